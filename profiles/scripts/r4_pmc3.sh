#!/bin/bash
# round 4, after the DP pruning and the one-pass index join: counter passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs) of the all-vs-all and metagenome chain-stage
# kernels + the unit counts of the same runs, and the SQ counters of the lane DP (instructions per anchor before / after: profiles/r3/r3a_chain_lane20_counters.md)
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
profiles/scripts/pmc.sh r4_ava "anchor_join4|anchor_emit_pairs|chain_lane20|select_kernel" --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
profiles/scripts/pmc.sh r4_meta "gsi_join_kernel|gsi_prefilter|chain_quad_deep|chain_chunk_list|select_tiny|pair_build_rows" --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
for w in "allvsall --refs 1000 --steps 2" "metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2"; do
  tag=$(echo $w | cut -d' ' -f1)
  python bench.py --workload $w --warmup 1 --cpu-sample 0 > /dev/null 2> /dev/null
  cp $(ls -t gpurun_out/bench_full_${tag}_*.json | head -1) gpurun_out/pmc/r4_units_$tag.json
done
profiles/scripts/pmc_sq.sh r4_lane_sq "chain_lane20|chain_quad_deep" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
profiles/scripts/pmc_sq.sh r4_quad_sq "chain_lane20|chain_quad_deep" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" -- --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
ls gpurun_out/pmc | head -40; cat gpurun_out/pmc/r4_lane_sq.sq.txt gpurun_out/pmc/r4_quad_sq.sq.txt
