#!/bin/bash
# round 3: counter passes again for the kernels that changed after the first collection (all-vs-all emit, metagenome join and DP)
cd "$GRAFT_REPO_ROOT"
profiles/scripts/pmc.sh r3_ava "anchor_join4|anchor_emit_pairs|chain_lane20|select_kernel" --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
profiles/scripts/pmc.sh r3_meta "anchor_join_probe|chain_quad_deep|chain_chunk_list|chunk_heads|anchor_emit_packed4|pref_count|select_kernel" --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
for w in "allvsall --refs 1000" "metagenome --refs 5000 --queries 100000 --api-queries 0"; do
  tag=$(echo $w | cut -d' ' -f1)
  python bench.py --workload $w --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/pmc/r3_units_$tag.json 2> /dev/null
done
profiles/scripts/pmc_sq.sh r3z_lane "chain_lane20_kernel" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
cat gpurun_out/pmc/r3z_lane.sq.txt
