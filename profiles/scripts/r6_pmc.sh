#!/bin/bash
# round 6: counter passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs) of every kernel that holds >= 3 % of a workload's step, the unit counts of the same runs,
# and the sketch scan kernels' counters again (their body became a template in round 4). Summarised by profiles/scripts/pmc_summary_r6.py -> profiles/r6/pmc_kernels.json
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
T="timeout 500"
$T profiles/scripts/pmc.sh r6_search "sketch_scan|sketch_emit" --workload search --steps 2 --warmup 1 --cpu-sample 0 --no-api
$T profiles/scripts/pmc_sq.sh r6_search_sq "sketch_scan" "SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_WAVES" -- --workload search --steps 2 --warmup 1 --cpu-sample 0 --no-api
$T profiles/scripts/pmc.sh r6_ava "gsl_walk|gsl_heads|chain_lane20|select_kernel|chunk_seeds|pair_reduce" --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
$T profiles/scripts/pmc.sh r6_meta "gsi_join_kernel|gsi_prefilter|chain_quad_deep|chain_chunk_list|select_tiny|select_kernel|pair_reduce|pair_build_rows" --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
$T profiles/scripts/pmc.sh r6_mammal "anchor_join4|anchor_emit_expand|item_next|chunk_hops|chain_lane20x|chain_chunk_list|select_huge|select_big|radix_sort|index_bucket|index_gather|pair_reduce" --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0
for w in "allvsall --refs 1000 --steps 2" "metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2" "mammalian --refs 8 --steps 2"; do
  tag=$(echo $w | cut -d' ' -f1)
  $T python3 bench.py --workload $w --warmup 1 --cpu-sample 0 > /dev/null 2> /dev/null
  cp $(ls -t gpurun_out/bench_full_${tag}_*.json | head -1) gpurun_out/pmc/r6_units_$tag.json
done
$T python3 bench.py --workload search --steps 2 --warmup 1 --cpu-sample 0 --no-api > /dev/null 2> /dev/null
cp $(ls -t gpurun_out/bench_full_search_*.json | head -1) gpurun_out/pmc/r6_units_search.json
ls gpurun_out/pmc | head -60
