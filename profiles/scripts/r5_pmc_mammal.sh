#!/bin/bash
# the Gb-scale kernels' counters again after the round's changes (batches of 2^28, chunk walk over item offsets): profiles/scripts/pmc_summary_r5.py mammalian
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
T="timeout 600"
$T profiles/scripts/pmc.sh r5_mammal "anchor_join4|anchor_emit_expand|chunk_hops|chain_lane20x|chain_chunk_list|select_huge|select_big|pair_reduce" --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0
$T python3 bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> /dev/null
cp $(ls -t gpurun_out/bench_full_mammalian_*.json | head -1) gpurun_out/pmc/r5_units_mammalian.json
ls -la gpurun_out/pmc; head -12 gpurun_out/pmc/r5_mammal.FETCH_SIZE.txt gpurun_out/pmc/r5_mammal.WRITE_SIZE.txt; tail -3 gpurun_out/pmc/r5_mammal.FETCH_SIZE.err
