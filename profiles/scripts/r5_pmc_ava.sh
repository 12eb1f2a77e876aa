#!/bin/bash
# the all-vs-all kernels' counters again at the end of the round (selection without its atomics, one-wave reduction, lane DP with three near predecessors): pmc_summary_r5.py allvsall
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
export PSK_PIPELINE=0
T="timeout 600"
$T profiles/scripts/pmc.sh r5_ava "gsl_walk|gsl_heads|chain_lane20|select_kernel|chunk_seeds|pair_reduce" --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
$T python3 bench.py --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> /dev/null
cp $(ls -t gpurun_out/bench_full_allvsall_*.json | head -1) gpurun_out/pmc/r5_units_allvsall.json
head -8 gpurun_out/pmc/r5_ava.FETCH_SIZE.txt gpurun_out/pmc/r5_ava.WRITE_SIZE.txt
