#!/bin/bash
# counter passes (FETCH_SIZE, WRITE_SIZE) of the 10 000 x 10 000 step itself: the walks, the heads kernel, the lane DP, the selection (1 timed step + the one-chain table step under the counters)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc
A="--refs 10000 --steps 1 --warmup 0 --cpu-sample 0 --no-host-leg --no-workloads"
timeout 1500 profiles/scripts/pmc.sh r6_ava10k "gsl_walk|gsl_heads|chain_lane20|select_kernel|chunk_seeds" $A
timeout 400 python3 bench.py $A > /dev/null 2> /dev/null
cp $(ls -t gpurun_out/bench_full_allvsall_*.json | head -1) gpurun_out/pmc/r6_units_allvsall10k.json
head -5 gpurun_out/pmc/r6_ava10k.FETCH_SIZE.txt gpurun_out/pmc/r6_ava10k.WRITE_SIZE.txt; tail -3 gpurun_out/pmc/r6_ava10k.FETCH_SIZE.err
