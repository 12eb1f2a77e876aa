#!/bin/bash
# round 3: long randomised parity sweep over the paths added this round (register-window DP, rows by length, one-synchronisation sketch), then the default bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3s
PSK_FUZZ_SEEDS=400 PSK_FUZZ_DB_SEEDS=80 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3s/fuzz_sweep.log 2>&1; echo "sweep rc=$?"; tail -4 gpurun_out/r3s/fuzz_sweep.log
start=$(date +%s)
timeout 1750 python bench.py > gpurun_out/r3s/bench_default.json 2> gpurun_out/r3s/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - start )) s" | tee -a gpurun_out/r3s/bench_default.err
python profiles/scripts/show_bench.py gpurun_out/r3s/bench_default.json | head -4
