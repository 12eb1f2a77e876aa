#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3z
start=$(date +%s)
timeout 1750 python bench.py > gpurun_out/r3z/bench_default.json 2> gpurun_out/r3z/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - start )) s" | tee -a gpurun_out/r3z/bench_default.err
tail -3 gpurun_out/r3z/bench_default.err
