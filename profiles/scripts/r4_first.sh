#!/bin/bash
# round 4, first call: the compact bench line, the self-spawned 2-rank dry run, the GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4a
python bench.py > gpurun_out/r4a/bench_default.out 2> gpurun_out/r4a/bench_default.err
tail -c 4200 gpurun_out/r4a/bench_default.out | tail -1 > gpurun_out/r4a/bench_default_line.json
python bench.py --gpus 2 --backend gloo --share-gpu --workload allvsall --refs 1000 --steps 3 --warmup 1 > gpurun_out/r4a/bench_spawn2.out 2> gpurun_out/r4a/bench_spawn2.err
tail -1 gpurun_out/r4a/bench_spawn2.out
python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r4a/bench_ava1000.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r4a/pytest_gpu.txt
wc -c gpurun_out/r4a/bench_default_line.json
