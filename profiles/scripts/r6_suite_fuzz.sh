#!/bin/bash
# the complete GPU suite, then the wide fuzz sweep (320 pair seeds + 48 database seeds under their forced kernel variants)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6s
(time timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/r6s/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r6s/pytest_gpu.txt
(time PSK_FUZZ_SEEDS=320 PSK_FUZZ_DB_SEEDS=48 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu) > gpurun_out/r6s/fuzz_sweep_368.txt 2>&1
tail -6 gpurun_out/r6s/fuzz_sweep_368.txt
