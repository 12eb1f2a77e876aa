#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4w
PSK_FUZZ_SEEDS=64 PSK_FUZZ_DB_SEEDS=48 timeout 2400 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_parity.py tests/test_gpu_small_query.py -m gpu -x -q > gpurun_out/r4w/pytest.txt 2>&1; tail -3 gpurun_out/r4w/pytest.txt
