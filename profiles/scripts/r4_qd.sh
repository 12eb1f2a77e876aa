#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4o
PSK_FUZZ_SEEDS=200 PSK_FUZZ_DB_SEEDS=60 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_parity.py -m gpu -x -q -n 4 > gpurun_out/r4o/pytest.txt 2>&1; grep -n "FAILED\|Error\|assert" gpurun_out/r4o/pytest.txt | head -20; tail -3 gpurun_out/r4o/pytest.txt
