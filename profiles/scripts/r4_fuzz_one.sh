#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4y
PSK_FUZZ_SEEDS=420 timeout 600 python -m pytest "tests/test_gpu_fuzz.py::test_random_pairs_match_oracle[257]" "tests/test_gpu_fuzz.py::test_random_pairs_match_oracle[369]" "tests/test_gpu_fuzz.py::test_random_pairs_match_oracle[417]" -m gpu -q -s > gpurun_out/r4y/fuzz_one.txt 2>&1; tail -30 gpurun_out/r4y/fuzz_one.txt | cut -c1-300
