#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4y
PSK_FUZZ_SEEDS=600 PSK_FUZZ_DB_SEEDS=150 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -n 4 > gpurun_out/r4y/fuzz_sweep.txt 2>&1; tail -5 gpurun_out/r4y/fuzz_sweep.txt
python - <<'PY'
import ctypes as C, pyskani_amd
PY
