#!/bin/bash
# the complete GPU suite + smoke (round 6)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6s
(time timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/r6s/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r6s/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6s/smoke.txt 2>&1; tail -2 gpurun_out/r6s/smoke.txt
