#!/bin/bash
# selected GPU tests: profiles/scripts/r6_tests_k.sh "<pytest -k expression>" [file]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6t
(time timeout 2400 python -m pytest ${2:-tests} -x -q -m gpu -k "$1") > gpurun_out/r6t/tests_k.txt 2>&1
tail -25 gpurun_out/r6t/tests_k.txt
