#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4u
python3 profiles/scripts/meta_tail.py > gpurun_out/r4u/meta_tail.txt 2>&1; tail -2 gpurun_out/r4u/meta_tail.txt
PSK_HIT_THREADS=1 python3 profiles/scripts/meta_tail.py > gpurun_out/r4u/meta_tail_1thread.txt 2>&1; tail -1 gpurun_out/r4u/meta_tail_1thread.txt
bash profiles/scripts/r4_gaps_meta.sh | head -12
