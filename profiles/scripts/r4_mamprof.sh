#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash profiles/scripts/prof.sh r4q_mammal --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4q_mammal_kernel_stats.csv 3 > gpurun_out/prof/r4q_kernel_stats_mammal.md; head -34 gpurun_out/prof/r4q_kernel_stats_mammal.md
