#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash profiles/scripts/prof.sh r4x_meta --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4x_meta_kernel_stats.csv 3 > gpurun_out/prof/r4x_kernel_stats_meta.md
head -16 gpurun_out/prof/r4x_kernel_stats_meta.md | cut -c1-150
