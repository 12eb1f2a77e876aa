#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash profiles/scripts/prof.sh r4j_meta --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4j_meta_kernel_stats.csv 3 > gpurun_out/prof/r4j_kernel_stats_meta.md; head -24 gpurun_out/prof/r4j_kernel_stats_meta.md
