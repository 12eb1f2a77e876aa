#!/bin/bash
# kernel table of the metagenome step (rocprofv3 --kernel-trace --stats) over 1 warm-up + 4 timed steps: which launches are per step and which are one-time (index builds)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6p
timeout 500 bash profiles/scripts/prof.sh r6p_meta --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r6p/prof_meta.log 2>&1
python3 profiles/summarize.py gpurun_out/prof/r6p_meta_kernel_stats.csv 5 > gpurun_out/r6p/kernel_stats_meta.md
head -45 gpurun_out/r6p/kernel_stats_meta.md | cut -c1-120
tail -1 gpurun_out/prof/r6p_meta.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'])"
