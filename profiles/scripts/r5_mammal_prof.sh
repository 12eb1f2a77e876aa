#!/bin/bash
# kernel table of the mammalian 8 x 3 Gb step (rocprofv3 --kernel-trace --stats)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5m
timeout 400 bash profiles/scripts/prof.sh r5m_mammal --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r5m/prof.log 2>&1
python3 profiles/summarize.py gpurun_out/prof/r5m_mammal_kernel_stats.csv 3 > gpurun_out/r5m/kernel_stats_mammal.md
head -32 gpurun_out/r5m/kernel_stats_mammal.md | cut -c1-110
tail -1 gpurun_out/r5m/prof.log | cut -c1-300
