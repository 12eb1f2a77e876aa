#!/bin/bash
# end of round 4: rocprofv3 --kernel-trace --stats summaries of the four workloads (profiles/r4/r4x_kernel_stats_*.md)
cd "$GRAFT_REPO_ROOT" || exit 1
bash profiles/scripts/prof.sh r4x_search --steps 10 --warmup 2 --cpu-sample 0 --no-api --no-workloads
python3 profiles/summarize.py gpurun_out/prof/r4x_search_kernel_stats.csv 12 > gpurun_out/prof/r4x_kernel_stats_search.md
bash profiles/scripts/prof.sh r4x_ava --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4x_ava_kernel_stats.csv 3 > gpurun_out/prof/r4x_kernel_stats_ava.md
bash profiles/scripts/prof.sh r4x_meta --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4x_meta_kernel_stats.csv 3 > gpurun_out/prof/r4x_kernel_stats_meta.md
bash profiles/scripts/prof.sh r4x_mammal --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4x_mammal_kernel_stats.csv 3 > gpurun_out/prof/r4x_kernel_stats_mammal.md
for w in search ava meta mammal; do echo "== $w"; head -12 gpurun_out/prof/r4x_kernel_stats_$w.md | cut -c1-150; done
