#!/bin/bash
# round 5: the 10 000 x 10 000 step: kernel table and HBM counters of the slice join's kernels
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r5e}
timeout 400 bash profiles/scripts/prof.sh ${tag}_ava10k --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/${tag}_ava10k_kernel_stats.csv 2 > gpurun_out/prof/${tag}_kernel_stats_ava10k.md
timeout 600 bash profiles/scripts/pmc.sh ${tag}_gsl10k "gsl_" --workload allvsall --refs 10000 --steps 1 --warmup 0 --cpu-sample 0
head -16 gpurun_out/prof/${tag}_kernel_stats_ava10k.md | cut -c1-160
for f in gpurun_out/pmc/${tag}_gsl10k*.txt; do echo "== $f"; cat $f; done
