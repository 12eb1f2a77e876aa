#!/bin/bash
# round 5: the slice join on the 1000 x 1000 all-vs-all step: kernel table (staged whole-line stores / plain 16-byte stores) and HBM counters of its three kernels
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r5b}
bash profiles/scripts/prof.sh ${tag}_ava --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/${tag}_ava_kernel_stats.csv 3 > gpurun_out/prof/${tag}_kernel_stats_ava.md
PSK_GSL_STAGE=0 bash profiles/scripts/prof.sh ${tag}_ava_nostage --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/${tag}_ava_nostage_kernel_stats.csv 3 > gpurun_out/prof/${tag}_kernel_stats_ava_nostage.md
bash profiles/scripts/pmc.sh ${tag}_gsl "gsl_" --workload allvsall --refs 1000 --steps 1 --warmup 1 --cpu-sample 0
PSK_GSL_STAGE=0 bash profiles/scripts/pmc.sh ${tag}_gsl_nostage "gsl_" --workload allvsall --refs 1000 --steps 1 --warmup 1 --cpu-sample 0
for w in ava ava_nostage; do echo "== $w"; head -14 gpurun_out/prof/${tag}_kernel_stats_$w.md | cut -c1-160; done
for f in gpurun_out/pmc/${tag}_gsl*.txt; do echo "== $f"; cat $f; done
