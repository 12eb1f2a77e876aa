#!/bin/bash
# the mammalian kernel table (rocprofv3 --kernel-trace --stats) and the bench's own HIP-event brackets FROM THE SAME RUN, and the brackets of a run without the profiler
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6m
timeout 400 python bench.py --workload mammalian --refs 8 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r6m/plain.json 2> gpurun_out/r6m/plain.err
timeout 500 bash profiles/scripts/prof.sh r6m_mammal --workload mammalian --refs 8 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r6m/prof.log 2>&1
python3 profiles/summarize.py gpurun_out/prof/r6m_mammal_kernel_stats.csv 4 > gpurun_out/r6m/kernel_stats_mammal.md
head -16 gpurun_out/r6m/kernel_stats_mammal.md | cut -c1-110
python3 - <<'PY'
import json
for tag, p in (("plain", "gpurun_out/r6m/plain.json"), ("under rocprofv3", "gpurun_out/prof/r6m_mammal.json")):
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print(tag, "ms_per_step", d["ms_per_step"], d.get("kernel_ms_per_step"))
PY
