#!/bin/bash
# default bench run (what the driver runs), the compact line kept
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4c
python bench.py > gpurun_out/r4c/bench_default.out 2> gpurun_out/r4c/bench_default.err
tail -1 gpurun_out/r4c/bench_default.out > gpurun_out/r4c/bench_default_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c/bench_default_line.json"))
print(d["ms_per_step"], d["value"], d["extras"]["workloads"]["metagenome_api"])
PY
