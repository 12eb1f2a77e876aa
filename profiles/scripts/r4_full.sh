#!/bin/bash
# round 4: the complete GPU suite and the default bench run (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4x
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r4x/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4x/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r4x/bench_default.out 2> gpurun_out/r4x/bench_default.err
tail -1 gpurun_out/r4x/bench_default.out > gpurun_out/r4x/bench_default_line.json
wc -c gpurun_out/r4x/bench_default_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4x/bench_default_line.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["extras"].get("host_packed_pairs_per_s"))
for k,v in d["extras"]["workloads"].items(): print(k, {a:b for a,b in v.items() if a in ("ms_per_step","value","hits","api_queries_per_s","api_queries_per_s_8_threads","oracle_check")}, v.get("roofline"))
PY
