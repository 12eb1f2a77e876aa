#!/bin/bash
# the complete GPU suite, smoke, the default bench run (round 6)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6s
(time timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/r6s/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r6s/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6s/smoke.txt 2>&1; tail -1 gpurun_out/r6s/smoke.txt
(time timeout 900 python bench.py > gpurun_out/r6s/bench_default_lines.txt 2> gpurun_out/r6s/bench_default.err)
tail -1 gpurun_out/r6s/bench_default_lines.txt > gpurun_out/r6s/bench_default_line.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r6s/bench_default_line.json"))
print("contract", d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["kernel_ms_per_step"] if "kernel_ms_per_step" in d else "", d["host_to_host"])
print(d["extras"].get("oracle_check"), d["extras"].get("scaling_model"))
for k, w in d["extras"]["workloads"].items(): print(k, w.get("ms_per_step"), w.get("value"), w.get("roofline"), w.get("oracle_check"))
PY
