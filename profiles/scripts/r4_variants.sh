#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4l
for v in contigs sv contigs sv plain; do
  python3 bench.py --workload allvsall --refs 1000 --variant $v --steps 2 --warmup 1 --cpu-sample 8 2> gpurun_out/r4l/$v.err | tail -1 > gpurun_out/r4l/line_$v.json
  python3 - <<PY
import json
d=json.load(open("gpurun_out/r4l/line_$v.json"))
print("$v", round(d["ms_per_step"],1), d["config"]["hits"], d["extras"].get("hits_digest"), d.get("roofline",{}).get("kernel"), round(d.get("roofline",{}).get("frac",0),3))
PY
done
grep -h "oracle_check" gpurun_out/bench_full_allvsall_*.json | tail -2 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_learned.py tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -2
