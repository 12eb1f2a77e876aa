#!/bin/bash
# two batches in flight: the tests that force it, then the all-vs-all steps with and without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5p
timeout 1200 python -m pytest tests/test_gpu_scale_paths.py -x -q -m gpu 2>&1 | tail -4
for pl in 0 -1; do
  for n in 10000 1000; do
    if [ $pl = 0 ]; then export PSK_PIPELINE=0; else unset PSK_PIPELINE; fi
    timeout 600 python bench.py --workload allvsall --refs $n --steps 3 --warmup 2 --cpu-sample 0 --no-host-leg 2> gpurun_out/r5p/ava_${n}_$pl.err | tail -1 > gpurun_out/r5p/ava_${n}_$pl.json
    python - $n $pl <<'PY'
import json, sys
n, pl = sys.argv[1:3]
try:
    d = json.loads(open(f"gpurun_out/r5p/ava_{n}_{pl}.json").read())
    print("pipeline", pl, n, round(d["ms_per_step"], 1), d["extras"].get("hits_digest"), d.get("kernel_ms_per_step"))
except Exception as e:
    print("pipeline", pl, n, "failed", e)
PY
  done
done
