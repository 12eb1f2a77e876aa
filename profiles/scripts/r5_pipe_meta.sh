#!/bin/bash
# two batches in flight on the metagenome rounds: tests that run contig rounds, then the step with and without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5p
PSK_PIPELINE=1 timeout 1200 python -m pytest tests/test_gpu_scale_paths.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4
for pl in 0 -1; do
    if [ $pl = 0 ]; then export PSK_PIPELINE=0; else unset PSK_PIPELINE; fi
    timeout 600 python bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 3 --warmup 2 --cpu-sample 0 2> gpurun_out/r5p/meta_$pl.err | tail -1 > gpurun_out/r5p/meta_$pl.json
    python - $pl <<'PY'
import json, sys
pl = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r5p/meta_{pl}.json").read())
    print("pipeline", pl, round(d["ms_per_step"], 1), d["extras"].get("hits"), d["config"].get("hits"), d.get("kernel_ms_per_step"))
    for k, v in d["extras"].items():
        if isinstance(v, dict) and "ms_per_step" in v: print("   ", k, round(v["ms_per_step"], 1), v.get("hits"))
except Exception as e:
    print("pipeline", pl, "failed", e)
PY
done
