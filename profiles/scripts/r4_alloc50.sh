#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
PSK_TRACE_ALLOC=1 PSK_TRACE_BATCH=1 timeout 2400 python bench.py --workload mammalian --refs 50 --stream --cpu-sample 0 > gpurun_out/r4x/m50.json 2> gpurun_out/r4x/m50_alloc.err
grep -c "psk alloc" gpurun_out/r4x/m50_alloc.err; grep "psk alloc" gpurun_out/r4x/m50_alloc.err | sort -t' ' -k1,1 | awk '{print}' | cut -c1-200 | sort -k6 -n -r 2>/dev/null | head -40
grep "psk batch" gpurun_out/r4x/m50_alloc.err | head -30 | cut -c1-200
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r4x/m50.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], sum(d["kernel_ms_per_step"].values()))
PY
