#!/bin/bash
# mammalian 8 x 3 Gb step against the batch size (query seeds per batch): PSK_BATCH_ITEMS_LOG2 = 27 (default for Gb-scale queries), 28, 29
mkdir -p gpurun_out/r5m
for lg in ${LOGS:-27 28 29}; do
  PSK_BATCH_ITEMS_LOG2=$lg timeout 600 python bench.py --workload mammalian --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r5m/mammal_$lg.json 2> gpurun_out/r5m/mammal_$lg.err
  python - "$lg" <<'PY'
import json, sys
lg = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r5m/mammal_{lg}.json").read().strip().splitlines()[-1])
    full = json.load(open(d["full"])) if d.get("full") else d
    k = full.get("kernel_roofline") or full.get("extras", {}).get("kernel_roofline") or {}
    print(lg, round(d["ms_per_step"], 2), {n: round(v.get("ms_per_step", 0), 1) for n, v in k.items()} if isinstance(k, dict) else "")
except Exception as e:
    print(lg, "failed", e)
PY
  tail -2 gpurun_out/r5m/mammal_$lg.err
done
