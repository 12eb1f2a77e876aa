#!/bin/bash
# Gb-scale path after a change: the tests that run it, then the mammalian step with its two oracle-checked pairs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5m
timeout 900 python -m pytest tests/test_gpu_big.py tests/test_gpu_parity.py -x -q -m gpu -k "big or item_hops or alternative_device or Gb or gb" 2>&1 | tail -5
PSK_FUZZ_SEEDS=${FUZZ:-32} timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k random_pairs 2>&1 | tail -3
timeout 600 python bench.py --workload mammalian --steps 2 --warmup 1 > gpurun_out/r5m/mammal_check.json 2> gpurun_out/r5m/mammal_check.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5m/mammal_check.json").read().strip().splitlines()[-1])
full = json.load(open(d["full"])) if d.get("full") else d
k = full.get("kernel_roofline") or {}
print(round(d["ms_per_step"], 2), {n: round(v.get("ms_per_step", 0), 1) for n, v in k.items()})
print(json.dumps(full.get("oracle_check"))[:600])
PY
tail -2 gpurun_out/r5m/mammal_check.err
