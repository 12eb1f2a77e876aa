#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3u
timeout 1800 python -m pytest tests/test_gpu_big.py tests/test_gpu_parity.py tests/test_gpu_scale_paths.py -m gpu -x -q 2>&1 | tail -3
python bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r3u/mammal8.json 2> gpurun_out/r3u/mammal8.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3u/mammal8.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 2), d["config"].get("hits"), {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()}, {k: v for k, v in d.get("extras", {}).items() if "oracle" in k})
PY
