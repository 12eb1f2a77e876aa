#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3t
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_big.py -m gpu -x -q > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3t/pytest.log
python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3t/ava1000.json 2> gpurun_out/r3t/ava1000.err
python bench.py --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/r3t/ava10k.json 2> gpurun_out/r3t/ava10k.err
for w in ava1000 ava10k; do python - gpurun_out/r3t/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("hits"), d.get("extras", {}).get("hits_digest"), {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()})
PY
done
