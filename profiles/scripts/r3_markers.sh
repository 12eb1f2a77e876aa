#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3y
timeout 1800 python -m pytest tests/test_gpu_big.py tests/test_gpu_parity.py tests/test_gpu_ingest.py -m gpu -x -q 2>&1 | tail -3
python bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r3y/mammal8.json 2> gpurun_out/r3y/mammal8.err
python bench.py --workload mammalian --refs 50 --stream > gpurun_out/r3y/mammalian_50x3Gb.json 2> gpurun_out/r3y/mammalian_50x3Gb.err
for w in mammal8 mammalian_50x3Gb; do python - gpurun_out/r3y/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 1), d["config"].get("hits"), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
done
