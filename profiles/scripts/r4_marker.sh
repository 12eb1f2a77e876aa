#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4z
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py tests/test_gpu_ingest.py -m gpu -x -q > gpurun_out/r4z/pytest_marker.txt 2>&1; tail -4 gpurun_out/r4z/pytest_marker.txt
python3 bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r4z/mammal.out 2> gpurun_out/r4z/mammal.err; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_mammalian_*.json"))[-1]
d=json.load(open(f)); print(round(d["ms_per_step"],1), d["config"].get("hits"), {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
