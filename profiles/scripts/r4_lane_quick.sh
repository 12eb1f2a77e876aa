#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4v
PSK_FUZZ_SEEDS=48 PSK_FUZZ_DB_SEEDS=40 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4v/pytest.txt 2>&1; tail -3 gpurun_out/r4v/pytest.txt
for n in 1000 10000; do python3 bench.py --workload allvsall --refs $n --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> gpurun_out/ava.err; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_allvsall_*.json"))[-1]
d=json.load(open(f)); print("n $n", round(d["ms_per_step"],1), d["config"].get("hits"), {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
done
