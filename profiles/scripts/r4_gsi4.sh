#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4s
PSK_FUZZ_SEEDS=48 PSK_FUZZ_DB_SEEDS=40 timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4s/pytest.txt 2>&1; tail -3 gpurun_out/r4s/pytest.txt
for fs in "" "--faster-small"; do python3 bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0 $fs > /dev/null 2> gpurun_out/r4s/meta.err; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_metagenome_*.json"))[-1]
d=json.load(open(f)); print("$fs", round(d["ms_per_step"],1), d["config"]["hits"], {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
done
