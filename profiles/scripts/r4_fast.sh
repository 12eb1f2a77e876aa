#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_small_query.py tests/test_database_gpu.py tests/test_gpu_learned.py tests/test_reference_goldens.py -m gpu -x -q > gpurun_out/r4t/pytest.txt 2>&1; tail -3 gpurun_out/r4t/pytest.txt
python3 bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 20000 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r4t/meta.out 2> gpurun_out/r4t/meta.err
python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_metagenome_*.json"))[-1]
d=json.load(open(f)); print(round(d["ms_per_step"],1), d["config"]["hits"], {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
print(json.dumps(d.get("api") or d.get("extras",{}).get("api") or [k for k in d])[:1500])
PY
