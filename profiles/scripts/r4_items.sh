#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4p
timeout 1500 python -m pytest tests/test_gpu_parity.py::test_alternative_device_paths_agree tests/test_gpu_big.py -m gpu -x -q > gpurun_out/r4p/pytest.txt 2>&1; tail -4 gpurun_out/r4p/pytest.txt
for v in 1 0; do PSK_HOPS_ITEMS=$v python3 bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 2 > /dev/null 2> gpurun_out/r4p/mammal_$v.err; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_mammalian_*.json"))[-1]
d=json.load(open(f)); print("PSK_HOPS_ITEMS=$v", round(d["ms_per_step"],1), d["config"]["hits"], d["extras"].get("oracle_check",{}).get("result"), {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
done
