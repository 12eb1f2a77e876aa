#!/bin/bash
# round 3: Gb-scale pairs after the single-walk chunk table: tests at 1 Gb, the 8 x 3 Gb line, the 50 x 3 Gb line, kernel stats of the 8 x 3 Gb run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_big.py -m gpu -x -q > gpurun_out/r3e/pytest_big.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3e/pytest_big.log
timeout 900 python bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 > gpurun_out/r3e/mammalian_8x3Gb.json 2> gpurun_out/r3e/mammalian_8x3Gb.err; echo "8x rc=$?"
timeout 1500 python bench.py --workload mammalian --refs 50 --stream --cpu-sample 0 > gpurun_out/r3e/mammalian_50x3Gb.json 2> gpurun_out/r3e/mammalian_50x3Gb.err; echo "50x rc=$?"
python - <<'PY'
import json
for f in ("mammalian_8x3Gb", "mammalian_50x3Gb"):
    d = json.loads(open(f"gpurun_out/r3e/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 1), d["config"]["hits"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()}, d["extras"].get("oracle_check", {}).get("result"))
PY
profiles/scripts/prof.sh r3e_mammal8 --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0
