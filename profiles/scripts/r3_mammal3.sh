#!/bin/bash
# round 3: Gb-scale chunk walk over the position stream (no successor array): tests at 1 Gb, A/B on 8 x 3 Gb, 50 x 3 Gb
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_big.py -m gpu -x -q > gpurun_out/r3j/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3j/pytest.log
for wd in 1 0; do
  PSK_WALK_DIRECT=$wd timeout 900 python bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 > gpurun_out/r3j/mammalian_8x3Gb_wd$wd.json 2> gpurun_out/r3j/mammalian_8x3Gb_wd$wd.err; echo "8x wd=$wd rc=$?"
done
python - <<'PY'
import json
for f in ("mammalian_8x3Gb_wd1", "mammalian_8x3Gb_wd0"):
    d = json.loads(open(f"gpurun_out/r3j/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 1), d["config"]["hits"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()}, d["extras"].get("oracle_check", {}).get("result"), d["extras"]["chain_work_per_step"])
PY
timeout 1500 python bench.py --workload mammalian --refs 50 --stream > gpurun_out/r3j/mammalian_50x3Gb.json 2> gpurun_out/r3j/mammalian_50x3Gb.err; echo "50x rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3j/mammalian_50x3Gb.json").read().strip().splitlines()[-1])
print("50x", round(d["ms_per_step"], 1), d["config"]["hits"], d["extras"]["phases_s"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()}, d["extras"].get("oracle_check", {}).get("result"))
PY
profiles/scripts/prof.sh r3j_mammal8 --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0
