#!/bin/bash
# round 3: Gb-scale pairs after (a) run starts in the packed records, (b) sixteen tree slots in the lane DP: tests, 8 x 3 Gb, kernel stats, 50 x 3 Gb
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3f
timeout 1500 python -m pytest tests/test_gpu_big.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3f/pytest.log
for xt in 1; do
  PSK_LANE_XTREES=$xt timeout 900 python bench.py --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r3f/mammalian_8x3Gb_xt$xt.json 2> gpurun_out/r3f/mammalian_8x3Gb_xt$xt.err; echo "8x xt=$xt rc=$?"
done
python - <<'PY'
import json
for f in ("mammalian_8x3Gb_xt1", "mammalian_8x3Gb_xt0"):
    d = json.loads(open(f"gpurun_out/r3f/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 1), d["config"]["hits"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
profiles/scripts/prof.sh r3f_mammal8 --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0
timeout 1500 python bench.py --workload mammalian --refs 50 --stream > gpurun_out/r3f/mammalian_50x3Gb.json 2> gpurun_out/r3f/mammalian_50x3Gb.err; echo "50x rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3f/mammalian_50x3Gb.json").read().strip().splitlines()[-1])
print("50x", round(d["ms_per_step"], 1), d["config"]["hits"], d["extras"]["phases_s"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()}, d["extras"].get("oracle_check", {}).get("result"))
PY
