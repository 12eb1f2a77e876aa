#!/bin/bash
# end of round 4: BASELINE configs[4] at full count on one GPU (50 x 3 Gb, ASCII streamed), with kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
start=$(date +%s)
timeout 2400 python bench.py --workload mammalian --refs 50 --stream --cpu-sample 0 > gpurun_out/r4x/mammalian_50x3Gb.json 2> gpurun_out/r4x/mammalian_50x3Gb.err; echo "rc=$? wall=$(( $(date +%s) - start )) s"
tail -3 gpurun_out/r4x/mammalian_50x3Gb.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4x/mammalian_50x3Gb.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"]["hits"], d["extras"].get("phases_s"), d["extras"].get("resident"))
print({k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
print(d["extras"].get("oracle_check"))
PY
