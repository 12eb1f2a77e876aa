#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4f
python3 bench.py --workload metagenome --refs 5000 --queries 10000 --api-queries 4000 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r4f/meta10k_api.out 2> gpurun_out/r4f/meta10k_api.err
python3 - <<'PY'
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_metagenome_*.json"))[-1]
d=json.load(open(f))
print(round(d["ms_per_step"], 1), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["extras"].get("api", {}).items() if k != "note"})
PY
