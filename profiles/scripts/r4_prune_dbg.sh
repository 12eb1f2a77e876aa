#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
PSK_DP_PRUNE=5 python3 bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep "^vote " | head -40
python3 bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> gpurun_out/dbg.err; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_metagenome_*.json"))[-1]
d=json.load(open(f)); print(round(d["ms_per_step"],1), d["config"]["hits"], {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
