#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for fs in "" "--faster-small"; do python3 bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0 $fs > /dev/null 2> gpurun_out/meta.err; tail -2 gpurun_out/meta.err | cut -c1-300; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_metagenome_*.json"))[-1]
d=json.load(open(f)); print("$fs", round(d["ms_per_step"],1), d["config"]["hits"], {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
done
