#!/bin/bash
# the contract job's timed steps one by one (extras.step_ms), three runs
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-host-leg --no-workloads 2>/dev/null | tail -1 > /dev/null
  python3 -c "
import json,glob
f=sorted(glob.glob('gpurun_out/bench_full_allvsall_*.json'))[-1]; d=json.load(open(f)); print(round(d['ms_per_step'],1), d['extras']['step_ms'])"
done
