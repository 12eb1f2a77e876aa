#!/bin/bash
# scratch growth during the contract job's steps (PSK_TRACE_ALLOC=1), with the steps' own times
cd "$GRAFT_REPO_ROOT" || exit 1
PSK_TRACE_ALLOC=1 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-host-leg --no-workloads 2> gpurun_out/alloc_trace.err | tail -1 > /dev/null
grep -c "psk alloc" gpurun_out/alloc_trace.err; grep "psk alloc" gpurun_out/alloc_trace.err | tail -30
python3 -c "
import json,glob
f=sorted(glob.glob('gpurun_out/bench_full_allvsall_*.json'))[-1]; d=json.load(open(f)); print(round(d['ms_per_step'],1), d['extras']['step_ms'])"
