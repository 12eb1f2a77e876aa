#!/bin/bash
# the single-GPU emulation of a rank's share (extras.scaling_model) under variants: two lanes forced in every round (PSK_PIPELINE=1), larger exchange batches
cd "$GRAFT_REPO_ROOT" || exit 1
A="--refs 10000 --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-workloads --emulate-rank-of 2 4 8"
run() { tag=$1; shift; "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$tag', 'N=1', round(d['ms_per_step'],1), {n: (v['rank_ms'], v['speedup_vs_1']) for n, v in d['extras']['scaling_model']['ranks'].items()})"; }
run default timeout 400 python bench.py $A
PSK_PIPELINE=1 run pipeline1 timeout 400 python bench.py $A
run batch512 timeout 400 python bench.py $A --exchange-batch 512
run batch1024 timeout 400 python bench.py $A --exchange-batch 1024
