#!/bin/bash
# the contract path at N = 2 as a dry run on ONE GPU (gloo, both ranks on device 0; never for reported numbers): the sharded job must give the single-process hit digest
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6n
timeout 400 python bench.py --refs 600 --steps 2 --warmup 1 --cpu-sample 0 --no-workloads --no-host-leg > gpurun_out/r6n/n1.txt 2> gpurun_out/r6n/n1.err
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --refs 600 --steps 2 --warmup 1 --cpu-sample 0 --exchange-batch 128 > gpurun_out/r6n/n2.txt 2> gpurun_out/r6n/n2.err
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --comm capi --refs 600 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r6n/n2_capi.txt 2> gpurun_out/r6n/n2_capi.err
for f in n1 n2 n2_capi; do tail -1 gpurun_out/r6n/$f.txt | python3 -c "
import json,sys
t=sys.stdin.read().strip()
try:
    d=json.loads(t); print('$f', d['n_gpus'], d['scaling'], round(d['ms_per_step'],1), d['config']['hits'], d['extras'].get('hits_digest'), d['extras'].get('exchange'))
except Exception as e: print('$f', 'no line', e, t[:200])"; tail -3 gpurun_out/r6n/$f.err | cut -c1-300; done
