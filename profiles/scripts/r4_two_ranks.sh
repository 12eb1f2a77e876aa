#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4x
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 --no-workloads --no-api --cpu-sample 0 > gpurun_out/r4x/two_ranks.out 2> gpurun_out/r4x/two_ranks.err; echo rc $?
tail -1 gpurun_out/r4x/two_ranks.out | cut -c1-700
timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --workload allvsall --refs 400 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r4x/two_ranks_ava.out 2> gpurun_out/r4x/two_ranks_ava.err; echo rc $?
tail -1 gpurun_out/r4x/two_ranks_ava.out | cut -c1-900
timeout 600 python bench.py --workload allvsall --refs 400 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('single', d['config'].get('hits'), d['config'].get('hits_digest'), d['n_gpus'])"
