#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
PSK_DP_PRUNE=4 timeout 300 python3 bench.py --workload allvsall --refs 1000 --steps 1 --warmup 0 --cpu-sample 0 2>gpurun_out/dbg1.err | grep "^vote " | head -50
