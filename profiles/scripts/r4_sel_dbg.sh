#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python3 bench.py --workload mammalian --refs 4 --steps 1 --warmup 0 --cpu-sample 0 2>gpurun_out/sel.err | grep "^pair " | head -40
tail -3 gpurun_out/sel.err | cut -c1-200
