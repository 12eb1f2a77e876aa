#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4g
NT=8 python3 profiles/scripts/query_threads8.py 2>&1 | tail -2
NT=1 python3 profiles/scripts/query_threads8.py 2>&1 | tail -1
rm -rf /tmp/q2; mkdir -p /tmp/q2
NT=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/q2 -o q -- python3 profiles/scripts/query_threads8.py > /dev/null 2> gpurun_out/r4g/trace.err
python3 profiles/scripts/trace_overlap.py /tmp/q2 | tee gpurun_out/r4g/overlap_8threads.txt
GPU_MAX_HW_QUEUES=8 NT=8 python3 profiles/scripts/query_threads8.py 2>&1 | tail -1
