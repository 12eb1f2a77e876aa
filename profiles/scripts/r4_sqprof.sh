#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4e
timeout 600 python -m pytest tests/test_gpu_small_query.py -x -q 2>&1 | tail -3
PSK_SQ_TEAM=0 timeout 600 python -m pytest tests/test_gpu_small_query.py -x -q 2>&1 | tail -1
PSK_SQ_PROFILE=1 timeout 300 python profiles/scripts/query_latency.py 2>&1 | tail -4 > gpurun_out/r4e/sq_profile.txt; cat gpurun_out/r4e/sq_profile.txt
PSK_SQ_PROFILE=1 PSK_SQ_TEAM=0 timeout 300 python profiles/scripts/query_latency.py 2>&1 | tail -2
rm -rf /tmp/q1; mkdir -p /tmp/q1
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/q1 -o q -- python3 profiles/scripts/query_latency.py > /dev/null 2> gpurun_out/r4e/trace.err
python3 profiles/scripts/query_timeline.py /tmp/q1 > gpurun_out/r4e/query_timeline.txt 2>&1; head -12 gpurun_out/r4e/query_timeline.txt
NQ=4000 timeout 600 python3 profiles/scripts/query_threads.py > gpurun_out/r4e/query_threads.txt 2>&1; tail -3 gpurun_out/r4e/query_threads.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
