#!/bin/bash
# round 4: the one-launch-sequence query - its own tests, the suites that call Database.query, latency, timeline, threads
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_small_query.py -x -q > gpurun_out/r4b/pytest_small.txt 2>&1
tail -25 gpurun_out/r4b/pytest_small.txt
timeout 300 python profiles/scripts/query_latency.py > gpurun_out/r4b/query_latency.txt 2>&1
tail -5 gpurun_out/r4b/query_latency.txt
rm -rf /tmp/q1; mkdir -p /tmp/q1
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/q1 -o q -- python3 profiles/scripts/query_latency.py > /dev/null 2> gpurun_out/r4b/trace.err
python3 profiles/scripts/query_timeline.py /tmp/q1 > gpurun_out/r4b/query_timeline.txt 2>&1; head -40 gpurun_out/r4b/query_timeline.txt
NQ=4000 timeout 600 python3 profiles/scripts/query_threads.py > gpurun_out/r4b/query_threads.txt 2>&1; tail -3 gpurun_out/r4b/query_threads.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_database_gpu.py tests/test_gpu_learned.py -m gpu -x -q > gpurun_out/r4b/pytest_more.txt 2>&1; tail -5 gpurun_out/r4b/pytest_more.txt
