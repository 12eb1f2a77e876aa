#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3q
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_database_gpu.py tests/test_gpu_learned.py tests/test_gpu_scale_paths.py -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3q/pytest.log
python3 profiles/scripts/query_latency.py > gpurun_out/r3q/query_latency.txt 2>&1; tail -3 gpurun_out/r3q/query_latency.txt
rm -rf /tmp/q1; mkdir -p /tmp/q1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/q1 -o q -- python3 profiles/scripts/query_latency.py > /dev/null 2> gpurun_out/r3q/trace.err
python3 profiles/scripts/query_timeline.py /tmp/q1 > gpurun_out/r3q/query_timeline.txt 2>&1; head -45 gpurun_out/r3q/query_timeline.txt
python3 profiles/scripts/query_threads.py > gpurun_out/r3q/threads_default.txt 2>&1; tail -2 gpurun_out/r3q/threads_default.txt
