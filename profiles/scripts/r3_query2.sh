#!/bin/bash
# round 3: per-contig Database.query work: tests, then the host-side split and the kernel timeline of one call, then the metagenome step's host tail
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3q
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_learned.py -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3q/pytest.log
python3 profiles/scripts/query_latency.py > gpurun_out/r3q/query_latency.txt 2>&1; tail -3 gpurun_out/r3q/query_latency.txt
rm -rf /tmp/q1; mkdir -p /tmp/q1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/q1 -o q -- python3 profiles/scripts/query_latency.py > /dev/null 2> gpurun_out/r3q/trace.err
python3 profiles/scripts/query_timeline.py /tmp/q1 > gpurun_out/r3q/query_timeline.txt 2>&1; head -60 gpurun_out/r3q/query_timeline.txt
python3 profiles/scripts/meta_tail.py > gpurun_out/r3q/meta_tail.txt 2>&1; tail -2 gpurun_out/r3q/meta_tail.txt
