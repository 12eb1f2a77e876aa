#!/bin/bash
# round 3: where one Database.query(name, contig) spends its time: host-side split, then the kernel timeline of one call
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3q
python3 profiles/scripts/query_latency.py > gpurun_out/r3q/query_latency.txt 2>&1; tail -3 gpurun_out/r3q/query_latency.txt
rm -rf /tmp/q1; mkdir -p /tmp/q1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/q1 -o q -- python3 profiles/scripts/query_latency.py > /dev/null 2> gpurun_out/r3q/trace.err
python3 profiles/scripts/query_timeline.py /tmp/q1 > gpurun_out/r3q/query_timeline.txt 2>&1; head -70 gpurun_out/r3q/query_timeline.txt
