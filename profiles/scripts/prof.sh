#!/bin/bash
# Usage (on the GPU box, from the repo root): profiles/scripts/prof.sh <tag> <bench.py args...>
# rocprofv3 kernel trace + stats of one bench.py invocation; only the kernel_stats CSV is kept (gpurun_out/<tag>_kernel_stats.csv).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof /tmp/prof/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/$tag -o $tag -- python3 bench.py "$@" > gpurun_out/prof/$tag.json 2> gpurun_out/prof/$tag.err
for f in $(find /tmp/prof/$tag -name "*kernel_stats.csv"); do cp "$f" gpurun_out/prof/${tag}_kernel_stats.csv; done
ls /tmp/prof/$tag/* | head -5 >> gpurun_out/prof/$tag.err
