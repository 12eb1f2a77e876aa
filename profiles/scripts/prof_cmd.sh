#!/bin/bash
# Usage (GPU box, repo root): profiles/scripts/prof_cmd.sh <tag> <python script> [args...]   -> gpurun_out/prof/<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof /tmp/prof/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/$tag -o $tag -- python3 "$@" > gpurun_out/prof/$tag.log 2> gpurun_out/prof/$tag.err
for f in $(find /tmp/prof/$tag -name "*kernel_stats.csv"); do cp "$f" gpurun_out/prof/${tag}_kernel_stats.csv; done
