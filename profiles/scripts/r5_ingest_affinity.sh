#!/bin/bash
# host-memory search (1 query vs 1 000 x 5 Mb from ASCII in host memory): packing threads x CPU affinity of the process
cd "$GRAFT_REPO_ROOT" || exit 1
lscpu | grep -E "^CPU\(s\)|NUMA node|Socket|Thread" | head -8
run() { # label, taskset spec or "", threads
  if [ -n "$2" ]; then pre="taskset -c $2"; else pre=""; fi
  PSK_INGEST_THREADS=$3 $pre timeout 200 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 threads=$3', 'host_to_host', round(d['host_to_host']['value']), 'pairs/s', round(d['host_to_host']['ms_per_step'],1), 'ms; ascii', round(d['extras'].get('host_ascii_pairs_per_s',0)))"
}
run all "" 8
run all "" 16
run socket0 0-63 8
run socket0 0-63 16
run socket0 0-63 32
run socket0+smt 0-63,128-191 16
run socket1 64-127 16
