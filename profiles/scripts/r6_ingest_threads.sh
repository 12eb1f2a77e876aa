#!/bin/bash
# the contract job from ASCII in HOST memory (host_to_host) with 8 / 16 / 32 / 64 packing threads, each twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for t in 8 16 32 64; do
  PSK_INGEST_THREADS=$t timeout 400 python bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-workloads 2>/dev/null | tail -1 > /dev/null
  python3 -c "
import json,glob
f=sorted(glob.glob('gpurun_out/bench_full_allvsall_*.json'))[-1]; d=json.load(open(f)); h=d['host_to_host']; print('threads $t', 'host_to_host', round(h['ms_per_step'],1), 'ingest_s', round(h['ingest_s'],3), 'GB/s', round(h['ingest_host_GBps'],1), 'query_s', round(h['query_s'],3))"
done; done
