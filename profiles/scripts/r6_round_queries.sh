#!/bin/bash
# metagenome step: one round of 100 000 queries instead of two of 65 536 / 34 464 (PSK_ROUND_QUERIES), each twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for q in default 100000 32768; do
  if [ $q = default ]; then unset PSK_ROUND_QUERIES; else export PSK_ROUND_QUERIES=$q; fi
  timeout 300 python bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('round_queries $q', round(d['ms_per_step'],1), 'hits', d['config']['hits'], {a: round(b,1) for a,b in k.items() if b > 1})"
done; done
