#!/bin/bash
# contig join: entries cut by work (pairs x query seeds) - the metagenome step for several caps, each twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for w in "$@"; do
  PSK_GSI_ENTRY_WORK=$w timeout 300 python bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('entry_work $w', round(d['ms_per_step'],1), 'hits', d['config']['hits'], {a: round(b,1) for a,b in k.items() if b > 1}, 'visited', d['extras']['chain_work_per_step']['index_entries_visited'])"
done; done
