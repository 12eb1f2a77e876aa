#!/bin/bash
# A/B of library builds on the 10 000 x 10 000 step as one chain: profiles/scripts/r5_lib_ab.sh <lib name without lib/.so> ... (each twice, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for v in "$@"; do
  export PSK_LIB_PATH=$PWD/pyskani_amd/lib$v.so
  PSK_PIPELINE=0 timeout 300 python bench.py --workload allvsall --refs ${REFS:-10000} --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],1), d['extras']['hits_digest'], 'join', k['anchor'], 'emit', k['anchor_emit'], 'dp', k['chain_chunk'], 'select', k['select'])"
done; done
