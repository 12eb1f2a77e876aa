#!/bin/bash
# A/B of library builds on the 10 000 x 10 000 step: profiles/scripts/r6_lib_ab.sh <variant> ... ("base" = libpyskani_amd.so; others libpyskani_amd_<variant>.so), each twice, interleaved;
# first as one chain of launches (PSK_PIPELINE=0: the brackets are the kernels' own), then with the default two batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
for pipe in 0 default; do for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset PSK_LIB_PATH; else export PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_$v.so; fi
  if [ $pipe = 0 ]; then export PSK_PIPELINE=0; else unset PSK_PIPELINE; fi
  timeout 300 python bench.py --refs ${REFS:-10000} --steps 2 --warmup 1 --cpu-sample 0 --no-host-leg --no-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v pipeline=$pipe', round(d['ms_per_step'],1), d['extras']['hits_digest'], 'join', round(k['anchor'],1), 'emit', round(k['anchor_emit'],1), 'dp', round(k['chain_chunk'],1), 'select', round(k['select'],1))"
done; done; done
