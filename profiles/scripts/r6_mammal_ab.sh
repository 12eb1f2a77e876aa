#!/bin/bash
# A/B of library builds on the mammalian 8 x 3 Gb step: profiles/scripts/r6_mammal_ab.sh <variant> ...   ("base" = libpyskani_amd.so; others libpyskani_amd_<variant>.so), each twice, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset PSK_LIB_PATH; else export PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_$v.so; fi
  timeout 300 python bench.py --workload mammalian --refs 8 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],1), 'hits', d['config']['hits'], {a: round(b,1) for a,b in k.items()})"
done; done
