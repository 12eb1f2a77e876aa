#!/bin/bash
# A/B of library builds on the metagenome step: profiles/scripts/r6_meta_ab.sh <variant> ... ("base" = libpyskani_amd.so), each twice, interleaved; first the contig-join tests on the default build
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_scale_paths.py tests/test_gpu_parity.py tests/test_gpu_small_query.py -x -q -m gpu -k "rescue or beyond or blocks or prefilter or alternative or contig or many" 2>&1 | tail -2
PSK_FUZZ_SEEDS=320 PSK_FUZZ_DB_SEEDS=160 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = base ]; then unset PSK_LIB_PATH; else export PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_$v.so; fi
  timeout 300 python bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],1), 'hits', d['config']['hits'], {a: round(b,1) for a,b in k.items() if b > 1})"
done; done
