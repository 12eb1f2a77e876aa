#!/bin/bash
# cooperative group selection: workgroups dealt by the pairs' candidate counts (default build) against equal groups (libpyskani_amd_equalgroups.so) - the tests that force the
# group selection, then the mammalian step both ways (interleaved), then the phases of the new default (variant build -DSEL_TRACE)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_big.py tests/test_gpu_parity.py -x -q -m gpu -k "big or large_pair or global_selection or item_hops" 2>&1 | tail -3
PSK_FUZZ_SEEDS=96 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for v in base equalgroups; do
  if [ "$v" = base ]; then unset PSK_LIB_PATH; else export PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_$v.so; fi
  timeout 300 python bench.py --workload mammalian --refs 8 --steps 3 --warmup 1 --cpu-sample $([ $rep = 1 ] && echo 8 || echo 0) 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],1), 'hits', d['config']['hits'], d['extras'].get('oracle_check'), {a: round(b,1) for a,b in k.items()})"
done; done
PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_seltrace.so timeout 300 python bench.py --workload mammalian --refs 8 --steps 1 --warmup 0 --cpu-sample 0 2>/dev/null | grep SEL_TRACE | head -12
