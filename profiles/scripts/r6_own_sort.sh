#!/bin/bash
# Gb-scale sketches: every sketch its own (u32, u32) index sort against the group sort of (u64, u32) - tests of the Gb-scale paths, then the mammalian step both ways, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_big.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do for v in own group; do
  if [ $v = group ]; then export PSK_INDEX_GROUP_SORT=1; else unset PSK_INDEX_GROUP_SORT; fi
  timeout 300 python bench.py --workload mammalian --refs 8 --steps 3 --warmup 1 --cpu-sample $([ $rep = 1 ] && echo 8 || echo 0) 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('$v', round(d['ms_per_step'],1), 'hits', d['config']['hits'], d['extras'].get('oracle_check'), {a: round(b,1) for a,b in k.items()})"
done; done
