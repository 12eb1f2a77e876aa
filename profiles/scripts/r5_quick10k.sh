#!/bin/bash
# round 5: one 10 000 x 10 000 step per variant (env given as arguments "K=V K=V" per variant), kernel brackets printed
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  echo "== $v"
  env $v timeout 300 python3 bench.py --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms_per_step'], d['extras']['hits_digest'])"
done
