#!/bin/bash
# round 5: where the emit walk's time goes - timing-only variants (PSK_GSL_STAGE=2: no anchor stores, 3: no line flush, 4: no staging; results are wrong)
cd "$GRAFT_REPO_ROOT" || exit 1
for v in 1 2 3 4 0; do
  PSK_GSL_STAGE=$v bash profiles/scripts/prof.sh r5abl_$v --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
  echo "== PSK_GSL_STAGE=$v"; python3 profiles/summarize.py gpurun_out/prof/r5abl_${v}_kernel_stats.csv 3 | grep gsl_
done
