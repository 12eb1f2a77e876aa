#!/bin/bash
# round 3: chunk-table rows by length for every DP kernel that shares a wave between chunks: tests, then each workload with and without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3r
timeout 2400 python -m pytest tests/test_gpu_scale_paths.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_big.py -m gpu -x -q > gpurun_out/r3r/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3r/pytest.log
for rs in on off; do
  unset PSK_ROW_SORT; if [ $rs = off ]; then export PSK_ROW_SORT=0; fi
  python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3r/ava1000_$rs.json 2> gpurun_out/r3r/ava1000_$rs.err
  python bench.py --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/r3r/mammal8_$rs.json 2> gpurun_out/r3r/mammal8_$rs.err
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-api --no-workloads > gpurun_out/r3r/search_$rs.json 2> gpurun_out/r3r/search_$rs.err
  for w in ava1000 mammal8 search; do python - gpurun_out/r3r/${w}_$rs.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()})
PY
  done
done
unset PSK_ROW_SORT
