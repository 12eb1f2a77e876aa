#!/bin/bash
# round 3: metagenome join through probe tables: tests, A/B, kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3h
timeout 1500 python -m pytest tests/test_gpu_scale_paths.py tests/test_gpu_fuzz.py tests/test_gpu_learned.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3h/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3h/pytest.log
for pb in on rsoff; do
  unset PSK_ROW_SORT; if [ $pb = rsoff ]; then export PSK_ROW_SORT=0; fi
  python bench.py --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0 > gpurun_out/r3h/meta100k_probe$pb.json 2> gpurun_out/r3h/meta100k_probe$pb.err
  python - gpurun_out/r3h/meta100k_probe$pb.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 1), d["config"]["hits"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
done
unset PSK_ROW_SORT
