#!/bin/bash
# round 3: the position-ordered probe join against the k-mer-ordered join + per-pair emit (tests, A/B timing, kernel stats)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3b
timeout 1200 python -m pytest tests/test_gpu_scale_paths.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
tail -15 gpurun_out/r3b/pytest.log
for pb in rows classic; do
  unset PSK_PROBE PSK_ROWS; if [ $pb = classic ]; then export PSK_ROWS=0; fi;
  python bench.py --workload allvsall --refs 1000 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r3b/ava1000_probe$pb.json 2> gpurun_out/r3b/ava1000_probe$pb.err
  python - gpurun_out/r3b/ava1000_probe$pb.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["config"]["hits"], d["extras"]["hits_digest"], {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
PY
done
unset PSK_PROBE PSK_ROWS
profiles/scripts/prof.sh r3b_ava_probe --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0
