#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3v
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_database_gpu.py -m gpu -x -q 2>&1 | tail -3
for sw in on off; do
  unset PSK_SCREEN_WAVE; if [ $sw = off ]; then export PSK_SCREEN_WAVE=0; fi
  python bench.py --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/r3v/ava10k_$sw.json 2> gpurun_out/r3v/ava10k_$sw.err
  python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3v/ava1000_$sw.json 2> gpurun_out/r3v/ava1000_$sw.err
  python bench.py --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0 > gpurun_out/r3v/meta100k_$sw.json 2> gpurun_out/r3v/meta100k_$sw.err
  for w in ava10k ava1000 meta100k; do python - gpurun_out/r3v/${w}_$sw.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("hits"), d.get("extras", {}).get("hits_digest"), "screen", round(d["kernel_ms_per_step"]["screen"], 2))
PY
  done
done
