#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3x
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_database_gpu.py tests/test_gpu_learned.py -m gpu -x -q 2>&1 | tail -3
for fs in "" "--faster-small"; do
  python bench.py --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0 $fs > gpurun_out/r3x/meta100k$fs.json 2> gpurun_out/r3x/meta100k$fs.err
  python - "gpurun_out/r3x/meta100k$fs.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("hits"), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
done
python bench.py --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/r3x/ava10k.json 2> gpurun_out/r3x/ava10k.err
python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3x/ava1000.json 2> gpurun_out/r3x/ava1000.err
for w in ava10k ava1000; do python - gpurun_out/r3x/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("hits"), d.get("extras", {}).get("hits_digest"), {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items()})
PY
done
