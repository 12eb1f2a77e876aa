#!/bin/bash
# round 3: chain selection with one full-length sort per pair: the whole GPU suite, then all-vs-all 1 000 and 10 000, metagenome, 8 x 3 Gb
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3t
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3t/pytest.log
python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3t/ava1000.json 2> gpurun_out/r3t/ava1000.err
python bench.py --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/r3t/ava10k.json 2> gpurun_out/r3t/ava10k.err
python bench.py --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0 > gpurun_out/r3t/meta100k.json 2> gpurun_out/r3t/meta100k.err
python bench.py --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/r3t/mammal8.json 2> gpurun_out/r3t/mammal8.err
for w in ava1000 ava10k meta100k mammal8; do python - gpurun_out/r3t/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 2), d["config"].get("hits"), d.get("extras", {}).get("hits_digest"), {k: round(v, 2) for k, v in d.get("kernel_ms_per_step", {}).items()})
PY
done
