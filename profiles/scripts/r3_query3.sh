#!/bin/bash
# round 3: per-contig Database.query: tests, one call's split and timeline, the bench's API leg (one thread / eight threads)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3q
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_database_gpu.py tests/test_gpu_learned.py -m gpu -x -q > gpurun_out/r3q/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3q/pytest.log
python3 profiles/scripts/query_latency.py > gpurun_out/r3q/query_latency.txt 2>&1; tail -3 gpurun_out/r3q/query_latency.txt
rm -rf /tmp/q1; mkdir -p /tmp/q1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/q1 -o q -- python3 profiles/scripts/query_latency.py > /dev/null 2> gpurun_out/r3q/trace.err
python3 profiles/scripts/query_timeline.py /tmp/q1 > gpurun_out/r3q/query_timeline.txt 2>&1; head -60 gpurun_out/r3q/query_timeline.txt
python3 bench.py --workload metagenome --refs 5000 --queries 10000 --api-queries 4000 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r3q/meta10k_api.json 2> gpurun_out/r3q/meta10k_api.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r3q/meta10k_api.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"], 1), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d["extras"].get("api", {}).items() if k != "note"})
PY
