#!/bin/bash
# round 4: packed host ingest - tests, the headline workload from host memory (api / host_ascii / host_packed), thread sweep; per-contig query threads with C-built Hit lists
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_small_query.py -x -q > gpurun_out/r4d/pytest.txt 2>&1; tail -4 gpurun_out/r4d/pytest.txt
python bench.py --no-workloads --cpu-sample 0 > gpurun_out/r4d/bench_search.out 2> gpurun_out/r4d/bench_search.err
python - <<'PY'
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_search_*.json"))[-1]
d=json.load(open(f)); e=d["extras"]
print({k:e[k] for k in ("api_pairs_per_s","host_ascii_pairs_per_s","host_packed_pairs_per_s")})
print(e["host_ascii_detail"]); print(e["host_packed_detail"])
PY
for t in 8 16 32 64; do PSK_INGEST_THREADS=$t python bench.py --no-workloads --cpu-sample 0 --steps 3 --warmup 1 > /dev/null 2>&1; python - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_search_*.json"))[-1]
e=json.load(open(f))["extras"]; print("threads $t", round(e["host_ascii_pairs_per_s"]), round(e["host_packed_pairs_per_s"]), e["host_packed_detail"]["host_GBps"])
PY
done 2>&1 | tee gpurun_out/r4d/ingest_threads.txt
NQ=4000 timeout 600 python3 profiles/scripts/query_threads.py > gpurun_out/r4d/query_threads.txt 2>&1; tail -3 gpurun_out/r4d/query_threads.txt
