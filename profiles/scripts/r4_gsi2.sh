#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4j
timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_scale_paths.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r4j/pytest.txt 2>&1; tail -12 gpurun_out/r4j/pytest.txt
for fs in "" "--faster-small"; do python3 bench.py --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0 $fs > /dev/null 2> gpurun_out/r4j/meta.err; python3 - <<PY
import json,glob
f=sorted(glob.glob("gpurun_out/bench_full_metagenome_*.json"))[-1]
d=json.load(open(f)); print("$fs", round(d["ms_per_step"],1), d["config"]["hits"], {k: round(v,1) for k,v in d["kernel_ms_per_step"].items()})
PY
done
bash profiles/scripts/prof.sh r4j_meta --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r4j_meta_kernel_stats.csv 3 > gpurun_out/prof/r4j_kernel_stats_meta.md; head -16 gpurun_out/prof/r4j_kernel_stats_meta.md
