#!/bin/bash
# end of round 6: the complete GPU suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats summaries of the workloads (profiles/r6/r6z_*)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6z
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r6z/r6z_pytest_gpu_tail.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6z/r6z_smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/r6z/r6z_bench_default_lines.txt 2> gpurun_out/r6z/r6z_bench_default.err
tail -1 gpurun_out/r6z/r6z_bench_default_lines.txt > gpurun_out/r6z/r6z_bench_default_line.json
cp $(ls -t gpurun_out/bench_full_allvsall_*.json | head -1) gpurun_out/r6z/r6z_bench_default_full.json
timeout 300 bash profiles/scripts/prof.sh r6z_search --workload search --steps 10 --warmup 2 --cpu-sample 0 --no-api
python3 profiles/summarize.py gpurun_out/prof/r6z_search_kernel_stats.csv 12 > gpurun_out/r6z/r6z_kernel_stats_search.md
export PSK_PIPELINE=0      # (kernel tables of the all-vs-all steps: one chain of launches - with two batches in flight a kernel's duration is that of a kernel sharing the chip)
timeout 300 bash profiles/scripts/prof.sh r6z_ava --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r6z_ava_kernel_stats.csv 3 > gpurun_out/r6z/r6z_kernel_stats_ava.md
timeout 500 bash profiles/scripts/prof.sh r6z_ava10k --refs 10000 --steps 1 --warmup 1 --cpu-sample 0 --no-host-leg --no-workloads
python3 profiles/summarize.py gpurun_out/prof/r6z_ava10k_kernel_stats.csv 2 > gpurun_out/r6z/r6z_kernel_stats_ava10k.md
unset PSK_PIPELINE
cat gpurun_out/r6z/r6z_pytest_gpu_tail.txt gpurun_out/r6z/r6z_smoke.txt; head -c 1800 gpurun_out/r6z/r6z_bench_default_line.json; echo; head -14 gpurun_out/r6z/r6z_kernel_stats_ava10k.md | cut -c1-110
