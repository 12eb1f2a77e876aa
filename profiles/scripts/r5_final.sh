#!/bin/bash
# end of round 5: the complete GPU suite, smoke, the default bench line, rocprofv3 --kernel-trace --stats summaries of the workloads (profiles/r5/r5x_*)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5x
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5x/r5x_pytest_gpu_tail.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5x/r5x_smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/r5x/r5x_bench_default_lines.txt 2> gpurun_out/r5x/r5x_bench_default.err
tail -1 gpurun_out/r5x/r5x_bench_default_lines.txt > gpurun_out/r5x/r5x_bench_default_line.json
cp $(ls -t gpurun_out/bench_full_search_*.json | head -1) gpurun_out/r5x/r5x_bench_default_full.json
timeout 300 bash profiles/scripts/prof.sh r5x_search --steps 10 --warmup 2 --cpu-sample 0 --no-api --no-workloads
python3 profiles/summarize.py gpurun_out/prof/r5x_search_kernel_stats.csv 12 > gpurun_out/r5x/r5x_kernel_stats_search.md
export PSK_PIPELINE=0      # (kernel tables of the all-vs-all steps: one chain of launches - with two batches in flight a kernel's duration is that of a kernel sharing the chip)
timeout 300 bash profiles/scripts/prof.sh r5x_ava --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r5x_ava_kernel_stats.csv 3 > gpurun_out/r5x/r5x_kernel_stats_ava.md
timeout 400 bash profiles/scripts/prof.sh r5x_ava10k --workload allvsall --refs 10000 --steps 1 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r5x_ava10k_kernel_stats.csv 2 > gpurun_out/r5x/r5x_kernel_stats_ava10k.md
unset PSK_PIPELINE
timeout 300 bash profiles/scripts/prof.sh r5x_meta --workload metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r5x_meta_kernel_stats.csv 3 > gpurun_out/r5x/r5x_kernel_stats_meta.md
timeout 300 bash profiles/scripts/prof.sh r5x_mammal --workload mammalian --refs 8 --steps 2 --warmup 1 --cpu-sample 0
python3 profiles/summarize.py gpurun_out/prof/r5x_mammal_kernel_stats.csv 3 > gpurun_out/r5x/r5x_kernel_stats_mammal.md
PSK_FUZZ_SEEDS=320 PSK_FUZZ_DB_SEEDS=48 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r5x/r5x_fuzz_sweep_368.txt
cat gpurun_out/r5x/r5x_fuzz_sweep_368.txt gpurun_out/r5x/r5x_pytest_gpu_tail.txt gpurun_out/r5x/r5x_smoke.txt; head -c 1500 gpurun_out/r5x/r5x_bench_default_line.json
