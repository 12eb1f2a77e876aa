#!/bin/bash
# round 3, closing run: full GPU suite, the default bench line, rocprofv3 kernel stats of the headline command and of the other workloads, counters of the Gb-scale kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3z
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r3z/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3z/pytest.log; tail -4 gpurun_out/r3z/pytest.log
start=$(date +%s)
timeout 1750 python bench.py > gpurun_out/r3z/bench_default.json 2> gpurun_out/r3z/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - start )) s" | tee -a gpurun_out/r3z/bench_default.err
profiles/scripts/prof.sh r3z_search --steps 10 --warmup 2 --cpu-sample 0 --no-api --no-workloads
profiles/scripts/prof.sh r3z_ava --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0
profiles/scripts/prof.sh r3z_meta --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
profiles/scripts/prof.sh r3z_mammal --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0
profiles/scripts/pmc.sh r3_mammal "chunk_hops_sliced|anchor_next|chain_lane20|chain_chunk_list|anchor_emit_expand|anchor_join4" --workload mammalian --refs 4 --steps 2 --warmup 1 --cpu-sample 0
python bench.py --workload mammalian --refs 4 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/pmc/r3_units_mammalian.json 2> /dev/null
python bench.py --workload mammalian --refs 50 --stream > gpurun_out/r3z/mammalian_50x3Gb.json 2> gpurun_out/r3z/mammalian_50x3Gb.err; echo "50x rc=$?"
ls gpurun_out/r3z gpurun_out/prof | tail -20
