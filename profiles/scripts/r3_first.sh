#!/bin/bash
# round 3, first GPU call: GPU test suite, the default bench line (with extras.workloads), kernel stats + SQ counters of the all-vs-all step
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -5 gpurun_out/r3a/pytest.log
start=$(date +%s)
timeout 1750 python bench.py > gpurun_out/r3a/bench_default.json 2> gpurun_out/r3a/bench_default.err; echo "bench rc=$? wall=$(( $(date +%s) - start )) s" | tee -a gpurun_out/r3a/bench_default.err
tail -c 600 gpurun_out/r3a/bench_default.err
profiles/scripts/prof.sh r3a_ava --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0
profiles/scripts/pmc_sq.sh r3a_ava "chain_lane20|anchor_join4|anchor_emit_pairs" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" -- --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
cat gpurun_out/pmc/r3a_ava.sq.txt | head -40
