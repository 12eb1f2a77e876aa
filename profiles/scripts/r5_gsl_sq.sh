#!/bin/bash
# round 5: SQ counters of the slice join's walk kernels on the 1000 x 1000 all-vs-all step (separate passes)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r5c_gsl}
rm -f gpurun_out/pmc/$tag.sq.txt
bash profiles/scripts/pmc_sq.sh $tag "gsl_walk" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH" -- --workload allvsall --refs 1000 --steps 1 --warmup 1 --cpu-sample 0
cat gpurun_out/pmc/$tag.sq.txt
