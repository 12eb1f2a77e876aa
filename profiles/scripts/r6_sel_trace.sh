#!/bin/bash
# phases of the cooperative group selection of Gb-scale pairs (variant build -DSEL_TRACE: device printf of wall-clock stamps per pair)
cd "$GRAFT_REPO_ROOT" || exit 1
PSK_LIB_PATH=$PWD/pyskani_amd/libpyskani_amd_seltrace.so timeout 300 python bench.py --workload mammalian --refs 8 --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | grep SEL_TRACE | tail -40
