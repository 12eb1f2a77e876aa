#!/bin/bash
# round 4: counter passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs) for the chain-stage kernels of the three large workloads + the unit counts of the same runs
cd "$GRAFT_REPO_ROOT" || exit 1
profiles/scripts/pmc.sh r4_ava "anchor_join4|anchor_emit_pairs|chain_lane20|select_kernel" --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
profiles/scripts/pmc.sh r4_meta "gsi_join_kernel|gsi_prefilter|chain_quad_deep|chain_chunk_list|select_tiny|pair_build_rows" --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
profiles/scripts/pmc.sh r4_mammal "anchor_join4|anchor_emit_expand|anchor_next|chunk_hops_sliced|chain_lane20x|chain_chunk_list|select_huge|select_big" --workload mammalian --refs 4 --steps 2 --warmup 1 --cpu-sample 0
for w in "allvsall --refs 1000 --steps 2" "metagenome --refs 5000 --queries 100000 --api-queries 0 --steps 2" "mammalian --refs 4 --steps 2"; do
  tag=$(echo $w | cut -d' ' -f1)
  python bench.py --workload $w --warmup 1 --cpu-sample 0 > /dev/null 2> /dev/null
  cp $(ls -t gpurun_out/bench_full_${tag}_*.json | head -1) gpurun_out/pmc/r4_units_$tag.json
done
ls gpurun_out/pmc | head -30
