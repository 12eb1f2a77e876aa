#!/bin/bash
# round 3: HBM traffic counters (FETCH_SIZE / WRITE_SIZE, one rocprofv3 pass each) of the dominant kernels of the non-search workloads,
# plus kernel stats of the metagenome step. Summarised by profiles/scripts/pmc_summary.py into profiles/r3/pmc_kernels.json.
cd "$GRAFT_REPO_ROOT"
profiles/scripts/pmc.sh r3_ava "anchor_join4|anchor_emit_pairs|chain_lane20|select_kernel" --workload allvsall --refs 1000 --steps 2 --warmup 1 --cpu-sample 0
profiles/scripts/pmc.sh r3_meta "anchor_join_probe|chain_quad_deep|chain_chunk_list|chunk_heads|anchor_emit_packed4|pref_count|select_kernel" --workload metagenome --refs 5000 --queries 20000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
profiles/scripts/pmc.sh r3_mammal "chunk_hops_sliced|anchor_next|chain_lane20|chain_chunk_list|anchor_emit_expand|anchor_join4" --workload mammalian --refs 4 --steps 2 --warmup 1 --cpu-sample 0
profiles/scripts/prof.sh r3_meta --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0
for w in "allvsall --refs 1000" "metagenome --refs 5000 --queries 20000 --api-queries 0" "mammalian --refs 4"; do
  tag=$(echo $w | cut -d' ' -f1)
  python bench.py --workload $w --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/pmc/r3_units_$tag.json 2> /dev/null
done
ls -la gpurun_out/pmc | tail -12
