#!/bin/bash
# round 4, after the item-space chunk tables: the Gb-scale counter passes again + the 50 x 3 Gb streamed run
cd "$GRAFT_REPO_ROOT" || exit 1
profiles/scripts/pmc.sh r4_mammal "anchor_join4|anchor_emit_expand|anchor_next|chunk_hops_sliced|chunk_hops_items|item_next|chain_lane20x|chain_chunk_list|select_huge|select_big" --workload mammalian --refs 4 --steps 2 --warmup 1 --cpu-sample 0
python bench.py --workload mammalian --refs 4 --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> /dev/null
cp $(ls -t gpurun_out/bench_full_mammalian_*.json | head -1) gpurun_out/pmc/r4_units_mammalian.json
python bench.py --workload mammalian --refs 50 --stream --cpu-sample 0 2> gpurun_out/pmc/stream50.err | tail -1 > gpurun_out/pmc/stream50_line.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc/stream50_line.json")); print(d["ms_per_step"], d["config"]["hits"], d["kernel_ms_per_step"])
PY
