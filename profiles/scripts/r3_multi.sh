#!/bin/bash
# round 3: full GPU suite + the sharded all-vs-all as a two-rank dry run on ONE GPU (gloo; both ranks on device 0) against the single-process run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest.log
tail -6 gpurun_out/r3c/pytest.log
python bench.py --workload allvsall --refs 1000 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/r3c/ava1000_1rank.json 2> gpurun_out/r3c/ava1000_1rank.err
for comm in torch; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --workload allvsall --refs 1000 --steps 3 --warmup 1 --backend gloo --share-gpu --comm $comm > gpurun_out/r3c/ava1000_2rank_$comm.json 2> gpurun_out/r3c/ava1000_2rank_$comm.err
  echo "2-rank $comm rc=$?"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --share-gpu --no-workloads > gpurun_out/r3c/search_2rank.json 2> gpurun_out/r3c/search_2rank.err; echo "search 2-rank rc=$?"
python - <<'PY'
import json
for f in ("ava1000_1rank", "ava1000_2rank_torch", "search_2rank"):
    try:
        d = json.loads(open(f"gpurun_out/r3c/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 1), d["value"], d["config"].get("hits"), d["extras"].get("hits_digest"), d["extras"].get("exchange"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 gpurun_out/r3c/ava1000_2rank_torch.err
