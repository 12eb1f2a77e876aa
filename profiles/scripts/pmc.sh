#!/bin/bash
# Usage (GPU box, repo root): profiles/scripts/pmc.sh <tag> <kernel regex> <bench.py args...>
# HBM traffic counters per kernel, one rocprofv3 pass per counter (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE do not share a pass).
tag=$1; shift; regex=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc/$tag.$c; mkdir -p /tmp/pmc/$tag.$c
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$regex" --output-format csv -d /tmp/pmc/$tag.$c -o $tag -- python3 bench.py "$@" > /dev/null 2> gpurun_out/pmc/$tag.$c.err
  f=$(find /tmp/pmc/$tag.$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" > gpurun_out/pmc/$tag.$c.txt <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == sys.argv[2]:
        k = r["Kernel_Name"].split("(")[0]; tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(tot, key=lambda k: -tot[k]):
    print(f"{k}\t{cnt[k]}\t{tot[k]:.6g}")
PY
done
