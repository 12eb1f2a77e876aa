#!/bin/bash
# Usage (GPU box, repo root): profiles/scripts/pmc_sq.sh <tag> <kernel regex> "<counters pass 1>" "<counters pass 2>" ... -- <bench.py args...>
tag=$1; shift; regex=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
n=0
for c in "${passes[@]}"; do
  n=$((n+1)); rm -rf /tmp/pmc/$tag.$n; mkdir -p /tmp/pmc/$tag.$n
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "$regex" --output-format csv -d /tmp/pmc/$tag.$n -o $tag -- python3 bench.py "$@" > /dev/null 2> gpurun_out/pmc/$tag.$n.err
  f=$(find /tmp/pmc/$tag.$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> gpurun_out/pmc/$tag.sq.txt <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"]); tot[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(tot):
    print(f"{k[0]}\t{k[1]}\t{cnt[k]}\t{tot[k]:.6g}\t{tot[k]/cnt[k]:.6g}")
PY
done
