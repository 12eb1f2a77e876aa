#!/bin/bash
# round 4: timeline of a headline (search) step: GPU idle gaps
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4u
rm -rf /tmp/gaps_search; mkdir -p /tmp/gaps_search
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_search -o g -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-api --no-workloads > gpurun_out/r4u/search.json 2> gpurun_out/r4u/search.err
python3 - /tmp/gaps_search > gpurun_out/r4u/search_gaps.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in csv.DictReader(open(f))]
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("sketch_scan_kernel")]
# the metagenome bench sketches the database once (first scans), then every step starts with one sketch_scan of the contigs: take the last two
a, b = scans[-2], scans[-1]
step = rows[a:b]
t0 = step[0][0]
busy = 0; prev_end = t0; gaps = []
for s, e, n in step:
    if s > prev_end: gaps.append((s - prev_end, (prev_end - t0) / 1e3, n))
    busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
print(f"step {(rows[b][0] - t0) / 1e6:.1f} ms, {len(step)} kernels, GPU busy {busy / 1e6:.1f} ms, idle {(rows[b][0] - t0 - busy) / 1e6:.1f} ms")
gaps.sort(reverse=True)
for g, at, n in gaps[:40]: print(f"  idle {g / 1e3:9.1f} us at +{at:9.1f} us before {n}")
tot = collections.Counter()
for s, e, n in step: tot[n] += e - s
for n, v in tot.most_common(40): print(f"  {v / 1e6:8.2f} ms  {n}")
PY
head -100 gpurun_out/r4u/search_gaps.txt
