#!/bin/bash
# round 4: timeline of a metagenome step: idle gaps, and every kernel over 300 us in start order (what runs outside the event brackets)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4u
rm -rf /tmp/gaps_meta; mkdir -p /tmp/gaps_meta
rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_meta -o g -- python3 bench.py --workload metagenome --refs 5000 --queries 100000 --steps 2 --warmup 1 --cpu-sample 0 --api-queries 0 > gpurun_out/r4u/meta.json 2> gpurun_out/r4u/meta.err
python3 - /tmp/gaps_meta > gpurun_out/r4u/meta_gaps.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]) for r in csv.DictReader(open(f))]
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("sketch_scan_kernel")]
a, b = scans[-2], scans[-1]
step = rows[a:b]
t0 = step[0][0]
busy = 0; prev_end = t0; gaps = []
for s, e, n in step:
    if s > prev_end: gaps.append((s - prev_end, (prev_end - t0) / 1e3, n))
    busy += max(0, e - max(s, prev_end)); prev_end = max(prev_end, e)
print(f"step {(rows[b][0] - t0) / 1e6:.1f} ms, {len(step)} kernels, GPU busy {busy / 1e6:.1f} ms, idle {(rows[b][0] - t0 - busy) / 1e6:.1f} ms")
gaps.sort(reverse=True)
for g, at, n in gaps[:15]: print(f"  idle {g / 1e3:9.1f} us at +{at:9.1f} us before {n}")
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in step: tot[n] += e - s; cnt[n] += 1
for n, v in tot.most_common(45): print(f"  {v / 1e6:8.2f} ms  x{cnt[n]:<4d} {n}")
print("in start order (kernels over 300 us):")
for s, e, n in step:
    if e - s > 300000: print(f"  +{(s - t0) / 1e6:8.2f} ms  {(e - s) / 1e6:7.2f} ms  {n}")
PY
head -150 gpurun_out/r4u/meta_gaps.txt
