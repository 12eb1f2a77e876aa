#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4u
rm -rf /tmp/gaps_search; mkdir -p /tmp/gaps_search
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gaps_search -o g -- python3 bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-api --no-workloads > gpurun_out/r4u/search.json 2> gpurun_out/r4u/search.err
python3 - /tmp/gaps_search > gpurun_out/r4u/search_seq.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]) for r in csv.DictReader(open(f))]
for g in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("sketch_scan_kernel")]
a, b = scans[-2], scans[-1]
t0 = rows[a][0]; prev = t0
for s, e, n in rows[a:b + 1]:
    print(f"+{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  run {(e - s) / 1e3:7.1f}  {n}")
    prev = max(prev, e)
PY
cat gpurun_out/r4u/search_seq.txt
