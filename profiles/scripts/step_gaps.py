"""Timeline of one steady-state bench step from a rocprofv3 kernel trace: kernels in start order with the idle gap before each.
Usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -o g -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-api
                 python3 profiles/scripts/step_gaps.py /tmp/gaps"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48]) for r in csv.DictReader(open(f))]
rows.sort()
scans = [i for i, r in enumerate(rows) if r[2].startswith("sketch_scan_kernel")]
a, b = scans[-2], scans[-1]          # the last complete step: from its sketch_scan to the next one
step = rows[a:b]
t0 = step[0][0]
print(f"step: {(rows[b][0] - t0) / 1e3:.1f} us from sketch_scan to the next sketch_scan, {len(step)} kernels, busy {sum(e - s for s, e, _ in step) / 1e3:.1f} us")
prev_end = t0
for s, e, n in step:
    gap = s - prev_end
    if gap > 8000 or e - s > 30000:
        print(f"  +{(s - t0) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  run {(e - s) / 1e3:8.1f}  {n}")
    prev_end = max(prev_end, e)
print(f"  tail gap to next step: {(rows[b][0] - prev_end) / 1e3:.1f} us")
