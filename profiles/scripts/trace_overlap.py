"""rocprofv3 kernel trace of a multi-threaded run -> how busy the GPU was and how the kernels of concurrent queries overlapped:
wall span, union of kernel intervals, sum of kernel durations, per-kernel count / mean duration."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(f))]
rows.sort()
tail = rows[len(rows) // 3:]          # skip warm-up / database load
t0, t1 = tail[0][0], max(e for _, e, _ in tail)
union, cur_s, cur_e = 0, None, None
for s, e, _ in tail:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
tot = sum(e - s for s, e, _ in tail)
print(f"span {(t1-t0)/1e6:.1f} ms, GPU busy (union) {union/1e6:.1f} ms = {union/(t1-t0):.2f}, sum of kernel time {tot/1e6:.1f} ms = {tot/(t1-t0):.2f} kernels in flight on average")
by = collections.defaultdict(list)
for s, e, n in tail: by[n].append(e - s)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"  {n:42s} n={len(v):6d} mean {sum(v)/len(v)/1e3:7.1f} us  total {sum(v)/1e6:8.1f} ms")
