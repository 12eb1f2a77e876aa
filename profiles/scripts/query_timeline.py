import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44]) for r in csv.DictReader(open(f))]
rows.sort()
# last complete query: from the second-to-last sketch_scan to the last one
sc = [i for i, r in enumerate(rows) if r[2].startswith("sketch_scan_kernel")]
a, b = sc[-2], sc[-1]
t0 = rows[a][0]; prev = t0
print(f"one sketch+query: {(rows[b][0]-t0)/1e3:.1f} us, {b-a} kernels, busy {sum(e-s for s,e,_ in rows[a:b])/1e3:.1f} us")
for s, e, n in rows[a:b]:
    print(f"  +{(s-t0)/1e3:7.1f} gap {(s-prev)/1e3:6.1f} run {(e-s)/1e3:6.1f}  {n}")
    prev = max(prev, e)
