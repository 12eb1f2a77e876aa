#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table of this library's kernels
(torch's data-generation kernels and rocclr copies are dropped). Usage: summarize.py in.csv steps > out.md"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
keep = []
for r in rows:
    n = r["Name"]
    if n.startswith("void at::") or "rocclr" in n:
        continue
    m = re.search(r"segmented_radix_sort_config<[^,]+, ([^,]+), ([^>]+)>", n)
    if m:
        short = f"rocprim segmented_radix_sort<{m.group(1).strip()},{m.group(2).strip().split('::')[-1]}>"
    elif "radix_sort_onesweep" in n or "radix_sort" in n:
        short = "rocprim device radix_sort (onesweep)" if "onesweep" in n else "rocprim radix_sort helper"
    elif "scan_impl" in n or "lookback_scan" in n:
        short = "rocprim scan (" + ("init" if "init_lookback" in n else "main") + ")"
    else:
        short = n.split("(")[0]
    keep.append((short, int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
tot = sum(k[2] for k in keep)
print("| kernel | calls | avg us | total ms | ms/step | share |")
print("|---|---|---|---|---|---|")
for s, c, t, a in sorted(keep, key=lambda k: -k[2]):
    print(f"| {s} | {c} | {a / 1e3:.1f} | {t / 1e6:.3f} | {t / 1e6 / steps:.3f} | {100 * t / tot:.1f}% |")
