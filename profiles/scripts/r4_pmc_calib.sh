#!/bin/bash
# round 4: what FETCH_SIZE / WRITE_SIZE report for known byte counts in this library's access shapes (profiles/micro/pmc_calib.hip)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc
[ -x profiles/micro/pmc_calib ] || hipcc --offload-arch=gfx950 -O3 profiles/micro/pmc_calib.hip -o profiles/micro/pmc_calib
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc/calib.$c; mkdir -p /tmp/pmc/calib.$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc/calib.$c -o calib -- ./profiles/micro/pmc_calib > gpurun_out/pmc/calib.$c.bytes 2> gpurun_out/pmc/calib.$c.err
  f=$(find /tmp/pmc/calib.$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" > gpurun_out/pmc/calib.$c.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") == sys.argv[2]:
        print(r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]), sep="\t")
PY
done
python3 - <<'PY'
asked = {}
for line in open("gpurun_out/pmc/calib.FETCH_SIZE.bytes"):
    p = line.split()
    if len(p) >= 2: asked[p[0]] = [int(x) for x in p[1:]]
out = ["| kernel | bytes asked for | bytes of touched 64-B lines | FETCH_SIZE (KiB x 1024) | WRITE_SIZE (KiB x 1024) | counter / asked | counter / lines |", "|---|---|---|---|---|---|---|"]
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for line in open(f"gpurun_out/pmc/calib.{c}.txt"):
        k, v = line.rstrip("\n").split("\t"); cnt.setdefault(k.replace("void ", "").strip(), {})[c] = float(v) * 1024.0
for k, a in asked.items():
    f, w = cnt.get(k, {}).get("FETCH_SIZE", 0.0), cnt.get(k, {}).get("WRITE_SIZE", 0.0)
    main = f if k.startswith("rd_") else w
    lines = a[1] if len(a) > 1 else a[0]
    out.append(f"| {k} | {a[0]:.4g} | {lines:.4g} | {f:.4g} | {w:.4g} | {main / a[0]:.3f} | {main / lines:.3f} |")
open("gpurun_out/pmc/calib_table.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
