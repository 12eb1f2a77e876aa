"""Development aid: the same all-vs-all batch through two joins (PSK_GSI_SLICE 1 / 0, read per round), records compared field by field.
usage: python tools/diff_join_paths.py [n_genomes] [variant]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 200
variant = sys.argv[2] if len(sys.argv) > 2 else "plain"
dev = torch.device("cuda:0")
n_families = max(1, n_total // 100)
anc_lens, fam_of = bench.family_layout(3, n_total, n_families)
r = bench.make_genomes(torch, dev, 3, 31, list(range(n_total)), fam_of, anc_lens, variant=variant)
buf, offs, lens = r[0], r[1], r[2]
gfc_list = r[3] if len(r) > 3 else None
torch.cuda.synchronize()
eng = bench.Engine(0)
names = (C.c_char_p * n_total)(*[f"g{i}".encode() for i in range(n_total)])
c_off, c_len, gfc, n = eng.layout(offs, lens, gfc_list)
out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, n)
res = {}
for mode in ("1", "0"):
    os.environ["PSK_GSI_SLICE"] = mode
    db = eng.make_db(names, out, n)      # (a fresh database per mode: the first builds the seed index, the second the per-sketch indexes)
    nh, (recs, qoffs) = eng.query_many(db, out, n, keep=True, raw=True)
    recs = recs.copy()
    recs["reserved"] = np.repeat(np.arange(n, dtype=np.uint32), np.diff(qoffs))
    res[mode] = recs
    print("mode", mode, "hits", nh, "digest", bench.records_digest(recs))
a, b = res["1"], res["0"]
if len(a) != len(b):
    print("hit counts differ", len(a), len(b))
    sys.exit(1)
bad = np.zeros(len(a), bool)
for f in a.dtype.names:
    d = a[f] != b[f]
    if d.any():
        print("field", f, "differs in", int(d.sum()), "records")
        bad |= d
for i in np.flatnonzero(bad)[:12]:
    print("record", i, "\n  slice :", a[i], "\n  pairs :", b[i])
print("differing records:", int(bad.sum()), "of", len(a))
