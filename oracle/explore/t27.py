"""Aggregation variants for the `median` and `robust` KATs (test_ani.py:49-61) on top of the oracle's
per-chunk (anchors, seeds) records: which natural reading of "median" / "10-90 % trimmed mean" lands on
0.9995 / 0.9977 while AF x2 and the raw mean 0.9946 stay met?  TEST-INFRASTRUCTURE scratch, not product."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import load_fasta_first_record
from oracle import oracle as O

ec = load_fasta_first_record("e.coli-EC590.fasta.gz"); k12 = load_fasta_first_record("e.coli-K12.fasta.gz")
ref, q = O.Sketch([ec]), O.Sketch([k12])
res = O.chain(ref, q)
ch = O.last_chunks()
A = ch["anchors"].astype(float); S = ch["seeds"].astype(float)
print("chunks", len(ch), "mean raw", res.ani)
K = 15
def vals(dS, clamp=True):
    r = A / np.maximum(S + dS, 1)
    if clamp: r = np.minimum(r, 1)
    return r ** (1 / K)

T_MED, T_ROB, T_MEAN = 0.9995, 0.9977, 0.9946
rows = []
for dS in (-1, 0, -2):
    for clamp in (True, False):
        v = vals(dS, clamp); n = len(v); sv = np.sort(v)
        o = np.argsort(v, kind="stable")
        # ---- medians
        med = {
            "upper(n/2)": sv[n // 2], "lower((n-1)/2)": sv[(n - 1) // 2], "mean-of-middle": 0.5 * (sv[(n - 1) // 2] + sv[n // 2]),
            "np.median": np.median(v),
        }
        for wn, w in (("wA", A), ("wS", S)):
            cw = np.cumsum(w[o]); med[f"weighted-{wn}"] = v[o][np.searchsorted(cw, cw[-1] / 2)]
        med["ratio-of-medians"] = (np.median(A) / np.median(S + dS)) ** (1 / K)
        med["pooled-middle-half"] = (A[o][n // 4: n - n // 4].sum() / (S + dS)[o][n // 4: n - n // 4].sum()) ** (1 / K)
        for name, m in med.items():
            rows.append(("median", f"dS={dS} clamp={clamp} {name}", m, abs(m - T_MED)))
        # ---- robust: trimmed means
        for lo_rule, hi_rule in itertools.product(("floor", "ceil", "round"), repeat=2):
            f = {"floor": np.floor, "ceil": np.ceil, "round": np.round}
            lo = int(f[lo_rule](0.1 * n)); hi = int(f[hi_rule](0.9 * n))
            for wn, w in (("u", np.ones(n)), ("wA", A), ("wS", S)):
                m = (sv[lo:hi] * w[o][lo:hi]).sum() / w[o][lo:hi].sum()
                rows.append(("robust", f"dS={dS} clamp={clamp} trim[{lo_rule} .1n,{hi_rule} .9n) {wn}", m, abs(m - T_ROB)))
        # trim by weight (10 % of anchors / seeds at each end)
        for wn, w in (("wA", A), ("wS", S)):
            cw = np.cumsum(w[o]) / w.sum(); keep = (cw > 0.1) & (cw <= 0.9)
            m = (sv[keep] * w[o][keep]).sum() / w[o][keep].sum()
            rows.append(("robust", f"dS={dS} clamp={clamp} trim-by-weight {wn}", m, abs(m - T_ROB)))
            rows.append(("robust", f"dS={dS} clamp={clamp} trim-by-weight {wn} unweighted-mean", sv[keep].mean(), abs(sv[keep].mean() - T_ROB)))
        # only-lower / only-upper trimming, winsorising, pooled ratio of the kept chunks
        lo, hi = n // 10, n - n // 10
        rows.append(("robust", f"dS={dS} clamp={clamp} drop-low-10%-only", sv[lo:].mean(), abs(sv[lo:].mean() - T_ROB)))
        rows.append(("robust", f"dS={dS} clamp={clamp} drop-high-10%-only", sv[:hi].mean(), abs(sv[:hi].mean() - T_ROB)))
        w_ = sv.copy(); w_[:lo] = sv[lo]; w_[hi:] = sv[hi - 1]
        rows.append(("robust", f"dS={dS} clamp={clamp} winsorised", w_.mean(), abs(w_.mean() - T_ROB)))
        pr = (A[o][lo:hi].sum() / (S + dS)[o][lo:hi].sum()) ** (1 / K)
        rows.append(("robust", f"dS={dS} clamp={clamp} pooled-ratio-of-kept", pr, abs(pr - T_ROB)))
        for fl, fh in ((0.05, 0.95), (0.2, 0.8), (0.25, 0.75), (0.1, 1.0), (0.15, 0.85)):
            m = sv[int(fl * n): int(fh * n)].mean()
            rows.append(("robust", f"dS={dS} clamp={clamp} trim[{fl},{fh}) u", m, abs(m - T_ROB)))
with open("/tmp/search/t27.txt", "w") as fo:
    for kind in ("median", "robust"):
        sel = sorted([r for r in rows if r[0] == kind], key=lambda r: r[3])
        print(f"== {kind}: {len(sel)} variants; within 5e-5: {sum(r[3] < 5e-5 for r in sel)}")
        for r in sel[:12]:
            print("  %-70s %.6f  |d|=%.2e" % (r[1], r[2], r[3]))
        for r in sel: fo.write("%s\t%s\t%.6f\t%.2e\n" % r)
