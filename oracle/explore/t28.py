"""One consistent (seed-count rule, weight) across mean / median / robust: table of all three per rule."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from conftest import load_fasta_first_record
from oracle import oracle as O
ec = load_fasta_first_record("e.coli-EC590.fasta.gz"); k12 = load_fasta_first_record("e.coli-K12.fasta.gz")
ref, q = O.Sketch([ec]), O.Sketch([k12])
res = O.chain(ref, q); ch = O.last_chunks()
A = ch["anchors"].astype(float); S = ch["seeds"].astype(float); n = len(A)
print("dS clamp w | mean(.9946) median(.9995) robust(.9977)")
for dS in (-2, -1, 0, 1):
    for clamp in (True, False):
        r = A / np.maximum(S + dS, 1)
        if clamp: r = np.minimum(r, 1)
        v = r ** (1 / 15); o = np.argsort(v, kind="stable"); sv = v[o]
        for wn, w in (("u", np.ones(n)), ("wA", A), ("wS", S + dS)):
            mean = (v * w).sum() / w.sum()
            cw = np.cumsum(w[o]); med = sv[np.searchsorted(cw, cw[-1] / 2)] if wn != "u" else sv[n // 2]
            lo, hi = n // 10, n * 9 // 10
            rob = (sv[lo:hi] * w[o][lo:hi]).sum() / w[o][lo:hi].sum()
            ok = lambda x, t: "*" if abs(x - t) < 5e-5 else " "
            print(f"{dS:+d} {int(clamp)} {wn:3s} | {mean:.6f}{ok(mean,.9946)} {med:.6f}{ok(med,.9995)} {rob:.6f}{ok(rob,.9977)}")
