"""Exploration driver (not the oracle, not shipped): hypothesis search vs pyskani KATs."""
import ctypes as C, gzip, os, sys, numpy as np, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
so = os.path.join(HERE, "libex.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "ex.c"), "-lm"])
L = C.CDLL(so)

seed_dt = np.dtype([("kmer", "<u8"), ("pos", "<u4"), ("contig", "<u4"), ("canon", "u1")], align=True)
anchor_dt = np.dtype([("qc", "<u4"), ("qp", "<u4"), ("rp", "<u4"), ("rc", "<u4"), ("rev", "u1")], align=True)
interval_dt = np.dtype([("chunk", "<i4"), ("qc", "<u4"), ("q0", "<u4"), ("q1", "<u4"), ("rc", "<u4"), ("r0", "<u4"), ("r1", "<u4"),
                        ("nanch", "<i4"), ("nseeds", "<i4"), ("score", "<f8"), ("rev", "<i4"), ("setsize", "<i4")], align=True)

class CP(C.Structure):
    _fields_ = [("frag_len", C.c_int), ("max_gap", C.c_double), ("anchor_score", C.c_double), ("min_anchors", C.c_int),
                ("band", C.c_int), ("bp_band", C.c_int), ("max_lin", C.c_double), ("k", C.c_int), ("mult_cap", C.c_int),
                ("chunk_mode", C.c_int), ("gapcost_mode", C.c_int), ("chainset_mode", C.c_int), ("require_mono", C.c_int), ("gap_w", C.c_double)]

def load(name):
    with gzip.open(os.path.join(ROOT, "tests/golden", name), "rt") as f:
        seq = []
        started = False
        for line in f:
            if line.startswith(">"):
                if started: break
                started = True
                continue
            seq.append(line.strip())
    return "".join(seq).encode()

def sketch(seq, c=125, mc=1000, k=15, hashvar=0, marker_mode=0):
    out = np.zeros(len(seq) // max(1, c // 3) + 1000, dtype=seed_dt)
    markers = np.zeros(len(seq) // max(1, c // 3) + 1000, dtype=np.uint64)
    nm = C.c_long(0)
    L.ex_sketch.restype = C.c_long
    n = L.ex_sketch(seq, C.c_long(len(seq)), c, mc, k, hashvar, 0, out.ctypes.data_as(C.c_void_p), markers.ctypes.data_as(C.c_void_p), C.byref(nm), marker_mode)
    return out[:n].copy(), np.unique(markers[:nm.value])

def chain(qs, rs, **kw):
    p = CP(frag_len=20000, max_gap=50, anchor_score=20, min_anchors=3, band=100, bp_band=2500, max_lin=5000, k=15, mult_cap=0,
           chunk_mode=0, gapcost_mode=0, chainset_mode=0, require_mono=1, gap_w=0.0)
    for a, b in kw.items(): setattr(p, a, b)
    out = np.zeros(2000000, dtype=interval_dt)
    na = C.c_long(0)
    A = np.zeros(4000000, dtype=anchor_dt); ch = np.zeros(4000000, dtype=np.int32)
    L.ex_chain.restype = C.c_long
    n = L.ex_chain(qs.ctypes.data_as(C.c_void_p), C.c_long(len(qs)), rs.ctypes.data_as(C.c_void_p), C.c_long(len(rs)), C.byref(p),
                   out.ctypes.data_as(C.c_void_p), C.c_long(len(out)), C.byref(na), A.ctypes.data_as(C.c_void_p), ch.ctypes.data_as(C.c_void_p))
    return out[:n].copy(), A[:na.value].copy(), ch[:na.value].copy()

if __name__ == "__main__":
    ec = load("e.coli-EC590.fasta.gz"); k12 = load("e.coli-K12.fasta.gz")
    print(len(ec), len(k12))
    s_ec, m_ec = sketch(ec); s_k, m_k = sketch(k12)
    print("seeds", len(s_ec), len(s_k), "markers", len(m_ec), len(m_k), "shared markers", len(np.intersect1d(m_ec, m_k)))
    iv, A, ch = chain(s_k, s_ec)
    print("anchors", len(A), "intervals", len(iv))
    np.save("/tmp/iv.npy", iv)
