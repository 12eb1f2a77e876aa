#!/usr/bin/env python3
"""tools/chunk_unit_sweep.py - what is ONE VALUE of the ANI estimate, and what does its denominator count?  (CPU only, test infrastructure.)

The oracle (oracle/skani_oracle.c) meets three of the four reachable known answers of the reference (/root/reference/src/pyskani/tests/test_ani.py:28-61, EC590 vs K-12:
reference_fraction 0.9246, query_fraction 0.9189, identity with learned_ani=False 0.9946) to the reference's own 4 decimals and misses the fourth, median=True -> 0.9995,
by 9e-5. Round 2 swept the AGGREGATION over the oracle's per-chunk values; this script keeps the oracle's CHAINING rule fixed (tools/chunk_unit_sweep.c is orc_chain's
passes 1-2 with the chunk-window switch; mode 0 is asserted to reproduce the oracle's numbers) and sweeps the UNIT a value is taken over and the seeds its denominator
counts:

    unit      chunk (all kept chains of a 20 kb query window together) | chain (every kept chain its own value)
    window    a chunk starts at its first anchor (the oracle) | fixed grid pos // 20 000
    seeds     between the unit's outermost kept anchors | inside the kept chains' own spans only | every query seed of the window
    denom     S, S - 1, S - 2, S - (number of chains of the unit)
    cover     (window seeds only) a chunk counts if its chains span at least 0 / 25 / 50 / 75 / 90 % of the window
    weights   mean: unweighted | by anchors | by seeds | by span.  median: lower / upper middle element | weighted by seeds
    roles     pyskani's (query K-12 chunked against reference EC590) | swapped

Every variant is scored on all four numbers: both aligned fractions (they depend on the chaining only), the mean (0.9946) and the median (0.9995). A variant "fits" when all
four are within 5e-5. Output: a table (markdown) of every variant's four numbers, best first; `--json` writes them for tests/test_oracle_kat.py's frozen record.

    python tools/chunk_unit_sweep.py [--top 60] [--json tests/golden/chunk_unit_sweep.json]
"""
import argparse
import ctypes as C
import gzip
import itertools
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
KAT = {"af_ref": 0.9246, "af_query": 0.9189, "mean": 0.9946, "median": 0.9995}      # test_ani.py:35-40, 56-61
K, CC, FRAG = 15, 125, 20000


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("window_mode", "fragment", "band", "bp_band", "max_gap", "gap_cost_x4", "min_anchors", "min_score_x4", "ref_overlap")]


chain_dtype = np.dtype([("score", "<i4"), ("q0", "<u4"), ("q1", "<u4"), ("r0", "<u4"), ("r1", "<u4"), ("rc", "<u4"), ("qc", "<u4"), ("nanch", "<u4"), ("order", "<u4"), ("chunk", "<u4")])


def build():
    so = os.path.join(HERE, "_chunk_unit_sweep.so")
    src = os.path.join(HERE, "chunk_unit_sweep.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-o", so, src])
    lib = C.CDLL(so)
    lib.sweep_chains.restype = C.c_int64
    lib.sweep_chains.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    return lib


def first_record(name):
    seq, started = [], False
    with gzip.open(os.path.join(ROOT, "tests", "golden", name), "rt") as f:
        for line in f:
            if line.startswith(">"):
                if started:
                    break
                started = True
                continue
            if line.strip():
                seq.append(line.strip())
    return "".join(seq).encode("ascii")


def chains_of(lib, qseeds, rseeds, window_mode, **over):
    p = Params(window_mode, FRAG, max(1, min(100, 2500 // CC)), 2500, 300, 2, 3, 180, 1)
    for k, v in over.items():
        setattr(p, k, v)
    cap = 1 << 16
    out = np.zeros(cap, chain_dtype)
    clo, cqc = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
    nch, na = C.c_uint64(), C.c_uint64()
    q = np.ascontiguousarray(qseeds)
    r = np.ascontiguousarray(rseeds)
    n = lib.sweep_chains(q.ctypes.data, len(q), r.ctypes.data, len(r), C.byref(p), out.ctypes.data, cap, clo.ctypes.data, cqc.ctypes.data, cap, C.byref(nch), C.byref(na))
    assert 0 <= n <= cap and nch.value <= cap
    return out[:n].copy(), clo[:nch.value].copy(), cqc[:nch.value].copy(), na.value


def seeds_in(qpos, lo, hi):
    """query seeds with lo <= pos <= hi (one contig: the fixtures' first records)"""
    return np.searchsorted(qpos, hi, side="right") - np.searchsorted(qpos, lo, side="left")


def wmedian(v, w):
    o = np.argsort(v, kind="stable")
    cw = np.cumsum(w[o])
    return v[o][np.searchsorted(cw, cw[-1] / 2.0)]


def evaluate(chains, chunk_lo, qpos, qlen, rlen, window_mode):
    """every (unit, seeds, denom, cover, weights) reading over one set of kept chains -> list of dict rows"""
    rows = []
    span = (chains["q1"] - chains["q0"]).astype(np.int64)
    covered = int((span + 1 + 2 * CC).sum())
    afq, afr = min(1.0, covered / qlen), min(1.0, covered / rlen)
    s_chain = seeds_in(qpos, chains["q0"], chains["q1"]).astype(np.int64)
    ck = chains["chunk"]
    ids = np.unique(ck)
    # per chunk: anchors, outermost span, sum of own-span seeds, number of chains, spans' sum
    a_c = np.array([chains["nanch"][ck == i].sum() for i in ids], np.int64)
    l_c = np.array([chains["q0"][ck == i].min() for i in ids], np.int64)
    r_c = np.array([chains["q1"][ck == i].max() for i in ids], np.int64)
    own_c = np.array([s_chain[ck == i].sum() for i in ids], np.int64)
    n_c = np.array([(ck == i).sum() for i in ids], np.int64)
    sp_c = np.array([span[ck == i].sum() for i in ids], np.int64)
    outer_c = seeds_in(qpos, l_c, r_c).astype(np.int64)
    wlo = chunk_lo[ids].astype(np.int64)
    whi = np.minimum(wlo + FRAG - (0 if window_mode == 0 else 1), qlen - 1)
    win_c = seeds_in(qpos, wlo, whi).astype(np.int64)
    wlen = (whi - wlo + 1).astype(np.float64)
    units = {"chunk": {"outer": (a_c, outer_c, n_c, r_c - l_c), "own": (a_c, own_c, n_c, sp_c), "window": (a_c, win_c, n_c, sp_c)},
             "chain": {"own": (chains["nanch"].astype(np.int64), s_chain, np.ones(len(chains), np.int64), span)}}
    for unit, by_seeds in units.items():
        for seeds, (a, s, nint, sp) in by_seeds.items():
            covers = (0.0, 0.25, 0.5, 0.75, 0.9) if seeds == "window" else (0.0,)
            for cover in covers:
                keep = (sp_c / wlen >= cover) if seeds == "window" else np.ones(len(a), bool)
                if keep.sum() < 3:
                    continue
                for denom in ("S", "S-1", "S-2", "S-n"):
                    d = {"S": s, "S-1": s - 1, "S-2": s - 2, "S-n": s - nint}[denom]
                    d = np.maximum(d, 1)[keep].astype(np.float64)
                    aa = a[keep].astype(np.float64)
                    v = np.minimum(1.0, aa / d) ** (1.0 / K)
                    w_all = {"none": np.ones(len(v)), "anchors": aa, "seeds": d, "span": np.maximum(sp[keep], 1).astype(np.float64)}
                    vs = np.sort(v)
                    meds = {"lower": vs[(len(vs) - 1) // 2], "upper": vs[len(vs) // 2], "wseeds": wmedian(v, d)}
                    for wname, w in w_all.items():
                        mean = float((v * w).sum() / w.sum())
                        for mname, med in meds.items():
                            rows.append({"unit": unit, "window": "first-anchor" if window_mode == 0 else "grid", "seeds": seeds, "denom": denom, "cover": cover, "mean_w": wname, "median": mname,
                                         "af_ref": afr, "af_query": afq, "mean_val": mean, "median_val": float(med), "n_values": int(len(v)), "median_S": float(np.median(d))})
    return rows


def sketch_salted(lib, seq, salt):
    cap = len(seq) // CC * 2 + 4096
    out = np.zeros(cap, O_SEED)
    n = lib.sweep_sketch(seq, len(seq), CC, K, C.c_uint64(salt), out.ctypes.data, cap)
    assert 0 <= n <= cap
    return out[:n].copy()


O_SEED = np.dtype([("kmer", "<u4"), ("pos", "<u4"), ("contig", "<u4"), ("canon", "<u4")])
VKEY = ("roles", "window", "unit", "seeds", "denom", "cover", "mean_w", "median")


def all_rows(lib, seeds_ec, seeds_k12, len_ec, len_k12, want=None):
    """every variant's four numbers for one pair of seed sets (pyskani's roles: reference EC590, query K-12)"""
    rows = []
    for roles in ("pyskani", "swapped"):
        qs, rs = (seeds_k12, seeds_ec) if roles == "pyskani" else (seeds_ec, seeds_k12)
        qlen, rlen = (len_k12, len_ec) if roles == "pyskani" else (len_ec, len_k12)
        qpos = qs["pos"].astype(np.int64)
        for wm in (0, 1):
            chains, clo, cqc, na = chains_of(lib, qs, rs, wm)
            if want is not None and roles == "pyskani" and wm == 0:      # the harness IS the oracle's chaining: same anchors, kept chains, covered bases
                assert na == want.n_anchors and len(chains) == want.n_intervals and int(chains["nanch"].sum()) == want.sum_chain_anchors
                assert int(((chains["q1"] - chains["q0"]).astype(np.int64) + 1 + 2 * CC).sum()) == want.covered_query
            for row in evaluate(chains, clo, qpos, qlen, rlen, wm):
                # the aligned fractions are reported against pyskani's roles: reference_fraction = covered / len(EC590), query_fraction = covered / len(K-12)
                if roles == "swapped":
                    row["af_ref"], row["af_query"] = row["af_query"], row["af_ref"]
                row["roles"] = roles
                rows.append(row)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--json", default=None)
    ap.add_argument("--salts", type=int, default=0, help="also re-draw the FracMinHash sample this many times (k-mer xor salt before the hash) and place the known answers in every variant's distribution")
    args = ap.parse_args()
    from oracle import oracle as O
    O.build()
    lib = build()
    lib.sweep_sketch.restype = C.c_int64
    lib.sweep_sketch.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64]
    ec, k12 = first_record("e.coli-EC590.fasta.gz"), first_record("e.coli-K12.fasta.gz")
    sk_ref, sk_q = O.Sketch([ec]), O.Sketch([k12])
    want = O.chain(sk_ref, sk_q)
    assert np.array_equal(sketch_salted(lib, ec, 0), sk_ref.seeds) and np.array_equal(sketch_salted(lib, k12, 0), sk_q.seeds)      # salt 0 = the oracle's seed sets
    rows = all_rows(lib, sk_ref.seeds, sk_q.seeds, len(ec), len(k12), want)
    for r in rows:
        r["err"] = {"af_ref": abs(r["af_ref"] - KAT["af_ref"]), "af_query": abs(r["af_query"] - KAT["af_query"]), "mean": abs(r["mean_val"] - KAT["mean"]), "median": abs(r["median_val"] - KAT["median"])}
        r["worst"] = max(r["err"].values())
        r["fits"] = r["worst"] < 5e-5
    rows.sort(key=lambda r: r["worst"])
    is_cur = lambda r: r["roles"] == "pyskani" and r["window"] == "first-anchor" and r["unit"] == "chunk" and r["seeds"] == "outer" and r["denom"] == "S-1" and r["mean_w"] == "none" and r["median"] == "upper"
    cur = [r for r in rows if is_cur(r)][0]
    assert abs(cur["mean_val"] - want.ani) < 1e-6, (cur["mean_val"], want.ani)
    print(f"{len(rows)} variants; {sum(r['fits'] for r in rows)} fit all four numbers at 5e-5; the oracle's rule: mean {cur['mean_val']:.6f} median {cur['median_val']:.6f} "
          f"AF {cur['af_ref']:.6f} / {cur['af_query']:.6f} (worst error {cur['worst']:.2e})\n")
    print("| roles | window | unit | seeds | denom | cover | mean weights | median | n | median S | AF ref | AF query | mean | median | worst err |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows[:args.top]:
        print(f"| {r['roles']} | {r['window']} | {r['unit']} | {r['seeds']} | {r['denom']} | {r['cover']} | {r['mean_w']} | {r['median']} | {r['n_values']} | {r['median_S']:.0f} | "
              f"{r['af_ref']:.5f} | {r['af_query']:.5f} | {r['mean_val']:.5f} | {r['median_val']:.5f} | {r['worst']:.1e} |")
    salted = None
    if args.salts > 0:
        # Re-draw the seed sample: the known answers are 4-decimal roundings of functions of WHICH 1-in-125 k-mers are seeds, and the seed set is the one thing the reference
        # exposes nothing about (sketch.rs:16-31). For every variant: mean and standard deviation of its four numbers over the samples, and where the known answers sit in them
        # (z = (known answer - mean) / sd). A variant whose |z| is small everywhere is CONSISTENT with the reference; the 4th decimal of any of them is decided by the sample.
        acc = {}
        rng = np.random.default_rng(2026)
        fit_count = {}
        for si in range(args.salts):
            salt = int(rng.integers(1, 1 << 62))
            for r in all_rows(lib, sketch_salted(lib, ec, salt), sketch_salted(lib, k12, salt), len(ec), len(k12)):
                key = tuple(r[k] for k in VKEY)
                acc.setdefault(key, []).append((r["af_ref"], r["af_query"], r["mean_val"], r["median_val"]))
                ok = max(abs(r["af_ref"] - KAT["af_ref"]), abs(r["af_query"] - KAT["af_query"]), abs(r["mean_val"] - KAT["mean"]), abs(r["median_val"] - KAT["median"])) < 5e-5
                fit_count[key] = fit_count.get(key, 0) + int(ok)
        kat = np.array([KAT["af_ref"], KAT["af_query"], KAT["mean"], KAT["median"]])
        salted = []
        for key, vals in acc.items():
            v = np.array(vals)
            mu, sd = v.mean(0), v.std(0, ddof=1)
            z = (kat - mu) / np.maximum(sd, 1e-9)
            salted.append(dict(zip(VKEY, key), mu=mu.tolist(), sd=sd.tolist(), z=z.tolist(), zmax=float(np.abs(z).max()), n=len(vals), fits=fit_count[key],
                               within=[float((np.abs(v[:, j] - kat[j]) < 5e-5).mean()) for j in range(4)]))
        salted.sort(key=lambda r: r["zmax"])
        cs = [r for r in salted if is_cur(r)][0]
        print(f"\n{args.salts} re-drawn seed samples. The oracle's rule: AF ref {cs['mu'][0]:.5f} +- {cs['sd'][0]:.5f}, AF query {cs['mu'][1]:.5f} +- {cs['sd'][1]:.5f}, mean {cs['mu'][2]:.5f} +- {cs['sd'][2]:.5f}, "
              f"median {cs['mu'][3]:.5f} +- {cs['sd'][3]:.5f}; z of the known answers {', '.join(f'{x:+.2f}' for x in cs['z'])}; samples within 5e-5 of each known answer: "
              f"{', '.join(f'{x:.0%}' for x in cs['within'])}; of all four at once: {cs['fits']} of {cs['n']}\n")
        print("| roles | window | unit | seeds | denom | cover | mean weights | median | AF ref (mean +- sd) | mean | median | z AF | z mean | z median | all four within 5e-5 |")
        print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
        for r in salted[:args.top]:
            print(f"| {r['roles']} | {r['window']} | {r['unit']} | {r['seeds']} | {r['denom']} | {r['cover']} | {r['mean_w']} | {r['median']} | {r['mu'][0]:.5f} +- {r['sd'][0]:.5f} | {r['mu'][2]:.5f} +- {r['sd'][2]:.5f} | "
                  f"{r['mu'][3]:.5f} +- {r['sd'][3]:.5f} | {r['z'][0]:+.2f} | {r['z'][2]:+.2f} | {r['z'][3]:+.2f} | {r['fits']} / {r['n']} |")
    if args.json:      # (the record tests/test_oracle_kat.py and oracle/README.md cite: the best 120 of each ranking and the oracle's own rule)
        keep = ("roles", "window", "unit", "seeds", "denom", "cover", "mean_w", "median", "n_values", "median_S", "af_ref", "af_query", "mean_val", "median_val", "worst", "fits")
        rnd = lambda x: [round(y, 6) for y in x] if isinstance(x, list) else (round(x, 6) if isinstance(x, float) else x)
        with open(args.json, "w") as f:
            json.dump({"kat": KAT, "n_variants": len(rows), "n_fit": sum(r["fits"] for r in rows), "oracle_rule": {k: rnd(cur[k]) for k in keep},
                       "rows_best": [{k: rnd(r[k]) for k in keep} for r in rows[:120]],
                       "salts": args.salts, "salted_oracle_rule": {k: rnd(v) for k, v in cs.items()} if salted else None,
                       "salted_best": [{k: rnd(v) for k, v in r.items()} for r in salted[:120]] if salted else None,
                       "salted_any_fit": max(r["fits"] for r in salted) if salted else None}, f, indent=0)
    return rows, salted


if __name__ == "__main__":
    main()
