#!/usr/bin/env python3
"""tools/dump_pyskani_goldens.py — for a machine that HAS the reference installed (`pip install pyskani`).

Emits the reference's own answers for (a) the two E. coli fixtures under the five flag sets of the reference's
tests (src/pyskani/tests/test_ani.py:28-61) and (b) bench.py's seeded synthetic family genomes, as JSON that can be
committed under tests/golden/ so that seed-level drift of this repository's restated algorithm could one day be pinned
against real pyskani output (today nothing the reference holds pins the seed sets).

It is never imported by the tests here, never shipped to the GPU box and does not touch /root/reference; it needs
`pyskani` and numpy only. Usage:  python tools/dump_pyskani_goldens.py [--out tests/golden/pyskani_goldens.json]
"""
import argparse
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIVERGENCE = (0.0005, 0.002, 0.005, 0.01, 0.02, 0.04, 0.07, 0.10)      # bench.py / SURVEY.md §8(d) family model
FLAG_SETS = {"default": {}, "learned_ani_false": {"learned_ani": False}, "learned_ani_true": {"learned_ani": True},
             "robust": {"robust": True}, "median": {"median": True}}


def first_record(path):
    """first FASTA record of a gzipped file (what the reference's test harness feeds, tests/test_ani.py:16-26)"""
    seq, seen = [], False
    with gzip.open(path, "rt") as f:
        for line in f:
            if line.startswith(">"):
                if seen:
                    break
                seen = True
            elif seen:
                seq.append(line.strip())
    return "".join(seq).encode()


def family(seed, n_members, length):
    """numpy twin of the family model (an iid ancestor, members with independent substitutions at the cycled rates)"""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    anc = rng.integers(0, 4, length, dtype=np.uint8)
    out = []
    for j in range(n_members):
        d = DIVERGENCE[j % len(DIVERGENCE)]
        mut = rng.random(length) < d
        shift = rng.integers(1, 4, length, dtype=np.uint8)
        out.append(lut[np.where(mut, (anc + shift) & 3, anc)].tobytes())
    return out


def hits_of(db, name, seq, **kw):
    return sorted(({"reference": h.reference_name, "identity": h.identity, "query_fraction": h.query_fraction,
                    "reference_fraction": h.reference_fraction} for h in db.query(name, seq, **kw)), key=lambda r: r["reference"])


def main():
    import pyskani
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "pyskani_goldens.json"))
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--length", type=int, default=1_000_000)
    args = ap.parse_args()
    out = {"pyskani_version": getattr(pyskani, "__version__", "?"), "fixtures": {}, "families": {}}
    g = os.path.join(ROOT, "tests", "golden")
    ec, k12 = first_record(os.path.join(g, "e.coli-EC590.fasta.gz")), first_record(os.path.join(g, "e.coli-K12.fasta.gz"))
    db = pyskani.Database()
    db.sketch("EC590", ec)
    for label, kw in FLAG_SETS.items():
        out["fixtures"][label] = hits_of(db, "K12", k12, **kw)
    for c, mc in ((125, 1000), (30, 200)):
        for seed in (11, 12):
            members = family(seed, args.members, args.length)
            db = pyskani.Database(compression=c, marker_compression=mc)
            for j, m in enumerate(members):
                db.sketch(f"m{j}", m)
            key = f"c{c}_mc{mc}_seed{seed}_n{args.members}_L{args.length}"
            out["families"][key] = {label: {f"m{j}": hits_of(db, f"m{j}", m, **kw) for j, m in enumerate(members)}
                                    for label, kw in (("learned_ani_false", {"learned_ani": False}), ("median", {"median": True}))}
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
