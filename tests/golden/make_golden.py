#!/usr/bin/env python3
"""Regenerates tests/golden/ecoli_oracle_golden.json from THIS repo's CPU oracle on the reference's two
FASTA fixtures. These are self-generated goldens that freeze the restatement (oracle/README.md); they are
not outputs of the reference, whose implementation (Rust crate skani v0.3.0) cannot be built or imported
here. ecoli_kat.json, by contrast, holds the reference's own known answers (test_ani.py:28-61)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_fasta_first_record  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
ec = load_fasta_first_record("e.coli-EC590.fasta.gz")
k12 = load_fasta_first_record("e.coli-K12.fasta.gz")
r, q = O.Sketch([ec]), O.Sketch([k12])
g = {}
for name, sk in (("EC590", r), ("K12", q)):
    s = sk.seeds
    g[name] = dict(n_seeds=len(s), n_markers=len(sk.markers),
                   seeds_sha256=hashlib.sha256(np.ascontiguousarray(s).tobytes()).hexdigest(),
                   markers_sha256=hashlib.sha256(sk.markers.tobytes()).hexdigest(),
                   first_seeds=[[int(x) for x in row] for row in s[:8].tolist()],
                   first_markers=[int(x) for x in sk.markers[:8]])
res = O.chain(r, q)
g["pair"] = {k: int(getattr(res, k)) for k in ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")}
g["pair"].update(ani=float(res.ani), af_query=float(res.af_query), af_ref=float(res.af_ref),
                 ani_median=float(O.chain(r, q, median=True).ani), ani_robust=float(O.chain(r, q, robust=True).ani))
g["_generated_by"] = "tests/golden/make_golden.py (this repo's oracle; NOT reference output)"
json.dump(g, open(os.path.join(HERE, "ecoli_oracle_golden.json"), "w"), indent=1)
print(g["pair"])
