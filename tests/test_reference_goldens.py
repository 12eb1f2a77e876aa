"""Consumer of tools/dump_pyskani_goldens.py: the day someone runs the dump on a machine that has the reference installed
(`pip install pyskani`) and commits tests/golden/pyskani_goldens.json, the oracle (CPU) and the HIP path (GPU) are held to every
triple the reference produced - the two E. coli fixtures under the reference's flag sets (src/pyskani/tests/test_ani.py:28-61) and
bench.py's seeded family genomes - at the reference's own tolerance (4 decimals, |d| < 5e-5), with no new code. Absent file: skipped,
with the reason. Flag sets that go through skani's embedded regression model (default, learned_ani=True, robust: lib.rs:611-614) are
compared only when a model is loaded ($PSK_MODEL_PATH); the raw ones (learned_ani=False, median) always."""
import json
import os
import sys

import pytest

from conftest import GOLDEN, ROOT

PATH = os.path.join(GOLDEN, "pyskani_goldens.json")
TOL = 5e-5          # the reference asserts 4 decimal places (test_ani.py:31-33)
RAW_SETS = {"learned_ani_false": dict(learned_ani=False), "median": dict(median=True)}
MODEL_SETS = {"default": dict(), "learned_ani_true": dict(learned_ani=True), "robust": dict(robust=True)}


def goldens():
    if not os.path.exists(PATH):
        pytest.skip("tests/golden/pyskani_goldens.json is absent: run tools/dump_pyskani_goldens.py where pyskani is installed "
                    "(the reference cannot be built in this image: Rust crate, un-vendored skani v0.3.0)")
    return json.load(open(PATH))


def tool():
    sys.path.insert(0, ROOT)
    from tools import dump_pyskani_goldens as T
    return T


def compare(got, want, what):
    """got / want: {reference name: (identity, query_fraction, reference_fraction)}"""
    assert set(got) == set(want), (what, set(got) ^ set(want))
    for name, w in want.items():
        for a, b, field in zip(got[name], w, ("identity", "query_fraction", "reference_fraction")):
            assert abs(a - b) < TOL, (what, name, field, a, b)


def as_triples(rows):
    return {r["reference"]: (r["identity"], r["query_fraction"], r["reference_fraction"]) for r in rows}


def family_cases(G):
    T = tool()
    for key, per_label in G.get("families", {}).items():
        parts = dict(p[0] == "c" and ("c", p[1:]) or p[:2] == "mc" and ("mc", p[2:]) or p[:4] == "seed" and ("seed", p[4:]) or p[0] == "n" and ("n", p[1:]) or ("L", p[1:]) for p in key.split("_"))
        c, mc, seed, n, L = (int(parts[k]) for k in ("c", "mc", "seed", "n", "L"))
        yield key, c, mc, T.family(seed, n, L), per_label


def test_golden_consumer_parses_its_own_key_format():
    """the consumer's reading of the dump's family keys, checked without the file (so that a typo here cannot hide behind the skip)"""
    fake = {"families": {"c30_mc200_seed11_n2_L5000": {"median": {}}}}
    (key, c, mc, members, per_label), = list(family_cases(fake))
    assert (c, mc, len(members), len(members[0])) == (30, 200, 2, 5000) and "median" in per_label


def oracle_consumer(G, oracle, fixtures=True):
    T = tool()
    n = 0
    if fixtures:
        ec, k12 = (T.first_record(os.path.join(GOLDEN, f)) for f in ("e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz"))
        refs = [("EC590", oracle.Sketch([ec]))]
        q = oracle.Sketch([k12])
        for label, kw in RAW_SETS.items():
            got = {n: (r.ani, r.af_query, r.af_ref) for n, r in oracle.query(refs, q, **kw)}
            compare(got, as_triples(G["fixtures"][label]), f"fixtures/{label}")
    for key, c, mc, members, per_label in family_cases(G):
        sk = [(f"m{j}", oracle.Sketch([m], c=c, marker_c=mc)) for j, m in enumerate(members)]
        for label, kw in RAW_SETS.items():
            for j in range(len(members)):
                got = {nm: (r.ani, r.af_query, r.af_ref) for nm, r in oracle.query(sk, sk[j][1], **kw)}
                compare(got, as_triples(per_label[label][f"m{j}"]), f"{key}/{label}/m{j}")
                n += len(got)
    return n


def test_oracle_against_reference_goldens(oracle):
    oracle_consumer(goldens(), oracle)


def test_consumer_end_to_end_on_a_self_made_file(oracle):
    """plumbing check, NOT parity: a goldens object in the dump's format whose triples come from the oracle itself must pass the
    consumer, and one with a triple moved by 1e-4 must fail it"""
    T = tool()
    key = "c30_mc200_seed11_n3_L60000"
    members = T.family(11, 3, 60000)
    sk = [(f"m{j}", oracle.Sketch([m], c=30, marker_c=200)) for j, m in enumerate(members)]
    rows = lambda j, kw: sorted(({"reference": n, "identity": r.ani, "query_fraction": r.af_query, "reference_fraction": r.af_ref}
                                 for n, r in oracle.query(sk, sk[j][1], **kw)), key=lambda r: r["reference"])
    G = {"families": {key: {label: {f"m{j}": rows(j, kw) for j in range(3)} for label, kw in RAW_SETS.items()}}}
    assert oracle_consumer(G, oracle, fixtures=False) >= 12
    G["families"][key]["median"]["m1"][0]["identity"] += 1e-4
    with pytest.raises(AssertionError):
        oracle_consumer(G, oracle, fixtures=False)


@pytest.mark.gpu
def test_hip_path_against_reference_goldens():
    G = goldens()
    T = tool()
    import pyskani_amd as psk
    ec, k12 = (T.first_record(os.path.join(GOLDEN, f)) for f in ("e.coli-EC590.fasta.gz", "e.coli-K12.fasta.gz"))
    db = psk.Database()
    db.sketch("EC590", ec)
    sets = dict(RAW_SETS)
    if db._model is not None:
        sets.update(MODEL_SETS)
    for label, kw in sets.items():
        got = {h.reference_name: (h.identity, h.query_fraction, h.reference_fraction) for h in db.query("K12", k12, **kw)}
        compare(got, as_triples(G["fixtures"][label]), f"fixtures/{label}")
    for key, c, mc, members, per_label in family_cases(G):
        db = psk.Database(compression=c, marker_compression=mc)
        for j, m in enumerate(members):
            db.sketch(f"m{j}", m)
        for label, kw in RAW_SETS.items():
            for j, m in enumerate(members):
                got = {h.reference_name: (h.identity, h.query_fraction, h.reference_fraction) for h in db.query(f"m{j}", m, **kw)}
                compare(got, as_triples(per_label[label][f"m{j}"]), f"{key}/{label}/m{j}")
