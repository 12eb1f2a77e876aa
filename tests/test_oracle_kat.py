"""CPU: the oracle against the reference's own known-answer test
(/root/reference/src/pyskani/tests/test_ani.py:28-61, fixtures copied to tests/golden/)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

KAT = json.load(open(os.path.join(GOLDEN, "ecoli_kat.json")))


@pytest.fixture(scope="module")
def pair(oracle, ecoli):
    ec, k12 = ecoli
    return oracle.Sketch([ec]), oracle.Sketch([k12])


def test_fixture_lengths(ecoli):
    ec, k12 = ecoli
    assert len(ec) == 4617703 and len(k12) == 4646332
    assert set(ec) <= set(b"ACGT") and set(k12) <= set(b"ACGT")


def test_screen_passes(oracle, pair):
    ref, q = pair
    ok, shared = oracle.screen(q, ref, 0.80, True)
    assert ok and shared > 0.5 * min(len(ref.markers), len(q.markers))


def test_aligned_fractions_match_reference_kat(oracle, pair):
    """reference_fraction / query_fraction to the reference's own tolerance (places=4)."""
    ref, q = pair
    res = oracle.chain(ref, q)
    assert abs(res.af_ref - KAT["reference_fraction"]) < 5e-5
    assert abs(res.af_query - KAT["query_fraction"]) < 5e-5


def test_raw_identity_matches_reference_kat(oracle, pair):
    """test_no_learned_ani (test_ani.py:35-40): identity 0.9946 to the reference's own 4 decimals."""
    ref, q = pair
    res = oracle.chain(ref, q)
    assert abs(res.ani - KAT["identity"]["no_learned_ani"]) < 5e-5


def test_median_identity_near_reference_kat(oracle, pair):
    """test_median (test_ani.py:56-61): the restatement gives 0.99959 against 0.9995 — inside
    BASELINE.json's 1e-4 tolerance, outside the reference's 4 decimals (oracle/README.md)."""
    ref, q = pair
    res = oracle.chain(ref, q, median=True)
    assert abs(res.ani - KAT["identity"]["median"]) < KAT["restatement_tolerance"]["median"]


@pytest.mark.xfail(strict=True, reason="skani source absent: median identity restated to 9e-5, not to the reference's 5e-5")
def test_median_identity_exact_reference_kat(oracle, pair):
    ref, q = pair
    assert abs(oracle.chain(ref, q, median=True).ani - KAT["identity"]["median"]) < 5e-5


def test_robust_raw_identity_is_recorded(oracle, pair):
    """test_robust's KAT (0.9977) includes skani's learned-ANI model, which is unobtainable here; the raw
    trimmed mean of the restatement is pinned by the self-generated golden only."""
    ref, q = pair
    assert abs(oracle.chain(ref, q, robust=True).ani - 0.99782) < 2e-5


def test_golden_counts(oracle, pair):
    """Self-generated goldens (tests/golden/ecoli_oracle_golden.json) freeze the restatement."""
    g = json.load(open(os.path.join(GOLDEN, "ecoli_oracle_golden.json")))
    ref, q = pair
    import hashlib
    for name, sk in (("EC590", ref), ("K12", q)):
        s = sk.seeds
        assert len(s) == g[name]["n_seeds"] and len(sk.markers) == g[name]["n_markers"]
        assert hashlib.sha256(np.ascontiguousarray(s).tobytes()).hexdigest() == g[name]["seeds_sha256"]
        assert hashlib.sha256(sk.markers.tobytes()).hexdigest() == g[name]["markers_sha256"]
    res = oracle.chain(ref, q)
    for key in ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds"):
        assert getattr(res, key) == g["pair"][key], key


def test_hash_known_values(oracle):
    lib = oracle.lib()
    # mm_hash64 restated with `!(key + (key << 21))` as the Rust expression parses
    def ref_hash(key):
        M = (1 << 64) - 1
        key = (~(key + (key << 21))) & M
        key ^= key >> 24
        key = (key + (key << 3) + (key << 8)) & M
        key ^= key >> 14
        key = (key + (key << 2) + (key << 4)) & M
        key ^= key >> 28
        key = (key + (key << 31)) & M
        return key
    for v in (0, 1, 0x3FFFFFFF, 123456789, (1 << 42) - 1):
        assert lib.orc_mm_hash64(v) == ref_hash(v)


def test_aggregation_variants_negative_result(oracle, pair):
    """is there a natural reading of `median` / `robust` that lands on 0.9995 / 0.9977 (test_ani.py:49-61)
    while the seed-count rule that meets the raw mean (0.9946) is kept? Recorded negative result (oracle/README.md,
    oracle/explore/t27.py, t28.py): with denominators `seeds - 1` every median variant (lower / upper / mean of the two
    middle values, weighted by anchors or seeds) gives 0.99959-0.99964, and every 10-90 % trimming variant (floor / ceil /
    round of the cut indices, trimmed by count or by weight, weighted or unweighted mean) gives 0.99781-0.99819. The
    robust KAT additionally includes skani's regression model in the reference (lib.rs:611-614 does not pass `robust`)."""
    ref, q = pair
    oracle.chain(ref, q)
    ch = oracle.last_chunks()
    a, s = ch["anchors"].astype(float), ch["seeds"].astype(float)
    v = np.sort(np.minimum(1.0, a / np.maximum(s - 1, 1)) ** (1 / 15))
    n = len(v)
    assert abs(v.mean() - 0.9946) < 5e-5
    meds = [v[n // 2], v[(n - 1) // 2], 0.5 * (v[n // 2] + v[(n - 1) // 2])]
    assert all(5e-5 < abs(m - 0.9995) < 1e-4 for m in meds)
    trims = [v[lo:hi].mean() for lo in (int(np.floor(.1 * n)), int(np.ceil(.1 * n))) for hi in (int(np.floor(.9 * n)), int(np.ceil(.9 * n)))]
    assert all(5e-5 < abs(t - 0.9977) < 3e-4 for t in trims)


def test_chunk_unit_sweep_negative_result(oracle):
    """with the current chaining rule fixed, is there a reading of what ONE VALUE of the estimate is (per chunk / per kept chain; seeds between the outermost
    anchors / inside chain spans only / of the whole window; S, S-1, S-2, S-chains; fixed grid / first-anchor windows; weighted / unweighted; both roles) that meets all four
    reachable known answers (test_ani.py:35-40, 56-61) at 5e-5? Recorded negative result (tools/chunk_unit_sweep.py, oracle/README.md round 6): none of 1 536, and the reason -
    over re-drawn FracMinHash samples (same algorithm, salted hash) the oracle's own rule moves by sd 1.5e-3 (AF), 5e-4 (mean), 1.6e-4 (median): the 4th decimal of every known
    answer is decided by the seed sample, which the reference does not expose; the oracle's rule is consistent with all four (|z| < 2)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("chunk_unit_sweep", os.path.join(os.path.dirname(GOLDEN), os.pardir, "tools", "chunk_unit_sweep.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    rec = json.load(open(os.path.join(GOLDEN, "chunk_unit_sweep.json")))
    assert rec["n_variants"] >= 1500 and rec["n_fit"] == 0 and rec["salted_any_fit"] <= 1 and rec["salts"] == 200
    argv = sys.argv
    sys.argv = ["chunk_unit_sweep.py", "--top", "0", "--salts", "24"]
    try:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            rows, salted = S.main()
    finally:
        sys.argv = argv
    assert len(rows) == rec["n_variants"] and not any(r["fits"] for r in rows)
    best = rows[0]
    assert abs(best["worst"] - rec["rows_best"][0]["worst"]) < 1e-6 and 5e-5 < best["worst"] < 1e-4
    cur = [r for r in salted if r["roles"] == "pyskani" and r["window"] == "first-anchor" and r["unit"] == "chunk" and r["seeds"] == "outer" and r["denom"] == "S-1" and r["mean_w"] == "none" and r["median"] == "upper"][0]
    assert all(abs(z) < 2.5 for z in cur["z"]), cur["z"]                       # the oracle's rule is consistent with every known answer ...
    assert cur["sd"][0] > 5e-4 and cur["sd"][2] > 1.5e-4 and cur["sd"][3] > 5e-5, cur["sd"]      # ... whose 4th decimals the seed sample decides (sd >= the 5e-5 tolerance)
    chain_rows = [r for r in salted if r["unit"] == "chain"]
    assert chain_rows and all(abs(r["z"][2]) > 3 for r in chain_rows)          # per-chain values are rejected on the mean
    own_rows = [r for r in salted if r["unit"] == "chunk" and r["seeds"] == "own"]
    assert own_rows and all(abs(r["z"][2]) > 3 for r in own_rows)              # so are seed counts inside the chains' own spans only


def test_oracle_model_matches_python_evaluator(oracle):
    """The oracle's restatement of gbdt 0.1.3 inference against the 20-line evaluator in tests/gbdt_util.py."""
    import gbdt_util as G
    rng = np.random.default_rng(5)
    trees = G.random_trees(rng, n_trees=7, depth=4)
    m = oracle.Model(trees, 98.5, 0.1, list(range(9)))
    for _ in range(200):
        row = [rng.uniform(lo, hi) for lo, hi in [(97, 100), (0, 1)] + [(1e3, 6e6)] * 6 + [(1e3, 3e4)]]
        if rng.random() < 0.3:
            row[int(rng.integers(0, 9))] = float(G.UNKNOWN)
        row = [float(np.float32(x)) for x in row]
        assert m.predict(row) == G.predict(trees, 98.5, 0.1, row)


def test_oracle_learned_ani_and_default_rule(oracle, pair):
    """lib.rs:611-614: with a model, learned_ani=None applies it when c >= 70 and not median; False never does."""
    import gbdt_util as G
    ref, q = pair
    trees = G.random_trees(np.random.default_rng(9), n_trees=3, depth=3)
    m = oracle.Model(trees, 99.0, 0.5, list(range(9)))
    raw = oracle.chain(ref, q)
    on = oracle.chain(ref, q, learned_ani=None, model=m)
    assert on.learned == 1 and abs(on.ani_raw - raw.ani) == 0 and on.ani != raw.ani
    lq = [4646332.0] * 3
    lr = [4617703.0] * 3
    row = [np.float32(raw.ani) * np.float32(100), np.float32(on.ani_std) * np.float32(100)] + lq + lr + [np.float32(raw.covered_query) / np.float32(raw.n_intervals)]
    want = min(1.0, max(0.0, float(np.float32(G.predict(trees, 99.0, 0.5, row)) * np.float32(0.01))))
    assert abs(on.ani - want) < 1e-7
    assert oracle.chain(ref, q, learned_ani=None, model=m, median=True).learned == 0
    assert oracle.chain(ref, q, learned_ani=False, model=m).learned == 0


def test_oracle_query_dedups_names(oracle):
    """lib.rs:51-55 + 616-637: a name sketched twice gives ONE hit, chained against the later sketch."""
    from conftest import random_genome, mutate
    rng = np.random.default_rng(3)
    g = random_genome(rng, 120_000)
    a, b = oracle.Sketch([mutate(rng, g, 0.05)]), oracle.Sketch([mutate(rng, g, 0.01)])
    q = oracle.Sketch([g])
    hits = oracle.query([("x", a), ("y", b), ("x", b)], q)
    assert [h[0] for h in hits] == ["y", "x"] or [h[0] for h in hits] == ["x", "y"]
    byname = dict(hits)
    assert byname["x"].ani == byname["y"].ani == oracle.chain(b, q).ani
