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
