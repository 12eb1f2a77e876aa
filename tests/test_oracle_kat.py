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


@pytest.mark.parametrize("mode", ["no_learned_ani", "median"])
def test_identity_near_reference_kat(oracle, pair, mode):
    """Raw-ANI KATs reachable without skani's GBDT weights. The restatement is within
    KAT['identity_tolerance_restatement'] of them, NOT within the reference's 4 decimals:
    parity of the ANI value is only partially pinned (oracle/README.md)."""
    ref, q = pair
    res = oracle.chain(ref, q, median=(mode == "median"))
    want = KAT["identity"][mode]
    assert abs(res.ani - want) < KAT["identity_tolerance_restatement"]


@pytest.mark.xfail(strict=True, reason="skani source absent: chunk ANI aggregation restated, not recovered to 4 decimals")
def test_identity_exact_reference_kat(oracle, pair):
    ref, q = pair
    assert abs(oracle.chain(ref, q).ani - KAT["identity"]["no_learned_ani"]) < 5e-5


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
