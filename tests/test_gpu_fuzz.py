"""GPU: randomised parity sweep. Random parameters (k, c, marker_c), contig structure, divergence, repeats and
strand flips; every integer intermediate of chaining and both sketches must equal the oracle's."""
import os

import numpy as np
import pytest

from conftest import mutate, random_genome

pytestmark = pytest.mark.gpu

INT_FIELDS = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")
COMP = bytes.maketrans(b"ACGT", b"TGCA")


@pytest.fixture(scope="module")
def psk():
    import pyskani_amd
    return pyskani_amd


def _case(rng):
    k = int(rng.integers(9, 17))
    c = int(rng.choice([8, 20, 30, 60, 100, 105, 125, 200, 400]))
    mc = int(c * rng.choice([2, 4, 8]))
    L = int(rng.integers(30000, 400000))
    anc = random_genome(rng, L)
    if rng.random() < 0.5:                      # plant a repeat family
        rep = random_genome(rng, int(rng.integers(300, 6000)))
        pieces, pos = [], 0
        for _ in range(int(rng.integers(2, 7))):
            cut = int(rng.integers(pos, L))
            pieces.append(anc[pos:cut]); pieces.append(mutate(rng, rep, rng.uniform(0, 0.03))); pos = cut
        pieces.append(anc[pos:])
        anc = b"".join(pieces)
    def split(seq):
        n = int(rng.integers(1, 6))
        cuts = sorted(int(x) for x in rng.integers(0, len(seq), size=n - 1))
        parts = [seq[a:b] for a, b in zip([0] + cuts, cuts + [len(seq)])]
        return [p[::-1].translate(COMP) if rng.random() < 0.3 else p for p in parts]
    ref = split(mutate(rng, anc, rng.uniform(0, 0.02)))
    lo = int(rng.integers(0, len(anc) // 2)); hi = int(rng.integers(lo + len(anc) // 4, len(anc)))
    qry = split(mutate(rng, anc[lo:hi], rng.uniform(0, 0.08), rng.uniform(0, 0.001)))
    return k, c, mc, ref, qry


@pytest.mark.parametrize("seed", range(int(os.environ.get("PSK_FUZZ_SEEDS", "24"))))   # PSK_FUZZ_SEEDS=1000 for a long sweep
def test_random_pairs_match_oracle(psk, oracle, seed, monkeypatch):
    rng = np.random.default_rng(5000 + seed)
    if seed % 4 == 0:
        monkeypatch.setenv("PSK_LANE_XTREES", "1")  # sixteen tree slots per chunk in the lane DP (twelve in LDS): the Gb-scale variant, forced on small pairs
        monkeypatch.setenv("PSK_CHAIN_LANE", "64")  # (a small launch would take the four-lanes-per-chunk kernel otherwise)
        if seed % 8 == 4:
            monkeypatch.setenv("PSK_CHUNK_HOPS", "1"); monkeypatch.setenv("PSK_HOPS_ITEMS", "1")      # ... and the Gb-scale chunk table: waves hopping over the items' offsets
    if seed % 4 == 3:
        monkeypatch.setenv("PSK_JOIN", "wide")      # the fallback join format gets a quarter of the sweep
        monkeypatch.setenv("PSK_CHAIN_WAVE_REG", "0")   # ... with the throughput DP kernels of wide bands (c < 105) instead of the small launch's register-window one
        if seed % 8 == 7:
            monkeypatch.setenv("PSK_CHUNK_HOPS", "1"); monkeypatch.setenv("PSK_HOPS_ITEMS", "1")
    if seed % 4 == 1:
        monkeypatch.setenv("PSK_JOIN_PAIRS", "1")   # ... and the pair-major join another,
        monkeypatch.setenv("PSK_EMIT_EXPAND", "1")  # with the anchor-major emit of Gb-scale pairs behind it
        if seed % 8 == 1:
            monkeypatch.setenv("PSK_PROBE", "1")    # ... through the database-wide seed index (the metagenome join) or, every other time, the references' probe tables
            if seed % 16 == 1:
                monkeypatch.setenv("PSK_GSI_JOIN", "0")
            elif seed % 32 == 9:
                monkeypatch.setenv("PSK_GSI_ONEPASS", "0")      # ... the index join with its count pass (without: anchors placed at the pairs' item offsets, one walk)
            elif seed % 32 == 25:
                monkeypatch.setenv("PSK_GSI_STAGE", "0")        # ... every anchor its own 16-byte store (default: an even-indexed anchor waits in LDS for its neighbour)
    if seed % 4 == 2:
        monkeypatch.setenv("PSK_EMIT_PAIRS", "1")   # ... and the one-workgroup-per-pair emit (join's pair totals) another,
        monkeypatch.setenv("PSK_CHUNK_HOPS", "0")   # chunk table included
        monkeypatch.setenv("PSK_SKETCH_SMALL", "0") # ... behind the count-then-allocate sketch pipeline (a single genome takes the one-synchronisation path otherwise)
        if seed % 8 == 6:
            monkeypatch.setenv("PSK_GSI_SLICE", "1")    # ... or (every other time) the seed-index join by (query, slice) waves - the all-vs-all default - forced on one pair:
            if seed % 16 == 14:                         # repeats (several anchors per seed and pair), strands, contigs, slices of 512 seeds; staged whole-line stores or plain ones
                monkeypatch.setenv("PSK_GSL_STAGE", "0")
    for _ in range(4):
        k, c, mc, ref, qry = _case(rng)
        kw = {"median": True} if rng.random() < 0.2 else ({"robust": True} if rng.random() < 0.2 else {})
        r, q = oracle.Sketch(ref, c=c, marker_c=mc, k=k), oracle.Sketch(qry, c=c, marker_c=mc, k=k)
        want = oracle.query([("ref", r)], q, **kw)
        db = psk.Database(compression=c, marker_compression=mc, k=k)
        db.sketch("ref", *ref)
        gs = db._sketch("q", qry, True)
        seeds, markers = gs.export()
        assert np.array_equal(seeds["kmer"], q.seeds["kmer"]) and np.array_equal(seeds["pos"], q.seeds["pos"]), (seed, k, c)
        assert np.array_equal(markers, q.markers), (seed, k, c, mc)
        got = db.query_sketches([gs], learned_ani=False, **kw)[0]      # the general path, under this seed's forced kernel variants
        one = db.query("q", *qry, learned_ani=False, **kw)             # ... and Database.query (a small query: the one-launch-sequence path): the same records
        assert [(h.reference_name, h.identity, h._raw["n_anchors"], h._raw["covered_query"], h._raw["sum_chunk_seeds"]) for h in one] == \
               [(h.reference_name, h.identity, h._raw["n_anchors"], h._raw["covered_query"], h._raw["sum_chunk_seeds"]) for h in got], (seed, k, c, mc, kw)
        assert len(got) == len(want), (seed, k, c, mc, kw)
        if want:
            for f in INT_FIELDS:
                assert got[0]._raw[f] == getattr(want[0][1], f), (seed, k, c, mc, kw, f)
            assert abs(got[0].identity - want[0][1].ani) < 1e-6, (seed, k, c, mc, kw)
            assert abs(got[0].query_fraction - want[0][1].af_query) < 1e-6 and abs(got[0].reference_fraction - want[0][1].af_ref) < 1e-6


@pytest.mark.parametrize("seed", range(int(os.environ.get("PSK_FUZZ_DB_SEEDS", "6"))))
def test_random_databases_match_oracle(psk, oracle, seed, monkeypatch):
    """Random small databases (1-3 families, 3-20 members, assorted parameters) against several queries through
    query_many: hit sets, every chaining integer and ANI / AF must equal the oracle's screen + chain loop."""
    rng = np.random.default_rng(9000 + seed)
    if seed % 2:
        monkeypatch.setenv("PSK_PREFILTER", "1")      # seed prefilter of rescued queries whatever the batch size
        if seed % 4 == 1:
            monkeypatch.setenv("PSK_GSI_SLICE", "1")  # ... and the rounds' pairs joined through the database-wide seed index by (query, slice) waves (the all-vs-all default)
    else:
        monkeypatch.setenv("PSK_JOIN_PAIRS", "1")     # the join of many small pairs whatever the batch size: through the database-wide seed index,
        monkeypatch.setenv("PSK_PROBE", "1")          # or (every other seed) through the references' probe tables
        if seed % 4 == 2:
            monkeypatch.setenv("PSK_GSI_JOIN", "0")
        elif seed % 8 == 4:
            monkeypatch.setenv("PSK_GSI_ONEPASS", "0")
        else:
            monkeypatch.setenv("PSK_BSI_SMALL", "0")  # ... the database-wide index in one walk (default: the index in blocks of 256 references)
    k = int(rng.integers(11, 17)); c = int(rng.choice([30, 60, 125, 200])); mc = int(c * rng.choice([4, 8]))
    fams = [random_genome(rng, int(rng.integers(60000, 250000))) for _ in range(int(rng.integers(1, 4)))]
    refs = []
    for i in range(int(rng.integers(3, 21))):
        a = fams[int(rng.integers(0, len(fams)))]
        refs.append((f"r{i}", mutate(rng, a, rng.uniform(0, 0.12), rng.uniform(0, 0.0005))))
    db = psk.Database(compression=c, marker_compression=mc, k=k)
    for n, sq in refs:
        db.sketch(n, sq)
    osk = [(n, oracle.Sketch([sq], c=c, marker_c=mc, k=k)) for n, sq in refs]
    queries = []
    for j in range(int(rng.integers(1, 6))):
        a = fams[int(rng.integers(0, len(fams)))]
        lo = int(rng.integers(0, len(a) // 2))
        queries.append((f"q{j}", mutate(rng, a[lo:lo + int(rng.integers(3000, len(a) - lo))], rng.uniform(0, 0.06))))
    queries.append(("unrelated", random_genome(rng, 50000)))
    fs = bool(rng.integers(0, 2))
    cutoff = float(rng.choice([0.0, 0.7, 0.85, 0.95])) or None
    got_all = db.query_many(queries, learned_ani=False, faster_small=fs, cutoff=cutoff)
    for (qn, qs), got in zip(queries, got_all):
        want = {n: r for n, r in oracle.query(osk, oracle.Sketch([qs], c=c, marker_c=mc, k=k), faster_small=fs, cutoff=cutoff)}
        g = {h.reference_name: h for h in got}
        assert set(g) == set(want), (seed, qn, sorted(set(g) ^ set(want)))
        for n, w in want.items():
            for f in INT_FIELDS:
                assert g[n]._raw[f] == getattr(w, f), (seed, qn, n, f)
            assert abs(g[n].identity - w.ani) < 1e-6 and abs(g[n].query_fraction - w.af_query) < 1e-6
        single = db.query(qn, qs, learned_ani=False, faster_small=fs, cutoff=cutoff)
        assert [(h.reference_name, h.identity) for h in single] == [(h.reference_name, h.identity) for h in got]


@pytest.mark.parametrize("onepass", ["1", "0"])
def test_index_join_reruns_with_its_count_pass_when_a_reference_repeats_the_query(psk, oracle, onepass, monkeypatch):
    """The join through the database-wide seed index places a pair's anchors at the pair's item offset - room for one anchor per query seed - and counts
    while it writes. A reference that holds the query's k-mers several times over (a tandem repeat of the query) needs more room: the batch is rerun with
    the count pass, same answers; references that hold it once beside it keep the single walk honest in the same batch."""
    monkeypatch.setenv("PSK_PROBE", "1"); monkeypatch.setenv("PSK_JOIN_PAIRS", "1")
    monkeypatch.setenv("PSK_GSI_ONEPASS", onepass)
    rng = np.random.default_rng(4242)
    unit = random_genome(rng, 4000)
    a = random_genome(rng, 60000)
    refs = [("tandem", a[:20000] + unit * 7 + a[20000:]), ("once", a[:30000] + unit + a[30000:]), ("mut", mutate(rng, a[:10000] + unit * 2 + a[10000:], 0.02)), ("none", random_genome(rng, 50000))]
    db = psk.Database(compression=30, marker_compression=200)
    for n, sq in refs:
        db.sketch(n, sq)
    osk = [(n, oracle.Sketch([sq], c=30, marker_c=200)) for n, sq in refs]
    queries = [("unit", unit), ("unit_mut", mutate(rng, unit, 0.03)), ("flank", a[15000:27000]), ("two", unit * 2)]
    got_all = db.query_many(queries, learned_ani=False)
    for (qn, qs), got in zip(queries, got_all):
        want = {n: r for n, r in oracle.query(osk, oracle.Sketch([qs], c=30, marker_c=200))}
        g = {h.reference_name: h for h in got}
        assert set(g) == set(want), (qn, sorted(set(g) ^ set(want)))
        for n, w in want.items():
            for f in INT_FIELDS:
                assert g[n]._raw[f] == getattr(w, f), (qn, n, f, g[n]._raw[f], getattr(w, f))
            assert abs(g[n].identity - w.ani) < 1e-6 and abs(g[n].query_fraction - w.af_query) < 1e-6
    assert any(h._raw["n_anchors"] > len(oracle.Sketch([unit], c=30, marker_c=200).seeds) for h in got_all[0])      # the case is what it claims: more anchors than query seeds
