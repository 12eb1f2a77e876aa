"""GPU: Database.query() of a small genome as ONE launch sequence (psk_query_host -> csrc/small_query.hip).

What the reference's pymethod does per call (lib.rs:549-660: sketch the query at :571, screen every reference :617-637, chain the
shortlist :646-653, keep ani > 0.1 :654) is held here to the CPU oracle - hit sets, every chain integer, ANI / AF at 1e-6 - and to
the library's general path (Database.query_sketches over the same bytes: psk_sketch_host + psk_query_many), record for record.
`psk_ctx_small_query_stats` says which path a call took, so a silent fallback cannot pass for the fused path."""
import ctypes as C

import numpy as np
import pytest

from conftest import mutate, random_genome

pytestmark = pytest.mark.gpu

INT_FIELDS = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")
REC_FIELDS = ("ani", "af_query", "af_ref", "ref_index", "n_chunks", "n_intervals", "n_anchors", "covered_query", "covered_ref",
              "sum_chain_anchors", "sum_chunk_seeds", "ani_raw", "ani_std", "learned")


@pytest.fixture(scope="module")
def psk():
    import pyskani_amd
    return pyskani_amd


def stats(db):
    t, r, g = C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert db._lib.psk_ctx_small_query_stats(db._ctx._h, C.byref(t), C.byref(r), C.byref(g)) == 0
    return t.value, r.value, g.value


def revcomp(seq):
    return seq[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))


def check_against_oracle(oracle, osk, hits, contigs, c, marker_c, **kw):
    want = {n: r for n, r in oracle.query(osk, oracle.Sketch(list(contigs), c=c, marker_c=marker_c), **kw)}
    got = {h.reference_name: h for h in hits}
    assert set(got) == set(want), set(got) ^ set(want)
    for n, w in want.items():
        for f in INT_FIELDS:
            assert got[n]._raw[f] == getattr(w, f), (n, f, got[n]._raw[f], getattr(w, f))
        assert abs(got[n].identity - w.ani) < 1e-6 and abs(got[n].query_fraction - w.af_query) < 1e-6 and abs(got[n].reference_fraction - w.af_ref) < 1e-6
    return len(want)


def same_records(db, name, contigs, hits, **kw):
    """the general path over the same bytes gives the same records, bit for bit"""
    general = db.query_sketches([db.sketch_only(name, *contigs)], **kw)[0]
    assert len(general) == len(hits)
    for a, b in zip(hits, general):
        for f in REC_FIELDS:
            assert a._raw[f] == b._raw[f] or (a._raw[f] != a._raw[f] and b._raw[f] != b._raw[f]), (a.reference_name, f, a._raw[f], b._raw[f])


@pytest.mark.parametrize("c,marker_c", [(30, 200), (125, 1000), (70, 500), (200, 1000)])
def test_contig_queries_take_the_fused_path_and_match_oracle_and_general_path(psk, oracle, c, marker_c):
    rng = np.random.default_rng(1000 + c)
    anc = [random_genome(rng, 260000) for _ in range(3)]
    refs = [(f"r{f}_{j}", mutate(rng, a, d, 0.0002 * (j % 2))) for f, a in enumerate(anc) for j, d in enumerate((0.0, 0.01, 0.03, 0.08))]
    refs.append(("rc", revcomp(anc[1])))                                       # the reverse strand of a whole family
    db = psk.Database(compression=c, marker_compression=marker_c)
    for n, s in refs:
        db.sketch(n, s)
    osk = [(n, oracle.Sketch([s], c=c, marker_c=marker_c)) for n, s in refs]
    t0, r0, g0 = stats(db)
    n_q = n_hits = 0
    for i in range(10):
        L = int(np.exp(rng.uniform(np.log(1500), np.log(60000 if c == 30 else 150000))))
        a = anc[i % 3]; st = int(rng.integers(0, len(a) - L))
        seq = mutate(rng, a[st:st + L], rng.uniform(0, 0.06))
        if i % 4 == 3:
            seq = revcomp(seq)
        for kw in (dict(faster_small=False), dict(faster_small=True), dict(median=True), dict(robust=True), dict(cutoff=0.9)):
            hits = db.query(f"q{i}", seq, learned_ani=False, **kw)
            n_hits += check_against_oracle(oracle, osk, hits, [seq], c, marker_c, **kw)
            same_records(db, f"q{i}", [seq], hits, learned_ani=False, **kw)
            n_q += 1
    t1, r1, g1 = stats(db)
    assert t1 - t0 == n_q and r1 == r0 and g1 == g0, (t1 - t0, n_q, r1 - r0, g1 - g0)      # every call ran as one launch sequence
    assert n_hits > 40


def test_multi_contig_queries_repeats_and_short_contigs(psk, oracle):
    """queries of several contigs (some below 500 bases: ignored, lib.rs:156), a tandem repeat (k-mers with several matches),
    a contig that is a reference's own substring (ANI 1), and a query with no seed at all"""
    rng = np.random.default_rng(77)
    a = random_genome(rng, 200000)
    unit = random_genome(rng, 3000)
    rep = a[:50000] + unit * 6 + a[50000:]
    refs = [("plain", a), ("rep", rep), ("mut", mutate(rng, a, 0.02)), ("other", random_genome(rng, 150000))]
    db = psk.Database(compression=30, marker_compression=200)
    for n, s in refs:
        db.sketch(n, s)
    osk = [(n, oracle.Sketch([s], c=30, marker_c=200)) for n, s in refs]
    t0, r0, g0 = stats(db)
    queries = [
        [a[1000:9000], b"ACGT" * 50, mutate(rng, a[60000:71000], 0.03), a[150000:150600]],
        [unit * 3 + a[50000:52000]],                                          # inside the repeat: every unit k-mer matches 6 times
        [mutate(rng, rep[45000:75000], 0.01)],
        [a[20000:20700]],
        [b"A" * 3000],                                                         # one k-mer, every position
        [revcomp(a[100000:130000]), a[5000:25000]],
    ]
    for i, contigs in enumerate(queries):
        for fs in (False, True):
            hits = db.query(f"m{i}", *contigs, learned_ani=False, faster_small=fs)
            check_against_oracle(oracle, osk, hits, contigs, 30, 200, faster_small=fs)
            same_records(db, f"m{i}", contigs, hits, learned_ani=False, faster_small=fs)
    t1, r1, g1 = stats(db)
    assert (t1 - t0) + (g1 - g0) == 2 * len(queries) and t1 - t0 >= 2 * (len(queries) - 2)      # the repeat-rich ones may exceed a capacity and rerun: same answers


def test_duplicate_names_and_growing_database(psk, oracle):
    """lib.rs:616-637: the shortlist is a set of NAMES - a name sketched twice yields one hit, against its later sketch; and the
    device tables the fused path reads follow the database as references are added between queries"""
    rng = np.random.default_rng(5)
    a = random_genome(rng, 120000)
    b = mutate(rng, a, 0.03)
    db = psk.Database(compression=30, marker_compression=200)
    db.sketch("x", a)
    q = mutate(rng, a[30000:60000], 0.01)
    h1 = db.query("q", q, learned_ani=False)
    assert [h.reference_name for h in h1] == ["x"]
    db.sketch("y", b)
    db.sketch("x", b)                                                          # "x" now stands for b
    h2 = db.query("q", q, learned_ani=False)
    assert sorted(h.reference_name for h in h2) == ["x", "y"]
    ob = oracle.Sketch([b], c=30, marker_c=200)
    want = oracle.chain(ob, oracle.Sketch([q], c=30, marker_c=200))
    for h in h2:
        assert h._raw["n_anchors"] == want.n_anchors and abs(h.identity - want.ani) < 1e-6
    same_records(db, "q", [q], h2, learned_ani=False)


def test_capacity_overflow_reruns_on_the_general_path(psk, oracle):
    """a query just inside the size limits whose seed count exceeds the LDS capacity (low-complexity sequence: far more seeds than
    bases / c) raises the flag and is answered by the general path"""
    rng = np.random.default_rng(9)
    a = random_genome(rng, 100000)
    for _ in range(64):                                                        # a period-40 repeat whose few k-mers are mostly selected: far more seeds than bases / c
        unit = random_genome(rng, 40)
        low = (unit * 2000)[:60000]
        if len(oracle.Sketch([low], c=30, marker_c=200).seeds) > 3300:
            break
    else:
        pytest.skip("no dense repeat unit found")
    refs = [("a", a), ("low", low[:8000] + a[:20000])]
    db = psk.Database(compression=30, marker_compression=200)
    for n, s in refs:
        db.sketch(n, s)
    osk = [(n, oracle.Sketch([s], c=30, marker_c=200)) for n, s in refs]
    t0, r0, g0 = stats(db)
    check_against_oracle(oracle, osk, db.query("lowq", low, learned_ani=False), [low], 30, 200)
    t1, r1, g1 = stats(db)
    assert (t1 - t0, r1 - r0, g1 - g0) == (0, 1, 1)                            # tried, flagged, rerun
    big = [a[:95000] + a[:95000]]                                             # too large for the fused path: never tried
    check_against_oracle(oracle, osk, db.query("big", *big, learned_ani=False), big, 30, 200)
    t2, r2, g2 = stats(db)
    assert (t2 - t1, r2 - r1, g2 - g1) == (0, 0, 1)


def test_many_references_long_shortlist(psk, oracle):
    """a rescued short contig (< 20 markers) passes the screen against EVERY reference (lib.rs:617-630): a shortlist longer than the
    records that cross with the status block, most pairs without a chain"""
    rng = np.random.default_rng(13)
    anc = [random_genome(rng, 30000) for _ in range(6)]
    refs = [(f"r{i}", mutate(rng, anc[i % 6], 0.002 * (i // 6))) for i in range(330)]
    db = psk.Database(compression=30, marker_compression=200)
    db.sketch_many([(n, s) for n, s in refs])
    osk = [(n, oracle.Sketch([s], c=30, marker_c=200)) for n, s in refs]
    t0, _, g0 = stats(db)
    q = mutate(rng, anc[2][4000:6500], 0.02)                                   # ~12 markers expected
    for fs in (False, True):
        hits = db.query("short", q, learned_ani=False, faster_small=fs)
        n = check_against_oracle(oracle, osk, hits, [q], 30, 200, faster_small=fs)
        assert fs or n >= 50
    t1, _, g1 = stats(db)
    assert t1 - t0 == 2 and g1 == g0


def test_concurrent_fused_queries_from_threads(psk, oracle):
    """the reference's query() takes &self and releases the GIL (lib.rs:551,569): concurrent calls, each on its own lane"""
    import threading
    rng = np.random.default_rng(21)
    anc = [random_genome(rng, 150000) for _ in range(2)]
    refs = [(f"r{f}_{j}", mutate(rng, a, d)) for f, a in enumerate(anc) for j, d in enumerate((0.0, 0.02, 0.05))]
    db = psk.Database(compression=30, marker_compression=200)
    for n, s in refs:
        db.sketch(n, s)
    qs = []
    for i in range(24):
        L = int(rng.integers(3000, 40000)); a = anc[i % 2]; st = int(rng.integers(0, len(a) - L))
        qs.append(mutate(rng, a[st:st + L], 0.02))
    serial = [[(h.reference_name, h.identity, h._raw["n_anchors"]) for h in db.query(f"q{i}", s, learned_ani=False)] for i, s in enumerate(qs)]
    out = [None] * len(qs)

    def work(k):
        for i in range(k, len(qs), 6):
            out[i] = [(h.reference_name, h.identity, h._raw["n_anchors"]) for h in db.query(f"q{i}", qs[i], learned_ani=False)]
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in th]; [t.join() for t in th]
    assert out == serial


def test_c_level_call_and_ctypes_route_return_the_same_hits(psk):
    """Database.query goes to the library through ONE C-level call (csrc/hitlist.c: query_host); the ctypes route it replaces
    ($PSK_PY_FASTCALL=0) gives the same hits, field for field, and errors surface the same way"""
    rng = np.random.default_rng(31)
    a = random_genome(rng, 150000)
    db = psk.Database(compression=30, marker_compression=200)
    for j, d in enumerate((0.0, 0.01, 0.04)):
        db.sketch(f"r{j}", mutate(rng, a, d))
    assert db._fast is not None
    qs = [(mutate(rng, a[1000:30000], 0.02),), (a[5000:9000], a[60000:90000]), (bytearray(a[100000:120000]),), (a[:2000].decode(),), (b"ACGT" * 100,)]
    for contigs in qs:
        fast = db.query("q", *contigs, learned_ani=False)
        keep, db._fast = db._fast, None
        try:
            slow = db.query("q", *contigs, learned_ani=False)
        finally:
            db._fast = keep
        assert len(fast) == len(slow)
        for x, y in zip(fast, slow):
            assert (x.identity, x.query_name, x.query_fraction, x.reference_name, x.reference_fraction, x.learned) == \
                   (y.identity, y.query_name, y.query_fraction, y.reference_name, y.reference_fraction, y.learned)
            assert all(x._raw[f] == y._raw[f] or x._raw[f] != x._raw[f] for f in REC_FIELDS)
    with pytest.raises(TypeError):
        db.query("q", 5)
