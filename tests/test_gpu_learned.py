"""GPU: learned-ANI regression (lib.rs:611-614), name de-duplication of the shortlist (lib.rs:51-55, 616-637) and the
round-1 advisor findings, through the C-ABI, against the oracle and the evaluator in tests/gbdt_util.py."""
import os
import warnings

import numpy as np
import pytest

import gbdt_util as G
from conftest import mutate, random_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psk():
    import pyskani_amd
    return pyskani_amd


@pytest.fixture(scope="module")
def trio():
    rng = np.random.default_rng(41)
    g = random_genome(rng, 400_000)
    cuts = [0, 90_000, 150_000, 150_400, 300_000, 400_000]          # one contig < 500 is dropped (lib.rs:156)
    q = mutate(rng, g, 0.015)
    return g, [q[a:b] for a, b in zip(cuts, cuts[1:])], mutate(rng, g, 0.04)


def test_model_predict_matches_python_evaluator(psk):
    rng = np.random.default_rng(5)
    trees = G.random_trees(rng, n_trees=40, depth=5)
    m = psk.Model.from_trees(trees, bias=98.5, shrinkage=0.1)
    assert m.shape == (40, 40 * 63, 9)
    rows = np.array([[rng.uniform(lo, hi) for lo, hi in [(97, 100), (0, 1)] + [(1e3, 6e6)] * 6 + [(1e3, 3e4)]] for _ in range(500)], np.float32)
    rows[rng.random(rows.shape) < 0.05] = G.UNKNOWN
    got = m.predict(rows)
    want = np.array([G.predict(trees, 98.5, 0.1, r) for r in rows], np.float32)
    assert np.array_equal(got, want)                                  # f32 accumulation in tree order: bit-exact


def test_model_json_loader(psk, tmp_path):
    """The serde-JSON shape of gbdt::GBDT; a custom feature order through "psk_features"."""
    rng = np.random.default_rng(6)
    trees = G.random_trees(rng, n_trees=5, depth=3)
    path = tmp_path / "model.json"
    path.write_text(G.to_gbdt_json(trees, 97.25, 0.3))
    m = psk.Model.from_file(str(path))
    row = np.array([[99.1, 0.4, 5e6, 4e6, 3e6, 5e6, 4e6, 3e6, 2e4]], np.float32)
    assert m.predict(row)[0] == np.float32(G.predict(trees, 97.25, 0.3, row[0]))
    feats = ["std100", "ani100", "af_query", "af_ref", "n_chunks", "q50_ref", "q50_query", "total_len_ref", "n_contigs_query"]
    m2 = psk.Model.from_json(G.to_gbdt_json(trees, 97.25, 0.3, features=feats))
    assert m2.shape[2] == 9
    with pytest.raises(ValueError):
        psk.Model.from_json('{"conf": {}, "trees": [{"tree": {"tree": []}}], "bias": 0}')
    with pytest.raises(ValueError):
        psk.Model.from_json(G.to_gbdt_json(trees, 0.0, 1.0, features=["ani100", "nonsense"]))
    with pytest.raises(ValueError):
        psk.Model.from_json("not json")
    with pytest.raises(KeyError):
        psk.Model.from_file(str(tmp_path / "absent.json"))


@pytest.mark.parametrize("features", [None, ["ani100", "std100", "af_query", "af_ref", "n_chunks", "q90_query", "q10_ref", "avg_chain_len",
                                             "total_len_query", "total_len_ref", "n_contigs_query", "n_contigs_ref"]])
def test_query_with_model_matches_oracle(psk, oracle, trio, features):
    ref, qcontigs, other = trio
    rng = np.random.default_rng(8)
    nf = len(features) if features else 9
    scales = None if features is None else [(97, 100), (0, 1), (0, 1), (0, 1), (1, 40), (1e3, 2e5), (1e3, 5e5), (1e3, 3e4), (3e5, 5e5), (3e5, 5e5), (1, 6), (1, 3)]
    trees = G.random_trees(rng, n_trees=12, depth=4, n_features=nf, scales=scales)
    model = psk.Model.from_trees(trees, bias=98.0, shrinkage=0.2, features=features)
    from pyskani_amd import _capi
    ids = list(range(9)) if features is None else [_capi.FEATURE_NAMES.index(f) for f in features]
    omodel = oracle.Model(trees, 98.0, 0.2, ids)
    db = psk.Database(model=model)
    db.sketch("ref", ref)
    db.sketch("other", other)
    orefs = [("ref", oracle.Sketch([ref])), ("other", oracle.Sketch([other]))]
    oq = oracle.Sketch(qcontigs)
    for kw in ({}, {"learned_ani": True}, {"learned_ani": False}, {"median": True}, {"robust": True}, {"median": True, "learned_ani": True}):
        hits = db.query("q", *qcontigs, **kw)
        want = oracle.query(orefs, oq, median=kw.get("median", False), robust=kw.get("robust", False),
                            learned_ani=kw.get("learned_ani", None), model=omodel)
        assert [h.reference_name for h in hits] == [n for n, _ in want]
        for h, (_, w) in zip(hits, want):
            assert h.learned == bool(w.learned)
            assert h._raw["learned"] == w.learned
            assert abs(h._raw["ani_raw"] - w.ani_raw) < 1e-6 and abs(h._raw["ani_std"] - w.ani_std) < 1e-6
            assert abs(h.identity - w.ani) < 2e-6, (kw, h.identity, w.ani)
        # the default rule: c >= 70 and not median (lib.rs:611-613)
        expect_learned = kw.get("learned_ani", not kw.get("median", False))
        assert all(h.learned == expect_learned for h in hits)
    # many-query path shares the stage
    a = db.query_many([("q", *qcontigs), ("r", ref)])
    b = [db.query("q", *qcontigs), db.query("r", ref)]
    assert [[(h.reference_name, h.identity, h.learned) for h in x] for x in a] == [[(h.reference_name, h.identity, h.learned) for h in x] for x in b]


def test_default_rule_needs_c_70(psk, trio):
    ref, qcontigs, _ = trio
    trees = G.random_trees(np.random.default_rng(2), n_trees=2, depth=2)
    db = psk.Database(compression=60, marker_compression=500, model=psk.Model.from_trees(trees, bias=98.0))
    db.sketch("ref", ref)
    assert [h.learned for h in db.query("q", *qcontigs)] == [False]
    assert [h.learned for h in db.query("q", *qcontigs, learned_ani=True)] == [True]


def test_no_model_behaviour(psk, trio, monkeypatch):
    """No model: learned_ani=True raises, the default warns once per Database and flags the hits as un-regressed."""
    monkeypatch.delenv("PSK_MODEL_PATH", raising=False)
    ref, qcontigs, _ = trio
    for _ in range(2):                 # every Database warns, not only the first of the process
        db = psk.Database()
        db.sketch("ref", ref)
        with pytest.raises(RuntimeError):
            db.query("q", *qcontigs, learned_ani=True)
        with pytest.warns(RuntimeWarning, match="RAW chain ANI"):
            hits = db.query("q", *qcontigs)
        assert len(hits) == 1 and hits[0].learned is False
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            db.query("q", *qcontigs)                           # once per Database
            db.query("q", *qcontigs, learned_ani=False)
            db.query("q", *qcontigs, median=True)              # default rule: median switches the model off


def test_model_from_environment(psk, trio, tmp_path, monkeypatch):
    ref, qcontigs, _ = trio
    trees = G.random_trees(np.random.default_rng(12), n_trees=4, depth=3)
    path = tmp_path / "m.json"
    path.write_text(G.to_gbdt_json(trees, 98.0, 0.25))
    monkeypatch.setenv("PSK_MODEL_PATH", str(path))
    db = psk.Database()
    db.sketch("ref", ref)
    hits = db.query("q", *qcontigs)
    assert hits[0].learned and hits[0]._raw["ani_raw"] != hits[0]._raw["ani"]


def test_duplicate_names_give_one_hit(psk, oracle):
    """lib.rs:51-55 (store keyed by name) + :616-637 (shortlist of names): one hit per NAME, against its last sketch —
    also when only the EARLIER entry passes the screen."""
    rng = np.random.default_rng(3)
    g = random_genome(rng, 150_000)
    far, near, unrelated = mutate(rng, g, 0.05), mutate(rng, g, 0.01), random_genome(rng, 150_000)
    db = psk.Database()
    for name, seq in (("x", far), ("y", near), ("x", near), ("z", near), ("z", unrelated)):
        db.sketch(name, seq)
    sk = {n: oracle.Sketch([s]) for n, s in (("far", far), ("near", near), ("unrelated", unrelated))}
    orefs = [("x", sk["far"]), ("y", sk["near"]), ("x", sk["near"]), ("z", sk["near"]), ("z", sk["unrelated"])]
    for many in (False, True):
        hits = db.query_many([("q", g)], learned_ani=False)[0] if many else db.query("q", g, learned_ani=False)
        want = oracle.query(orefs, oracle.Sketch([g]))
        # "z": its first entry passes the screen, so the name is shortlisted and chained against the unrelated later sketch -> no hit
        assert [h.reference_name for h in hits] == [n for n, _ in want] == ["y", "x"]
        for h, (_, w) in zip(hits, want):
            assert h._raw["n_anchors"] == w.n_anchors and abs(h.identity - w.ani) < 1e-6


def test_duplicate_names_on_disk(psk, tmp_path):
    rng = np.random.default_rng(4)
    g = random_genome(rng, 100_000)
    db = psk.Database(str(tmp_path / "sep"), format="separated")        # Folder.store overwrites silently, lib.rs:57-58
    db.sketch("x", mutate(rng, g, 0.05)); db.sketch("x", mutate(rng, g, 0.01)); db.flush()
    opened = psk.Database.open(str(tmp_path / "sep"))
    hits = opened.query("q", g, learned_ani=False)
    assert len(hits) == 1 and hits[0].identity > 0.985
    con = psk.Database(str(tmp_path / "con"))                            # Consolidated.store rejects duplicates, lib.rs:66-72
    con.sketch("x", g)
    with pytest.raises(ValueError):
        con.sketch("x", g)


def test_save_onto_own_folder(psk, tmp_path):
    """save(path, overwrite=True) onto the folder an opened database reads from must not destroy it."""
    rng = np.random.default_rng(6)
    g = random_genome(rng, 80_000)
    folder = str(tmp_path / "db")
    with psk.Database(folder) as db:
        db.sketch("a", g); db.sketch("b", mutate(rng, g, 0.02))
    opened = psk.Database.open(folder)
    before = [(h.reference_name, h.identity) for h in opened.query("q", g, learned_ani=False)]
    opened.save(folder, overwrite=True)
    assert [(h.reference_name, h.identity) for h in opened.query("q", g, learned_ani=False)] == before
    again = psk.Database.load(folder)
    assert [(h.reference_name, h.identity) for h in again.query("q", g, learned_ani=False)] == before and len(before) == 2
    opened.save(folder, overwrite=True, format="separated")
    assert os.path.exists(os.path.join(folder, "a.sketch"))


def test_pack_unpack_device_records(psk, oracle):
    """psk_sketch_pack / psk_sketch_unpack (the multi-GPU exchange records): unpacked sketches are byte-identical to the
    originals (seeds, markers, contig tables) and chain to the same result."""
    import ctypes as C
    import torch
    rng = np.random.default_rng(21)
    g = random_genome(rng, 200_000)
    genomes = [[g[:70_000], g[70_000:70_300], g[70_300:]], [mutate(rng, g, 0.02)], [b"ACGT" * 50], [random_genome(rng, 30_000)] * 2]
    db = psk.Database()
    sketches = [db._sketch(f"s{i}", c, True) for i, c in enumerate(genomes)] + [db._sketch("m", genomes[1], False)]
    sizes = [s.pack_size() for s in sketches]
    assert all(sz % 16 == 0 and sz >= 64 for sz in sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    buf = torch.zeros(int(offs[-1]), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()      # the fill runs on torch's stream, the library packs on its own: order them
    for s, o, sz in zip(sketches, offs, sizes):
        s.pack_into(buf.data_ptr() + int(o), sz)
    with pytest.raises(ValueError):
        sketches[0].pack_into(buf.data_ptr(), sizes[0] - 16)
    torch.cuda.synchronize()
    moved = buf.clone()                                    # what an all-gather would deliver
    torch.cuda.synchronize()
    got = psk.Sketch.unpack(db._ctx, moved.data_ptr(), offs[:-1], [s.name for s in sketches])
    for a, b in zip(sketches, got):
        sa, ma = a.export(); sb, mb = b.export()
        assert sa.tobytes() == sb.tobytes() and ma.tobytes() == mb.tobytes()
        assert a.to_record().to_bytes() == b.to_record().to_bytes()
        assert a._info()[1:] == b._info()[1:]
    ref = psk.Database()
    ref.sketch("ref", g)
    want = [[(h.identity, h._raw["n_anchors"], h._raw["covered_query"]) for h in x] for x in ref.query_sketches(sketches[:4], learned_ani=False)]
    have = [[(h.identity, h._raw["n_anchors"], h._raw["covered_query"]) for h in x] for x in ref.query_sketches(got[:4], learned_ani=False)]
    assert want == have and len(want[0]) == 1 and len(want[1]) == 1
    bad = moved.clone(); bad[int(offs[1])] = 0
    torch.cuda.synchronize()
    with pytest.raises(ValueError):
        psk.Sketch.unpack(db._ctx, bad.data_ptr(), offs[:-1], [s.name for s in sketches])
    # with the buffer's extent given, a record that would run past it is refused before anything is read through its header
    total = int(offs[-1])
    ok = psk.Sketch.unpack(db._ctx, moved.data_ptr(), offs[:-1], [s.name for s in sketches], capacity=total)
    assert len(ok) == len(sketches)
    with pytest.raises(ValueError, match="past the buffer|beyond the buffer"):
        psk.Sketch.unpack(db._ctx, moved.data_ptr(), offs[:-1], [s.name for s in sketches], capacity=total - 16)      # the last record is truncated
    with pytest.raises(ValueError, match="beyond the buffer"):
        psk.Sketch.unpack(db._ctx, moved.data_ptr(), offs[:-1], [s.name for s in sketches], capacity=int(offs[-2]) + 8)   # not even its header fits


def test_query_edge_cases(psk):
    """Empty database, no query, queries without a hit / without seeds / sketched with seed=False, references sketched
    with seed=False: the errors and empty results the reference's loop structure implies (lib.rs:617-657)."""
    rng = np.random.default_rng(77)
    g = random_genome(rng, 60_000)
    empty = psk.Database()
    assert empty.query("q", g, learned_ani=False) == [] and empty.query_many([("q", g)], learned_ani=False) == [[]]
    assert empty.query_many([], learned_ani=False) == []
    db = psk.Database()
    db.sketch("a", g)
    db.sketch("short", b"ACGT" * 100)                       # below MIN_LENGTH_CONTIG: a reference without seeds or markers
    assert db.query_many([], learned_ani=False) == []
    assert db.query("unrelated", random_genome(rng, 60_000), learned_ani=False) == []
    assert db.query("tiny", b"ACGT" * 50, learned_ani=False) == []              # query without seeds: no hits, no error
    res = db.query_many([("self", g), ("tiny", b"ACGT" * 50), ("none", random_genome(rng, 50_000)), ("again", g)], learned_ani=False)
    assert [len(r) for r in res] == [1, 0, 0, 1] and res[0][0].identity == 1.0 and res[3][0].reference_name == "a"
    with pytest.raises(ValueError, match="seed=False"):
        db.query("noseed", g, seed=False, learned_ani=False)                     # passes the screen, cannot be chained
    db2 = psk.Database()
    db2.sketch("markers_only", g, seed=False)
    with pytest.raises(ValueError, match="seed=False"):
        db2.query("q", g, learned_ani=False)
    assert db2.query("unrelated", random_genome(rng, 60_000), learned_ani=False) == []   # nothing shortlisted: no error
    other = psk.Database(compression=60)
    with pytest.raises(ValueError, match="different parameters"):
        db.query_sketches([other.sketch_only("x", g)], learned_ani=False)


@pytest.mark.parametrize("kw", [{}, {"median": True}, {"robust": True}, {"faster_small": True}])
def test_many_small_pairs_against_oracle(psk, oracle, kw, monkeypatch):
    """A batch of > 4 096 small pairs (short contigs, most of them rescued against every reference, lib.rs:538-541) takes the
    paths a handful of pairs never sees: the list of live pairs, the wave-per-pair reduction of short chunk tables, the
    pair-major join. Every hit against the oracle; and the same batch with those paths switched off, and with the seed
    prefilter of rescued contigs (default from 2^20 pairs on) forced on."""
    rng = np.random.default_rng(91)
    anc = [random_genome(rng, 90_000) for _ in range(4)]
    refs = [(f"r{f}_{j}", mutate(rng, anc[f], 0.004 * j)) for f in range(4) for j in range(15)]          # 60 references
    contigs = []
    for i in range(84):
        a = anc[i % 4]
        st = int(rng.integers(0, len(a) - 9000))
        L = int(rng.integers(1500, 3500)) if i % 3 else int(rng.integers(5000, 9000))                    # two thirds below 20 markers
        contigs.append((f"c{i}", mutate(rng, a[st:st + L], rng.uniform(0, 0.04))))
    db = psk.Database(compression=30, marker_compression=200)
    db.sketch_many(refs)
    got = db.query_many(contigs, learned_ani=False, **kw)
    orefs = [(n, oracle.Sketch([g], c=30, marker_c=200)) for n, g in refs]
    n_pairs_chained = 0
    step = 1 if not kw else 4                      # the oracle side is the slow one: every contig for the default flags, a quarter otherwise
    for (name, seq), hits in list(zip(contigs, got))[::step]:
        want = oracle.query(orefs, oracle.Sketch([seq], c=30, marker_c=200), median=kw.get("median", False), robust=kw.get("robust", False),
                            faster_small=kw.get("faster_small", False))
        assert [h.reference_name for h in hits] == [n for n, _ in want], name
        for h, (_, w) in zip(hits, want):
            for f in ("n_anchors", "n_chunks", "n_intervals", "covered_query", "sum_chain_anchors", "sum_chunk_seeds"):
                assert h._raw[f] == getattr(w, f), (name, h.reference_name, f)
            assert abs(h.identity - w.ani) < 1e-6 and abs(h._raw["ani_std"] - w.ani_std) < 1e-6
        n_pairs_chained += len(hits)
    assert n_pairs_chained > 500 // step
    if not kw:
        ref = [[(h.reference_name, h.identity, h._raw["n_anchors"]) for h in hs] for hs in got]
        for var, val in (("PSK_REDUCE_SMALL", "0"), ("PSK_JOIN_PAIRS", "0"), ("PSK_JOIN_PAIRS", "1"), ("PSK_PREFILTER", "1")):
            monkeypatch.setenv(var, val)
            alt = db.query_many(contigs, learned_ani=False)
            monkeypatch.delenv(var)
            assert [[(h.reference_name, h.identity, h._raw["n_anchors"]) for h in hs] for hs in alt] == ref, var


def test_pack_many_and_device_sketching(psk):
    """psk_sketch_pack_many: one call packs a batch (one header upload, one copy launch, one synchronisation); the records unpack to
    byte-identical sketches. Database.sketch_many_device: genomes staged in HBM by the caller give the same sketches as sketch()."""
    import ctypes as C
    import torch
    from pyskani_amd import _capi
    rng = np.random.default_rng(33)
    g = random_genome(rng, 150_000)
    genomes = [[g[:60_000], g[60_000:60_400], g[60_400:]], [mutate(rng, g, 0.03)], [random_genome(rng, 20_000)] * 3, [b"ACGT" * 40]]
    db = psk.Database()
    sketches = [db._sketch(f"s{i}", c, True) for i, c in enumerate(genomes)]
    sizes = [s.pack_size() for s in sketches]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    buf = torch.empty(int(offs[-1]), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    lib = db._lib
    handles = (C.c_void_p * len(sketches))(*[s._h for s in sketches])
    c_offs = (C.c_uint64 * len(sketches))(*[int(o) for o in offs[:-1]])
    _capi.check(lib.psk_sketch_pack_many(handles, len(sketches), C.c_void_p(buf.data_ptr()), c_offs, int(offs[-1])))
    assert lib.psk_sketch_pack_many(handles, len(sketches), C.c_void_p(buf.data_ptr()), c_offs, int(offs[-1]) - 16) == _capi.PSK_EINVAL
    got = psk.Sketch.unpack(db._ctx, buf.data_ptr(), offs[:-1], [s.name for s in sketches])
    for a, b in zip(sketches, got):
        sa, ma = a.export(); sb, mb = b.export()
        assert sa.tobytes() == sb.tobytes() and ma.tobytes() == mb.tobytes()
    # device-resident ingest through the Database class
    flat, coffs, clens, gfc = [], [], [], [0]
    total = 0
    for cs in genomes:
        for c in cs:
            coffs.append(total); clens.append(len(c)); flat.append(np.frombuffer(c, np.uint8)); total += (len(c) + 31) & ~15
            pad = ((len(c) + 31) & ~15) - len(c)
            flat.append(np.zeros(pad, np.uint8))
        gfc.append(len(coffs))
    dev = torch.from_numpy(np.concatenate(flat + [np.zeros(64, np.uint8)])).cuda()
    torch.cuda.synchronize()
    db2 = psk.Database()
    db2.sketch_many_device([f"s{i}" for i in range(len(genomes))], dev.data_ptr(), coffs, clens, gfc)
    assert len(db2) == len(genomes)
    for i, a in enumerate(sketches):
        sa, ma = a.export(); sb, mb = db2._full_sketch(i).export()
        assert sa.tobytes() == sb.tobytes() and ma.tobytes() == mb.tobytes()


def test_clock_probe_and_work_counters(psk):
    import ctypes as C
    from pyskani_amd import _capi
    rng = np.random.default_rng(4)
    db = psk.Database()
    lib, ctx = db._lib, db._ctx._h
    mhz, ms = C.c_double(), C.c_double()
    _capi.check(lib.psk_ctx_clock_probe(ctx, C.byref(mhz), C.byref(ms)))
    assert 500.0 < mhz.value < 3000.0 and 0.1 < ms.value < 20.0, (mhz.value, ms.value)      # MI355X: up to 2.4 GHz
    g = random_genome(rng, 200_000)
    db.sketch("a", g); db.sketch("b", mutate(rng, g, 0.02)); db.sketch("z", random_genome(rng, 200_000))
    p, i, a = C.c_uint64(), C.c_uint64(), C.c_uint64()
    _capi.check(lib.psk_ctx_work(ctx, None, None, None, 1))
    hits = db.query("q", mutate(rng, g, 0.01), learned_ani=False)
    _capi.check(lib.psk_ctx_work(ctx, C.byref(p), C.byref(i), C.byref(a), 0))
    assert len(hits) == 2 and p.value == 2 and a.value == sum(int(h._raw["n_anchors"]) for h in hits)
    assert i.value > 0 and i.value % 2 == 0      # two pairs of the same query: 2 x its seed count
