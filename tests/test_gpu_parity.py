"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.
Bit-exact for seed sets, marker sets, shared-marker counts and every integer intermediate of
chaining; ANI / AF within 1e-6 (BASELINE.json's tolerance is 1e-4)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, mutate, random_genome

pytestmark = pytest.mark.gpu

INT_FIELDS = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")


@pytest.fixture(scope="module")
def psk():
    import pyskani_amd
    return pyskani_amd


def gpu_sketch(psk, contigs, **kw):
    db = psk.Database(**kw)
    return db, db._sketch("g", contigs, True)


def assert_sketch_equal(gs, osk):
    seeds, markers = gs.export()
    o = osk.seeds
    assert len(seeds) == len(o)
    for f in ("kmer", "pos", "contig", "canon"):
        assert np.array_equal(seeds[f], o[f]), f
    assert np.array_equal(markers, osk.markers)


def test_sketch_ecoli_bit_exact(psk, oracle, ecoli):
    for seq in ecoli:
        db, gs = gpu_sketch(psk, [seq])
        assert_sketch_equal(gs, oracle.Sketch([seq]))


@pytest.mark.parametrize("k,c,mc", [(15, 125, 1000), (15, 30, 200), (16, 125, 1000), (11, 50, 400), (13, 200, 1000), (8, 7, 20)])
def test_sketch_params_and_ragged_contigs(psk, oracle, k, c, mc):
    rng = np.random.default_rng(k * 1000 + c)
    lens = [499, 500, 501, 16383, 16384, 16385, 16384 * 2 + 21, 40000, 20, 0, 777, 65536 + 15]
    contigs = [random_genome(rng, n) for n in lens]
    # lower case, N runs and IUPAC codes must all map to 0 like skani's BYTE_TO_SEQ
    b = bytearray(contigs[7]); b[100:180] = b"N" * 80; b[5000:5040] = b"acgtnRYKM-" * 4; b[16380:16390] = b"NNNNNnnnnn"; contigs[7] = bytes(b)
    contigs[3] = contigs[3].lower()
    db, gs = gpu_sketch(psk, contigs, compression=c, marker_compression=mc, k=k)
    assert_sketch_equal(gs, oracle.Sketch(contigs, c=c, marker_c=mc, k=k))


@pytest.mark.parametrize("k,c", [(15, 125), (16, 20), (9, 5), (1, 3), (2, 2)])
def test_sketch_arbitrary_bytes(psk, oracle, k, c):
    """Every one of the 256 byte values, scattered: only ACGT/acgt are bases, everything else is base 0
    (the v_perm fast path of the packer must hand every other byte to the checked path)."""
    rng = np.random.default_rng(700 + k)
    contigs = []
    for n in (70000, 16384 * 3, 5000):
        b = bytearray(random_genome(rng, n))
        pos = rng.choice(n, size=n // 50, replace=False)
        for p_, v in zip(pos, rng.integers(0, 256, size=len(pos))):
            b[p_] = int(v)
        contigs.append(bytes(b))
    contigs.append(bytes(rng.integers(0, 256, size=20000, dtype=np.uint8)))      # pure noise
    contigs.append(bytes(range(256)) * 40)
    db, gs = gpu_sketch(psk, contigs, compression=c, marker_compression=4 * c, k=k)
    assert_sketch_equal(gs, oracle.Sketch(contigs, c=c, marker_c=4 * c, k=k))


def test_marker_paths(psk, oracle):
    """Marker sets: the one-workgroup-per-genome LDS path (default for genomes up to ~6 Mb at marker_c = 1000), its
    overflow fallback (a repeat-rich genome with more raw markers than the LDS holds) and the segmented-sort path
    forced through PSK_MARKER_SEGSORT must all give the oracle's sets."""
    rng = np.random.default_rng(77)
    unit = None
    for _ in range(200):                       # a 1 kb unit that carries at least two marker windows
        u = random_genome(rng, 1000)
        if len(oracle.Sketch([u * 3]).markers) >= 2:
            unit = u
            break
    assert unit is not None
    repeat = unit * 6000                       # 6 Mb: expected 6 000 raw markers, really >= 12 000 -> device overflow flag
    plain = random_genome(rng, 700000)
    want_r, want_p = oracle.Sketch([repeat]), oracle.Sketch([plain])
    db, gs = gpu_sketch(psk, [repeat]); assert_sketch_equal(gs, want_r)
    db, gs = gpu_sketch(psk, [plain]); assert_sketch_equal(gs, want_p)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import random_genome\n"
        "import pyskani_amd\n"
        "rng = np.random.default_rng(78); g = random_genome(rng, 700000)\n"
        "db = pyskani_amd.Database(); db.sketch('a', g)\n"
        "import hashlib\n"
        "sk = db._sketch('x', [g], True); seeds, markers = sk.export()\n"
        "print(len(seeds), len(markers), hashlib.sha256(markers.tobytes()).hexdigest())\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for force in (False, True, "sliced"):      # "sliced": the distinct-marker pass of few LARGE genomes (a segment cut into slices), forced on this one
        env = dict(os.environ); env.pop("PSK_MARKER_SEGSORT", None); env.pop("PSK_MARKER_SLICES", None)
        if force:
            env["PSK_MARKER_SEGSORT"] = "1"
        if force == "sliced":
            env["PSK_MARKER_SLICES"] = "7"
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600).decode().strip())
    assert outs[0] == outs[1] == outs[2], outs


def test_tagged_marker_sort_is_redone_at_full_size_when_markers_outnumber_its_cap(oracle):
    """The tagged marker sort of large genomes (forced here: PSK_MARKER_SEGSORT=1) takes twice the EXPECTED number of raw markers + 65 536 entries instead
    of the seed count. A genome whose raw markers outnumber that - a 10-base period that carries a marker, 1.2 Mb of it: >= 120 000 raw markers where
    2 x 1.5 Mb / 200 + 65 536 = 80 536 are allowed for - is flagged on the device and sorted again at the seed count; same sets as the oracle either way."""
    rng = np.random.default_rng(79)
    unit = None
    for _ in range(400):                       # a 10-base period one of whose ten 21-mers is a marker at marker_c = 200
        u = random_genome(rng, 10)
        if len(oracle.Sketch([u * 300], c=30, marker_c=200).markers) >= 1:
            unit = u
            break
    assert unit is not None
    import tempfile
    out = os.path.join(tempfile.mkdtemp(), "sk.npz")
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import random_genome\n"
        "import pyskani_amd\n"
        "rng = np.random.default_rng(80); g = random_genome(rng, 150000) + %r * 120000 + random_genome(rng, 150000)\n"
        "db = pyskani_amd.Database(compression=30, marker_compression=200)\n"
        "sk = db._sketch('x', [g], True); seeds, markers = sk.export()\n"
        "np.savez(%r, kmer=seeds['kmer'], pos=seeds['pos'], markers=markers)\n"
    ) % (ROOT, os.path.join(ROOT, "tests"), unit, out)
    env = dict(os.environ); env["PSK_MARKER_SEGSORT"] = "1"; env["PSK_SKETCH_SMALL"] = "0"
    subprocess.check_call([sys.executable, "-c", code], env=env, timeout=600)
    rng2 = np.random.default_rng(80); g = random_genome(rng2, 150000) + unit * 120000 + random_genome(rng2, 150000)
    want = oracle.Sketch([g], c=30, marker_c=200)
    got = np.load(out)
    assert np.array_equal(got["kmer"], want.seeds["kmer"]) and np.array_equal(got["pos"], want.seeds["pos"])
    assert np.array_equal(got["markers"], want.markers)


def test_sketch_empty_and_short(psk, oracle):
    db, gs = gpu_sketch(psk, [b"ATGC" * 100])        # 400 bp: below MIN_LENGTH_CONTIG (test_database.py:13)
    seeds, markers = gs.export()
    assert len(seeds) == 0 and len(markers) == 0
    db, gs = gpu_sketch(psk, [])
    assert len(gs.export()[0]) == 0


def test_sketch_one_sync_path_and_its_fallback(psk, oracle, monkeypatch):
    """A genome sketched on its own goes through the pipeline with ONE host synchronisation, its arrays sized from the expected seed
    and marker counts; input whose counts do not fit (low complexity: a short period selects every k-mer or none) must fall back to
    the count-then-allocate path with the same result. Both against the oracle, and the general path (PSK_SKETCH_SMALL=0) too."""
    rng = np.random.default_rng(11)
    period = random_genome(rng, 37)
    cases = [
        [random_genome(rng, 12000)],
        [random_genome(rng, 700), random_genome(rng, 45000), random_genome(rng, 300), random_genome(rng, 20000)],      # short contigs are dropped (lib.rs:156)
        [period * 2000],                                      # 74 kb of a 37-base period: 37 distinct k-mers, each selected ~2 000 times or never
        [random_genome(rng, 5000) + b"ACGT" * 8000 + random_genome(rng, 5000)],
        [b"A" * 40000],
    ]
    for c, mc in ((30, 200), (8, 16), (125, 1000)):
        for contigs in cases:
            want = oracle.Sketch(contigs, c=c, marker_c=mc)
            for small in ("1", "0"):
                monkeypatch.setenv("PSK_SKETCH_SMALL", small)
                db = psk.Database(compression=c, marker_compression=mc)
                gs = db._sketch("x", contigs, True)
                assert_sketch_equal(gs, want)


def test_sketch_str_and_buffer_inputs(psk, oracle):
    rng = np.random.default_rng(5)
    seq = random_genome(rng, 30000)
    want = oracle.Sketch([seq])
    for obj in (seq.decode(), bytearray(seq), memoryview(seq), np.frombuffer(seq, dtype=np.uint8)):
        db, gs = gpu_sketch(psk, [obj])
        assert_sketch_equal(gs, want)


def test_invalid_k_is_value_error(psk):
    with pytest.raises(ValueError):
        db = psk.Database(k=17)
        db.sketch("x", b"A" * 1000)


def chain_gpu(psk, ref_contigs, q_contigs, **kw):
    db = psk.Database()
    db.sketch("ref", *ref_contigs)
    return db.query("q", *q_contigs, learned_ani=False, **kw)


def check_pair(psk, oracle, ref_contigs, q_contigs, **kw):
    r, q = oracle.Sketch(ref_contigs), oracle.Sketch(q_contigs)
    want = oracle.query([("ref", r)], q, **kw)
    got = chain_gpu(psk, ref_contigs, q_contigs, **kw)
    assert len(got) == len(want)
    if want:
        w, g = want[0][1], got[0]
        for f in INT_FIELDS:
            assert g._raw[f] == getattr(w, f), f
        assert abs(g.identity - w.ani) < 1e-6
        assert abs(g.query_fraction - w.af_query) < 1e-6
        assert abs(g.reference_fraction - w.af_ref) < 1e-6
    return got


@pytest.mark.parametrize("kw", [{}, {"median": True}, {"robust": True}])
def test_chain_ecoli_matches_oracle(psk, oracle, ecoli, kw):
    ec, k12 = ecoli
    got = check_pair(psk, oracle, [ec], [k12], **kw)
    assert len(got) == 1 and got[0].reference_name == "ref" and got[0].query_name == "q"


def test_chain_ecoli_reference_kat_fractions(psk, ecoli):
    """test_ani.py:35-40 through the HIP path: fractions to the reference's 4 places."""
    ec, k12 = ecoli
    db = psk.Database()
    db.sketch("EC590", ec)
    hits = db.query("K12", k12, learned_ani=False)
    assert len(hits) == 1
    assert abs(hits[0].reference_fraction - 0.9246) < 5e-5
    assert abs(hits[0].query_fraction - 0.9189) < 5e-5
    assert abs(hits[0].identity - 0.9946) < 5e-5     # test_no_learned_ani, test_ani.py:35-40
    med = db.query("K12", k12, median=True)
    assert abs(med[0].identity - 0.9995) < 1e-4      # test_median, test_ani.py:56-61 (see tests/test_oracle_kat.py)


def test_chain_multicontig_repeats_and_strands(psk, oracle):
    rng = np.random.default_rng(11)
    base = random_genome(rng, 300000)
    rep = random_genome(rng, 3000)
    # reference: 3 contigs with a repeated element; query: rearranged, partly reverse-complemented, mutated
    ref = [base[:120000] + rep + base[120000:150000], base[150000:260000] + rep, rep + base[260000:]]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    q1 = mutate(rng, base[50000:200000], 0.03, 0.0005)
    q2 = mutate(rng, base[200000:] + rep, 0.01)[::-1].translate(comp)
    check_pair(psk, oracle, ref, [q2, q1, b"ACGT" * 50])


def test_chain_many_repeat_copies_per_chunk(psk, oracle):
    """Seven diverged copies of a 12 kb element in the reference against one copy in the query: the query chunks
    each carry seven qualifying chain trees, more than the lane-per-chunk kernel keeps in registers, so they go
    through its overflow list to the wave kernel; plus a tandem case on both strands."""
    rng = np.random.default_rng(111)
    elem = random_genome(rng, 12000)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    ref = []
    parts = []
    for c in range(7):
        copy = mutate(rng, elem, 0.01 + 0.004 * c)
        if c % 3 == 2:
            copy = copy[::-1].translate(comp)
        parts.append(random_genome(rng, 20000 + 1000 * c) + copy)
    ref.append(b"".join(parts[:4]) + random_genome(rng, 5000))
    ref.append(b"".join(parts[4:]) + random_genome(rng, 30000))
    flank_l, flank_r = random_genome(rng, 60000), random_genome(rng, 60000)
    query = [flank_l + elem + flank_r, mutate(rng, ref[0][:50000], 0.02)]
    res = check_pair(psk, oracle, ref, query)
    assert res and res[0]._raw["n_intervals"] >= 2


def test_chain_unrelated_gives_no_hit(psk, oracle):
    rng = np.random.default_rng(12)
    a, b = random_genome(rng, 200000), random_genome(rng, 200000)
    assert check_pair(psk, oracle, [a], [b]) == []
    # cutoff=None screens at 0.80; a tiny genome (< 20 markers) is rescued unless faster_small
    small = a[:8000]
    check_pair(psk, oracle, [a], [small])
    check_pair(psk, oracle, [a], [small], faster_small=True)


def test_query_many_refs_matches_oracle(psk, oracle):
    rng = np.random.default_rng(13)
    anc = [random_genome(rng, 400000) for _ in range(3)]
    refs = []
    for f, a in enumerate(anc):
        for j, d in enumerate((0.002, 0.01, 0.03, 0.06, 0.1)):
            refs.append((f"f{f}_m{j}", mutate(rng, a, d, 0.0002)))
    qseq = mutate(rng, anc[1], 0.02)
    db = psk.Database()
    for name, s in refs:
        db.sketch(name, s)
    got = {h.reference_name: h for h in db.query("q", qseq, learned_ani=False)}
    osk = [(n, oracle.Sketch([s])) for n, s in refs]
    want = {n: r for n, r in oracle.query(osk, oracle.Sketch([qseq]))}
    assert set(got) == set(want) and len(want) == 5
    for n, w in want.items():
        for f in INT_FIELDS:
            assert got[n]._raw[f] == getattr(w, f), (n, f)
        assert abs(got[n].identity - w.ani) < 1e-6
    # screen counts through the C-ABI
    import ctypes as C
    q = db._sketch("q", [qseq], True)
    n = len(db)
    flags = np.zeros(n, np.uint8); shared = np.zeros(n, np.uint32)
    from pyskani_amd import _capi
    _capi.check(db._lib.psk_screen(db._h, q._h, 0.80, 1, flags.ctypes.data_as(C.c_void_p), shared.ctypes.data_as(C.c_void_p)))
    oq = oracle.Sketch([qseq])
    for i, (nme, r) in enumerate(osk):
        ok, sh = oracle.screen(oq, r, 0.80, True)
        assert sh == shared[i] and ok == bool(flags[i]), nme


def test_alternative_device_paths_agree(ecoli):
    """Every switchable device path must give the same integers: the default (lane-per-chunk DP), the wave-per-chunk DP
    (PSK_CHAIN_LANE=0), both ways of building the chunk table (PSK_CHUNK_HOPS=1 pointer chase, =0 head walk) the lane-serial transliteration of the oracle (PSK_CHAIN_SERIAL=1), and the radix-sorted k-mer index
    (PSK_INDEX_RADIX=1) against the one-workgroup-per-sketch builder, and the wide (lower bound, count) join format
    (PSK_JOIN=wide), the one-wave-per-pair reference-major join (PSK_JOIN_PAIRS=1) against the default four-tile packed merge join, and the one-workgroup-per-pair emit from the join's pair
    totals (PSK_EMIT_PAIRS=1; the default for batches of >= 1 024 mid-sized pairs; with PSK_CHUNK_HOPS=0 it also builds the chunk
    table) against scan + emit over all items."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_fasta_first_record as L\n"
        "import pyskani_amd\n"
        "db = pyskani_amd.Database(); db.sketch('EC590', L('e.coli-EC590.fasta.gz'))\n"
        "h = db.query('K12', L('e.coli-K12.fasta.gz'), learned_ani=False)[0]\n"
        "print(h._raw['n_chunks'], h._raw['n_intervals'], h._raw['covered_query'], h._raw['covered_ref'], h._raw['sum_chain_anchors'], h._raw['sum_chunk_seeds'], repr(h.identity))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    outs = {}
    for name, extra in (("default", {}), ("wave_dp", {"PSK_CHAIN_LANE": "0"}), ("lane_dp", {"PSK_CHAIN_LANE": "64"}), ("quad_dp", {"PSK_CHAIN_LANE": "q"}), ("hops", {"PSK_CHUNK_HOPS": "1"}), ("hops_over_items", {"PSK_CHUNK_HOPS": "1", "PSK_HOPS_ITEMS": "1"}), ("hops_over_items_wide", {"PSK_CHUNK_HOPS": "1", "PSK_HOPS_ITEMS": "1", "PSK_JOIN": "wide"}), ("walk", {"PSK_CHUNK_HOPS": "0"}), ("serial", {"PSK_CHAIN_SERIAL": "1"}), ("radix_index", {"PSK_INDEX_RADIX": "1"}), ("wide_join", {"PSK_JOIN": "wide"}), ("pair_join", {"PSK_JOIN_PAIRS": "1"}), ("emit_per_pair", {"PSK_EMIT_PAIRS": "1"}), ("emit_per_pair_with_chunk_table", {"PSK_EMIT_PAIRS": "1", "PSK_CHUNK_HOPS": "0"}),
                        ("reduce_by_workgroup", {"PSK_REDUCE_WAVE": "0"}), ("reduce_by_workgroup_only", {"PSK_REDUCE_SMALL": "0"}), ("rows_in_table_order", {"PSK_ROW_SORT": "0"})):      # (the pair's ~230 chunk rows: reduced by one wave, four rows per lane, by default)
        env = dict(os.environ)
        for k in ("PSK_HOPS_ITEMS", "PSK_CHAIN_SERIAL", "PSK_CHAIN_LANE", "PSK_CHUNK_HOPS", "PSK_INDEX_RADIX", "PSK_JOIN", "PSK_JOIN_PAIRS", "PSK_EMIT_PAIRS", "PSK_REDUCE_WAVE", "PSK_REDUCE_SMALL", "PSK_ROW_SORT"):
            env.pop(k, None)
        env.update(extra)
        outs[name] = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=600).decode().strip()
    assert len(set(outs.values())) == 1, outs


@pytest.mark.parametrize("c", [3, 40])
def test_item_hops_dense_seeds(psk, oracle, monkeypatch, c):
    """The Gb-scale chunk table (waves hopping over item offsets, chunk_hops_items_kernel) where one 20 kb fragment holds more seeds than the
    wave stages at a time (c = 3: ~6 700 against a window of 4 096 - the successor is searched in global memory) and where a window holds several
    fragments (c = 40): contigs with and without anchors, a query contig shorter than a fragment."""
    monkeypatch.setenv("PSK_CHUNK_HOPS", "1"); monkeypatch.setenv("PSK_HOPS_ITEMS", "1")
    rng = np.random.default_rng(77 + c)
    anc = random_genome(rng, 260000)
    ref = [anc[:120000], mutate(rng, anc[120000:], 0.01)]
    qry = [mutate(rng, anc[5000:95000], 0.02), random_genome(rng, 30000), mutate(rng, anc[130000:142000], 0.01), mutate(rng, anc[150000:255000], 0.03, 0.0005)]
    r, q = oracle.Sketch(ref, c=c, marker_c=8 * c, k=15), oracle.Sketch(qry, c=c, marker_c=8 * c, k=15)
    want = oracle.query([("ref", r)], q)
    db = psk.Database(compression=c, marker_compression=8 * c, k=15)
    db.sketch("ref", *ref)
    got = db.query_sketches([db._sketch("q", qry, True)], learned_ani=False)[0]
    assert len(got) == len(want) == 1
    for f in INT_FIELDS:
        assert got[0]._raw[f] == getattr(want[0][1], f), (c, f)
    assert abs(got[0].identity - want[0][1].ani) < 1e-6


def test_learned_ani_without_model_raises(psk, ecoli):
    db = psk.Database()
    db.sketch("a", ecoli[0][:100000])
    with pytest.raises(RuntimeError):
        db.query("q", ecoli[0][:100000], learned_ani=True)
    with pytest.warns(RuntimeWarning):
        import pyskani_amd.database as D
        D._warned_no_model = False
        db.query("q", ecoli[0][:100000])


def test_full_size_properties(psk):
    """BASELINE configs[1]-sized property checks: self-query gives ANI 1 and AF ~1; ANI ordering follows divergence."""
    rng = np.random.default_rng(21)
    g = random_genome(rng, 5_000_000)
    db = psk.Database()
    db.sketch("self", g)
    for j, d in enumerate((0.005, 0.02, 0.05)):
        db.sketch(f"d{j}", mutate(rng, g, d))
    hits = {h.reference_name: h for h in db.query("q", g, learned_ani=False)}
    assert hits["self"].identity == 1.0 and hits["self"].query_fraction > 0.99
    assert hits["self"].identity > hits["d0"].identity > hits["d1"].identity > hits["d2"].identity
    for j, d in enumerate((0.005, 0.02, 0.05)):
        # conftest.mutate redraws the base uniformly, so the substitution rate is 0.75 d
        assert abs(hits[f"d{j}"].identity - (1 - 0.75 * d)) < 0.003


@pytest.mark.parametrize("k,c,mc", [(16, 60, 500), (11, 200, 800), (13, 100, 1000), (15, 250, 1000)])
def test_chain_other_sketch_parameters(psk, oracle, k, c, mc):
    """Chaining away from the defaults: look-back bands 41 (wave-per-chunk DP), 12, 25 (one over the lane kernels'
    window) and 10; k-mer widths 22 to 32 bits through the bucketed index."""
    rng = np.random.default_rng(900 + k)
    anc = random_genome(rng, 500000)
    ref = [anc[:200000], mutate(rng, anc[200000:], 0.004)]
    qry = [mutate(rng, anc[100000:450000], 0.03, 0.0003)]
    r, q = oracle.Sketch(ref, c=c, marker_c=mc, k=k), oracle.Sketch(qry, c=c, marker_c=mc, k=k)
    want = oracle.query([("ref", r)], q)
    db = psk.Database(compression=c, marker_compression=mc, k=k)
    db.sketch("ref", *ref)
    got = db.query("q", *qry, learned_ani=False)
    assert len(got) == len(want) == 1
    for f in INT_FIELDS:
        assert got[0]._raw[f] == getattr(want[0][1], f), f
    assert abs(got[0].identity - want[0][1].ani) < 1e-6


def test_metagenome_mode_short_contigs(psk, oracle, monkeypatch):
    """BASELINE configs[3] in miniature: c=30 / marker_c=200, short contigs as separate queries, both
    faster_small settings; every hit list must equal the oracle's."""
    rng = np.random.default_rng(31)
    anc = [random_genome(rng, 300000) for _ in range(2)]
    refs = [(f"r{f}_{j}", mutate(rng, a, d)) for f, a in enumerate(anc) for j, d in enumerate((0.0, 0.02, 0.05))]
    db = psk.Database(compression=30, marker_compression=200)
    for n, s in refs:
        db.sketch(n, s)
    osk = [(n, oracle.Sketch([s], c=30, marker_c=200)) for n, s in refs]
    contigs = []
    for i in range(6):
        L = int(np.exp(rng.uniform(np.log(2000), np.log(50000))))
        a = anc[i % 2]; st = int(rng.integers(0, len(a) - L))
        contigs.append((f"ctg{i}", mutate(rng, a[st:st + L], rng.uniform(0, 0.05))))
    for fs, prefilter in ((False, None), (True, None), (False, "1")):      # "1": seed prefilter of the rescued contigs forced on (default from 2^20 pairs)
        if prefilter:
            monkeypatch.setenv("PSK_PREFILTER", prefilter)
        got_all = db.query_many(contigs, learned_ani=False, faster_small=fs)
        for (name, seq), got in zip(contigs, got_all):
            want = {n: r for n, r in oracle.query(osk, oracle.Sketch([seq], c=30, marker_c=200), faster_small=fs)}
            g = {h.reference_name: h for h in got}
            assert set(g) == set(want), (name, fs, set(g) ^ set(want))
            for n, w in want.items():
                for f in INT_FIELDS:
                    assert g[n]._raw[f] == getattr(w, f), (name, n, f)
                assert abs(g[n].identity - w.ani) < 1e-6 and abs(g[n].query_fraction - w.af_query) < 1e-6
            # query_many == repeated query
            single = {h.reference_name: h.identity for h in db.query(name, seq, learned_ani=False, faster_small=fs)}
            assert single == {n: h.identity for n, h in g.items()}


def test_all_vs_all_query_many_matches_oracle(psk, oracle):
    """BASELINE configs[2] in miniature: every genome queried against a database of all of them in ONE
    psk_query_many call (pairs with different queries share kernel launches)."""
    rng = np.random.default_rng(41)
    anc = [random_genome(rng, 250000), random_genome(rng, 180000), random_genome(rng, 90000)]
    genomes = []
    for f, a in enumerate(anc):
        for j, d in enumerate((0.0, 0.01, 0.04, 0.08)):
            m = mutate(rng, a, d, 0.0003)
            cut = int(rng.integers(20000, len(m) - 20000))
            genomes.append((f"g{f}_{j}", [m[:cut], m[cut:]] if j % 2 else [m]))
    db = psk.Database()
    for n, contigs in genomes:
        db.sketch(n, *contigs)
    # the batched screen has four implementations (workgroup per pair / inverted marker index with the count row in LDS, looked up by waves
    # through a bucket table or by one lane per marker / the same with a count matrix in HBM): all must agree
    per_mode = {}
    for mode in ("brute", "inv", "inv_lane", "inv_global"):
        os.environ["PSK_SCREEN"] = mode.split("_")[0]
        if mode == "inv_global":
            os.environ["PSK_SCREEN_GLOBAL"] = "1"
        if mode == "inv_lane":
            os.environ["PSK_SCREEN_WAVE"] = "0"
        try:
            per_mode[mode] = db.query_many([(n, *contigs) for n, contigs in genomes], learned_ani=False)
        finally:
            os.environ.pop("PSK_SCREEN", None); os.environ.pop("PSK_SCREEN_GLOBAL", None); os.environ.pop("PSK_SCREEN_WAVE", None)
    for mode in ("inv", "inv_lane", "inv_global"):
        assert [[(h.reference_name, h.identity) for h in hs] for hs in per_mode["brute"]] == [[(h.reference_name, h.identity) for h in hs] for hs in per_mode[mode]], mode
    got_all = per_mode["inv"]
    osk = [(n, oracle.Sketch(contigs)) for n, contigs in genomes]
    n_hits = 0
    for (n, contigs), got, (_, oq) in zip(genomes, got_all, osk):
        want = {rn: r for rn, r in oracle.query(osk, oq)}
        g = {h.reference_name: h for h in got}
        assert set(g) == set(want), (n, set(g) ^ set(want))
        assert all(h.query_name == n for h in got)
        for rn, w in want.items():
            for f in INT_FIELDS:
                assert g[rn]._raw[f] == getattr(w, f), (n, rn, f)
            assert abs(g[rn].identity - w.ani) < 1e-6 and abs(g[rn].reference_fraction - w.af_ref) < 1e-6
        n_hits += len(got)
    assert n_hits >= 3 * 16 - 6     # every within-family pair is a hit (a few 8 %-vs-8 % pairs may fall below 0.15 AF)


def test_chain_large_pair_global_selection(psk, oracle):
    """> 1 024 candidate chains per pair (genomes beyond ~10 Mb) leave the LDS selection kernel for the
    workgroup-per-pair kernel on global scratch; it must agree with the oracle, conflicts included."""
    rng = np.random.default_rng(51)
    base = random_genome(rng, 24_000_000)
    rep = random_genome(rng, 6000)
    ref = [base[:9_000_000] + rep, base[9_000_000:17_000_000] + rep + base[17_000_000:]]
    q = mutate(rng, base[:12_000_000] + rep + base[12_000_000:] + rep, 0.02, 0.0002)
    got = check_pair(psk, oracle, ref, [q[:15_000_000], q[15_000_000:]])
    assert len(got) == 1 and got[0]._raw["n_chunks"] > 1024
    check_pair(psk, oracle, ref, [q[:15_000_000], q[15_000_000:]], median=True)


def test_median_and_robust_beyond_lds_capacity(psk, oracle):
    """> 4 096 chunk values per pair: median / trimmed mean sort in global scratch instead of LDS."""
    rng = np.random.default_rng(52)
    base = random_genome(rng, 86_000_000)
    q = mutate(rng, base, 0.03)
    for kw in ({"median": True}, {"robust": True}, {}):
        got = check_pair(psk, oracle, [base], [q], **kw)
        assert len(got) == 1 and got[0]._raw["n_chunks"] > 4096


def test_concurrent_queries_from_threads(psk, oracle):
    """`query` takes &self in the reference (lib.rs:551): concurrent queries from Python threads are legal — its only route
    to parallelism. ctypes releases the GIL around the C call and the library runs each call on its own execution lane
    (stream + scratch), so they overlap on the device; the FIRST queries also race to build the shared k-mer indexes and
    device tables, which the library does under the database's exclusive lock."""
    import threading
    rng = np.random.default_rng(61)
    anc = [random_genome(rng, 200000), random_genome(rng, 150000)]
    db = psk.Database()
    for f, a in enumerate(anc):
        for j, d in enumerate((0.0, 0.02, 0.05)):
            db.sketch(f"r{f}_{j}", mutate(rng, a, d))
    queries = [mutate(rng, anc[i % 2], 0.01 * (i + 1)) for i in range(8)]
    got = [None] * len(queries)
    errors = []

    def work(i):
        try:
            for _ in range(3):
                got[i] = sorted((h.reference_name, h.identity, h._raw["n_anchors"]) for h in db.query(f"q{i}", queries[i], learned_ani=False))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(queries))]      # nothing is indexed yet: the threads race on it
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    want = [sorted((h.reference_name, h.identity, h._raw["n_anchors"]) for h in db.query(f"q{i}", q, learned_ani=False)) for i, q in enumerate(queries)]
    assert got == want and all(len(w) == 3 for w in want)
    # borrow rules of the PyO3 class: a writer excludes everything, readers exclude writers
    with psk.Database._Borrow(db, False):
        with pytest.raises(RuntimeError, match="Already borrowed"):
            db.sketch("x", anc[0])
        db.query("ok", queries[0], learned_ani=False)                       # a second reader is fine
    with psk.Database._Borrow(db, True):
        with pytest.raises(RuntimeError, match="Already mutably borrowed"):
            db.query("q", queries[0], learned_ani=False)
        with pytest.raises(RuntimeError, match="Already borrowed"):
            db.sketch("x", anc[0])
    db.sketch("x", anc[0])
    assert len(db) == 7


def test_threads_share_query_sketches_and_large_batches(psk):
    """Stress of what round 3 made asynchronous: the k-mer index of a single query sketch is launched without a wait (another lane that
    meets the same sketch waits for the build's event), and the hits of a batch of more than 4 096 pairs cross on a copy stream while
    the next batch runs. Eight threads query the SAME `Sketch` objects one at a time in different orders while two more run the whole
    set as one batch; every result must equal the single-threaded one."""
    import threading
    rng = np.random.default_rng(63)
    anc = random_genome(rng, 150000)
    db = psk.Database(compression=30, marker_compression=200)
    db.sketch_many([(f"r{j}", mutate(rng, anc, 0.0005 * j)) for j in range(120)])      # 120 references of one family: every contig chains against all of them
    contigs = []
    for i in range(64):
        L = int(rng.integers(5000, 20000)); st = int(rng.integers(0, len(anc) - L))
        contigs.append(mutate(rng, anc[st:st + L], 0.01))

    def key(hits):
        return [(h.reference_name, h.identity, int(h._raw["n_anchors"]), int(h._raw["sum_chain_anchors"])) for h in hits]
    want = [key(h) for h in db.query_sketches([db.sketch_only(f"c{i}", c) for i, c in enumerate(contigs)], learned_ani=False)]
    assert sum(len(w) for w in want) > 4096      # more pairs than the host-filtered small batch takes
    for rep in range(2):
        sketches = [db.sketch_only(f"c{i}", c) for i, c in enumerate(contigs)]      # fresh: nothing indexed, the threads race on every one of them
        errors = []

        def one_by_one(t):
            try:
                order = np.random.default_rng(100 + t).permutation(len(sketches))
                for i in order:
                    got = key(db.query_sketches([sketches[int(i)]], learned_ani=False)[0])
                    if got != want[int(i)]:
                        errors.append(("single", t, int(i)))
            except Exception as e:      # noqa: BLE001
                errors.append(e)

        def all_at_once(t):
            try:
                for _ in range(3):
                    got = [key(h) for h in db.query_sketches(sketches, learned_ani=False)]      # 64 x 120 = 7 680 pairs: the large-batch path
                    if got != want:
                        errors.append(("batch", t))
            except Exception as e:      # noqa: BLE001
                errors.append(e)
        threads = [threading.Thread(target=one_by_one, args=(t,)) for t in range(8)] + [threading.Thread(target=all_at_once, args=(t,)) for t in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors[:5]


def test_lanes_overlap_queries(psk):
    """Throughput of per-contig queries from 1 vs 4 host threads against one database (reported, and required not to be
    slower: with one stream per context the threads used to serialise completely)."""
    import threading, time
    rng = np.random.default_rng(62)
    anc = random_genome(rng, 2_000_000)
    db = psk.Database(compression=30, marker_compression=200)
    db.sketch_many([(f"r{j}", mutate(rng, anc, 0.002 * j)) for j in range(20)])
    contigs = []
    for i in range(256):
        a = int(rng.integers(0, len(anc) - 20000))
        contigs.append(mutate(rng, anc[a:a + int(rng.integers(3000, 20000))], 0.01))
    want = [len(db.query("c", c, learned_ani=False)) for c in contigs]

    def run(n_threads):
        got = [None] * len(contigs)

        def work(t):
            for i in range(t, len(contigs), n_threads):
                got[i] = len(db.query("c", contigs[i], learned_ani=False))
        ts = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        assert got == want
        return len(contigs) / dt
    r1 = run(1); r4 = run(4); r1b = run(1)
    print(f"\nper-contig queries/s: 1 thread {r1:.0f} / {r1b:.0f}, 4 threads {r4:.0f}")
    assert r4 > 0.9 * max(r1, r1b)


def test_min_records_equal_the_raw_records(psk):
    """psk_query_many_min (20-byte records: what the reference's Hit holds, hit.rs:77-104 - the default of the record-level entry points) against
    psk_query_many (80 bytes with every chaining integer): the same hits in the same order, the same three floats bit for bit, `query` = the
    index of the hit's query in the call. A batch the host filters (few pairs) and one the device selects (> 4 096 pairs: the records are converted
    by the selection itself), over two calls on one database (the per-round numbering starts again in every call)."""
    import ctypes as C
    rng = np.random.default_rng(2024)
    anc = [random_genome(rng, 120_000) for _ in range(3)]
    genomes = [(f"g{f}_{j}", mutate(rng, anc[f], 0.003 * j)) for f in range(3) for j in range(40)]      # 120 genomes: 3 x 1 600 pairs in one batch
    db = psk.Database()
    db.sketch_many(genomes)
    sk = db._sketch_many(genomes, True)
    for part in (sk[:5], sk):
        n = len(part)
        handles = (C.c_void_p * n)(*[x._h for x in part])
        raw, roffs = db.query_handles(handles, n, learned_ani=False, raw=True)
        small, soffs = db.query_handles(handles, n, learned_ani=False)
        assert raw.dtype.itemsize == 80 and small.dtype.itemsize == 20
        assert len(raw) == len(small) >= n * 20 and np.array_equal(roffs, soffs)
        for f in ("ani", "af_query", "af_ref"):
            assert np.array_equal(raw[f].view(np.uint32), small[f].view(np.uint32)), f
        assert np.array_equal(raw["ref_index"], small["ref_index"])
        assert np.array_equal(small["query"], np.repeat(np.arange(n, dtype=np.uint32), np.diff(soffs)))      # (no model: bit 31 is clear)
        assert not raw["learned"].any() and not raw["reserved"].any()


def test_min_records_number_their_queries_through_the_whole_call():
    """psk_hit_min.query is the index of the hit's query in the CALL, also when the call runs as several rounds of queries (here 37 per round, and chain batches of
    2^18 seeds): every round numbers its own queries from zero on the device, the host adds the round's first index."""
    code = r'''
import sys, os, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np
from conftest import random_genome, mutate
import pyskani_amd as psk
rng = np.random.default_rng(31)
anc = [random_genome(rng, 80_000) for _ in range(2)]
genomes = [(f"g{i}", mutate(rng, anc[i %% 2], 0.002 * (i // 2))) for i in range(100)]
db = psk.Database()
db.sketch_many(genomes)
sk = db._sketch_many(genomes, True)
h = (C.c_void_p * 100)(*[x._h for x in sk])
raw, ro = db.query_handles(h, 100, learned_ani=False, raw=True)
small, so = db.query_handles(h, 100, learned_ani=False)
assert len(raw) == len(small) > 100 * 40 and np.array_equal(ro, so)
assert np.array_equal(small["query"], np.repeat(np.arange(100, dtype=np.uint32), np.diff(so)))
assert np.array_equal(raw["ani"].view(np.uint32), small["ani"].view(np.uint32)) and np.array_equal(raw["ref_index"], small["ref_index"])
print("ok", len(small))
''' % (ROOT, ROOT)
    env = dict(os.environ, PSK_ROUND_QUERIES="37", PSK_BATCH_ITEMS_LOG2="18")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_sharded_database_over_rccl_world1(psk):
    """The N>1 code path on the real backend: torch.distributed "nccl" (= RCCL) with a one-rank group, real
    Database underneath. (Two ranks cannot share this box's single GPU under RCCL; world 2 runs on gloo in
    tests/test_parallel_cpu.py.)"""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
from conftest import random_genome, mutate
import pyskani_amd as psk
from pyskani_amd.parallel import ShardedDatabase
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.default_rng(5)
anc = random_genome(rng, 150000)
refs = [mutate(rng, anc, d) for d in (0.0, 0.03, 0.06)] + [random_genome(rng, 150000)]
names = ["r%%d" %% i for i in range(len(refs))]
sdb = ShardedDatabase(dist, device=dev)
assert sdb.coll_device == dev
sdb.sketch_all(names, lambda i: (refs[i],))
q = mutate(rng, anc, 0.01)
got = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in sdb.query("q", q, learned_ani=False)]
db = psk.Database()
for n, r in zip(names, refs): db.sketch(n, r)
want = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in db.query("q", q, learned_ani=False)]
assert got == want and len(got) == 3, (got, want)
# the same exchange steps through the library's OWN RCCL communicator (psk_comm_create / psk_gather_hits / psk_gather_sketches)
sdb2 = ShardedDatabase(dist, device=dev, comm="capi")
sdb2.sketch_all(names, lambda i: (refs[i],))
got2 = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in sdb2.query("q", q, learned_ani=False)]
assert got2 == want, (got2, want)
for s in (sdb, sdb2):
    ava = s.all_vs_all(batch=3, learned_ani=False)
    full = db.query_many(list(zip(names, refs)), learned_ani=False)
    for n, hits in zip(names, full):
        w = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in hits]
        g = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in ava[n]]
        assert g == w, (s.comm.kind, n, g, w)
assert sdb2.comm.kind == "capi" and sdb2.stats["collective_s"] > 0
# the exchange's records: 20 bytes by default (psk_hit_min), the 80-byte psk_hit with every chaining integer on request - the same hits either way
small = sdb2.all_vs_all_records(batch=3, learned_ani=False)
sdb3 = ShardedDatabase(dist, device=dev, comm=sdb2.comm, raw=True)
sdb3.sketch_all(names, lambda i: (refs[i],))
big = sdb3.all_vs_all_records(batch=3, learned_ani=False)
assert small.dtype.itemsize == 20 and big.dtype.itemsize == 80 and len(small) == len(big) > 0
assert np.array_equal(small["ani"].view(np.uint32), big["ani"].view(np.uint32)) and np.array_equal(small["ref_index"], big["ref_index"])
assert np.array_equal(small["query"], big["reserved"]) and big["n_anchors"].min() > 0
sdb2.comm.close()
dist.destroy_process_group()
print("sharded ok")
''' % (ROOT, ROOT)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sharded ok" in r.stdout, r.stdout + r.stderr


def test_sharded_all_vs_all_two_ranks_share_the_gpu(psk):
    """ShardedDatabase.all_vs_all with two ranks (gloo, both on this box's one GPU): shards sketch their own genomes,
    exchange sketch records, query them against the local shard, gather hits. Must equal one Database.query_many."""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch, torch.distributed as dist
from conftest import random_genome, mutate
import pyskani_amd as psk
from pyskani_amd.parallel import ShardedDatabase
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.default_rng(17)
anc = [random_genome(rng, 120000), random_genome(rng, 90000)]
genomes = [mutate(rng, anc[i %% 2], 0.01 * (i // 2)) for i in range(7)]
names = ["g%%d" %% i for i in range(7)]
sdb = ShardedDatabase(dist, device=torch.device("cuda", 0))
n_local = sdb.sketch_all(names, lambda i: (genomes[i],))
assert n_local == (4 if rank == 0 else 3)
got = sdb.all_vs_all(batch=2, learned_ani=False)
db = psk.Database()
for n, g in zip(names, genomes): db.sketch(n, g)
want = db.query_many(list(zip(names, genomes)), learned_ani=False)
for n, hits in zip(names, want):
    w = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in hits]
    g = [(h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in got[n]]
    assert g == w, (n, g, w)
assert sum(len(v) for v in got.values()) >= 7 + 12
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "all-vs-all ok")
''' % (ROOT, ROOT)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("all-vs-all ok" in o for o in outs), outs
