"""GPU: the pipelined host-ingest entry point (psk_sketch_many_host / Database.sketch_many) against per-genome
sketching and the oracle; 32-bit offset guards."""
import numpy as np
import pytest

from conftest import mutate, random_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psk():
    import pyskani_amd
    return pyskani_amd


def export_bytes(sk):
    s, m = sk.export()
    return s.tobytes(), m.tobytes(), sk._info()[1:]


@pytest.mark.parametrize("packed", ["0", "1"])
def test_sketch_many_equals_per_genome_sketch(psk, oracle, monkeypatch, packed):
    """Ragged genomes (empty, all-short, multi-contig, odd lengths) through the pipeline = psk_sketch_host one by one; as ASCII
    over PCIe and 2-bit packed on the host's worker threads (every byte that is not ACGT/acgt packs to 0, like the kernels' own table)."""
    monkeypatch.setenv("PSK_INGEST_PACKED", packed)
    rng = np.random.default_rng(31)
    genomes = [("g0", random_genome(rng, 300_001)),
               ("g1", random_genome(rng, 40_000), random_genome(rng, 499), random_genome(rng, 77_777), b""),
               ("g2",), ("g3", b"ACGT" * 100),
               ("g4", "acgtn" * 3000, bytearray(random_genome(rng, 16_385)), memoryview(random_genome(rng, 501)))]
    genomes += [(f"h{i}", random_genome(rng, int(rng.integers(500, 60_000)))) for i in range(40)]
    db = psk.Database()
    many = db._sketch_many(genomes, True)
    for g, sk in zip(genomes, many):
        one = db._sketch(g[0], g[1:], True)
        assert export_bytes(sk) == export_bytes(one), g[0]
        assert sk.name == g[0]
    o = oracle.Sketch([bytes(c, "ascii") if isinstance(c, str) else bytes(c) for c in genomes[4][1:]])
    s, m = many[4].export()
    assert np.array_equal(s["kmer"], o.seeds["kmer"]) and np.array_equal(s["pos"], o.seeds["pos"]) and np.array_equal(m, o.markers)


@pytest.mark.parametrize("packed", ["0", "1"])
def test_sketch_many_spans_sub_batches_and_slots(psk, monkeypatch, packed):
    """More ASCII than one sub-batch (192 MB of ASCII / 96 MB of packed words) and genomes straddling the 32 MB staging slots; 3 worker threads."""
    monkeypatch.setenv("PSK_INGEST_THREADS", "3")
    monkeypatch.setenv("PSK_INGEST_PACKED", packed)
    rng = np.random.default_rng(32)
    base = random_genome(rng, 9_000_000)
    genomes = [(f"g{i}", base[i * 1000: i * 1000 + 7_000_000 + 1013 * i], base[:600 + i]) for i in range(62)]   # ~440 MB
    db = psk.Database()
    many = db._sketch_many(genomes, True)
    for i in (0, 4, 5, 26, 27, 44, 54, 55, 61):
        assert export_bytes(many[i]) == export_bytes(db._sketch("x", genomes[i][1:], True)), i


def test_database_sketch_many_and_query(psk, tmp_path):
    rng = np.random.default_rng(33)
    anc = random_genome(rng, 150_000)
    genomes = [(f"r{i}", mutate(rng, anc, 0.01 * i)) for i in range(6)] + [("far", random_genome(rng, 150_000))]
    a, b = psk.Database(), psk.Database(str(tmp_path / "db"))
    for g in genomes:
        a.sketch(*g)
    b.sketch_many(genomes)
    b.flush()
    q = mutate(rng, anc, 0.005)
    ha = [(h.reference_name, h.identity, h.query_fraction) for h in a.query("q", q, learned_ani=False)]
    hb = [(h.reference_name, h.identity, h.query_fraction) for h in b.query("q", q, learned_ani=False)]
    assert ha == hb and len(ha) == 6
    hc = [(h.reference_name, h.identity, h.query_fraction) for h in psk.Database.load(str(tmp_path / "db")).query("q", q, learned_ani=False)]
    assert hc == ha
    with pytest.raises(TypeError):
        a.sketch_many([(b"bytes-name", q)])


def _repeat_genome(oracle, copies):
    """A 25-base unit whose central 15-mer is a seed (picked with the oracle), repeated `copies` times."""
    rng = np.random.default_rng(34)
    g = random_genome(rng, 50_000)
    s = oracle.Sketch([g]).seeds
    p = int(s["pos"][len(s) // 2])                 # window [p-20, p]
    unit = g[p - 22:p + 3]
    assert len(oracle.Sketch([unit * 40]).seeds) >= 40
    return unit * copies


def test_anchor_total_beyond_32_bits_is_refused(psk, oracle):
    """a k-mer present ~47k times on both sides gives > 2^31 anchors; the 32-bit offsets must not wrap silently."""
    g = _repeat_genome(oracle, 47_000)
    db = psk.Database()
    db.sketch("rep", g)
    with pytest.raises(OverflowError, match="anchors"):
        db.query("q", g, learned_ani=False)
    # ... and a moderately repetitive pair still chains, identically to the oracle
    small = _repeat_genome(oracle, 300)
    db2 = psk.Database()
    db2.sketch("rep", small)
    h = db2.query("q", small, learned_ani=False)
    want = oracle.chain(oracle.Sketch([small]), oracle.Sketch([small]))
    assert (len(h) == 1 and h[0]._raw["n_anchors"] == want.n_anchors and abs(h[0].identity - want.ani) < 1e-6) or (not h and want.ani <= 0.1)
