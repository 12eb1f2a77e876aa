"""GPU: storage behaviour of `Database`: the file layout the reference's tests pin (as a table), error behaviour, and round trips
through open/load/save."""
import os
import pathlib

import numpy as np
import pytest

from conftest import mutate, random_genome

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def psk():
    import pyskani_amd
    return pyskani_amd


# What the reference's storage tests pin (src/pyskani/tests/test_database.py:11-42), as DATA: for every on-disk format, which files of the
# database folder exist after two genomes were sketched and which appear only with flush() (the marker file, and the consolidated format's index,
# are written on flush: lib.rs:187-227, 662-686). Inputs of 400 bases never reach the seeding code (lib.rs:156): this is file-layout behaviour only.
LAYOUT = {
    # format: (files present right after sketch(), files that flush() adds)
    "separated": (("{name}.sketch",), ("markers.bin",)),
    "consolidated": (("sketches.db",), ("index.db", "markers.bin")),
}
GENOMES = (("test1", b"ATGC" * 100), ("test2", b"TTGC" * 100))


def _present(folder, patterns):
    names = {pat.format(name=n) for pat in patterns for n, _ in GENOMES}
    return {f: os.path.exists(os.path.join(folder, f)) for f in sorted(names)}


def test_memory_database_has_no_path(psk):
    db = psk.Database()
    db.sketch(*GENOMES[0])
    assert db.path is None


@pytest.mark.parametrize("fmt", sorted(LAYOUT))
def test_folder_layout_before_and_after_flush(psk, tmp_path, fmt):
    at_once, on_flush = LAYOUT[fmt]
    folder = str(tmp_path)
    db = psk.Database(folder, format=fmt)
    for name, seq in GENOMES:
        db.sketch(name, seq)
    assert all(_present(folder, at_once).values()), _present(folder, at_once)
    assert not any(_present(folder, on_flush).values()), _present(folder, on_flush)
    db.flush()
    assert all(_present(folder, at_once + on_flush).values()), _present(folder, at_once + on_flush)
    assert db.path == pathlib.Path(folder)


def test_errors_like_the_reference(psk, tmp_path):
    with pytest.raises(ValueError):                              # lib.rs:407-409
        psk.Database(str(tmp_path / "x"), format="bogus")
    d = str(tmp_path / "dup")
    db = psk.Database(d)                                         # default format: consolidated (lib.rs:403)
    db.sketch("a", b"ACGT" * 200)
    with pytest.raises(ValueError):                              # duplicate name, lib.rs:66-72
        db.sketch("a", b"ACGT" * 200)
    db.flush()
    with pytest.raises(FileExistsError):                         # folder already holds markers.bin, lib.rs:395-399
        psk.Database(d)
    with pytest.raises(FileExistsError):                         # save without overwrite, lib.rs:688-692
        db.save(d)
    with pytest.raises(OSError):
        psk.Database.open(str(tmp_path / "missing"))


def _family(rng):
    anc = [random_genome(rng, 150000) for _ in range(2)]
    refs = [(f"r{f}_{j}", [mutate(rng, a, d)]) for f, a in enumerate(anc) for j, d in enumerate((0.0, 0.02, 0.06))]
    refs[1] = (refs[1][0], [refs[1][1][0][:60000], refs[1][1][0][60000:]])      # one multi-contig reference
    return anc, refs


def _hits(db, q):
    return sorted((h.reference_name, h.identity, h.query_fraction, h.reference_fraction) for h in db.query("q", q, learned_ani=False))


@pytest.mark.parametrize("fmt", ["separated", "consolidated"])
def test_round_trip_open_load_save(psk, tmp_path, fmt):
    rng = np.random.default_rng(7)
    anc, refs = _family(rng)
    q = mutate(rng, anc[0], 0.01)
    mem = psk.Database()
    for n, c in refs:
        mem.sketch(n, *c)
    want = _hits(mem, q)
    assert len(want) == 3
    # written while sketching, markers on context-manager exit (lib.rs:426-434)
    d1 = str(tmp_path / "d1")
    with psk.Database(d1, format=fmt) as db:
        for n, c in refs:
            db.sketch(n, *c)
        assert _hits(db, q) == want
    assert os.path.exists(os.path.join(d1, "markers.bin"))
    opened = psk.Database.open(d1)                               # markers resident, sketches read lazily
    assert opened.path == pathlib.Path(d1) and len(opened) == len(refs)
    assert opened.compression == 125 and opened.marker_compression == 1000
    assert _hits(opened, q) == want
    many = opened.query_many([("q", q)], learned_ani=False)
    assert sorted((h.reference_name, h.identity) for h in many[0]) == [(n, i) for n, i, _, _ in want]
    loaded = psk.Database.load(d1)                               # everything resident, in-memory database
    assert loaded.path is None and _hits(loaded, q) == want
    # save from memory in the other layout, reopen
    d2 = str(tmp_path / "d2")
    other = "consolidated" if fmt == "separated" else "separated"
    mem.save(d2, format=other)
    names = set(os.listdir(d2))
    assert "markers.bin" in names and (("sketches.db" in names and "index.db" in names) if other == "consolidated" else "r0_0.sketch" in names)
    assert _hits(psk.Database.open(d2), q) == want
    opened.save(str(tmp_path / "d3"), format="separated")       # save from a lazily opened database
    assert _hits(psk.Database.load(str(tmp_path / "d3")), q) == want


def test_sketch_record_round_trip_is_bit_exact(psk, tmp_path):
    rng = np.random.default_rng(8)
    contigs = [random_genome(rng, 40000), random_genome(rng, 700), b"ACGT" * 10]
    db = psk.Database(compression=30, marker_compression=200, k=13)
    sk = db._sketch("g", contigs, True)
    rec = sk.to_record()
    from pyskani_amd import storage
    back = storage.Record.from_bytes(rec.to_bytes())
    assert back.params == (30, 200, 13) and back.name == "g"
    assert np.array_equal(back.contig_lens, [40000, 700])
    assert np.array_equal(back.seeds, rec.seeds) and np.array_equal(back.markers, rec.markers)
    sk2 = psk.Sketch.from_record(db._ctx, back)
    s1, m1 = sk.export(); s2, m2 = sk2.export()
    assert np.array_equal(s1, s2) and np.array_equal(m1, m2)
    with pytest.raises(ValueError):
        storage.Record.from_bytes(b"XXXX" + rec.to_bytes()[4:])
