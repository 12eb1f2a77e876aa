"""CPU: the host-side pieces of the boundary that involve no GPU - the 2-bit packer of the ingest pipeline (csrc/pack_host.cpp,
`psk_pack2bit_host`) against a numpy restatement of the oracle's byte table (oracle/skani_oracle.c BYTE_TO_SEQ: A 0, C 1, G 2, T 3,
case-insensitive, everything else 0), and the C-level construction of a query's `Hit` list (csrc/hitlist.c) against its Python twin."""
import ctypes as C

import numpy as np
import pytest


def np_pack(seq):
    lut = np.zeros(256, np.uint32)
    for ch, v in ((b"C", 1), (b"c", 1), (b"G", 2), (b"g", 2), (b"T", 3), (b"t", 3)):
        lut[ch[0]] = v
    codes = lut[np.frombuffer(seq, np.uint8)]
    n = len(codes)
    words = np.zeros((n + 15) // 16, np.uint32)
    for i in range(n):
        words[i // 16] |= codes[i] << np.uint32(30 - 2 * (i % 16))
    return words


def lib_pack(lib, seq, mode):
    n = len(seq)
    out = np.full((n + 15) // 16 + 2, 0xDEADBEEF, np.uint32)          # two guard words: nothing is written past ceil(n / 16)
    lib.psk_pack2bit_host(C.c_char_p(seq), n, out.ctypes.data_as(C.c_void_p), mode)
    assert out[-1] == 0xDEADBEEF and out[-2] == 0xDEADBEEF
    return out[:-2]


@pytest.mark.parametrize("mode", [0, 1])
def test_pack2bit_matches_the_byte_table(mode):
    from pyskani_amd import _capi
    lib = _capi.load()
    rng = np.random.default_rng(3)
    every = bytes(range(256)) * 3                                      # every byte value, at every position of a word
    for seq in [b"", b"A", b"ACGT", b"acgtn" * 7, every, every[1:], every[5:700]]:
        assert np.array_equal(lib_pack(lib, seq, mode), np_pack(seq)), (mode, len(seq))
    for _ in range(40):
        n = int(rng.integers(1, 700))
        seq = np.frombuffer(b"ACGTacgtNn-*\x00\xff", np.uint8)[rng.integers(0, 14, n)].tobytes()
        assert np.array_equal(lib_pack(lib, seq, mode), np_pack(seq)), (mode, n)
    big = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 200_003)].tobytes()
    assert np.array_equal(lib_pack(lib, big, 0), lib_pack(lib, big, 1))


def test_hit_lists_built_in_c_equal_the_python_construction():
    from pyskani_amd import _capi, database
    if database._hitlist is None:
        pytest.skip("pyskani_amd/_hitlist is not built (make -C pyskani_amd/csrc)")
    dt = np.dtype(_capi.Hit)
    rng = np.random.default_rng(5)
    n = 300
    recs = np.zeros(n, dt)
    recs["ani"] = rng.random(n).astype(np.float32); recs["af_query"] = rng.random(n).astype(np.float32); recs["af_ref"] = rng.random(n).astype(np.float32)
    recs["ref_index"] = rng.integers(0, 50, n); recs["learned"] = rng.integers(0, 2, n); recs["n_anchors"] = rng.integers(0, 1 << 40, n)
    names = [f"ref{i}" for i in range(50)]
    fast = database.Hit._from_records(recs, "query", names)
    keep, database._hitlist = database._hitlist, None
    try:
        slow = database.Hit._from_records(recs, "query", names)
    finally:
        database._hitlist = keep
    assert len(fast) == len(slow) == n and all(type(h) is database.Hit for h in fast)
    for a, b in zip(fast, slow):
        assert (a.identity, a.query_name, a.query_fraction, a.reference_name, a.reference_fraction, a.learned) == \
               (b.identity, b.query_name, b.query_fraction, b.reference_name, b.reference_fraction, b.learned)
        assert a._raw["n_anchors"] == b._raw["n_anchors"] and repr(a) == repr(b)
    with pytest.raises(IndexError):
        database.Hit._from_records(recs, "query", names[:10])
    assert database.Hit._from_records(recs[:0], "q", names) == []


def test_c_level_query_call_passes_bytes_and_builds_hits_without_a_gpu():
    """_hitlist.query_host is the per-contig Database.query as one call from the interpreter (lib.rs:549-660): it hands the contigs' bytes to the
    function it was given, with the interpreter lock released, and turns the records into Hit objects. Here the function is a stand-in made with
    ctypes (no GPU, no library compute): what arrives and what comes back is checked, a failing status comes back as an int."""
    import ctypes as C
    from pyskani_amd import _capi, database
    if database._hitlist is None:
        pytest.skip("pyskani_amd/_hitlist is not built (make -C pyskani_amd/csrc)")
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p; libc.malloc.argtypes = [C.c_size_t]
    seen = {}
    PROTO = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64))

    def fake(db, contigs, lens, nc, seed, opts, hits, n):
        seen["db"], seen["seed"], seen["opts"] = db, seed, opts
        seen["contigs"] = [C.string_at(contigs[i], lens[i]) for i in range(nc)]
        if seen.get("fail"):
            return 7
        k = seen["n_out"]
        n[0] = k
        if k:
            buf = libc.malloc(k * C.sizeof(_capi.Hit))
            arr = (_capi.Hit * k).from_address(buf)
            for i in range(k):
                C.memset(C.addressof(arr[i]), 0, C.sizeof(_capi.Hit))
                arr[i].ani = 0.5 + 0.001 * i; arr[i].af_query = 0.25; arr[i].af_ref = 0.75; arr[i].ref_index = i % 3; arr[i].learned = i & 1; arr[i].n_anchors = 1000 + i
            hits[0] = buf
        else:
            hits[0] = None
        return 0
    cb = PROTO(fake)
    fn = C.cast(cb, C.c_void_p).value
    free_fn = C.cast(libc.free, C.c_void_p).value
    opts = _capi.QueryOpts()
    names = ["a", "b", "c"]
    seen["n_out"] = 40
    contigs = (b"ACGT" * 10, b"", b"TTTT" * 3) + tuple(b"G" * i for i in range(20))      # more than the call's on-stack arrays hold
    out = database._hitlist.query_host(fn, free_fn, 0x1234, contigs, 1, C.addressof(opts), database.Hit, "q", names)
    assert seen["db"] == 0x1234 and seen["seed"] == 1 and seen["opts"] == C.addressof(opts) and seen["contigs"] == list(contigs)
    assert len(out) == 40 and all(type(h) is database.Hit for h in out)
    for i, h in enumerate(out):
        assert h.reference_name == names[i % 3] and h.query_name == "q" and h.learned == bool(i & 1)
        assert h.identity == float(np.float32(0.5 + 0.001 * i)) and h.query_fraction == 0.25 and h.reference_fraction == 0.75
        assert h._raw["n_anchors"] == 1000 + i and h._raw["ref_index"] == i % 3
    seen["n_out"] = 0
    assert database._hitlist.query_host(fn, free_fn, 1, (b"ACGT",), 0, C.addressof(opts), database.Hit, "q", names) == [] and seen["seed"] == 0
    seen["fail"] = True
    assert database._hitlist.query_host(fn, free_fn, 1, (), 0, C.addressof(opts), database.Hit, "q", names) == 7
    with pytest.raises(TypeError):
        database._hitlist.query_host(fn, free_fn, 1, ("ACGT",), 0, C.addressof(opts), database.Hit, "q", names)
    with pytest.raises(TypeError):
        database._hitlist.query_host(fn, free_fn, 1, [b"ACGT"], 0, C.addressof(opts), database.Hit, "q", names)
    with pytest.raises(ValueError):
        database._hitlist.query_host(0, free_fn, 1, (b"ACGT",), 0, C.addressof(opts), database.Hit, "q", names)
    seen["fail"] = False; seen["n_out"] = 5
    with pytest.raises(IndexError):
        database._hitlist.query_host(fn, free_fn, 1, (b"ACGT",), 0, C.addressof(opts), database.Hit, "q", names[:1])
