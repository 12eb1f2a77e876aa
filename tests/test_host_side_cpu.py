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
