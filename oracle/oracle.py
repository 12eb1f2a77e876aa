"""ctypes binding of the CPU oracle. TEST INFRASTRUCTURE ONLY: importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from pyskani_amd/."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libskani_oracle.so")

seed_dtype = np.dtype([("kmer", "<u4"), ("pos", "<u4"), ("contig", "<u4"), ("canon", "<u4")])
chunk_dtype = np.dtype([("contig", "<u4"), ("left", "<u4"), ("right", "<u4"), ("anchors", "<u4"),
                        ("seeds", "<u4"), ("n_intervals", "<u4")])


class _Sketch(C.Structure):
    _fields_ = [("c", C.c_int), ("marker_c", C.c_int), ("k", C.c_int), ("n_contigs", C.c_uint32),
                ("contig_len", C.POINTER(C.c_uint32)), ("total_len", C.c_uint64),
                ("n_seeds", C.c_uint64), ("seeds", C.c_void_p),
                ("n_markers", C.c_uint64), ("markers", C.POINTER(C.c_uint64))]


class QueryOpts(C.Structure):
    _fields_ = [("learned_ani", C.c_int), ("median", C.c_int), ("robust", C.c_int),
                ("screen_val", C.c_double), ("rescue_small", C.c_int), ("min_aligned_frac", C.c_double)]


class Result(C.Structure):
    _fields_ = [("ani", C.c_float), ("af_query", C.c_float), ("af_ref", C.c_float),
                ("n_anchors", C.c_uint64), ("n_chunks", C.c_uint32), ("n_intervals", C.c_uint32),
                ("covered_query", C.c_uint64), ("covered_ref", C.c_uint64),
                ("sum_chain_anchors", C.c_uint64), ("sum_chunk_seeds", C.c_uint64)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libskani_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_sketch_new.restype = C.POINTER(_Sketch)
        _lib.orc_sketch_new.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.c_uint32,
                                        C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.orc_sketch_free.argtypes = [C.POINTER(_Sketch)]
        _lib.orc_screen.argtypes = [C.POINTER(_Sketch), C.POINTER(_Sketch), C.c_double, C.c_int,
                                    C.POINTER(C.c_uint64)]
        _lib.orc_chain.argtypes = [C.POINTER(_Sketch), C.POINTER(_Sketch), C.POINTER(QueryOpts),
                                   C.POINTER(Result)]
        _lib.orc_mm_hash64.restype = C.c_uint64
        _lib.orc_mm_hash64.argtypes = [C.c_uint64]
        _lib.orc_last_chunks.restype = C.c_uint32
        _lib.orc_last_chunks.argtypes = [C.POINTER(C.c_void_p)]
    return _lib


class Sketch:
    """Owns an orc_sketch. Mirrors what `Database._sketch` builds (lib.rs:140-185)."""

    def __init__(self, contigs, c=125, marker_c=1000, k=15, seed=True):
        contigs = [bytes(x) for x in contigs]
        self._keep = contigs
        n = len(contigs)
        arr = (C.c_char_p * max(n, 1))(*contigs)
        lens = (C.c_uint64 * max(n, 1))(*[len(x) for x in contigs])
        self._p = lib().orc_sketch_new(arr, lens, n, c, marker_c, k, int(seed))
        if not self._p:
            raise ValueError("invalid sketch parameters")

    def __del__(self):
        if getattr(self, "_p", None):
            lib().orc_sketch_free(self._p)
            self._p = None

    @property
    def seeds(self):
        s = self._p.contents
        if s.n_seeds == 0:
            return np.zeros(0, dtype=seed_dtype)
        buf = (C.c_char * (s.n_seeds * seed_dtype.itemsize)).from_address(s.seeds)
        return np.frombuffer(buf, dtype=seed_dtype).copy()

    @property
    def markers(self):
        s = self._p.contents
        return np.ctypeslib.as_array(s.markers, shape=(s.n_markers,)).copy() if s.n_markers else np.zeros(0, np.uint64)

    @property
    def total_len(self):
        return self._p.contents.total_len

    @property
    def contig_lens(self):
        s = self._p.contents
        return np.ctypeslib.as_array(s.contig_len, shape=(s.n_contigs,)).copy() if s.n_contigs else np.zeros(0, np.uint32)


def screen(q, r, screen_val=0.80, rescue_small=True):
    shared = C.c_uint64(0)
    ok = lib().orc_screen(q._p, r._p, screen_val, int(rescue_small), C.byref(shared))
    return bool(ok), shared.value


def chain(ref, query, median=False, robust=False, min_aligned_frac=0.15):
    o = QueryOpts(0, int(median), int(robust), 0.0, 1, min_aligned_frac)
    res = Result()
    rc = lib().orc_chain(ref._p, query._p, C.byref(o), C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_chain failed: %d" % rc)
    return res


def last_chunks():
    p = C.c_void_p()
    n = lib().orc_last_chunks(C.byref(p))
    if n == 0:
        return np.zeros(0, dtype=chunk_dtype)
    buf = (C.c_char * (n * chunk_dtype.itemsize)).from_address(p.value)
    return np.frombuffer(buf, dtype=chunk_dtype).copy()


def query(refs, q, *, median=False, robust=False, cutoff=None, faster_small=False):
    """The screen + chain loops of Database.query (lib.rs:603-657) over a list of (name, Sketch)."""
    screen_val = cutoff if cutoff else 0.80
    hits = []
    for name, r in refs:
        ok, _ = screen(q, r, screen_val, not faster_small)
        if not ok:
            continue
        res = chain(r, q, median=median, robust=robust)
        if res.ani > 0.1:
            hits.append((name, res))
    return hits
