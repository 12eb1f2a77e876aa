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
                ("n_markers", C.c_uint64), ("markers", C.POINTER(C.c_uint64)), ("kindex", C.c_void_p)]


class Node(C.Structure):
    _fields_ = [("feature", C.c_int32), ("threshold", C.c_float), ("left", C.c_int32), ("right", C.c_int32),
                ("value", C.c_float), ("missing", C.c_int32), ("is_leaf", C.c_int32)]


class _Model(C.Structure):
    _fields_ = [("nodes", C.POINTER(Node)), ("first", C.POINTER(C.c_uint32)), ("n_trees", C.c_uint32),
                ("n_features", C.c_uint32), ("features", C.POINTER(C.c_int32)), ("bias", C.c_float), ("shrinkage", C.c_float)]


class Model:
    """Flattened regression trees for orc_chain / orc_model_predict. `trees` = list of node lists
    [(feature, threshold, left, right, value, missing, is_leaf), ...] with children relative to the tree."""

    def __init__(self, trees, bias, shrinkage, features):
        flat = [n for t in trees for n in t]
        self._nodes = (Node * max(len(flat), 1))(*[Node(*n) for n in flat])
        firsts = np.concatenate([[0], np.cumsum([len(t) for t in trees])]).astype(np.uint32)
        self._first = (C.c_uint32 * len(firsts))(*firsts.tolist())
        self._feat = (C.c_int32 * len(features))(*features)
        self.c = _Model(self._nodes, self._first, len(trees), len(features), self._feat, bias, shrinkage)

    def predict(self, row):
        arr = (C.c_float * len(row))(*row)
        return lib().orc_model_predict(C.byref(self.c), arr)


class QueryOpts(C.Structure):
    _fields_ = [("learned_ani", C.c_int), ("median", C.c_int), ("robust", C.c_int),
                ("screen_val", C.c_double), ("rescue_small", C.c_int), ("min_aligned_frac", C.c_double),
                ("model", C.POINTER(_Model))]


class Result(C.Structure):
    _fields_ = [("ani", C.c_float), ("af_query", C.c_float), ("af_ref", C.c_float),
                ("n_anchors", C.c_uint64), ("n_chunks", C.c_uint32), ("n_intervals", C.c_uint32),
                ("covered_query", C.c_uint64), ("covered_ref", C.c_uint64),
                ("sum_chain_anchors", C.c_uint64), ("sum_chunk_seeds", C.c_uint64),
                ("ani_raw", C.c_float), ("ani_std", C.c_float), ("learned", C.c_uint32)]


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
        _lib.orc_model_predict.restype = C.c_float
        _lib.orc_model_predict.argtypes = [C.POINTER(_Model), C.POINTER(C.c_float)]
        _lib.orc_query_refs.restype = C.c_uint32
        _lib.orc_query_refs.argtypes = [C.POINTER(C.POINTER(_Sketch)), C.c_uint32, C.POINTER(_Sketch), C.POINTER(QueryOpts),
                                        C.POINTER(C.c_uint32), C.POINTER(Result), C.c_uint32]
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


def chain(ref, query, median=False, robust=False, min_aligned_frac=0.15, learned_ani=False, model=None):
    la = -1 if learned_ani is None else int(bool(learned_ani))
    o = QueryOpts(la, int(median), int(robust), 0.0, 1, min_aligned_frac, C.pointer(model.c) if model is not None else None)
    res = Result()
    rc = lib().orc_chain(ref._p, query._p, C.byref(o), C.byref(res))
    if rc != 0:
        raise RuntimeError("orc_chain failed: %d" % rc)
    return res


def query_count(ref_sketches, q, faster_small=False):
    """Number of hits of `q` against a list of Sketch objects; the whole screen + chain loop runs in C (no GIL held)."""
    arr = (C.POINTER(_Sketch) * len(ref_sketches))(*[r._p for r in ref_sketches])
    o = QueryOpts(0, 0, 0, 0.0, int(not faster_small), 0.15, None)
    return lib().orc_query_refs(arr, len(ref_sketches), q._p, C.byref(o), None, None, 0)


def last_chunks():
    p = C.c_void_p()
    n = lib().orc_last_chunks(C.byref(p))
    if n == 0:
        return np.zeros(0, dtype=chunk_dtype)
    buf = (C.c_char * (n * chunk_dtype.itemsize)).from_address(p.value)
    return np.frombuffer(buf, dtype=chunk_dtype).copy()


def query(refs, q, *, median=False, robust=False, cutoff=None, faster_small=False, learned_ani=False, model=None):
    """The screen + chain loops of Database.query (lib.rs:603-657) over a list of (name, Sketch). As in the
    reference, the marker list keeps every entry while the sketch store is keyed by name (lib.rs:51-55, 501-508):
    a name is shortlisted when ANY of its entries passes (lib.rs:616-637) and chained once, against its LAST sketch."""
    screen_val = cutoff if cutoff else 0.80
    store = {name: r for name, r in refs}              # later sketch of a name replaces the earlier one
    order = {name: i for i, (name, _) in enumerate(refs)}
    shortlist = set()
    for name, r in refs:
        ok, _ = screen(q, r, screen_val, not faster_small)
        if ok:
            shortlist.add(name)
    hits = []
    for name in sorted(shortlist, key=order.get):      # the reference's order is HashSet order, i.e. unspecified
        res = chain(store[name], q, median=median, robust=robust, learned_ani=learned_ani, model=model)
        if res.ani > 0.1:
            hits.append((name, res))
    return hits
