"""Host-side mirror of pyskani's `Database` / `Hit` / `Sketch` classes for the hot path.

Same names, argument meaning and error behaviour as the PyO3 classes in
/root/reference/src/pyskani/_skani/{lib,hit,sketch}.rs; all arithmetic runs in HIP kernels
behind the C-ABI of include/pyskani_amd.h. Persistence (`path=`, `load/open/save/flush`)
keeps the reference's file names and flush semantics with this build's own byte format (storage.py).
"""
import collections
import ctypes as C
import itertools
import operator
import os
import pathlib
import threading
import warnings

import numpy as np

from . import _capi
from . import storage as _storage

_CHARP1, _U64_1, _HITP = C.c_char_p * 1, C.c_uint64 * 1, C.POINTER(_capi.Hit)

try:      # C-level construction of a query's Hit list (csrc/hitlist.c, built beside the library); host logic only - the Python construction below is its twin
    from . import _hitlist
except ImportError:
    _hitlist = None

_ctx_lock = threading.Lock()
_ctxs = {}


class Context:
    """One GPU: owns the psk_ctx handle."""

    def __init__(self, device=0):
        self._lib = _capi.load()
        h = C.c_void_p()
        _capi.check(self._lib.psk_ctx_create(device, C.byref(h)))
        self._h = h
        self.device = device

    def close(self):
        """Destroy the context. Every Database / Sketch created on it must be gone first: their device blocks
        return to the context's pool when they are freed. Contexts are deliberately NOT destroyed by the garbage
        collector (interpreter shutdown frees objects in arbitrary order); process exit releases the GPU memory."""
        if getattr(self, "_h", None):
            self._lib.psk_ctx_destroy(self._h)
            self._h = None

    def synchronize(self):
        _capi.check(self._lib.psk_ctx_synchronize(self._h))


def default_context(device=0):
    with _ctx_lock:
        if device not in _ctxs:
            _ctxs[device] = Context(device)
        return _ctxs[device]


def release_default_context(device=0):
    """Destroy the shared context of `device` (and with it the device blocks its pool keeps for reuse). Every Database, Sketch and
    Model made on it must be gone. For programs that run several large jobs one after another in one process (bench.py)."""
    with _ctx_lock:
        ctx = _ctxs.pop(device, None)
    if ctx is not None:
        ctx.close()


def _as_bytes(obj):
    """utils::Text::new (utils.rs:74-102): str -> UTF-8 view, bytes/bytearray -> view, buffers -> copy."""
    if isinstance(obj, str):
        return obj.encode("utf-8")
    if isinstance(obj, bytes):
        return obj
    if isinstance(obj, bytearray):
        return bytes(obj)
    try:
        return bytes(memoryview(obj))
    except TypeError:
        raise TypeError(f"expected str, bytes, bytearray or buffer, found {type(obj).__name__}")


class _PyHitBase:
    """Pure-Python twin of csrc/hitlist.c's HitBase (used when the C module is not built): the same constructor, the same read-only fields."""

    __slots__ = ("_f",)

    def __new__(cls, identity, query_name, query_fraction, reference_name, reference_fraction, learned=False, keep=None, idx=0):
        self = object.__new__(cls)
        f32 = lambda x: float(np.float32(x))
        object.__setattr__(self, "_f", (f32(identity), query_name, f32(query_fraction), reference_name, f32(reference_fraction), bool(learned), keep, int(idx)))
        return self

    def __setattr__(self, name, value):
        raise AttributeError(f"attribute '{name}' of 'Hit' objects is not writable")

    identity = property(lambda self: self._f[0])
    query_name = property(lambda self: self._f[1])
    query_fraction = property(lambda self: self._f[2])
    reference_name = property(lambda self: self._f[3])
    reference_fraction = property(lambda self: self._f[4])
    learned = property(lambda self: self._f[5])
    _keep = property(lambda self: self._f[6])
    _idx = property(lambda self: self._f[7])


_HIT_BASE = _hitlist.HitBase if _hitlist is not None else _PyHitBase      # (fixed at import: what Hit is made of)


class Hit(_HIT_BASE):
    """A single hit found when querying a `Database` with a genome (hit.rs:18-104): `identity`, `query_name`, `query_fraction`,
    `reference_name`, `reference_fraction`, read-only like the reference's getters (hit.rs:77-104).

    The fields sit in a C structure (csrc/hitlist.c: HitBase - one allocation per hit) so that a query's hits are built by ONE C-level pass
    over the library's records (`Hit._from_records`): with a Python-level constructor per hit, 200 hits cost more than the GPU work of the
    query that found them. Like the reference's class, hits compare and hash by identity, not by value, and are no sequences.
    `learned` is not in the reference: True when `identity` came out of the learned-ANI regression model, False when it is the raw chain ANI
    (the reference applies skani's embedded model by default, lib.rs:611-614; this build needs the model file)."""

    __slots__ = ()

    def __new__(cls, identity, query_name, query_fraction, reference_name, reference_fraction):
        identity = float(np.float32(identity))
        query_fraction = float(np.float32(query_fraction))
        reference_fraction = float(np.float32(reference_fraction))
        if identity < 0.0 or identity > 1.0:                      # hit.rs:34-37
            raise ValueError(f"Invalid value for `identity`: {identity}")
        if query_fraction < 0.0 or query_fraction > 1.0:          # hit.rs:38-41
            raise ValueError(f"Invalid value for `query_fraction`: {query_fraction}")
        if reference_fraction < 0.0 or reference_fraction > 1.0:  # hit.rs:42-48
            raise ValueError(f"Invalid value for `reference_fraction`: {reference_fraction}")
        return super().__new__(cls, identity, str(query_name), query_fraction, str(reference_name), reference_fraction)

    @classmethod
    def _from_records(cls, recs, qname, names):
        """[Hit] from a numpy array of psk_hit records. Values come from the library and are valid by construction, so the range
        checks of the constructor are skipped; the integer intermediates stay reachable as `hit._raw[field]` (the record, looked up on demand)."""
        n = len(recs)
        if _hitlist is not None and isinstance(names, list):      # the same objects, built in C (csrc/hitlist.c)
            return _hitlist.build(cls, recs, n, qname, names, recs)
        return list(map(_HIT_BASE.__new__, itertools.repeat(cls, n), recs["ani"].tolist(), itertools.repeat(qname), recs["af_query"].tolist(),
                        map(names.__getitem__, recs["ref_index"].tolist()), recs["af_ref"].tolist(), (recs["learned"] != 0).tolist(), itertools.repeat(recs), range(n)))

    def __repr__(self):  # hit.rs:61-74
        return ("Hit(identity={!r}, query_name={!r}, query_fraction={!r}, reference_name={!r}, "
                "reference_fraction={!r})").format(self.identity, self.query_name, self.query_fraction,
                                                    self.reference_name, self.reference_fraction)

    def __reduce__(self):
        return (_hit_restore, ((self.identity, self.query_name, self.query_fraction, self.reference_name, self.reference_fraction, self.learned),))

    @property
    def _raw(self):
        """the psk_hit record behind the hit (numpy void: chaining integers, raw ANI), None for a hit built by hand"""
        src = self._keep
        if type(src) is bytes:      # the records of a per-contig query, kept as the bytes the library returned (_hitlist.query_host)
            src = np.frombuffer(src, dtype=Database._HIT_DTYPE)
        return None if src is None else src[self._idx]


def _hit_restore(fields):
    return _HIT_BASE.__new__(Hit, *fields)


class Sketch:
    """A sketched genome resident in HBM (sketch.rs:4-31)."""

    def __init__(self, ctx, handle, name, owned=True):
        self._ctx = ctx
        self._h = handle
        self._name = name
        self._owned = owned

    def __del__(self):
        if getattr(self, "_owned", False) and getattr(self, "_h", None):
            self._ctx._lib.psk_sketch_free(self._h)
            self._h = None

    def _info(self):
        p = _capi.Params()
        ns, nm, tl, nc = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        _capi.check(self._ctx._lib.psk_sketch_info(self._h, C.byref(p), C.byref(ns), C.byref(nm), C.byref(tl), C.byref(nc)))
        return p, ns.value, nm.value, tl.value, nc.value

    @property
    def name(self):
        return self._name

    @property
    def c(self):
        return self._info()[0].c

    @property
    def amino_acid(self):
        return False  # use_aa is hard-wired false, lib.rs:416

    def to_record(self):
        """Host copy as a storage.Record (params, name, contig lengths, seeds, markers)."""
        p, ns, nm, tl, nc = self._info()
        lens = np.zeros(nc, dtype="<u4")
        _capi.check(self._ctx._lib.psk_sketch_contig_lens(self._h, lens.ctypes.data_as(C.c_void_p)))
        seeds, markers = self.export()
        return _storage.Record((p.c, p.marker_c, p.k), self._name, lens, seeds, markers)

    @classmethod
    def from_record(cls, ctx, record, markers_only=False):
        """psk_sketch_import: a device-resident sketch from a storage.Record."""
        params = _capi.Params(*record.params)
        seeds = np.zeros(0, _storage.SEED_DTYPE) if markers_only else record.seeds
        h = C.c_void_p()
        _capi.check(ctx._lib.psk_sketch_import(
            ctx._h, C.byref(params), record.contig_lens.ctypes.data_as(C.c_void_p), len(record.contig_lens),
            seeds.ctypes.data_as(C.c_void_p), len(seeds), record.markers.ctypes.data_as(C.c_void_p), len(record.markers),
            0 if markers_only else 1, C.byref(h)))
        return cls(ctx, h, record.name)

    def pack_size(self):
        """Bytes of this sketch as a packed device record (psk_sketch_pack_size)."""
        n = C.c_uint64()
        _capi.check(self._ctx._lib.psk_sketch_pack_size(self._h, C.byref(n)))
        return n.value

    def pack_into(self, device_ptr, capacity):
        """Write the packed record at a 16-byte aligned DEVICE address (e.g. inside a torch uint8 tensor)."""
        _capi.check(self._ctx._lib.psk_sketch_pack(self._h, C.c_void_p(device_ptr), capacity))

    @classmethod
    def unpack(cls, ctx, device_ptr, offsets, names, capacity=0):
        """Packed records at device_ptr + offsets[i] -> device-resident sketches (psk_sketch_unpack). `capacity`: the bytes readable at
        device_ptr - a record whose header points beyond them is refused (0: unchecked, the caller vouches for the records)."""
        n = len(offsets)
        offs = (C.c_uint64 * max(n, 1))(*[int(o) for o in offsets])
        out = (C.c_void_p * max(n, 1))()
        _capi.check(ctx._lib.psk_sketch_unpack(ctx._h, C.c_void_p(device_ptr), int(capacity), offs, n, out))
        return [cls(ctx, C.c_void_p(out[i]), names[i]) for i in range(n)]

    def export(self):
        """(seeds, markers) copied back to the host — used by the parity tests."""
        _, ns, nm, _, _ = self._info()
        seeds = np.zeros(ns, dtype=[("kmer", "<u4"), ("pos", "<u4"), ("contig", "<u4"), ("canon", "<u4")])
        markers = np.zeros(nm, dtype=np.uint64)
        _capi.check(self._ctx._lib.psk_sketch_export(self._h, seeds.ctypes.data_as(C.c_void_p), markers.ctypes.data_as(C.c_void_p)))
        return seeds, markers


class Model:
    """Learned-ANI regression model (skani::regression::get_model, lib.rs:614), resident in HBM.

    skani's trained gradient-boosted trees are embedded in the skani crate, which is not part of the reference tree,
    so the weights must be supplied: `Model.from_file(path)` reads the serde-JSON of a `gbdt::GBDT` (crate gbdt 0.1.3,
    Cargo.lock:1608); `Model.from_trees` takes explicit trees. Inference runs on the GPU (csrc/model.hip)."""

    def __init__(self, ctx, handle):
        self._ctx, self._h = ctx, handle

    def __del__(self):
        if getattr(self, "_h", None):
            self._ctx._lib.psk_model_free(self._h)
            self._h = None

    @classmethod
    def from_file(cls, path, device=0):
        ctx = default_context(device)
        h = C.c_void_p()
        _capi.check(ctx._lib.psk_model_load_file(ctx._h, os.fsencode(path), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_json(cls, text, device=0):
        ctx = default_context(device)
        data = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        h = C.c_void_p()
        _capi.check(ctx._lib.psk_model_load_json(ctx._h, data, len(data), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_trees(cls, trees, bias=0.0, shrinkage=1.0, features=None, device=0):
        """trees: list of node lists [(feature, threshold, left, right, value, missing, is_leaf), ...], children
        relative to the tree; features: names from _capi.FEATURE_NAMES (default: _capi.DEFAULT_FEATURES)."""
        ctx = default_context(device)
        flat = [n for t in trees for n in t]
        nodes = (_capi.TreeNode * max(len(flat), 1))(*[_capi.TreeNode(*n, 0) for n in flat])
        first = (C.c_uint32 * max(len(trees), 1))(*np.concatenate([[0], np.cumsum([len(t) for t in trees])])[:len(trees)].astype(int).tolist())
        feats = None
        if features is not None:
            feats = (C.c_int32 * len(features))(*[_capi.FEATURE_NAMES.index(f) for f in features])
        h = C.c_void_p()
        _capi.check(ctx._lib.psk_model_create(ctx._h, nodes, len(flat), first, len(trees), bias, shrinkage, feats,
                                              len(features) if features is not None else 0, C.byref(h)))
        return cls(ctx, h)

    @property
    def shape(self):
        nt, nn, nf = C.c_uint32(), C.c_uint64(), C.c_uint32()
        _capi.check(self._ctx._lib.psk_model_info(self._h, C.byref(nt), C.byref(nn), C.byref(nf)))
        return nt.value, nn.value, nf.value

    def predict(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, self.shape[2])
        out = np.zeros(len(rows), np.float32)
        _capi.check(self._ctx._lib.psk_model_predict(self._h, rows.ctypes.data_as(C.c_void_p), len(rows), out.ctypes.data_as(C.c_void_p)))
        return out


class Database:
    """A database storing sketched genomes (lib.rs:132-137, 368-660): sketches live in HBM; with a `path`
    they are also written to a folder in the reference's layout (storage.py)."""

    def __init__(self, path=None, *, compression=125, marker_compression=1000, k=15, format=None, device=0, model=None):
        """`device` and `model` are additions to the reference signature (lib.rs:369): the GPU to use, and the
        learned-ANI regression model (a `Model` or a path to a gbdt JSON file; default: $PSK_MODEL_PATH if set)."""
        self._ctx = default_context(device)
        self._model = None
        self._warned_no_model = False
        self._opts_cache = {}
        self._cache_lock = threading.Lock()
        self._device = device
        self._lib = self._ctx._lib
        self._params = _capi.Params(int(compression), int(marker_compression), int(k))
        # PyO3 borrow rules of the reference: `sketch` takes &mut self (lib.rs:479), `query` takes &self (lib.rs:551).
        # Concurrent queries are fine (they overlap on the GPU's execution lanes); a sketch while anything else runs, or
        # anything while a sketch runs, raises the same RuntimeError PyO3 raises.
        self._borrow_lock = threading.Lock()
        self._readers = 0
        self._writer = False
        self._names = []                # insertion order = markers Vec (lib.rs:501-504)
        self._resident = []             # per ref: True when the full sketch (not only its markers) sits in HBM
        self._n_lazy = 0                # references that are NOT resident (an `open`ed database): 0 = every query goes through psk_query
        self._cache = collections.OrderedDict()   # lazily loaded sketches of a disk-backed database
        self._h = None
        if path is None:
            self._storage = None        # DatabaseStorage::Memory, lib.rs:378
        else:
            folder = os.fsdecode(path)
            if not os.path.exists(folder):
                try:
                    os.makedirs(folder)
                except OSError as err:  # lib.rs:385-391
                    raise OSError(err.errno, f"Failed to create {folder}") from err
            if os.path.exists(os.path.join(folder, "markers.bin")):        # lib.rs:395-399
                raise FileExistsError(os.path.join(folder, "markers.bin"))
            kind = "consolidated" if format is None else format             # lib.rs:400-403
            if kind == "consolidated":
                self._storage = _storage.Consolidated(folder)
            elif kind == "separated":
                self._storage = _storage.Folder(folder)
            else:
                raise ValueError(f"invalid format: {kind}")                 # lib.rs:407-409
        if path is None and format not in (None, "consolidated", "separated"):
            raise ValueError(f"invalid format: {format}")
        h = C.c_void_p()
        _capi.check(self._lib.psk_db_create(self._ctx._h, C.byref(self._params), C.byref(h)))
        self._h = h
        # addresses the C-level query call needs (csrc/hitlist.c does not link against the library); $PSK_PY_FASTCALL=0: the ctypes route (tests, A/B)
        self._fast = None
        if os.environ.get("PSK_PY_FASTCALL", "1") != "0":
            self._fast = (C.cast(self._lib.psk_query_host, C.c_void_p).value, C.cast(self._lib.psk_free, C.c_void_p).value)
        if model is None and os.environ.get("PSK_MODEL_PATH"):
            model = os.environ["PSK_MODEL_PATH"]
        if model is not None:
            self.load_model(model)

    def load_model(self, model):
        """Attach the learned-ANI regression model used by `query` (lib.rs:611-614): a `Model`, or the path of a
        gbdt JSON file. `None` detaches it."""
        if model is not None and not isinstance(model, Model):
            model = Model.from_file(model, device=self._device)
        self._model = model
        self._opts_cache = {}

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.psk_db_destroy(self._h)
            self._h = None

    class _Borrow:
        def __init__(self, db, mutable):
            self.db, self.mutable = db, mutable

        def __enter__(self):
            db = self.db
            with db._borrow_lock:
                if self.mutable:
                    if db._writer or db._readers:
                        raise RuntimeError("Already borrowed")
                    db._writer = True
                else:
                    if db._writer:
                        raise RuntimeError("Already mutably borrowed")
                    db._readers += 1

        def __exit__(self, *exc):
            db = self.db
            with db._borrow_lock:
                if self.mutable:
                    db._writer = False
                else:
                    db._readers -= 1
            return False

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.flush()                    # lib.rs:426-434
        return False

    @property
    def path(self):
        """`pathlib.Path` or `None`: where sketches are stored (lib.rs:437-452)."""
        return None if self._storage is None else pathlib.Path(self._storage.path)

    @property
    def compression(self):
        return self._params.c           # lib.rs:456-458

    @property
    def marker_compression(self):
        return self._params.marker_c    # lib.rs:462-464

    def __len__(self):
        return self._lib.psk_db_size(self._h)

    # -- persistence ---------------------------------------------------------
    def _marker_records(self):
        recs = []
        for i, name in enumerate(self._names):
            sk = Sketch(self._ctx, C.c_void_p(self._lib.psk_db_sketch(self._h, i)), name, owned=False)
            recs.append(sk.to_record().markers_only())
        return recs

    def flush(self):
        """Memory: nothing. Folder: markers.bin. Consolidated: markers.bin + index.db (lib.rs:217-227, 734-740)."""
        if self._storage is not None:
            self._storage.flush((self._params.c, self._params.marker_c, self._params.k), self._marker_records())

    @classmethod
    def open(cls, path, device=0):
        """Markers in HBM, full sketches read from the folder when a query shortlists them (lib.rs:294-337)."""
        folder = os.fsdecode(path)
        params, records = _storage.read_markers(os.path.join(folder, "markers.bin"))
        db = cls(compression=params[0], marker_compression=params[1], k=params[2], device=device)
        index_path, sk_path = os.path.join(folder, "index.db"), os.path.join(folder, "sketches.db")
        if os.path.exists(index_path) and os.path.exists(sk_path):          # lib.rs:313-328
            db._storage = _storage.Consolidated(folder, _storage.read_index(index_path))
        else:
            db._storage = _storage.Folder(folder)
        for rec in records:
            sk = Sketch.from_record(db._ctx, rec, markers_only=True)
            sk._owned = False
            _capi.check(db._lib.psk_db_add(db._h, rec.name.encode("utf-8"), sk._h))
            db._names.append(rec.name)
            db._resident.append(False)
            db._n_lazy += 1
        return db

    @classmethod
    def load(cls, path, device=0):
        """`open`, then every sketch pulled into memory; the result is an in-memory database (lib.rs:249-275)."""
        disk = cls.open(path, device=device)
        db = cls(compression=disk._params.c, marker_compression=disk._params.marker_c, k=disk._params.k, device=device)
        for name in disk._names:
            sk = Sketch.from_record(db._ctx, disk._storage.load(name))
            sk._owned = False
            _capi.check(db._lib.psk_db_add(db._h, name.encode("utf-8"), sk._h))
            db._names.append(name)
            db._resident.append(True)
        return db

    def _full_sketch(self, i):
        """The chainable sketch of reference i: resident in the db, else read from disk (small LRU)."""
        if self._resident[i]:
            return Sketch(self._ctx, C.c_void_p(self._lib.psk_db_sketch(self._h, i)), self._names[i], owned=False)
        name = self._names[i]
        with self._cache_lock:                                              # `query` takes &self: callers may be concurrent
            sk = self._cache.get(name)
            if sk is not None:
                self._cache.move_to_end(name)
                return sk
        sk = Sketch.from_record(self._ctx, self._storage.load(name))        # KeyError / OSError as lib.rs:95-122
        with self._cache_lock:
            self._cache[name] = sk
            while len(self._cache) > 256:
                self._cache.popitem(last=False)
        return sk

    def save(self, path, overwrite=False, format=None):
        """Write the whole database to `path` (lib.rs:662-726). `format` follows the DOCUMENTED meaning
        ("consolidated" default, "separated" = one file per genome); the reference's match arms at
        lib.rs:696-700 are inverted relative to its own docstring and are not reproduced."""
        folder = os.fsdecode(path)
        if not os.path.exists(folder):
            try:
                os.makedirs(folder)
            except OSError as err:
                raise OSError(err.errno, f"Failed to create {folder}") from err
        markers_path = os.path.join(folder, "markers.bin")
        if not overwrite and os.path.exists(markers_path):
            raise FileExistsError(markers_path)                              # lib.rs:688-692
        kind = "consolidated" if format is None else format
        if kind not in ("consolidated", "separated"):
            raise ValueError(f"invalid format: {kind}")
        # a database opened from `folder` reads its sketches lazily from the very files this call replaces: every record
        # is read BEFORE the target is touched, and sketches.db is rebuilt under a temporary name and renamed at the end
        last = {name: i for i, name in enumerate(self._names)}             # the store is keyed by name (lib.rs:51-55)
        records = [self._full_sketch(last[name]).to_record() for name in dict.fromkeys(self._names)]
        if kind == "consolidated":
            target = _storage.Consolidated(folder, file_name="sketches.db.tmp")
            tmp = os.path.join(folder, "sketches.db.tmp")
            if os.path.exists(tmp):
                os.remove(tmp)
        else:
            target = _storage.Folder(folder)
        for rec in records:
            target.store(rec)
        if kind == "consolidated":
            if not os.path.exists(tmp):
                open(tmp, "wb").close()
            os.replace(tmp, os.path.join(folder, "sketches.db"))
            target.file_name = "sketches.db"
        by_name = {rec.name: rec for rec in records}
        markers = [by_name[name].markers_only() for name in self._names]
        target.flush((self._params.c, self._params.marker_c, self._params.k), markers)
        if self._storage is not None and os.path.realpath(self._storage.path) == os.path.realpath(folder):
            self._storage = target      # saved onto its own folder: the old offsets / files are gone
            with self._cache_lock:
                self._cache.clear()

    # -- the hot path ------------------------------------------------------------------------
    def _sketch(self, name, contigs, seed):
        """Database::_sketch (lib.rs:140-185): one genome from N contigs."""
        views = [_as_bytes(c) for c in contigs]
        n = len(views)
        arr = (C.c_char_p * max(n, 1))(*views)
        lens = (C.c_uint64 * max(n, 1))(*[len(v) for v in views])
        h = C.c_void_p()
        _capi.check(self._lib.psk_sketch_host(self._ctx._h, C.byref(self._params), arr, lens, n, int(bool(seed)), C.byref(h)))
        return Sketch(self._ctx, h, name)

    def _sketch_many(self, genomes, seed):
        """[(name, contig, ...)] -> [Sketch] through the pipelined host-ingest entry point (psk_sketch_many_host)."""
        views, first = [], [0]
        for g in genomes:
            if not isinstance(g[0], str):
                raise TypeError("name must be a str")
            views.extend(_as_bytes(c) for c in g[1:])
            first.append(len(views))
        n, nc = len(genomes), len(views)
        arr = (C.c_char_p * max(nc, 1))(*views)
        lens = (C.c_uint64 * max(nc, 1))(*[len(v) for v in views])
        gfc = (C.c_uint32 * (n + 1))(*first)
        out = (C.c_void_p * max(n, 1))()
        _capi.check(self._lib.psk_sketch_many_host(self._ctx._h, C.byref(self._params), arr, lens, gfc, n, int(bool(seed)), out))
        return [Sketch(self._ctx, C.c_void_p(out[i]), genomes[i][0]) for i in range(n)]

    def sketch_many(self, genomes, *, seed=True):
        """Add many reference genomes in one call: `genomes` = [(name, contig, ...), ...]. Same result as
        `for g in genomes: db.sketch(*g)` (lib.rs:477-510 applied per genome), but the contigs cross PCIe through a
        pinned, double-buffered pipeline that overlaps the copies with the sketch kernels — the way to load a large
        reference set from host memory. An addition to the reference API."""
        with Database._Borrow(self, True):
            sketches = self._sketch_many(genomes, seed)
            if self._storage is not None:
                for sk in sketches:
                    self._storage.store(sk.to_record())
            names = (C.c_char_p * max(len(sketches), 1))(*[sk.name.encode("utf-8") for sk in sketches])
            handles = (C.c_void_p * max(len(sketches), 1))(*[sk._h for sk in sketches])
            _capi.check(self._lib.psk_db_add_batch(self._h, names, handles, len(sketches)))
            for sk in sketches:
                sk._owned = False
                self._names.append(sk.name)
                self._resident.append(True)
        return None

    def sketch_many_device(self, names, device_ptr, contig_offsets, contig_lengths, genome_first_contig=None, *, seed=True):
        """Add many reference genomes whose ASCII already sits in HBM (psk_sketch_batch_device + psk_db_add_batch): contig i is
        bytes [contig_offsets[i], +contig_lengths[i]) of the device allocation at `device_ptr` (offsets 16-byte aligned, 16 bytes
        of slack after the last contig); genome g owns contigs [genome_first_contig[g], genome_first_contig[g+1]) (default: one
        contig per genome). For callers that stage genomes on the GPU themselves (bench.py, a multi-GPU shard)."""
        n = len(names)
        nc = len(contig_offsets)
        gfc = list(genome_first_contig) if genome_first_contig is not None else list(range(n + 1))
        if len(gfc) != n + 1 or gfc[-1] != nc or len(contig_lengths) != nc:
            raise ValueError("genome_first_contig must hold len(names) + 1 entries ending at the number of contigs")
        with Database._Borrow(self, True):
            c_off = (C.c_uint64 * max(nc, 1))(*[int(x) for x in contig_offsets])
            c_len = (C.c_uint64 * max(nc, 1))(*[int(x) for x in contig_lengths])
            c_gfc = (C.c_uint32 * (n + 1))(*gfc)
            out = (C.c_void_p * max(n, 1))()
            _capi.check(self._lib.psk_sketch_batch_device(self._ctx._h, C.byref(self._params), C.c_void_p(device_ptr), c_off, c_len, c_gfc, n, int(bool(seed)), out))
            if self._storage is not None:
                for i in range(n):
                    self._storage.store(Sketch(self._ctx, C.c_void_p(out[i]), names[i], owned=False).to_record())
            c_names = (C.c_char_p * max(n, 1))(*[nm.encode("utf-8") for nm in names])
            _capi.check(self._lib.psk_db_add_batch(self._h, c_names, out, n))
            self._names.extend(names)
            self._resident.extend([True] * n)
        return None

    def sketch(self, name, *contigs, seed=True):
        """Add a reference genome to the database (lib.rs:477-510)."""
        if not isinstance(name, str):
            raise TypeError("name must be a str")
        with Database._Borrow(self, True):          # PyO3's &mut self borrow
            sk = self._sketch(name, contigs, seed)
            if self._storage is not None:              # lib.rs:505-508: written at once, markers only on flush
                self._storage.store(sk.to_record())
            sk._owned = False                          # ownership moves into the db (lib.rs:501-508)
            _capi.check(self._lib.psk_db_add(self._h, name.encode("utf-8"), sk._h))
            self._names.append(name)
            self._resident.append(True)
        return None

    def _opts(self, learned_ani, median, robust, cutoff, faster_small):
        # default rule: learned ANI when c >= 70 and not median (lib.rs:611-613, docstring :522-527)
        learned = bool(learned_ani) if learned_ani is not None else (self._params.c >= 70 and not median)
        if learned and self._model is None:
            if learned_ani is not None:       # explicit learned_ani=True without a model
                raise RuntimeError("learned_ani=True needs a regression model: skani's GBDT weights are embedded in the "
                                   "skani crate; load them with Database(model=...), Database.load_model() or $PSK_MODEL_PATH")
            # default call: the reference would apply skani's embedded model here. Say so once per Database and mark
            # every Hit (Hit.learned is False): identities are the raw chain ANI, ~7e-4 above the regressed value on
            # the reference's own E. coli test pair (test_ani.py:28-40).
            if not self._warned_no_model:
                warnings.warn("pyskani_amd: no learned-ANI regression model is loaded, so identities are the RAW chain ANI "
                              "(what pyskani returns for learned_ani=False), not pyskani's default regressed value; "
                              "pass learned_ani=False to accept that explicitly, or load a model "
                              "(Database(model=...), Database.load_model(), $PSK_MODEL_PATH)", RuntimeWarning, stacklevel=3)
                self._warned_no_model = True
            learned = False
        key = (learned, bool(median), bool(robust), bool(faster_small), float(cutoff) if cutoff else 0.0, id(self._model) if learned else 0)
        o = self._opts_cache.get(key)
        if o is None:      # (a ctypes structure per distinct flag set, not per call: per-contig queries are microseconds apart)
            o = _capi.QueryOpts(1 if learned else 0, int(bool(median)), int(bool(robust)), int(bool(faster_small)),
                                float(cutoff) if cutoff else 0.0, 0.0, self._model._h if (learned and self._model is not None) else None)
            o._keep = self._model
            if len(self._opts_cache) < 64:
                self._opts_cache[key] = o
        return o

    _HIT_DTYPE = np.dtype(_capi.Hit)
    _HIT_MIN_DTYPE = np.dtype(_capi.HitMin)

    def _hit(self, r, qname):
        """One psk_hit (ctypes struct) -> Hit (the lazy path: a handful of hits)."""
        return self._hits(np.frombuffer(bytes(r), dtype=self._HIT_DTYPE), qname)[0]

    def _hits(self, recs, qname):
        """A numpy array of psk_hit records -> [Hit] (one C-level pass: Hit._from_records)."""
        return Hit._from_records(recs, qname, self._names)

    def _hits_from_ptr(self, hits_p, lo, hi, qname):
        if hi <= lo:
            return []
        return self._hits(_capi.hit_records(hits_p, lo, hi, self._HIT_DTYPE), qname)

    def query_many(self, genomes, *, seed=True, learned_ani=None, median=False, robust=False, cutoff=None,
                   faster_small=False):
        """[(name, contigs...)] -> list of hit lists; equals [self.query(name, *contigs, ...) for ...].
        An addition to the reference API for all-vs-all / many-bin workloads."""
        sketches = self._sketch_many(genomes, seed)
        return self.query_sketches(sketches, learned_ani=learned_ani, median=median, robust=robust, cutoff=cutoff,
                                   faster_small=faster_small)

    def sketch_only(self, name, *contigs, seed=True):
        """Sketch a genome WITHOUT adding it: the `Sketch` can be queried (`query_sketches`) or shipped to another
        rank as a record (`Sketch.to_record().to_bytes()`), e.g. by `parallel.ShardedDatabase.all_vs_all`."""
        return self._sketch(name, contigs, seed)

    def query_sketches(self, sketches, *, learned_ani=None, median=False, robust=False, cutoff=None, faster_small=False):
        """query_many for genomes that are already sketched (`Sketch` objects made with this database's parameters)."""
        opts = self._opts(learned_ani, median, robust, cutoff, faster_small)
        n = len(sketches)
        with Database._Borrow(self, False):
            return self._query_sketches(sketches, opts, n)

    # ---- record-level entry points (no Python object per hit): what parallel.ShardedDatabase and bench.py use
    def sketch_handles(self):
        """ctypes array of the psk_sketch* of every reference, in insertion order (borrowed: the database owns them)."""
        if self._n_lazy:
            raise RuntimeError("sketch_handles needs a memory-resident database")
        n = len(self._names)
        return (C.c_void_p * max(n, 1))(*[self._lib.psk_db_sketch(self._h, i) for i in range(n)])

    def query_handles(self, handles, n, *, learned_ani=None, median=False, robust=False, cutoff=None, faster_small=False, raw=False):
        """psk_query_many_min over n raw sketch handles -> (records, offsets): a numpy array of 20-byte psk_hit_min records (own memory;
        `query` = index of the hit's query among the handles, bit 31 = learned) and the n+1 int64 offsets of every query's hits in it.
        raw=True: psk_query_many, the 80-byte psk_hit records with every chaining integer (parity tests)."""
        opts = self._opts(learned_ani, median, robust, cutoff, faster_small)
        with Database._Borrow(self, False):
            hits_p = C.POINTER(_capi.Hit if raw else _capi.HitMin)()
            offs = (C.c_uint64 * (n + 1))()
            _capi.check((self._lib.psk_query_many if raw else self._lib.psk_query_many_min)(self._h, handles, n, C.byref(opts), C.byref(hits_p), offs))
            try:
                total = int(offs[n])
                recs = _capi.hit_records(hits_p, 0, total, self._HIT_DTYPE if raw else self._HIT_MIN_DTYPE)
            finally:
                if hits_p:
                    self._lib.psk_free(hits_p)
        return recs, np.frombuffer(offs, dtype=np.uint64).astype(np.int64)

    def query_records(self, name, *contigs, seed=True, learned_ani=None, median=False, robust=False, cutoff=None, faster_small=False):
        """Database.query returning the psk_hit records (numpy structured array, ref_index = insertion index) instead of `Hit`s."""
        opts = self._opts(learned_ani, median, robust, cutoff, faster_small)
        with Database._Borrow(self, False):
            if self._n_lazy:
                raise RuntimeError("query_records needs a memory-resident database")
            return self._query_host(contigs, seed, opts)

    def _query_host(self, contigs, seed, opts):
        """lib.rs:571 + 569-659 in ONE library call (psk_query_host): the query is sketched from the caller's bytes (not stored) and
        queried; a contig-sized query runs as one launch sequence with one synchronisation. Returns the psk_hit records."""
        views = [_as_bytes(c) for c in contigs]
        nc = len(views)
        if nc == 1:      # (the per-contig call: array types made once, not per call)
            arr = _CHARP1(views[0]); lens = _U64_1(len(views[0]))
        else:
            arr = (C.c_char_p * max(nc, 1))(*views)
            lens = (C.c_uint64 * max(nc, 1))(*map(len, views))
        hits_p = _HITP()
        n = C.c_uint64(0)
        _capi.check(self._lib.psk_query_host(self._h, arr, lens, nc, 1 if seed else 0, C.byref(opts), C.byref(hits_p), C.byref(n)))
        try:      # (one C-level copy into an immutable bytes object; a structured np.empty + memmove costs five times as much for a hundred hits)
            return _capi.hit_records(hits_p, 0, n.value, self._HIT_DTYPE)      # (an array that owns its memory and can be written: callers add shard offsets in place)
        finally:
            if hits_p:
                self._lib.psk_free(hits_p)

    def _query_sketches(self, sketches, opts, n):
        if self._n_lazy:      # `open`ed database: sketches come from disk per query
            return [self._query_lazy(s.name, s, opts) for s in sketches]
        arr = (C.c_void_p * max(n, 1))(*[s._h for s in sketches])
        hits_p = C.POINTER(_capi.Hit)()
        offs = (C.c_uint64 * (n + 1))()
        _capi.check(self._lib.psk_query_many(self._h, arr, n, C.byref(opts), C.byref(hits_p), offs))
        try:
            return [self._hits_from_ptr(hits_p, offs[i], offs[i + 1], sketches[i].name) for i in range(n)]
        finally:
            if hits_p:
                self._lib.psk_free(hits_p)

    def _query_lazy(self, qname, q, opts):
        """Database.query for an `open`ed database: screen on the resident markers, read the shortlisted
        sketches from disk, chain them (lib.rs:617-657 with the Folder/Consolidated arms of `load`)."""
        n = len(self._names)
        if n == 0:
            return []
        flags = np.zeros(n, np.uint8)
        screen_val = opts.cutoff if opts.cutoff != 0.0 else 0.80
        _capi.check(self._lib.psk_screen(self._h, q._h, screen_val, int(not opts.faster_small), flags.ctypes.data_as(C.c_void_p), None))
        last = {nm: i for i, nm in enumerate(self._names)}                 # shortlist of NAMES, lib.rs:616-637
        idx = sorted({last[self._names[i]] for i in range(n) if flags[i]})
        if not idx:
            return []
        sketches = [self._full_sketch(i) for i in idx]
        arr = (C.c_void_p * len(idx))(*[s._h for s in sketches])
        res = (_capi.Hit * len(idx))()
        _capi.check(self._lib.psk_chain(self._ctx._h, arr, len(idx), q._h, C.byref(opts), res))
        out = []
        for i, r in zip(idx, res):
            if r.ani > 0.1:                                               # lib.rs:654
                r.ref_index = i
                out.append(self._hit(r, qname))
        return out

    def query(self, name, *contigs, seed=True, learned_ani=None, median=False, robust=False, cutoff=None,
              faster_small=False):
        """Query the database with a genome (lib.rs:549-660); returns a list of `Hit`."""
        if not isinstance(name, str):
            raise TypeError("name must be a str")
        with Database._Borrow(self, False):
            return self._query(name, contigs, seed, learned_ani, median, robust, cutoff, faster_small)

    def _query(self, name, contigs, seed, learned_ani, median, robust, cutoff, faster_small):
        opts = self._opts(learned_ani, median, robust, cutoff, faster_small)
        if self._n_lazy:
            return self._query_lazy(name, self._sketch(name, contigs, seed), opts)
        if _hitlist is not None and self._fast is not None:
            # the whole call in C (csrc/hitlist.c: arguments, psk_query_host with the interpreter lock released, the Hit list): what a query costs under the
            # lock decides how far concurrent per-contig queries scale (lib.rs:569)
            views = contigs if all(type(c) is bytes for c in contigs) else tuple(_as_bytes(c) for c in contigs)
            r = _hitlist.query_host(self._fast[0], self._fast[1], self._h.value, views, 1 if seed else 0, C.addressof(opts), Hit, name, self._names)
            if type(r) is int:
                _capi.check(r)
            return r
        recs = self._query_host(contigs, seed, opts)
        return self._hits(recs, name) if len(recs) else []
