"""Host-side mirror of pyskani's `Database` / `Hit` / `Sketch` classes for the hot path.

Same names, argument meaning and error behaviour as the PyO3 classes in
/root/reference/src/pyskani/_skani/{lib,hit,sketch}.rs; all arithmetic runs in HIP kernels
behind the C-ABI of include/pyskani_amd.h. Persistence (`path=`, `load/open/save/flush`)
keeps the reference's file names and flush semantics with this build's own byte format (storage.py).
"""
import collections
import ctypes as C
import os
import pathlib
import threading
import warnings

import numpy as np

from . import _capi
from . import storage as _storage

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


class Hit:
    """A single hit found when querying a `Database` with a genome (hit.rs:18-104)."""

    __slots__ = ("_identity", "_query_name", "_query_fraction", "_reference_name", "_reference_fraction", "_raw")

    def __init__(self, identity, query_name, query_fraction, reference_name, reference_fraction):
        identity = float(np.float32(identity))
        query_fraction = float(np.float32(query_fraction))
        reference_fraction = float(np.float32(reference_fraction))
        if identity < 0.0 or identity > 1.0:                      # hit.rs:34-37
            raise ValueError(f"Invalid value for `identity`: {identity}")
        if query_fraction < 0.0 or query_fraction > 1.0:          # hit.rs:38-41
            raise ValueError(f"Invalid value for `query_fraction`: {query_fraction}")
        if reference_fraction < 0.0 or reference_fraction > 1.0:  # hit.rs:42-48
            raise ValueError(f"Invalid value for `reference_fraction`: {reference_fraction}")
        self._identity = identity
        self._query_name = str(query_name)
        self._query_fraction = query_fraction
        self._reference_name = str(reference_name)
        self._reference_fraction = reference_fraction
        self._raw = None

    def __repr__(self):  # hit.rs:61-74
        return ("Hit(identity={!r}, query_name={!r}, query_fraction={!r}, reference_name={!r}, "
                "reference_fraction={!r})").format(self.identity, self.query_name, self.query_fraction,
                                                    self.reference_name, self.reference_fraction)

    identity = property(lambda self: self._identity)
    query_name = property(lambda self: self._query_name)
    query_fraction = property(lambda self: self._query_fraction)
    reference_name = property(lambda self: self._reference_name)
    reference_fraction = property(lambda self: self._reference_fraction)


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

    def export(self):
        """(seeds, markers) copied back to the host — used by the parity tests."""
        _, ns, nm, _, _ = self._info()
        seeds = np.zeros(ns, dtype=[("kmer", "<u4"), ("pos", "<u4"), ("contig", "<u4"), ("canon", "<u4")])
        markers = np.zeros(nm, dtype=np.uint64)
        _capi.check(self._ctx._lib.psk_sketch_export(self._h, seeds.ctypes.data_as(C.c_void_p), markers.ctypes.data_as(C.c_void_p)))
        return seeds, markers


_warned_no_model = False


class Database:
    """A database storing sketched genomes (lib.rs:132-137, 368-660): sketches live in HBM; with a `path`
    they are also written to a folder in the reference's layout (storage.py)."""

    def __init__(self, path=None, *, compression=125, marker_compression=1000, k=15, format=None, device=0):
        self._ctx = default_context(device)
        self._lib = self._ctx._lib
        self._params = _capi.Params(int(compression), int(marker_compression), int(k))
        self._lock = threading.Lock()   # `sketch` takes &mut self (lib.rs:479)
        self._names = []                # insertion order = markers Vec (lib.rs:501-504)
        self._resident = []             # per ref: True when the full sketch (not only its markers) sits in HBM
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

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.psk_db_destroy(self._h)
            self._h = None

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
        if name in self._cache:
            self._cache.move_to_end(name)
            return self._cache[name]
        sk = Sketch.from_record(self._ctx, self._storage.load(name))        # KeyError / OSError as lib.rs:95-122
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
        if kind == "consolidated":
            target = _storage.Consolidated(folder)
            if overwrite and os.path.exists(os.path.join(folder, "sketches.db")):
                os.remove(os.path.join(folder, "sketches.db"))
        elif kind == "separated":
            target = _storage.Folder(folder)
        else:
            raise ValueError(f"invalid format: {kind}")
        markers = []
        for i in range(len(self._names)):
            rec = self._full_sketch(i).to_record()
            target.store(rec)
            markers.append(rec.markers_only())
        target.flush((self._params.c, self._params.marker_c, self._params.k), markers)

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

    def sketch(self, name, *contigs, seed=True):
        """Add a reference genome to the database (lib.rs:477-510)."""
        if not isinstance(name, str):
            raise TypeError("name must be a str")
        if not self._lock.acquire(blocking=False):
            raise RuntimeError("Already borrowed")   # PyO3's &mut self borrow error
        try:
            sk = self._sketch(name, contigs, seed)
            if self._storage is not None:              # lib.rs:505-508: written at once, markers only on flush
                self._storage.store(sk.to_record())
            sk._owned = False                          # ownership moves into the db (lib.rs:501-508)
            _capi.check(self._lib.psk_db_add(self._h, name.encode("utf-8"), sk._h))
            self._names.append(name)
            self._resident.append(True)
        finally:
            self._lock.release()
        return None

    def _opts(self, learned_ani, median, robust, cutoff, faster_small):
        global _warned_no_model
        # default rule: learned ANI when c >= 70 and not median (lib.rs:611-613, docstring :522-527)
        learned = learned_ani if learned_ani is not None else (self._params.c >= 70 and not median)
        if learned and learned_ani is None:
            # the GBDT weights live inside the skani crate and are not available to this build: say so
            if not _warned_no_model:
                warnings.warn("pyskani_amd: no learned-ANI regression model is available; returning the raw "
                              "chain ANI (pass learned_ani=False to silence, learned_ani=True raises)", RuntimeWarning, stacklevel=3)
                _warned_no_model = True
            learned = False
        return _capi.QueryOpts(1 if learned else 0, int(bool(median)), int(bool(robust)), int(bool(faster_small)),
                               float(cutoff) if cutoff else 0.0, 0.0)

    def _hit(self, r, qname):
        ref_name = self._lib.psk_db_name(self._h, r.ref_index).decode("utf-8")
        hit = Hit(r.ani, qname, r.af_query, ref_name, r.af_ref)
        hit._raw = {f: getattr(r, f) for f, _ in _capi.Hit._fields_}
        return hit

    def query_many(self, genomes, *, seed=True, learned_ani=None, median=False, robust=False, cutoff=None,
                   faster_small=False):
        """[(name, contigs...)] -> list of hit lists; equals [self.query(name, *contigs, ...) for ...].
        An addition to the reference API for all-vs-all / many-bin workloads."""
        sketches = [self._sketch(g[0], g[1:], seed) for g in genomes]
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
        if not all(self._resident):      # `open`ed database: sketches come from disk per query
            return [self._query_lazy(s.name, s, opts) for s in sketches]
        arr = (C.c_void_p * max(n, 1))(*[s._h for s in sketches])
        hits_p = C.POINTER(_capi.Hit)()
        offs = (C.c_uint64 * (n + 1))()
        _capi.check(self._lib.psk_query_many(self._h, arr, n, C.byref(opts), C.byref(hits_p), offs))
        try:
            return [[self._hit(hits_p[j], sketches[i].name) for j in range(offs[i], offs[i + 1])] for i in range(n)]
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
        idx = [i for i in range(n) if flags[i]]
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
        q = self._sketch(name, contigs, seed)
        opts = self._opts(learned_ani, median, robust, cutoff, faster_small)
        if not all(self._resident):
            return self._query_lazy(name, q, opts)
        hits_p = C.POINTER(_capi.Hit)()
        n = C.c_uint64(0)
        _capi.check(self._lib.psk_query(self._h, q._h, C.byref(opts), C.byref(hits_p), C.byref(n)))
        out = []
        try:
            for i in range(n.value):
                out.append(self._hit(hits_p[i], name))
        finally:
            if hits_p:
                self._lib.psk_free(hits_p)
        return out
