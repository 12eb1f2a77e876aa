"""Multi-GPU plumbing for the query path: one process per GPU, references sharded across ranks.

Every (query, ref) pair is independent (lib.rs:617-657), so the path has exactly two exchange steps and nothing else is
collective: the all-gather of the per-shard HIT LISTS (search and all-vs-all), and — all-vs-all only — the all-gather of the
shards' SKETCHES as the query side, moved as packed device records HBM -> xGMI -> HBM.

Two interchangeable transports carry them:
  TorchComm   torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests and one-GPU dry runs)
  CapiComm    the library's own RCCL communicator behind the C-ABI (psk_comm_create / psk_gather_hits / psk_gather_sketches,
              include/pyskani_amd.h): what a non-Python host binds; torch.distributed (any backend) only hands the 128-byte id round
Hits travel as the library's 20-byte psk_hit_min records (numpy structured arrays; SURVEY.md §8e's record): the GLOBAL reference index
in `ref_index`, the global query index in `query` (bit 31: the regression model produced the ANI). `raw=True` moves the 80-byte psk_hit
records with every chaining integer instead (global query index in `reserved`): parity tests. No Python object exists per hit until a
caller asks for `Hit`s at the very end.
"""
import ctypes as C
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch  # noqa: F401  (before the HIP library: torch bundles its own HIP runtime, which must initialise first)

from . import _capi

HIT_VALS = 3  # ani, af_query, af_ref
HIT_DTYPE = np.dtype(_capi.Hit)
HIT_MIN_DTYPE = np.dtype(_capi.HitMin)
HIT_BYTES = HIT_DTYPE.itemsize
QUERY_MASK = 0x7FFFFFFF      # psk_hit_min.query: the index below bit 31, `learned` in it


def _qfield(dtype):
    """name of the field that carries the query index in records of `dtype`"""
    return "query" if "query" in dtype.names else "reserved"


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` of `world`."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def weighted_shard_cuts(weights, world):
    """Contiguous shards balanced on `weights` (genome lengths ~ seed counts): cut point r is the first
    index whose prefix weight reaches r/world of the total. Returns world+1 cut indices; every rank computes the same."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    if n == 0 or w.sum() <= 0:
        return [shard_bounds(n, r, world)[0] for r in range(world)] + [n]
    pre = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world):
        c = int(np.searchsorted(pre, pre[-1] * r / world, side="left"))
        # the genome straddling the target goes to whichever side leaves the smaller imbalance
        if c > 0 and abs(pre[c - 1] - pre[-1] * r / world) <= abs(pre[min(c, n)] - pre[-1] * r / world):
            c -= 1
        cuts.append(min(max(c, cuts[-1]), n))
    return cuts + [n]


def _al16(x):
    return (int(x) + 15) & ~15


def all_gather_hit_records(recs, dist, device="cpu", group=None, state=None):
    """All-gather ragged per-shard arrays of psk_hit records. Returns (all records in rank order, per-rank counts), identical on
    every rank.

    ONE collective in the steady state: every rank sends a fixed number of rows of one record each (20 or 80 bytes) — row 0 carries its true count — sized by
    `state["cap"]`, the largest count seen lately. If a count does not fit (every rank sees every header, so all agree), the
    capacity grows and the gather is repeated once. The capacity decays (halves towards the latest maximum), so one large exchange
    does not tax every later small one. `state` is a dict the CALLER owns (one per ShardedDatabase); without it every call
    exchanges the counts first."""
    world = dist.get_world_size(group)
    recs = np.asarray(recs)
    rec_dtype = recs.dtype if recs.dtype.names else HIT_DTYPE      # psk_hit or psk_hit_min records: rows of their own width
    recs = np.ascontiguousarray(recs, dtype=rec_dtype).reshape(-1)
    row_bytes = rec_dtype.itemsize
    n = recs.shape[0]
    cap = int(state.get("cap", 0)) if state is not None else 0
    raw = recs.view(np.uint8).reshape(n, row_bytes)
    while True:
        mine = torch.empty((cap + 1, row_bytes), dtype=torch.uint8, device=device)
        head = np.zeros(row_bytes, np.uint8)
        head[:8] = np.frombuffer(np.int64(n).tobytes(), np.uint8)
        k = min(n, cap)
        host = np.concatenate([head[None, :], raw[:k]], axis=0) if k else head[None, :]
        mine[:1 + k] = torch.from_numpy(host).to(device)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        heads = torch.stack([p[0, :8] for p in parts]).cpu().numpy()
        counts = [int(np.frombuffer(heads[r].tobytes(), np.int64)[0]) for r in range(world)]
        if max(counts) <= cap:
            break
        cap = max(counts)      # the same on every rank: they all saw the same headers
    if state is not None:
        state["cap"] = max(max(counts), cap // 2)
    if sum(counts):
        got = np.concatenate([p[1:1 + c].cpu().numpy() for p, c in zip(parts, counts) if c], axis=0)
        out = np.ascontiguousarray(got).view(rec_dtype).reshape(-1)
    else:
        out = np.zeros(0, rec_dtype)
    return out, counts


def all_gather_hits(idx, vals, dist, device="cpu", group=None, state=None):
    """All-gather ragged per-shard hit lists given as plain arrays.

    idx:  int64 [n_local, 2] = (global query index, global ref index), both below 2^32; vals: float32 [n_local, 3] =
    (ani, af_query, af_ref). Returns (idx, vals) concatenated over ranks in rank order, identical on every rank. A thin front
    of all_gather_hit_records: the indices travel in the records' 32-bit fields - exact below 2^32, OverflowError beyond (they are
    never truncated silently)."""
    idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1, 2)
    vals = np.ascontiguousarray(vals, dtype=np.float32).reshape(-1, HIT_VALS)
    assert len(idx) == len(vals)
    if len(idx) and (int(idx.min()) < 0 or int(idx.max()) >= 1 << 32):
        raise OverflowError("hit indices must lie in [0, 2^32): the exchange records carry them in 32-bit fields")
    recs = np.zeros(len(idx), HIT_DTYPE)
    recs["reserved"] = idx[:, 0]; recs["ref_index"] = idx[:, 1]
    recs["ani"] = vals[:, 0]; recs["af_query"] = vals[:, 1]; recs["af_ref"] = vals[:, 2]
    got, _ = all_gather_hit_records(recs, dist, device=device, group=group, state=state)
    out_idx = np.stack([got["reserved"].astype(np.int64), got["ref_index"].astype(np.int64)], axis=1).reshape(-1, 2)
    out_vals = np.stack([got["ani"], got["af_query"], got["af_ref"]], axis=1).astype(np.float32).reshape(-1, HIT_VALS)
    return out_idx, out_vals


class TorchComm:
    """The two exchange steps over torch.distributed."""

    kind = "torch"

    def __init__(self, dist, group=None, device="cpu"):
        self.dist, self.group, self.device = dist, group, device
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._hit_state = {}
        self.bytes_sent = 0

    def gather_hit_records(self, recs):
        state = self._hit_state.setdefault(np.asarray(recs).dtype.itemsize, {})      # (one capacity per record width)
        out, counts = all_gather_hit_records(recs, self.dist, device=self.device, group=self.group, state=state)
        self.bytes_sent += (state.get("cap", 0) + 1) * out.dtype.itemsize * (self.world - 1)
        return out, counts

    def gather_sketch_handles(self, ctx, handles, cuda_device):
        """handles: ctypes array of this rank's psk_sketch* -> (ctypes array of everybody's handles on this GPU, counts per rank).
        TWO collectives: (count, bytes) of every rank, then one uint8 device tensor per rank = [u64 sizes[n]] pad16 [records],
        packed by ONE psk_sketch_pack_many, moved by one all-gather (HBM -> xGMI -> HBM with backend "nccl"), unpacked by ONE
        psk_sketch_unpack."""
        lib, dist, world = ctx._lib, self.dist, self.world
        n = len(handles)
        sizes = np.zeros(n, np.uint64)
        sz = C.c_uint64()
        for i in range(n):
            _capi.check(lib.psk_sketch_pack_size(handles[i], C.byref(sz))); sizes[i] = sz.value
        table = _al16(8 * n)
        offs = table + np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        mybytes = int(offs[-1])
        nb = torch.tensor([n, mybytes], dtype=torch.int64, device=self.device)
        nbs = [torch.empty_like(nb) for _ in range(world)]
        dist.all_gather(nbs, nb, group=self.group)
        nbs = torch.stack(nbs).cpu().numpy()
        counts = [int(x) for x in nbs[:, 0]]
        width = max(16, _al16(int(nbs[:, 1].max())))
        big = torch.empty((world + 1, width), dtype=torch.uint8, device=cuda_device)      # pad bytes are never read
        send = big[world]
        if n:
            send[:8 * n] = torch.from_numpy(sizes.view(np.uint8)).to(cuda_device)
        torch.cuda.current_stream(cuda_device).synchronize()      # the library packs on its own (non-blocking) stream
        if n:
            c_offs = (C.c_uint64 * n)(*[int(o) for o in offs[:-1]])
            _capi.check(lib.psk_sketch_pack_many(handles, n, C.c_void_p(send.data_ptr()), c_offs, mybytes))
        if self.device == "cpu":      # gloo: the records take the host hop (CPU tests, one-GPU dry runs)
            hsend = send.cpu()
            hparts = [torch.empty_like(hsend) for _ in range(world)]
            dist.all_gather(hparts, hsend, group=self.group)
            for r in range(world):
                big[r].copy_(hparts[r])
        else:
            dist.all_gather(list(big[:world].unbind(0)), send, group=self.group)
        self.bytes_sent += width * (world - 1)
        maxn = max(counts) if counts else 0
        tables = big[:world, :8 * maxn].cpu().numpy() if maxn else np.zeros((world, 0), np.uint8)
        torch.cuda.current_stream(cuda_device).synchronize()      # the library reads the gathered tensor on its own stream (not a device-wide wait: with overlap the previous round is being queried)
        roffs = []
        for r in range(world):
            sz_r = np.ascontiguousarray(tables[r, :8 * counts[r]]).view(np.uint64)
            o = r * width + _al16(8 * counts[r])
            for s in sz_r:
                roffs.append(o); o += int(s)
        total = len(roffs)
        out = (C.c_void_p * max(total, 1))()
        if total:
            c_roffs = (C.c_uint64 * total)(*roffs)
            _capi.check(lib.psk_sketch_unpack(ctx._h, C.c_void_p(big.data_ptr()), int(big.numel()), c_roffs, total, out))      # (a record that points beyond the gathered buffer is refused)
        return out, counts


class CapiComm:
    """The two exchange steps through the library's own RCCL communicator (psk_comm_*): the id is made by rank 0 and handed
    round with torch.distributed's broadcast_object_list (any backend — it is the host program's bootstrap channel, not the
    data path)."""

    kind = "capi"

    def __init__(self, ctx, dist, group=None):
        lib = ctx._lib
        self._lib, self.ctx = lib, ctx
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = (C.c_uint8 * _capi.COMM_ID_BYTES)()
        if self.rank == 0:
            _capi.check(lib.psk_comm_unique_id(ident))
        box = [bytes(ident)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        ident = (C.c_uint8 * _capi.COMM_ID_BYTES).from_buffer_copy(box[0])
        h = C.c_void_p()
        _capi.check(lib.psk_comm_create(ctx._h, self.rank, self.world, ident, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.psk_comm_destroy(self._h)
            self._h = None

    @property
    def bytes_sent(self):
        b = C.c_uint64()
        _capi.check(self._lib.psk_comm_info(self._h, None, None, C.byref(b), None))
        return b.value

    def gather_hit_records(self, recs):
        recs = np.asarray(recs)
        small = recs.dtype == HIT_MIN_DTYPE
        dtype = HIT_MIN_DTYPE if small else HIT_DTYPE
        recs = np.ascontiguousarray(recs, dtype=dtype).reshape(-1)
        out_p = C.POINTER(_capi.HitMin if small else _capi.Hit)()
        n_all = C.c_uint64()
        counts = (C.c_uint64 * self.world)()
        fn = self._lib.psk_gather_hits_min if small else self._lib.psk_gather_hits
        _capi.check(fn(self._h, recs.ctypes.data_as(C.c_void_p), len(recs), C.byref(out_p), C.byref(n_all), counts))
        try:
            n = n_all.value
            out = _capi.hit_records(out_p, 0, n, dtype)
        finally:
            if out_p:
                self._lib.psk_free(out_p)
        return out, [int(c) for c in counts]

    def gather_sketch_handles(self, ctx, handles, cuda_device=None):
        n = len(handles)
        out_pp = C.POINTER(C.c_void_p)()
        counts = (C.c_uint32 * self.world)()
        _capi.check(self._lib.psk_gather_sketches(self._h, handles, n, C.byref(out_pp), counts))
        total = sum(counts)
        out = (C.c_void_p * max(total, 1))()
        for i in range(total):
            out[i] = out_pp[i]
        self._lib.psk_free(out_pp)
        return out, [int(c) for c in counts]


class ShardedDatabase:
    """A `Database` whose references are sharded over the ranks of a torch.distributed process group.

    One process per GPU. `sketch_all` gives rank r the contiguous shard `shard_bounds(n, r, world)` of the
    reference list; `query` runs the ordinary `Database.query` (lib.rs:549-660) on the local shard and
    all-gathers the hit lists, so every rank returns the same hits, in global reference order. The only
    collective on that path is the gather: pairs are independent (lib.rs:617-657). `all_vs_all` adds the one
    other exchange the path has: the shards' sketches, all-gathered in batches as the query side.

    `local` is any object with the `Database` interface; by default a `pyskani_amd.Database` on this rank's GPU.
    `comm`: "torch" (collectives through torch.distributed), "capi" (the library's RCCL communicator, psk_comm_*), or an instance.
    `stats` accumulates where the wall time of the exchange-carrying calls went: seconds inside psk_* calls, inside collectives
    (with `overlap` the sketch gather of round b + 1 runs beside round b's query: the two sums then overlap), and in Python around them.
    `raw`: the hit lists travel as 80-byte psk_hit records with every chaining integer (parity tests) instead of the 20-byte psk_hit_min.
    """

    def __init__(self, dist, local=None, device=None, group=None, comm="torch", raw=False, **params):
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.coll_device = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
        if local is None:
            from .database import Database
            local = Database(device=(device.index if device is not None and device.index is not None else 0), **params)
        self.local = local
        if comm == "capi":
            self.comm = CapiComm(local._ctx, dist, group)
        elif comm == "torch":
            self.comm = TorchComm(dist, group, self.coll_device)
        elif hasattr(comm, "gather_hit_records"):
            self.comm = comm           # a transport the caller made once and reuses (creating an RCCL communicator is not free)
        else:
            raise ValueError("comm must be 'torch', 'capi' or a TorchComm / CapiComm instance")
        self.names = []          # GLOBAL reference names, identical on every rank
        self._lo = 0
        self._cuts = None        # shard cut points (world + 1), identical on every rank
        self.device = device
        self.stats = {"psk_s": 0.0, "collective_s": 0.0, "exposed_collective_s": 0.0, "total_s": 0.0}      # (exposed: what the calling thread WAITED for; with overlap the rest of collective_s ran beside psk_s)
        self.raw = bool(raw)
        self.rec_dtype = HIT_DTYPE if raw else HIT_MIN_DTYPE
        self.trace = None        # a list: all_vs_all_records appends (event, round) as it goes (tests)

    def __len__(self):
        return len(self.names)

    def _shard(self, r):
        if self._cuts is not None:
            return self._cuts[r], self._cuts[r + 1]
        return shard_bounds(len(self.names), r, self.world)

    def sketch_all(self, names, fetch, weights=None):
        """Add references `names` (the same list on every rank); `fetch(i)` returns the contigs (a tuple of
        bytes-like) of global reference i and is called only for this rank's shard. `weights` (one number per
        reference, the same on every rank — genome length is the natural choice: seeds ~ length / c) balances the
        contiguous shards on seed count instead of genome count."""
        if self.names:
            raise RuntimeError("ShardedDatabase.sketch_all may be called once: shards are contiguous")
        self._set_names(names, weights)
        hi = self._shard(self.rank)[1]
        for i in range(self._lo, hi):
            self.local.sketch(self.names[i], *fetch(i))
        return hi - self._lo

    def _set_names(self, names, weights=None):
        self.names = list(names)
        if weights is not None:
            if len(weights) != len(self.names):
                raise ValueError("weights must hold one entry per reference")
            self._cuts = weighted_shard_cuts(weights, self.world)
        self._lo = self._shard(self.rank)[0]

    def adopt_local(self, names, weights=None):
        """For callers that filled `local` themselves (e.g. from device-resident genomes, psk_sketch_batch_device): declare the
        GLOBAL name list; this rank's shard must already hold exactly its share, in order."""
        self._set_names(names, weights)
        lo, hi = self._shard(self.rank)
        if len(self.local) != hi - lo:
            raise ValueError(f"rank {self.rank}: the local database holds {len(self.local)} references, its shard is {hi - lo}")

    def _timed(self, key, fn, *a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            self.stats[key] += time.perf_counter() - t0

    def query(self, name, *contigs, **opts):
        from .database import Hit
        t0 = time.perf_counter()
        dt, qf = self.rec_dtype, _qfield(self.rec_dtype)
        failure = None
        recs = np.zeros(0, dt)
        try:
            if hasattr(self.local, "query_records"):
                got_local = self._timed("psk_s", self.local.query_records, name, *contigs, **opts)
                recs = np.zeros(len(got_local), dt)
                for f in dt.names:
                    if f in got_local.dtype.names:
                        recs[f] = got_local[f]
                if qf == "query" and "learned" in got_local.dtype.names:
                    recs["query"] = np.where(got_local["learned"] != 0, np.uint32(0x80000000), np.uint32(0))
            else:       # a stand-in local database (tests): Hit objects in, records out
                by_name = {n: j for j, n in enumerate(self.names[self._lo:self._lo + len(self.local)])}
                local = self.local.query(name, *contigs, **opts)
                recs = np.zeros(len(local), dt)
                for i, h in enumerate(local):
                    raw = getattr(h, "_raw", None)
                    recs[i]["ref_index"] = int(raw["ref_index"]) if raw is not None else by_name[h.reference_name]
                    recs[i]["ani"], recs[i]["af_query"], recs[i]["af_ref"] = h.identity, h.query_fraction, h.reference_fraction
            recs["ref_index"] += self._lo
            if qf == "reserved":
                recs["reserved"] = 0
        except Exception as e:      # a rank whose local query failed still enters the collective (its peers are waiting in it) and raises afterwards
            failure = e
            recs = np.zeros(0, dt)
        t_wait = time.perf_counter()
        got, _ = self._timed("collective_s", self.comm.gather_hit_records, recs)
        self.stats["exposed_collective_s"] += time.perf_counter() - t_wait
        self.stats["total_s"] += time.perf_counter() - t0
        if failure is not None:
            raise failure
        learned = ((got["query"] >> 31) != 0) if qf == "query" else (got["learned"] != 0)
        base = Hit.__mro__[1]      # (the storage class takes `learned`; Hit's own constructor is the reference's five arguments, hit.rs:26-33)
        return [base.__new__(Hit, float(r["ani"]), name, float(r["af_query"]), self.names[int(r["ref_index"])], float(r["af_ref"]), bool(l)) for r, l in zip(got, learned.tolist())]

    def all_vs_all_records(self, batch=1024, overlap=True, **opts):
        """Every genome of the job against every other (and itself), as RECORDS: a numpy array of psk_hit_min (psk_hit with `raw`)
        sorted by (query, reference) with GLOBAL indices (`query` - `reserved` in raw records - and `ref_index`), identical on every rank.

        Each rank's shard IS its share of the genomes, so the query side is the all-gather of the shards' sketches, `batch`
        genomes per rank at a time, as packed device records (HBM -> xGMI -> HBM); each received batch is queried against
        the local shard in one psk_query_many(_min), the hits get their global indices with numpy, and ONE all-gather of the hit
        records ends the call. No Python object per hit or per pair anywhere. With `overlap` the gather of round b + 1 is
        issued (on a helper thread: its collectives and the library's packing run on their own streams) BEFORE round b is queried,
        so the exchange leaves the critical path: two rounds' gathered sketches are alive at a time. `batch` = 1 024: a round of ~5 Mb genomes (1 024 x its ~100 relatives x 40 000
        seeds) is then above the 2^31 (pair, seed) items from which the library keeps two batches in flight on two lanes; at 256 a rank ran every round as one chain (single-GPU
        emulation of a rank's share of the 10 000-genome job, bench.py extras.scaling_model: 70 against 77 ms per rank at N = 8). 2 x 1 024 x N packed sketches of 0.9 MB are alive: 14 GB at N = 8."""
        t_all = time.perf_counter()
        local = self.local
        n_local = len(local)
        sizes = [self._shard(r)[1] - self._shard(r)[0] for r in range(self.world)]
        dev = self.device if self.device is not None else torch.device("cuda", local._device)
        lib = local._lib
        mine_all = local.sketch_handles()            # ctypes array of the shard's psk_sketch*
        chunks = []
        rounds = (max(sizes) + batch - 1) // batch if sizes else 0
        dt, qf = self.rec_dtype, _qfield(self.rec_dtype)
        trace = self.trace
        if len(self.names) >= 1 << 31:      # (the records' index fields are 32-bit, psk_hit_min's query index 31: refuse rather than wrap)
            raise OverflowError("a sharded job of 2^31 genomes or more does not fit the hit records' index fields")

        def gather(b):
            if trace is not None:
                trace.append(("gather_start", b))
            i0, i1 = min(b * batch, n_local), min((b + 1) * batch, n_local)
            mine_n = (C.c_void_p * (i1 - i0))(*[mine_all[i] for i in range(i0, i1)])
            out = self._timed("collective_s", self.comm.gather_sketch_handles, local._ctx, mine_n, dev)
            if trace is not None:
                trace.append(("gather_end", b))
            return out
        pool = ThreadPoolExecutor(max_workers=1) if (overlap and rounds > 1) else None
        nxt = pool.submit(gather, 0) if pool else None
        nxt_taken = False      # the future in `nxt` has been waited for (its sketches are in `held`)
        held = []      # gathered handle sets not yet freed: (handles, total)
        failure = None
        try:
            for b in range(rounds):
                t_wait = time.perf_counter()
                handles, counts = nxt.result() if pool else gather(b)
                self.stats["exposed_collective_s"] += time.perf_counter() - t_wait
                nxt_taken = True
                total = sum(counts)
                held.append((handles, total))
                if pool and b + 1 < rounds:
                    nxt = pool.submit(gather, b + 1)      # round b + 1's sketches travel while round b is queried
                    nxt_taken = False
                if total and failure is None:
                    if trace is not None:
                        trace.append(("query_start", b))
                    try:
                        recs, offs = self._timed("psk_s", local.query_handles, handles, total, raw=self.raw, **opts)
                    except Exception as e:      # a rank whose local query failed keeps entering the remaining collectives (its peers are waiting in them) with nothing to add, and raises at the end
                        failure = e
                        lib.psk_sketch_free_many(handles, total)
                        held.pop()
                        continue
                    if trace is not None:
                        trace.append(("query_end", b))
                    # global query index of every hit: the batch's queries are rank-major, rank r contributes counts[r]
                    qglob = np.concatenate([self._shard(r)[0] + b * batch + np.arange(counts[r], dtype=np.int64) for r in range(self.world)])
                    if qf == "query":      # the library numbered the hits' queries within the call; bit 31 (learned) rides along
                        q = recs["query"]
                        recs["query"] = (qglob[q & QUERY_MASK].astype(np.uint32) | (q & np.uint32(0x80000000)))
                    else:
                        recs["reserved"] = np.repeat(qglob, np.diff(offs)).astype(np.uint32)
                    recs["ref_index"] += self._lo
                    chunks.append(recs)
                lib.psk_sketch_free_many(handles, total)
                held.pop()
        finally:
            if pool:
                pool.shutdown(wait=True)
                if not nxt_taken and nxt is not None and nxt.exception() is None:      # an error between two rounds: the gather in flight has its sketches too
                    h, c = nxt.result()
                    held.append((h, sum(c)))
            for h, t in held:
                lib.psk_sketch_free_many(h, t)
        mine_recs = np.concatenate(chunks) if (chunks and failure is None) else np.zeros(0, dt)
        t_wait = time.perf_counter()
        got, _ = self._timed("collective_s", self.comm.gather_hit_records, mine_recs)
        self.stats["exposed_collective_s"] += time.perf_counter() - t_wait
        if failure is not None:
            self.stats["total_s"] += time.perf_counter() - t_all
            raise failure
        qkey = (got["query"] & QUERY_MASK) if qf == "query" else got["reserved"]
        order = np.lexsort((got["ref_index"], qkey))
        out = got[order]
        self.stats["total_s"] += time.perf_counter() - t_all
        return out

    def all_vs_all(self, batch=1024, **opts):
        """`all_vs_all_records` as {query_name: [Hit, ...]}, identical on every rank, hits in global reference order."""
        recs = self.all_vs_all_records(batch=batch, **opts)
        from .database import Hit
        out = {n: [] for n in self.names}
        names = self.names
        qidx = (recs["query"] & QUERY_MASK) if "query" in recs.dtype.names else recs["reserved"]
        for q, r, ani, afq, afr in zip(qidx.tolist(), recs["ref_index"].tolist(), recs["ani"].tolist(), recs["af_query"].tolist(), recs["af_ref"].tolist()):
            qn = names[q]
            out[qn].append(Hit(ani, qn, afq, names[r], afr))
        return out
