"""Multi-GPU plumbing for the query path: one process per GPU, references sharded across ranks,
one exchange step — an all-gather of the per-shard hit lists. The collective runs
through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests);
nothing else on the data path is collective: every (query, ref) pair is independent (lib.rs:617-657).
"""
import numpy as np
import torch  # noqa: F401  (before the HIP library: torch bundles its own HIP runtime, which must initialise first)

HIT_VALS = 3  # ani, af_query, af_ref


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


_HIT_CAP = {}      # rows the gathered buffer held last time, per process group: the next call sends that many (+ the header row)


def all_gather_hits(idx, vals, dist, device="cpu", group=None):
    """All-gather ragged per-shard hit lists.

    idx:  int64 [n_local, 2] = (global query index, global ref index) — integers travel as integers (exact for any
          database size); vals: float32 [n_local, 3] = (ani, af_query, af_ref).
    Returns (idx, vals) concatenated over ranks in rank order, identical on every rank.

    ONE collective in the steady state: every rank sends a fixed number of int64 rows — row 0 carries its true count, the
    others (query, ref, the three floats bit-cast into two int64) — sized by the largest count seen so far. If a count does
    not fit (every rank sees every header, so all agree), the capacity grows and the gather is repeated once; the first
    call therefore exchanges the counts alone and then the lists.
    """
    import torch
    world = dist.get_world_size(group)
    idx = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1, 2)
    vals = np.ascontiguousarray(vals, dtype=np.float32).reshape(-1, HIT_VALS)
    assert len(idx) == len(vals)
    n = idx.shape[0]
    rows = np.zeros((n, 4), dtype=np.int64)
    rows[:, :2] = idx
    v4 = np.zeros((n, 4), dtype=np.float32)
    v4[:, :HIT_VALS] = vals
    rows[:, 2:] = v4.view(np.int64)
    key = id(group) if group is not None else 0
    cap = _HIT_CAP.get(key, 0)
    while True:
        mine = torch.zeros((cap + 1, 4), dtype=torch.int64, device=device)
        mine[0, 0] = n
        k = min(n, cap)
        if k:
            mine[1:1 + k] = torch.from_numpy(rows[:k]).to(device)
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        host = [p.cpu().numpy() for p in parts]
        counts = [int(h[0, 0]) for h in host]
        if max(counts) <= cap:
            break
        cap = max(counts)      # the same on every rank: they all saw the same headers
    _HIT_CAP[key] = cap
    got = np.concatenate([h[1:1 + c] for h, c in zip(host, counts)], axis=0) if sum(counts) else np.zeros((0, 4), np.int64)
    out_idx = np.ascontiguousarray(got[:, :2])
    out_vals = np.ascontiguousarray(got[:, 2:]).view(np.float32).reshape(-1, 4)[:, :HIT_VALS].copy()
    return out_idx, out_vals


def all_gather_sketches(sketches, ctx, dist, device, group=None):
    """The exchange step of an all-vs-all: every rank contributes a list of device-resident sketches
    and receives everybody's, as device-resident sketches on ITS GPU. Records are packed into one uint8 device tensor
    (psk_sketch_pack), moved by ONE all-gather of that tensor (RCCL over xGMI with backend "nccl"; no host hop),
    and unpacked in place (psk_sketch_unpack). A second, small all-gather carries the record sizes.
    Returns a list over ranks of lists of `Sketch`."""
    import torch
    from .database import Sketch
    world = dist.get_world_size(group)
    sizes = [s.pack_size() for s in sketches]
    n = torch.tensor([len(sizes)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    counts = [int(x.item()) for x in ns]
    m = max(max(counts), 1)
    mine = torch.zeros(m, dtype=torch.int64, device=device)
    if sizes:
        mine[:len(sizes)] = torch.tensor(sizes, dtype=torch.int64)
    all_sizes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(all_sizes, mine, group=group)
    all_sizes = [t.cpu().numpy()[:c] for t, c in zip(all_sizes, counts)]
    offs = [np.concatenate([[0], np.cumsum(sz)]).astype(np.int64) for sz in all_sizes]     # record sizes are multiples of 16
    width = max(int(max(o[-1] for o in offs)), 16)
    buf = torch.empty(width, dtype=torch.uint8, device=device)      # pad bytes are never read; a fill kernel on torch's stream
    torch.cuda.current_stream(device).synchronize()                   # would race the library's own (non-blocking) stream
    rank = dist.get_rank(group)
    for s, o, sz in zip(sketches, offs[rank][:-1], sizes):
        s.pack_into(buf.data_ptr() + int(o), int(sz))
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    torch.cuda.synchronize(device)              # the library reads the gathered tensors on its own stream
    out = []
    for r in range(world):
        names = [f"rank{r}_{j}" for j in range(counts[r])]
        out.append(Sketch.unpack(ctx, parts[r].data_ptr(), offs[r][:-1], names) if counts[r] else [])
    return out


class ShardedDatabase:
    """A `Database` whose references are sharded over the ranks of a torch.distributed process group.

    One process per GPU. `sketch_all` gives rank r the contiguous shard `shard_bounds(n, r, world)` of the
    reference list; `query` runs the ordinary `Database.query` (lib.rs:549-660) on the local shard and
    all-gathers the hit lists, so every rank returns the same hits, in global reference order. The only
    collective on that path is the gather: pairs are independent (lib.rs:617-657). `all_vs_all` adds the one
    other exchange the path has: the shards' sketches, all-gathered in batches as the query side.

    `local` is any object with the `Database` interface; by default a `pyskani_amd.Database` on this rank's GPU.
    """

    def __init__(self, dist, local=None, device=None, group=None, **params):
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.coll_device = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
        if local is None:
            from .database import Database
            local = Database(device=(device.index if device is not None and device.index is not None else 0), **params)
        self.local = local
        self.names = []          # GLOBAL reference names, identical on every rank
        self._lo = 0
        self._cuts = None        # shard cut points (world + 1), identical on every rank
        self.device = device

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
        self.names = list(names)
        if weights is not None:
            if len(weights) != len(self.names):
                raise ValueError("weights must hold one entry per reference")
            self._cuts = weighted_shard_cuts(weights, self.world)
        self._lo, hi = self._shard(self.rank)
        for i in range(self._lo, hi):
            self.local.sketch(self.names[i], *fetch(i))
        return hi - self._lo

    def _global_index(self, h, by_name):
        raw = getattr(h, "_raw", None)        # the library's own reference index when the hit carries it; else by name
        return self._lo + int(raw["ref_index"]) if raw is not None else by_name[h.reference_name]

    def query(self, name, *contigs, **opts):
        from .database import Hit
        by_name = {n: self._lo + j for j, n in enumerate(self.names[self._lo:self._lo + len(self.local)])}
        local = self.local.query(name, *contigs, **opts)
        idx = np.array([[0, self._global_index(h, by_name)] for h in local], dtype=np.int64).reshape(-1, 2)
        vals = np.array([[h.identity, h.query_fraction, h.reference_fraction] for h in local], dtype=np.float32).reshape(-1, HIT_VALS)
        idx, vals = all_gather_hits(idx, vals, self.dist, device=self.coll_device, group=self.group)
        return [Hit(float(v[0]), name, float(v[1]), self.names[int(i[1])], float(v[2])) for i, v in zip(idx, vals)]

    def all_vs_all(self, batch=256, **opts):
        """Every genome of the job against every other (and itself). Each rank's shard IS its share of the genomes,
        so the query side is the all-gather of the shards' sketches, `batch` genomes per rank at a time, as packed
        device records (`all_gather_sketches`: HBM -> xGMI -> HBM); each received batch is queried against the local
        shard with `Database.query_sketches`, and the hit lists are all-gathered at the end. Returns
        {query_name: [Hit, ...]}, identical on every rank, hits in global reference order."""
        import torch
        from .database import Hit
        local = self.local
        n_local = len(local)
        by_name = {n: self._lo + j for j, n in enumerate(self.names[self._lo:self._lo + n_local])}
        sizes = [self._shard(r)[1] - self._shard(r)[0] for r in range(self.world)]
        dev = self.device if self.device is not None else torch.device("cuda", local._device)
        rows_i, rows_v = [], []
        for b in range((max(sizes) + batch - 1) // batch if sizes else 0):
            i0, i1 = min(b * batch, n_local), min((b + 1) * batch, n_local)
            mine = [local._full_sketch(i) for i in range(i0, i1)]
            for r, sketches in enumerate(all_gather_sketches(mine, local._ctx, self.dist, dev, group=self.group)):
                qbase = self._shard(r)[0] + b * batch
                for j, hits in enumerate(local.query_sketches(sketches, **opts) if sketches else []):
                    for h in hits:
                        rows_i.append([qbase + j, self._global_index(h, by_name)])
                        rows_v.append([h.identity, h.query_fraction, h.reference_fraction])
        idx, vals = all_gather_hits(np.array(rows_i, dtype=np.int64).reshape(-1, 2), np.array(rows_v, dtype=np.float32).reshape(-1, HIT_VALS),
                                    self.dist, device=self.coll_device, group=self.group)
        out = {n: [] for n in self.names}
        for t in np.lexsort((idx[:, 1], idx[:, 0])):
            qn = self.names[int(idx[t, 0])]
            out[qn].append(Hit(float(vals[t, 0]), qn, float(vals[t, 1]), self.names[int(idx[t, 1])], float(vals[t, 2])))
        return out
