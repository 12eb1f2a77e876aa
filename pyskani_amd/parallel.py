"""Multi-GPU plumbing for the query path: one process per GPU, references sharded across ranks,
one exchange step — an all-gather of the per-shard hit lists. The collective runs
through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests);
nothing else on the data path is collective: every (query, ref) pair is independent (lib.rs:617-657).
"""
import numpy as np

HIT_COLS = 4  # global ref index, ani, af_query, af_ref


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` of `world`."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_hits(local_hits, dist, device="cpu", group=None):
    """All-gather ragged per-shard hit lists.

    local_hits: float32 array [n_local, HIT_COLS] whose column 0 already holds GLOBAL ref indices.
    Returns the concatenation over ranks (rank order), identical on every rank. Two collectives:
    an all-gather of the counts, then one all-gather of lists padded to the largest count.
    """
    import torch
    world = dist.get_world_size(group)
    local = np.ascontiguousarray(local_hits, dtype=np.float32).reshape(-1, HIT_COLS)
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    m = max(max(counts), 1)
    mine = torch.zeros((m, HIT_COLS), dtype=torch.float32, device=device)
    if local.shape[0]:
        mine[:local.shape[0]] = torch.from_numpy(local).to(device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = [p[:c].cpu().numpy() for p, c in zip(parts, counts)]
    return np.concatenate(out, axis=0) if out else np.zeros((0, HIT_COLS), np.float32)


def all_gather_bytes(blobs, dist, device="cpu", group=None):
    """All-gather a list of byte strings per rank: returns, on every rank, the lists of all ranks (rank order).
    Two collectives: lengths (padded to the longest list), then one padded uint8 all-gather."""
    import torch
    world = dist.get_world_size(group)
    n = torch.tensor([len(blobs)], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    counts = [int(x.item()) for x in ns]
    m = max(max(counts), 1)
    lens = torch.zeros(m, dtype=torch.int64, device=device)
    if blobs:
        lens[:len(blobs)] = torch.tensor([len(b) for b in blobs], dtype=torch.int64, device=device)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    all_lens = [l.cpu().numpy() for l in all_lens]
    width = max(int(max(int(l.sum()) for l in all_lens)), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=device)
    if blobs:
        flat = np.frombuffer(b"".join(blobs), dtype=np.uint8)
        mine[:len(flat)] = torch.from_numpy(flat.copy()).to(device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    out = []
    for r in range(world):
        buf = parts[r].cpu().numpy().tobytes()
        pos, lst = 0, []
        for k in range(counts[r]):
            ln = int(all_lens[r][k]); lst.append(buf[pos:pos + ln]); pos += ln
        out.append(lst)
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

    def __len__(self):
        return len(self.names)

    def _shard(self, r):
        return shard_bounds(len(self.names), r, self.world)

    def sketch_all(self, names, fetch):
        """Add references `names` (the same list on every rank); `fetch(i)` returns the contigs (a tuple of
        bytes-like) of global reference i and is called only for this rank's shard."""
        if self.names:
            raise RuntimeError("ShardedDatabase.sketch_all may be called once: shards are contiguous")
        self.names = list(names)
        self._lo, hi = self._shard(self.rank)
        for i in range(self._lo, hi):
            self.local.sketch(self.names[i], *fetch(i))
        return hi - self._lo

    def _global_index(self, h, by_name):
        raw = getattr(h, "_raw", None)        # the library's own reference index when the hit carries it; else by name
        return self._lo + raw["ref_index"] if raw else by_name[h.reference_name]

    def query(self, name, *contigs, **opts):
        from .database import Hit
        by_name = {n: self._lo + j for j, n in enumerate(self.names[self._lo:self._lo + len(self.local)])}
        local = self.local.query(name, *contigs, **opts)
        # column 0 travels as float32: exact below 2^24 references per job
        rows = np.array([[self._global_index(h, by_name), h.identity, h.query_fraction, h.reference_fraction] for h in local],
                        dtype=np.float32).reshape(-1, HIT_COLS)
        allh = all_gather_hits(rows, self.dist, device=self.coll_device, group=self.group)
        return [Hit(float(r[1]), name, float(r[2]), self.names[int(r[0])], float(r[3])) for r in allh]

    def all_vs_all(self, batch=256, **opts):
        """Every genome of the job against every other (and itself). Each rank's shard IS its share of the genomes,
        so the query side is the all-gather of the shards' sketches, `batch` genomes per rank at a time: records are
        exported from HBM, exchanged as bytes, imported on the receiving GPU, queried against the local shard with
        `Database.query_sketches`, and the hit lists all-gathered. Returns {query_name: [Hit, ...]}, identical on
        every rank, hits in global reference order."""
        from .database import Hit, Sketch
        from .storage import Record
        local = self.local
        n_local = len(local)
        by_name = {n: self._lo + j for j, n in enumerate(self.names[self._lo:self._lo + n_local])}
        sizes = [self._shard(r)[1] - self._shard(r)[0] for r in range(self.world)]
        rows_q, rows_h = [], []
        for b in range((max(sizes) + batch - 1) // batch if sizes else 0):
            i0, i1 = min(b * batch, n_local), min((b + 1) * batch, n_local)
            blobs = [local._full_sketch(i).to_record().to_bytes() for i in range(i0, i1)]
            for r, lst in enumerate(all_gather_bytes(blobs, self.dist, device=self.coll_device, group=self.group)):
                qbase = self._shard(r)[0] + b * batch
                sketches = [Sketch.from_record(local._ctx, Record.from_bytes(x)) for x in lst]
                for j, hits in enumerate(local.query_sketches(sketches, **opts) if sketches else []):
                    for h in hits:
                        rows_q.append([qbase + j, 0, 0, 0])
                        rows_h.append([self._global_index(h, by_name), h.identity, h.query_fraction, h.reference_fraction])
        a = all_gather_hits(np.array(rows_h, dtype=np.float32).reshape(-1, HIT_COLS), self.dist, device=self.coll_device, group=self.group)
        q = all_gather_hits(np.array(rows_q, dtype=np.float32).reshape(-1, HIT_COLS), self.dist, device=self.coll_device, group=self.group)
        out = {n: [] for n in self.names}
        for t in np.lexsort((a[:, 0], q[:, 0])):
            qn = self.names[int(q[t, 0])]
            out[qn].append(Hit(float(a[t, 1]), qn, float(a[t, 2]), self.names[int(a[t, 0])], float(a[t, 3])))
        return out
