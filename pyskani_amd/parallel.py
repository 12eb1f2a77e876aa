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


class ShardedDatabase:
    """A `Database` whose references are sharded over the ranks of a torch.distributed process group.

    One process per GPU. `sketch_all` gives rank r the contiguous shard `shard_bounds(n, r, world)` of the
    reference list; `query` runs the ordinary `Database.query` (lib.rs:549-660) on the local shard and
    all-gathers the hit lists, so every rank returns the same hits, in global reference order. The only
    collective on the data path is that gather: pairs are independent (lib.rs:617-657).

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

    def sketch_all(self, names, fetch):
        """Add references `names` (the same list on every rank); `fetch(i)` returns the contigs (a tuple of
        bytes-like) of global reference i and is called only for this rank's shard."""
        if self.names:
            raise RuntimeError("ShardedDatabase.sketch_all may be called once: shards are contiguous")
        self.names = list(names)
        self._lo, hi = shard_bounds(len(self.names), self.rank, self.world)
        for i in range(self._lo, hi):
            self.local.sketch(self.names[i], *fetch(i))
        return hi - self._lo

    def query(self, name, *contigs, **opts):
        from .database import Hit
        index = {n: self._lo + j for j, n in enumerate(self.names[self._lo:self._lo + len(self.local)])}
        local = self.local.query(name, *contigs, **opts)

        def gidx(h):   # the library's own reference index when the hit carries it; else by name
            raw = getattr(h, "_raw", None)
            return self._lo + raw["ref_index"] if raw else index[h.reference_name]
        # column 0 travels as float32: exact below 2^24 references per job
        rows = np.array([[gidx(h), h.identity, h.query_fraction, h.reference_fraction] for h in local],
                        dtype=np.float32).reshape(-1, HIT_COLS)
        allh = all_gather_hits(rows, self.dist, device=self.coll_device, group=self.group)
        return [Hit(float(r[1]), name, float(r[2]), self.names[int(r[0])], float(r[3])) for r in allh]
