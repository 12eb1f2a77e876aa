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
