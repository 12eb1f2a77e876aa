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
# every third global ref "hits"; ANI encodes the global index so the gather can be checked. Indices travel as int64:
# 2**24 + 1 is not representable in float32 (ADVICE r1) and must survive the gather
BIG = 2 ** 24 + 1
gidx = np.array([g for g in range(lo, hi) if g %% 3 == 0], dtype=np.int64)
idx = np.stack([np.full_like(gidx, 5), gidx + BIG], axis=1).reshape(-1, 2)
vals = np.stack([0.9 + gidx / 1000, np.full(len(gidx), 0.5), np.full(len(gidx), 0.25)], axis=1).astype(np.float32).reshape(-1, 3)
gi, gv = all_gather_hits(idx, vals, dist)
want = np.array([g for g in range(n_refs) if g %% 3 == 0], dtype=np.int64)
assert gi.dtype == np.int64 and np.array_equal(gi[:, 1], want + BIG) and np.all(gi[:, 0] == 5), (gi, want)
assert np.allclose(gv[:, 0], 0.9 + want / 1000) and np.all(gv[:, 1] == 0.5) and np.all(gv[:, 2] == 0.25)
# a rank with no hits at all
ei, ev = all_gather_hits(np.zeros((0, 2), np.int64) if rank == 1 else idx, np.zeros((0, 3), np.float32) if rank == 1 else vals, dist)
n0 = len([g for g in range(*shard_bounds(n_refs, 0, world)) if g %% 3 == 0])
assert len(ei) == len(ev) == n0 and np.array_equal(ei[:, 1], want[:n0] + BIG)
# the one-collective exchange keeps a capacity from call to call: a larger list (capacity grows, second gather), then a smaller one
st = {}                                                                # the caller owns the capacity (one per ShardedDatabase)
caps = []
for n_each in (40, 3, 0, 17):
    bi = np.stack([np.arange(n_each, dtype=np.int64) + 1000 * rank, np.arange(n_each, dtype=np.int64) * 7 + rank], axis=1).reshape(-1, 2)
    bv = (np.arange(3 * n_each, dtype=np.float32).reshape(-1, 3) + rank) / 8
    n_mine = n_each if rank == 0 else n_each // 2                      # ragged: rank 1 sends half
    oi, ov = all_gather_hits(bi[:n_mine], bv[:n_mine], dist, state=st)
    caps.append(st["cap"])
    w0 = np.stack([np.arange(n_each, dtype=np.int64), np.arange(n_each, dtype=np.int64) * 7], axis=1).reshape(-1, 2)
    w1 = np.stack([np.arange(n_each // 2, dtype=np.int64) + 1000, np.arange(n_each // 2, dtype=np.int64) * 7 + 1], axis=1).reshape(-1, 2)
    assert np.array_equal(oi, np.concatenate([w0, w1])), (n_each, oi)
    v0 = np.arange(3 * n_each, dtype=np.float32).reshape(-1, 3) / 8
    v1 = ((np.arange(3 * n_each, dtype=np.float32).reshape(-1, 3) + 1) / 8)[:n_each // 2]
    assert ov.dtype == np.float32 and np.array_equal(ov, np.concatenate([v0, v1])), n_each
assert caps == [40, 20, 10, 17], caps                                  # grows to fit, decays by halves: one big exchange does not tax later ones
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
# shards balanced on weights (genome lengths ~ seed counts): one 3 Gb genome beside six 5 Mb ones
from pyskani_amd.parallel import weighted_shard_cuts
sdb2 = ShardedDatabase(dist, local=FakeLocal())
w = [5e6, 5e6, 3e9, 5e6, 5e6, 5e6, 5e6]
sdb2.sketch_all(names, lambda i: (names[i].encode(),), weights=w)
assert sdb2._cuts == weighted_shard_cuts(w, world) == [0, 3, 7], sdb2._cuts
assert [h.reference_name for h in sdb2.query("q", b"ACGT")] == ["g0", "g2", "g4", "g6"]
# all-vs-all over stand-ins: 20-byte records (psk_hit_min: query index below bit 31, `learned` in it), three rounds of two genomes per rank, and the
# exchange of round b + 1 STARTED before round b's query returns (the stand-in query waits for it: without the overlap it would time out)
import ctypes as C, threading
from pyskani_amd.parallel import TorchComm, HIT_MIN_DTYPE, HIT_DTYPE, QUERY_MASK
N = 11
class Lib:
    freed = 0
    def psk_sketch_free_many(self, handles, n): Lib.freed += n
class ALocal:
    _device = 0; _ctx = None
    def __init__(self, lo, hi): self.lo, self.hi, self._lib, self.round, self.comm = lo, hi, Lib(), 0, None
    def __len__(self): return self.hi - self.lo
    def sketch_handles(self): return (C.c_void_p * max(1, self.hi - self.lo))(*[1000 + g for g in range(self.lo, self.hi)])      # a genome's "handle" is 1000 + its global index
    def query_handles(self, handles, total, raw=False, **kw):
        b = self.round; self.round += 1
        if b + 1 < self.comm.rounds: assert self.comm.started[b + 1].wait(20), "round %%d's gather had not started while round %%d was queried" %% (b + 1, b)
        dt = HIT_DTYPE if raw else HIT_MIN_DTYPE
        rows, offs = [], [0]
        for qi in range(total):
            q = handles[qi] - 1000
            for r in range(self.lo, self.hi):
                if (q + r) %% 3 == 0:
                    rec = np.zeros(1, dt)
                    rec["ani"], rec["af_query"], rec["af_ref"], rec["ref_index"] = 0.5 + q / 100 + r / 10000, 0.25, 0.75, r - self.lo
                    if raw: rec["learned"] = q %% 2
                    else: rec["query"] = qi | ((q %% 2) << 31)
                    rows.append(rec)
            offs.append(len(rows))
        return (np.concatenate(rows) if rows else np.zeros(0, dt)), np.array(offs, np.int64)
class AComm(TorchComm):
    def __init__(self, dist, rounds): super().__init__(dist); self.rounds, self.calls, self.started = rounds, 0, [threading.Event() for _ in range(rounds)]
    def gather_sketch_handles(self, ctx, handles, dev):
        b = self.calls; self.calls += 1
        self.started[b].set()
        box = [None] * self.world
        self.dist.all_gather_object(box, [int(h) for h in handles])
        flat = [h for part in box for h in part]
        return (C.c_void_p * max(1, len(flat)))(*flat), [len(part) for part in box]
for raw in (False, True):
    lo, hi = shard_bounds(N, rank, world)
    loc = ALocal(lo, hi)
    comm = AComm(dist, rounds=3)
    loc.comm = comm
    sdb3 = ShardedDatabase(dist, local=loc, comm=comm, raw=raw)
    sdb3.adopt_local(["g%%d" %% i for i in range(N)])
    sdb3.trace = []
    Lib.freed = 0
    recs = sdb3.all_vs_all_records(batch=2)
    assert recs.dtype == (HIT_DTYPE if raw else HIT_MIN_DTYPE) and recs.dtype.itemsize == (80 if raw else 20)
    want = [(q, r) for q in range(N) for r in range(N) if (q + r) %% 3 == 0]
    qs = (recs["reserved"] if raw else recs["query"] & QUERY_MASK).tolist()
    assert list(zip(qs, recs["ref_index"].tolist())) == want, (raw, list(zip(qs, recs["ref_index"].tolist()))[:8])
    assert np.allclose(recs["ani"], [0.5 + q / 100 + r / 10000 for q, r in want], atol=1e-6)
    learned = (recs["learned"] != 0) if raw else ((recs["query"] >> 31) != 0)
    assert learned.tolist() == [q %% 2 == 1 for q, r in want]
    assert Lib.freed == N, Lib.freed                      # every gathered handle set was released (N genomes over the three rounds)
    tr = sdb3.trace
    for b in range(2):
        assert tr.index(("gather_start", b + 1)) < tr.index(("query_end", b)), tr       # the next round's exchange runs beside this round's query
    assert comm.calls == 3
# a rank whose local query fails in round 1 (ADVICE r5): it keeps entering the remaining sketch gathers and the hit gather with nothing to add - its peer returns
# (with its own shard's hits only) instead of blocking in a collective the failed rank never enters - and raises when the call is over; every gathered handle is released
class Failing(ALocal):
    def query_handles(self, handles, total, raw=False, **kw):
        if rank == 1 and self.round == 1:
            self.round += 1
            raise RuntimeError("local query failed")
        return super().query_handles(handles, total, raw=raw, **kw)
lo, hi = shard_bounds(N, rank, world)
loc = Failing(lo, hi)
comm = AComm(dist, rounds=3)
loc.comm = comm
sdb4 = ShardedDatabase(dist, local=loc, comm=comm)
sdb4.adopt_local(["g%%d" %% i for i in range(N)])
Lib.freed = 0
try:
    recs = sdb4.all_vs_all_records(batch=2)
    assert rank == 0, "the failed rank must raise"
    r0 = shard_bounds(N, 0, world)
    assert sorted(set(recs["ref_index"].tolist())) == [r for r in range(*r0) if any((q + r) %% 3 == 0 for q in range(N))], recs["ref_index"]
except RuntimeError as e:
    assert rank == 1 and "local query failed" in str(e), (rank, e)
assert comm.calls == 3 and Lib.freed == N, (comm.calls, Lib.freed)
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


def test_weighted_shard_cuts():
    from pyskani_amd.parallel import weighted_shard_cuts
    rng = np.random.default_rng(0)
    for n in (0, 1, 5, 50, 1000):
        for world in (1, 2, 3, 8):
            w = rng.integers(1, 100, n) * (1 + 50 * (rng.random(n) < 0.05))
            cuts = weighted_shard_cuts(w, world)
            assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts, cuts[1:]))
            if n >= 20 * world:
                loads = [w[a:b].sum() for a, b in zip(cuts, cuts[1:])]
                assert max(loads) - min(loads) <= 2 * w.max()          # never worse than one genome either side
    assert weighted_shard_cuts([1, 1, 1, 1], 2) == [0, 2, 4]
    assert weighted_shard_cuts([0, 0, 0], 2) == [0, 2, 3]             # no weight: fall back to counts


def test_all_gather_hits_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]
