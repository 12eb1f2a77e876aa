"""8 host threads of Database.query_records (no Python object per hit) over the metagenome bench's shape, smaller: for a kernel trace"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyskani_amd as psk
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", np.uint8)
n_fam, per = 10, 100
anc = [lut[rng.integers(0, 4, 2_000_000)] for _ in range(n_fam)]
def mut(a, d):
    m = rng.random(len(a)) < d
    b = a.copy(); b[m] = lut[rng.integers(0, 4, int(m.sum()))]; return b.tobytes()
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many([(f"r{f}_{j}", mut(anc[f], 0.001 * j)) for f in range(n_fam) for j in range(per)])
NQ = int(os.environ.get("NQ", "4000")); NT = int(os.environ.get("NT", "8"))
contigs = []
for i in range(NQ):
    a = anc[i % n_fam]; L = int(np.exp(rng.uniform(np.log(2000), np.log(50000)))); st = int(rng.integers(0, len(a) - L))
    contigs.append(mut(a[st:st + L], rng.uniform(0, 0.05)))
for c in contigs[:50]:
    db.query("w", c, learned_ani=False)
def work(lo, hi):
    for i in range(lo, hi):
        db.query_records("q", contigs[i], learned_ani=False)
for rep in range(2):
    th = [threading.Thread(target=work, args=(k * NQ // NT, (k + 1) * NQ // NT)) for k in range(NT)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
    print(NT, "threads:", round(NQ / (time.perf_counter() - t0)), "queries/s", flush=True)
