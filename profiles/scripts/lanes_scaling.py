"""Per-contig Database.query() from T host threads (lanes): queries per second for T = 1, 2, 4, 8, 16 ($PSK_LANES must allow T)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pyskani_amd as psk
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", np.uint8)
anc = lut[rng.integers(0, 4, 2_000_000)]
def mut(a, d):
    m = rng.random(len(a)) < d
    b = a.copy(); b[m] = lut[rng.integers(0, 4, int(m.sum()))]; return b.tobytes()
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many([(f"r{j}", mut(anc, 0.001 * j)) for j in range(200)])
contigs = [mut(anc[a:a + 10000], 0.01) for a in rng.integers(0, len(anc) - 10000, 1600)]
for c in contigs[:50]:
    db.query("w", c, learned_ani=False)
def work(lo, hi):
    for c in contigs[lo:hi]:
        db.query("q", c, learned_ani=False)
for T in (1, 2, 4, 8, 16):
    n = len(contigs)
    th = [threading.Thread(target=work, args=(i * n // T, (i + 1) * n // T)) for i in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    print(f"{T:2d} threads: {n / dt:7.0f} queries/s")
