"""Where the time of ONE Database.query(name, contig) goes (host-side wall clock, 10 kb contig against 200 x 2 Mb references):
sketch call, query call, Python object construction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ctypes as C
import pyskani_amd as psk
from pyskani_amd import _capi
rng = np.random.default_rng(1)
lut = np.frombuffer(b"ACGT", np.uint8)
anc = lut[rng.integers(0, 4, 2_000_000)]
def mut(a, d):
    m = rng.random(len(a)) < d
    b = a.copy(); b[m] = lut[rng.integers(0, 4, int(m.sum()))]; return b.tobytes()
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many([(f"r{j}", mut(anc, 0.001 * j)) for j in range(200)])
contigs = [mut(anc[a:a + 10000], 0.01) for a in rng.integers(0, len(anc) - 10000, 300)]
for c in contigs[:20]:
    db.query("w", c, learned_ani=False)
N = len(contigs)
t0 = time.perf_counter(); sk = [db._sketch("q", [c], True) for c in contigs]; t_sk = (time.perf_counter() - t0) / N
opts = db._opts(False, False, False, None, False)
lib = db._lib
t0 = time.perf_counter()
nh = 0
for s in sk:
    hits_p = C.POINTER(_capi.Hit)(); n = C.c_uint64(0)
    _capi.check(lib.psk_query(db._h, s._h, C.byref(opts), C.byref(hits_p), C.byref(n)))
    nh += n.value
    lib.psk_free(hits_p)
t_q = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for c in contigs:
    db.query("q", c, learned_ani=False)
t_all = (time.perf_counter() - t0) / N
print(f"per query: sketch {t_sk * 1e3:.3f} ms, psk_query {t_q * 1e3:.3f} ms ({nh / N:.0f} hits), whole Database.query {t_all * 1e3:.3f} ms")
