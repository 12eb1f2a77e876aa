import sys; sys.argv=['x']
import numpy as np
from ex import sketch, chain
from t22 import select
rng = np.random.default_rng(1)
L = 2_000_000
g = rng.integers(0, 4, L).astype(np.uint8)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
def mutate(g, d):
    m = rng.random(len(g)) < d
    h = g.copy(); h[m] = (h[m] + rng.integers(1, 4, m.sum())) % 4
    return h
ref = lut[g].tobytes(); s_r, _ = sketch(ref)
for d in (0.0, 0.01, 0.03, 0.05, 0.08, 0.12, 0.16):
    q = lut[mutate(g, d)].tobytes(); s_q, _ = sketch(q)
    qpos = np.sort(s_q['pos'])
    iv, A, ch = chain(s_q, s_r, chunk_mode=1, band=20, bp_band=2500, max_gap=300, gap_w=0.5)
    kept = select(iv, 45, 3, True)
    if len(kept)==0: print(d, "no chains"); continue
    cid = kept['chunk']
    Ac = np.bincount(cid, weights=kept['nanch']); m = Ac > 0
    mn = np.full(cid.max()+1, 10**10); mx = np.zeros(cid.max()+1, dtype=int)
    np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
    S = np.searchsorted(qpos, mx[m], 'right') - np.searchsorted(qpos, mn[m], 'left')
    v = np.minimum(1, Ac[m]/np.maximum(S-1,1))**(1/15)
    v0 = np.minimum(1, Ac[m]/S)**(1/15)
    N = (kept['q1'].astype(int)-kept['q0']+251).sum()
    print(f"true {1-d:.3f} nint {len(kept)} AF {min(1,N/L):.3f} | A/(S-1) {v.mean():.4f}  A/S {v0.mean():.4f}")
