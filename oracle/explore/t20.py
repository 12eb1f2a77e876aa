import sys, itertools
from t4 import sketch, chain, np
from t18 import select
rng = np.random.default_rng(1)
L = 2_000_000
g = rng.integers(0, 4, L).astype(np.uint8)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
def mutate(g, d):
    m = rng.random(len(g)) < d
    h = g.copy(); h[m] = (h[m] + rng.integers(1, 4, m.sum())) % 4
    return h
ref = lut[g].tobytes(); s_r, _ = sketch(ref)
for d in (0.01, 0.03, 0.05, 0.08, 0.12, 0.16):
    q = lut[mutate(g, d)].tobytes(); s_q, _ = sketch(q)
    qpos = np.sort(s_q['pos'])
    for cm in (1,):
        iv, A, ch = chain(s_q, s_r, chunk_mode=cm, band=100, bp_band=2500)
        kept = select(iv)
        if len(kept)==0: print(d, "no chains"); continue
        ns = np.searchsorted(qpos, kept['q1'], 'right') - np.searchsorted(qpos, kept['q0'], 'left')
        cid = kept['chunk']
        Ac = np.bincount(cid, weights=kept['nanch']); Sc = np.bincount(cid, weights=ns); m = Ac > 0
        mn = np.full(cid.max()+1, 10**10); mx = np.zeros(cid.max()+1, dtype=int)
        np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
        Sl = np.searchsorted(qpos, mx[m], 'right') - np.searchsorted(qpos, mn[m], 'left')
        a = np.minimum(1, Ac[m]/Sc[m]) ** (1/15); al = np.minimum(1, Ac[m]/Sl) ** (1/15)
        span = (kept['q1'].astype(int)-kept['q0']+251).sum()
        # extended per-interval
        nse = np.searchsorted(qpos, kept['q1'].astype(int)+125, 'right') - np.searchsorted(qpos, kept['q0'].astype(int)-125, 'left')
        ae = np.minimum(1, kept['nanch']/nse)**(1/15)
        Sce = np.bincount(cid, weights=nse); ace = np.minimum(1, Ac[m]/Sce[m])**(1/15)
        print(f"true {1-d:.3f} anchors {len(A)} nint {len(kept)} AF {span/L:.3f} | in-span chunk {a.mean():.4f} LR {al.mean():.4f} ext-c int {ae.mean():.4f} ext-c chunk {ace.mean():.4f} global-LR {(Ac[m].sum()/Sl.sum())**(1/15):.4f}")
