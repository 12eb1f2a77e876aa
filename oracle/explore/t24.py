import sys; sys.argv=['x']
import numpy as np
from t22 import *
cfg=(1, 20, 2500, 300, 2, 45, 3, True)
cm, band, bp, mg, gm, ms, ma, rc = cfg
iv, A, ch = chain(s_k, s_ec, chunk_mode=cm, band=band, bp_band=bp, max_gap=mg, gapcost_mode=gm)
kept = select(iv, ms, ma, rc)
o = np.lexsort((kept['q0'], kept['chunk'])); kept = kept[o]
cid = kept['chunk']; n_int=len(kept)
Ac = np.bincount(cid, weights=kept['nanch']); m = Ac > 0
mn = np.full(cid.max() + 1, 10**10); mx = np.zeros(cid.max() + 1, dtype=int)
np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
S = np.searchsorted(qpos, mx[m], 'right') - np.searchsorted(qpos, mn[m], 'left')
A_=Ac[m]
print("n chunks", m.sum(), "n_int", n_int, "cov_q", (kept['q1'].astype(int)-kept['q0']).sum()+251*n_int)
for name, a, s in (("lr", A_, S), ("lr-1", A_, S-1)):
    v = np.minimum(1, a/s)**(1/15); oo=np.argsort(v); n=len(v)
    print(name, "mean %.5f med %.5f"%(v.mean(), v[oo][n//2]))
    print("  around median:", [(int(a[i]), int(s[i]), round(float(v[i]),5)) for i in oo[n//2-14:n//2+6]])
    miss = s - a
    print("  misses histogram", np.bincount(np.minimum(miss.astype(int),6)))
