import sys; sys.argv=['x']
import numpy as np
from t22 import *
def fam(qs, rs, LQ, LR, qpos, label):
    iv, A, ch = chain(qs, rs, chunk_mode=1, band=20, bp_band=2500, max_gap=300, gap_w=0.5)
    kept = select(iv, 45, 3, True)
    o = np.lexsort((kept['q0'], kept['chunk'])); kept = kept[o]
    cid = kept['chunk']; n_int=len(kept)
    Ac = np.bincount(cid, weights=kept['nanch']); m = Ac > 0
    mn = np.full(cid.max() + 1, 10**10); mx = np.zeros(cid.max() + 1, dtype=int)
    np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
    S = np.searchsorted(qpos, mx[m], 'right') - np.searchsorted(qpos, mn[m], 'left')
    A_=Ac[m]; n=len(A_)
    N = (kept['q1'].astype(int)-kept['q0']).sum()+251*n_int
    print(label, "chunks", n, "ints", n_int, "N", N, "afq %.6f afr(shared) %.6f afr(own) %.6f"%(N/LQ, N/LR, ((kept['r1'].astype(int)-kept['r0']).sum()+251*n_int)/LR))
    for name, a, s in (("A/(S-1)", A_, S-1), ("(A+1)/S", A_+1, S), ("A/S", A_, S), ("(A+.5)/S",A_+0.5,S)):
        v = np.minimum(1, a/np.maximum(s,1))**(1/15); sv=np.sort(v)
        print("   %-9s mean %.5f med %.5f rob %.5f"%(name, v.mean(), sv[n//2], sv[n//10:n-n//10].mean()))
fam(s_k, s_ec, LQ, LR, qpos, "K12 query")
fam(s_ec, s_k, LR, LQ, np.sort(s_ec['pos']), "EC query")
