import sys
from t4 import *
from t15 import select
def agg(qs, rs, LQ, LR, **kw):
    iv, A, ch = chain(qs, rs, **kw)
    kept = select(iv)
    qpos = np.sort(qs['pos'])
    n_int = len(kept)
    span = (kept['q1'].astype(int)-kept['q0']).sum()
    ns = np.searchsorted(qpos, kept['q1'], 'right') - np.searchsorted(qpos, kept['q0'], 'left')
    cid = kept['chunk']
    Ac = np.bincount(cid, weights=kept['nanch']); Sc = np.bincount(cid, weights=ns); m = Ac > 0
    a = np.minimum(1, Ac[m]/Sc[m]) ** (1/15); s=np.sort(a); n=len(s)
    ai = np.minimum(1, kept['nanch']/ns) ** (1/15); si=np.sort(ai); ni=len(si)
    print(kw, "anchors", len(A), "nint", n_int, "need/int %.1f"%((0.9189*LQ-span)/n_int), "AFq(2c+1) %.6f AFr %.6f" % ((span+251*n_int)/LQ, ((kept['r1'].astype(int)-kept['r0']).sum()+251*n_int)/LR),
      f"chunk: n {n} mean {a.mean():.5f} med {s[n//2]:.5f} rob {s[n//10:n-n//10].mean():.5f} wA {(a*Ac[m]).sum()/Ac[m].sum():.5f} wS {(a*Sc[m]).sum()/Sc[m].sum():.5f} | intv: mean {ai.mean():.5f} med {si[ni//2]:.5f}")
for cm in (0,1):
  for mc in (0, 1, 2, 3, 4, 6, 9, 16):
    agg(s_k, s_ec, LQ, LR, chunk_mode=cm, band=100, bp_band=2500, mult_cap=mc)
