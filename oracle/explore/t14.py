from t13 import select
from t4 import *
def agg(qs, rs, LQ, LR, **kw):
    iv, A, ch = chain(qs, rs, **kw)
    kept = select(iv)
    qpos = np.sort(qs['pos'])
    n_int = len(kept)
    span = (kept['q1'].astype(int)-kept['q0']).sum()
    print(kw, "nint", n_int, "AFq(2c+1) %.6f AFr %.6f" % ((span+251*n_int)/LQ, ((kept['r1'].astype(int)-kept['r0']).sum()+251*n_int)/LR))
    for e in (0, 15, 62, 125):
        lo = np.searchsorted(qpos, kept['q0'].astype(int) - e, 'left'); hi = np.searchsorted(qpos, kept['q1'].astype(int) + e, 'right')
        ns = hi - lo
        cid = kept['chunk']
        Ac = np.bincount(cid, weights=kept['nanch']); Sc = np.bincount(cid, weights=ns); m = Ac > 0
        a = np.minimum(1, Ac[m]/Sc[m]) ** (1/15); s=np.sort(a); n=len(s)
        ai = np.minimum(1, kept['nanch']/ns) ** (1/15); si=np.sort(ai); ni=len(si)
        # leftmost-rightmost
        mn = np.full(cid.max()+1, 10**10); mx = np.zeros(cid.max()+1, dtype=int)
        np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
        Sl = np.searchsorted(qpos, mx[m]+e, 'right') - np.searchsorted(qpos, mn[m]-e, 'left')
        al = np.minimum(1, Ac[m]/Sl) ** (1/15); sl=np.sort(al)
        print(f" e={e:3d} chunk: mean {a.mean():.5f} med {s[n//2]:.5f} rob {s[n//10:n-n//10].mean():.5f} wA {(a*Ac[m]).sum()/Ac[m].sum():.5f} wS {(a*Sc[m]).sum()/Sc[m].sum():.5f} | intv: mean {ai.mean():.5f} med {si[ni//2]:.5f} wA {(ai*kept['nanch']).sum()/kept['nanch'].sum():.5f} | LR: mean {al.mean():.5f} med {sl[n//2]:.5f} rob {sl[n//10:n-n//10].mean():.5f} wA {(al*Ac[m]).sum()/Ac[m].sum():.5f}")
agg(s_k, s_ec, LQ, LR, chunk_mode=1, band=100, bp_band=2500)
agg(s_k, s_ec, LQ, LR, chunk_mode=1, band=50, bp_band=2500)
print("switched")
agg(s_ec, s_k, LR, LQ, chunk_mode=1, band=100, bp_band=2500)
