from t4 import *
def ext(qs, rs, LQ, LR, es, **kw):
    kept, arr, ani = run(qs, rs, LQ, LR, verbose=False, **kw)
    qpos = np.sort(qs['pos'])
    for e in es:
        lo = np.searchsorted(qpos, kept['q0'].astype(int) - e, 'left'); hi = np.searchsorted(qpos, kept['q1'].astype(int) + e, 'right')
        ns = hi - lo
        cid = kept['chunk']
        A = np.bincount(cid, weights=kept['nanch']); S = np.bincount(cid, weights=ns)
        m = A > 0
        a = np.minimum(1, A[m]/S[m]) ** (1/15)
        ai = np.minimum(1, kept['nanch']/ns) ** (1/15)
        s = np.sort(a); n = len(s); si = np.sort(ai); ni = len(si)
        span = (kept['q1'].astype(int) - kept['q0'] + 2*e).sum()
        print(f"e={e:4d} chunk: mean {a.mean():.5f} med {s[n//2]:.5f} rob {s[n//10:n-n//10].mean():.5f} wS {(a*S[m]).sum()/S[m].sum():.5f} wA {(a*A[m]).sum()/A[m].sum():.5f} | intv: mean {ai.mean():.5f} med {si[ni//2]:.5f} wA {(ai*kept['nanch']).sum()/kept['nanch'].sum():.5f} wS {(ai*ns).sum()/ns.sum():.5f}| AFq {span/LQ:.5f} AFr {span/LR:.5f}")
print("K12 as query, maxgap 50"); ext(s_k, s_ec, LQ, LR, [0, 15, 30, 62, 100, 125, 150, 200, 250, 300, 400])
print("K12 as query, maxgap 300"); ext(s_k, s_ec, LQ, LR, [0, 62, 125, 200, 250], max_gap=300)
print("EC as query, maxgap 50"); ext(s_ec, s_k, LR, LQ, [0, 62, 125, 200, 250])
