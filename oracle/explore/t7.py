from t4 import *
def allseeds(qs, rs, LQ, **kw):
    kept, arr, ani = run(qs, rs, LQ, LR, verbose=False, **kw)
    qpos = np.sort(qs['pos'])
    # fixed grid chunk id = pos//20000 ; kept['chunk'] index is sequential over chunks with anchors; map through q0
    cid = kept['q0'] // 20000
    tot = np.bincount(qpos // 20000)
    A = np.bincount(cid, weights=kept['nanch'], minlength=len(tot))
    m = A > 0
    r = np.minimum(1, A[m] / tot[m]); a = r ** (1/15)
    s = np.sort(a); n = len(s)
    print(kw, "chunks", n, "mean", a.mean(), "median", s[n//2], "robust", s[n//10:n-n//10].mean(), "w", (a*tot[m]).sum()/tot[m].sum())
    return a, A[m], tot[m]
for mg in (50, 300):
    a, A, T = allseeds(s_k, s_ec, LQ, max_gap=mg)
print(np.sort(a)[100:130])
