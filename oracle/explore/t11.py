from t4 import *
def v1(qs, rs, LQ, LR, **kw):
    kept, arr, ani = run(qs, rs, LQ, LR, verbose=False, **kw)
    ch = {}
    for g in kept:
        c = ch.setdefault(int(g['chunk']), [10**10, 0, 0, 0, 10**10, 0])
        c[0] = min(c[0], int(g['q0'])); c[1] = max(c[1], int(g['q1'])); c[2] += g['nanch']; c[3]+=1
        c[4] = min(c[4], int(g['r0'])); c[5] = max(c[5], int(g['r1']))
    v = np.array(list(ch.values()))
    span = (v[:,1]-v[:,0]).sum()
    print(kw, "chunks", len(v), "span", span, "AFq", span/LQ, "AFr", span/LR, "need", 0.9189*LQ, "per chunk", (0.9189*LQ-span)/len(v), "per int", (0.9189*LQ-span)/len(kept))
v1(s_k, s_ec, LQ, LR)
v1(s_ec, s_k, LR, LQ)
