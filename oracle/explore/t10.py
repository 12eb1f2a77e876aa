from t4 import *
def v1(qs, rs, LQ, LR, e=0, **kw):
    kept, arr, ani = run(qs, rs, LQ, LR, verbose=False, **kw)
    qpos = np.sort(qs['pos'])
    ch = {}
    for g in kept:
        c = ch.setdefault(int(g['chunk']), [10**10, 0, 0, 0])
        c[0] = min(c[0], int(g['q0'])); c[1] = max(c[1], int(g['q1'])); c[2] += g['nanch']; c[3]+=1
    v = np.array(list(ch.values()))
    S = np.searchsorted(qpos, v[:,1]+e, 'right') - np.searchsorted(qpos, v[:,0]-e, 'left')
    a = np.minimum(1, v[:,2]/S) ** (1/15)
    s = np.sort(a); n = len(s)
    print(kw, "e",e,"n", n, "mean %.5f med %.5f rob %.5f wS %.5f" % (a.mean(), s[n//2], s[n//10:n-n//10].mean(), (a*S).sum()/S.sum()))
for e in (0, 62, 125):
    v1(s_k, s_ec, LQ, LR, e)
    v1(s_k, s_ec, LQ, LR, e, max_gap=300)
