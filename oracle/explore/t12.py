from t4 import *
import itertools
def v1(qs, rs, LQ, LR, min_score=45, **kw):
    kept, arr, ani = run(qs, rs, LQ, LR, verbose=False, min_score=min_score, **kw)
    qpos = np.sort(qs['pos'])
    ch = {}
    for g in kept:
        c = ch.setdefault(int(g['chunk']), [10**10, 0, 0, 0])
        c[0] = min(c[0], int(g['q0'])); c[1] = max(c[1], int(g['q1'])); c[2] += g['nanch']; c[3]+=1
    v = np.array(list(ch.values()))
    S = np.searchsorted(qpos, v[:,1], 'right') - np.searchsorted(qpos, v[:,0], 'left')
    a = np.minimum(1, v[:,2]/S) ** (1/15)
    s = np.sort(a); n = len(s)
    span = (kept['q1'].astype(int)-kept['q0']).sum()
    print(kw, min_score, "n", n, "nint", len(kept), "mean %.5f med %.5f rob %.5f | span %d +2c: AFq %.5f AFr %.5f" % (a.mean(), s[n//2], s[n//10:n-n//10].mean(), span, (span+250*len(kept))/LQ, (span+250*len(kept))/LR))
for band, bp, cm in itertools.product((20, 50, 100), (0, 2500), (0, 1)):
    v1(s_k, s_ec, LQ, LR, band=band, bp_band=bp, chunk_mode=cm)
for ms in (0, 20, 45, 60, 100):
    v1(s_k, s_ec, LQ, LR, min_score=ms)
