from t4 import *
def stats(kept, qs):
    qpos = np.sort(qs['pos'])
    lo = np.searchsorted(qpos, kept['q0'], 'left'); hi = np.searchsorted(qpos, kept['q1'], 'right')
    ns = hi - lo
    r = np.minimum(1, kept['nanch'] / ns); ani = r ** (1/15)
    s = np.sort(ani); n = len(s)
    print(" per-interval: n", n, "mean", ani.mean(), "w-anch", (ani*kept['nanch']).sum()/kept['nanch'].sum(), "w-seeds", (ani*ns).sum()/ns.sum(), "median", s[n//2], np.median(ani), "robust", s[n//10:n-n//10].mean())
for mg in (50, 100, 200, 300, 1000):
    for gm in (0, 2, 1):
        print("max_gap", mg, "gapmode", gm)
        kept, arr, ani = run(s_k, s_ec, LQ, LR, max_gap=mg, gapcost_mode=gm)
        stats(kept, s_k)
