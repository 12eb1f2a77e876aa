from t4 import *
import itertools, time
def select(iv, min_score=45, min_anch=3, refcheck=False):
    good = iv[(iv['nanch'] >= min_anch) & (iv['score'] >= min_score)]
    order = np.argsort(-good['score'], kind='stable')
    kept = []; bychunk = {}; allk = []
    for idx in order:
        g = good[idx]
        lst = bychunk.setdefault(int(g['chunk']), [])
        ok = True
        for h in lst:
            if not (g['q1'] < h['q0'] or g['q0'] > h['q1']): ok = False; break
        if ok and refcheck:
            for h in allk:
                if h['rc']==g['rc'] and not (g['r1'] < h['r0'] or g['r0'] > h['r1']): ok = False; break
        if ok: lst.append(g); kept.append(g); allk.append(g)
    return np.array(kept, dtype=iv.dtype)
T_Q, T_R = 0.9189*LQ, 0.9246*LR
res=[]
for cm, band, bp, mg, gm, rc in itertools.product((0,1),(20,50,100),(0,2500),(50,100,300),(0,2),(False,True)):
    iv, A, ch = chain(s_k, s_ec, chunk_mode=cm, band=band, bp_band=bp, max_gap=mg, gapcost_mode=gm)
    kept = select(iv, refcheck=rc)
    span = (kept['q1'].astype(int)-kept['q0']).sum(); rspan=(kept['r1'].astype(int)-kept['r0']).sum()
    n = len(kept)
    print(cm, band, bp, mg, gm, rc, "n", n, "span", span, rspan, "need/int %.1f %.1f" % ((T_Q-span)/n, (T_R-rspan)/n), flush=True)
