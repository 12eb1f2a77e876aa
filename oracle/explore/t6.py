from t4 import *
def chunked(kept, qs, A_all=None, frag=20000, verbose=True):
    pass
def run2(qs, rs, LQ, LR, min_score=45, min_anch=3, **kw):
    iv, A, ch = chain(qs, rs, chunk_mode=2, **kw)
    good = iv[(iv['nanch'] >= min_anch) & (iv['score'] >= min_score)]
    order = np.argsort(-good['score'], kind='stable')
    kept = []
    for idx in order:
        g = good[idx]; ok = True
        for h in kept:
            if not (g['q1'] < h['q0'] or g['q0'] > h['q1']): ok = False; break
        if ok: kept.append(g)
    kept = np.array(kept, dtype=iv.dtype)
    qpos = np.sort(qs['pos'])
    lo = np.searchsorted(qpos, kept['q0'], 'left'); hi = np.searchsorted(qpos, kept['q1'], 'right')
    ns = hi - lo
    span = (kept['q1'].astype(int) - kept['q0']).sum(); rspan = (kept['r1'].astype(int) - kept['r0']).sum()
    print("cands", len(good), "kept", len(kept), "qspan", span, "AFq", span/LQ, "rspan", rspan, "AFr", rspan/LR, "need", 0.9189*LQ, "d/int", (0.9189*LQ-span)/len(kept))
    print(" global ani", (kept['nanch'].sum()/ns.sum())**(1/15), "anchors", kept['nanch'].sum(), "seeds", ns.sum())
    return kept, ns
if __name__ == "__main__":
  for mg in (50,100,300):
    for band in (20,50,100):
      for bp in (0,2500):
        print("maxgap",mg,"band",band,"bp",bp)
        run2(s_k, s_ec, LQ, LR, max_gap=mg, band=band, bp_band=bp)
