import sys, itertools
from t4 import *
def select(iv, min_score=45, min_anch=3):
    good = iv[(iv['nanch'] >= min_anch) & (iv['score'] >= min_score)]
    order = np.argsort(-good['score'], kind='stable')
    kept = []; bychunk = {}
    for idx in order:
        g = good[idx]
        lst = bychunk.setdefault(int(g['chunk']), [])
        ok = True
        for h in lst:
            if not (g['q1'] < h['q0'] or g['q0'] > h['q1']): ok = False; break
        if ok: lst.append(g); kept.append(g)
    return np.array(kept, dtype=iv.dtype)
qpos = np.sort(s_k['pos'])
for cm, band, bp in itertools.product((0,1),(50,100),(0,2500)):
    iv, A, ch = chain(s_k, s_ec, chunk_mode=cm, band=band, bp_band=bp)
    for ma, ms in itertools.product((3,4,5,6),(45,60,80,100,150)):
        kept = select(iv, ms, ma)
        n_int=len(kept); span=(kept['q1'].astype(int)-kept['q0']).sum()
        ns = np.searchsorted(qpos, kept['q1'], 'right') - np.searchsorted(qpos, kept['q0'], 'left')
        cid = kept['chunk']
        Ac = np.bincount(cid, weights=kept['nanch']); Sc = np.bincount(cid, weights=ns); m = Ac > 0
        mn = np.full(cid.max()+1, 10**10); mx = np.zeros(cid.max()+1, dtype=int)
        np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
        Sl = np.searchsorted(qpos, mx[m], 'right') - np.searchsorted(qpos, mn[m], 'left')
        a = np.minimum(1, Ac[m]/Sc[m]) ** (1/15); s=np.sort(a); n=len(s)
        al = np.minimum(1, Ac[m]/Sl) ** (1/15); sl=np.sort(al)
        print(cm,band,bp,ma,ms,"nint",n_int,"need/int %.1f"%((0.9189*LQ-span)/n_int), "span: mean %.5f med %.5f | LR: mean %.5f med %.5f rob %.5f"%(a.mean(), s[n//2], al.mean(), sl[n//2], sl[n//10:n-n//10].mean()))
