import sys; sys.argv=['x']
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
iv, A, ch = chain(s_k, s_ec, chunk_mode=1, band=100, bp_band=2500)
kept = select(iv)
qpos = np.sort(s_k['pos'])
cid = kept['chunk']
ns = np.searchsorted(qpos, kept['q1'], 'right') - np.searchsorted(qpos, kept['q0'], 'left')
tot_def=0; tot_def_lr=0
for c in np.unique(cid):
    m = cid==c
    k = kept[m]; o=np.argsort(k['q0']); k=k[o]; n=ns[m][o]
    Sl = np.searchsorted(qpos, k['q1'].max(), 'right') - np.searchsorted(qpos, k['q0'].min(), 'left')
    a = (k['nanch'].sum()/n.sum())**(1/15); al=(k['nanch'].sum()/Sl)**(1/15)
    tot_def += 1-a; tot_def_lr += 1-al
    if len(k)>1 or a<0.99:
        print(c, "ani %.4f lr %.4f"%(a,al), [(int(x['q0']), int(x['q1'])-int(x['q0']), int(x['nanch']), int(s), round(float(x['score'])), int(x['rev']), int(x['r0'])) for x,s in zip(k,n)])
print(tot_def, tot_def_lr, "target", 0.0054*len(np.unique(cid)))
