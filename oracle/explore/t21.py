import sys, itertools
from t4 import *
from t18 import select
qpos = np.sort(s_k['pos'])
out = open("/tmp/t21.txt", "w")
T = dict(afq=0.9189, afr=0.9246, mean=0.9946, med=0.9995)
for cm, band, bp, mg, gm in itertools.product((0,1),(20,50,100,200),(0,2500),(50,100,200,300),(0,2)):
    iv, A, ch = chain(s_k, s_ec, chunk_mode=cm, band=band, bp_band=bp, max_gap=mg, gapcost_mode=gm)
    for ma, ms in ((3,45),(4,45),(5,45),(3,80),(3,100)):
        kept = select(iv, ms, ma)
        n_int=len(kept); span=(kept['q1'].astype(int)-kept['q0']).sum(); rspan=(kept['r1'].astype(int)-kept['r0']).sum()
        cid = kept['chunk']
        Ac = np.bincount(cid, weights=kept['nanch']); m = Ac > 0
        mn = np.full(cid.max()+1, 10**10); mx = np.zeros(cid.max()+1, dtype=int)
        np.minimum.at(mn, cid, kept['q0'].astype(int)); np.maximum.at(mx, cid, kept['q1'].astype(int))
        Sl = np.searchsorted(qpos, mx[m], 'right') - np.searchsorted(qpos, mn[m], 'left')
        al = np.minimum(1, Ac[m]/Sl) ** (1/15); sl=np.sort(al); n=len(sl)
        for ext in (250, 251, 265, 266):
            afq=(span+ext*n_int)/LQ; afr=(rspan+ext*n_int)/LR
            err = abs(afq-T['afq'])+abs(afr-T['afr'])+abs(al.mean()-T['mean'])+abs(sl[n//2]-T['med'])
            out.write(f"{err:.6f} cm={cm} band={band} bp={bp} mg={mg} gm={gm} ma={ma} ms={ms} ext={ext} n={n} nint={n_int} afq={afq:.5f} afr={afr:.5f} mean={al.mean():.5f} med={sl[n//2]:.5f} rob={sl[n//10:n-n//10].mean():.5f}\n")
    out.flush()
