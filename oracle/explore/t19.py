import sys, itertools
from t4 import *
from t18 import select
qpos = np.sort(s_k['pos'])
iv, A, ch = chain(s_k, s_ec, chunk_mode=1, band=100, bp_band=2500)
kept = select(iv)
o = np.lexsort((kept['q0'], kept['chunk'])); kept = kept[o]
ns = np.searchsorted(qpos, kept['q1'], 'right') - np.searchsorted(qpos, kept['q0'], 'left')
cid = kept['chunk']
def evalS(extra, label):
    Ac = np.bincount(cid, weights=kept['nanch']); Sc = np.bincount(cid, weights=ns + extra); m = Ac > 0
    a = np.minimum(1, Ac[m]/Sc[m]) ** (1/15); s=np.sort(a); n=len(s)
    print(label, "mean %.5f med %.5f rob %.5f" % (a.mean(), s[n//2], s[n//10:n-n//10].mean()))
# gap seeds between adjacent intervals in same chunk
gap_seeds = np.zeros(len(kept)); gap_len = np.zeros(len(kept)); same = np.zeros(len(kept), bool)
for i in range(1, len(kept)):
    if cid[i] == cid[i-1]:
        gap_len[i] = int(kept['q0'][i]) - int(kept['q1'][i-1])
        gap_seeds[i] = np.searchsorted(qpos, kept['q0'][i], 'left') - np.searchsorted(qpos, kept['q1'][i-1], 'right')
        same[i] = (kept['rev'][i]==kept['rev'][i-1]) and (kept['rc'][i]==kept['rc'][i-1])
for G in (0, 200, 500, 1000, 1500, 2000, 2500, 3000, 4000, 5000, 7500, 10000, 20000):
    evalS(np.where(gap_len <= G, gap_seeds, 0), f"G={G}")
