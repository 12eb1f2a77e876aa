import sys
from t15 import *
rpos = np.sort(s_ec['pos'])
nr = np.searchsorted(rpos, kept['r1'], 'right') - np.searchsorted(rpos, kept['r0'], 'left')
def rep(name, Sint):
    Ac = np.bincount(cid, weights=kept['nanch']); Sc = np.bincount(cid, weights=Sint); m=Ac>0
    a = np.minimum(1, Ac[m]/Sc[m])**(1/15); s=np.sort(a); n=len(s)
    ai = np.minimum(1, kept['nanch']/Sint)**(1/15); si=np.sort(ai); ni=len(si)
    print(name, "chunk mean %.5f med %.5f rob %.5f wA %.5f| int mean %.5f med %.5f" % (a.mean(), s[n//2], s[n//10:n-n//10].mean(), (a*Ac[m]).sum()/Ac[m].sum(), ai.mean(), si[ni//2]))
rep("q", ns); rep("r", nr); rep("max", np.maximum(ns,nr)); rep("avg", (ns+nr)/2); rep("min", np.minimum(ns,nr))
