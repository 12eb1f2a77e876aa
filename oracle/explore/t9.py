from t4 import *
kept, arr, ani = run(s_k, s_ec, LQ, LR, verbose=False)
s=np.sort(ani); n=len(s)
print(n, s[n//2-25:n//2+25].round(5))
# arr columns: anchors, seeds, span, nint
o=np.argsort(ani)
print(arr[o][n//2-25:n//2+25,:2])
