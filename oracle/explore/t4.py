from ex import *
import bisect
ec = load("e.coli-EC590.fasta.gz"); k12 = load("e.coli-K12.fasta.gz")
s_ec, m_ec = sketch(ec); s_k, m_k = sketch(k12)
LQ, LR = len(k12), len(ec)

def run(qs, rs, LQ, LR, min_score=45, min_anch=3, verbose=True, **kw):
    iv, A, ch = chain(qs, rs, **kw)
    qpos = np.sort(qs['pos'])
    good = iv[(iv['nanch'] >= min_anch) & (iv['score'] >= min_score)]
    # greedy non-overlap on query within chunk by score
    order = np.argsort(-good['score'], kind='stable')
    kept = []
    bychunk = {}
    for idx in order:
        g = good[idx]
        lst = bychunk.setdefault(int(g['chunk']), [])
        ok = True
        for h in lst:
            if not (g['q1'] < h['q0'] or g['q0'] > h['q1']): ok = False; break
        if ok: lst.append(g); kept.append(g)
    kept = np.array(kept, dtype=iv.dtype)
    # seeds in each interval
    lo = np.searchsorted(qpos, kept['q0'], 'left'); hi = np.searchsorted(qpos, kept['q1'], 'right')
    ns = hi - lo
    # per chunk aggregate
    chunks = {}
    for g, n in zip(kept, ns):
        c = chunks.setdefault(int(g['chunk']), [0, 0, 0, 0])
        c[0] += g['nanch']; c[1] += n; c[2] += int(g['q1']) - int(g['q0']); c[3] += 1
    arr = np.array([v for v in chunks.values()], dtype=float)
    ratio = np.minimum(1, arr[:, 0] / arr[:, 1])
    ani = ratio ** (1 / 15)
    span = arr[:, 2].sum(); nint = arr[:, 3].sum()
    if verbose:
        print("kept intervals", len(kept), "chunks", len(arr), "span", span, "AFq", span / LQ, "AFr", span / LR, "need", 0.9189 * LQ, "delta/int", (0.9189 * LQ - span) / nint)
        print(" mean", ani.mean(), "w-anch", (ani * arr[:, 0]).sum() / arr[:, 0].sum(), "w-seeds", (ani * arr[:, 1]).sum() / arr[:, 1].sum(), "global", (arr[:,0].sum()/arr[:,1].sum())**(1/15), "median", np.median(ani), np.sort(ani)[len(ani)//2])
        s = np.sort(ani); n = len(s)
        print(" robust", s[n//10: n - n//10].mean(), "seeds/chunk median", np.median(arr[:,1]))
    return kept, arr, ani
if __name__ == "__main__":
    print("query=K12"); run(s_k, s_ec, LQ, LR)
    print("query=EC590 (switched)"); run(s_ec, s_k, LR, LQ)
