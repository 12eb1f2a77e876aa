import sys, time, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import numpy as np, torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n_refs = 5000
buf, offs, lens = bench.make_genomes(torch, dev, seed_shared=2, seed_members=3, n_refs=n_refs, n_families=50)
eng = bench.Engine(0); eng.set_params(30, 200)
names = (C.c_char_p * n_refs)(*[f"r{i}".encode() for i in range(n_refs)])
handles = eng.sketch_device(buf.data_ptr(), offs[:n_refs], lens[:n_refs])
db = eng.make_db(names, handles, n_refs)
cbuf, coffs, clens = bench.make_contigs(torch, dev, buf, offs, lens, n_refs, 10000, seed=4)
torch.cuda.synchronize()
acc = {}
def T(k, f):
    t0 = time.perf_counter(); r = f(); acc[k] = acc.get(k, 0.0) + time.perf_counter() - t0; return r
for it in range(5):
    if it == 2: acc.clear()
    qh = T("sketch", lambda: eng.sketch_device(cbuf.data_ptr(), coffs, clens))
    T("query_many", lambda: eng.query_many(db, qh, len(coffs)))
    def fr():
        for h in qh: eng.lib.psk_sketch_free(h)
    T("free", fr)
print({k: round(v / 3 * 1e3, 2) for k, v in acc.items()})
