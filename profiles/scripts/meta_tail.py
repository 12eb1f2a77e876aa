"""Host-side split of one metagenome step (100 k contigs x 5 000 references): the sketch call, psk_query_many, psk_free of the
hit array, psk_sketch_free_many - the GPU is idle during all but the first two (profiles/r3/r3i_metagenome_gaps.txt)."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n_refs, n_queries = 5000, int(os.environ.get("NQ", "100000"))
n_families = n_refs // 100
anc_lens, fam_of = bench.family_layout(4, n_refs, n_families)
buf, offs, lens = bench.make_genomes(torch, dev, 4, 41, list(range(n_refs)), fam_of, anc_lens)
eng = bench.Engine(0, 30, 200)
names = (C.c_char_p * n_refs)(*[f"r{i}".encode() for i in range(n_refs)])
db = eng.make_db(names, eng.sketch_device(buf.data_ptr(), offs, lens), n_refs)
cbuf, coffs, clens = bench.make_contigs(torch, dev, buf, offs, lens, n_refs, n_queries, seed=4)
torch.cuda.synchronize()
c_off, c_len, gfc, nq = eng.layout(coffs, clens)
opts = eng.capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None)
acc = {}
def T(k, f):
    t0 = time.perf_counter(); r = f(); acc[k] = acc.get(k, 0.0) + time.perf_counter() - t0; return r
R = 3
for it in range(1 + R):
    if it == 1: acc.clear()
    qh = T("sketch_batch_device", lambda: eng.sketch_device_c(cbuf.data_ptr(), c_off, c_len, gfc, nq))
    hits_p = C.POINTER(eng.capi.Hit)(); o = (C.c_uint64 * (nq + 1))()
    T("psk_query_many", lambda: eng.capi.check(eng.lib.psk_query_many(db, qh, nq, C.byref(opts), C.byref(hits_p), o)))
    T("psk_free(hits)", lambda: eng.lib.psk_free(hits_p))
    T("psk_sketch_free_many", lambda: eng.lib.psk_sketch_free_many(qh, nq))
    T("ctx_synchronize", eng.sync)
print(int(o[nq]), "hits;", {k: round(v / R * 1e3, 2) for k, v in acc.items()}, "ms per step")
