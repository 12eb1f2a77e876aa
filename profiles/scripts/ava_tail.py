"""Host-side split of one all-vs-all step (1 000 x 1 000): sketch call, database build, psk_query_many, record copy, psk_free, psk_db_destroy."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n = 1000
anc_lens, fam_of = bench.family_layout(3, n, n // 100)
buf, offs, lens = bench.make_genomes(torch, dev, 3, 31, list(range(n)), fam_of, anc_lens)
torch.cuda.synchronize()
eng = bench.Engine(0)
names = (C.c_char_p * n)(*[f"g{i}".encode() for i in range(n)])
c_off, c_len, gfc, nn = eng.layout(offs, lens)
opts = eng.capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None)
acc = {}
def T(k, f):
    t0 = time.perf_counter(); r = f(); acc[k] = acc.get(k, 0.0) + time.perf_counter() - t0; return r
R = 4
for it in range(1 + R):
    if it == 1: acc.clear()
    out = T("sketch_batch_device", lambda: eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn))
    db = T("db_create+add_batch", lambda: eng.make_db(names, out, nn))
    hits_p = C.POINTER(eng.capi.Hit)(); o = (C.c_uint64 * (nn + 1))()
    T("psk_query_many", lambda: eng.capi.check(eng.lib.psk_query_many(db, out, nn, C.byref(opts), C.byref(hits_p), o)))
    nh = int(o[nn])
    T("records copy", lambda: np.frombuffer((eng.capi.Hit * nh).from_address(C.addressof(hits_p.contents)), dtype=eng.hit_dtype).copy())
    T("psk_free(hits)", lambda: eng.lib.psk_free(hits_p))
    T("psk_db_destroy", lambda: eng.lib.psk_db_destroy(db))
    T("ctx_synchronize", eng.sync)
print(nh, "hits;", {k: round(v / R * 1e3, 3) for k, v in acc.items()}, "ms per step; sum", round(sum(acc.values()) / R * 1e3, 2))
