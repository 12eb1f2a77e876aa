import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes as C
from pyskani_amd import _capi
lib = _capi.load()
ctx = C.c_void_p(); _capi.check(lib.psk_ctx_create(0, C.byref(ctx)))
params = _capi.Params(125, 1000, 15)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000   # default 8 contigs x 125 Mb = 1 Gb genome
DO_QUERY = (sys.argv[3] != '0') if len(sys.argv) > 3 else True
offs, lens, total = [], [], 0
for i in range(2 * NC):
    offs.append(total); lens.append(L); total += (L + 31) & ~15
buf = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
for i in range(NC):
    a = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
    buf[offs[i]:offs[i] + L] = lut[a.long()]
    mut = torch.rand((L,), generator=g, device=dev) < 0.01
    b = torch.where(mut, (a + torch.randint(1, 4, (L,), generator=g, device=dev, dtype=torch.uint8)) & 3, a)
    buf[offs[NC + i]:offs[NC + i] + L] = lut[b.long()]
    del a, b, mut
torch.cuda.synchronize()
c_off = (C.c_uint64 * (2 * NC))(*offs); c_len = (C.c_uint64 * (2 * NC))(*lens)
gfc = (C.c_uint32 * 3)(0, NC, 2 * NC)
out = (C.c_void_p * 2)()
t = time.time()
_capi.check(lib.psk_sketch_batch_device(ctx, C.byref(params), C.c_void_p(buf.data_ptr()), c_off, c_len, gfc, 2, 1, out))
print("sketch 2 x %d Mb: %.3f s" % (NC * L // 1000000, time.time() - t), flush=True)
ns, nm, tl, nc = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
for i in range(2):
    _capi.check(lib.psk_sketch_info(out[i], None, C.byref(ns), C.byref(nm), C.byref(tl), C.byref(nc)))
    print(" genome", i, "seeds", ns.value, "markers", nm.value, "len", tl.value, "contigs", nc.value, flush=True)
if not DO_QUERY:
    sys.exit(0)
db = C.c_void_p(); _capi.check(lib.psk_db_create(ctx, C.byref(params), C.byref(db)))
_capi.check(lib.psk_db_add(db, b"ref", out[0]))
opts = _capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None)
hits = C.POINTER(_capi.Hit)(); n = C.c_uint64(0)
_capi.check(lib.psk_ctx_set_timing(ctx, 1))
for rep in range(2):   # the first query also builds the two k-mer indexes
    t = time.time()
    _capi.check(lib.psk_query(db, out[1], C.byref(opts), C.byref(hits), C.byref(n)))
    print("query %d: %.3f s, hits %d" % (rep, time.time() - t, n.value))
for k in ("sketch_sort", "screen", "anchor", "chain_chunk", "select", "pair_reduce"):
    ms, cnt = C.c_double(0), C.c_uint64(0)
    _capi.check(lib.psk_ctx_timing(ctx, k.encode(), C.byref(ms), C.byref(cnt)))
    print("  %-12s %8.3f ms over %d launches" % (k, ms.value, cnt.value))
_capi.check(lib.psk_ctx_set_timing(ctx, 0))
for i in range(n.value):
    h = hits[i]; print(" ani %.5f afq %.4f afr %.4f chunks %d intervals %d anchors %d" % (h.ani, h.af_query, h.af_ref, h.n_chunks, h.n_intervals, h.n_anchors))
opts2 = _capi.QueryOpts(0, 1, 0, 0, 0.0, 0.0, None)
rc = lib.psk_query(db, out[1], C.byref(opts2), C.byref(hits), C.byref(n))
print("median on 50k chunks ->", rc, lib.psk_last_error())
