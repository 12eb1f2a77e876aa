"""Sketch throughput on metagenome-like input (BASELINE configs[3] shape): many genomes of short contigs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyskani_amd import _capi
lib = _capi.load()
ctx = C.c_void_p(); _capi.check(lib.psk_ctx_create(0, C.byref(ctx)))
params = _capi.Params(30, 200, 15)
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
n_gen, n_ctg = 400, 100
lens = np.exp(rng.uniform(np.log(2000), np.log(50000), size=n_gen * n_ctg)).astype(np.int64)
offs = np.concatenate([[0], np.cumsum((lens + 31) & ~15)])[:-1]
total = int(offs[-1] + lens[-1] + 64)
buf = torch.randint(0, 4, (total,), device=dev, dtype=torch.uint8)
buf = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[buf.long()]
torch.cuda.synchronize()
c_off = (C.c_uint64 * len(offs))(*[int(x) for x in offs]); c_len = (C.c_uint64 * len(lens))(*[int(x) for x in lens])
gfc = (C.c_uint32 * (n_gen + 1))(*[g * n_ctg for g in range(n_gen + 1)])
out = (C.c_void_p * n_gen)()
for it in range(4):
    if it == 1:
        _capi.check(lib.psk_ctx_set_timing(ctx, 1))
    t = time.perf_counter()
    _capi.check(lib.psk_sketch_batch_device(ctx, C.byref(params), C.c_void_p(buf.data_ptr()), c_off, c_len, gfc, n_gen, 1, out))
    dt = time.perf_counter() - t
    for i in range(n_gen): lib.psk_sketch_free(out[i])
    print(f"{n_gen} genomes x {n_ctg} contigs, {lens.sum()/1e6:.0f} Mb: {dt*1e3:.2f} ms = {lens.sum()/dt/1e12:.3f} T bases/s", flush=True)
for k in ("sketch_scan", "sketch_emit", "sketch_sort"):
    ms, cnt = C.c_double(0), C.c_uint64(0)
    _capi.check(lib.psk_ctx_timing(ctx, k.encode(), C.byref(ms), C.byref(cnt)))
    print(f"  {k:12s} {ms.value / max(1, cnt.value):8.3f} ms per launch ({cnt.value} launches)")
