"""Host-side wall time of each C-ABI call of one bench step (search workload), averaged over steps."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench

def main(steps=10):
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    buf, offs, lens = bench.make_genomes(torch, dev, seed_shared=2, seed_members=3, n_refs=bench.N_REFS, n_families=bench.N_FAMILIES)
    torch.cuda.synchronize()
    eng = bench.Engine(0); lib, capi = eng.lib, eng.capi
    n = len(offs)
    names = (C.c_char_p * bench.N_REFS)(*[f"r{i}".encode() for i in range(bench.N_REFS)])
    acc = {}
    def T(k, f):
        t0 = time.perf_counter(); r = f(); acc[k] = acc.get(k, 0.0) + time.perf_counter() - t0; return r
    for it in range(steps + 2):
        if it == 2: acc.clear()
        c_off = T("py_arrays", lambda: (C.c_uint64 * n)(*offs)); c_len = (C.c_uint64 * n)(*lens); gfc = (C.c_uint32 * (n + 1))(*range(n + 1)); out = (C.c_void_p * n)()
        T("sketch_batch", lambda: capi.check(lib.psk_sketch_batch_device(eng.ctx, C.byref(eng.params), C.c_void_p(buf.data_ptr()), c_off, c_len, gfc, n, 1, out)))
        db = C.c_void_p(); T("db_create", lambda: capi.check(lib.psk_db_create(eng.ctx, C.byref(eng.params), C.byref(db))))
        T("db_add_batch", lambda: capi.check(lib.psk_db_add_batch(db, names, out, n - 1)))
        opts = capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None) if len(capi.QueryOpts._fields_) > 6 else capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0); hits_p = C.POINTER(capi.Hit)(); nh = C.c_uint64(0)
        T("query", lambda: capi.check(lib.psk_query(db, out[n - 1], C.byref(opts), C.byref(hits_p), C.byref(nh))))
        T("free", lambda: (lib.psk_free(hits_p), lib.psk_sketch_free(out[n - 1]), lib.psk_db_destroy(db)))
    print({k: round(v / steps * 1e3, 3) for k, v in acc.items()}, "ms per step; sum", round(sum(acc.values()) / steps * 1e3, 3))

if __name__ == "__main__":
    main()
