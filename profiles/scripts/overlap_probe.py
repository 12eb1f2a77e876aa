"""Does running two chaining batches at once (two lanes = two HIP streams) beat running them one after the other?
All-vs-all of 1 000 genomes through psk_query_many: one call for all queries vs two host threads with half the queries each."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
n = 1000
buf, offs, lens = bench.make_genomes(torch, dev, seed_shared=2, seed_members=3, n_refs=n, n_families=10)
torch.cuda.synchronize()
eng = bench.Engine(0)
names = (C.c_char_p * n)(*[f"r{i}".encode() for i in range(n)])
handles = eng.sketch_device(buf.data_ptr(), offs[:n], lens[:n])
db = eng.make_db(names, handles, n)
def run(lo, hi, out, k):
    sub = (C.c_void_p * (hi - lo))(*[handles[i] for i in range(lo, hi)])
    out[k] = eng.query_many(db, sub, hi - lo)
for mode in ("one call", "two threads", "one call", "two threads", "four threads"):
    res = {}
    eng.capi.check(eng.lib.psk_ctx_synchronize(eng.ctx))
    t0 = time.perf_counter()
    if mode == "one call":
        run(0, n, res, 0)
    else:
        k = 2 if mode == "two threads" else 4
        th = [threading.Thread(target=run, args=(i * n // k, (i + 1) * n // k, res, i)) for i in range(k)]
        [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    print(f"{mode:12s} {dt * 1e3:7.1f} ms  hits {sum(res.values())}")
