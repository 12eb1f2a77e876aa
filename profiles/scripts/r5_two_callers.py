"""Would two batches in flight help the all-vs-all step? One psk_query_many call over all queries against two concurrent callers (two host threads, two lanes)
with half of the queries each, same database, indexes already built. python profiles/scripts/r5_two_callers.py [n_genomes] [callers]"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B

n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda:0")
anc_lens, fam_of = B.family_layout(3, n_total, max(1, n_total // 100))
buf, offs, lens = B.make_genomes(torch, dev, 3, 31, list(range(n_total)), fam_of, anc_lens, variant="plain")
torch.cuda.synchronize()
eng = B.Engine(0)
names = (C.c_char_p * n_total)(*[f"g{i}".encode() for i in range(n_total)])
c_off, c_len, gfc, n = eng.layout(offs, lens, None)
out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, n)
db = eng.make_db(names, out, n)
n_callers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cuts = [n * i // n_callers for i in range(n_callers + 1)]
parts = [((type(out)._type_ * (cuts[i + 1] - cuts[i]))(*out[cuts[i]:cuts[i + 1]]), cuts[i + 1] - cuts[i]) for i in range(n_callers)]
print("warm", eng.query_many(db, out, n), flush=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nh = eng.query_many(db, out, n)
    t1 = time.perf_counter()
    res = [0] * n_callers
    def run(i, h, k): res[i] = eng.query_many(db, h, k)
    th = [threading.Thread(target=run, args=(i, parts[i][0], parts[i][1])) for i in range(n_callers)]
    t2 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t3 = time.perf_counter()
    print(f"one caller: {1e3 * (t1 - t0):.1f} ms ({nh} hits)   {n_callers} callers: {1e3 * (t3 - t2):.1f} ms ({sum(res)} hits)", flush=True)
