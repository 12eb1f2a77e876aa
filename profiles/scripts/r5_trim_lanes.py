"""One context, an all-vs-all of 4 000 x 5 Mb genomes (two lanes, tens of GB of chain scratch each), then 8 x 3 Gb genomes all-vs-all: the Gb-scale rounds size their batches by the
free memory and take back what the idle lane holds. PSK_TRACE_BATCH=1 python profiles/scripts/r5_trim_lanes.py  (stderr: pairs and items of every batch)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B

dev = torch.device("cuda:0")
eng = B.Engine(0)
def free_gb(): return torch.cuda.mem_get_info()[0] / 2**30
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 125
anc_lens, fam_of = B.family_layout(3, n, n // 100)
buf, offs, lens = B.make_genomes(torch, dev, 3, 31, list(range(n)), fam_of, anc_lens, variant="plain")
torch.cuda.synchronize()
names = (C.c_char_p * n)(*[f"g{i}".encode() for i in range(n)])
c_off, c_len, gfc, nn = eng.layout(offs, lens, None)
out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
db = eng.make_db(names, out, nn)
print("all-vs-all hits", eng.query_many(db, out, nn), "free GiB", round(free_gb(), 1), flush=True)
eng.lib.psk_db_destroy(db); del buf; torch.cuda.empty_cache()
print("after the all-vs-all: free GiB", round(free_gb(), 1), flush=True)
g = 8
buf, offs, lens, gfc_l = B.make_big_genomes(torch, dev, g, 24, mb * 1_000_000, 4, seed=5)
torch.cuda.synchronize()      # (the generator runs on torch's stream, the library on its own)
names = (C.c_char_p * g)(*[f"m{i}".encode() for i in range(g)])
c_off, c_len, gfc, nn = eng.layout(offs, lens, gfc_l)
for rep in range(3):
    handles = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
    db = eng.make_db(names, handles, nn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nh = eng.query_many(db, handles, nn)
    print("8 x 3 Gb query", round(1e3 * (time.perf_counter() - t0), 1), "ms, hits", nh, "free GiB", round(free_gb(), 1), flush=True)
    print("   same sketches, same database, again:", eng.query_many(db, handles, nn), flush=True)
    db2 = eng.make_db(names, handles, nn)
    print("   same sketches, new database:", eng.query_many(db2, handles, nn), flush=True)
    eng.lib.psk_db_destroy(db2)
    eng.lib.psk_db_destroy(db)
