"""Does the FIRST sketch + query of Gb-scale genomes in a context give the hits of the later ones? (fresh context; 8 x 3 Gb or argv[1] genomes of argv[2] Mb contigs)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
dev = torch.device("cuda:0")
eng = B.Engine(0)
g = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 125
buf, offs, lens, gfc_l = B.make_big_genomes(torch, dev, g, 24, mb * 1_000_000, 4, seed=5)
torch.cuda.synchronize()
names = (C.c_char_p * g)(*[f"m{i}".encode() for i in range(g)])
c_off, c_len, gfc, nn = eng.layout(offs, lens, gfc_l)
for rep in range(3):
    handles = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
    db = eng.make_db(names, handles, nn)
    t0 = time.perf_counter()
    nh = eng.query_many(db, handles, nn)
    print("rep", rep, "query", round(1e3 * (time.perf_counter() - t0), 1), "ms, hits", nh, flush=True)
    eng.lib.psk_db_destroy(db)
