"""The contract job's steps WITHOUT the bracket timers (psk_ctx_set_timing never called), one by one: are the slow second / third steps of bench.py's timed loop the timers'?
python profiles/scripts/r6_step_times_plain.py [timing]   ("timing": switch the timers on after four steps, as bench.py's timed loop does)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
dev = torch.device("cuda:0")
eng = B.Engine(0)
n = 10000
anc_lens, fam_of = B.family_layout(3, n, n // 100)
buf, offs, lens = B.make_genomes(torch, dev, 3, 31, list(range(n)), fam_of, anc_lens, variant="plain")
torch.cuda.synchronize()
names = (C.c_char_p * n)(*[f"g{i}".encode() for i in range(n)])
c_off, c_len, gfc, nn = eng.layout(offs, lens, None)
ts = []
for s in range(14):
    if len(sys.argv) > 1 and s == 4:
        print("clock", eng.clock_probe()["shader_clock_mhz"]); eng.capi.check(eng.lib.psk_ctx_set_timing(eng.ctx, 1)); eng.timing("reset"); eng.work(reset=True)
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
    db = eng.make_db(names, out, nn)
    nh = eng.query_many(db, out, nn)
    eng.lib.psk_db_destroy(db)
    ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print(sys.argv[1:], ts)
