import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
buf, offs, lens = bench.make_genomes(torch, dev, seed_shared=2, seed_members=3, n_refs=200, n_families=10)
torch.cuda.synchronize()
eng = bench.Engine(0); lib, capi = eng.lib, eng.capi
names = (C.c_char_p * 200)(*[f"r{i}".encode() for i in range(200)])
for it in range(3):
    if it == 1:
        capi.check(lib.psk_ctx_set_timing(eng.ctx, 1)); eng.timing("reset")
    eng.step_all_vs_all(buf.data_ptr(), offs, lens, names)
print(os.environ.get("PSK_LIB_PATH"), "sketch_sort ms per step:", eng.timing("sketch_sort")[0] / 2)
