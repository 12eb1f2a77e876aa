import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, bench
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
buf, offs, lens = bench.make_genomes(torch, dev, seed_shared=2, seed_members=3, n_refs=300, n_families=10)
torch.cuda.synchronize()
eng = bench.Engine(0)
names = (C.c_char_p * 300)(*[f"r{i}".encode() for i in range(300)])
def free_mb(): return torch.cuda.mem_get_info()[0] / 2**20
for it in range(6):
    t0 = time.perf_counter()
    for _ in range(40):
        eng.step(buf.data_ptr(), offs, lens, names)
        eng.step_all_vs_all(buf.data_ptr(), offs[:101], lens[:101], names) if False else None
    print(f"round {it}: {(time.perf_counter()-t0)/40*1e3:.3f} ms/step, free {free_mb():.0f} MiB", flush=True)
