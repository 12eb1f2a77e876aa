#!/usr/bin/env python3
"""bench.py — genome-pairs/sec (sketch + ANI) of the MI355X-native pyskani hot path.

One "step" = one pass of the whole hot path over one batch of synthetic genomes that are already
resident in HBM as ASCII: sketch every reference and the query (FracMinHash seeds, marker sets,
k-mer index), load them into a fresh database, run Database.query (marker screen -> seed-index
lookup -> chaining -> ANI/AF) and bring the hit list back to the host. Nothing is cached between
steps. At N=1 the workload is BASELINE.json configs[1]: 1 query vs 1 000 synthetic ~5 Mb refs,
c=125, marker_c=1000, k=15. For N>1 the references are sharded: every rank holds its own 1 000
references (weak scaling), the query is replicated, and the per-shard hit lists are all-gathered
with RCCL (torch.distributed backend "nccl").

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     — dominant kernel (sketch_scan) timed with HIP events on the library's stream
  cpu_baseline — the CPU oracle (oracle/, a port) on a bounded sample of the same workload
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_REFS = 1000            # per GPU
N_FAMILIES = 10
DIVERGENCE = (0.0005, 0.002, 0.005, 0.01, 0.02, 0.04, 0.07, 0.10)   # SURVEY.md §8(d) family model
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def make_genomes(torch, device, seed_shared, seed_members, n_refs, n_families):
    """Family model of SURVEY.md §8(d): ancestors of iid ACGT, L ~ U[4.5, 5.5] Mb (shared by all
    ranks); members carry independent substitutions at the cycled rates above (rank-specific);
    the last genome is the query (family 0, d = 0.02, shared). Built on the GPU; returns one uint8
    ASCII tensor plus per-genome (offset, length), every offset 16-byte aligned."""
    gs = torch.Generator(device=device)
    gs.manual_seed(seed_shared)
    gm = torch.Generator(device=device)
    gm.manual_seed(seed_members)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    rng = np.random.default_rng(seed_shared)
    lens = [int(rng.integers(4_500_000, 5_500_001)) for _ in range(n_families)]
    per_fam = max(1, n_refs // n_families)
    n_genomes = n_refs + 1
    offs, glen, total = [], [], 0
    fam_of = [min(i // per_fam, n_families - 1) for i in range(n_refs)] + [0]
    for i in range(n_genomes):
        offs.append(total)
        glen.append(lens[fam_of[i]])
        total += (lens[fam_of[i]] + 15 + 16) & ~15
    buf = torch.zeros(total + 64, dtype=torch.uint8, device=device)
    anc = [torch.randint(0, 4, (L,), generator=gs, device=device, dtype=torch.uint8) for L in lens]
    for i in range(n_genomes):
        a = anc[fam_of[i]]
        g = gs if i == n_refs else gm
        d = 0.02 if i == n_refs else DIVERGENCE[i % len(DIVERGENCE)]
        mut = torch.rand(a.shape, generator=g, device=device) < d
        shift = torch.randint(1, 4, a.shape, generator=g, device=device, dtype=torch.uint8)
        codes = torch.where(mut, (a + shift) & 3, a)
        buf[offs[i]:offs[i] + glen[i]] = lut[codes.long()]
        del mut, shift, codes
    return buf, offs, glen


class Engine:
    """Thin driver over the C-ABI for device-resident genomes."""

    def __init__(self, device):
        from pyskani_amd import _capi
        self.capi = _capi
        self.lib = _capi.load()
        self.ctx = C.c_void_p()
        _capi.check(self.lib.psk_ctx_create(device, C.byref(self.ctx)))
        self.params = _capi.Params(125, 1000, 15)

    def _layout(self, offs, lens, n):
        """ctypes views of the (constant) genome layout, built once: they describe the resident input"""
        key = (id(offs), n)
        if getattr(self, "_layout_key", None) != key:
            self._layout_key = key
            self._c_off = (C.c_uint64 * n)(*offs[:n]); self._c_len = (C.c_uint64 * n)(*lens[:n])
            self._gfc = (C.c_uint32 * (n + 1))(*range(n + 1))
        return self._c_off, self._c_len, self._gfc

    def step(self, d_ptr, offs, lens, names):
        lib, capi = self.lib, self.capi
        n = len(offs)
        c_off, c_len, gfc = self._layout(offs, lens, n)
        out = (C.c_void_p * n)()
        capi.check(lib.psk_sketch_batch_device(self.ctx, C.byref(self.params), C.c_void_p(d_ptr), c_off, c_len, gfc, n, 1, out))
        db = C.c_void_p()
        capi.check(lib.psk_db_create(self.ctx, C.byref(self.params), C.byref(db)))
        try:
            capi.check(lib.psk_db_add_batch(db, names, out, n - 1))
            opts = capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None)
            hits_p = C.POINTER(capi.Hit)()
            nh = C.c_uint64(0)
            capi.check(lib.psk_query(db, out[n - 1], C.byref(opts), C.byref(hits_p), C.byref(nh)))
            hits = np.zeros((nh.value, 4), dtype=np.float32)
            if nh.value:   # one structured view of the returned psk_hit array instead of a ctypes loop
                rec = np.frombuffer((capi.Hit * nh.value).from_address(C.addressof(hits_p.contents)), dtype=np.dtype(capi.Hit))
                hits[:, 0] = rec["ref_index"]; hits[:, 1] = rec["ani"]; hits[:, 2] = rec["af_query"]; hits[:, 3] = rec["af_ref"]
            if hits_p:
                lib.psk_free(hits_p)
        finally:
            lib.psk_sketch_free(out[n - 1])
            lib.psk_db_destroy(db)
        return hits

    def step_all_vs_all(self, d_ptr, offs, lens, names):
        """BASELINE configs[2] shape on one GPU: every genome against a database of all of them."""
        lib, capi = self.lib, self.capi
        n = len(offs) - 1                      # the trailing query genome is not used here
        c_off, c_len, gfc = self._layout(offs, lens, n)
        out = (C.c_void_p * n)()
        capi.check(lib.psk_sketch_batch_device(self.ctx, C.byref(self.params), C.c_void_p(d_ptr), c_off, c_len, gfc, n, 1, out))
        db = C.c_void_p()
        capi.check(lib.psk_db_create(self.ctx, C.byref(self.params), C.byref(db)))
        try:
            capi.check(lib.psk_db_add_batch(db, names, out, n))
            opts = capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None)
            hits_p = C.POINTER(capi.Hit)()
            offsets = (C.c_uint64 * (n + 1))()
            capi.check(lib.psk_query_many(db, out, n, C.byref(opts), C.byref(hits_p), offsets))
            nh = int(offsets[n])
            if hits_p:
                lib.psk_free(hits_p)
        finally:
            lib.psk_db_destroy(db)
        return nh

    def timing(self, kernel):
        ms, n = C.c_double(0), C.c_uint64(0)
        self.capi.check(self.lib.psk_ctx_timing(self.ctx, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def cpu_baseline(fetch, n_sample):
    """The CPU oracle (a port of the restated skani path, single thread like one pyskani call) on
    1 query vs the first n_sample references of this rank's shard. `fetch(i)` returns genome i's bytes
    (i = -1: the query); only oracle time is counted, not the D2H copies that feed it."""
    from oracle import oracle as O
    O.build()
    secs = 0.0
    g = fetch(-1)
    t0 = time.perf_counter(); q = O.Sketch([g]); secs += time.perf_counter() - t0
    refs = []
    for i in range(n_sample):
        g = fetch(i)
        t0 = time.perf_counter(); refs.append((str(i), O.Sketch([g]))); secs += time.perf_counter() - t0
    del g
    t0 = time.perf_counter(); hits = O.query(refs, q); secs += time.perf_counter() - t0
    return n_sample / secs, secs, len(hits)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--refs", type=int, default=N_REFS, help="references per GPU (BASELINE configs[1]: 1000)")
    ap.add_argument("--workload", choices=["search", "allvsall"], default="search",
                    help="search = BASELINE configs[1] (the headline); allvsall = configs[2] shape on this GPU's genomes (extra, not the headline)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only for dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry-run aid: every rank uses device 0 (needs --backend gloo); never for reported numbers")
    ap.add_argument("--cpu-sample", type=int, default=1000, help="references in the CPU-baseline sample (0 = skip); 1000 = the whole workload, ~10-20 s")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through `python -m torch.distributed.run "
              f"--nproc-per-node {args.gpus} ...`; measuring {world} GPU(s)", file=sys.stderr)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if args.backend == "nccl" else "cpu"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    n_refs = args.refs
    buf, offs, lens = make_genomes(torch, device, seed_shared=2, seed_members=1000 * rank + 3, n_refs=n_refs, n_families=N_FAMILIES)
    torch.cuda.synchronize()

    from pyskani_amd.parallel import all_gather_hits
    eng = Engine(local_rank)
    names = (C.c_char_p * n_refs)(*[f"r{rank}_{i}".encode() for i in range(n_refs)])

    def step():
        if args.workload == "allvsall":
            return eng.step_all_vs_all(buf.data_ptr(), offs, lens, names)
        hits = eng.step(buf.data_ptr(), offs, lens, names)
        if world > 1:   # exchange step: all-gather of per-shard hit lists (RCCL over xGMI)
            idx = np.stack([np.zeros(len(hits), np.int64), hits[:, 0].astype(np.int64) + rank * n_refs], axis=1)   # (query, GLOBAL ref index)
            return all_gather_hits(idx, hits[:, 1:4], dist, device=coll_device)[0].shape[0]
        return hits.shape[0]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.capi.check(eng.lib.psk_ctx_synchronize(eng.ctx))

    for _ in range(args.warmup):
        n_hits = step()
    eng.capi.check(eng.lib.psk_ctx_set_timing(eng.ctx, 1))
    eng.timing("reset")
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_hits = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=coll_device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    scan_ms, scan_n = eng.timing("sketch_scan")
    kernel_ms = {k: eng.timing(k)[0] / max(1, args.steps) for k in ("sketch_scan", "sketch_emit", "sketch_sort", "screen", "anchor", "chain_chunk", "select", "pair_reduce")}
    eng.capi.check(eng.lib.psk_ctx_set_timing(eng.ctx, 0))

    if rank == 0:
        pairs = (n_refs * n_refs if args.workload == "allvsall" else n_refs) * world * args.steps
        value = pairs / dt
        # sketch_scan: ALGORITHMIC bytes per launch = sum over the launch's genomes of
        # L (ASCII read) + L/4 (2-bit packed write)   [SURVEY.md §8(d) B_sk, first two terms; DESIGN.md §4]
        bases = float(sum(lens))
        alg_bytes = bases * 1.25 * args.steps / max(1, scan_n)   # per launch (a step may split into sub-batches)
        avg_s = (scan_ms / max(1, scan_n)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_sketch_scan.json")
        if os.path.exists(pmc):   # PMC pass is offline (rocprofv3 --pmc, separate runs); scaled by this launch's bases
            traffic = json.load(open(pmc))["traffic_bytes_per_base"] * bases * args.steps / max(1, scan_n)
        line = {
            "metric": "genome-pairs/sec (sketch+ANI)", "value": value, "unit": "genome-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": (f"all-vs-all {n_refs} x {n_refs} synthetic ~5 Mb genomes per GPU" if args.workload == "allvsall" else f"1 query vs {n_refs} synthetic ~5 Mb refs per GPU") + f" (10 families x {n_refs // N_FAMILIES}), c=125 marker_c=1000 k=15",
                       "refs_per_gpu": n_refs, "hits": int(n_hits), "parallelism": f"refs sharded over {world} GPU(s)"},
            "roofline": {"kernel": "sketch_scan_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_s * 1e3, "launches": int(scan_n),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "priced against HBM as the contract asks; the kernel is integer-VALU-issue bound (one 64-bit mix per base: "
                                 "15 four-cycle + 7 two-cycle wave instructions per base, DESIGN.md section 4, profiles/micro/valu_rates.hip)"},
            "kernel_ms_per_step": kernel_ms,
            # SURVEY.md §8(d) side figures, whole job
            "extras": {"genomes_sketched_per_s": (n_refs + (0 if args.workload == "allvsall" else 1)) * world * args.steps / dt,
                       "bases_sketched_per_s": bases * world * args.steps / dt,
                       "reported_hits_per_step": int(n_hits)},
        }
        if args.cpu_sample > 0 and world == 1 and args.workload == "search":
            ns = min(args.cpu_sample, n_refs)
            v, secs, nh = cpu_baseline(lambda i: bytes(buf[offs[i]:offs[i] + lens[i]].cpu().numpy()), ns)
            line["cpu_baseline"] = {"value": v, "unit": "genome-pairs/s", "cores": 1, "kind": "port",
                                    "sample": f"1 query vs the first {ns} of {n_refs} refs of the same workload (sketch {ns + 1} genomes + {ns} screens + {nh} chained hits), {secs:.1f} s of oracle time"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
