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
  cpu_baseline — the CPU oracle (oracle/, a port) on a bounded sample of the same workload: one core (what a
                 pyskani call uses, lib.rs:493,569) and all host cores (one query / reference per thread)
  extras       — at N=1: the same workload through the pyskani-shaped API from HOST memory
                 (api_pairs_per_s: 1 000 x Database.sketch(bytes) + Database.query(bytes);
                  host_ascii_pairs_per_s: Database.sketch_many + query, the pipelined ingest)

Other workloads (not the headline; `--workload`):
  allvsall    BASELINE configs[2] shape on one GPU: every genome against all (families of 100)
  metagenome  BASELINE configs[3] shape: short contigs (2-50 kb) against a resident database of 5 Mb references,
              c=30 marker_c=200
  mammalian   BASELINE configs[4] shape at reduced count: all-vs-all of --refs genomes of 24 x --contig-mb Mb contigs
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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

    def set_params(self, c, marker_c, k=15):
        self.params = self.capi.Params(c, marker_c, k)

    def sketch_device(self, d_ptr, offs, lens):
        n = len(offs)
        key = (id(offs), n)       # ctypes views of a constant layout are built once (they describe the resident input)
        if getattr(self, "_sd_key", None) != key:
            self._sd_key = key
            self._sd = ((C.c_uint64 * n)(*offs), (C.c_uint64 * n)(*lens), (C.c_uint32 * (n + 1))(*range(n + 1)))
        c_off, c_len, gfc = self._sd
        out = (C.c_void_p * n)()
        self.capi.check(self.lib.psk_sketch_batch_device(self.ctx, C.byref(self.params), C.c_void_p(d_ptr), c_off, c_len, gfc, n, 1, out))
        return out

    def sketch_device_contigs(self, d_ptr, c_offs, c_lens, gfc):
        n, nc = len(gfc) - 1, len(c_offs)
        c_off = (C.c_uint64 * nc)(*c_offs); c_len = (C.c_uint64 * nc)(*c_lens); g = (C.c_uint32 * (n + 1))(*gfc)
        out = (C.c_void_p * n)()
        self.capi.check(self.lib.psk_sketch_batch_device(self.ctx, C.byref(self.params), C.c_void_p(d_ptr), c_off, c_len, g, n, 1, out))
        return out

    def make_db(self, names, handles, n):
        db = C.c_void_p()
        self.capi.check(self.lib.psk_db_create(self.ctx, C.byref(self.params), C.byref(db)))
        self.capi.check(self.lib.psk_db_add_batch(db, names, handles, n))
        return db

    def query_many(self, db, handles, n, faster_small=False):
        opts = self.capi.QueryOpts(0, 0, 0, int(faster_small), 0.0, 0.0, None)
        hits_p = C.POINTER(self.capi.Hit)()
        offsets = (C.c_uint64 * (n + 1))()
        self.capi.check(self.lib.psk_query_many(db, handles, n, C.byref(opts), C.byref(hits_p), offsets))
        nh = int(offsets[n])
        if hits_p:
            self.lib.psk_free(hits_p)
        return nh

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
            if nh:      # algorithmic work of the chain stage, for the per-kernel roofline figures: anchors and (pair, query seed) items of the hits
                rec = np.frombuffer((capi.Hit * nh).from_address(C.addressof(hits_p.contents)), dtype=np.dtype(capi.Hit))
                per_q = np.diff(np.frombuffer(offsets, dtype=np.uint64).astype(np.int64))
                seeds = np.zeros(n, np.int64)
                ns = C.c_uint64()
                for i in range(n):
                    capi.check(lib.psk_sketch_info(out[i], None, C.byref(ns), None, None, None)); seeds[i] = ns.value
                self.last_chain_work = {"anchors": int(rec["n_anchors"].sum()), "items": int((per_q * seeds).sum()), "pairs": nh}
            if hits_p:
                lib.psk_free(hits_p)
        finally:
            lib.psk_db_destroy(db)
        return nh

    def timing(self, kernel):
        ms, n = C.c_double(0), C.c_uint64(0)
        self.capi.check(self.lib.psk_ctx_timing(self.ctx, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def cpu_baseline(fetch, n_sample, threads):
    """The CPU oracle (a port of the restated skani path) on 1 query vs the first n_sample references of this rank's
    shard. `fetch(i)` returns genome i's bytes (i = -1: the query); only oracle time is counted, not the D2H copies
    that feed it. Two figures: ONE core (a pyskani call is single-threaded, lib.rs:493,569) and ALL host cores with one
    reference per thread (what a user gets from the GIL release) — ctypes drops the GIL, so plain threads scale."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    genomes = [fetch(i) for i in range(n_sample)]
    gq = fetch(-1)
    t0 = time.perf_counter()
    q = O.Sketch([gq])
    refs = [(str(i), O.Sketch([g])) for i, g in enumerate(genomes)]
    hits = O.query(refs, q)
    secs1 = time.perf_counter() - t0
    del refs

    def one(i):
        r = O.Sketch([genomes[i]])
        return O.query_count([r], q)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        nh = sum(ex.map(one, range(n_sample)))
    secs_all = time.perf_counter() - t0
    assert nh == len(hits)
    return n_sample / secs1, secs1, len(hits), n_sample / secs_all, secs_all


def cpu_baseline_allvsall(fetch, n_refs, n_queries, threads):
    """CPU oracle on a SUB-SAMPLE of the all-vs-all workload (sub-sample configs 3-5 and extrapolate
    linearly in pairs): every reference is sketched once (all cores), then n_queries of them are queried against all
    n_refs (screen every reference, chain the shortlist). One-core figure from the first few queries alone."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        sk = list(ex.map(lambda i: O.Sketch([fetch(i)]), range(n_refs)))
    t_sketch_all = time.perf_counter() - t0
    step = max(1, n_refs // n_queries)
    qs = list(range(0, n_refs, step))[:n_queries]

    def one(qi):      # screen every reference + chain the shortlist, one C call (the interpreter lock is released inside)
        return O.query_count(sk, sk[qi])
    n1 = min(4, len(qs))
    t0 = time.perf_counter()
    h1 = [one(q) for q in qs[:n1]]
    t_one = (time.perf_counter() - t0) / n1                     # seconds per query on one core
    t0 = time.perf_counter()
    g = fetch(0); O.Sketch([g]); t_sk1 = time.perf_counter() - t0  # seconds per sketch on one core
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        hall = list(ex.map(one, qs))
    t_q_all = time.perf_counter() - t0
    # extrapolation to the full n_refs x n_refs job
    secs_1core = n_refs * t_sk1 + n_refs * t_one
    secs_all = t_sketch_all + t_q_all * (n_refs / len(qs))
    pairs = float(n_refs) * n_refs
    return {"value": pairs / secs_1core, "unit": "genome-pairs/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"{len(qs)} of {n_refs} queries against all {n_refs} references ({sum(hall)} chained hits), every reference sketched once; "
                      f"extrapolated linearly in queries to {n_refs} x {n_refs}; one core: {t_sk1 * 1e3:.1f} ms per sketch, {t_one:.2f} s per query ({n1} queries timed)",
            "all_cores": {"value": pairs / secs_all, "cores": threads, "seconds_extrapolated": secs_all,
                          "measured": {"sketch_all_refs_s": t_sketch_all, "queries_s": t_q_all, "queries": len(qs)}},
            "note": "the repo's own C restatement (the Rust reference cannot be built here); flat arrays where skani uses hash maps: a conservative floor for the speed-up"}


def cpu_baseline_metagenome(fetch_ref, n_cpu_refs, contigs, threads, faster_small):
    """CPU oracle on a bounded sample of the metagenome workload: a database of the first n_cpu_refs references (sketched with all
    cores, untimed like the GPU side's resident database), then every sampled contig as its own query (sketch + screen of every
    reference + chaining of the shortlist): one core for the first few, all cores (one contig per thread) for all of them."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        sk = list(ex.map(lambda i: O.Sketch([fetch_ref(i)], c=30, marker_c=200), range(n_cpu_refs)))
    t_db = time.perf_counter() - t0

    def one(c):
        return O.query_count(sk, O.Sketch([c], c=30, marker_c=200), faster_small=faster_small)
    n1 = min(16, len(contigs))
    t0 = time.perf_counter()
    h1 = [one(c) for c in contigs[:n1]]
    t_one = (time.perf_counter() - t0) / n1
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        hall = list(ex.map(one, contigs))
    t_all = time.perf_counter() - t0
    return {"value": 1.0 / t_one, "unit": "queries/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"{len(contigs)} of the contigs, each its own query against a database of the first {n_cpu_refs} references only (a tenth of the GPU run's 5 000: "
                      f"the CPU figure is flattered by it); {sum(hall)} hits; one core: {t_one * 1e3:.1f} ms per query ({n1} timed); database sketched in {t_db:.1f} s on {threads} threads (not counted)",
            "all_cores": {"value": len(contigs) / t_all, "cores": threads, "seconds": t_all},
            "note": "the repo's own C restatement (the Rust reference cannot be built here)"}


def api_rates(psk, genomes, query):
    """The drop-in path a pyskani user calls, from ASCII in HOST memory (SURVEY.md §8d 'Metric'):
    (a) n x Database.sketch(name, bytes) + one Database.query(name, bytes); (b) Database.sketch_many + query."""
    out = {}
    n = len(genomes)
    for label, bulk in (("api", False), ("host_ascii", True), ("api", False), ("host_ascii", True)):   # second pass = warm
        db = psk.Database()
        t0 = time.perf_counter()
        if bulk:
            db.sketch_many([(f"r{i}", g) for i, g in enumerate(genomes)])
        else:
            for i, g in enumerate(genomes):
                db.sketch(f"r{i}", g)
        t1 = time.perf_counter()
        hits = db.query("q", query, learned_ani=False)
        t2 = time.perf_counter()
        out[label] = {"pairs_per_s": n / (t2 - t0), "sketch_s": t1 - t0, "query_ms": (t2 - t1) * 1e3, "hits": len(hits),
                      "host_GBps": sum(len(g) for g in genomes) / (t1 - t0) / 1e9}
        del db
    return out


def make_big_genomes(torch, device, n_genomes, n_contigs, contig_len, fam_size, seed):
    """BASELINE configs[4] shape: genomes of n_contigs x contig_len bases (24 x 125 Mb = 3 Gb), families of fam_size members that
    carry independent substitutions (rates cycled from DIVERGENCE[:4]) on a shared ancestor. Built contig by contig on the GPU
    (never more than one ancestor contig live). Returns the ASCII buffer, per-contig offsets / lengths and genome_first_contig."""
    g = torch.Generator(device=device); g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    stride = (contig_len + 15 + 16) & ~15
    buf = torch.zeros(n_genomes * n_contigs * stride + 64, dtype=torch.uint8, device=device)
    offs = [(gi * n_contigs + c) * stride for gi in range(n_genomes) for c in range(n_contigs)]
    lens = [contig_len] * (n_genomes * n_contigs)
    n_fam = (n_genomes + fam_size - 1) // fam_size
    for f in range(n_fam):
        for c in range(n_contigs):
            a = torch.randint(0, 4, (contig_len,), generator=g, device=device, dtype=torch.uint8)
            for j in range(fam_size):
                gi = f * fam_size + j
                if gi >= n_genomes:
                    break
                d = DIVERGENCE[:4][j % 4]
                mut = torch.rand(a.shape, generator=g, device=device) < d
                shift = torch.randint(1, 4, a.shape, generator=g, device=device, dtype=torch.uint8)
                o = offs[gi * n_contigs + c]
                buf[o:o + contig_len] = lut[torch.where(mut, (a + shift) & 3, a).long()]
                del mut, shift
            del a
    gfc = [gi * n_contigs for gi in range(n_genomes + 1)]
    return buf, offs, lens, gfc


def make_contigs(torch, device, buf, offs, lens, n_refs, n_contigs, seed):
    """BASELINE configs[3] / SURVEY.md §8(d) config 4: query contigs = substrings of random references, length
    log-uniform in [2 kb, 50 kb], extra divergence U[0, 5 %]. Built on the GPU; 16-byte aligned offsets."""
    rng = np.random.default_rng(seed)
    clen = np.exp(rng.uniform(np.log(2000), np.log(50000), n_contigs)).astype(np.int64)
    src = rng.integers(0, n_refs, n_contigs)
    div = rng.uniform(0, 0.05, n_contigs)
    coffs, total = [], 0
    for L in clen:
        coffs.append(total); total += (int(L) + 15 + 16) & ~15
    out = torch.zeros(total + 64, dtype=torch.uint8, device=device)
    g = torch.Generator(device=device); g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    inv = torch.zeros(256, dtype=torch.uint8, device=device); inv[list(b"ACGT")] = torch.arange(4, dtype=torch.uint8, device=device)
    for i in range(n_contigs):
        L = int(clen[i]); r = int(src[i]); st = int(rng.integers(0, lens[r] - L))
        piece = inv[buf[offs[r] + st: offs[r] + st + L].long()]
        mut = torch.rand(L, generator=g, device=device) < float(div[i])
        shift = torch.randint(1, 4, (L,), generator=g, device=device, dtype=torch.uint8)
        out[coffs[i]:coffs[i] + L] = lut[torch.where(mut, (piece + shift) & 3, piece).long()]
    return out, coffs, [int(x) for x in clen]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--refs", type=int, default=None, help="references per GPU (search / allvsall default 1000 = BASELINE configs[1]; metagenome default 5000)")
    ap.add_argument("--workload", choices=["search", "allvsall", "metagenome", "mammalian"], default="search",
                    help="search = BASELINE configs[1] (the headline); allvsall = configs[2] shape on this GPU's genomes; metagenome = configs[3] shape (extras, not the headline)")
    ap.add_argument("--queries", type=int, default=10000, help="metagenome: number of query contigs")
    ap.add_argument("--contig-mb", type=int, default=125, help="mammalian: contig length in Mb (24 contigs per genome; 125 = 3 Gb genomes)")
    ap.add_argument("--api-queries", type=int, default=2000, help="metagenome: contigs also sent one by one through Database.query() from host bytes (0 = skip)")
    ap.add_argument("--faster-small", action="store_true", help="metagenome: Database.query(faster_small=True) (no rescue of contigs with < 20 markers)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only for dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry-run aid: every rank uses device 0 (needs --backend gloo); never for reported numbers")
    ap.add_argument("--cpu-sample", type=int, default=1000, help="references in the CPU-baseline sample (0 = skip); 1000 = the whole workload, ~10-20 s")
    ap.add_argument("--no-api", action="store_true", help="skip the host-memory API extras (N=1 search only)")
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

    n_refs = args.refs if args.refs is not None else (5000 if args.workload == "metagenome" else (8 if args.workload == "mammalian" else N_REFS))
    n_families = N_FAMILIES if args.workload == "search" else max(1, n_refs // 100)     # families of 100 for configs 3-4
    big = None
    if args.workload == "mammalian":      # BASELINE configs[4] shape at reduced count: n_refs genomes of 24 contigs, families of 4, all-vs-all
        big = make_big_genomes(torch, device, n_refs, 24, args.contig_mb * 1_000_000, 4, seed=5 + rank)
        buf, offs, lens = big[0], big[1], big[2]
    else:
        buf, offs, lens = make_genomes(torch, device, seed_shared=2, seed_members=1000 * rank + 3, n_refs=n_refs, n_families=n_families)
    torch.cuda.synchronize()

    from pyskani_amd.parallel import all_gather_hits
    eng = Engine(local_rank)
    names = (C.c_char_p * n_refs)(*[f"r{rank}_{i}".encode() for i in range(n_refs)])
    meta_state = {}
    if args.workload == "metagenome":
        eng.set_params(30, 200)
        t0 = time.perf_counter()
        handles = eng.sketch_device(buf.data_ptr(), offs[:n_refs], lens[:n_refs])
        meta_state["db"] = eng.make_db(names, handles, n_refs)
        eng.capi.check(eng.lib.psk_ctx_synchronize(eng.ctx))
        meta_state["db_build_s"] = time.perf_counter() - t0
        cbuf, coffs, clens = make_contigs(torch, device, buf, offs, lens, n_refs, args.queries, seed=4 + rank)
        torch.cuda.synchronize()
        meta_state.update(cbuf=cbuf, coffs=coffs, clens=clens)

    def step():
        if args.workload == "allvsall":
            return eng.step_all_vs_all(buf.data_ptr(), offs, lens, names)
        if args.workload == "mammalian":     # sketch every genome, load the database, every genome against all of them
            handles = eng.sketch_device_contigs(buf.data_ptr(), big[1], big[2], big[3])
            db = eng.make_db(names, handles, n_refs)
            try:
                return eng.query_many(db, handles, n_refs)
            finally:
                eng.lib.psk_db_destroy(db)
        if args.workload == "metagenome":   # sketch every contig, query them all against the resident database
            qh = eng.sketch_device(meta_state["cbuf"].data_ptr(), meta_state["coffs"], meta_state["clens"])
            try:
                return eng.query_many(meta_state["db"], qh, len(meta_state["coffs"]), args.faster_small)
            finally:
                for h in qh:
                    eng.lib.psk_sketch_free(h)
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
        if args.workload == "allvsall":
            pairs_per_step, sketched = n_refs * n_refs, n_refs
            bases = float(sum(lens[:n_refs]))
            wl = f"all-vs-all {n_refs} x {n_refs} synthetic ~5 Mb genomes per GPU ({n_families} families x {n_refs // n_families}), c=125 marker_c=1000 k=15"
        elif args.workload == "mammalian":
            pairs_per_step, sketched = n_refs * n_refs, n_refs
            bases = float(sum(lens))
            wl = (f"mammalian scale (BASELINE configs[4] shape at reduced count): all-vs-all of {n_refs} synthetic genomes of 24 x {args.contig_mb} Mb contigs "
                  f"({24 * args.contig_mb / 1000:.1f} Gb each; families of 4, substitution rates {DIVERGENCE[:4]}), c=125 marker_c=1000 k=15")
        elif args.workload == "metagenome":
            pairs_per_step, sketched = args.queries * n_refs, args.queries
            bases = float(sum(meta_state["clens"]))
            wl = (f"metagenome: {args.queries} contigs (2-50 kb, log-uniform, 0-5 % divergence) vs a resident database of {n_refs} synthetic ~5 Mb refs "
                  f"({n_families} families), c=30 marker_c=200 k=15, faster_small={args.faster_small}")
        else:
            pairs_per_step, sketched = n_refs, n_refs + 1
            bases = float(sum(lens))
            wl = f"1 query vs {n_refs} synthetic ~5 Mb refs per GPU (10 families x {n_refs // N_FAMILIES}), c=125 marker_c=1000 k=15"
        value = pairs_per_step * world * args.steps / dt
        # sketch_scan: ALGORITHMIC bytes per launch = sum over the launch's genomes of
        # L (ASCII read) + L/4 (2-bit packed write)   [SURVEY.md §8(d) B_sk, first two terms; DESIGN.md §4]
        alg_bytes = bases * 1.25 * args.steps / max(1, scan_n)   # per launch (a step may split into sub-batches)
        avg_s = (scan_ms / max(1, scan_n)) * 1e-3
        achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = valu = None
        pmc = os.path.join(ROOT, "profiles", "r2", "r2n_pmc_sketch_scan.json")
        if os.path.exists(pmc) and args.workload not in ("metagenome", "mammalian"):
            # OFFLINE counters (rocprofv3 --pmc in separate passes, profiles/r2/r2n_pmc_sketch_scan.json), scaled by this launch's bases:
            # HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) and VALU wave-instructions (SQ_INSTS_VALU = 26.8 per 64 bases, 61 % of them
            # four-cycle and 39 % two-cycle by profiles/micro/valu_rates.hip = 3.2 cycles on average)
            pm = json.load(open(pmc))
            traffic = pm["traffic_bytes_per_base"] * bases * args.steps / max(1, scan_n)
            insts = pm["valu_wave_instructions_per_launch"] / pm["bases_per_launch"] * bases * args.steps / max(1, scan_n)
            simd_cycles_peak = avg_s * 2.4e9 * 1024          # 256 CUs x 4 SIMDs at the 2.4 GHz maximum clock
            valu = {"valu_issue_frac": insts * 3.2 / simd_cycles_peak if avg_s > 0 else None,
                    "valu_issue_frac_at_1p93GHz": insts * 3.2 / (avg_s * 1.93e9 * 1024) if avg_s > 0 else None,
                    "valu_wave_instructions_per_launch": insts,
                    "source": "SQ_INSTS_VALU measured offline (profiles/r2/r2n_pmc_sketch_scan.json), cycle classes from profiles/micro/valu_rates.hip; "
                              "launch duration measured live; 1.93 GHz = GRBM_GUI_ACTIVE clock of the profiled launch"}
        line = {
            "metric": "genome-pairs/sec (sketch+ANI)", "value": value, "unit": "genome-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wl, "refs_per_gpu": n_refs, "hits": int(n_hits), "parallelism": f"refs sharded over {world} GPU(s)"},
            "roofline": {"kernel": "sketch_scan_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": "offline rocprofv3 --pmc pass (profiles/r2/r2n_pmc_sketch_scan.json), scaled by bases" if traffic else None,
                         "avg_launch_ms": avg_s * 1e3, "launches": int(scan_n),
                         "algorithmic_bytes_per_launch": alg_bytes, "valu": valu,
                         "note": "priced against HBM as the contract asks; the kernel's real roof is integer VALU issue (one 64-bit mix per base: "
                                 "15 four-cycle + 7 two-cycle wave instructions per base, DESIGN.md section 4): see valu.valu_issue_frac"},
            "kernel_ms_per_step": kernel_ms,
            # SURVEY.md §8(d) side figures, whole job
            "extras": {"genomes_sketched_per_s": sketched * world * args.steps / dt,
                       "bases_sketched_per_s": bases * world * args.steps / dt,
                       "reported_hits_per_step": int(n_hits)},
        }
        if args.workload == "allvsall" and getattr(eng, "last_chain_work", None):
            # per-kernel roofline of the chain stage: ALGORITHMIC bytes (SURVEY.md §8d B_ch terms) / HIP-event kernel time
            w = eng.last_chain_work
            kr = {}
            for name, key, nbytes, what in (("anchor_join", "anchor", 16.0 * w["items"], "8 B query k-mer + order read and 8 B record written per (pair, query seed)"),
                                            ("chain_lane", "chain_chunk", 16.0 * w["anchors"], "16 B per anchor read")):
                t = kernel_ms[key] * 1e-3
                if t > 0:
                    kr[name] = {"achieved_GBps": nbytes / t / 1e9, "frac_of_hbm_peak": nbytes / t / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": nbytes,
                                "ms_per_step": kernel_ms[key], "bytes": what}
            line["extras"]["chain_kernel_roofline"] = {"work_per_step": w, "kernels": kr,
                                                       "note": "neither kernel is HBM-bound: chain_lane is VALU-issue bound, the join is bound by dependent-load latency and issue "
                                                               "(profiles/r2/r2s_pmc_allvsall1000.md, r2e_pmc_join_kernels_sq.txt)"}
        if args.workload == "metagenome":
            line["extras"].update(queries_per_s=args.queries * world * args.steps / dt, db_build_s=meta_state["db_build_s"])
            if world == 1 and args.cpu_sample > 0:
                n_cpu_refs = min(n_refs, 500)
                href = buf[:offs[n_cpu_refs]].cpu().numpy()
                chost0 = meta_state["cbuf"].cpu().numpy()
                sample = [chost0[meta_state["coffs"][i]:meta_state["coffs"][i] + meta_state["clens"][i]].tobytes() for i in range(min(512, args.queries))]
                line["cpu_baseline"] = cpu_baseline_metagenome(lambda i: href[offs[i]:offs[i] + lens[i]].tobytes(), n_cpu_refs, sample, os.cpu_count() or 1, args.faster_small)
                del href, chost0
            if world == 1 and args.api_queries > 0:
                # the same contigs ONE AT A TIME through the pyskani-shaped API, from host bytes: Database.query(name, contig)
                import pyskani_amd as psk
                host = buf.cpu().numpy()
                pdb = psk.Database(compression=30, marker_compression=200)
                t0 = time.perf_counter()
                pdb.sketch_many([(f"r{i}", host[offs[i]:offs[i] + lens[i]].tobytes()) for i in range(n_refs)])
                t_load = time.perf_counter() - t0
                del host
                chost = meta_state["cbuf"].cpu().numpy()
                nq = min(args.api_queries, args.queries)
                contigs = [chost[meta_state["coffs"][i]:meta_state["coffs"][i] + meta_state["clens"][i]].tobytes() for i in range(nq)]
                for c in contigs[:20]:
                    pdb.query("w", c, learned_ani=False, faster_small=args.faster_small)
                t0 = time.perf_counter()
                nh = sum(len(pdb.query(f"c{i}", c, learned_ani=False, faster_small=args.faster_small)) for i, c in enumerate(contigs))
                t_q = time.perf_counter() - t0
                # ... and the same calls from eight host threads (the reference's query() releases the GIL: threads are its route to
                # concurrency; here every thread's call runs on its own lane of the context)
                import threading
                def _work(lo, hi):
                    for i in range(lo, hi):
                        pdb.query(f"c{i}", contigs[i], learned_ani=False, faster_small=args.faster_small)
                th = [threading.Thread(target=_work, args=(k * nq // 8, (k + 1) * nq // 8)) for k in range(8)]
                t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t_q8 = time.perf_counter() - t0
                many = pdb.query_many([(f"c{i}", c) for i, c in enumerate(contigs)], learned_ani=False, faster_small=args.faster_small)
                assert sum(len(h) for h in many) == nh
                t0 = time.perf_counter()
                pdb.query_many([(f"c{i}", c) for i, c in enumerate(contigs)], learned_ani=False, faster_small=args.faster_small)
                t_many = time.perf_counter() - t0
                line["extras"].update(api_queries_per_s=nq / t_q, api_query_ms=t_q / nq * 1e3, api_queries_per_s_8_threads=nq / t_q8, api_query_many_per_s=nq / t_many,
                                      api_db_load_s=t_load, api_hits=nh,
                                      api_note=f"{nq} contigs from host bytes through pyskani_amd.Database: one Database.query() per contig (from one host thread, and from eight), and one Database.query_many() for all of them")
        if world == 1 and args.workload == "search" and (args.cpu_sample > 0 or not args.no_api):
            host = buf.cpu().numpy()
            fetch = lambda i: host[offs[i]:offs[i] + lens[i]].tobytes()
            if not args.no_api:
                import pyskani_amd as psk
                r = api_rates(psk, [fetch(i) for i in range(n_refs)], fetch(-1))
                line["extras"].update(api_pairs_per_s=r["api"]["pairs_per_s"], host_ascii_pairs_per_s=r["host_ascii"]["pairs_per_s"],
                                      api_detail=r["api"], host_ascii_detail=r["host_ascii"],
                                      api_note="same 1 query vs refs workload from ASCII bytes in HOST memory through pyskani_amd.Database: "
                                               "api = n x sketch() + query(); host_ascii = sketch_many() (pinned, double-buffered H2D pipeline) + query()")
            if args.cpu_sample > 0:
                ns = min(args.cpu_sample, n_refs)
                threads = os.cpu_count() or 1
                v1, secs1, nh, vall, secs_all = cpu_baseline(fetch, ns, threads)
                line["cpu_baseline"] = {"value": v1, "unit": "genome-pairs/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                                        "sample": f"1 query vs the first {ns} of {n_refs} refs of the same workload (sketch {ns + 1} genomes + {ns} screens + {nh} chained hits), {secs1:.1f} s of oracle time",
                                        "all_cores": {"value": vall, "cores": threads, "seconds": secs_all,
                                                      "how": "same sample, one reference (sketch + screen + chain) per thread over every hardware thread of the host"},
                                        "note": "the repo's own C restatement (the Rust reference cannot be built here); it keeps seeds in flat arrays where skani inserts "
                                                "into hash maps, so it is if anything faster than the Rust path: a conservative floor for the speed-up"}
        if world == 1 and args.workload == "allvsall" and args.cpu_sample > 0:
            host = buf.cpu().numpy()
            line["cpu_baseline"] = cpu_baseline_allvsall(lambda i: host[offs[i]:offs[i] + lens[i]].tobytes(), n_refs,
                                                         min(args.cpu_sample, 256, n_refs), os.cpu_count() or 1)
        print(json.dumps(line), flush=True)
    if meta_state.get("db"):
        eng.lib.psk_db_destroy(meta_state["db"])
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
