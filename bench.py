#!/usr/bin/env python3
"""bench.py — genome-pairs/sec (sketch + ANI) of the MI355X-native pyskani hot path.

One "step" = one pass of the whole hot path over one batch of synthetic genomes that are already resident in HBM as ASCII:
sketch every genome (FracMinHash seeds, marker sets, k-mer index), load them into a fresh database, run the query (marker
screen -> seed-index lookup -> chaining -> ANI/AF) and bring the hit list back to the host. Nothing is cached between steps.

The ONE JSON line (rank 0) is the CONTRACT JOB - the job `north_star`'s target is quoted on, BASELINE.json configs[2] on the GPUs given: all-vs-all of 10 000
synthetic ~5 Mb genomes (100 families of 100), 10^8 pairs, ~10^6 of them chained, c=125 marker_c=1000 k=15; a FIXED job (`scaling: "strong"`): at N>1 the references are
sharded over the ranks, the shards' sketches all-gathered as the query side and the 20-byte hit records all-gathered once, so `--gpus N` reports the same job as N=1.
`host_to_host` beside `value` is the same job from ASCII in HOST memory; `extras.oracle_check` = 8 random hits of the step recomputed by the CPU oracle;
`extras.scaling_model` = a SINGLE-GPU EMULATION of rank 0's share of the 2-, 4- and 8-way job (no multi-GPU node was available: a prediction, not a measurement).
At N=1 the default run then also measures the other configurations `north_star` names, each as an entry of `extras.workloads` with its own ms/step, hits,
dominant-kernel `roofline` and a bounded `cpu_baseline`:
  search_1k         BASELINE configs[1]: 1 query vs 1 000 synthetic ~5 Mb refs (the headline of rounds 1-5), with its host-memory API legs
  allvsall_1k_*     SURVEY.md 8(d)'s harder generator variants (genomes cut into contigs / with block rearrangements), 8 hits of each recomputed by the oracle
  metagenome_100k   BASELINE configs[3]: 100 000 contigs (2-50 kb) vs a resident database of 5 000 x ~5 Mb refs, c=30 marker_c=200,
                    with the rescue of short contigs on (default) and off (`faster_small`)
  mammalian_8x3Gb   BASELINE configs[4] shape at reduced count: all-vs-all of 8 genomes of 24 x 125 Mb contigs; two of the chained
                    pairs are checked against the CPU oracle outside the timed region
(`--no-workloads` skips them; `--workload X` runs X alone as the line; `--refs N` without `--workload` runs the contract job at N genomes.)

Every line / entry carries
  roofline      the workload's dominant kernel among those with a stated algorithmic byte count, timed with HIP events
                on the library's stream: achieved = algorithmic bytes / kernel time, against the 8 TB/s HBM peak; `traffic` = HBM bytes
                from an offline rocprofv3 --pmc pass scaled by the run's units when profiles/ holds one for that kernel, else null
  cpu_baseline  the CPU oracle (oracle/, a port — the Rust reference cannot be built here) on a bounded sample: one core (what a
                pyskani call uses, lib.rs:493,569) and all host cores
  clock         shader clock under an integer-VALU load, probed before the timed loop (sketch_scan is bound by VALU issue: its
                time follows the clock the box holds)

N>1 (one process per GPU, launched by torch.distributed.run): the contract job shards its FIXED set of genomes over the ranks (strong scaling): every rank sketches
its share, the shards' sketches are all-gathered as packed device records in batches (the query side; round b + 1 travels while round b is queried), queried against
the local shard, and the hit records are all-gathered. `--workload search` shards 1 000 references per GPU (weak scaling), the query is replicated and the per-shard hit
lists are all-gathered. `--comm torch` moves the exchange steps through torch.distributed (backend nccl = RCCL), `--comm capi` through the library's own RCCL
communicator (psk_comm_*).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_REFS = 1000            # per GPU (search)
N_FAMILIES = 10
DIVERGENCE = (0.0005, 0.002, 0.005, 0.01, 0.02, 0.04, 0.07, 0.10)   # SURVEY.md §8(d) family model
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
KERNELS = ("sketch_scan", "sketch_emit", "sketch_sort", "screen", "anchor", "anchor_emit", "chain_chunk", "select", "pair_reduce")
PORT_NOTE = ("the repo's own C restatement (the Rust reference cannot be built here); it keeps seeds in flat arrays where skani inserts "
             "into hash maps, so it is if anything faster than the Rust path: a conservative floor for the speed-up")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


# ------------------------------------------------------------------ the contract line: compact (< 4 kB) on stdout, everything else on disk
LINE_BUDGET = 4096      # the driver keeps an 8 kB stdout tail: the LAST line must fit well inside it (round 3's 21 kB line could not be parsed)


def _sig(x, digits=5):
    """floats to `digits` significant digits (the full-precision object is in the file named by `full`)"""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    return x


def _clip(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "…"


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if d is not None and k in d}


def compact_roofline(r, short=False):
    if not r:
        return None
    keys = ("kernel", "frac", "traffic") if short else ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_of_copy_bw", "traffic", "traffic_source", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch")
    out = _pick(r, keys)
    if "traffic_source" in out:
        out["traffic_source"] = _clip(out["traffic_source"], 56)
    return out


def compact_cpu(c, short=False):
    if not c:
        return None
    if short:
        out = _pick(c, ("value", "cores"))
        if "all_cores" in c:
            out["all_cores"] = _pick(c["all_cores"], ("value", "cores"))
        return out
    out = _pick(c, ("value", "unit", "cores", "kind", "cpu"))
    out["sample"] = _clip(c.get("sample", ""), 160)
    if "all_cores" in c:
        out["all_cores"] = _pick(c["all_cores"], ("value", "cores"))
    return out


def compact_workload(e):
    """one entry of extras.workloads in the compact line: ms_per_step, value, unit, hits, roofline{kernel, frac, traffic}, cpu_baseline{value, cores}"""
    if not e:
        return None
    if "ms_per_step" not in e:      # the API leg: rates only
        return {k: _sig(v) for k, v in e.items() if isinstance(v, (int, float)) and not isinstance(v, bool) and k not in ("api_query_ms", "api_db_load_s")}
    out = _pick(e, ("ms_per_step", "value", "unit", "hits", "hits_digest"))
    if str(e.get("scaling", "")).startswith(("strong", "weak")):
        out["scaling"] = e["scaling"]
    out["roofline"] = compact_roofline(e.get("roofline"), short=True)
    if e.get("host_to_host") and "value" in e["host_to_host"]:
        out["host_to_host"] = _pick(e["host_to_host"], ("value", "ms_per_step", "ingest", "vs_cpu_all_cores"))
    if e.get("exchange"):
        out["exchange"] = _pick(e["exchange"], ("ranks", "backend", "bytes_sent_per_rank_last_step", "collective_s_last_step", "psk_s_last_step"))
    if e.get("cpu_baseline"):
        out["cpu_baseline"] = compact_cpu(e["cpu_baseline"], short=True)
    if "oracle_check" in e:
        out["oracle_check"] = e["oracle_check"].get("result")
    return out


def compact_line(full, full_path=None):
    """The contract line the driver parses. Everything bulky (per-kernel tables, prose notes, per-workload detail) stays in `full`
    (written to `full_path`, and each workload's full entry is its own EARLIER stdout line)."""
    line = {k: _sig(full[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in full}
    cfg = dict(full.get("config", {}))
    cfg["workload"] = _clip(cfg.get("workload", ""), 200)
    for k in list(cfg):
        if isinstance(cfg[k], str) and k != "workload":
            cfg[k] = _clip(cfg[k], 120)
    line["config"] = cfg
    line["roofline"] = compact_roofline(full.get("roofline"))
    if full.get("roofline") and full["roofline"].get("valu"):
        line["roofline"]["valu_issue_frac"] = _sig(full["roofline"]["valu"].get("valu_issue_frac_at_probed_clock"))
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    if full.get("host_to_host"):
        line["host_to_host"] = _pick(full["host_to_host"], ("value", "unit", "ms_per_step", "ingest", "vs_cpu_1_core", "vs_cpu_all_cores"))
    if "clock" in full and full["clock"]:
        line["clock_mhz"] = _sig(full["clock"].get("shader_clock_mhz"))
    if "copy_bw" in full:
        line["copy_bw_GBps"] = _sig(full["copy_bw"].get("GBps"))
    if "kernel_ms_per_step" in full:
        line["kernel_ms_per_step"] = {k: _sig(v, 4) for k, v in full["kernel_ms_per_step"].items()}
    ex_full = full.get("extras", {})
    ex = {}
    for k in ("genomes_sketched_per_s", "bases_sketched_per_s", "api_pairs_per_s", "host_ascii_pairs_per_s", "host_packed_pairs_per_s", "queries_per_s", "workloads_wall_s", "hits_digest"):
        if k in ex_full:
            ex[k] = _sig(ex_full[k])
    if "exchange" in ex_full:
        ex["exchange"] = _pick(ex_full["exchange"], ("ranks", "backend", "bytes_sent_per_rank_last_step", "collective_s_last_step", "psk_s_last_step", "outside_psk_and_collectives_frac"))
    if "oracle_check" in ex_full:
        ex["oracle_check"] = ex_full["oracle_check"].get("result")
    if ex_full.get("scaling_model"):
        ex["scaling_model"] = {"kind": _clip(ex_full["scaling_model"].get("kind", ""), 60),
                               "ranks": {n: _pick(v, ("rank_ms", "speedup_vs_1", "pairs_chained", "index_lookups", "bytes_received")) for n, v in ex_full["scaling_model"].get("ranks", {}).items()}}
    if "workloads" in ex_full:
        ex["workloads"] = {k: compact_workload(v) for k, v in ex_full["workloads"].items() if v}
    line["extras"] = ex
    if full_path:
        line["full"] = full_path
    if len(json.dumps(line)) >= LINE_BUDGET:      # never exceed the budget: drop the least important parts, in this order
        line.pop("kernel_ms_per_step", None)
    if len(json.dumps(line)) >= LINE_BUDGET and "scaling_model" in ex:
        ex["scaling_model"]["ranks"] = {n: _pick(v, ("rank_ms", "speedup_vs_1")) for n, v in ex["scaling_model"]["ranks"].items()}
    if len(json.dumps(line)) >= LINE_BUDGET and "workloads" in ex:
        ex["workloads"] = {k: _pick(v, ("ms_per_step", "value", "unit")) for k, v in ex["workloads"].items()}
    while len(json.dumps(line)) >= LINE_BUDGET and ex.get("workloads"):
        ex["workloads"].popitem()
        ex["workloads_truncated"] = True
    return line


def write_full(full, tag):
    """the complete object: profiles/r6/ (tracked) and gpurun_out/ (what travels back from the GPU box). Returns the repo-relative path."""
    name = f"bench_full_{tag}_{time.strftime('%Y%m%d_%H%M%S')}.json"
    rel = None
    for d in (os.path.join("gpurun_out"), os.path.join("profiles", "r6")):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            with open(os.path.join(ROOT, d, name), "w") as f:
                json.dump(full, f, indent=1)
            rel = os.path.join(d, name)
        except OSError:
            pass
    return rel


def spawn_decision(gpus, env):
    """bench.py --gpus N without a launcher starts its own N ranks: True when this process must become the launcher's parent.
    (The driver's own launch sets WORLD_SIZE; a rank never spawns.)"""
    return gpus > 1 and "WORLD_SIZE" not in env and "RANK" not in env


def spawn_ranks(gpus, argv):
    """Start `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a CHILD (never exec: this process stays the
    parent and exits with the child's code), pass its output through, and repeat rank 0's contract line as the LAST stdout line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last = None
    for ln in proc.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            if last is not None:
                print(last, flush=True)
            last = t
        else:
            sys.stdout.write(ln); sys.stdout.flush()
    rc = proc.wait()
    if last is not None:
        print(last, flush=True)
    return rc


def device_copy_bandwidth(torch, device, nbytes=1 << 30, reps=5):
    """SURVEY.md §8(d)'s second denominator: the rate of a plain device-to-device copy on this box (bytes read + bytes written per second)"""
    a = torch.empty(nbytes, dtype=torch.uint8, device=device); b = torch.empty_like(a)
    a.zero_(); b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return {"GBps": 2.0 * nbytes / (ms * 1e-3) / 1e9, "bytes": nbytes, "ms": ms, "how": "torch copy_ of 1 GiB device to device, read + written bytes over the HIP-event time"}


# ------------------------------------------------------------------ synthetic data (built on the GPU)
def family_layout(seed_shared, n_genomes, n_families):
    """Family model of SURVEY.md §8(d): ancestors of iid ACGT, L ~ U[4.5, 5.5] Mb; genome i belongs to family i // (n / families).
    Returns (ancestor lengths, family of every genome)."""
    rng = np.random.default_rng(seed_shared)
    lens = [int(rng.integers(4_500_000, 5_500_001)) for _ in range(n_families)]
    per_fam = max(1, n_genomes // n_families)
    return lens, [min(i // per_fam, n_families - 1) for i in range(n_genomes)]


def make_genomes(torch, device, seed_shared, seed_members, ids, fam_of, anc_lens, query_family=None, variant="plain"):
    """Members `ids` (GLOBAL genome indices) of the family model: each carries independent substitutions at the cycled rates on its
    family's ancestor, drawn from a generator seeded by (seed_members, global index) — so a rank that builds only its shard gets
    the same genomes as a rank that builds them all. With `query_family` set, one more genome follows: the query (d = 0.02).
    Returns one uint8 ASCII tensor plus per-genome (offset, length), every offset 16-byte aligned.
    variant: "sv" = on top of the substitutions, 20 block rearrangements per member (5-50 kb each: inverted in place - reverse
    complement - or moved elsewhere), drawn from a numpy generator seeded by the genome's global index; "contigs" = the genome cut into 1-80
    contigs at random 16-byte aligned points: then the return value is (buf, contig offsets, contig lengths, genome_first_contig)."""
    gs = torch.Generator(device=device)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    fams = sorted({fam_of[i] for i in ids} | ({query_family} if query_family is not None else set()))
    anc = {}
    for f in fams:      # every rank draws the ancestors it needs from the shared seed
        gs.manual_seed(seed_shared * 1_000_003 + f)
        anc[f] = torch.randint(0, 4, (anc_lens[f],), generator=gs, device=device, dtype=torch.uint8)
    todo = [(i, fam_of[i], DIVERGENCE[i % len(DIVERGENCE)]) for i in ids]
    if query_family is not None:
        todo.append((-1, query_family, 0.02))
    offs, glen, total = [], [], 0
    for _, f, _ in todo:
        offs.append(total); glen.append(anc_lens[f]); total += (anc_lens[f] + 15 + 16) & ~15
    buf = torch.zeros(total + 64, dtype=torch.uint8, device=device)
    gm = torch.Generator(device=device)
    for j, (i, f, d) in enumerate(todo):
        a = anc[f]
        gm.manual_seed((seed_members if i >= 0 else seed_shared) * 1_000_003 + (i if i >= 0 else 999_983))
        mut = torch.rand(a.shape, generator=gm, device=device) < d
        shift = torch.randint(1, 4, a.shape, generator=gm, device=device, dtype=torch.uint8)
        codes = torch.where(mut, (a + shift) & 3, a)
        if variant == "sv":
            r = np.random.default_rng((seed_members, 77, i if i >= 0 else 999_983))
            for _ in range(20):
                L = int(r.integers(5_000, 50_001)); s0 = int(r.integers(0, codes.numel() - L))
                if r.random() < 0.5:      # inversion: the block's reverse complement in place
                    codes[s0:s0 + L] = 3 - torch.flip(codes[s0:s0 + L], (0,))
                else:                     # translocation: the block cut out and put back elsewhere
                    block = codes[s0:s0 + L].clone(); rest = torch.cat((codes[:s0], codes[s0 + L:]))
                    t0 = int(r.integers(0, rest.numel() + 1))
                    codes = torch.cat((rest[:t0], block, rest[t0:]))
        buf[offs[j]:offs[j] + glen[j]] = lut[codes.long()]
        del mut, shift, codes
    if variant == "contigs":
        c_off, c_len, gfc = [], [], [0]
        for j, (i, _, _) in enumerate(todo):
            r = np.random.default_rng((seed_members, 78, i if i >= 0 else 999_983))
            n = int(r.integers(1, 81))
            cuts = sorted({int(x) & ~15 for x in r.integers(16, glen[j] - 16, n - 1)}) if n > 1 else []
            edges = [0] + cuts + [glen[j]]
            for a0, a1 in zip(edges[:-1], edges[1:]):
                if a1 > a0:
                    c_off.append(offs[j] + a0); c_len.append(a1 - a0)
            gfc.append(len(c_off))
        return buf, c_off, c_len, gfc
    return buf, offs, glen


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


def fill_big_genome(torch, device, buf, gi, n_contigs, contig_len, fam_size, seed):
    """Genome gi of the configs[4] family model into `buf` (n_contigs x stride bytes, reused from genome to genome): ancestor contig c of
    family gi // fam_size drawn from a generator seeded by (seed, family, contig), the member's substitutions from (seed, genome, contig) -
    any genome can be (re)built alone, so 50 x 3 Gb never has to be resident as ASCII. Returns the contig offsets / lengths."""
    g = torch.Generator(device=device)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    stride = (contig_len + 15 + 16) & ~15
    fam, j = gi // fam_size, gi % fam_size
    d = DIVERGENCE[:4][j % 4]
    for c in range(n_contigs):
        g.manual_seed((seed * 1_000_003 + fam) * 1009 + c)
        a = torch.randint(0, 4, (contig_len,), generator=g, device=device, dtype=torch.uint8)
        g.manual_seed(((seed + 7) * 1_000_003 + gi) * 1009 + c)
        mut = torch.rand(a.shape, generator=g, device=device) < d
        shift = torch.randint(1, 4, a.shape, generator=g, device=device, dtype=torch.uint8)
        buf[c * stride:c * stride + contig_len] = lut[torch.where(mut, (a + shift) & 3, a).long()]
        del a, mut, shift
    return [c * stride for c in range(n_contigs)], [contig_len] * n_contigs


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


# ------------------------------------------------------------------ the library, through its C-ABI
class Engine:
    """Thin driver over the C-ABI for device-resident genomes."""

    def __init__(self, device, c=125, marker_c=1000, k=15):
        from pyskani_amd import _capi
        self.capi = _capi
        self.lib = _capi.load()
        self.ctx = C.c_void_p()
        _capi.check(self.lib.psk_ctx_create(device, C.byref(self.ctx)))
        self.params = _capi.Params(c, marker_c, k)
        self.hit_dtype = np.dtype(_capi.Hit)
        self.hit_min_dtype = np.dtype(_capi.HitMin)

    def close(self):
        if self.ctx:
            self.lib.psk_ctx_destroy(self.ctx)
            self.ctx = None

    def sync(self):
        self.capi.check(self.lib.psk_ctx_synchronize(self.ctx))

    def clock_probe(self):
        mhz, ms = C.c_double(), C.c_double()
        self.capi.check(self.lib.psk_ctx_clock_probe(self.ctx, C.byref(mhz), C.byref(ms)))
        return {"shader_clock_mhz": mhz.value, "probe_ms": ms.value,
                "how": "fixed integer-VALU micro-kernel (v_alignbit_b32 chains, 4 cycles per wave64 instruction, 8 waves per SIMD) timed with HIP events just before the timed loop"}

    def sketch_device(self, d_ptr, offs, lens, gfc=None):
        n = len(gfc) - 1 if gfc is not None else len(offs)
        nc = len(offs)
        c_off = (C.c_uint64 * nc)(*offs); c_len = (C.c_uint64 * nc)(*lens)
        g = (C.c_uint32 * (n + 1))(*(gfc if gfc is not None else range(n + 1)))
        return self.sketch_device_c(d_ptr, c_off, c_len, g, n)

    def layout(self, offs, lens, gfc=None):
        """ctypes views of a constant genome layout, built once: they describe the resident input"""
        n = len(gfc) - 1 if gfc is not None else len(offs)
        nc = len(offs)
        return ((C.c_uint64 * nc)(*offs), (C.c_uint64 * nc)(*lens), (C.c_uint32 * (n + 1))(*(gfc if gfc is not None else range(n + 1))), n)

    def sketch_device_c(self, d_ptr, c_off, c_len, gfc, n):
        out = (C.c_void_p * n)()
        self.capi.check(self.lib.psk_sketch_batch_device(self.ctx, C.byref(self.params), C.c_void_p(d_ptr), c_off, c_len, gfc, n, 1, out))
        return out

    def make_db(self, names, handles, n):
        db = C.c_void_p()
        self.capi.check(self.lib.psk_db_create(self.ctx, C.byref(self.params), C.byref(db)))
        self.capi.check(self.lib.psk_db_add_batch(db, names, handles, n))
        return db

    def query_many(self, db, handles, n, faster_small=False, keep=False, raw=False):
        """psk_query_many_min -> number of hits (keep=True: the 20-byte psk_hit_min records and the per-query offsets as numpy arrays). The timed steps
        take these records - what the reference's Hit holds (hit.rs:77-104); raw=True: psk_query_many, the 80-byte psk_hit with every chaining integer,
        which the oracle checks outside the timed regions read."""
        opts = self.capi.QueryOpts(0, 0, 0, int(faster_small), 0.0, 0.0, None)
        hits_p = C.POINTER(self.capi.Hit if raw else self.capi.HitMin)()
        offsets = (C.c_uint64 * (n + 1))()
        self.capi.check((self.lib.psk_query_many if raw else self.lib.psk_query_many_min)(db, handles, n, C.byref(opts), C.byref(hits_p), offsets))
        nh = int(offsets[n])
        recs = None
        if keep:
            recs = (self.capi.hit_records(hits_p, 0, nh, self.hit_dtype if raw else self.hit_min_dtype),
                    np.frombuffer(offsets, dtype=np.uint64).astype(np.int64))
        if hits_p:
            self.lib.psk_free(hits_p)
        return (nh, recs) if keep else nh

    def query_one(self, db, handle):
        opts = self.capi.QueryOpts(0, 0, 0, 0, 0.0, 0.0, None)
        hits_p = C.POINTER(self.capi.Hit)()
        nh = C.c_uint64(0)
        self.capi.check(self.lib.psk_query(db, handle, C.byref(opts), C.byref(hits_p), C.byref(nh)))
        recs = self.capi.hit_records(hits_p, 0, nh.value, self.hit_dtype)
        if hits_p:
            self.lib.psk_free(hits_p)
        return recs

    def timing(self, kernel):
        ms, n = C.c_double(0), C.c_uint64(0)
        self.capi.check(self.lib.psk_ctx_timing(self.ctx, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def work(self, reset=False):
        p, i, a = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.capi.check(self.lib.psk_ctx_work(self.ctx, C.byref(p), C.byref(i), C.byref(a), int(reset)))
        lk, vis, cands, rows = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.capi.check(self.lib.psk_ctx_join_work(self.ctx, C.byref(lk), C.byref(vis), C.byref(cands), C.byref(rows), int(reset)))
        return {"chained_pairs": p.value, "items": i.value, "anchors": a.value, "index_lookups": lk.value, "index_entries_visited": vis.value, "candidates": cands.value, "chunk_rows": rows.value}


def timed_loop(eng, step, steps, warmup, fence):
    """W untimed warm-up steps, the clock probe, then EXACTLY K timed steps bracketed by fence() on both sides. Returns seconds,
    the last step's result, per-kernel HIP-event ms per step + launch counts, chain-stage work per step, and the clock record."""
    res = None
    for _ in range(warmup):
        res = step()
    clock = eng.clock_probe()
    eng.capi.check(eng.lib.psk_ctx_set_timing(eng.ctx, 1))
    eng.timing("reset")
    eng.work(reset=True)
    fence()
    t0 = time.perf_counter()
    marks = [t0]
    for _ in range(steps):
        res = step()
        marks.append(time.perf_counter())      # (every step ends with its results on the host: the marks are the steps' own wall times; nothing is added to the timed region but this call)
    fence()
    dt = time.perf_counter() - t0
    timed_loop.last_step_ms = [round((b - a) * 1e3, 2) for a, b in zip(marks, marks[1:])]
    kern = {k: eng.timing(k) for k in KERNELS}
    eng.capi.check(eng.lib.psk_ctx_set_timing(eng.ctx, 0))
    work = {k: v / max(1, steps) for k, v in eng.work(reset=True).items()}
    return dt, res, kern, work, clock


def load_pmc():
    """offline counter passes (rocprofv3 --pmc, separate runs; profiles/scripts/pmc_summary.py writes the JSON): HBM bytes per unit of work of
    the profiled kernels, {workload: {timer name: {"bytes_per_unit": x, "unit": "base" | "item" | "anchor", "source": ...}}}"""
    out = {}
    rel = next((r for r in ("profiles/r6/r6_pmc_sketch_scan.json", "profiles/r5/r5_pmc_sketch_scan.json") if os.path.exists(os.path.join(ROOT, *r.split("/")))), "profiles/r2/r2n_pmc_sketch_scan.json")
    p = os.path.join(ROOT, *rel.split("/"))
    if os.path.exists(p):      # (counted again in round 5: the kernel's body became a template in round 4)
        d = json.load(open(p))
        ss = {"bytes_per_unit": d["traffic_bytes_per_base"], "unit": "base", "source": rel, "valu_per_base": d["valu_wave_instructions_per_launch"] / d["bases_per_launch"]}
        for wl in ("search", "allvsall", "metagenome", "mammalian"):
            out.setdefault(wl, {})["sketch_scan"] = ss
    # (round 4: scale factors calibrated per access shape, profiles/r4/r4k_pmc_calibration.md; the round-3 file for whatever the newer one lacks)
    for rel in (("profiles", "r3", "pmc_kernels.json"), ("profiles", "r4", "pmc_kernels.json"), ("profiles", "r5", "pmc_kernels.json"), ("profiles", "r6", "pmc_kernels.json")):
        p = os.path.join(ROOT, *rel)
        if os.path.exists(p):
            for wl, timers in json.load(open(p)).items():
                if wl.startswith("_"):
                    continue
                for k, v in timers.items():
                    out.setdefault(wl, {})[k] = dict(out.get(wl, {}).get(k, {}), **dict(v, source="/".join(rel)))      # (keeps what an earlier file alone carries: sketch_scan's VALU count)
    return out


def kernel_rooflines(kern, steps, units, pmc):
    """Per-kernel roofline table of one workload. `units` = per STEP {bases, c, marker_c, items, anchors, index_lookups, index_entries_visited, candidates,
    chunk_rows, chained_pairs} (the library's own work counters: psk_ctx_work / psk_ctx_join_work). ALGORITHMIC bytes (SURVEY.md §8d B_sk / B_ch terms):
       sketch_scan   L (ASCII read) + L/4 (2-bit packed write)                                    per base
       sketch_emit   L/8 (seed mask) + L/4 (packed read) + 20 L/c (seed records) + 8 L/marker_c   per base
       sketch_sort   20 B per seed (k-mer index) + 16 B per marker
       the join, per-pair merge join (no index lookups counted):
         anchor        8 B (query k-mer + its position order) read + 8 B record written           per (pair, query seed) item
         anchor_emit   8 B record read per item + 16 B anchor written per anchor
       the join through the database-wide seed index (index_lookups > 0: one lookup per query SEED finds its matches in every reference):
         anchor        COUNT walk (where it runs: the two-pass forms): 12 B per lookup (k-mer, two bucket bounds) + 12 B per index entry of the looked-up runs
         anchor_emit   EMIT walk: the same reads + 8 B per lookup (the seed's position and contig|strand) + 16 B per anchor written
       chain_chunk   16 B per anchor read (the DP; its candidates are a few bytes per chunk)
       select        32 B per candidate chain (seven 4-byte fields read, the verdict written) + 8 B per chunk row (its table entry) + 128 B per chunk row (chunk_seeds_kernel,
                     inside this timer: two binary searches of ~16 four-byte probes in the query's seed positions)
       pair_reduce   32 B per chunk row (the chunk's totals) + 80 B per pair (the record written)
    A bracket that did not do the work its formula counts (a join pass that was not launched: ~0 ms) would price above the HBM peak: such rows carry no fraction
    (tests/test_bench_line_cpu.py rejects frac > 1). The screen (an inverted-index lookup, below 3 % of every step) is reported as time only."""
    bases, items, anchors = units["bases"], units.get("items", 0.0), units.get("anchors", 0.0)
    lookups, visited = units.get("index_lookups", 0.0), units.get("index_entries_visited", 0.0)
    cands, rows, pairs = units.get("candidates", 0.0), units.get("chunk_rows", 0.0), units.get("chained_pairs", 0.0)
    alg = {"sketch_scan": (1.25 * bases, "L + L/4 per base"),
           "sketch_sort": ((20.0 / units["c"] + 16.0 / units["marker_c"]) * bases, "k-mer index: 20 B per seed (key, position|meta, order); marker sets: 16 B per marker; per base: 20/c + 16/marker_c"),
           "sketch_emit": ((0.125 + 0.25 + 20.0 / units["c"] + 8.0 / units["marker_c"]) * bases, "L/8 + L/4 + 20 L/c + 8 L/marker_c per base"),
           "chain_chunk": (16.0 * anchors, "16 B per anchor"),
           "select": (32.0 * cands + 136.0 * rows, "32 B per candidate chain + 8 B per chunk row (its table entry) + 128 B per chunk row (chunk_seeds: two binary searches of ~16 probes in the query's seed positions)"),
           "pair_reduce": (32.0 * rows + 80.0 * pairs, "32 B per chunk row + 80 B per pair")}
    if lookups > 0:
        alg["anchor"] = (12.0 * lookups + 12.0 * visited, "seed-index COUNT walk: 12 B per query-seed lookup + 12 B per index entry visited")
        alg["anchor_emit"] = (20.0 * lookups + 12.0 * visited + 16.0 * anchors, "seed-index EMIT walk: 20 B per query-seed lookup + 12 B per index entry visited + 16 B per anchor")
    else:
        alg["anchor"] = (16.0 * items, "16 B per (pair, query seed) item")
        alg["anchor_emit"] = (8.0 * items + 16.0 * anchors, "8 B per item + 16 B per anchor")
    unit_n = {"base": bases, "item": items, "anchor": anchors, "lookup": lookups, "candidate": cands, "row": rows}
    table = {}
    for k in KERNELS:
        ms_total, launches = kern[k]
        ms_step = ms_total / max(1, steps)
        row = {"ms_per_step": ms_step, "launches_per_step": launches / max(1, steps)}
        if k in alg and alg[k][0] > 0 and ms_step > 0 and launches > 0:
            b = alg[k][0]
            frac = b / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS
            if frac <= 1.0:      # (above the peak: the bracket was empty - e.g. the one-walk index join has no COUNT pass)
                row.update(algorithmic_bytes_per_step=b, bytes=alg[k][1], achieved_GBps=b / (ms_step * 1e-3) / 1e9, frac_of_hbm_peak=frac)
                pm = pmc.get(k)
                same_join = k not in ("anchor", "anchor_emit") or (lookups > 0) == bool(pm.get("index_join", False)) if pm else False      # (a join's counters only price the join that was measured)
                if pm and unit_n.get(pm.get("unit"), 0) > 0 and same_join:
                    row["traffic_bytes_per_step"] = pm["bytes_per_unit"] * unit_n[pm["unit"]]      # (the counter pass states its own unit)
                    row["traffic_source"] = f"offline rocprofv3 --pmc pass ({pm['source']}), scaled by {pm['unit']}s"
        table[k] = row
    return table


def roofline_of(table, steps, prefer=None):
    """The `roofline` object of a line: the dominant kernel among those with an algorithmic byte count (or `prefer`)."""
    cands = [k for k, r in table.items() if "algorithmic_bytes_per_step" in r]
    if not cands:
        return None
    k = prefer if prefer in cands else max(cands, key=lambda x: table[x]["ms_per_step"])
    r = table[k]
    launches = max(1.0, r["launches_per_step"])
    top = max(table, key=lambda x: table[x]["ms_per_step"])
    out = {"kernel": k, "bound": "hbm", "achieved": r["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac_of_hbm_peak"],
           "traffic": r.get("traffic_bytes_per_step", None) and r["traffic_bytes_per_step"] / launches,
           "traffic_source": r.get("traffic_source"),
           "avg_launch_ms": r["ms_per_step"] / launches, "launches": int(round(launches * steps)),
           "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_step"] / launches, "bytes": r["bytes"]}
    if top != k:
        out["note"] = f"largest timed bracket of the step is '{top}' ({table[top]['ms_per_step']:.2f} ms), which has no stated HBM byte count (sort / LDS-latency work); '{k}' is the largest kernel that has one"
    return out


# ------------------------------------------------------------------ CPU baselines (the oracle: test infrastructure, used here as the checker's clock)
def cpu_baseline_search(fetch, n_sample, threads):
    """The CPU oracle on 1 query vs the first n_sample references. `fetch(i)` returns genome i's bytes (i = -1: the query); only
    oracle time is counted. ONE core (a pyskani call is single-threaded, lib.rs:493,569) and ALL host cores with one reference per
    thread (what a user gets from the GIL release) — ctypes drops the GIL, so plain threads scale."""
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
    return {"value": n_sample / secs1, "unit": "genome-pairs/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"1 query vs the first {n_sample} refs of the same workload (sketch {n_sample + 1} genomes + {n_sample} screens + {len(hits)} chained hits), {secs1:.1f} s of oracle time",
            "all_cores": {"value": n_sample / secs_all, "cores": threads, "seconds": secs_all,
                          "how": "same sample, one reference (sketch + screen + chain) per thread over every hardware thread of the host"},
            "note": PORT_NOTE}


def cpu_baseline_allvsall(fetch, n_refs, n_queries, threads):
    """CPU oracle on a SUB-SAMPLE of the all-vs-all workload (sub-sample configs 3-5 and extrapolate linearly in
    pairs): every reference is sketched once (all cores), then n_queries of them are queried against all n_refs (screen every
    reference, chain the shortlist). One-core figure from the first few queries alone."""
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
    [one(q) for q in qs[:n1]]
    t_one = (time.perf_counter() - t0) / n1                     # seconds per query on one core
    t0 = time.perf_counter()
    O.Sketch([fetch(0)]); t_sk1 = time.perf_counter() - t0       # seconds per sketch on one core
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        hall = list(ex.map(one, qs))
    t_q_all = time.perf_counter() - t0
    secs_1core = n_refs * t_sk1 + n_refs * t_one                # extrapolation to the full n_refs x n_refs job
    secs_all = t_sketch_all + t_q_all * (n_refs / len(qs))
    pairs = float(n_refs) * n_refs
    return {"value": pairs / secs_1core, "unit": "genome-pairs/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"{len(qs)} of {n_refs} queries against all {n_refs} references ({sum(hall)} chained hits), every reference sketched once; "
                      f"extrapolated linearly in queries to {n_refs} x {n_refs}; one core: {t_sk1 * 1e3:.1f} ms per sketch, {t_one:.2f} s per query ({n1} queries timed)",
            "all_cores": {"value": pairs / secs_all, "cores": threads, "seconds_extrapolated": secs_all,
                          "measured": {"sketch_all_refs_s": t_sketch_all, "queries_s": t_q_all, "queries": len(qs)}},
            "note": PORT_NOTE}


def cpu_baseline_metagenome(fetch_ref, n_cpu_refs, n_refs, contigs, threads, faster_small):
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
    [one(c) for c in contigs[:n1]]
    t_one = (time.perf_counter() - t0) / n1
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        hall = list(ex.map(one, contigs))
    t_all = time.perf_counter() - t0
    return {"value": 1.0 / t_one, "unit": "queries/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"{len(contigs)} of the contigs, each its own query against a database of the first {n_cpu_refs} of the {n_refs} references "
                      f"(the CPU figure is flattered by the smaller database); {sum(hall)} hits; one core: {t_one * 1e3:.1f} ms per query ({n1} timed); "
                      f"database sketched in {t_db:.1f} s on {threads} threads (not counted)",
            "all_cores": {"value": len(contigs) / t_all, "cores": threads, "seconds": t_all},
            "note": PORT_NOTE}


def api_rates(psk, genomes, query):
    """The drop-in path a pyskani user calls, from ASCII in HOST memory (SURVEY.md §8d 'Metric'):
    (a) n x Database.sketch(name, bytes) + one Database.query(name, bytes); (b) Database.sketch_many + query with the bytes crossing PCIe
    as ASCII (PSK_INGEST_PACKED=0); (c) the same with the ingest worker threads packing 2 bits per base (the default: L / 4 bytes over PCIe)."""
    out = {}
    n = len(genomes)
    modes = (("api", False, None), ("host_ascii", True, "0"), ("host_packed", True, "1"))
    for label, bulk, packed in modes + modes:   # second pass = warm
        if packed is None:
            os.environ.pop("PSK_INGEST_PACKED", None)
        else:
            os.environ["PSK_INGEST_PACKED"] = packed
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
    os.environ.pop("PSK_INGEST_PACKED", None)
    return out


# ------------------------------------------------------------------ workloads
class Job:
    """what every workload runner needs: torch, the device, the distributed state, the arguments"""

    def __init__(self, torch, device, local_rank, rank, world, dist, coll_device, args):
        self.torch, self.device, self.local_rank, self.rank, self.world, self.dist, self.coll_device, self.args = torch, device, local_rank, rank, world, dist, coll_device, args
        self.pmc = load_pmc()

    def fence(self, eng):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        eng.sync()

    def max_over_ranks(self, dt):
        if self.world > 1:
            t = self.torch.tensor([dt], device=self.coll_device, dtype=self.torch.float64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return dt


def run_search(job, steps, warmup, n_refs, cpu_sample, with_api):
    """BASELINE configs[1]: 1 query vs n_refs references per GPU (weak scaling for N>1: every rank holds its own n_refs)."""
    torch, rank, world = job.torch, job.rank, job.world
    anc_lens, fam_of = family_layout(2, n_refs, N_FAMILIES)
    buf, offs, lens = make_genomes(torch, job.device, 2, 1000 * rank + 3, list(range(n_refs)), fam_of, anc_lens, query_family=0)
    torch.cuda.synchronize()
    eng = Engine(job.local_rank)
    names = (C.c_char_p * n_refs)(*[f"r{rank}_{i}".encode() for i in range(n_refs)])
    c_off, c_len, gfc, n = eng.layout(offs, lens)
    comm = None
    if world > 1:
        from pyskani_amd import parallel
        if job.args.comm == "capi":
            from pyskani_amd.database import Context
            shim = Context.__new__(Context); shim._lib, shim._h = eng.lib, eng.ctx
            comm = parallel.CapiComm(shim, job.dist)
        else:
            comm = parallel.TorchComm(job.dist, None, job.coll_device)

    sent = {"bytes": 0}

    def step():
        out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, n)
        db = eng.make_db(names, out, n - 1)
        try:
            recs = eng.query_one(db, out[n - 1])
        finally:
            eng.lib.psk_sketch_free(out[n - 1])
            eng.lib.psk_db_destroy(db)
        if comm is not None:   # exchange step: all-gather of the per-shard hit records (RCCL over xGMI), global reference indices
            b0 = comm.bytes_sent
            recs["ref_index"] += rank * n_refs
            recs, _ = comm.gather_hit_records(recs)
            sent["bytes"] = comm.bytes_sent - b0
        return len(recs)

    dt, n_hits, kern, work, clock = timed_loop(eng, step, steps, warmup, lambda: job.fence(eng))
    dt = job.max_over_ranks(dt)
    line = None
    if rank == 0:
        bases = float(sum(lens))
        table = kernel_rooflines(kern, steps, {"bases": bases, "c": 125, "marker_c": 1000, **work}, job.pmc.get("search", {}))
        roof = roofline_of(table, steps, prefer="sketch_scan")
        pm = job.pmc.get("search", {}).get("sketch_scan")
        if pm and "valu_per_base" in pm and roof:
            # VALU issue: SQ_INSTS_VALU per base measured offline (26.8 wave-instructions per 64 bases, 61 % four-cycle and 39 % two-cycle by
            # profiles/micro/valu_rates.hip = 3.2 cycles on average), launch duration measured live, clock probed live
            insts = pm["valu_per_base"] * bases / max(1.0, table["sketch_scan"]["launches_per_step"])
            avg_s = roof["avg_launch_ms"] * 1e-3
            roof["valu"] = {"valu_issue_frac_at_probed_clock": insts * 3.2 / (avg_s * clock["shader_clock_mhz"] * 1e6 * 1024) if avg_s > 0 and clock["shader_clock_mhz"] > 0 else None,
                            "valu_issue_frac_at_2p4GHz": insts * 3.2 / (avg_s * 2.4e9 * 1024) if avg_s > 0 else None,
                            "valu_wave_instructions_per_launch": insts,
                            "source": f"SQ_INSTS_VALU measured offline ({pm['source']}), cycle classes from profiles/micro/valu_rates.hip; launch duration and clock measured live"}
            roof["note"] = ("priced against HBM as the contract asks; the kernel's real roof is integer VALU issue (one 64-bit mix per base: "
                            "15 four-cycle + 7 two-cycle wave instructions per base, DESIGN.md section 4): see valu")
        line = {
            "metric": "genome-pairs/sec (sketch+ANI)", "value": n_refs * world * steps / dt, "unit": "genome-pairs/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"1 query vs {n_refs} synthetic ~5 Mb refs per GPU (10 families x {n_refs // N_FAMILIES}), c=125 marker_c=1000 k=15; device-resident ASCII in, hit list on host out (the same step from ASCII in host memory, packed ingest: host_to_host)",
                       "refs_per_gpu": n_refs, "hits": int(n_hits), "parallelism": f"refs sharded over {world} GPU(s)" + (f", hit lists all-gathered ({job.args.comm})" if world > 1 else "")},
            "roofline": roof, "clock": clock,
            "kernel_ms_per_step": {k: table[k]["ms_per_step"] for k in KERNELS},
            "extras": {"genomes_sketched_per_s": (n_refs + 1) * world * steps / dt, "bases_sketched_per_s": bases * world * steps / dt,
                       "reported_hits_per_step": int(n_hits), "chain_work_per_step": work},
        }
        if comm is not None:
            line["extras"]["exchange"] = {"ranks": int(job.dist.get_world_size()), "backend": job.args.backend, "bytes_sent_per_rank_last_step": int(sent["bytes"])}
        if world == 1 and (cpu_sample > 0 or with_api):
            host = buf.cpu().numpy()
            fetch = lambda i: host[offs[i]:offs[i] + lens[i]].tobytes()
            if with_api:
                import pyskani_amd as psk
                r = api_rates(psk, [fetch(i) for i in range(n_refs)], fetch(-1))
                psk.database.release_default_context(job.local_rank)
                hp = r["host_packed"]
                line["host_to_host"] = {"value": hp["pairs_per_s"], "unit": "genome-pairs/s", "ms_per_step": n_refs / hp["pairs_per_s"] * 1e3, "ingest": "packed (2 bits per base over PCIe)",
                                        "note": "the contract's timed region (SURVEY.md 8d): ASCII contigs in HOST memory -> Database.sketch_many + Database.query -> hit list on host; `value` above is the same step from device-resident ASCII"}
                line["extras"].update(api_pairs_per_s=r["api"]["pairs_per_s"], host_ascii_pairs_per_s=r["host_ascii"]["pairs_per_s"], host_packed_pairs_per_s=r["host_packed"]["pairs_per_s"],
                                      api_detail=r["api"], host_ascii_detail=r["host_ascii"], host_packed_detail=r["host_packed"],
                                      api_note="same 1 query vs refs workload from ASCII bytes in HOST memory through pyskani_amd.Database: "
                                               "api = n x sketch() + query(); host_ascii = sketch_many() (pinned, double-buffered H2D pipeline, ASCII over PCIe) + query(); "
                                               "host_packed = the same with the ingest threads packing 2 bits per base (the library's default for genomes of long contigs)")
            if cpu_sample > 0:
                line["cpu_baseline"] = cpu_baseline_search(fetch, min(cpu_sample, n_refs), os.cpu_count() or 1)
                if "host_to_host" in line:
                    line["host_to_host"]["vs_cpu_1_core"] = line["host_to_host"]["value"] / line["cpu_baseline"]["value"]
                    if "all_cores" in line["cpu_baseline"]:
                        line["host_to_host"]["vs_cpu_all_cores"] = line["host_to_host"]["value"] / line["cpu_baseline"]["all_cores"]["value"]
            del host
    if comm is not None and hasattr(comm, "close"):
        comm.close()
    del buf
    eng.close()
    torch.cuda.empty_cache()
    return line


def records_digest(recs):
    """digest of a hit-record array sorted by (query, reference): the query and reference indices and the bits of the three floats - the fields of psk_hit_min, which
    both record kinds carry (psk_hit: the query index in `reserved`). ANI and both aligned fractions are functions of every chaining integer, which the oracle
    checks compare one by one on the raw records."""
    h = hashlib.sha256()
    q = (recs["query"] & np.uint32(0x7FFFFFFF)) if "query" in recs.dtype.names else recs["reserved"]
    h.update(np.ascontiguousarray(q, dtype=np.uint32).tobytes())
    for f in ("ref_index", "ani", "af_query", "af_ref"):
        h.update(np.ascontiguousarray(recs[f]).tobytes())
    return h.hexdigest()[:16]


def run_allvsall(job, steps, warmup, n_total, cpu_queries, variant="plain", verify_hits=0, host_leg=False, model_ranks=()):
    """BASELINE configs[2] shape: every genome against a database of all of them (families of 100). N=1: one psk_query_many. N>1: the
    FIXED job of n_total genomes is sharded over the ranks (strong scaling) and run through parallel.ShardedDatabase.all_vs_all_records."""
    torch, rank, world, args = job.torch, job.rank, job.world, job.args
    n_families = max(1, n_total // 100)
    anc_lens, fam_of = family_layout(3, n_total, n_families)
    from pyskani_amd.parallel import shard_bounds
    lo, hi = shard_bounds(n_total, rank, world)
    n_local = hi - lo
    gfc_list = None
    if variant == "contigs":
        buf, offs, lens, gfc_list = make_genomes(torch, job.device, 3, 31, list(range(lo, hi)), fam_of, anc_lens, variant=variant)
    else:
        buf, offs, lens = make_genomes(torch, job.device, 3, 31, list(range(lo, hi)), fam_of, anc_lens, variant=variant)
    torch.cuda.synchronize()
    all_names = [f"g{i}" for i in range(n_total)]
    bases_local = float(sum(lens))
    line = None
    if variant != "plain" and world != 1:
        raise SystemExit("the generator variants run at N=1")
    if world == 1:
        eng = Engine(job.local_rank)
        names = (C.c_char_p * n_total)(*[s.encode() for s in all_names])
        c_off, c_len, gfc, n = eng.layout(offs, lens, gfc_list)
        last = {}

        def step():
            out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, n)
            db = eng.make_db(names, out, n)
            try:
                nh, (recs, qoffs) = eng.query_many(db, out, n, keep=True)
                last["recs"], last["offs"] = recs, qoffs
                return nh
            finally:
                eng.lib.psk_db_destroy(db)
        dt, n_hits, kern, work, clock = timed_loop(eng, step, steps, warmup, lambda: job.fence(eng))
        step_ms = list(getattr(timed_loop, "last_step_ms", []))      # every timed step's own wall time (the line's ms_per_step is their mean)
        recs = last["recs"]      # psk_hit_min: `query` = the hit's query within the call
        digest = records_digest(recs)
        # The timed steps keep two batches in flight on two lanes (query_many.hip: rounds of >= 2^31 (pair, seed) items): their kernels share the chip and a bracket's
        # duration says little about the kernel. The per-kernel table comes from ONE more step run as a single chain of launches (PSK_PIPELINE=0), outside the timed region.
        table_step = None
        if "PSK_PIPELINE" not in os.environ:
            os.environ["PSK_PIPELINE"] = "0"
            try:
                dt1, _, kern, work, _ = timed_loop(eng, step, 1, 0, lambda: job.fence(eng))
                if records_digest(last["recs"]) != digest:
                    raise SystemExit("all-vs-all: the one-chain step and the two-lane steps disagree")
                table_step = {"mode": "one chain of launches (PSK_PIPELINE=0), one step outside the timed region", "ms_per_step": dt1 * 1e3}
            finally:
                del os.environ["PSK_PIPELINE"]
        table = kernel_rooflines(kern, 1 if table_step else steps, {"bases": bases_local, "c": 125, "marker_c": 1000, **work}, job.pmc.get("allvsall10k" if (n_total >= 10000 and "allvsall10k" in job.pmc) else "allvsall", {}))      # (the 10 000 x 10 000 job has a counter pass of its own)
        shape = {"plain": "single-contig genomes, substitutions only", "contigs": "every genome cut into 1-80 contigs",
                 "sv": "20 block inversions / translocations of 5-50 kb per genome on top of the substitutions"}[variant]
        line = {"ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "value": float(n_total) * n_total * steps / dt, "unit": "genome-pairs/s",
                "workload": f"all-vs-all {n_total} x {n_total} synthetic ~5 Mb genomes on one GPU ({n_families} families x {n_total // n_families}; {shape}), c=125 marker_c=1000 k=15; device-resident ASCII",
                "hits": int(n_hits), "hits_digest": digest, "chain_work_per_step": work, "bases_sketched_per_s": bases_local * steps / dt,
                "roofline": roofline_of(table, 1 if table_step else steps), "kernel_roofline": table, "clock": clock, "scaling": "strong", "step_ms": step_ms}
        if table_step:
            line["kernel_table_step"] = table_step
        host = None
        if (cpu_queries > 0 and variant == "plain") or host_leg:
            try:
                host = buf.cpu().numpy()
            except (RuntimeError, MemoryError) as e:      # (50 GB of pageable host memory for 10 000 genomes)
                line["host_to_host"] = {"skipped": f"no host copy of the genomes: {e}"[:200]}
        if cpu_queries > 0 and variant == "plain" and host is not None:
            line["cpu_baseline"] = cpu_baseline_allvsall(lambda i: host[offs[i]:offs[i] + lens[i]].tobytes(), n_total, min(cpu_queries, n_total), os.cpu_count() or 1)
        if host_leg and host is not None:
            # The contract's timed region for this job: ASCII contigs in HOST memory -> hit list on host. The genomes cross PCIe through the
            # library's ingest pipeline (worker threads pack 2 bits per base into pinned slots, L/4 bytes per genome over PCIe, sketch kernels on the previous
            # sub-batch), then the database is loaded and queried as in the device-resident step. The two halves do not overlap: an all-vs-all query needs every
            # reference in the database before its first round.
            ptrs = (C.c_char_p * n)(*[C.cast(C.c_void_p(host.ctypes.data + offs[i]), C.c_char_p) for i in range(n)])
            hl = (C.c_uint64 * n)(*[int(x) for x in lens])
            hg = (C.c_uint32 * (n + 1))(*range(n + 1))
            best = None
            for _ in range(2):      # (the second pass is the warm one: pinned slots and device blocks exist)
                outh = (C.c_void_p * n)()
                eng.sync()
                t0 = time.perf_counter()
                eng.capi.check(eng.lib.psk_sketch_many_host(eng.ctx, C.byref(eng.params), ptrs, hl, hg, n, 1, outh))
                t1 = time.perf_counter()
                db = eng.make_db(names, outh, n)
                try:
                    nh2, (recs2, _) = eng.query_many(db, outh, n, keep=True)
                finally:
                    eng.lib.psk_db_destroy(db)
                t2 = time.perf_counter()
                best = {"ms_per_step": (t2 - t0) * 1e3, "value": float(n_total) * n_total / (t2 - t0), "unit": "genome-pairs/s", "ingest": "packed (2 bits per base over PCIe)",
                        "ingest_s": t1 - t0, "ingest_host_GBps": float(sum(lens)) / (t1 - t0) / 1e9, "query_s": t2 - t1, "hits": int(nh2), "hits_digest": records_digest(recs2)}
            assert best["hits_digest"] == digest, "the host-memory run and the device-resident run disagree"
            if line.get("cpu_baseline"):
                best["vs_cpu_1_core"] = best["value"] / line["cpu_baseline"]["value"]
                if "all_cores" in line["cpu_baseline"]:
                    best["vs_cpu_all_cores"] = best["value"] / line["cpu_baseline"]["all_cores"]["value"]
            line["host_to_host"] = best
        del host
        if verify_hits > 0 and len(recs):      # outside the timed region: the same step once more with the 80-byte records, random hits recomputed by the CPU oracle, every chain integer compared
            out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, n)
            db = eng.make_db(names, out, n)
            try:
                _, (raw, roffs) = eng.query_many(db, out, n, keep=True, raw=True)
            finally:
                eng.lib.psk_db_destroy(db)
            raw["reserved"] = np.repeat(np.arange(n, dtype=np.uint32), np.diff(roffs))
            assert records_digest(raw) == digest, "the 80-byte and the 20-byte records of the same step disagree"
            line["oracle_check"] = allvsall_verify(buf, offs, lens, gfc_list, raw, verify_hits)
        if model_ranks and variant == "plain":
            line["scaling_model"] = scaling_model(eng, buf, offs, lens, n_total, args.exchange_batch, line["ms_per_step"], tuple(model_ranks))
        eng.close()
    else:
        import pyskani_amd as psk
        from pyskani_amd.parallel import ShardedDatabase
        # the timed step builds a fresh local Database every time (nothing cached): sketch the shard, load it, exchange + query
        from pyskani_amd import parallel
        from pyskani_amd.database import default_context
        ctx = default_context(job.local_rank)
        comm = parallel.CapiComm(ctx, job.dist) if args.comm == "capi" else parallel.TorchComm(job.dist, None, job.coll_device)      # made once, reused by every step
        state = {}

        def step():
            sent0 = comm.bytes_sent
            db = psk.Database(device=job.local_rank)
            db.sketch_many_device(all_names[lo:hi], buf.data_ptr(), offs, lens)
            sdb = ShardedDatabase(job.dist, local=db, device=job.device, comm=comm)
            sdb.adopt_local(all_names)
            recs = sdb.all_vs_all_records(batch=args.exchange_batch, learned_ani=False)
            state["stats"], state["bytes_sent"], state["recs"] = dict(sdb.stats), comm.bytes_sent - sent0, recs
            return len(recs)
        eng = Engine.__new__(Engine)      # the timers / counters of the Database's own context
        from pyskani_amd import _capi
        eng.capi, eng.lib, eng.ctx, eng.hit_dtype = _capi, ctx._lib, ctx._h, np.dtype(_capi.Hit)
        dt, n_hits, kern, work, clock = timed_loop(eng, step, steps, warmup, lambda: job.fence(eng))
        dt = job.max_over_ranks(dt)
        if rank == 0:
            st = state["stats"]
            other = st["total_s"] - st["psk_s"] - st.get("exposed_collective_s", st["collective_s"])      # (the gathers of later rounds run beside psk_s on a helper thread: only what the calling thread waited for is exposed)
            table = kernel_rooflines(kern, steps, {"bases": bases_local, "c": 125, "marker_c": 1000, **work}, job.pmc.get("allvsall", {}))
            line = {"ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "value": float(n_total) * n_total * steps / dt, "unit": "genome-pairs/s",
                    "workload": f"all-vs-all {n_total} x {n_total} synthetic ~5 Mb genomes sharded over {world} GPU(s) ({n_families} families), c=125 marker_c=1000 k=15, device-resident ASCII; "
                                f"query side = all-gather of the shards' packed sketch records, {args.exchange_batch} genomes per rank and round; hit records all-gathered once ({args.comm})",
                    "hits": int(n_hits), "hits_digest": records_digest(state["recs"]), "scaling": "strong",
                    "exchange": {"ranks": int(job.dist.get_world_size()), "backend": args.backend, "record_bytes": int(state["recs"].dtype.itemsize), "overlap": True, "bytes_sent_per_rank_last_step": int(state["bytes_sent"]), "collective_s_last_step": st["collective_s"], "exposed_collective_s_last_step": st.get("exposed_collective_s"), "psk_s_last_step": st["psk_s"],
                                 "python_s_last_step": other, "all_vs_all_s_last_step": st["total_s"],
                                 "outside_psk_and_collectives_frac": other / st["total_s"] if st["total_s"] > 0 else None,
                                 "note": "rank 0's split of the last step's ShardedDatabase.all_vs_all_records call (sketching the shard and loading the database come before it)"},
                    "roofline": roofline_of(table, steps), "kernel_roofline": table, "clock": clock, "chain_work_per_step_rank0": work}
        if hasattr(comm, "close"):
            comm.close()
    del buf
    torch.cuda.empty_cache()
    return line


XGMI_LINK_GBS = 153.0      # MI355X_MICROARCH.md: 7 point-to-point xGMI links per GPU, ~153 GB/s each


def scaling_model(eng, buf, offs, lens, n_total, batch, t1_ms, ranks=(2, 4, 8)):
    """A SINGLE-GPU EMULATION of one rank's share of the N-way strong-scaling job - not a measurement of N GPUs (none was available to this build). For N in `ranks` this
    GPU plays rank 0: it sketches rank 0's shard (n_total / N genomes), loads it as the local database and queries it with every round ShardedDatabase.all_vs_all_records
    would hand it (parallel.py: `batch` genomes of EVERY rank per round, rank-major; the other ranks' sketches are sketched here beforehand, untimed - on N GPUs they arrive
    over xGMI while the previous round is queried). Reported per N: the rank's time (sketch + load + rounds), its work counters, the bytes it would send / receive, the
    exchange time those bytes cost over direct xGMI links (overlapped with the queries except for the first round), and the implied speed-up against the measured N = 1 step.
    The job's families are consecutive in insertion order, so a rank chains 1 / N of the pairs and only its own genomes' seeds are looked up (a query without a passing
    reference in the shard costs its share of the marker screen and nothing else): what does not shrink with N is the screen of all n_total queries and the per-round set-up."""
    capi, lib = eng.capi, eng.lib
    n_all = len(offs)
    c_off, c_len, gfc, n = eng.layout(offs, lens)
    everyone = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, n)      # stand-ins for the gathered sketches
    pack = np.zeros(n_all, np.int64)
    sz = C.c_uint64()
    for i in range(n_all):
        capi.check(lib.psk_sketch_pack_size(C.c_void_p(everyone[i]), C.byref(sz)))
        pack[i] = sz.value
    out = {"kind": "single-GPU emulation of rank 0's share (NOT a multi-GPU measurement)", "batch": batch, "n1_measured_ms": t1_ms, "ranks": {}}
    from pyskani_amd.parallel import shard_bounds
    try:
        for N in ranks:
            bounds = [shard_bounds(n_total, r, N) for r in range(N)]
            lo, hi = bounds[0]
            m = hi - lo
            s_off, s_len, s_gfc, _ = eng.layout(offs[lo:hi], lens[lo:hi])
            names = (C.c_char_p * m)(*[f"g{i}".encode() for i in range(lo, hi)])
            per_batch = {}
            for bt in sorted({batch, max(b1 - b0 for b0, b1 in bounds)}):
                rounds = (max(b1 - b0 for b0, b1 in bounds) + bt - 1) // bt
                best = None
                for rep in range(2):      # (the second pass is the warm one)
                    eng.sync(); eng.work(reset=True)
                    t0 = time.perf_counter()
                    mine = eng.sketch_device_c(buf.data_ptr(), s_off, s_len, s_gfc, m)
                    eng.sync(); t1 = time.perf_counter()
                    db = eng.make_db(names, mine, m)
                    t2 = time.perf_counter()
                    hits, round_s, recv = 0, [], 0
                    try:
                        for b in range(rounds):
                            ids = [i for (b0, b1) in bounds for i in range(min(b0 + b * bt, b1), min(b0 + (b + 1) * bt, b1))]
                            if not ids:
                                continue
                            hs = (C.c_void_p * len(ids))(*[everyone[i] for i in ids])
                            tr = time.perf_counter()
                            hits += eng.query_many(db, hs, len(ids))
                            round_s.append(time.perf_counter() - tr)
                            recv += int(pack[[i for i in ids if not (lo <= i < hi)]].sum())
                    finally:
                        lib.psk_db_destroy(db)
                    t3 = time.perf_counter()
                    w = eng.work(reset=True)
                    cur = {"rank_ms": (t3 - t0) * 1e3, "sketch_ms": (t1 - t0) * 1e3, "db_load_ms": (t2 - t1) * 1e3, "rounds": rounds, "rounds_ms": sum(round_s) * 1e3, "first_round_ms": round_s[0] * 1e3 if round_s else 0.0,
                           "hits_rank": int(hits), "pairs_chained": int(w["chained_pairs"]), "index_lookups": int(w["index_lookups"]), "index_entries_visited": int(w["index_entries_visited"]), "anchors": int(w["anchors"]),
                           "bytes_sent": int(pack[lo:hi].sum()) + 20 * int(hits), "bytes_received": recv + 20 * int(hits) * (N - 1),
                           "exchange_ms_model": (recv / max(1, N - 1)) / (XGMI_LINK_GBS * 1e9) * 1e3, "first_round_exchange_ms_model": (recv / max(1, N - 1) / max(1, rounds)) / (XGMI_LINK_GBS * 1e9) * 1e3}
                    if best is None or cur["rank_ms"] < best["rank_ms"]:
                        best = cur
                best["speedup_vs_1"] = t1_ms / (best["rank_ms"] + best["first_round_exchange_ms_model"])
                best["efficiency"] = best["speedup_vs_1"] / N
                per_batch[bt] = best
            pick = min(per_batch, key=lambda k: per_batch[k]["rank_ms"])
            e = dict(per_batch[batch], batch=batch)
            if pick != batch:
                e["best_batch"] = dict(per_batch[pick], batch=pick)
            out["ranks"][str(N)] = e
    finally:
        lib.psk_sketch_free_many(everyone, n)
    out["note"] = ("exchange_ms_model: the sketches a rank receives from ONE peer over that peer's direct xGMI link at %.0f GB/s (all peers send at once); only the first round's share is "
                   "exposed (round b + 1 travels while round b is queried), which speedup_vs_1 adds to rank_ms. The hit all-gather (20 B per hit) is latency-bound and not priced." % XGMI_LINK_GBS)
    return out


def allvsall_verify(buf, offs, lens, gfc, recs, k):
    """k random hits of an all-vs-all step -> both genomes to the host -> oracle sketches -> oracle.chain: the GPU record must carry the same integers"""
    from oracle import oracle as O
    O.build()
    pick = np.random.default_rng(5).choice(len(recs), size=min(k, len(recs)), replace=False)
    def contigs_of(g):
        rng_c = range(gfc[g], gfc[g + 1]) if gfc is not None else (g,)
        return [buf[offs[c]:offs[c] + lens[c]].cpu().numpy().tobytes() for c in rng_c]
    fields = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")
    checked = []
    for i in pick:
        h = recs[int(i)]
        q, r = int(h["reserved"]), int(h["ref_index"])
        want = O.chain(O.Sketch(contigs_of(r)), O.Sketch(contigs_of(q)))
        for f in fields:
            assert int(h[f]) == int(getattr(want, f)), (q, r, f, int(h[f]), int(getattr(want, f)))
        assert abs(float(h["ani"]) - want.ani) < 1e-6 and abs(float(h["af_query"]) - want.af_query) < 1e-6 and abs(float(h["af_ref"]) - want.af_ref) < 1e-6
        checked.append({"query": q, "ref": r, "ani": float(h["ani"]), "n_anchors": int(h["n_anchors"])})
    return {"pairs": checked, "fields": list(fields) + ["ani (1e-6)", "af_query (1e-6)", "af_ref (1e-6)"], "result": f"{len(checked)} random hits bit-exact"}


def run_metagenome(job, steps, warmup, n_refs, n_queries, settings, cpu_contigs, api_queries):
    """BASELINE configs[3]: n_queries contigs against a RESIDENT database of n_refs references, c=30 marker_c=200. One entry per
    `faster_small` setting in `settings`; the database and the contigs are built once."""
    torch, args = job.torch, job.args
    n_families = max(1, n_refs // 100)
    anc_lens, fam_of = family_layout(4, n_refs, n_families)
    buf, offs, lens = make_genomes(torch, job.device, 4, 41 + 1000 * job.rank, list(range(n_refs)), fam_of, anc_lens)
    torch.cuda.synchronize()
    eng = Engine(job.local_rank, 30, 200)
    names = (C.c_char_p * n_refs)(*[f"r{i}".encode() for i in range(n_refs)])
    t0 = time.perf_counter()
    handles = eng.sketch_device(buf.data_ptr(), offs, lens)
    db = eng.make_db(names, handles, n_refs)
    eng.sync()
    db_build_s = time.perf_counter() - t0
    cbuf, coffs, clens = make_contigs(torch, job.device, buf, offs, lens, n_refs, n_queries, seed=4 + job.rank)
    torch.cuda.synchronize()
    c_off, c_len, gfc, nq = eng.layout(coffs, clens)
    bases = float(sum(clens))
    out = {}
    for faster_small in settings:
        def step():   # sketch every contig, query them all against the resident database
            qh = eng.sketch_device_c(cbuf.data_ptr(), c_off, c_len, gfc, nq)
            try:
                return eng.query_many(db, qh, nq, faster_small)
            finally:
                eng.lib.psk_sketch_free_many(qh, nq)
        dt, n_hits, kern, work, clock = timed_loop(eng, step, steps, warmup, lambda: job.fence(eng))
        dt = job.max_over_ranks(dt)
        table = kernel_rooflines(kern, steps, {"bases": bases, "c": 30, "marker_c": 200, **work}, job.pmc.get("metagenome", {}))
        entry = {"ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "value": n_queries * job.world * steps / dt, "unit": "queries/s",
                 "pairs_per_s": float(n_queries) * n_refs * job.world * steps / dt,
                 "workload": f"metagenome: {n_queries} contigs (2-50 kb, log-uniform, 0-5 % divergence) vs a resident database of {n_refs} synthetic ~5 Mb refs ({n_families} families), "
                             f"c=30 marker_c=200 k=15, faster_small={faster_small}, device-resident ASCII; a step = sketch every contig + psk_query_many + hits to host",
                 "hits": int(n_hits), "chain_work_per_step": work, "db_build_s": db_build_s,
                 "roofline": roofline_of(table, steps), "kernel_roofline": table, "clock": clock}
        if job.world == 1 and cpu_contigs > 0 and job.rank == 0:
            n_cpu_refs = min(n_refs, 500)
            href = buf[:offs[n_cpu_refs - 1] + lens[n_cpu_refs - 1]].cpu().numpy()
            k = min(cpu_contigs, n_queries)
            chost0 = cbuf[:coffs[k - 1] + clens[k - 1]].cpu().numpy()
            sample = [chost0[coffs[i]:coffs[i] + clens[i]].tobytes() for i in range(k)]
            entry["cpu_baseline"] = cpu_baseline_metagenome(lambda i: href[offs[i]:offs[i] + lens[i]].tobytes(), n_cpu_refs, n_refs, sample, os.cpu_count() or 1, faster_small)
            del href, chost0
        out["faster_small" if faster_small else "rescue"] = entry
    eng.lib.psk_db_destroy(db)
    eng.close()      # the resident database, its indexes, probe tables and scratch go before the API leg builds its own Database
    if job.world == 1 and api_queries > 0 and job.rank == 0:
        out["api"] = metagenome_api(job, buf, offs, lens, n_refs, cbuf, coffs, clens, min(api_queries, n_queries), settings[0])
    del buf, cbuf
    torch.cuda.empty_cache()
    return out


def metagenome_api(job, buf, offs, lens, n_refs, cbuf, coffs, clens, nq, faster_small):
    """the same contigs ONE AT A TIME through the pyskani-shaped API, from host bytes: Database.query(name, contig) — the call SURVEY.md
    §8d config 4 specifies — from one host thread and from eight (the reference's query() releases the GIL: threads are its route to
    concurrency; here every thread's call runs on its own lane of the context), and one Database.query_many for all of them"""
    import threading
    import pyskani_amd as psk
    host = buf.cpu().numpy()
    pdb = psk.Database(compression=30, marker_compression=200, device=job.local_rank)
    refs = [(f"r{i}", host[offs[i]:offs[i] + lens[i]].tobytes()) for i in range(n_refs)]      # (the bytes objects a caller would hold: made outside the timed call)
    t0 = time.perf_counter()
    pdb.sketch_many(refs)
    t_load = time.perf_counter() - t0
    del host, refs
    chost = cbuf[:coffs[nq - 1] + clens[nq - 1]].cpu().numpy()
    contigs = [chost[coffs[i]:coffs[i] + clens[i]].tobytes() for i in range(nq)]
    for c in contigs[:20]:
        pdb.query("w", c, learned_ani=False, faster_small=faster_small)
    t0 = time.perf_counter()
    nh = sum(len(pdb.query(f"c{i}", c, learned_ani=False, faster_small=faster_small)) for i, c in enumerate(contigs))
    t_q = time.perf_counter() - t0

    def _work(lo, hi):
        for i in range(lo, hi):
            pdb.query(f"c{i}", contigs[i], learned_ani=False, faster_small=faster_small)
    th = [threading.Thread(target=_work, args=(k * nq // 8, (k + 1) * nq // 8)) for k in range(8)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t_q8 = time.perf_counter() - t0
    many = pdb.query_many([(f"c{i}", c) for i, c in enumerate(contigs)], learned_ani=False, faster_small=faster_small)
    assert sum(len(h) for h in many) == nh
    t0 = time.perf_counter()
    pdb.query_many([(f"c{i}", c) for i, c in enumerate(contigs)], learned_ani=False, faster_small=faster_small)
    t_many = time.perf_counter() - t0
    del pdb, many
    import gc
    gc.collect()
    psk.database.release_default_context(job.local_rank)      # its pool would keep tens of GB the next workload needs
    return {"queries": nq, "api_queries_per_s": nq / t_q, "api_query_ms": t_q / nq * 1e3, "api_queries_per_s_8_threads": nq / t_q8, "api_query_many_per_s": nq / t_many,
            "api_db_load_s": t_load, "api_hits": nh, "faster_small": faster_small,
            "note": f"{nq} contigs from host bytes through pyskani_amd.Database: one Database.query() per contig (from one host thread, and from eight), and one Database.query_many() for all of them"}


def run_mammalian(job, steps, warmup, n_genomes, contig_mb, verify_pairs):
    """BASELINE configs[4] shape at reduced count: all-vs-all of n_genomes genomes of 24 x contig_mb Mb contigs (families of 4). Outside
    the timed region, `verify_pairs` of the chained pairs are recomputed by the CPU oracle and every integer of the chain is compared;
    the oracle's time for them is the CPU baseline."""
    torch = job.torch
    buf, offs, lens, gfc = make_big_genomes(torch, job.device, n_genomes, 24, contig_mb * 1_000_000, 4, seed=5 + job.rank)
    torch.cuda.synchronize()
    eng = Engine(job.local_rank)
    names = (C.c_char_p * n_genomes)(*[f"m{i}".encode() for i in range(n_genomes)])
    c_off, c_len, c_gfc, n = eng.layout(offs, lens, gfc)
    last = {}

    def step():     # sketch every genome, load the database, every genome against all of them
        handles = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, c_gfc, n)
        db = eng.make_db(names, handles, n)
        try:
            nh, (recs, qoffs) = eng.query_many(db, handles, n, keep=True, raw=True)      # (a few dozen hits: the oracle check below reads their chaining integers)
            last["recs"], last["offs"] = recs, qoffs
            return nh
        finally:
            eng.lib.psk_db_destroy(db)
    dt, n_hits, kern, work, clock = timed_loop(eng, step, steps, warmup, lambda: job.fence(eng))
    dt = job.max_over_ranks(dt)
    bases = float(sum(lens))
    table = kernel_rooflines(kern, steps, {"bases": bases, "c": 125, "marker_c": 1000, **work}, job.pmc.get("mammalian", {}))
    entry = {"ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "value": float(n_genomes) * n_genomes * job.world * steps / dt, "unit": "genome-pairs/s",
             "workload": f"mammalian scale (BASELINE configs[4] shape at reduced count): all-vs-all of {n_genomes} synthetic genomes of 24 x {contig_mb} Mb contigs "
                         f"({24 * contig_mb / 1000:.1f} Gb each; families of 4, substitution rates {DIVERGENCE[:4]}), c=125 marker_c=1000 k=15",
             "hits": int(n_hits), "chain_work_per_step": work, "bases_sketched_per_s": bases * job.world * steps / dt,
             "roofline": roofline_of(table, steps), "kernel_roofline": table, "clock": clock}
    if job.rank == 0 and verify_pairs > 0 and n_hits:
        entry["oracle_check"], entry["cpu_baseline"] = mammalian_verify(buf, offs, lens, gfc, last["recs"], last["offs"], n_genomes, verify_pairs, int(n_hits))
    del buf
    eng.close()
    torch.cuda.empty_cache()
    return entry


def run_mammalian_stream(job, n_genomes, contig_mb, fam_size, verify_pairs):
    """BASELINE configs[4] at its full count on ONE GPU: n_genomes genomes of 24 x contig_mb Mb (50 x 3 Gb = 150 Gb of sequence) whose ASCII is
    never resident together - every genome is built into one reused device buffer (untimed), sketched from there (timed) and dropped; what
    stays in HBM is what the path needs: the sketches and, once chained, the k-mer indexes of all n_genomes. Then one database, every genome
    against all of them (psk_query_many), hits to the host. ms_per_step = the sum of the sketch calls + database + query: one pass, steps = 1."""
    torch = job.torch
    contig_len = contig_mb * 1_000_000
    stride = (contig_len + 15 + 16) & ~15
    buf = torch.zeros(24 * stride + 64, dtype=torch.uint8, device=job.device)
    eng = Engine(job.local_rank)
    clock = eng.clock_probe()
    eng.capi.check(eng.lib.psk_ctx_set_timing(eng.ctx, 1)); eng.timing("reset"); eng.work(reset=True)
    handles = (C.c_void_p * n_genomes)()
    t_sketch, t_gen = 0.0, 0.0
    for gi in range(n_genomes):
        t0 = time.perf_counter()
        offs, lens = fill_big_genome(torch, job.device, buf, gi, 24, contig_len, fam_size, seed=5)
        torch.cuda.synchronize()
        t_gen += time.perf_counter() - t0
        t0 = time.perf_counter()
        h = eng.sketch_device(buf.data_ptr(), offs, lens, gfc=[0, 24])
        eng.sync()
        t_sketch += time.perf_counter() - t0
        handles[gi] = h[0]
    names = (C.c_char_p * n_genomes)(*[f"m{i}".encode() for i in range(n_genomes)])
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    db = eng.make_db(names, handles, n_genomes)
    nh, (recs, qoffs) = eng.query_many(db, handles, n_genomes, keep=True, raw=True)
    eng.sync()
    t_query = time.perf_counter() - t0
    free1 = torch.cuda.mem_get_info()[0]
    kern = {k: eng.timing(k) for k in KERNELS}
    work = eng.work(reset=True)
    dt = t_sketch + t_query
    bases = float(n_genomes) * 24 * contig_len
    table = kernel_rooflines(kern, 1, {"bases": bases, "c": 125, "marker_c": 1000, **work}, job.pmc.get("mammalian", {}))
    seeds = C.c_uint64(); n_seeds = 0
    for gi in range(n_genomes):
        eng.capi.check(eng.lib.psk_sketch_info(handles[gi], None, C.byref(seeds), None, None, None)); n_seeds += seeds.value
    entry = {"ms_per_step": dt * 1e3, "steps": 1, "warmup": 0, "value": float(n_genomes) * n_genomes / dt, "unit": "genome-pairs/s",
             "workload": f"mammalian scale, BASELINE configs[4] at full count on one GPU: all-vs-all of {n_genomes} synthetic genomes of 24 x {contig_mb} Mb contigs "
                         f"({24 * contig_mb / 1000:.1f} Gb each; families of {fam_size}, substitution rates {DIVERGENCE[:4]}), c=125 marker_c=1000 k=15; ASCII streamed genome by genome "
                         f"through one device buffer, sketches + k-mer indexes of all {n_genomes} resident",
             "hits": int(nh), "chain_work_per_step": work, "bases_sketched_per_s": bases / t_sketch,
             "phases_s": {"sketch_all": t_sketch, "database_and_query": t_query, "generate_untimed": t_gen},
             "resident": {"seeds": int(n_seeds), "sketch_bytes_est": int(n_seeds) * 20, "index_bytes_est": int(n_seeds) * 24,
                          "device_free_before_query_GB": free0 / 1e9, "device_free_after_query_GB": free1 / 1e9},
             "roofline": roofline_of(table, 1), "kernel_roofline": table, "clock": clock}
    eng.lib.psk_db_destroy(db)
    if verify_pairs > 0 and nh:
        # genomes 0 and 1 rebuilt (the generator is per genome), copied to the host and chained by the oracle
        contigs = {}
        for gi in (0, 1):
            offs, lens = fill_big_genome(torch, job.device, buf, gi, 24, contig_len, fam_size, seed=5)
            contigs[gi] = [buf[o:o + L].cpu().numpy().tobytes() for o, L in zip(offs, lens)]
        entry["oracle_check"], entry["cpu_baseline"] = mammalian_verify_host(contigs, recs, qoffs, n_genomes, verify_pairs, int(nh))
    del buf
    eng.close()
    torch.cuda.empty_cache()
    return entry


def mammalian_verify(buf, offs, lens, gfc, recs, qoffs, n_genomes, verify_pairs, n_hits):
    """Two genomes of one family -> host -> oracle sketches -> oracle.chain for the ordered pairs between them; the hits of the last
    timed step must carry the same integers. Returns (check record, cpu_baseline extrapolated from the oracle's times)."""
    qa, qb = 0, 1                       # two members of family 0
    contigs = {g: [buf[offs[c]:offs[c] + lens[c]].cpu().numpy().tobytes() for c in range(gfc[g], gfc[g + 1])] for g in (qa, qb)}
    return mammalian_verify_host(contigs, recs, qoffs, n_genomes, verify_pairs, n_hits)


def mammalian_verify_host(contigs, recs, qoffs, n_genomes, verify_pairs, n_hits):
    """contigs = {0: [...], 1: [...]}: two genomes of one family as host bytes."""
    from oracle import oracle as O
    O.build()
    qa, qb = 0, 1
    t0 = time.perf_counter()
    sk = {g: O.Sketch(contigs[g]) for g in (qa, qb)}
    t_sketch = (time.perf_counter() - t0) / 2
    del contigs
    fields = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")
    checked, t_chain = [], 0.0
    for q, r in ((qa, qb), (qb, qa))[:verify_pairs]:
        t0 = time.perf_counter()
        want = O.chain(sk[r], sk[q])
        t_chain += time.perf_counter() - t0
        mine = recs[qoffs[q]:qoffs[q + 1]]
        hit = mine[mine["ref_index"] == r]
        assert len(hit) == 1, f"pair ({q}, {r}) is missing from the GPU hits"
        for f in fields:
            assert int(hit[0][f]) == int(getattr(want, f)), (q, r, f, int(hit[0][f]), int(getattr(want, f)))
        assert abs(float(hit[0]["ani"]) - want.ani) < 1e-6 and abs(float(hit[0]["af_query"]) - want.af_query) < 1e-6
        checked.append({"query": q, "ref": r, "ani": float(hit[0]["ani"]), "n_anchors": int(hit[0]["n_anchors"]), "n_chunks": int(hit[0]["n_chunks"])})
    t_chain /= max(1, len(checked))
    secs = n_genomes * t_sketch + n_hits * t_chain      # screens are negligible at this scale
    check = {"pairs": checked, "fields": list(fields) + ["ani (1e-6)", "af_query (1e-6)"], "result": "bit-exact integers"}
    cpu = {"value": float(n_genomes) * n_genomes / secs, "unit": "genome-pairs/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
           "sample": f"2 of the {n_genomes} genomes sketched ({t_sketch:.1f} s each) and {len(checked)} of the {n_hits} chained pairs ({t_chain:.1f} s each) by the oracle on one core; "
                     f"extrapolated to {n_genomes} sketches + {n_hits} chains = {secs:.0f} s",
           "note": PORT_NOTE}
    return check, cpu


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--refs", type=int, default=None, help="search: references per GPU (default 1000 = BASELINE configs[1]); allvsall: TOTAL genomes of the job (default 1000); metagenome: database size (default 5000); mammalian: genomes (default 8)")
    ap.add_argument("--workload", choices=["search", "allvsall", "metagenome", "mammalian"], default=None,
                    help="default (no flag) = the contract job: all-vs-all of 10 000 genomes (BASELINE configs[2], the job north_star's target is quoted on; a FIXED job, sharded over the ranks at N>1: strong scaling), "
                         "with the other configurations under extras.workloads at N=1; a named workload runs alone as the line (search = BASELINE configs[1]: 1 query vs 1 000 refs per GPU, weak scaling)")
    ap.add_argument("--no-workloads", action="store_true", help="the default run at N=1: skip extras.workloads (search 1 x 1 000, all-vs-all 1k variants, metagenome 100k, mammalian 8 x 3 Gb)")
    ap.add_argument("--queries", type=int, default=10000, help="metagenome: number of query contigs")
    ap.add_argument("--contig-mb", type=int, default=125, help="mammalian: contig length in Mb (24 contigs per genome; 125 = 3 Gb genomes)")
    ap.add_argument("--api-queries", type=int, default=10000, help="metagenome: contigs also sent one by one through Database.query() from host bytes (0 = skip)")
    ap.add_argument("--stream", action="store_true", help="mammalian: build and sketch one genome at a time through a reused device buffer (BASELINE configs[4] at full count: --refs 50), families of 10")
    ap.add_argument("--faster-small", action="store_true", help="metagenome: Database.query(faster_small=True) (no rescue of contigs with < 20 markers)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL over xGMI; gloo only for dry runs)")
    ap.add_argument("--comm", choices=["torch", "capi"], default="torch", help="N>1: who moves the exchange steps — torch.distributed, or the library's own RCCL communicator (psk_comm_*)")
    ap.add_argument("--exchange-batch", type=int, default=1024, help="allvsall N>1: genomes per rank and round of the sketch all-gather (1 024: a round of 5 Mb genomes is then large enough - 2^31 (pair, seed) items - for two batches in flight; the single-GPU emulation of a rank: 7.35 x at N = 8 against 6.79 x with 256)")
    ap.add_argument("--share-gpu", action="store_true", help="dry-run aid: every rank uses device 0 (needs --backend gloo); never for reported numbers")
    ap.add_argument("--cpu-sample", type=int, default=1000, help="CPU-baseline sample size (0 = skip): search: references (1000 = the whole workload, ~10-20 s); allvsall: queries (capped at 128); metagenome: contigs (capped at 512)")
    ap.add_argument("--variant", choices=["plain", "contigs", "sv"], default="plain", help="allvsall at N=1: the generator variant of SURVEY.md 8(d) - genomes cut into 1-80 contigs / 20 block rearrangements per genome")
    ap.add_argument("--no-api", action="store_true", help="skip the host-memory API extras (N=1 search only)")
    ap.add_argument("--no-host-leg", action="store_true", help="the default run at N=1: skip the job from ASCII in HOST memory (host_to_host; needs ~55 GB of host memory)")
    ap.add_argument("--emulate-rank-of", type=int, nargs="*", default=None, help="N=1 all-vs-all: after the timed steps, time rank 0's share of the N-way job on this one GPU for every N given (default run: 2 4 8) -> extras.scaling_model; an emulation, not a measurement")
    args = ap.parse_args()

    # --gpus N without a launcher: this process starts the N ranks itself, BEFORE torch is imported or any HIP call is made
    if spawn_decision(args.gpus, os.environ):
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

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
    job = Job(torch, device, local_rank, rank, world, dist, coll_device, args)
    job.copy_bw = device_copy_bandwidth(torch, device)
    cpu_n = args.cpu_sample if world == 1 else 0

    def as_line(entry, workload, extra_cfg=None):
        """a non-headline workload run alone: its entry re-shaped into the contract's line"""
        pairs_value = entry.get("pairs_per_s", entry["value"]) if entry["unit"] != "genome-pairs/s" else entry["value"]
        line = {"metric": "genome-pairs/sec (sketch+ANI)", "value": pairs_value, "unit": "genome-pairs/s", "n_gpus": world, "steps": entry["steps"], "warmup": entry["warmup"],
                "ms_per_step": entry["ms_per_step"], "higher_is_better": True, "scaling": entry.get("scaling", "weak"), "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": dict({"workload": entry["workload"], "hits": entry["hits"]}, **(extra_cfg or {})),
                "roofline": entry["roofline"], "clock": entry["clock"], "kernel_ms_per_step": {k: v["ms_per_step"] for k, v in entry["kernel_roofline"].items()},
                "extras": {k: v for k, v in entry.items() if k not in ("roofline", "clock", "cpu_baseline", "workload", "ms_per_step", "steps", "warmup", "value", "unit", "hits")}}
        if "cpu_baseline" in entry:
            line["cpu_baseline"] = entry["cpu_baseline"]
        return line

    line = None
    contract = args.workload is None      # no --workload: the contract job, 10 000 x 10 000 (with --refs: the same job at that size)
    if contract:
        args.workload = "allvsall"
    if args.workload == "search":
        line = run_search(job, args.steps, args.warmup, args.refs or N_REFS, cpu_n, with_api=(world == 1 and not args.no_api))
    elif args.workload == "allvsall":
        n_total = args.refs or (10000 if contract else 1000)
        full_job = contract and args.variant == "plain"
        model_ranks = args.emulate_rank_of if args.emulate_rank_of is not None else ([2, 4, 8] if (full_job and not args.no_workloads) else [])
        e = run_allvsall(job, args.steps, args.warmup, n_total, min(cpu_n, 128), variant=args.variant, verify_hits=8 if (cpu_n > 0 and world == 1) else 0,
                         host_leg=(full_job and world == 1 and not args.no_host_leg), model_ranks=model_ranks if world == 1 else ())
        if e:
            line = as_line(e, "allvsall", {"genomes": n_total, "parallelism": f"references sharded over {world} GPU(s)" + (f"; the shards' sketches all-gathered as the query side, hit records all-gathered once ({args.comm})" if world > 1 else "")})
            line["scaling"] = "strong"
            if e.get("host_to_host"):
                line["host_to_host"] = e["host_to_host"]
            if e.get("exchange"):
                line["extras"]["exchange"] = e["exchange"]
        if full_job and world == 1 and not args.no_workloads:
            t0 = time.perf_counter()
            wl = {}
            sl = run_search(job, 10, 2, N_REFS, cpu_n, with_api=not args.no_api)      # BASELINE configs[1]: the former headline
            sl.update(workload=sl["config"]["workload"], hits=sl["config"]["hits"], kernel_roofline={})
            wl["search_1k"] = sl
            # SURVEY.md §8d's harder shapes (every pair of the contract job is the easiest chaining case: one contig, substitutions only)
            wl["allvsall_1k_contigs"] = run_allvsall(job, 2, 1, 1000, 0, variant="contigs", verify_hits=8 if cpu_n > 0 else 0)
            wl["allvsall_1k_sv"] = run_allvsall(job, 2, 1, 1000, 0, variant="sv", verify_hits=8 if cpu_n > 0 else 0)
            meta = run_metagenome(job, 2, 1, 5000, 100000, (False, True), min(cpu_n, 512), 10000)
            wl["metagenome_100k"], wl["metagenome_100k_faster_small"], wl["metagenome_api"] = meta["rescue"], meta["faster_small"], meta.get("api")
            wl["mammalian_8x3Gb"] = run_mammalian(job, 2, 1, 8, 125, 2 if cpu_n > 0 else 0)
            line["extras"]["workloads"] = wl
            line["extras"]["workloads_wall_s"] = time.perf_counter() - t0
    elif args.workload == "metagenome":
        m = run_metagenome(job, args.steps, args.warmup, args.refs or 5000, args.queries, (args.faster_small,), min(cpu_n, 512), args.api_queries)
        if rank == 0:
            e = m["faster_small" if args.faster_small else "rescue"]
            line = as_line(e, "metagenome")
            line["extras"]["queries_per_s"] = e["value"]
            if m.get("api"):
                line["extras"]["api"] = m["api"]
    else:
        e = (run_mammalian_stream(job, args.refs or 50, args.contig_mb, 10, 2 if cpu_n > 0 else 0) if args.stream
             else run_mammalian(job, args.steps, args.warmup, args.refs or 8, args.contig_mb, 2 if cpu_n > 0 else 0))
        line = as_line(e, "mammalian") if rank == 0 else None
    if rank == 0 and line is not None:
        line["n_gpus"] = world
        line["copy_bw"] = job.copy_bw
        for r in [line.get("roofline")] + [w.get("roofline") for w in line.get("extras", {}).get("workloads", {}).values() if w]:
            if r and r.get("achieved") and job.copy_bw["GBps"] > 0:
                r["frac_of_copy_bw"] = r["achieved"] / job.copy_bw["GBps"]
        # every workload's full entry is its own stdout line, the complete object goes to disk, the LAST line is the compact contract line
        for k, w in line.get("extras", {}).get("workloads", {}).items():
            if w:
                print(json.dumps({"workload_entry": k, **w}), flush=True)
        path = write_full(line, args.workload + (f"_n{world}" if world > 1 else ""))
        print(json.dumps(compact_line(line, path)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
