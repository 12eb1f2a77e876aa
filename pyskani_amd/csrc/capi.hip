// C-ABI surface of libpyskani_amd.so: context, sketch handles, database, query driver.
// See include/pyskani_amd.h for the reference call sites each entry point replaces.
#include "common.h"
#include <cstdarg>
#include <algorithm>

static thread_local char g_err[512] = "";

void psk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- large hit arrays: a registry of the blocks handed out (address -> bytes) and ONE kept block (see HitList in common.h)
#include <sys/mman.h>
#include <unordered_map>
namespace {
struct HitBlocks {
    std::mutex mu;
    std::unordered_map<void*, size_t> live;
    void* kept = nullptr; size_t kept_bytes = 0;
    const bool keep = !(getenv("PSK_HIT_CACHE") && getenv("PSK_HIT_CACHE")[0] == '0');
};
HitBlocks& hit_blocks() { static HitBlocks* hb = new HitBlocks(); return *hb; }      // never destroyed: psk_free may run during process exit
}
void* hit_block_alloc(size_t bytes) {
    const size_t al = (size_t)2 << 20, rounded = (bytes + al - 1) / al * al;
    HitBlocks& hb = hit_blocks();
    void* q = nullptr; size_t have = 0;
    {
        std::lock_guard<std::mutex> lk(hb.mu);
        if (hb.kept && hb.kept_bytes >= rounded) { q = hb.kept; have = hb.kept_bytes; hb.kept = nullptr; hb.kept_bytes = 0; }
    }
    if (!q) {
        q = aligned_alloc(al, rounded);
        if (!q) return nullptr;
        (void)madvise(q, rounded, MADV_HUGEPAGE);
        have = rounded;
    }
    std::lock_guard<std::mutex> lk(hb.mu);
    hb.live[q] = have;
    return q;
}
bool hit_block_free(void* p) {
    HitBlocks& hb = hit_blocks();
    void* drop = nullptr;
    {
        std::lock_guard<std::mutex> lk(hb.mu);
        auto it = hb.live.find(p);
        if (it == hb.live.end()) return false;
        const size_t bytes = it->second;
        hb.live.erase(it);
        if (hb.keep && bytes > hb.kept_bytes) { drop = hb.kept; hb.kept = p; hb.kept_bytes = bytes; }      // the larger of the two stays
        else drop = p;
    }
    free(drop);
    return true;
}
void hit_block_trim() {
    HitBlocks& hb = hit_blocks();
    void* drop;
    { std::lock_guard<std::mutex> lk(hb.mu); drop = hb.kept; hb.kept = nullptr; hb.kept_bytes = 0; }
    free(drop);
}

static void ingest_release(psk_ctx* c);   // host-ingest pipeline resources (defined with psk_sketch_many_host)
static psk_status ingest_impl(psk_ctx* ctx, Lane* lane, const psk_params* p, const uint8_t* const* contigs, const uint64_t* lens,
                              const uint32_t* genome_first_contig, uint32_t n_genomes, int want_seeds, psk_sketch** out);

extern "C" {

const char* psk_last_error(void) { return g_err; }
const char* psk_version(void) { return "pyskani_amd 0.5.0 (gfx950; algorithm: skani 0.3.0 restatement)"; }
int psk_abi_version(void) { return PSK_ABI_VERSION; }
void psk_free(void* p) { if (p && !hit_block_free(p)) free(p); }

psk_status psk_ctx_create(int device, psk_ctx** out) {
    if (!out) { psk_set_error("ctx_create: NULL out"); return PSK_EINVAL; }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        psk_set_error("no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return PSK_EHIP;
    }
    if (device < 0 || device >= n) { psk_set_error("device %d out of range (0..%d)", device, n - 1); return PSK_EINVAL; }
    PSK_HIP(hipSetDevice(device));
    psk_ctx* c = new psk_ctx();
    c->device = device;
    if (const char* ml = getenv("PSK_LANES")) c->max_lanes = std::max(1, std::min(16, atoi(ml)));
    { LaneGuard first(c); if (!first.lane) { delete c; psk_set_error("hipStreamCreate failed"); return PSK_EHIP; } }   // the first lane exists from the start
    *out = c;
    return PSK_OK;
}

// every lane idle and held: for operations on the whole context
namespace {
struct AllLanes {
    psk_ctx* c; size_t n = 0;
    explicit AllLanes(psk_ctx* ctx) : c(ctx) {
        std::unique_lock<std::mutex> lk(c->lanes_mu);
        c->lanes_cv.wait(lk, [&] { for (char b : c->busy) if (b) return false; return true; });
        n = c->lanes.size();
        for (size_t i = 0; i < n; i++) c->busy[i] = 1;
    }
    ~AllLanes() { { std::lock_guard<std::mutex> lk(c->lanes_mu); for (size_t i = 0; i < n; i++) c->busy[i] = 0; } c->lanes_cv.notify_all(); }
};
}

void psk_ctx_destroy(psk_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    {
        AllLanes all(c);
        for (Lane* L : c->lanes) { if (L->stream) (void)hipStreamSynchronize(L->stream); }
        ingest_release(c);
        for (Lane* L : c->lanes) { L->release_all(); delete L; }
        c->lanes.clear(); c->busy.clear(); all.n = 0;
    }
    c->pool_drain();
    delete c;
    hit_block_trim();
}

psk_status psk_ctx_synchronize(psk_ctx* c) {
    if (!c) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    PSK_HIP(hipSetDevice(c->device));
    AllLanes all(c);
    for (Lane* L : c->lanes) PSK_HIP(hipStreamSynchronize(L->stream));
    return PSK_OK;
}

psk_status psk_ctx_set_timing(psk_ctx* c, int on) {
    if (!c) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    AllLanes all(c);
    c->timing = on != 0;
    return PSK_OK;
}

psk_status psk_ctx_timing(psk_ctx* c, const char* kernel, double* total_ms, uint64_t* launches) {
    if (!c || !kernel) { psk_set_error("ctx_timing: NULL argument"); return PSK_EINVAL; }
    PSK_HIP(hipSetDevice(c->device));
    AllLanes all(c);
    for (Lane* L : c->lanes) {
        PSK_HIP(hipStreamSynchronize(L->stream));
        for (TimerRec& r : L->pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->acc_ms[r.id] += ms; c->acc_n[r.id]++; }
            (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        }
        L->pending.clear();
    }
    for (int i = 0; i < K_COUNT; i++) if (!strcmp(kernel, KERNEL_NAMES[i])) {
        if (total_ms) *total_ms = c->acc_ms[i];
        if (launches) *launches = c->acc_n[i];
        return PSK_OK;
    }
    if (!strcmp(kernel, "reset")) { for (int i = 0; i < K_COUNT; i++) { c->acc_ms[i] = 0; c->acc_n[i] = 0; } return PSK_OK; }
    psk_set_error("unknown kernel name '%s'", kernel);
    return PSK_EINVAL;
}

// ---- shader-clock probe: a fixed integer-VALU load (8 independent chains of v_alignbit_b32 per lane, 8 waves per SIMD), ~1 ms.
// v_alignbit_b32 issues at 16 lanes per SIMD and cycle on gfx950 (profiles/micro/valu_rates.hip: 4 cycles per wave64
// instruction), so clock = wave-instructions per SIMD x 4 / duration. bench.py runs it before a timed loop: sketch_scan is bound by
// VALU issue, so its time follows the clock the box holds, and the record lets box-to-box spread be told from a regression.
}  // extern "C"
namespace {
constexpr int CLK_CHAINS = 8, CLK_UNROLL = 32, CLK_ITERS = 256, CLK_BLOCKS = 2048;
__global__ __launch_bounds__(256) void clock_probe_kernel(uint32_t* __restrict__ out, int iters, uint32_t c) {
    uint32_t a[CLK_CHAINS];
#pragma unroll
    for (int j = 0; j < CLK_CHAINS; j++) a[j] = threadIdx.x * 7u + j + blockIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < CLK_UNROLL; u++) {
#pragma unroll
            for (int j = 0; j < CLK_CHAINS; j++) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[j]) : "v"(c));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < CLK_CHAINS; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}
extern "C" {
psk_status psk_ctx_clock_probe(psk_ctx* c, double* mhz, double* ms_out) {
    if (!c || !mhz) { psk_set_error("clock_probe: NULL argument"); return PSK_EINVAL; }
    PSK_LANE(lg, c);
    Lane* lane = lg.lane;
    PSK_TRY(lane->s_misc.reserve(4 * (size_t)CLK_BLOCKS * 256));
    int cus = 0;
    PSK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    hipEvent_t a, b;
    PSK_HIP(hipEventCreate(&a)); PSK_HIP(hipEventCreate(&b));
    float best = 0;
    for (int rep = 0; rep < 4; rep++) {      // the first launch is a warm-up (clock ramp); the shortest of the rest counts
        PSK_HIP(hipEventRecord(a, lane->stream));
        hipLaunchKernelGGL(clock_probe_kernel, dim3(CLK_BLOCKS), dim3(256), 0, lane->stream, (uint32_t*)lane->s_misc.p, CLK_ITERS, 0x9E3779B9u);
        PSK_HIP(hipEventRecord(b, lane->stream));
        PSK_HIP(hipEventSynchronize(b));
        float ms = 0;
        PSK_HIP(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && (best == 0 || ms < best)) best = ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    const double wave_inst = (double)CLK_BLOCKS * 4 * CLK_CHAINS * CLK_UNROLL * CLK_ITERS;
    *mhz = best > 0 ? wave_inst * 4.0 / ((double)cus * 4.0) / (best * 1e-3) / 1e6 : 0.0;
    if (ms_out) *ms_out = best;
    return PSK_OK;
}

psk_status psk_ctx_work(psk_ctx* c, uint64_t* pairs, uint64_t* items, uint64_t* anchors, int reset) {
    if (!c) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    if (pairs) *pairs = c->w_pairs.load();
    if (items) *items = c->w_items.load();
    if (anchors) *anchors = c->w_anchors.load();
    if (reset) { c->w_pairs = 0; c->w_items = 0; c->w_anchors = 0; }
    return PSK_OK;
}

psk_status psk_ctx_join_work(psk_ctx* c, uint64_t* lookups, uint64_t* visited, uint64_t* candidates, uint64_t* rows, int reset) {
    if (!c) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    if (lookups) *lookups = c->w_lookups.load();
    if (visited) *visited = c->w_visited.load();
    if (candidates) *candidates = c->w_cands.load();
    if (rows) *rows = c->w_rows.load();
    if (reset) { c->w_lookups = 0; c->w_visited = 0; c->w_cands = 0; c->w_rows = 0; }
    return PSK_OK;
}

psk_status psk_ctx_small_query_stats(psk_ctx* c, uint64_t* taken, uint64_t* rerun, uint64_t* general) {
    if (!c) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    if (taken) *taken = c->sq_taken.load();
    if (rerun) *rerun = c->sq_rerun.load();
    if (general) *general = c->sq_general.load();
    return PSK_OK;
}
psk_status psk_device_alloc(psk_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) { psk_set_error("device_alloc: NULL argument"); return PSK_EINVAL; }
    PSK_HIP(hipSetDevice(c->device));
    PSK_HIP(hipMalloc(dptr, bytes ? bytes : 16));
    return PSK_OK;
}
psk_status psk_device_free(psk_ctx* c, void* dptr) {
    if (!c) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    PSK_HIP(hipSetDevice(c->device));
    PSK_HIP(hipFree(dptr));
    return PSK_OK;
}
psk_status psk_memcpy_h2d(psk_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || (!dst && bytes) || (!src && bytes)) { psk_set_error("memcpy_h2d: NULL argument"); return PSK_EINVAL; }
    PSK_LANE(lg, c);
    PSK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, lg.lane->stream));
    PSK_HIP(hipStreamSynchronize(lg.lane->stream));
    return PSK_OK;
}

psk_status psk_sketch_batch_device(psk_ctx* ctx, const psk_params* p, const uint8_t* d_bases, const uint64_t* contig_off,
                                   const uint64_t* contig_len, const uint32_t* genome_first_contig, uint32_t n_genomes,
                                   int want_seeds, psk_sketch** out) {
    if (!ctx || !out || (n_genomes && (!genome_first_contig || !contig_off || !contig_len))) { psk_set_error("sketch_batch: NULL argument"); return PSK_EINVAL; }
    PSK_LANE(lg, ctx);
    Lane* lane = lg.lane;
    // seed offsets are 32-bit per launch: size sub-batches so that the expected seed count
    // (bases / c) stays far below 2^31; if a launch still overflows (PSK_ELIMIT) halve and retry
    const uint64_t c_eff = p && p->c > 0 ? (uint64_t)p->c : 1;
    uint64_t limit = std::min<uint64_t>(1ull << 36, (1ull << 30) * c_eff);
    uint32_t g0 = 0;
    while (g0 < n_genomes) {
        uint32_t g1 = g0;
        uint64_t bases = 0;
        while (g1 < n_genomes) {
            uint64_t gb = 0;
            for (uint32_t c = genome_first_contig[g1]; c < genome_first_contig[g1 + 1]; c++) gb += contig_len[c] + TILE_BASES;
            if (g1 > g0 && bases + gb > limit) break;
            bases += gb; g1++;
        }
        psk_status rc = sketch_batch_impl(lane, p, d_bases, contig_off, contig_len, genome_first_contig + g0, g1 - g0, want_seeds, out + g0);
        if (rc == PSK_ELIMIT && g1 - g0 > 1) { limit = bases / 2; continue; }
        if (rc != PSK_OK) {
            for (uint32_t g = 0; g < g0; g++) { delete out[g]; out[g] = nullptr; }
            return rc;
        }
        g0 = g1;
    }
    return PSK_OK;
}

psk_status psk_sketch_host(psk_ctx* ctx, const psk_params* p, const uint8_t* const* contigs, const uint64_t* lens,
                           uint32_t n_contigs, int want_seeds, psk_sketch** out) {
    if (!ctx || !p || !out || (n_contigs && (!contigs || !lens))) { psk_set_error("sketch_host: NULL argument"); return PSK_EINVAL; }
    PSK_LANE(lg, ctx);
    Lane* lane = lg.lane;
    // stage the contigs in HBM at 16-byte aligned offsets (short contigs are dropped later, lib.rs:156)
    std::vector<uint64_t> off(n_contigs), len(n_contigs);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_contigs; i++) {
        off[i] = total; len[i] = lens[i];
        if (lens[i] >= MIN_LENGTH_CONTIG) total += (lens[i] + 15) & ~15ull;
    }
    if (total >= (64u << 20)) {   // a large genome: through the pinned staging pipeline (a plain copy from pageable memory moves 12-20 GB/s, the pipeline ~50).
                                  // Measured for 5 MB genomes: the pipeline's start-up (producer thread, staging hand-offs) costs more than it saves (0.43 vs 0.26 ms)
        const uint32_t gfc[2] = {0, n_contigs};
        return ingest_impl(ctx, lane, p, contigs, lens, gfc, 1, want_seeds, out);
    }
    PSK_TRY(lane->s_misc.reserve(total + 64));
    uint8_t* d = (uint8_t*)lane->s_misc.p;
    for (uint32_t i = 0; i < n_contigs; i++)
        if (lens[i] >= MIN_LENGTH_CONTIG) PSK_HIP(hipMemcpyAsync(d + off[i], contigs[i], lens[i], hipMemcpyHostToDevice, lane->stream));
    uint32_t gfc[2] = {0, n_contigs};
    return sketch_batch_impl(lane, p, d, off.data(), len.data(), gfc, 1, want_seeds, out);
}

void psk_sketch_free(psk_sketch* s) { delete s; }
void psk_sketch_free_many(psk_sketch* const* sketches, uint32_t n) { if (sketches) for (uint32_t i = 0; i < n; i++) delete sketches[i]; }

psk_status psk_sketch_info(const psk_sketch* s, psk_params* p, uint64_t* n_seeds, uint64_t* n_markers, uint64_t* total_len, uint32_t* n_contigs) {
    if (!s) { psk_set_error("NULL sketch"); return PSK_EINVAL; }
    if (p) *p = s->params;
    if (n_seeds) *n_seeds = s->n_seeds;
    if (n_markers) *n_markers = s->n_markers;
    if (total_len) *total_len = s->total_len;
    if (n_contigs) *n_contigs = (uint32_t)s->contig_len.size();
    return PSK_OK;
}

psk_status psk_sketch_export(const psk_sketch* s, psk_seed* seeds, uint64_t* markers) {
    if (!s) { psk_set_error("NULL sketch"); return PSK_EINVAL; }
    PSK_LANE(lg, s->ctx);
    Lane* ctx = lg.lane;
    if (seeds && s->n_seeds) {
        size_t n = s->n_seeds;
        std::vector<uint32_t> kmer(n), pos(n), meta(n);
        PSK_HIP(hipMemcpyAsync(kmer.data(), s->store->seed_kmer + s->seed_off, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(hipMemcpyAsync(pos.data(), s->store->seed_pos + s->seed_off, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(hipMemcpyAsync(meta.data(), s->store->seed_meta + s->seed_off, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < n; i++) { seeds[i].kmer = kmer[i]; seeds[i].pos = pos[i]; seeds[i].contig = meta[i] >> 1; seeds[i].canon = meta[i] & 1; }
    }
    if (markers && s->n_markers) {
        PSK_HIP(hipMemcpyAsync(markers, s->store->markers + s->marker_off, 8 * (size_t)s->n_markers, hipMemcpyDeviceToHost, ctx->stream));
        PSK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PSK_OK;
}

psk_status psk_sketch_contig_lens(const psk_sketch* s, uint32_t* lens) {
    if (!s || (!lens && !s->contig_len.empty())) { psk_set_error("contig_lens: NULL argument"); return PSK_EINVAL; }
    for (size_t i = 0; i < s->contig_len.size(); i++) lens[i] = s->contig_len[i];
    return PSK_OK;
}

/* Rebuild a device-resident sketch from its exported form (Database.open/load). */
psk_status psk_sketch_import(psk_ctx* ctx, const psk_params* p, const uint32_t* contig_lens, uint32_t n_contigs,
                             const psk_seed* seeds, uint64_t n_seeds, const uint64_t* markers, uint64_t n_markers,
                             int has_seeds, psk_sketch** out) {
    if (!ctx || !p || !out || (n_contigs && !contig_lens) || (n_seeds && !seeds) || (n_markers && !markers)) { psk_set_error("sketch_import: NULL argument"); return PSK_EINVAL; }
    if (p->k < 1 || p->k > 16 || p->c < 1 || p->marker_c < 1) { psk_set_error("invalid sketch parameters"); return PSK_EINVAL; }
    if (n_seeds >= 0x7FFFFFF0ull || n_markers >= 0x7FFFFFF0ull) { psk_set_error("sketch too large to import"); return PSK_ELIMIT; }
    *out = nullptr;
    PSK_LANE(lg, ctx);
    Lane* lane = lg.lane;
    std::unique_ptr<psk_sketch> s(new psk_sketch());
    s->ctx = ctx; s->params = *p; s->has_seeds = has_seeds != 0;
    s->contig_len.assign(contig_lens, contig_lens + n_contigs);
    for (uint32_t i = 0; i < n_contigs; i++) s->total_len += contig_lens[i];
    s->contig_seed_start.assign(n_contigs + 1, 0);
    const size_t ns = (size_t)n_seeds;
    std::vector<uint32_t> kmer(ns), pos(ns), meta(ns), cstart(n_contigs + 1, 0);
    std::vector<uint64_t> pm(ns);
    for (size_t i = 0; i < ns; i++) {
        const psk_seed& sd = seeds[i];
        if (sd.contig >= n_contigs || sd.pos >= contig_lens[sd.contig] || sd.canon > 1 ||
            (i && (sd.contig < seeds[i - 1].contig || (sd.contig == seeds[i - 1].contig && sd.pos <= seeds[i - 1].pos)))) {
            psk_set_error("sketch_import: seed %zu is out of range or out of (contig,pos) order", i);
            return PSK_EINVAL;
        }
        kmer[i] = sd.kmer; pos[i] = sd.pos; meta[i] = (sd.contig << 1) | sd.canon; pm[i] = ((uint64_t)sd.pos << 32) | meta[i];
        cstart[sd.contig + 1]++;
    }
    for (uint32_t c = 0; c < n_contigs; c++) cstart[c + 1] += cstart[c];
    s->contig_seed_start = cstart;
    for (uint64_t i = 1; i < n_markers; i++) if (markers[i] <= markers[i - 1]) { psk_set_error("sketch_import: markers must be sorted and distinct"); return PSK_EINVAL; }
    auto store = std::make_shared<SketchStore>();
    store->ctx = ctx;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t b_kmer = 0, b_pos = al(b_kmer + 4 * ns), b_meta = al(b_pos + 4 * ns), b_pm = al(b_meta + 4 * ns), b_cs = al(b_pm + 8 * ns), b_end = al(b_cs + 4 * (size_t)(n_contigs + 1));
    PSK_TRY(ctx->pool_alloc(b_end, &store->base, &store->bytes));
    char* sb = (char*)store->base;
    store->seed_kmer = (uint32_t*)(sb + b_kmer); store->seed_pos = (uint32_t*)(sb + b_pos); store->seed_meta = (uint32_t*)(sb + b_meta);
    store->seed_pm = (uint64_t*)(sb + b_pm); store->contig_seed_start = (uint32_t*)(sb + b_cs);
    PSK_TRY(ctx->pool_alloc(8 * ((size_t)n_markers + 1), &store->mbase, &store->mbytes));
    store->markers = (uint64_t*)store->mbase;
    hipStream_t st = lane->stream;
    if (ns) {
        PSK_HIP(hipMemcpyAsync(store->seed_kmer, kmer.data(), 4 * ns, hipMemcpyHostToDevice, st));
        PSK_HIP(hipMemcpyAsync(store->seed_pos, pos.data(), 4 * ns, hipMemcpyHostToDevice, st));
        PSK_HIP(hipMemcpyAsync(store->seed_meta, meta.data(), 4 * ns, hipMemcpyHostToDevice, st));
        PSK_HIP(hipMemcpyAsync(store->seed_pm, pm.data(), 8 * ns, hipMemcpyHostToDevice, st));
    }
    PSK_HIP(hipMemcpyAsync(store->contig_seed_start, cstart.data(), 4 * (size_t)(n_contigs + 1), hipMemcpyHostToDevice, st));
    if (n_markers) PSK_HIP(hipMemcpyAsync(store->markers, markers, 8 * (size_t)n_markers, hipMemcpyHostToDevice, st));
    PSK_HIP(hipStreamSynchronize(st));
    s->store = store;
    s->seed_off = 0; s->n_seeds = n_seeds; s->marker_off = 0; s->n_markers = n_markers; s->contig_off = 0;
    *out = s.release();
    return PSK_OK;
}

psk_status psk_db_create(psk_ctx* ctx, const psk_params* p, psk_db** out) {
    if (!ctx || !p || !out) { psk_set_error("db_create: NULL argument"); return PSK_EINVAL; }
    if (p->k < 1 || p->k > 16 || p->c < 1 || p->marker_c < 1) { psk_set_error("invalid sketch parameters (c=%d marker_c=%d k=%d)", p->c, p->marker_c, p->k); return PSK_EINVAL; }
    psk_db* db = new psk_db();
    db->ctx = ctx; db->params = *p;
    *out = db;
    return PSK_OK;
}

void psk_db_destroy(psk_db* db) {
    if (!db) return;
    (void)hipSetDevice(db->ctx->device);
    for (psk_sketch* s : db->refs) delete s;
    db->d_marker_ptr.release(); db->d_marker_n.release();
    db->inv_key.release(); db->inv_ref.release(); db->inv_tmp.release(); db->inv_bucket.release();
    db->d_refdesc.release(); db->d_canon.release();
    db->gsi_key.release(); db->gsi_val.release(); db->gsi_bucket.release();
    db->bsi_key.release(); db->bsi_val.release(); db->bsi_bucket.release(); db->bsi_base.release();
    delete db;
}

psk_status psk_db_add(psk_db* db, const char* name, psk_sketch* s) {
    if (!db || !name || !s) { delete s; psk_set_error("db_add: NULL argument"); return PSK_EINVAL; }
    if (s->ctx != db->ctx) { delete s; psk_set_error("db_add: sketch belongs to another context"); return PSK_EINVAL; }
    std::unique_lock<std::shared_mutex> lk(db->rw);      // exclusive, as `&mut self` makes Database::sketch (lib.rs:479)
    db->refs.push_back(s);
    db->names.emplace_back(name);
    db->note_added((uint32_t)db->refs.size() - 1);
    db->tables_dirty = true; db->inv_dirty = true; db->desc_dirty = true; db->small_state = 0; db->gsi_key.release(); db->gsi_val.release(); db->gsi_bucket.release(); db->gsi_state = 0;
    db->bsi_key.release(); db->bsi_val.release(); db->bsi_bucket.release(); db->bsi_base.release(); db->bsi_state = 0;
    return PSK_OK;
}

psk_status psk_db_add_batch(psk_db* db, const char* const* names, psk_sketch* const* sketches, uint32_t n) {
    if (!db || (n && (!names || !sketches))) { psk_set_error("db_add_batch: NULL argument"); return PSK_EINVAL; }
    std::unique_lock<std::shared_mutex> lk(db->rw);
    for (uint32_t i = 0; i < n; i++)
        if (!sketches[i] || !names[i] || sketches[i]->ctx != db->ctx) { psk_set_error("db_add_batch: bad entry %u", i); return PSK_EINVAL; }
    for (uint32_t i = 0; i < n; i++) {
        db->refs.push_back(sketches[i]);
        db->names.emplace_back(names[i]);
        db->note_added((uint32_t)db->refs.size() - 1);
    }
    db->tables_dirty = true; db->inv_dirty = true; db->desc_dirty = true; db->small_state = 0; db->gsi_key.release(); db->gsi_val.release(); db->gsi_bucket.release(); db->gsi_state = 0;
    db->bsi_key.release(); db->bsi_val.release(); db->bsi_bucket.release(); db->bsi_base.release(); db->bsi_state = 0;
    return PSK_OK;
}

// the three readers take the database lock shared: a concurrent psk_db_add may be growing the vectors
uint32_t psk_db_size(const psk_db* db) { if (!db) return 0; std::shared_lock<std::shared_mutex> lk(db->rw); return (uint32_t)db->refs.size(); }
const char* psk_db_name(const psk_db* db, uint32_t i) { if (!db) return nullptr; std::shared_lock<std::shared_mutex> lk(db->rw); return i < db->names.size() ? db->names[i].c_str() : nullptr; }
const psk_sketch* psk_db_sketch(const psk_db* db, uint32_t i) { if (!db) return nullptr; std::shared_lock<std::shared_mutex> lk(db->rw); return i < db->refs.size() ? db->refs[i] : nullptr; }

psk_status psk_screen(psk_db* db, const psk_sketch* q, double screen_val, int rescue_small, uint8_t* pass, uint32_t* shared) {
    if (!db || !q || !pass) { psk_set_error("screen: NULL argument"); return PSK_EINVAL; }
    PSK_LANE(lg, db->ctx);
    return screen_impl(lg.lane, db, q, screen_val, rescue_small, pass, shared);
}

psk_status psk_chain(psk_ctx* ctx, const psk_sketch* const* refs, uint32_t n_refs, const psk_sketch* q, const psk_query_opts* o, psk_hit* out) {
    if (!ctx) { psk_set_error("NULL ctx"); return PSK_EINVAL; }
    PSK_LANE(lg, ctx);
    return chain_impl(lg.lane, refs, n_refs, q, o, out);
}

static psk_status hits_out(HitList& all, psk_hit** hits) {
    if (!all.p) { all.p = (psk_hit*)malloc(sizeof(psk_hit)); if (!all.p) { psk_set_error("out of host memory"); return PSK_ENOMEM; } }      // no hit: still a pointer psk_free takes
    *hits = all.release();      // the list's own malloc'd array: no copy
    return PSK_OK;
}

psk_status psk_query(psk_db* db, const psk_sketch* q, const psk_query_opts* o, psk_hit** hits, uint64_t* n_hits) {
    if (!db || !q || !o || !hits || !n_hits) { psk_set_error("query: NULL argument"); return PSK_EINVAL; }
    *hits = nullptr; *n_hits = 0;
    PSK_LANE(lg, db->ctx);
    HitList all;
    uint64_t offs[2];
    PSK_TRY(query_many_impl(lg.lane, db, &q, 1, o, all, offs));
    const uint64_t nh = all.n;
    PSK_TRY(hits_out(all, hits));
    *n_hits = nh;
    return PSK_OK;
}

/* Database.query as the reference runs it (lib.rs:549-660): the query genome arrives as host ASCII contigs, is sketched (lib.rs:571, not stored)
 * and queried. Same hits as psk_sketch_host + psk_query + psk_sketch_free. A small genome (a contig, a bin of a few hundred kb) takes the
 * one-launch-sequence path of small_query.hip - one upload, four kernels, one download, one synchronisation; anything else, or a call that
 * exceeds that path's capacities, runs the two general calls here. */
psk_status psk_query_host(psk_db* db, const uint8_t* const* contigs, const uint64_t* lens, uint32_t n_contigs, int want_seeds,
                          const psk_query_opts* o, psk_hit** hits, uint64_t* n_hits) {
    if (!db || !o || !hits || !n_hits || (n_contigs && (!contigs || !lens))) { psk_set_error("query_host: NULL argument"); return PSK_EINVAL; }
    *hits = nullptr; *n_hits = 0;
    HitList all;
    if (want_seeds) {
        PSK_LANE(lg, db->ctx);
        bool done = false;
        PSK_TRY(query_host_small(lg.lane, db, contigs, lens, n_contigs, o, all, &done));
        if (!done) db->ctx->sq_general++;
        if (done) {
            const uint64_t nh = all.n;
            PSK_TRY(hits_out(all, hits));
            *n_hits = nh;
            return PSK_OK;
        }
        all.n = 0;
    }
    psk_sketch* q = nullptr;
    PSK_TRY(psk_sketch_host(db->ctx, &db->params, contigs, lens, n_contigs, want_seeds, &q));
    const psk_status rc = psk_query(db, q, o, hits, n_hits);
    delete q;
    return rc;
}

/* Many queries against one database. Same result as n_queries x psk_query; hits of query i are
 * hits[offsets[i] .. offsets[i+1]). */
psk_status psk_query_many(psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o,
                          psk_hit** hits, uint64_t* offsets) {
    if (!db || (!queries && n_queries) || !o || !hits || !offsets) { psk_set_error("query_many: NULL argument"); return PSK_EINVAL; }
    *hits = nullptr;
    offsets[0] = 0;
    PSK_LANE(lg, db->ctx);
    HitList all;
    PSK_TRY(query_many_impl(lg.lane, db, queries, n_queries, o, all, offsets));
    return hits_out(all, hits);
}

psk_status psk_query_many_min(psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o,
                              psk_hit_min** hits, uint64_t* offsets) {
    if (!db || (!queries && n_queries) || !o || !hits || !offsets) { psk_set_error("query_many_min: NULL argument"); return PSK_EINVAL; }
    *hits = nullptr;
    offsets[0] = 0;
    PSK_LANE(lg, db->ctx);
    HitListMin all;
    PSK_TRY(query_many_min_impl(lg.lane, db, queries, n_queries, o, all, offsets));
    if (!all.p) { all.p = (psk_hit_min*)malloc(sizeof(psk_hit_min)); if (!all.p) { psk_set_error("out of host memory"); return PSK_ENOMEM; } }      // no hit: still a pointer psk_free takes
    *hits = all.release();
    return PSK_OK;
}

}  // extern "C"


// ------------------------------------------------------------------ host-ASCII ingest pipeline
// psk_sketch_many_host: many genomes whose contigs sit in ordinary (pageable) host memory. The boundary of the reference
// hands over host buffers (lib.rs:485-489), so from ASCII in host memory the path is bound by PCIe, not by the sketch
// kernels (5 MB per genome against ~4 µs of sketch_scan). Three stages overlap:
//   worker threads  pageable -> pinned staging slots   (a single memcpy thread moves ~10 GB/s, PCIe Gen5 x16 ~55 GB/s)
//   copy stream     pinned slot -> device sub-batch buffer (hipMemcpyAsync, one DMA per 32 MB slot)
//   ctx stream      sketch_batch_impl over the previous sub-batch (double-buffered device ASCII)
#include <condition_variable>
#include <functional>
#include <thread>
#include <atomic>
#include <map>
namespace {
struct ParallelFor {   // persistent workers; run(n, fn) executes fn(0..n-1) on the workers and the caller
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv, cv_done;
    std::function<void(int)> fn; int n = 0, next = 0, active = 0; uint64_t gen = 0; bool stop = false;
    explicit ParallelFor(int workers) {
        for (int i = 0; i < workers; i++) th.emplace_back([this] {
            uint64_t seen = 0;
            for (;;) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || (gen != seen && next < n); });
                if (stop) return;
                seen = gen;
                while (next < n) { int i = next++; active++; lk.unlock(); fn(i); lk.lock(); active--; }
                if (active == 0) cv_done.notify_all();
            }
        });
    }
    void run(int count, std::function<void(int)> f) {
        std::unique_lock<std::mutex> lk(mu);
        fn = std::move(f); n = count; next = 0; gen++;
        cv.notify_all();
        while (next < n) { int i = next++; active++; lk.unlock(); fn(i); lk.lock(); active--; }
        cv_done.wait(lk, [&] { return active == 0 && next >= n; });
    }
    ~ParallelFor() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); }
};
constexpr size_t INGEST_SLOT = 32u << 20;       // pinned staging slot
constexpr int INGEST_SLOTS = 4;
struct IngestRes {   // per-context resources of the pipeline, created on first use
    hipStream_t copy = nullptr;
    void* pinned[INGEST_SLOTS] = {nullptr}; hipEvent_t slot_free[INGEST_SLOTS] = {nullptr};
    Scratch dev[2]; hipEvent_t ready[2] = {nullptr, nullptr};
    std::unique_ptr<ParallelFor> pool;
    std::mutex mu;
};
std::mutex g_ingest_mu;
std::map<psk_ctx*, IngestRes*> g_ingest;
}  // namespace
static void ingest_release(psk_ctx* c) {
    std::lock_guard<std::mutex> g(g_ingest_mu);
    auto it = g_ingest.find(c);
    if (it == g_ingest.end()) return;
    IngestRes* R = it->second;
    if (R->copy) (void)hipStreamSynchronize(R->copy);
    for (int i = 0; i < INGEST_SLOTS; i++) { if (R->pinned[i]) (void)hipHostFree(R->pinned[i]); if (R->slot_free[i]) (void)hipEventDestroy(R->slot_free[i]); }
    for (int i = 0; i < 2; i++) { R->dev[i].release(); if (R->ready[i]) (void)hipEventDestroy(R->ready[i]); }
    if (R->copy) (void)hipStreamDestroy(R->copy);
    delete R;
    g_ingest.erase(it);
}

extern "C" {

psk_status psk_sketch_many_host(psk_ctx* ctx, const psk_params* p, const uint8_t* const* contigs, const uint64_t* lens,
                                const uint32_t* genome_first_contig, uint32_t n_genomes, int want_seeds, psk_sketch** out) {
    if (!ctx || !p || !out || (n_genomes && (!genome_first_contig || !lens || !contigs))) { psk_set_error("sketch_many_host: NULL argument"); return PSK_EINVAL; }
    for (uint32_t g = 0; g < n_genomes; g++) out[g] = nullptr;
    if (!n_genomes) return PSK_OK;
    PSK_LANE(lg, ctx);
    return ingest_impl(ctx, lg.lane, p, contigs, lens, genome_first_contig, n_genomes, want_seeds, out);
}

}  // extern "C"

static psk_status ingest_impl(psk_ctx* ctx, Lane* lane, const psk_params* p, const uint8_t* const* contigs, const uint64_t* lens,
                              const uint32_t* genome_first_contig, uint32_t n_genomes, int want_seeds, psk_sketch** out) {
    for (uint32_t g = 0; g < n_genomes; g++) out[g] = nullptr;
    IngestRes* R;
    {
        std::lock_guard<std::mutex> g(g_ingest_mu);
        IngestRes*& slot = g_ingest[ctx];
        if (!slot) {
            // built aside and published only once every resource exists: a failure half-way (4 x 32 MB of pinned memory, events, the
            // stream) leaves no half-initialised entry for the next call to trip over
            IngestRes* fresh = new IngestRes();
            auto init = [&]() -> psk_status {
                PSK_HIP(hipStreamCreateWithFlags(&fresh->copy, hipStreamNonBlocking));
                for (int i = 0; i < INGEST_SLOTS; i++) {
                    PSK_HIP(hipHostMalloc(&fresh->pinned[i], INGEST_SLOT, hipHostMallocDefault));
                    PSK_HIP(hipEventCreateWithFlags(&fresh->slot_free[i], hipEventDisableTiming));
                }
                for (int i = 0; i < 2; i++) PSK_HIP(hipEventCreateWithFlags(&fresh->ready[i], hipEventDisableTiming));
                return PSK_OK;
            };
            const psk_status irc = init();
            if (irc != PSK_OK) {
                for (int i = 0; i < INGEST_SLOTS; i++) { if (fresh->pinned[i]) (void)hipHostFree(fresh->pinned[i]); if (fresh->slot_free[i]) (void)hipEventDestroy(fresh->slot_free[i]); }
                for (int i = 0; i < 2; i++) if (fresh->ready[i]) (void)hipEventDestroy(fresh->ready[i]);
                if (fresh->copy) (void)hipStreamDestroy(fresh->copy);
                delete fresh;
                g_ingest.erase(ctx);
                return irc;
            }
            const char* env = getenv("PSK_INGEST_THREADS");
            // (measured on the 256-thread MI355X hosts, 1 000 x 5 Mb genomes: packed ingest 33.2 k genomes/s with 8 workers, 28.1 k with 16, 26.8 k with 32, 21.0 k with 64 -
            // more workers only contend for the memory controllers; the plain ASCII copy is PCIe-bound at 8.4-8.9 k whatever the count: profiles/r4/r4d_ingest_threads.txt)
            const unsigned hw = std::thread::hardware_concurrency();
            int nt = env ? atoi(env) : (int)std::min(8u, std::max(1u, hw / 2));
            fresh->pool.reset(new ParallelFor(std::max(0, nt - 1)));
            slot = fresh;
        }
        R = slot;
    }
    std::lock_guard<std::mutex> ingest_lock(R->mu);     // one pipelined ingest per context at a time (its staging slots are shared)
    const uint32_t n_contigs = genome_first_contig[n_genomes];
    // PACKED ingest: the worker threads turn ASCII into 2-bit words (pack_host.cpp) while they fill the pinned slots, so a genome crosses PCIe as
    // L / 4 bytes and sketch_scan reads the words (its phase 1 becomes a copy). Every kept contig starts a 4 KB tile in the packed layout, so the mode
    // is taken where that is at most half of the ASCII (genomes of long contigs: always; bins of thousands of 500-base contigs: never).
    // PSK_INGEST_PACKED=0 / 1 forbid / force it (tests, A/B).
    auto tiles_of = [&](uint32_t c) -> uint64_t { return lens[c] >= MIN_LENGTH_CONTIG ? (lens[c] + TILE_BASES - 1) / TILE_BASES : 0; };
    bool packed;
    {
        uint64_t ascii_total = 0, tiles_total = 0;
        for (uint32_t c = 0; c < n_contigs; c++) if (lens[c] >= MIN_LENGTH_CONTIG) { ascii_total += lens[c]; tiles_total += tiles_of(c); }
        const char* pe = getenv("PSK_INGEST_PACKED");
        packed = pe ? pe[0] == '1' : tiles_total * (uint64_t)(4 * TILE_WORDS) * 2 <= ascii_total;
    }
    constexpr uint64_t TILE_BYTES = 4ull * TILE_WORDS;
    // sub-batches of ~192 MB of ASCII (whole genomes; packed: ~96 MB of words = 384 MB of ASCII); device offsets of the kept contigs, 16-byte aligned
    // (packed: off[] stays zero - the tile tables of sketch_batch_impl place the contigs - and tile0[c] is the contig's first tile in its sub-batch)
    const uint64_t SUB = packed ? (96ull << 20) : (192ull << 20);
    struct Sub { uint32_t g0, g1; uint64_t bytes; };
    std::vector<Sub> subs;
    std::vector<uint64_t> off(n_contigs, 0), len64(lens, lens + n_contigs), tile0(packed ? n_contigs : 0, 0);
    for (uint32_t g = 0; g < n_genomes;) {
        Sub s{g, g, 0};
        while (s.g1 < n_genomes) {
            uint64_t gb = 0;
            for (uint32_t c = genome_first_contig[s.g1]; c < genome_first_contig[s.g1 + 1]; c++) if (lens[c] >= MIN_LENGTH_CONTIG) gb += packed ? tiles_of(c) * TILE_BYTES : ((lens[c] + 15) & ~15ull);
            if (s.g1 > s.g0 && s.bytes + gb > SUB) break;
            uint64_t o = s.bytes;
            for (uint32_t c = genome_first_contig[s.g1]; c < genome_first_contig[s.g1 + 1]; c++) {
                if (packed) { tile0[c] = o / TILE_BYTES; o += tiles_of(c) * TILE_BYTES; }
                else { off[c] = o; if (lens[c] >= MIN_LENGTH_CONTIG) o += (lens[c] + 15) & ~15ull; }
            }
            s.bytes += gb; s.g1++;
        }
        subs.push_back(s);
        g = s.g1;
    }
    // consumer state shared with the producer thread
    std::mutex m; std::condition_variable cv;
    int produced = 0, consumed = 0; psk_status prod_rc = PSK_OK; std::string prod_err;
    const int device = ctx->device;
    std::thread producer([&] {
        (void)hipSetDevice(device);
        int slot_i = 0;
        for (size_t b = 0; b < subs.size(); b++) {
            { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return consumed + 2 > (int)b; }); }     // device buffer b % 2 is free again
            const Sub& s = subs[b];
            Scratch& D = R->dev[b & 1];
            psk_status rc = D.reserve(s.bytes + 64);
            hipError_t e = hipSuccess;
            const uint32_t c0 = genome_first_contig[s.g0], c1 = genome_first_contig[s.g1];
            for (uint64_t lo = 0; lo < s.bytes && rc == PSK_OK && e == hipSuccess; lo += INGEST_SLOT) {
                const uint64_t hi = std::min<uint64_t>(s.bytes, lo + INGEST_SLOT);
                char* pin = (char*)R->pinned[slot_i];
                e = hipEventSynchronize(R->slot_free[slot_i]);      // its previous DMA has drained
                if (e != hipSuccess) break;
                const int parts = (int)std::min<uint64_t>(64, (hi - lo + (1u << 20) - 1) >> 20);
                if (packed) R->pool->run(parts, [&](int t) {      // tiles [ta, tz) of the sub-batch: 16 384 bases -> 1 024 words each, zeros behind a contig's end
                    const uint64_t ta = (lo + (hi - lo) * (uint64_t)t / parts) / TILE_BYTES, tz = t + 1 == parts ? hi / TILE_BYTES : (lo + (hi - lo) * (uint64_t)(t + 1) / parts) / TILE_BYTES;
                    uint32_t l = c0, r = c1;      // first contig whose tiles end after ta
                    while (l < r) { const uint32_t mid = (l + r) >> 1; if (tile0[mid] + tiles_of(mid) <= ta) l = mid + 1; else r = mid; }
                    uint32_t c = l;
                    for (uint64_t tl = ta; tl < tz; tl++) {
                        while (c < c1 && tile0[c] + tiles_of(c) <= tl) c++;
                        uint32_t* dst = (uint32_t*)(pin + (tl * TILE_BYTES - lo));
                        uint32_t words = 0;
                        if (c < c1) {
                            const uint64_t pos0 = (tl - tile0[c]) * TILE_BASES, nb = std::min<uint64_t>(TILE_BASES, lens[c] - pos0);
                            psk_pack2bit_host(contigs[c] + pos0, nb, dst, 0);
                            words = (uint32_t)((nb + 15) / 16);
                        }
                        if (words < (uint32_t)TILE_WORDS) memset(dst + words, 0, 4 * (size_t)(TILE_WORDS - words));
                    }
                });
                else
                R->pool->run(parts, [&](int t) {
                    const uint64_t a = lo + (hi - lo) * (uint64_t)t / parts, z = lo + (hi - lo) * (uint64_t)(t + 1) / parts;
                    // first contig whose device range ends after a
                    uint32_t l = c0, r = c1;
                    while (l < r) { uint32_t mid = (l + r) >> 1; if (off[mid] + (lens[mid] >= MIN_LENGTH_CONTIG ? lens[mid] : 0) <= a) l = mid + 1; else r = mid; }
                    for (uint32_t c = l; c < c1 && off[c] < z; c++) {
                        if (lens[c] < MIN_LENGTH_CONTIG) continue;
                        const uint64_t s0 = std::max(a, off[c]), s1 = std::min(z, off[c] + lens[c]);
                        if (s0 < s1) memcpy(pin + (s0 - lo), contigs[c] + (s0 - off[c]), s1 - s0);
                    }
                });
                e = hipMemcpyAsync((char*)D.p + lo, pin, hi - lo, hipMemcpyHostToDevice, R->copy);
                if (e == hipSuccess) e = hipEventRecord(R->slot_free[slot_i], R->copy);
                slot_i = (slot_i + 1) % INGEST_SLOTS;
            }
            if (e == hipSuccess && rc == PSK_OK) e = hipEventRecord(R->ready[b & 1], R->copy);
            std::lock_guard<std::mutex> l(m);
            if (rc != PSK_OK || e != hipSuccess) { prod_rc = rc != PSK_OK ? rc : PSK_EHIP; prod_err = e != hipSuccess ? hipGetErrorString(e) : psk_last_error(); produced = (int)subs.size(); cv.notify_all(); return; }
            produced = (int)b + 1;
            cv.notify_all();
        }
    });
    psk_status rc = PSK_OK;
    for (size_t b = 0; b < subs.size() && rc == PSK_OK; b++) {
        { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return produced > (int)b; }); if (prod_rc != PSK_OK) { rc = prod_rc; psk_set_error("ingest: %s", prod_err.c_str()); break; } }
        const Sub& s = subs[b];
        hipError_t e = hipStreamWaitEvent(lane->stream, R->ready[b & 1], 0);
        if (e != hipSuccess) { psk_set_error("hipStreamWaitEvent: %s", hipGetErrorString(e)); rc = PSK_EHIP; break; }
        rc = sketch_batch_impl(lane, p, (const uint8_t*)R->dev[b & 1].p, off.data(), len64.data(), genome_first_contig + s.g0, s.g1 - s.g0, want_seeds, out + s.g0,
                               packed ? (const uint32_t*)R->dev[b & 1].p : nullptr);
        { std::lock_guard<std::mutex> l(m); consumed = (int)b + 1; }
        cv.notify_all();
    }
    { std::lock_guard<std::mutex> l(m); consumed = (int)subs.size() + 2; }
    cv.notify_all();
    producer.join();
    (void)hipStreamSynchronize(R->copy);
    if (rc != PSK_OK) for (uint32_t g = 0; g < n_genomes; g++) { delete out[g]; out[g] = nullptr; }
    return rc;
}
