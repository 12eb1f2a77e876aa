// Internal types shared by the translation units of libpyskani_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sys/mman.h>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <condition_variable>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/pyskani_amd.h"

// ---- algorithm constants (normative definition: oracle/skani_oracle.c) ----
constexpr int K_MARKER = 21;
constexpr uint32_t MIN_LENGTH_CONTIG = 500;  // lib.rs:156
constexpr uint32_t FRAGMENT_LENGTH = 20000;
constexpr int MAX_GAP_LENGTH = 300;
constexpr uint32_t MIN_ANCHORS = 3;
constexpr int BP_CHAIN_BAND = 2500;
constexpr int MAX_CHAIN_BAND = 100;      // look-back in anchors = clamp(BP_CHAIN_BAND / c, 1, MAX_CHAIN_BAND)
// chaining scores are kept doubled so that the gap cost |dq - dr| / 2 stays integral
constexpr int ANCHOR_SCORE2 = 40;        // anchor score 20
constexpr int MIN_SCORE2 = 90;           // 0.75 * 3 * 20 = 45
constexpr uint32_t SMALL_MARKER_COUNT = 20;

// ---- sketch tiling ----
constexpr int TILE_BASES = 16384;               // bases per workgroup
constexpr int TILE_THREADS = 256;
constexpr int TILE_WORDS = TILE_BASES / 16;     // 2-bit packed u32 words per tile
constexpr int TILE_MASKS = TILE_BASES / 64;     // u64 seed-mask words per tile (one per thread)

void psk_set_error(const char* fmt, ...);

#define PSK_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            psk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                          __LINE__);                                                          \
            return _e == hipErrorOutOfMemory ? PSK_ENOMEM : PSK_EHIP;                         \
        }                                                                                     \
    } while (0)

#define PSK_TRY(expr)                  \
    do {                               \
        psk_status _s = (expr);        \
        if (_s != PSK_OK) return _s;   \
    } while (0)

// grow-only device scratch buffer
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    psk_status reserve(size_t bytes) {
        if (bytes <= cap) return PSK_OK;
        static const bool trace = getenv("PSK_TRACE_ALLOC") != nullptr;      // diagnostics: every growth of a scratch buffer with its cost
        struct timespec t0{}, t1{};
        if (trace) clock_gettime(CLOCK_MONOTONIC, &t0);
        const size_t old = cap;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        PSK_HIP(hipMalloc(&p, want));
        cap = want;
        if (trace) { clock_gettime(CLOCK_MONOTONIC, &t1); fprintf(stderr, "[psk alloc] scratch %.1f MB -> %.1f MB: %.1f ms\n", old / 1e6, want / 1e6, (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6); }
        return PSK_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// kernels whose launches can be bracketed by HIP events on the ctx stream (psk_ctx_timing)
enum KernelId { K_SKETCH_SCAN = 0, K_SKETCH_EMIT, K_SKETCH_SORT, K_SCREEN, K_ANCHOR, K_CHAIN_CHUNK, K_SELECT, K_PAIR_REDUCE, K_ANCHOR_EMIT, K_COUNT };
static const char* const KERNEL_NAMES[K_COUNT] = {"sketch_scan", "sketch_emit", "sketch_sort", "screen", "anchor", "chain_chunk", "select", "pair_reduce", "anchor_emit"};
struct TimerRec { int id; hipEvent_t a, b; };

// One GPU (the public psk_ctx handle): the HBM block pool and the execution lanes. A LANE is what a call runs on: its own HIP
// stream, scratch buffers and pinned staging, so that calls from different host threads overlap on the device (the reference's
// `query` takes &self and releases the GIL, lib.rs:551,569: concurrent queries are its only route to parallelism).
struct Lane;
struct psk_ctx {
    int device = 0;
    bool timing = false;
    std::mutex stat_mu;
    double acc_ms[K_COUNT] = {0};
    uint64_t acc_n[K_COUNT] = {0};
    // work done by the chain stage since the last reset (psk_ctx_work): what the algorithmic bytes of its kernels are counted from
    std::atomic<uint64_t> w_pairs{0}, w_items{0}, w_anchors{0};
    // ... of the joins through the database-wide seed index: query seeds looked up (per walk) and index entries in the runs they found; and, while the timers
    // are on, candidate chains the selection read and chunk-table rows the reduce read (psk_ctx_join_work)
    std::atomic<uint64_t> w_lookups{0}, w_visited{0}, w_cands{0}, w_rows{0};
    // psk_query_host calls that ran as one launch sequence / were rerun on the general path because a capacity was exceeded / never qualified (psk_ctx_small_query_stats)
    std::atomic<uint64_t> sq_taken{0}, sq_rerun{0}, sq_general{0};
    // lanes: created on demand, at most max_lanes; a call takes a free one (LaneGuard) and gives it back
    std::mutex lanes_mu;
    std::condition_variable lanes_cv;
    std::vector<Lane*> lanes;
    std::vector<char> busy;
    int max_lanes = 8;             // per-contig queries from host threads: 2.9 k/s from one thread, 6.5 k from four, 8.6 k from eight, no more from sixteen (profiles/r3/r3q_query_threads.txt)
    std::mutex index_mu;           // k-mer index builds mutate sketches: one at a time
    uint64_t index_visit = 0;      // (under index_mu)
    std::mutex huge_mu;            // select_huge_kernel's workgroups spin at barriers and must all be resident: one such launch in flight per device
    // device block pool: sketch stores are recycled instead of hipMalloc/hipFree'd per batch
    struct PoolBlock { void* p; size_t bytes; };
    std::vector<PoolBlock> pool;
    std::mutex pool_mu;
    psk_status pool_alloc(size_t bytes, void** out, size_t* got) {
        {
            std::lock_guard<std::mutex> lk(pool_mu);
            int best = -1;
            for (int i = 0; i < (int)pool.size(); i++)
                if (pool[i].bytes >= bytes && pool[i].bytes <= bytes + bytes / 4 + (1 << 20) && (best < 0 || pool[i].bytes < pool[best].bytes)) best = i;
            if (best >= 0) { *out = pool[best].p; *got = pool[best].bytes; pool.erase(pool.begin() + best); return PSK_OK; }
        }
        size_t want = bytes + bytes / 16 + 256;
        hipError_t e = hipMalloc(out, want);
        if (e == hipErrorOutOfMemory) {      // blocks the pool keeps for reuse are the first thing to give back
            (void)hipGetLastError();
            pool_drain();
            e = hipMalloc(out, want);
        }
        if (e != hipSuccess) { psk_set_error("hipMalloc of %zu bytes failed: %s", want, hipGetErrorString(e)); return e == hipErrorOutOfMemory ? PSK_ENOMEM : PSK_EHIP; }
        *got = want;
        return PSK_OK;
    }
    void pool_release(void* p, size_t bytes) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(pool_mu);
        size_t held = 0;
        for (auto& b : pool) held += b.bytes;
        if (held + bytes > (size_t)64 << 30 || pool.size() >= 16) { (void)hipFree(p); return; }   // cap what the pool keeps
        pool.push_back({p, bytes});
    }
    void pool_drain() { std::lock_guard<std::mutex> lk(pool_mu); for (auto& b : pool) (void)hipFree(b.p); pool.clear(); }
};

struct Lane {
    psk_ctx* dev = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;      // hits of a large batch cross to the host here while the next batch computes (created on first use)
    Scratch q_sel;                          // two halves of selected hits: the one being copied is not the one the next batch writes
    psk_status copy_lane(hipStream_t* out) {
        if (!copy_stream) PSK_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        *out = copy_stream;
        return PSK_OK;
    }
    hipStream_t side_stream = nullptr; hipEvent_t side_fork = nullptr, side_join = nullptr;      // a few latency-bound waves beside a kernel that fills the chip (chain_run: the chunk walk of Gb-scale pairs beside the emit)
    psk_status side_lane(hipStream_t* out) {
        if (!side_stream) {
            PSK_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
            PSK_HIP(hipEventCreateWithFlags(&side_fork, hipEventDisableTiming));
            PSK_HIP(hipEventCreateWithFlags(&side_join, hipEventDisableTiming));
        }
        *out = side_stream;
        return PSK_OK;
    }
    bool holds_huge = false;       // between the launch of select_huge_kernel and the synchronisation that follows it
    void huge_acquire() { if (!holds_huge) { dev->huge_mu.lock(); holds_huge = true; } }
    void huge_release() { if (holds_huge) { (void)hipStreamSynchronize(stream); dev->huge_mu.unlock(); holds_huge = false; } }
    std::vector<TimerRec> pending;
    void t_begin(int id, hipStream_t st = nullptr) {
        if (!dev->timing) return;
        TimerRec r{id, nullptr, nullptr};
        (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, st ? st : stream);
        pending.push_back(r);
    }
    void t_end(hipStream_t st = nullptr) { if (dev->timing && !pending.empty()) (void)hipEventRecord(pending.back().b, st ? st : stream); }
    psk_status pool_alloc(size_t bytes, void** out, size_t* got) { return dev->pool_alloc(bytes, out, got); }
    // resources of one in-flight sketch sub-batch: own stream + scratch, so that sketch_emit / sorts of
    // sub-batch j overlap sketch_scan of sub-batch j+1
    struct JobRes {
        hipStream_t stream = nullptr; bool own_stream = false;
        hipEvent_t scan_done = nullptr;
        Scratch s_desc, s_packed, s_mask, s_counts, s_offs, s_tmp, s_mark, s_slices;
        void* pinned = nullptr; size_t pinned_cap = 0;
        psk_status pin(size_t bytes, void** out) {
            if (bytes > pinned_cap) {
                if (pinned) (void)hipHostFree(pinned);
                pinned = nullptr; pinned_cap = 0;
                size_t want = bytes * 2 + 4096;
                PSK_HIP(hipHostMalloc(&pinned, want, hipHostMallocDefault));
                pinned_cap = want;
            }
            *out = pinned;
            return PSK_OK;
        }
    };
    std::vector<JobRes*> jobs;
    psk_status job(size_t j, JobRes** out) {
        while (jobs.size() <= j) {
            JobRes* r = new JobRes();
            if (jobs.empty()) r->stream = stream;
            else { PSK_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)); r->own_stream = true; }
            PSK_HIP(hipEventCreateWithFlags(&r->scan_done, hipEventDisableTiming));
            jobs.push_back(r);
        }
        *out = jobs[j];
        return PSK_OK;
    }
    void jobs_release() {
        for (JobRes* r : jobs) {
            Scratch* all[] = {&r->s_desc, &r->s_packed, &r->s_mask, &r->s_counts, &r->s_offs, &r->s_tmp, &r->s_mark, &r->s_slices};
            for (Scratch* s : all) s->release();
            if (r->pinned) (void)hipHostFree(r->pinned);
            if (r->scan_done) (void)hipEventDestroy(r->scan_done);
            if (r->own_stream) (void)hipStreamDestroy(r->stream);
            delete r;
        }
        jobs.clear();
    }
    Scratch s_desc, s_packed, s_mask, s_counts, s_offs, s_tmp, s_mark, s_flags, s_misc;  // sketch
    Scratch q_a, q_b, q_c, q_d, q_e, q_f, q_g, q_h, q_i;                                  // query
    Scratch q_j;                                                                           // slice join (slice_join.hip): wave table, per-(pair, slice) records and bitmaps of a batch
    Scratch q_k;                                                                           // seed-index joins: per batch entry, the index blocks that hold one of its pairs' references (gsl_blocks_kernel)
    Scratch q_small;                                                                       // the one-launch-sequence query's workspace (small_query.hip)
    uint32_t sq_last_short = 192;                                                          // ... and the shortlist length of its last call on this lane: sizes the next chain launch
    void* h_pinned = nullptr;      // pinned host staging for small D2H/H2D
    size_t h_pinned_cap = 0;
    psk_status pinned(size_t bytes, void** out) {
        if (bytes > h_pinned_cap) {
            if (h_pinned) (void)hipHostFree(h_pinned);
            h_pinned = nullptr; h_pinned_cap = 0;
            size_t want = bytes * 2 + 4096;
            PSK_HIP(hipHostMalloc(&h_pinned, want, hipHostMallocDefault));
            h_pinned_cap = want;
        }
        *out = h_pinned;
        return PSK_OK;
    }
    void release_all() {
        Scratch* all[] = {&s_desc, &s_packed, &s_mask, &s_counts, &s_offs, &s_tmp, &s_mark, &s_flags, &s_misc, &q_a, &q_b, &q_c, &q_d, &q_e, &q_f, &q_g, &q_h, &q_i, &q_sel, &q_small, &q_j, &q_k};
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
        if (side_stream) { (void)hipStreamSynchronize(side_stream); (void)hipStreamDestroy(side_stream); side_stream = nullptr; if (side_fork) (void)hipEventDestroy(side_fork); if (side_join) (void)hipEventDestroy(side_join); side_fork = side_join = nullptr; }
        for (Scratch* s : all) s->release();
        jobs_release();
        if (h_pinned) (void)hipHostFree(h_pinned);
        h_pinned = nullptr; h_pinned_cap = 0;
        for (TimerRec& r : pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        pending.clear();
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
    }
};

// takes a free lane of the context for the duration of one call (creates one if all are busy and fewer than max_lanes exist)
struct LaneGuard {
    psk_ctx* c; Lane* lane = nullptr; int idx = -1;
    // nowait: no lane (lane == nullptr) rather than a wait when all are busy - a call that would like a SECOND lane must not wait for one (eight such callers would wait for each other)
    explicit LaneGuard(psk_ctx* ctx, bool nowait = false) : c(ctx) {
        std::unique_lock<std::mutex> lk(c->lanes_mu);
        for (;;) {
            for (size_t i = 0; i < c->lanes.size(); i++) if (!c->busy[i]) { idx = (int)i; break; }
            if (idx >= 0) break;
            if ((int)c->lanes.size() < c->max_lanes) {
                (void)hipSetDevice(c->device);
                Lane* L = new Lane();
                L->dev = c; L->device = c->device;
                if (hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking) != hipSuccess) { delete L; if (c->lanes.empty()) return; }
                else { c->lanes.push_back(L); c->busy.push_back(0); idx = (int)c->lanes.size() - 1; break; }
            }
            if (nowait) return;
            c->lanes_cv.wait(lk);
        }
        c->busy[idx] = 1; lane = c->lanes[idx];
        lk.unlock();
        (void)hipSetDevice(c->device);
    }
    ~LaneGuard() { if (idx >= 0) { lane->huge_release(); if (lane->side_stream) (void)hipStreamSynchronize(lane->side_stream);      /* (an error exit between the side stream's fork and join leaves it reading this lane's scratch) */
                   { std::lock_guard<std::mutex> lk(c->lanes_mu); c->busy[idx] = 0; } c->lanes_cv.notify_one(); } }
    LaneGuard(const LaneGuard&) = delete;
    LaneGuard& operator=(const LaneGuard&) = delete;
};
// The chain-stage scratch of the context's IDLE lanes (tens of GB each after an all-vs-all) goes back to the driver: a call that is about to size its batches by the free
// memory (Gb-scale pairs) would otherwise plan around buffers nobody is using. The lanes stay; their next call allocates again.
inline void psk_trim_idle_lanes(psk_ctx* c, Lane* self) {
    std::vector<Lane*> idle;
    {
        std::lock_guard<std::mutex> lk(c->lanes_mu);
        for (size_t i = 0; i < c->lanes.size(); i++) if (!c->busy[i] && c->lanes[i] != self) { c->busy[i] = 1; idle.push_back(c->lanes[i]); }
    }
    for (Lane* L : idle) {
        (void)hipStreamSynchronize(L->stream);
        Scratch* big[] = {&L->q_b, &L->q_c, &L->q_d, &L->q_e, &L->q_g, &L->q_j, &L->q_k, &L->q_sel};
        for (Scratch* s : big) s->release();
    }
    if (!idle.empty()) {
        { std::lock_guard<std::mutex> lk(c->lanes_mu); for (Lane* L : idle) for (size_t i = 0; i < c->lanes.size(); i++) if (c->lanes[i] == L) c->busy[i] = 0; }
        c->lanes_cv.notify_all();
    }
}
#define PSK_LANE(guard, ctx) LaneGuard guard(ctx); if (!guard.lane) { psk_set_error("no execution lane (hipStreamCreate failed)"); return PSK_EHIP; }

// grow-only device buffer taken from (and returned to) the context's block pool: for the per-database tables, which a
// benchmark step or a short-lived Database would otherwise hipMalloc / hipFree every time (30-40 us each)
struct PoolScratch {
    psk_ctx* dev = nullptr;
    void* p = nullptr;
    size_t cap = 0;
    PoolScratch() = default;
    PoolScratch(const PoolScratch&) = delete;
    PoolScratch& operator=(const PoolScratch&) = delete;
    PoolScratch(PoolScratch&& o) noexcept : dev(o.dev), p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    PoolScratch& operator=(PoolScratch&& o) noexcept { if (this != &o) { release(); dev = o.dev; p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    ~PoolScratch() { release(); }      // a local (the prefilter's scratch of one round) goes back to the pool on every exit path
    psk_status reserve(psk_ctx* d, size_t bytes) {
        if (bytes <= cap) return PSK_OK;
        release();
        dev = d;
        PSK_TRY(dev->pool_alloc(bytes + bytes / 4 + 256, &p, &cap));
        return PSK_OK;
    }
    void release() { if (p && dev) dev->pool_release(p, cap); p = nullptr; cap = 0; }
};

// k-mer-sorted reference index of a group of sketches, built on first chaining use (ensure_index)
struct IndexStore {
    psk_ctx* ctx = nullptr;
    void* base = nullptr;
    size_t bytes = 0;
    uint64_t* key = nullptr;   // slot<<32 | kmer, ascending: a sketch's slice is sorted by k-mer, stable in (contig,pos)
    uint32_t* perm = nullptr;  // index of the seed in the sketch's (contig,pos)-ordered arrays
    uint64_t* pms = nullptr;   // pos<<32 | meta of the seed at each index position (saves the perm -> seed_pm hop)
    uint32_t* km32 = nullptr;  // the sorted k-mers alone (low word of key): what the join streams, half the bytes
    uint32_t* bucket = nullptr; // per sketch nb+1 offsets: bucket b = entries whose k-mer >> bshift == b (a lookup is one
                               // table read plus a scan of ~4 keys instead of a 15-level binary search)
    // an index that was launched WITHOUT waiting for it (ensure_index(..., lazy): the one query sketch of a psk_query call): the stream it
    // was built on and the event behind the build; a lane on another stream waits for the event before it reads the index
    hipStream_t built_on = nullptr; hipEvent_t ready = nullptr;
    ~IndexStore();
};

inline IndexStore::~IndexStore() {
    if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); }      // (the block goes back to a pool other lanes allocate from)
    if (ctx) ctx->pool_release(base, bytes); else if (base) (void)hipFree(base);
}

// Probe tables of a group of sketches (ensure_probe): the reference side of the join of batches of MANY SMALL pairs (metagenome
// contigs: a few hundred query seeds against a reference of 10^5; anchor_join_probe_kernel). One 64-byte line holds up to
// PROBE_SLOTS distinct k-mers and, for each, exactly what the packed join record holds (x: the reference position of the match, or
// where the k-mer's run starts in the reference index when it matches more than once; y: ref contig << 1 | strand, count << 24): a
// lookup is ONE line read where the k-mer-sorted index costs a bucket-table read, a four-step search, two key reads and a position
// read, each waiting for the one before - and a contig's seeds have nothing to merge with: they fall ~500 index entries apart. line(km) = mulhi(km left-aligned, lines) is
// a multiplicative hash of the k-mer (~2.5 k-mers per line); a k-mer whose line is full sits in the next line with room.
constexpr uint32_t PROBE_SLOTS = 5;
constexpr uint32_t PROBE_EMPTY = 0xFFFFFFFFu;      // no canonical k-mer of k <= 16 (its reverse complement, 0, is smaller)
struct ProbeLine { uint32_t k[PROBE_SLOTS]; uint32_t pad; uint2 v[PROBE_SLOTS]; };
static_assert(sizeof(ProbeLine) == 64, "one probe line = one 64-byte cache line");
struct ProbeStore {
    psk_ctx* ctx = nullptr;
    void* base = nullptr;
    size_t bytes = 0;
    ~ProbeStore() { if (ctx) ctx->pool_release(base, bytes); else if (base) (void)hipFree(base); }
};

// Storage shared by the sketches of one batch: one device allocation, sliced.
struct SketchStore {
    psk_ctx* ctx = nullptr;          // blocks go back to ctx's pool (the ctx must outlive its sketches)
    void* base = nullptr;
    size_t bytes = 0, mbytes = 0;
    // slices (device pointers into base)
    uint32_t* seed_kmer = nullptr;   // (contig,pos) order
    uint32_t* seed_pos = nullptr;
    uint32_t* seed_meta = nullptr;   // contig<<1 | canon
    uint64_t* seed_pm = nullptr;     // pos<<32 | meta, (contig,pos) order: the value array of the lazy index sort
    uint64_t* markers = nullptr;     // per genome sorted unique
    uint32_t* contig_seed_start = nullptr;  // per kept contig (+1 sentinel per batch), global seed offsets
    void* mbase = nullptr;           // second allocation: the marker sets
    ~SketchStore() {
        if (ctx) { ctx->pool_release(base, bytes); ctx->pool_release(mbase, mbytes); }
        else { if (base) (void)hipFree(base); if (mbase) (void)hipFree(mbase); }
    }
};

// device-side description of one sketch: what either side of a (query, reference) pair contributes to the chain kernels,
// plus the per-sketch inputs of the learned-ANI regression. Pairs are assembled from two of these ON THE DEVICE.
struct SketchDesc {
    const uint32_t* key; const uint64_t* pms; const uint32_t* perm; const uint32_t* bucket;   // k-mer index (null until built)
    const uint32_t* pos; const uint32_t* meta; const uint32_t* seed_pos_base; const uint32_t* contig_start;
    const uint32_t* kmer;   // seed k-mers in (contig,pos) order
    uint64_t total_len;
    uint32_t n;             // seeds, once the k-mer index exists (0 before)
    uint32_t bshift, rows;  // bucket shift; rows of the chunk table a pair with this sketch as the query needs
    uint32_t n_contigs;
    float lenq[3];          // contig-length quantiles {q90, q50, q10}
    uint32_t tab_lines;     // probe table (null / 0 until built: ensure_probe)
    const ProbeLine* tab;
};

// The per-sketch contig tables: a contig of a metagenome is a sketch of ONE contig, and 100 000 of them are made and dropped per step - two heap blocks each
// (std::vector) were a third of the host time between the last kernel of a step and the first of the next. Up to four entries live in the object itself.
struct SmallVecU32 {
    uint32_t inl[4]; uint32_t* p = inl; uint32_t n = 0, cap = 4;
    SmallVecU32() = default;
    SmallVecU32(const SmallVecU32&) = delete; SmallVecU32& operator=(const SmallVecU32&) = delete;
    ~SmallVecU32() { if (p != inl) delete[] p; }
    void grow(size_t c) {
        if (c <= cap) return;
        const size_t nc = c > 2 * (size_t)cap ? c : 2 * (size_t)cap;
        uint32_t* q = new uint32_t[nc];
        memcpy(q, p, 4 * (size_t)n);
        if (p != inl) delete[] p;
        p = q; cap = (uint32_t)nc;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    uint32_t& operator[](size_t i) { return p[i]; }
    const uint32_t& operator[](size_t i) const { return p[i]; }
    const uint32_t* begin() const { return p; }
    const uint32_t* end() const { return p + n; }
    void push_back(uint32_t v) { grow((size_t)n + 1); p[n++] = v; }
    void resize(size_t c) { grow(c); for (size_t i = n; i < c; i++) p[i] = 0; n = (uint32_t)c; }
    void assign(size_t c, uint32_t v) { grow(c); for (size_t i = 0; i < c; i++) p[i] = v; n = (uint32_t)c; }
    template <class It> void assign(It a, It b) { const size_t c = (size_t)(b - a); grow(c); for (size_t i = 0; i < c; i++) p[i] = (uint32_t)a[i]; n = (uint32_t)c; }
    SmallVecU32& operator=(const std::vector<uint32_t>& v) { assign(v.begin(), v.end()); return *this; }
};

struct psk_sketch {
    psk_ctx* ctx = nullptr;
    psk_params params{};
    std::shared_ptr<SketchStore> store;
    uint64_t seed_off = 0, n_seeds = 0;      // slice of store->seed_* / idx_*
    uint64_t marker_off = 0, n_markers = 0;  // slice of store->markers
    uint64_t contig_off = 0;                 // slice of store->contig_seed_start
    SmallVecU32 contig_len;                  // kept contigs
    SmallVecU32 contig_seed_start;           // host copy, LOCAL offsets, n_contigs+1
    uint64_t total_len = 0;
    bool has_seeds = true;
    // contig-length quantiles {q90, q50, q10} (features of the learned-ANI regression): sorted lengths at n*9/10, n/2, n/10
    void len_quantiles(float out[3]) const {
        out[0] = out[1] = out[2] = 0.f;
        if (contig_len.empty()) return;
        if (contig_len.size() == 1) { out[0] = out[1] = out[2] = (float)contig_len[0]; return; }
        std::vector<uint32_t> v(contig_len.begin(), contig_len.end());
        std::sort(v.begin(), v.end());
        const size_t n = v.size();
        out[0] = (float)v[std::min(n - 1, n * 9 / 10)]; out[1] = (float)v[std::min(n - 1, n / 2)]; out[2] = (float)v[std::min(n - 1, n / 10)];
    }
    mutable uint64_t visit = 0;               // ensure_index / ensure_probe: the call that last listed this sketch (under index_mu; replaces a hash set of 10^5 pointers per round)
    mutable std::shared_ptr<IndexStore> idx;  // built on first chaining use
    mutable uint64_t idx_off = 0;
    mutable uint64_t idx_boff = 0;      // first entry of this sketch's bucket table in idx->bucket
    mutable uint32_t idx_bshift = 0;
    mutable std::shared_ptr<ProbeStore> ptab;  // built on first use by a batch of many small pairs (ensure_probe)
    mutable uint64_t ptab_off = 0;      // first line of this sketch in ptab
    mutable uint32_t ptab_lines = 0;
};

struct psk_db {
    psk_ctx* ctx = nullptr;
    // queries hold it shared while they compute; adding references and (re)building the shared device tables
    // (marker table, inverted index, k-mer indexes of references, descriptor table) take it exclusively
    mutable std::shared_mutex rw;
    psk_params params{};
    std::vector<psk_sketch*> refs;
    std::deque<std::string> names;       // deque: psk_db_name() pointers stay valid while sketches are added
    // lib.rs:51-55 + 616-637: the sketch store is a map keyed by name (a later sketch of a name replaces the earlier
    // one) while the marker list keeps both entries; a query shortlists NAMES, so every passing entry of a name
    // stands for the name's LAST sketch and yields one hit. canon[i] = last index holding names[i].
    std::vector<uint32_t> name_slot;     // open-addressed table over names: slot = (last index holding the name) + 1, 0 = empty
    std::vector<uint32_t> canon;
    bool has_dups = false, canon_dirty = false;
    static uint32_t name_hash(const std::string& s) { uint32_t h = 2166136261u; for (unsigned char c : s) h = (h ^ c) * 16777619u; return h; }
    void note_added(uint32_t i) {
        canon_dirty = true;
        if (name_slot.size() < 2 * (size_t)(i + 1) + 2) {      // keep the load below 1/2
            size_t cap = 64; while (cap < 4 * (size_t)(i + 1)) cap <<= 1;
            std::vector<uint32_t> grown(cap, 0);
            for (uint32_t v : name_slot) if (v) { size_t k = name_hash(names[v - 1]) & (cap - 1); while (grown[k]) k = (k + 1) & (cap - 1); grown[k] = v; }
            name_slot.swap(grown);
        }
        const size_t mask = name_slot.size() - 1;
        size_t k = name_hash(names[i]) & mask;
        while (name_slot[k] && names[name_slot[k] - 1] != names[i]) k = (k + 1) & mask;
        if (name_slot[k]) { has_dups = true; const uint32_t prev = name_slot[k] - 1; for (uint32_t j = 0; j < i; j++) if (canon[j] == prev) canon[j] = i; }
        name_slot[k] = i + 1;
        canon.push_back(i);
    }
    // device tables for the screen kernel, rebuilt lazily
    bool tables_dirty = true;
    PoolScratch d_marker_ptr, d_marker_n;
    // inverted marker index for many-query screens: every (marker, ref) of the db sorted by marker
    bool inv_dirty = true;
    PoolScratch inv_key, inv_ref, inv_tmp, inv_bucket;      // inv_bucket: first entry of every bucket of the marker's top inv_bits bits (2^inv_bits + 1 offsets)
    uint64_t inv_n = 0; int inv_bits = 0;
    // device table of SketchDesc, one per reference (refreshed when references are added or indexed)
    bool desc_dirty = true;
    uint64_t desc_indexed = 0; uint32_t desc_n = 0;
    PoolScratch d_refdesc, d_canon;
    std::vector<SketchDesc> h_refdesc;
    // database-wide seed index (seed_index.hip build_gsi): EVERY reference's seeds sorted by k-mer (stable: within a k-mer by reference, contig, position).
    // One lookup per query seed finds its matches in all references at once: the seed prefilter of a rescued contig in the one-launch-sequence
    // query, the join of batches of many small pairs (metagenome). gsi_val = ref << 48 | contig << 33 | pos << 1 | (fwd < rc); built once per
    // database state for databases of <= 65 536 references with <= 32 768 contigs each and < 2^31 seeds in all, dropped when references are added.
    int gsi_state = 0;      // 0 = not built, 1 = built, 2 = this database cannot have one (limits, memory)
    PoolScratch gsi_key, gsi_val, gsi_bucket;
    uint64_t gsi_n = 0; int gsi_shift = 0;
    // the same index in BLOCKS of 2^BSI_BLOG consecutive references, each block sorted by k-mer with a bucket table of its own (seed_index.hip build_bsi): what the seed-index
    // join of mid-sized pairs walks (slice_join.hip). A random genome of L bases holds a given 15-mer with probability 2 L / 4^15 ~ 1 %, and a k-mer that is a seed in one
    // genome is a seed in every genome that holds it (seeds are chosen by content): one run of the database-wide index holds ~0.01 N chance entries beside the query's
    // relatives - 93 of 147 entries per lookup at 10 000 genomes. A query's passing references are few and usually neighbours in insertion order: walking only the
    // blocks that hold one of them leaves 256 x 0.01 = 2.4 chance entries per lookup whatever the database size.
    int bsi_state = 0;      // 0 = not built, 1 = built, 2 = this database cannot have one
    // Entries carry BLOCK-LOCAL reference ids (bsi_val = local ref << 48 | contig << 33 | pos << 1 | (fwd < rc), local ref = reference & 255) and the bucket tables offsets
    // WITHIN their block (block b's entries start at bsi_base[b], a 64-bit offset): the number of references and the seeds of the database are bounded by memory only
    // (12 bytes per seed); a block of 256 references must stay below 2^31 seeds (one radix sort), a reference below 32 768 contigs.
    PoolScratch bsi_key, bsi_val, bsi_bucket, bsi_base;
    uint64_t bsi_n = 0; int bsi_shift = 0; uint32_t bsi_nb1 = 0, bsi_blocks = 0;      // nb1 = bucket-table entries per block (2^bits + 1)
    // the one-launch-sequence query (small_query.hip): 1 = every device table it reads is up to date, 2 = this database cannot take it; reset when references are added
    std::atomic<int> small_state{0};
};

// learned-ANI regression model: flattened trees in HBM
struct ModelNode { int32_t feature; float threshold; int32_t left, right; float value; int32_t missing, is_leaf, menu; };  // menu = psk_feature id of `feature`
struct ModelDev { const ModelNode* nodes; const uint32_t* first; uint32_t n_trees; uint32_t n_features; float bias, shrinkage; };
struct psk_model {
    psk_ctx* ctx = nullptr;
    void* base = nullptr;
    ModelDev dev{};
    uint64_t n_nodes = 0;
    std::vector<int32_t> features;   // psk_feature id of every position of the model's feature vector
    ~psk_model() { if (base) (void)hipFree(base); }
};
// hits[p].ani <- model(features of pair p) / 100 for every valid hit; launched on st after pair_reduce.
// pair_qr[p] = (index into qd, index into rd) of pair p.
void learned_apply_launch(const psk_model* m, psk_hit* d_hits, const uint2* pair_qr, const SketchDesc* qd, const SketchDesc* rd,
                          uint32_t n_pairs, hipStream_t st);

// ---- sketch kernel descriptors (sketch.hip; the host side of small_query.hip fills them too) ----
struct ContigDesc {
    uint64_t byte_off;     // offset of the contig in the ASCII buffer (16-byte aligned)
    uint32_t len;
    uint32_t first_tile;   // tiles of one contig are consecutive
    uint32_t genome;       // genome index inside the batch
    uint32_t contig_index; // index among the genome's kept contigs
    uint32_t pad0, pad1;
};

struct SketchConsts {
    uint64_t thr, thr_marker;
    uint32_t kmask;     // (1 << 2k) - 1
    int k, d, rshift;   // d = distance from window end to the seed's last base; rshift = 2k-2
    int delta;          // 16 - (21-k)/2: delay (bases) that puts every seed's FIRST base at a fixed index in sketch_scan
};


// ---- the one-launch-sequence query of a small genome from host bytes (psk_query_host; small_query.hip) -------------------------
// Fixed capacities: a call whose genome does not fit them takes the general path (psk_sketch_host + psk_query).
constexpr int BSI_BLOG = 8;                 // references per block of the blocked seed index: 2^8
constexpr uint32_t SQ_MAX_TILES = 64, SQ_MAX_DESC = 64;     // tiles of 16 384 bases / kept contigs of the query
constexpr uint32_t SQ_SEEDS = 3072;        // query seeds, and anchors of one (query, reference) pair, the fused chain kernel holds in LDS
constexpr uint32_t SQ_MARKERS = 2048;      // raw query markers the screen workgroup sorts in LDS
constexpr uint32_t SQ_ROWS = 64;           // chunk-table rows of a pair
constexpr uint32_t SQ_CANDS = 256;         // candidate chains of a pair
constexpr uint32_t SQ_HITS_FIRST = 256;    // hit records that cross with the status words in the one download
constexpr uint32_t SQ_F_SEEDS = 1u, SQ_F_MARKERS = 2u, SQ_F_PAIR = 4u;      // flags: a capacity was exceeded -> the general path reruns the call
struct SmallQHead {
    uint32_t n_seeds, n_markers_raw, n_markers, n_short;      // seeds of the query, raw / distinct markers, shortlisted references
    uint32_t flags, n_hits;                                    // n_hits: records with ani > 0.1 (lib.rs:654), appended by the chain workgroups (the host orders them by reference)
    unsigned long long n_anchors;                              // anchors over all pairs (psk_ctx_work)
    uint32_t coff[SQ_MAX_DESC + 1];                            // first seed of every kept contig
    uint32_t done;                                             // chain workgroups that have finished: the last one hands the status block and the hits to the host
    uint32_t pad1[128 - 9 - (SQ_MAX_DESC + 1)];
};
static_assert(sizeof(SmallQHead) == 512, "the status block is 512 bytes, the hits follow it");
struct SmallQSketch {      // device arrays of the query's sketch (lane-owned, reused from call to call)
    const uint8_t* d_bases; const ContigDesc* d_desc; const uint32_t* d_tci; const uint4* d_tinfo; const uint32_t* d_cft;
    uint32_t n_desc, n_tiles;
    uint32_t *d_packed; uint64_t* d_mask; uint32_t *d_cnt, *d_toff, *d_tmc;
    uint32_t *seed_kmer, *seed_pos, *seed_meta; uint64_t* seed_pm; uint64_t* mstage;
    SmallQHead* head;
};
// sketch_scan + sketch_emit (tile offsets computed by the emit waves themselves) on `st`; no synchronisation (sketch.hip)
psk_status small_query_sketch_enqueue(Lane* ctx, const psk_params* p, const SmallQSketch& A, hipStream_t st);

// ---- shared device helpers ----
__device__ __forceinline__ uint64_t mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}

// mm_hash64 of a key below 2^32 (k <= 16), arranged for gfx950's VALU: there 32-bit add/sub/and/or/xor/not/mov and
// right shifts issue at twice the rate of every other integer instruction (profiles/micro/valu_rates.hip), so
// the cost is the number of multiplies, funnel and 64-bit shifts. P = key * (2^21+1) < 2^53, hence
// ~P ^ (~P >> 24) = P ^ (P >> 24) ^ 0xFFFFFF00'00000000 with (P >> 56) = 0: the NOT costs one xor of the high word.
__device__ __forceinline__ uint64_t mm_hash64_u32(uint32_t key) {
    const uint64_t P = (uint64_t)key * 0x200001u;
    uint32_t lo = (uint32_t)P, hi = (uint32_t)(P >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24);
    hi ^= 0xFFFFFF00u;
    uint64_t r = (uint64_t)lo * 265u;
    hi = hi * 265u + (uint32_t)(r >> 32);
    uint64_t x = ((uint64_t)hi << 32) | (uint32_t)r;
    x ^= x >> 14;
    lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
    r = (uint64_t)lo * 21u;
    hi = hi * 21u + (uint32_t)(r >> 32);
    x = ((uint64_t)hi << 32) | (uint32_t)r;
    x ^= x >> 28;
    lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
    r = (uint64_t)lo * 0x80000001u;
    hi = (hi << 31) + (uint32_t)(r >> 32) + hi;
    return ((uint64_t)hi << 32) | (uint32_t)r;
}

// host-side entry points implemented across the translation units
psk_status sketch_batch_impl(Lane* ctx, const psk_params* p, const uint8_t* d_bases,
                             const uint64_t* contig_off, const uint64_t* contig_len,
                             const uint32_t* genome_first_contig, uint32_t n_genomes,
                             int want_seeds, psk_sketch** out, const uint32_t* d_packed_in = nullptr);
// d_packed_in: the bases 2-bit packed (pack_host.cpp), TILE_WORDS words per tile, tiles numbered over the kept contigs (length >= 500) in
// the order given, every contig starting a tile, 8 words of slack behind the last; d_bases and contig_off are then not read
extern "C" void psk_pack2bit_host(const uint8_t* src, uint64_t n, uint32_t* dst, int mode);
// sorts the seeds of every not-yet-indexed sketch by k-mer (stable) into its idx_* slice
psk_status ensure_index(Lane* ctx, const psk_sketch* const* refs, uint32_t n, bool lazy = false);
// builds the probe table of every indexed sketch of the list that lacks one (sketches of 256 .. 2^20 seeds)
psk_status ensure_probe(Lane* ctx, const psk_sketch* const* refs, uint32_t n);
// line of a k-mer: multiplicative hash, then scaled to the table. (NOT the k-mer's own top bits: canonical k-mers are the smaller of a
// k-mer and its reverse complement, so their density falls linearly from 2 at zero - twice the average load per line at the low end
// and probe chains hundreds of lines long; measured: 60 x slower.)
__device__ __forceinline__ uint32_t probe_line(uint32_t km, uint32_t lines) { return __umulhi(km * 2654435761u, lines); }
psk_status screen_impl(Lane* ctx, psk_db* db, const psk_sketch* query, double screen_val, int rescue_small,
                       uint8_t* pass, uint32_t* shared);
psk_status chain_pairs_impl(Lane* ctx, const psk_sketch* const* refs, const psk_sketch* const* queries, uint32_t n,
                            const psk_query_opts* o, psk_hit* out);
// growing array of hits in malloc'd memory: what the query entry points hand to the caller (psk_free) without another copy
// Large hit arrays (>= 8 MB) come from hit_block_alloc and go back through hit_block_free (psk_free routes them there): 2 MB-aligned,
// advised as huge pages, and the last one released is kept for the next call - a 600 MB result is 150 000 first-touch page faults and
// a 26 ms munmap otherwise, all of it with the GPU idle (profiles/r3/r3q_meta_tail.txt). PSK_HIT_CACHE=0: no block is kept.
void* hit_block_alloc(size_t bytes);      // nullptr: out of host memory
bool hit_block_free(void* p);             // false: not one of these blocks (the caller frees it)
void hit_block_trim();                    // drop the kept block (psk_ctx_destroy)
template <class H>
struct HitListT {
    H* p = nullptr; size_t n = 0, cap = 0;
    HitListT() = default;
    HitListT(const HitListT&) = delete; HitListT& operator=(const HitListT&) = delete;
    ~HitListT() { drop(p); }
    static void drop(void* q) { if (q && !hit_block_free(q)) free(q); }
    // room for `want` hits in all; pages that are never touched cost nothing
    bool reserve(size_t want) {
        if (want <= cap) return true;
        if (n && want < 2 * cap) want = 2 * cap;      // a list that already holds hits moves them when it grows: at least double, so a long result is copied O(1) times
        const size_t bytes = sizeof(H) * want;
        H* q = (H*)(bytes >= ((size_t)8 << 20) ? hit_block_alloc(bytes) : malloc(bytes));
        if (!q) return false;
        if (n) memcpy(q, p, sizeof(H) * n);
        drop(p);
        p = q; cap = want;
        return true;
    }
    bool reserve_for(size_t k) { return n + k <= cap || reserve(std::max<size_t>(n + k, cap + cap / 2 + 64)); }      // room for k more hits (the caller writes them, then adds k to n)
    bool append(const H* src, size_t k) {
        if (n + k > cap && !reserve(std::max<size_t>(n + k, cap + cap / 2 + 64))) return false;
        if (k) memcpy(p + n, src, sizeof(H) * k);
        n += k;
        return true;
    }
    H* release() { H* q = p; p = nullptr; n = cap = 0; return q; }
};
using HitList = HitListT<psk_hit>;
using HitListMin = HitListT<psk_hit_min>;      // the 20-byte records of psk_query_many_min: what the reference's Hit holds (hit.rs:77-104)
// Database.query for n_queries sketches (lib.rs:569-659): hits of query i are all[offsets[i] .. offsets[i+1]), ref insertion order
psk_status query_many_impl(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o,
                           HitList& all, uint64_t* offsets);
// ... the same with psk_hit_min records (query = index of the query within the call | learned << 31)
psk_status query_many_min_impl(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o,
                               HitListMin& all, uint64_t* offsets);
psk_status chain_impl(Lane* ctx, const psk_sketch* const* refs, uint32_t n_refs,
                      const psk_sketch* query, const psk_query_opts* o, psk_hit* out);
// Database.query from host bytes (lib.rs:549-660 with the _sketch call inside it): the one-launch-sequence path for a small genome
// (small_query.hip; *done = false: not eligible or a capacity was exceeded - the caller takes psk_sketch_host + query_many_impl)
constexpr uint32_t SQ_MAX_REFS = 24 * 1024;      // the screen workgroup keeps one shared-marker counter per reference in LDS
psk_status query_host_small(Lane* ctx, psk_db* db, const uint8_t* const* contigs, const uint64_t* lens, uint32_t n_contigs, const psk_query_opts* o,
                            HitList& all, bool* done);
// what that path reads on the device (seed_index.hip): marker table, inverted marker index, canon table, every reference indexed and described.
// Called with the database locked SHARED through `sh`; *ok = false: this database cannot take the path (a reference without seeds, other parameters)
psk_status small_query_prepare(Lane* ctx, psk_db* db, std::shared_lock<std::shared_mutex>& sh, bool* ok);
