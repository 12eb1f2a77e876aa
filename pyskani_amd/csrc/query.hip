// Marker screen, seed-index lookup, chunked chaining and ANI/AF reduction on gfx950.
// Replaces, for Database::query (/root/reference/src/pyskani/_skani/lib.rs:569-659):
//   skani::screen::check_markers_quickly   (call site lib.rs:623-628)  -> screen_kernel
//   skani::chain::chain_seeds              (call site lib.rs:652-653)  -> anchor_* / chunk_* /
//                                                                        chain_chunk / pair_reduce
// Semantics are normative in oracle/skani_oracle.c (orc_screen / orc_chain).
#include "common.h"
#include "chain_dev.h"
#include "slice_join.h"
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <algorithm>
#include <deque>
#include <unordered_map>
#include <thread>

// ------------------------------------------------------------------ screen
static inline size_t al256s(size_t x) { return (x + 255) & ~(size_t)255; }


__global__ __launch_bounds__(256) void screen_kernel(const MarkerSet* __restrict__ refs, const uint64_t* __restrict__ qm,
                                                     uint32_t nq, double thresh, int rescue_small,
                                                     uint8_t* __restrict__ pass, uint32_t* __restrict__ shared_out) {
    __shared__ uint32_t s_cnt[4];
    const MarkerSet r = refs[blockIdx.x];
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < nq; i += blockDim.x) {
        uint64_t m = qm[i];
        uint32_t lo = 0, hi = r.n;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (r.p[mid] < m) lo = mid + 1; else hi = mid; }
        cnt += (lo < r.n && r.p[lo] == m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sh = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        uint32_t small = nq < r.n ? nq : r.n;
        int ok;
        if (rescue_small && small < SMALL_MARKER_COUNT) ok = 1;
        else if (small == 0) ok = 0;
        else ok = ((double)sh / (double)small) > thresh;
        pass[blockIdx.x] = (uint8_t)ok;
        shared_out[blockIdx.x] = sh;
    }
}

static psk_status upload_marker_table(Lane* ctx, psk_db* db);
psk_status screen_impl(Lane* ctx, psk_db* db, const psk_sketch* q, double screen_val, int rescue_small, uint8_t* pass, uint32_t* shared) {
    std::shared_lock<std::shared_mutex> sh(db->rw);
    if (db->tables_dirty) {      // shared device tables are (re)built under the exclusive lock: no other lane is reading them
        sh.unlock();
        { std::unique_lock<std::shared_mutex> ex(db->rw); PSK_TRY(upload_marker_table(ctx, db)); }
        sh.lock();
    }
    const uint32_t n = (uint32_t)db->refs.size();
    if (n == 0) return PSK_OK;
    hipStream_t st = ctx->stream;
    PSK_TRY(ctx->q_a.reserve((size_t)n * 8));
    uint8_t* d_pass = (uint8_t*)ctx->q_a.p;
    uint32_t* d_shared = (uint32_t*)((char*)ctx->q_a.p + (((size_t)n + 3) & ~(size_t)3));
    const uint64_t* qm = q->store ? q->store->markers + q->marker_off : nullptr;
    double thresh = pow(screen_val, (double)K_MARKER);
    ctx->t_begin(K_SCREEN);
    hipLaunchKernelGGL(screen_kernel, dim3(n), dim3(256), 0, st, (const MarkerSet*)db->d_marker_ptr.p, qm, (uint32_t)q->n_markers,
                       thresh, rescue_small, d_pass, d_shared);
    ctx->t_end();
    void* hp;
    PSK_TRY(ctx->pinned((size_t)n * 8 + 16, &hp));
    uint8_t* h_pass = (uint8_t*)hp;
    uint32_t* h_shared = (uint32_t*)((char*)hp + (((size_t)n + 3) & ~(size_t)3));
    PSK_HIP(hipMemcpyAsync(h_pass, d_pass, n, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipMemcpyAsync(h_shared, d_shared, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipStreamSynchronize(st));
    memcpy(pass, h_pass, n);
    if (shared) memcpy(shared, h_shared, sizeof(uint32_t) * n);
    return PSK_OK;
}

// many queries x all refs: one workgroup per (ref, query); pass[q * n_refs + r]
__global__ __launch_bounds__(256) void screen_many_kernel(const MarkerSet* __restrict__ refs, const MarkerSet* __restrict__ queries,
                                                          uint32_t n_refs, double thresh, int rescue_small, uint8_t* __restrict__ pass) {
    __shared__ uint32_t s_cnt[4];
    const MarkerSet r = refs[blockIdx.x];
    const MarkerSet q = queries[blockIdx.y];
    const uint32_t small = q.n < r.n ? q.n : r.n;
    uint32_t cnt = 0;
    if (!(rescue_small && small < SMALL_MARKER_COUNT) && small > 0) {
        for (uint32_t i = threadIdx.x; i < q.n; i += blockDim.x) {
            uint64_t m = q.p[i];
            uint32_t lo = 0, hi = r.n;
            while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (r.p[mid] < m) lo = mid + 1; else hi = mid; }
            cnt += (lo < r.n && r.p[lo] == m);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sh = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        int ok;
        if (rescue_small && small < SMALL_MARKER_COUNT) ok = 1;
        else if (small == 0) ok = 0;
        else ok = ((double)sh / (double)small) > thresh;
        pass[(size_t)blockIdx.y * n_refs + blockIdx.x] = (uint8_t)ok;
    }
}

// The same for marker sets too large for one workgroup per pair (a 3 Gb genome holds 3 M markers: 60 ms in one workgroup): the
// query's markers are cut into slices of SCREEN_SLICE, one workgroup per (reference, query, slice) adds its shared-marker count
// to the pair's cell, inv_decide_kernel applies the pass rule.
constexpr uint32_t SCREEN_SLICE = 16384;
__global__ __launch_bounds__(256) void screen_slice_kernel(const MarkerSet* __restrict__ refs, const MarkerSet* __restrict__ queries,
                                                           uint32_t n_refs, uint32_t* __restrict__ count) {
    __shared__ uint32_t s_cnt[4];
    const MarkerSet r = refs[blockIdx.x];
    const MarkerSet q = queries[blockIdx.y];
    const uint32_t i0 = blockIdx.z * SCREEN_SLICE, i1 = i0 + SCREEN_SLICE < q.n ? i0 + SCREEN_SLICE : q.n;
    uint32_t cnt = 0;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
        const uint64_t m = q.p[i];
        uint32_t lo = 0, hi = r.n;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (r.p[mid] < m) lo = mid + 1; else hi = mid; }
        cnt += (lo < r.n && r.p[lo] == m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]; if (t) atomicAdd(&count[(size_t)blockIdx.y * n_refs + blockIdx.x], t); }
}

static psk_status upload_marker_table(Lane* ctx, psk_db* db) {
    const uint32_t n = (uint32_t)db->refs.size();
    if (!db->tables_dirty) return PSK_OK;
    std::vector<MarkerSet> h(n);
    for (uint32_t i = 0; i < n; i++) {
        const psk_sketch* r = db->refs[i];
        h[i].p = r->store ? r->store->markers + r->marker_off : nullptr;
        h[i].n = (uint32_t)r->n_markers; h[i].pad = 0;
    }
    PSK_TRY(db->d_marker_ptr.reserve(ctx->dev, sizeof(MarkerSet) * n));
    PSK_HIP(hipMemcpyAsync(db->d_marker_ptr.p, h.data(), sizeof(MarkerSet) * n, hipMemcpyHostToDevice, ctx->stream));
    PSK_HIP(hipStreamSynchronize(ctx->stream));
    db->tables_dirty = false;
    return PSK_OK;
}

// ---- inverted marker index: for 10^6+ (query, ref) pairs the screen costs O(shared markers), not O(Q x R x M) ----
__global__ __launch_bounds__(256) void inv_gather_kernel(const MarkerSet* __restrict__ refs, const uint32_t* __restrict__ roff,
                                                         uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const MarkerSet r = refs[blockIdx.x];
    const uint32_t o = roff[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < r.n; i += blockDim.x) { key[o + i] = r.p[i]; val[o + i] = blockIdx.x; }
}
// one lane per (query, marker): every ref that holds the marker gets +1 in the query's row of the count matrix
__global__ __launch_bounds__(256) void inv_lookup_kernel(const MarkerSet* __restrict__ queries, const uint32_t* __restrict__ qoff, uint32_t nq,
                                                         uint32_t n_items, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                         uint32_t n_inv, uint32_t n_refs, uint32_t* __restrict__ count) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    uint32_t lo = 0, hi = nq - 1;
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (qoff[mid] <= i) lo = mid; else hi = mid - 1; }
    const uint32_t q = lo;
    const uint64_t m = queries[q].p[i - qoff[q]];
    uint32_t l = 0, h = n_inv;
    while (l < h) { uint32_t mid = (l + h) >> 1; if (key[mid] < m) l = mid + 1; else h = mid; }
    for (; l < n_inv && key[l] == m; l++) atomicAdd(&count[(size_t)q * n_refs + val[l]], 1u);
}
__global__ __launch_bounds__(256) void inv_decide_kernel(const MarkerSet* __restrict__ refs, const MarkerSet* __restrict__ queries,
                                                         uint32_t n_refs, uint32_t nq, const uint32_t* __restrict__ count,
                                                         double thresh, int rescue_small, uint8_t* __restrict__ pass) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n_refs * nq) return;
    const uint32_t q = (uint32_t)(t / n_refs), r = (uint32_t)(t % n_refs);
    const uint32_t a = queries[q].n, b = refs[r].n, small = a < b ? a : b;
    int ok;
    if (rescue_small && small < SMALL_MARKER_COUNT) ok = 1;
    else if (small == 0) ok = 0;
    else ok = ((double)count[t] / (double)small) > thresh;
    pass[t] = (uint8_t)ok;
}

// The same screen with the query's row of the count matrix kept in LDS: one workgroup per query, shared-marker counts per
// reference by LDS atomics, the pass rule applied in place. No count matrix in HBM, no memset, no decide pass (n_refs * 4 B of
// LDS per workgroup: databases up to INV_LDS_REFS references).
constexpr uint32_t INV_LDS_REFS = 36 * 1024;
__global__ __launch_bounds__(512) void inv_screen_lds_kernel(const MarkerSet* __restrict__ refs, const MarkerSet* __restrict__ queries,
                                                             const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint32_t n_inv,
                                                             uint32_t n_refs, double thresh, int rescue_small, uint8_t* __restrict__ pass) {
    extern __shared__ uint32_t s_count[];
    const MarkerSet q = queries[blockIdx.x];
    for (uint32_t r = threadIdx.x; r < n_refs; r += blockDim.x) s_count[r] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < q.n; i += blockDim.x) {
        const uint64_t m = q.p[i];
        uint32_t l = 0, h = n_inv;
        while (l < h) { uint32_t mid = (l + h) >> 1; if (key[mid] < m) l = mid + 1; else h = mid; }
        for (; l < n_inv && key[l] == m; l++) atomicAdd(&s_count[val[l]], 1u);
    }
    __syncthreads();
    uint8_t* row = pass + (size_t)blockIdx.x * n_refs;
    for (uint32_t r = threadIdx.x; r < n_refs; r += blockDim.x) {
        const uint32_t b = refs[r].n, small = q.n < b ? q.n : b;
        int ok;
        if (rescue_small && small < SMALL_MARKER_COUNT) ok = 1;
        else if (small == 0) ok = 0;
        else ok = ((double)s_count[r] / (double)small) > thresh;
        row[r] = (uint8_t)ok;
    }
}

// bucket[b] = first entry of the sorted inverted index whose marker >> shift is >= b (b = 0 .. nb)
__global__ __launch_bounds__(256) void inv_bucket_kernel(const uint64_t* __restrict__ key, uint32_t n, int shift, uint32_t nb, uint32_t* __restrict__ bucket) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = (uint32_t)(key[i] >> shift);
    const uint32_t from = i ? (uint32_t)(key[i - 1] >> shift) + 1u : 0u;
    for (uint32_t x = from; x <= b; x++) bucket[x] = i;
    if (i == n - 1) for (uint32_t x = b + 1; x <= nb; x++) bucket[x] = n;
}

// The inverted screen, one workgroup per query as above, with the lookup done by WAVES: a marker's bucket (its top bits index a table of first entries,
// ~16 entries per bucket) is read with one coalesced load, the lanes that hold the marker itself add to their reference's counter. Per marker that is
// three dependent round trips (table, keys, references) shared by 64 lanes, four markers in flight per wave - where one lane per marker walked a 26-step
// binary search over the whole index and then its ~100 matches one dependent load at a time (59 -> 9 ms per 10 000 x 10 000 screen).
__global__ __launch_bounds__(512) void inv_screen_wave_kernel(const MarkerSet* __restrict__ refs, const MarkerSet* __restrict__ queries,
                                                              const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, const uint32_t* __restrict__ bucket, int shift,
                                                              uint32_t n_refs, double thresh, int rescue_small, uint8_t* __restrict__ pass) {
    extern __shared__ uint32_t s_count[];
    const MarkerSet q = queries[blockIdx.x];
    for (uint32_t r = threadIdx.x; r < n_refs; r += blockDim.x) s_count[r] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    constexpr int U = 4;
    for (uint32_t i0 = wave * U; i0 < q.n; i0 += n_waves * U) {
        uint64_t m[U]; uint32_t lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; u++) { m[u] = i0 + u < q.n ? q.p[i0 + u] : ~0ull; }
#pragma unroll
        for (int u = 0; u < U; u++) { const uint32_t b = (uint32_t)(m[u] >> shift); lo[u] = 0; hi[u] = 0; if (i0 + u < q.n) { lo[u] = bucket[b]; hi[u] = bucket[b + 1]; } }
        uint64_t k[U];
#pragma unroll
        for (int u = 0; u < U; u++) k[u] = lo[u] + lane < hi[u] ? key[lo[u] + lane] : ~0ull;
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (lo[u] + lane < hi[u] && k[u] == m[u]) atomicAdd(&s_count[val[lo[u] + lane]], 1u);
            for (uint32_t x = lo[u] + 64u + lane; x < hi[u]; x += 64u) if (key[x] == m[u]) atomicAdd(&s_count[val[x]], 1u);      // a bucket of more than 64 entries (rare)
        }
    }
    __syncthreads();
    uint8_t* row = pass + (size_t)blockIdx.x * n_refs;
    for (uint32_t r = threadIdx.x; r < n_refs; r += blockDim.x) {
        const uint32_t b = refs[r].n, small = q.n < b ? q.n : b;
        int ok;
        if (rescue_small && small < SMALL_MARKER_COUNT) ok = 1;
        else if (small == 0) ok = 0;
        else ok = ((double)s_count[r] / (double)small) > thresh;
        row[r] = (uint8_t)ok;
    }
}

static psk_status build_inverted(Lane* ctx, psk_db* db) {
    if (!db->inv_dirty) return PSK_OK;
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)db->refs.size();
    std::vector<uint32_t> roff(n + 1, 0);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; i++) { roff[i] = (uint32_t)tot; tot += db->refs[i]->n_markers; }
    roff[n] = (uint32_t)tot;
    if (tot >= 0x7FFFFFF0ull) { psk_set_error("database holds too many markers for one inverted index"); return PSK_ELIMIT; }
    db->inv_n = tot;
    PSK_TRY(db->inv_key.reserve(ctx->dev, 8 * (tot + 1)));
    PSK_TRY(db->inv_ref.reserve(ctx->dev, 4 * (tot + 1)));
    PSK_TRY(db->inv_tmp.reserve(ctx->dev, 12 * (tot + 1) + 4 * (size_t)(n + 1)));
    uint64_t* k_in = (uint64_t*)db->inv_tmp.p; uint32_t* v_in = (uint32_t*)(k_in + tot + 1); uint32_t* d_roff = v_in + tot + 1;
    PSK_HIP(hipMemcpyAsync(d_roff, roff.data(), 4 * (size_t)(n + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(inv_gather_kernel, dim3(n), dim3(256), 0, st, (const MarkerSet*)db->d_marker_ptr.p, d_roff, k_in, v_in);
    if (tot) {
        size_t tmp = 0;
        PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, k_in, (uint64_t*)db->inv_key.p, v_in, (uint32_t*)db->inv_ref.p, (int)tot, 0, 2 * K_MARKER, st));
        PSK_TRY(ctx->q_c.reserve(tmp));
        PSK_HIP(hipcub::DeviceRadixSort::SortPairs(ctx->q_c.p, tmp, k_in, (uint64_t*)db->inv_key.p, v_in, (uint32_t*)db->inv_ref.p, (int)tot, 0, 2 * K_MARKER, st));
    }
    {   // bucket table on the markers' top bits: ~16 entries per bucket (markers are 2 K_MARKER-bit canonical k-mers)
        int bits = 4; while (bits < 24 && (16ull << bits) < tot) bits++;
        if (bits > 2 * K_MARKER) bits = 2 * K_MARKER;
        db->inv_bits = bits;
        const uint32_t nb = 1u << bits;
        PSK_TRY(db->inv_bucket.reserve(ctx->dev, 4 * ((size_t)nb + 2)));
        if (tot) hipLaunchKernelGGL(inv_bucket_kernel, dim3((uint32_t)((tot + 255) / 256)), dim3(256), 0, st, (const uint64_t*)db->inv_key.p, (uint32_t)tot, 2 * K_MARKER - bits, nb, (uint32_t*)db->inv_bucket.p);
        else PSK_HIP(hipMemsetAsync(db->inv_bucket.p, 0, 4 * ((size_t)nb + 2), st));
    }
    PSK_HIP(hipStreamSynchronize(st));
    db->inv_dirty = false;
    return PSK_OK;
}

// Screens nq queries against every reference of the db; the pass matrix [nq][n_refs] STAYS ON THE DEVICE (d_pass, caller-owned).
// `keep` holds the host staging of the async uploads until the caller's next stream synchronisation.
struct ScreenStaging { std::deque<std::vector<MarkerSet>> hq; std::deque<std::vector<uint32_t>> qoff; };
static psk_status screen_many_device(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t nq, double screen_val, int rescue_small,
                                     uint8_t* d_pass, ScreenStaging& keep) {
    const uint32_t n = (uint32_t)db->refs.size();
    if (n == 0 || nq == 0) return PSK_OK;
    hipStream_t st = ctx->stream;
    PSK_TRY(upload_marker_table(ctx, db));
    const double thresh = pow(screen_val, (double)K_MARKER);
    // small jobs: one workgroup per (ref, query). Large jobs: inverted index + count matrix.
    const char* force = getenv("PSK_SCREEN");    // "inv" / "brute" for tests
    const bool use_inv = force ? !strcmp(force, "inv") : ((uint64_t)n * nq >= (1ull << 18));
    if (use_inv) PSK_TRY(build_inverted(ctx, db));
    const uint32_t per = std::max<uint32_t>(1, std::min<uint32_t>(65535, (use_inv ? (1u << 26) : (1u << 24)) / n));   // queries per launch
    // per sub-launch: query marker table + offsets (+ the count matrix of the inverted-index path), side by side in q_a
    const size_t slot_bytes = al256s(sizeof(MarkerSet) * per) + al256s(4 * (size_t)(per + 1));
    const uint32_t n_sub = (nq + per - 1) / per;
    PSK_TRY(ctx->q_a.reserve(slot_bytes * n_sub + 4 * (size_t)std::min<uint64_t>((uint64_t)per * n, use_inv ? ~0ull : (1ull << 22)) + 512));
    uint32_t* d_cnt = (uint32_t*)((char*)ctx->q_a.p + al256s(slot_bytes * n_sub));
    for (uint32_t b = 0, sub = 0; b < nq; b += per, sub++) {
        const uint32_t m = std::min(per, nq - b);
        keep.hq.emplace_back(m); keep.qoff.emplace_back(m + 1, 0u);
        std::vector<MarkerSet>& hq = keep.hq.back(); std::vector<uint32_t>& qoff = keep.qoff.back();
        uint64_t items = 0;
        uint32_t max_qm = 0;
        for (uint32_t i = 0; i < m; i++) {
            const psk_sketch* q = queries[b + i];
            hq[i].p = q->store ? q->store->markers + q->marker_off : nullptr; hq[i].n = (uint32_t)q->n_markers; hq[i].pad = 0;
            qoff[i] = (uint32_t)items; items += q->n_markers;
            max_qm = std::max(max_qm, (uint32_t)q->n_markers);
        }
        qoff[m] = (uint32_t)items;
        if (items >= 0xFFFFFFF0ull) { psk_set_error("too many query markers in one screen launch"); return PSK_ELIMIT; }
        char* Bq = (char*)ctx->q_a.p + slot_bytes * sub;
        MarkerSet* d_q = (MarkerSet*)Bq; uint32_t* d_qoff = (uint32_t*)(Bq + al256s(sizeof(MarkerSet) * per));
        uint8_t* pass_b = d_pass + (size_t)b * n;
        PSK_HIP(hipMemcpyAsync(d_q, hq.data(), sizeof(MarkerSet) * m, hipMemcpyHostToDevice, st));
        ctx->t_begin(K_SCREEN);
        if (use_inv && n <= INV_LDS_REFS && !getenv("PSK_SCREEN_GLOBAL")) {
            static bool lds_attr = false;
            if (!lds_attr) {
                PSK_HIP(hipFuncSetAttribute((const void*)inv_screen_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * INV_LDS_REFS)));
                PSK_HIP(hipFuncSetAttribute((const void*)inv_screen_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * INV_LDS_REFS)));
                lds_attr = true;
            }
            const bool wave_off = getenv("PSK_SCREEN_WAVE") && getenv("PSK_SCREEN_WAVE")[0] == '0';      // "0": one lane per marker, binary search over the whole index (A/B, tests)
            if (!wave_off && db->inv_n)
                hipLaunchKernelGGL(inv_screen_wave_kernel, dim3(m), dim3(512), 4 * (size_t)n, st, (const MarkerSet*)db->d_marker_ptr.p, d_q,
                                   (const uint64_t*)db->inv_key.p, (const uint32_t*)db->inv_ref.p, (const uint32_t*)db->inv_bucket.p, 2 * K_MARKER - db->inv_bits, n, thresh, rescue_small, pass_b);
            else
            hipLaunchKernelGGL(inv_screen_lds_kernel, dim3(m), dim3(512), 4 * (size_t)n, st, (const MarkerSet*)db->d_marker_ptr.p, d_q,
                               (const uint64_t*)db->inv_key.p, (const uint32_t*)db->inv_ref.p, (uint32_t)db->inv_n, n, thresh, rescue_small, pass_b);
        } else if (use_inv) {
            PSK_HIP(hipMemcpyAsync(d_qoff, qoff.data(), 4 * (size_t)(m + 1), hipMemcpyHostToDevice, st));
            PSK_HIP(hipMemsetAsync(d_cnt, 0, 4 * (size_t)m * n, st));
            if (items && db->inv_n)
                hipLaunchKernelGGL(inv_lookup_kernel, dim3((uint32_t)((items + 255) / 256)), dim3(256), 0, st, d_q, d_qoff, m, (uint32_t)items,
                                   (const uint64_t*)db->inv_key.p, (const uint32_t*)db->inv_ref.p, (uint32_t)db->inv_n, n, d_cnt);
            const size_t cells = (size_t)m * n;
            hipLaunchKernelGGL(inv_decide_kernel, dim3((uint32_t)((cells + 255) / 256)), dim3(256), 0, st, (const MarkerSet*)db->d_marker_ptr.p, d_q, n, m, d_cnt, thresh, rescue_small, pass_b);
        } else if (max_qm > 4 * SCREEN_SLICE && (uint64_t)m * n <= (1u << 22)) {   // few pairs of very large marker sets: slice the queries
            PSK_HIP(hipMemsetAsync(d_cnt, 0, 4 * (size_t)m * n, st));
            hipLaunchKernelGGL(screen_slice_kernel, dim3(n, m, (max_qm + SCREEN_SLICE - 1) / SCREEN_SLICE), dim3(256), 0, st, (const MarkerSet*)db->d_marker_ptr.p, d_q, n, d_cnt);
            const size_t cells = (size_t)m * n;
            hipLaunchKernelGGL(inv_decide_kernel, dim3((uint32_t)((cells + 255) / 256)), dim3(256), 0, st, (const MarkerSet*)db->d_marker_ptr.p, d_q, n, m, d_cnt, thresh, rescue_small, pass_b);
        } else {
            hipLaunchKernelGGL(screen_many_kernel, dim3(n, m), dim3(256), 0, st, (const MarkerSet*)db->d_marker_ptr.p, d_q, n, thresh, rescue_small, pass_b);
        }
        ctx->t_end();
    }
    return PSK_OK;
}

// ------------------------------------------------------------------ anchors
// One (reference, query) pair of a launch. Pairs may mix queries (query_many / all-vs-all).
struct PairDesc {
    const uint32_t* r_key; const uint64_t* r_pms;                             // ref index slice: k-mers ascending; r_pms = the seeds' pos<<32|meta in the same order
    const uint32_t* q_key; const uint32_t* q_perm;                            // query index slice: the join walks the query in k-mer order
    const uint32_t* q_pos; const uint32_t* q_meta;                            // query seeds, (contig,pos) order
    const uint32_t* q_kmer;                                                   // their k-mers, same order
    uint32_t q_nc, pad_;                                                      // kept contigs of the query
    const uint32_t* q_seed_pos_base;   // base of the query's store (q_contig_start holds offsets into it)
    const uint32_t* q_contig_start;
    uint64_t q_total_len, r_total_len;
    uint32_t r_n, q_n;
    const uint32_t* r_bucket; uint32_t r_bshift;                              // ref index bucket table (IndexStore::bucket)
    uint32_t r_tab_lines; const ProbeLine* r_tab;                             // ref probe table (null until built: ensure_probe)
};
// sbase[p] = first (pair, query seed) item of pair p in lb/cnt/aoff; cbase[p] = first row of pair p in the chunk table

__device__ __forceinline__ uint32_t find_le(const uint32_t* __restrict__ base, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n - 1;   // largest p in [0,n) with base[p] <= x
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (base[mid] <= x) lo = mid; else hi = mid - 1; }
    return lo;
}

// The pair of a workgroup's first item (or of a chunk-table row) comes from a table filled once per launch sequence
// (pair_table_kernel): a per-workgroup binary search over up to 2^20 pair offsets was a chain of ~20 DEPENDENT global
// loads in front of every workgroup of every kernel below — with nothing else to overlap, that latency was their run time.
__global__ __launch_bounds__(256) void pair_table_kernel(const uint32_t* __restrict__ sbase, const uint32_t* __restrict__ cbase, uint32_t n,
                                                         uint32_t n_tiles, uint32_t n_items, uint32_t n_rows,
                                                         uint32_t* __restrict__ blk_pair, uint32_t* __restrict__ row_pair, uint32_t* __restrict__ misc, uint2* __restrict__ lb_tail) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 64) misc[t] = 0;                  // the launch sequence's status words and counters start from zero (two memsets less)
    if (t == 64) *lb_tail = make_uint2(0, 0);
    if (t < n_tiles) {                        // pair of the first item of every 256-item tile
        const uint64_t x = (uint64_t)t * 256u;
        blk_pair[t] = find_le(sbase, n, x < n_items ? (uint32_t)x : n_items - 1);
    } else if (t - n_tiles < n_rows) {        // pair of every chunk-table row
        row_pair[t - n_tiles] = find_le(cbase, n, t - n_tiles);
    }
}
// pair of item x given the pair of the workgroup's first item: a short forward walk (a pair usually holds far more items
// than a workgroup has threads; pairs without items are stepped over)
__device__ __forceinline__ uint32_t pair_from_hint(const uint32_t* __restrict__ base, uint32_t n, uint32_t x, uint32_t p) {
    while (p + 1 < n && base[p + 1] <= x) p++;
    return p;
}

struct CountOf { __host__ __device__ uint32_t operator()(const uint2& v) const { return v.y; } };

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8, each with its own L2). Renumbering
// them so that consecutive LOGICAL workgroups share an XCD keeps the ~150 workgroups that join one (ref, query) pair
// - and re-read the same index slices - on one L2 instead of filling all eight.
__device__ __forceinline__ uint32_t xcd_block_id() {
    const uint32_t nb = gridDim.x, b = blockIdx.x, xcd = b & 7u, q = nb >> 3, r = nb & 7u;
    return xcd * q + (xcd < r ? xcd : r) + (b >> 3);
}

// The same with the XCDs taking turns every `g` logical workgroups (g ~ the workgroups of a few pairs): neighbouring pairs - one
// query against neighbouring references of its family - are joined at the same time on the eight XCDs, so what the device as a
// whole has in flight is ONE family's reference indices (64 MB for 100 x 5 Mb: they stay in the 256 MB memory-side cache) rather
// than the eight families that eight contiguous eighths of a large batch span. The last nb % (8 g) workgroups keep their number.
__device__ __forceinline__ uint32_t xcd_group_block_id(uint32_t g) {
    const uint32_t nb = gridDim.x, b = blockIdx.x, full = nb / (8u * g) * (8u * g);
    if (b >= full) return b;
    const uint32_t xcd = b & 7u, k = b >> 3;
    return ((k / g) * 8u + xcd) * g + (k % g);
}

// range of index entries of `key` equal to km: bucket table (the k-mer's top bits give ~4 entries), short scan, galloping
// upper bound for repeats
__device__ __forceinline__ void lookup_lane(const uint32_t* __restrict__ key, uint32_t rn, const uint32_t* __restrict__ bucket, uint32_t bshift,
                                            uint32_t km, uint32_t& lo, uint32_t& cnt) {
    lo = 0; cnt = 0;
    uint32_t hi = 0;
    if (rn) {
        const uint32_t bk = km >> bshift;
        lo = bucket[bk]; hi = bucket[bk + 1];
    }
    while (lo < hi && key[lo] < km) lo++;
    if (lo < rn && key[lo] == km) {
        uint32_t step = 1;
        while (lo + step < rn && key[lo + step] == km) step <<= 1;
        uint32_t a = lo + (step >> 1), b = lo + step < rn ? lo + step : rn;   // key[a]==km, key[b]!=km or b==n
        while (a + 1 < b) { uint32_t mid = (a + b) >> 1; if (key[mid] == km) a = mid; else b = mid; }
        cnt = b - lo;
    }
}

// 64-bit anchor total of the workgroup: the offsets the scan produces are 32-bit, the host compares the two totals
// (repeat-rich pairs can exceed 2^32 anchors: a k-mer present 10^5 times on both sides already does)
__device__ __forceinline__ void block_total(uint32_t cnt, uint32_t lb, unsigned long long* __restrict__ block_sum) {
    __shared__ unsigned long long s_ws[4];
    unsigned long long c64 = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c64 += __shfl_xor(c64, o);
    if ((threadIdx.x & 63) == 0) s_ws[threadIdx.x >> 6] = c64;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[lb] = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
}

// WIDE join format (fallback): one lane per (pair, query seed), (lower bound, count) per item
__global__ __launch_bounds__(256) void anchor_count_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                           uint32_t n_pairs, uint32_t n_items,
                                                           uint2* __restrict__ lbcnt_out, unsigned long long* __restrict__ block_sum,
                                                           const uint32_t* __restrict__ blk_pair) {
    const uint32_t lb = xcd_block_id();
    uint32_t i = lb * blockDim.x + threadIdx.x;
    const uint32_t p = pair_from_hint(sbase, n_pairs, i < n_items ? i : n_items - 1, blk_pair[lb]);
    uint32_t cnt = 0;
    if (i < n_items) {
        const PairDesc& P = pairs[p];
        // lane i takes the i-th query seed in K-MER order: neighbouring lanes search neighbouring keys
        const uint32_t iq = i - sbase[p];
        const uint32_t km = P.q_key[iq];
        const uint32_t dst = sbase[p] + P.q_perm[iq];     // results are stored in (contig,pos) order
        uint32_t lo;
        lookup_lane(P.r_key, P.r_n, P.r_bucket, P.r_bshift, km, lo, cnt);
        lbcnt_out[dst] = make_uint2(lo, cnt);      // one 8-byte scattered store per item
    }
    block_total(cnt, lb, block_sum);
}

// PACKED join format (default): per (pair, query seed) y = (ref contig << 1 | ref strand bit) of the first match | count << 24, and
// x = the reference position of the match when there is ONE - nearly all items: the emit kernel then reads nothing at random -
// or, for a k-mer with several matches, the index of the run's first entry in the reference's k-mer index (the emit kernel reads
// the run's positions from there; it used to look the k-mer up again, which is what every item of a Gb-scale pair - six chance
// 15-mer matches per seed - went through). Counts >= 255 or reference contig numbers >= 2^23 raise `need_wide` and the host
// reruns the batch in the wide format.
// The lookup itself is a MERGE: a wave's 64 query k-mers are consecutive in k-mer order, so their matches sit in one short
// stretch of the reference's sorted k-mers. The wave reads the bucket table twice (its first and last k-mer), stages that
// stretch in LDS with coalesced loads and every lane searches it there; only waves whose stretch exceeds JOIN_WIN entries
// (a sparse query against a dense reference) or that straddle two pairs fall back to one independent lookup per lane.
constexpr int JOIN_WIN = 256;
struct PackedCount { __host__ __device__ uint32_t operator()(const uint2& v) const { return v.y >> 24; } };
// The join kernels are bound by the LATENCY of their chain of dependent loads (pair table -> pair descriptor -> query k-mer ->
// bucket table -> reference k-mers -> reference position) and by instruction issue, at a wave residency the register file
// already caps (profiles/r2/r2e_pmc_join_kernels_sq.txt: 79 % of residency waiting, 7 waves per SIMD). The *4 variants put
// JT = 4 tiles of 256 items through every stage TOGETHER - four independent chains in flight per wave instead of one - and
// amortise the pair lookup over 1 024 items (emit: 37.7 -> 26.1 ms, join: 40.7 -> 38.6 ms per 10^5 pairs).
constexpr int JT = 4;
__global__ __launch_bounds__(256) void anchor_join4_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                           uint32_t n_pairs, uint32_t n_items, uint32_t n_tiles,
                                                           uint2* __restrict__ item_out, unsigned long long* __restrict__ block_sum,
                                                           uint32_t* __restrict__ need_wide, const uint32_t* __restrict__ blk_pair,
                                                           uint32_t* __restrict__ pair_cnt, uint32_t xcd_group) {
    __shared__ uint32_t s_key[JT][4][JOIN_WIN];
    const uint32_t lb = xcd_group ? xcd_group_block_id(xcd_group) : xcd_block_id();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t it[JT], p[JT], km[JT], dst[JT], lo[JT], cnt[JT], w_lo[JT], wn[JT];
    bool valid[JT], coop[JT], done[JT];
    uint32_t hint[JT];
#pragma unroll
    for (int t = 0; t < JT; t++) { const uint32_t tile = lb * JT + t; hint[t] = blk_pair[tile < n_tiles ? tile : n_tiles - 1]; }
    // The tile's pair is almost always the pair of its first item: its descriptor and item range are fetched on that assumption
    // together with the next pair's start that confirms it - one round trip instead of two in the kernel's chain of dependent loads
    // (join 37.0 -> 35.4 ms per 10^5 pairs; fetching the reference-side fields the same way as well gains nothing more)
    uint32_t nxs[JT], bs[JT];
    const uint32_t* qk[JT]; const uint32_t* qpm[JT];
#pragma unroll
    for (int t = 0; t < JT; t++) {
        it[t] = (lb * JT + t) * 256u + threadIdx.x;
        valid[t] = it[t] < n_items;
        p[t] = hint[t];
        nxs[t] = p[t] + 1 < n_pairs ? sbase[p[t] + 1] : 0xFFFFFFFFu;
        bs[t] = sbase[p[t]];
        qk[t] = pairs[p[t]].q_key; qpm[t] = pairs[p[t]].q_perm;
    }
#pragma unroll
    for (int t = 0; t < JT; t++) {
        const uint32_t x = valid[t] ? it[t] : n_items - 1;
        if (nxs[t] <= x) {      // a tile across a pair boundary (or pairs without items in between): the walk
            p[t] = pair_from_hint(sbase, n_pairs, x, p[t]);
            bs[t] = sbase[p[t]]; qk[t] = pairs[p[t]].q_key; qpm[t] = pairs[p[t]].q_perm;
        }
    }
#pragma unroll
    for (int t = 0; t < JT; t++) {
        km[t] = 0; dst[t] = 0; lo[t] = 0; cnt[t] = 0; coop[t] = false; done[t] = false; w_lo[t] = 0; wn[t] = 0;
        if (valid[t]) {
            const uint32_t iq = it[t] - bs[t];
            km[t] = qk[t][iq];
            dst[t] = bs[t] + qpm[t][iq];     // results are stored in (contig,pos) order
        }
    }
    // bucket reads of every tile whose wave joins one pair
#pragma unroll
    for (int t = 0; t < JT; t++) {
        const unsigned long long vm = __ballot(valid[t]);
        if (vm) {
            const int l0 = __ffsll((long long)vm) - 1, l1 = 63 - __clzll((long long)vm);
            const uint32_t p0 = __shfl(p[t], l0);
            if (__all(!valid[t] || p[t] == p0)) {
                const PairDesc& P0 = pairs[p0];
                if (P0.r_n == 0) done[t] = true;
                else {
                    const uint32_t km_a = __shfl(km[t], l0), km_b = __shfl(km[t], l1);
                    w_lo[t] = P0.r_bucket[km_a >> P0.r_bshift];
                    wn[t] = P0.r_bucket[(km_b >> P0.r_bshift) + 1] - w_lo[t];
                    coop[t] = wn[t] <= (uint32_t)JOIN_WIN;
                }
            }
        }
    }
    // the stretches of reference k-mers, staged in LDS
#pragma unroll
    for (int t = 0; t < JT; t++) if (coop[t]) {
        const uint32_t* __restrict__ rk = pairs[__shfl(p[t], __ffsll((long long)__ballot(valid[t])) - 1)].r_key;
        for (uint32_t j = lane; j < wn[t]; j += 64) s_key[t][wave][j] = rk[w_lo[t] + j];
    }
    lds_wave_sync();
#pragma unroll
    for (int t = 0; t < JT; t++) {
        if (coop[t]) {
            done[t] = true;
            if (valid[t]) {      // (starting from the lane's own bucket entry instead of a binary search was measured: slower)
                const uint32_t* sk = s_key[t][wave];
                uint32_t a = 0, b = wn[t];
                while (a < b) { const uint32_t mid = (a + b) >> 1; if (sk[mid] < km[t]) a = mid + 1; else b = mid; }
                lo[t] = w_lo[t] + a;
                uint32_t e = a;
                while (e < wn[t] && sk[e] == km[t]) e++;      // equal k-mers share a bucket: the run ends inside the stretch
                cnt[t] = e - a;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < JT; t++)
        if (!done[t] && valid[t]) { const PairDesc& P = pairs[p[t]]; lookup_lane(P.r_key, P.r_n, P.r_bucket, P.r_bshift, km[t], lo[t], cnt[t]); }
    uint64_t pm[JT];
#pragma unroll
    for (int t = 0; t < JT; t++) pm[t] = (valid[t] && cnt[t]) ? pairs[p[t]].r_pms[lo[t]] : 0ull;
#pragma unroll
    for (int t = 0; t < JT; t++) {
        if (valid[t]) {
            uint32_t x = 0, y = 0;
            if (cnt[t]) {
                const uint32_t rmeta = (uint32_t)pm[t];
                x = cnt[t] > 1 ? lo[t] : (uint32_t)(pm[t] >> 32);      // one match: its reference position; a run: where it starts in the reference index
                if (cnt[t] >= 255u || (rmeta >> 24)) { atomicOr(need_wide, 1u); y = (rmeta & 0xFFFFFFu) | (255u << 24); }
                else y = rmeta | (cnt[t] << 24);
            }
            item_out[dst[t]] = make_uint2(x, y);      // one 8-byte scattered store per item
        }
    }
    {   // 64-bit anchor total of the workgroup (the host compares it with the 32-bit offsets the scan produces), and - for
        // anchor_emit_pairs_kernel, which starts every pair at the prefix of these - the anchors per PAIR: one atomic per workgroup
        // when all its items belong to one pair (39 of 40 workgroups of a 5 Mb pair), one per matching item otherwise
        __shared__ unsigned long long s_ws[4];
        __shared__ uint32_t s_wp[4];
        unsigned long long c64 = 0;
#pragma unroll
        for (int t = 0; t < JT; t++) c64 += cnt[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c64 += __shfl_xor(c64, o);
        uint32_t wp = 0xFFFFFFFFu;      // the wave's pair; 0xFFFFFFFE: more than one; 0xFFFFFFFF: no item
        if (pair_cnt) {
#pragma unroll
            for (int t = 0; t < JT; t++) {
                const unsigned long long vm = __ballot(valid[t]);
                if (!vm) continue;
                const uint32_t p0 = __shfl(p[t], __ffsll((long long)vm) - 1);
                const bool uni = __all(!valid[t] || p[t] == p0);
                if (!uni || (wp != 0xFFFFFFFFu && wp != p0)) wp = 0xFFFFFFFEu; else if (wp == 0xFFFFFFFFu) wp = p0;
            }
        }
        if (lane == 0) { s_ws[wave] = c64; s_wp[wave] = wp; }
        __syncthreads();
        if (threadIdx.x == 0) block_sum[lb] = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3];
        if (pair_cnt) {
            uint32_t bp = 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < 4; w++) { const uint32_t x = s_wp[w]; if (x == 0xFFFFFFFFu) continue; if (bp == 0xFFFFFFFFu) bp = x; else if (bp != x) bp = 0xFFFFFFFEu; }
            if (bp < 0xFFFFFFFEu) { if (threadIdx.x == 0) { const unsigned long long tot = s_ws[0] + s_ws[1] + s_ws[2] + s_ws[3]; if (tot) atomicAdd(&pair_cnt[bp], (uint32_t)tot); } }
            else if (bp == 0xFFFFFFFEu) {      // a workgroup across a pair boundary: one atomic per (wave, tile, pair), not per item (same-address atomics serialise)
#pragma unroll
                for (int t = 0; t < JT; t++) {
                    unsigned long long todo = __ballot(valid[t] && cnt[t]);
                    while (todo) {
                        const uint32_t p0 = __shfl(p[t], __ffsll((long long)todo) - 1);
                        const bool mine = valid[t] && cnt[t] && p[t] == p0;
                        uint32_t v = mine ? cnt[t] : 0;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        if (lane == 0) atomicAdd(&pair_cnt[p0], v);
                        todo &= ~__ballot(mine);
                    }
                }
            }
        }
    }
}

// Batches of many SMALL pairs (metagenome contigs, every short one rescued against every reference): the records' positions
// do not depend on the order the pairs are joined in, so the join alone runs REFERENCE-major - one wave per pair, pairs visited
// in the order of `order[]` (pair ids sorted by reference) - and the ~2 000 contigs that probe one reference's 1.3 MB index find
// it in L2 instead of each fetching its hundred scattered lines from HBM. Everything downstream keeps the query-major layout.
__global__ __launch_bounds__(256) void anchor_join_pairs_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                const uint32_t* __restrict__ order, uint32_t n_pairs,
                                                                uint2* __restrict__ item_out, unsigned long long* __restrict__ block_sum,
                                                                uint32_t* __restrict__ need_wide) {
    const uint32_t lb = xcd_block_id();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = lb * 4 + wave;
    uint32_t total = 0;
    if (w < n_pairs) {
        const uint32_t p = order[w];
        const PairDesc P = pairs[p];
        const uint32_t base = sbase[p];
        // Four query seeds per lane go through every stage TOGETHER (k-mer, bucket bounds, a lower-bound search that all four
        // step through in lockstep, the entry found and its successor, the reference position): a stage is one round trip to
        // L2 for four independent loads instead of one - the kernel's time is that chain of round trips (at c = 30 a 5 Mb
        // reference has ~10 entries per bucket: the linear scan of lookup_lane was five of them).
        constexpr int U = 4;
        for (uint32_t i0 = 0; i0 < P.q_n; i0 += 64 * U) {
            uint32_t iq[U], km[U], lo[U], hi[U], k0[U], k1[U], cnt[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) { iq[u] = i0 + u * 64 + lane; ok[u] = iq[u] < P.q_n && P.r_n != 0; km[u] = ok[u] ? P.q_key[iq[u]] : 0; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                lo[u] = 0; hi[u] = 0;
                if (ok[u]) { const uint32_t bk = km[u] >> P.r_bshift; lo[u] = P.r_bucket[bk]; hi[u] = P.r_bucket[bk + 1]; }
            }
            for (;;) {      // lower bound of km in [lo, hi): one probe per seed and step
                bool any = false;
                uint32_t mid[U], kv[U];
#pragma unroll
                for (int u = 0; u < U; u++) { mid[u] = (lo[u] + hi[u]) >> 1; any = any || lo[u] < hi[u]; }
                if (!__any(any)) break;
#pragma unroll
                for (int u = 0; u < U; u++) kv[u] = lo[u] < hi[u] ? P.r_key[mid[u]] : 0;
#pragma unroll
                for (int u = 0; u < U; u++) if (lo[u] < hi[u]) { if (kv[u] < km[u]) lo[u] = mid[u] + 1; else hi[u] = mid[u]; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {      // equal k-mers share a bucket, but the run may be the last thing in the index
                k0[u] = (ok[u] && lo[u] < P.r_n) ? P.r_key[lo[u]] : 0xFFFFFFFFu;
                k1[u] = (ok[u] && lo[u] + 1 < P.r_n) ? P.r_key[lo[u] + 1] : 0xFFFFFFFFu;
            }
            uint64_t pm[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                cnt[u] = 0;
                if (ok[u] && k0[u] == km[u]) {
                    cnt[u] = 1;
                    if (k1[u] == km[u]) {      // a repeat (rare): gallop for the end of the run
                        uint32_t step = 2;
                        while (lo[u] + step < P.r_n && P.r_key[lo[u] + step] == km[u]) step <<= 1;
                        uint32_t a2 = lo[u] + (step >> 1), b2 = lo[u] + step < P.r_n ? lo[u] + step : P.r_n;
                        while (a2 + 1 < b2) { const uint32_t m2 = (a2 + b2) >> 1; if (P.r_key[m2] == km[u]) a2 = m2; else b2 = m2; }
                        cnt[u] = b2 - lo[u];
                    }
                }
                pm[u] = cnt[u] ? P.r_pms[lo[u]] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (iq[u] >= P.q_n) continue;
                uint32_t x = 0, y = 0;
                if (cnt[u]) {
                    const uint32_t rmeta = (uint32_t)pm[u];
                    x = cnt[u] > 1 ? lo[u] : (uint32_t)(pm[u] >> 32);
                    if (cnt[u] >= 255u || (rmeta >> 24)) { atomicOr(need_wide, 1u); y = (rmeta & 0xFFFFFFu) | (255u << 24); }
                    else y = rmeta | (cnt[u] << 24);
                }
                item_out[base + P.q_perm[iq[u]]] = make_uint2(x, y);
                total += cnt[u];
            }
        }
    }
    block_total(total, lb, block_sum);
}

// The same batches - many SMALL pairs, visited reference-major - through the references' PROBE TABLES (common.h): a contig's few hundred
// seeds fall ~500 entries apart in a 5 Mb reference's index at c = 30, so there is nothing to merge and every (pair, query seed) is an
// independent lookup - in the k-mer index a chain of eight dependent reads (two bucket bounds, a four-step search, two keys, the
// position), in the table ONE 64-byte line (a fifth of the lookups a second read of the same line, one in ten the next line). With
// no order to exploit the query is walked in (contig, position) order: the records land where the emit kernels read them with
// coalesced stores, the query's index and its scatter are not touched. Four seeds per lane in flight.
__device__ __forceinline__ int probe_slot_of(const ProbeLine* __restrict__ tab, uint32_t lines, uint32_t km, uint4 K, uint32_t& ln) {
    for (;;) {      // the fifth slot / the next line only where the first four are taken
        int sl = K.x == km ? 0 : K.y == km ? 1 : K.z == km ? 2 : K.w == km ? 3 : -1;
        if (sl < 0 && K.w != PROBE_EMPTY) {
            const uint32_t k4 = tab[ln].k[4];
            if (k4 == km) sl = 4;
            else if (k4 != PROBE_EMPTY) { ln = ln + 1 < lines ? ln + 1 : 0; K = *(const uint4*)(tab + ln); continue; }
        }
        return sl;
    }
}
__global__ __launch_bounds__(256) void anchor_join_probe_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                const uint32_t* __restrict__ order, uint32_t n_pairs,
                                                                uint2* __restrict__ item_out, unsigned long long* __restrict__ block_sum,
                                                                uint32_t* __restrict__ need_wide, uint32_t* __restrict__ aoff_local, uint32_t* __restrict__ pair_cnt) {
    // aoff_local / pair_cnt: the wave walks its pair's seeds in position order anyway - it leaves every item's anchor offset WITHIN the pair (a running count) and
    // the pair's total, so that the offsets of a batch are one scan over its 2 M pairs instead of one over its 700 M items (19 ms per metagenome step)
    const uint32_t lb = xcd_block_id();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t w = lb * 4 + wave;
    uint32_t total = 0;
    if (w < n_pairs) {
        const uint32_t p = order[w];
        const PairDesc& P = pairs[p];
        const uint32_t qn = P.q_n, lines = P.r_tab_lines;
        const ProbeLine* __restrict__ tab = P.r_tab;
        const uint32_t* __restrict__ q_kmer = P.q_kmer;
        uint2* __restrict__ out = item_out + sbase[p];
        uint32_t* __restrict__ loc = aoff_local ? aoff_local + sbase[p] : nullptr;
        uint32_t run = 0;      // anchors of the pair before the items of this step
        constexpr int U = 4;
        for (uint32_t j0 = 0; j0 < qn; j0 += 64 * U) {
            uint32_t km[U], ln[U];
            uint4 K[U];
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t j = j0 + u * 64 + lane; km[u] = j < qn ? q_kmer[j] : 0u; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t j = j0 + u * 64 + lane;
                ln[u] = lines ? probe_line(km[u], lines) : 0u;
                K[u] = (j < qn && lines) ? *(const uint4*)(tab + ln[u]) : make_uint4(PROBE_EMPTY, PROBE_EMPTY, PROBE_EMPTY, PROBE_EMPTY);
            }
            int sl[U];
#pragma unroll
            for (int u = 0; u < U; u++) sl[u] = (j0 + u * 64 + lane < qn && lines) ? probe_slot_of(tab, lines, km[u], K[u], ln[u]) : -1;
            uint2 rec[U];
#pragma unroll
            for (int u = 0; u < U; u++) rec[u] = sl[u] >= 0 ? tab[ln[u]].v[sl[u]] : make_uint2(0u, 0u);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t j = j0 + u * 64 + lane;
                if (j >= qn) continue;
                const uint32_t c = rec[u].y >> 24;
                if (c == 255u) atomicOr(need_wide, 1u);      // a count or contig number the packed entry cannot hold: the host reruns the batch in the wide format
                out[j] = rec[u];
                total += c;
            }
            if (loc) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t j = j0 + u * 64 + lane;
                    const uint32_t c = j < qn ? rec[u].y >> 24 : 0u;
                    uint32_t incl = c;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
                    if (j < qn) loc[j] = run + incl - c;
                    run += __shfl(incl, 63);
                }
            }
        }
        if (pair_cnt && lane == 0) pair_cnt[p] = run;
    }
    block_total(total, lb, block_sum);
}

__global__ __launch_bounds__(256) void anchor_emit_packed4_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                  uint32_t n_pairs, uint32_t n_items, uint32_t n_tiles,
                                                                  const uint2* __restrict__ item, const uint32_t* __restrict__ aoff,
                                                                  uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                                  const uint32_t* __restrict__ blk_pair, const uint32_t* __restrict__ pstart_local) {
    // (pstart_local: aoff holds offsets WITHIN the item's pair - the probe join's own running counts -, the pair's first anchor is added here)
    const uint32_t lb = xcd_block_id();
    uint32_t i[JT], p[JT], c[JT], dst[JT], qp[JT], qm[JT], hint[JT];
    uint2 rec[JT];
    bool act[JT];
#pragma unroll
    for (int t = 0; t < JT; t++) { const uint32_t tile = lb * JT + t; hint[t] = blk_pair[tile < n_tiles ? tile : n_tiles - 1]; }
#pragma unroll
    for (int t = 0; t < JT; t++) {
        i[t] = (lb * JT + t) * 256u + threadIdx.x;
        act[t] = i[t] < n_items;
        rec[t] = act[t] ? item[i[t]] : make_uint2(0, 0);
        dst[t] = act[t] ? aoff[i[t]] : 0;
    }
#pragma unroll
    for (int t = 0; t < JT; t++) {
        c[t] = rec[t].y >> 24;
        act[t] = act[t] && c[t] != 0;
        p[t] = pair_from_hint(sbase, n_pairs, i[t] < n_items ? i[t] : n_items - 1, hint[t]);
        unsigned long long d64 = dst[t];
        if (pstart_local && act[t]) { d64 += pstart_local[p[t]]; dst[t] = (uint32_t)d64; }
        if (act[t] && d64 + c[t] > cap) { atomicOr(err, 2u); act[t] = false; }   // beyond the optimistic capacity: the host reruns the batch
    }
#pragma unroll
    for (int t = 0; t < JT; t++) {
        qp[t] = 0; qm[t] = 0;
        if (act[t]) { const PairDesc& P = pairs[p[t]]; const uint32_t j0 = i[t] - sbase[p[t]]; qp[t] = P.q_pos[j0]; qm[t] = P.q_meta[j0]; }
    }
#pragma unroll
    for (int t = 0; t < JT; t++) {
        if (!act[t]) continue;
        if (c[t] == 1) {
            const uint32_t d = dst[t];
            anc[d] = make_uint4(qp[t], rec[t].x, (rec[t].y & 0xFFFFFEu) | ((rec[t].y ^ qm[t]) & 1u), qm[t] >> 1);   // (q pos, r pos, ref contig << 1 | reverse_match, q contig)
        } else {      // a k-mer with several matches: the record holds where its run starts in the reference index
            const PairDesc& P = pairs[p[t]];
            const uint32_t l = rec[t].x;
            for (uint32_t j = 0; j < c[t]; j++) {
                const uint64_t pm = P.r_pms[l + j];
                const uint32_t rmeta = (uint32_t)pm;
                anc[dst[t] + j] = make_uint4(qp[t], (uint32_t)(pm >> 32), (rmeta & ~1u) | ((rmeta ^ qm[t]) & 1u), qm[t] >> 1);
            }
        }
    }
}

// Emit for batches whose k-mers match MANY times (Gb-scale pairs: a 15-mer has ~6 chance matches in 3 Gb, so a pair of 24 M seeds yields
// 155 M anchors). The item-major kernels above give every item's run to ONE lane - 6.5 sixteen-byte stores a lane at a stride of 104
// bytes: 64 separate requests per store instruction, 0.57 TB/s for 80 GB of anchors. Here the wave works ANCHOR-major: its 64 items'
// records, offsets and query sides go to LDS, then lane k takes output slot first + k, first + 64 + k, ...: the owning item by a binary
// search over the 64 offsets, the match's position from r_pms[run start + j] - consecutive lanes read consecutive entries of a run -
// and ONE contiguous kilobyte of anchors per store instruction. Same anchors at the same places.
__global__ __launch_bounds__(256) void anchor_emit_expand_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                 uint32_t n_pairs, uint32_t n_items,
                                                                 const uint2* __restrict__ item, const uint32_t* __restrict__ aoff,
                                                                 uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                                 const uint32_t* __restrict__ blk_pair) {
    __shared__ uint32_t s_dst[4][64], s_x[4][64], s_y[4][64], s_qp[4][64], s_qm[4][64];
    __shared__ const uint64_t* s_pms[4][64];
    const uint32_t lb = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t i = lb * 256u + threadIdx.x;
    const bool act = i < n_items;
    const uint2 rec = act ? item[i] : make_uint2(0u, 0u);
    const uint32_t d0 = act ? aoff[i] : 0u;
    const uint32_t c = rec.y >> 24;
    uint32_t qp = 0, qm = 0;
    const uint64_t* pms = nullptr;
    if (act && c) {
        const uint32_t p = pair_from_hint(sbase, n_pairs, i, blk_pair[lb]);
        const PairDesc& P = pairs[p];
        const uint32_t j0 = i - sbase[p];
        qp = P.q_pos[j0]; qm = P.q_meta[j0]; pms = P.r_pms;
    }
    // the wave's items with a match, compacted to the front (offsets ascending): lane l of the compacted list
    const unsigned long long live = __ballot(act && c != 0);
    const uint32_t n_live = (uint32_t)__popcll(live);
    if (n_live == 0) return;
    if (act && c) {
        const uint32_t r = (uint32_t)__popcll(live & ((1ull << lane) - 1ull));
        s_dst[wave][r] = d0; s_x[wave][r] = rec.x; s_y[wave][r] = rec.y; s_qp[wave][r] = qp; s_qm[wave][r] = qm; s_pms[wave][r] = pms;
    }
    lds_wave_sync();
    const uint32_t first = s_dst[wave][0];
    const uint32_t last_c = s_y[wave][n_live - 1] >> 24;
    const unsigned long long end = (unsigned long long)s_dst[wave][n_live - 1] + last_c;      // one past the wave's last anchor
    if (end > cap) { if (lane == 0) atomicOr(err, 2u); }      // beyond the optimistic capacity: the host reruns the batch with the true total
    const unsigned long long stop = end < cap ? end : cap;
    for (unsigned long long o = (unsigned long long)first + lane; o < stop; o += 64) {
        uint32_t a = 0, b = n_live;      // owner = last compacted item with dst <= o
        while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (s_dst[wave][mid] <= (uint32_t)o) a = mid; else b = mid; }
        const uint32_t j = (uint32_t)o - s_dst[wave][a];
        const uint32_t x = s_x[wave][a], y = s_y[wave][a], qpa = s_qp[wave][a], qma = s_qm[wave][a];
        if ((y >> 24) == 1) {
            anc[o] = make_uint4(qpa, x, (y & 0xFFFFFEu) | ((y ^ qma) & 1u), qma >> 1);   // (q pos, r pos, ref contig << 1 | reverse_match, q contig)
        } else {
            const uint64_t pm = s_pms[wave][a][x + j];      // x = where the k-mer's run starts in the reference index
            const uint32_t rmeta = (uint32_t)pm;
            anc[o] = make_uint4(qpa, (uint32_t)(pm >> 32), (rmeta & ~1u) | ((rmeta ^ qma) & 1u), qma >> 1);
        }
    }
}

__global__ __launch_bounds__(256) void anchor_emit_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                          uint32_t n_pairs, uint32_t n_items,
                                                          const uint2* __restrict__ lbcnt,
                                                          const uint32_t* __restrict__ aoff,
                                                          uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                          const uint32_t* __restrict__ blk_pair) {
    const uint32_t lb = xcd_block_id();
    uint32_t i = lb * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t p = pair_from_hint(sbase, n_pairs, i, blk_pair[lb]);
    const uint2 lc = lbcnt[i];
    const uint32_t c = lc.y;
    if (c == 0) return;
    const PairDesc& P = pairs[p];
    const uint32_t j0 = i - sbase[p];
    uint32_t l = lc.x, dst = aoff[i];
    if ((uint64_t)dst + c > cap) { atomicOr(err, 2u); return; }   // beyond the optimistic capacity: the host reruns the batch with the true total
    uint32_t qp = P.q_pos[j0], qm = P.q_meta[j0];
    for (uint32_t j = 0; j < c; j++) {
        uint64_t pm = P.r_pms[l + j];        // (pos, meta) of the ref seed, stored in index order
        uint32_t rmeta = (uint32_t)pm;
        anc[dst + j] = make_uint4(qp, (uint32_t)(pm >> 32), (rmeta & ~1u) | ((rmeta ^ qm) & 1u), qm >> 1);   // ref contig << 1 | reverse_match
    }
}

// Emit for batches of many mid-sized pairs (all-vs-all): the join already counted every pair's anchors (pair_cnt), their prefix
// is where each pair's anchors start, so ONE WAVE PER PAIR (EP_W) walks the pair's packed records in item order, JT x 64 at a time,
// with a running offset - no per-item offsets array, no scan over the items: the records are read once (DeviceScan read them,
// wrote 4 B/item of offsets, and the emit kernel read both again). The next round's records are in flight while the current ones
// are written out. Same anchors at the same positions as the scan + emit path.
struct Widen { __host__ __device__ unsigned long long operator()(const uint32_t& v) const { return v; } };
// ... and because the workgroup sees the pair's items in (contig, position) order anyway, it also builds the pair's CHUNK TABLE
// (chunk_heads_kernel's rows: a chunk runs from its head anchor to the first anchor more than FRAGMENT_LENGTH further on the
// query): every lane leaves its items' keys and in-wave offsets in LDS, and once the round's anchors are written wave 0 steps from
// head to head through the round's keys with 64-wide compares - the separate pass over all anchors (16 B each) that
// chunk_heads_kernel makes is gone.
constexpr int EP_W = 1;                 // waves per workgroup (EP_T threads, JT x EP_T items per round). Measured per 10^5 pairs of the all-vs-all step: 8 waves 160 ms, 4: 139.6, 2: 136.6, 1: 134.5 - the fewer waves wait at the round's barrier for wave 0's walk from head to head, the better
constexpr int EP_T = 64 * EP_W;
__global__ __launch_bounds__(EP_T) __attribute__((amdgpu_waves_per_eu(5, 8))) void anchor_emit_pairs_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase, uint32_t n_pairs,
                                                                const uint2* __restrict__ item, const unsigned long long* __restrict__ poff,
                                                                uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                                const uint32_t* __restrict__ cbase, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks) {
    __shared__ __attribute__((aligned(16))) uint32_t s_wt[2][JT][EP_W];
    __shared__ unsigned long long s_key[2][JT * EP_T];     // (q contig << 32 | q pos) + 1 of the items with a match, 0 otherwise
    __shared__ uint32_t s_pre[2][JT * EP_T];               // anchors of the item's wave and sub-tile before it
    const uint32_t p = blockIdx.x;
    const uint32_t s0 = sbase[p], s1 = sbase[p + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long run0 = poff[p], total = poff[p + 1] - run0;
    if (err[5] || *(const unsigned long long*)(err + 16) > cap) {      // this attempt is rerun whatever it produces (see pair_guard_kernel; err + 16: the 64-bit anchor total) - no anchors, no chunk table
        if (chunks != nullptr && threadIdx.x == 0) n_chunks[p] = 0;
        return;
    }
    const bool heads = chunks != nullptr && total >= MIN_ANCHORS;     // fewer: no chain can form, no chunk table, every later kernel skips the pair
    if (chunks != nullptr && !heads && threadIdx.x == 0) n_chunks[p] = 0;
    if (s0 == s1) return;
    const PairDesc& P = pairs[p];
    const uint32_t* __restrict__ q_pos = P.q_pos; const uint32_t* __restrict__ q_meta = P.q_meta;
    unsigned long long run = run0;
    // chunk walk (wave 0; uniform over its lanes): current head anchor, its key + FRAGMENT_LENGTH, rows written
    const uint32_t row0 = heads ? cbase[p] : 0, max_chunks = heads ? cbase[p + 1] - row0 : 0;
    unsigned long long lim1 = 0; uint32_t h = 0, n_rows = 0; bool have = false;
    uint2 nxt[JT];
#pragma unroll
    for (int t = 0; t < JT; t++) { const uint32_t i = s0 + t * (uint32_t)EP_T + threadIdx.x; nxt[t] = i < s1 ? item[i] : make_uint2(0, 0); }
    for (uint32_t c0 = s0, it = 0; c0 < s1; c0 += JT * (uint32_t)EP_T, it++) {
        uint2 rec[JT];
        uint32_t c[JT], incl[JT], qp[JT], qm[JT];
#pragma unroll
        for (int t = 0; t < JT; t++) { rec[t] = nxt[t]; c[t] = rec[t].y >> 24; }
        // the query side of every matching item (does not wait for the offsets), THEN the next records: the wait for the former
        // leaves the latter in flight (vector-memory loads complete in order)
#pragma unroll
        for (int t = 0; t < JT; t++) {
            qp[t] = 0; qm[t] = 0;
            if (c[t]) { const uint32_t j0 = c0 - s0 + t * (uint32_t)EP_T + threadIdx.x; qp[t] = q_pos[j0]; qm[t] = q_meta[j0]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < JT; t++) { const uint32_t i = c0 + (JT + t) * (uint32_t)EP_T + threadIdx.x; nxt[t] = (i >= c0 && i < s1) ? item[i] : make_uint2(0, 0); }      // (a second round in flight was measured: one wave per SIMD fewer, slower)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < JT; t++) {
            uint32_t v = c[t];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(v, o); if (lane >= o) v += x; }
            incl[t] = v;
            if (lane == 63) s_wt[it & 1][t][wave] = v;
        }
        if (heads) {
#pragma unroll
            for (int t = 0; t < JT; t++) {
                s_key[it & 1][t * EP_T + threadIdx.x] = c[t] ? ((((unsigned long long)(qm[t] >> 1)) << 32) | qp[t]) + 1ull : 0ull;
                s_pre[it & 1][t * EP_T + threadIdx.x] = incl[t] - c[t];
            }
        }
        if (EP_W == 1) lds_wave_sync(); else __syncthreads();
        uint32_t agg = 0;
        unsigned long long dst[JT];
#pragma unroll
        for (int t = 0; t < JT; t++) {
            uint32_t before = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < EP_W; w++) { const uint32_t x = s_wt[it & 1][t][w]; before += w < wave ? x : 0; tot += x; }
            dst[t] = run + agg + before + (incl[t] - c[t]);
            agg += tot;
        }
#pragma unroll
        for (int t = 0; t < JT; t++) {
            if (!c[t]) continue;
            if (dst[t] + c[t] > cap) { atomicOr(err, 2u); continue; }   // beyond the optimistic capacity: the host reruns the batch
            const uint32_t d = (uint32_t)dst[t];
            if (c[t] == 1) {
                anc[d] = make_uint4(qp[t], rec[t].x, (rec[t].y & 0xFFFFFEu) | ((rec[t].y ^ qm[t]) & 1u), qm[t] >> 1);   // (q pos, r pos, ref contig << 1 | reverse_match, q contig)
            } else {      // a k-mer with several matches: the record holds where its run starts in the reference index
                const uint32_t l = rec[t].x;
                for (uint32_t j = 0; j < c[t]; j++) {
                    const uint64_t pm = P.r_pms[l + j];
                    const uint32_t rmeta = (uint32_t)pm;
                    anc[d + j] = make_uint4(qp[t], (uint32_t)(pm >> 32), (rmeta & ~1u) | ((rmeta ^ qm[t]) & 1u), qm[t] >> 1);
                }
            }
        }
        if (heads && wave == 0) {      // heads among this round's items: the first item with a match and a key beyond the current head's reach, again and again
            const unsigned long long* sk = s_key[it & 1];
            uint32_t sp = 0;
            while (sp < (uint32_t)(JT * EP_T)) {
                const uint32_t idx = sp + lane;
                const unsigned long long k1 = idx < (uint32_t)(JT * EP_T) ? sk[idx] : 0ull;
                const unsigned long long bal = __ballot(k1 > lim1);      // lim1 = 0 before the pair's first anchor: any match starts the first chunk
                if (!bal) { sp += 64; continue; }
                const uint32_t j = sp + (uint32_t)__ffsll((long long)bal) - 1;
                const uint32_t t = j / (uint32_t)EP_T, w = (j >> 6) & (uint32_t)(EP_W - 1);
                unsigned long long b = run + s_pre[it & 1][j];
                for (uint32_t tt = 0; tt < t; tt++) for (uint32_t ww = 0; ww < (uint32_t)EP_W; ww++) b += s_wt[it & 1][tt][ww];
                for (uint32_t ww = 0; ww < w; ww++) b += s_wt[it & 1][t][ww];
                const uint32_t bc = b < cap ? (uint32_t)b : cap;
                if (have) {
                    if (lane == 0) { if (n_rows < max_chunks) chunks[(size_t)row0 + n_rows] = make_uint2(h, bc); else atomicOr(err, 1u); }
                    n_rows++;
                }
                have = true; h = bc; lim1 = sk[j] + FRAGMENT_LENGTH;
                sp = j + 1;
            }
        }
        run += agg;
    }
    if (heads && wave == 0 && have) {
        const unsigned long long e = run0 + total;
        const uint32_t pend = e < cap ? (uint32_t)e : cap;
        if (lane == 0) {
            if (n_rows < max_chunks) chunks[(size_t)row0 + n_rows] = make_uint2(h, pend); else atomicOr(err, 1u);
            n_chunks[p] = n_rows + 1 < max_chunks ? n_rows + 1 : max_chunks;
        }
    }
}

// pstart from the 64-bit prefix of the pairs' anchor counts (clamped into the optimistically sized anchor arrays)
// (need_wide: the join met a count or contig number its packed records cannot hold - the host reruns the batch in the wide format whatever this attempt
// produces, so every pair is left EMPTY here and the chunk, DP and selection kernels of the attempt have nothing to do: a genome whose k-mers repeat
// 47 000 times spent minutes chaining 24 M clamped anchors before the rerun refused it)
__global__ __launch_bounds__(256) void pair_start64_kernel(const unsigned long long* __restrict__ poff, uint32_t n_pairs, uint32_t* __restrict__ pstart, uint32_t cap,
                                                           const uint32_t* __restrict__ need_wide) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= n_pairs) { const unsigned long long a = poff[p]; pstart[p] = *need_wide ? 0u : (a < cap ? (uint32_t)a : cap); }
}

// An attempt that will be rerun whatever it produces - the join asked for the wide format, or the anchor total does not fit the capacity the arrays were sized
// for (or the 32-bit offsets) - leaves every pair EMPTY: chunk tables, DP and selection then have nothing to do. (A genome whose k-mer repeats 47 000 times, met by
// a context whose arrays a Gb-scale batch had grown, spent nine minutes in the lane-serial DP of 75 M-anchor chunks before its total was looked at.)
__global__ __launch_bounds__(256) void pair_guard_kernel(const uint32_t* __restrict__ need_wide, const unsigned long long* __restrict__ total64, unsigned long long cap,
                                                         uint32_t* __restrict__ pstart, uint32_t n_pairs) {
    if (!*need_wide && *total64 <= cap) return;
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= n_pairs) pstart[p] = 0;
}

// pstart[p] = first anchor of pair p (pstart[n_pairs] = total)
__global__ __launch_bounds__(256) void pair_start_kernel(const uint32_t* __restrict__ aoff, const uint32_t* __restrict__ sbase, uint32_t n_pairs, uint32_t* __restrict__ pstart, uint32_t cap,
                                                         const unsigned long long* __restrict__ bsum, uint32_t n_sum, unsigned long long* __restrict__ total64,
                                                         const uint32_t* __restrict__ need_wide) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (aoff && p <= n_pairs) { const uint32_t a = aoff[sbase[p]]; pstart[p] = *need_wide ? 0u : (a < cap ? a : cap); }   // inside the (optimistically sized) anchor arrays whatever the counts were
    if (n_sum && blockIdx.x == 0) {   // small launches: the 64-bit anchor total here instead of a device-wide reduction (two launches fewer)
        __shared__ unsigned long long s_t[4];
        unsigned long long t = 0;
        for (uint32_t i = threadIdx.x; i < n_sum; i += blockDim.x) t += bsum[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if ((threadIdx.x & 63) == 0) s_t[threadIdx.x >> 6] = t;
        __syncthreads();
        if (threadIdx.x == 0) *total64 = s_t[0] + s_t[1] + s_t[2] + s_t[3];
    }
}

// Chunk table of one pair, one wave per pair. A chunk starts at anchor h and ends before the first anchor b of the
// same pair with (qc, qp) > (qc, qp)(h) + FRAGMENT_LENGTH. The walk from head to head is serial, so the wave stages
// a window of anchor keys in LDS with coalesced loads and finds each boundary with 64-wide compares + ballot
// (a chunk is ~190 anchors at c = 125: three rounds).
constexpr int HEAD_WIN = 2048;
__global__ __launch_bounds__(64) void chunk_heads_kernel(const uint32_t* __restrict__ pstart, const uint4* __restrict__ anc,
                                                         const uint32_t* __restrict__ cbase, uint32_t n_pairs, uint2* __restrict__ chunks,
                                                         uint32_t* __restrict__ n_chunks, uint32_t* __restrict__ err) {
    __shared__ unsigned long long s_key[HEAD_WIN];
    const uint32_t p = blockIdx.x;
    if (p >= n_pairs) return;
    const int lane = threadIdx.x;
    const uint32_t row0 = cbase[p], max_chunks = cbase[p + 1] - row0;
    const uint32_t pend = pstart[p + 1];
    uint32_t h = pstart[p], n = 0;
    if (pend - h < MIN_ANCHORS) { if (lane == 0) n_chunks[p] = 0; return; }   // no chain can form (>= MIN_ANCHORS anchors): no chunk table, every later kernel skips the pair
    uint32_t w0 = h, wn = 0;
    auto load_window = [&](uint32_t from) {
        lds_wave_sync();
        w0 = from; wn = pend - w0 < (uint32_t)HEAD_WIN ? pend - w0 : (uint32_t)HEAD_WIN;
        for (uint32_t i = lane; i < wn; i += 64) { const uint4 a = anc[w0 + i]; s_key[i] = ((unsigned long long)a.w << 32) | a.x; }
        lds_wave_sync();
    };
    if (h < pend) load_window(h);
    while (h < pend) {
        const unsigned long long limit = s_key[h - w0] + FRAGMENT_LENGTH;     // h is always inside the window
        uint32_t sp = h + 1, b = pend;
        for (;;) {
            if (sp >= pend) { b = pend; break; }
            if (sp >= w0 + wn) load_window(sp);
            const uint32_t wend = w0 + wn;
            const uint32_t idx = sp + lane;
            const unsigned long long bal = __ballot(idx < wend && s_key[idx - w0] > limit);
            if (bal) { b = sp + (uint32_t)__ffsll((long long)bal) - 1; break; }
            sp = sp + 64 < wend ? sp + 64 : wend;
        }
        if (lane == 0) { if (n < max_chunks) chunks[(size_t)row0 + n] = make_uint2(h, b); else atomicOr(err, 1u); }
        n++; h = b;
        if (h < pend && (h < w0 || h >= w0 + wn)) load_window(h);
    }
    if (lane == 0) n_chunks[p] = n < max_chunks ? n : max_chunks;
}

// nxt[a] = first anchor of the same pair that starts a new chunk if a chunk starts at a
// COARSE = 1: the successor of every 64th anchor only, into nxt[a / 64]; COARSE = 2: every anchor, searched between the successors of
// the two 64th anchors around it (nxt is monotone within a pair: 6-7 probes next to each other instead of 13 across megabytes;
// 16.6 -> 8.0 + 1.5 ms per launch over the 8 x 3 Gb step's anchors); COARSE = 0: every anchor on its own (few, small pairs).
// (Staging a workgroup's common range of keys in LDS - one round of coalesced loads, the probes as LDS reads - was measured SLOWER: 13.2 ms per
// launch, profiles/r3/experiments/mammalian8_anchor_next_lds_window_kernel_stats.md: the probes of neighbouring lanes already share cache lines.)
template <int COARSE>
__global__ __launch_bounds__(256) void anchor_next_kernel(const uint4* __restrict__ anc,
                                                          const uint32_t* __restrict__ pstart, uint32_t n_pairs,
                                                          const uint32_t* __restrict__ coarse, uint32_t* __restrict__ nxt) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = pstart[n_pairs];      // the grid covers the capacity, the device knows the total
    const uint32_t a = COARSE == 1 ? t * 64u : t;
    if (a >= total) return;
    const uint32_t p = find_le(pstart, n_pairs, a);
    const uint32_t pend = pstart[p + 1];
    const uint64_t key = ((uint64_t)anc[a].w << 32) + (uint64_t)anc[a].x + FRAGMENT_LENGTH;   // first b with (qc,qp) > key
    uint32_t l = a + 1, h = pend;
    if (COARSE == 2) {
        const uint32_t g = a >> 6;
        if (g * 64u >= pstart[p]) { const uint32_t c0 = coarse[g]; l = c0 > l ? c0 : l; }      // the 64th anchor before a is of the same pair: nxt(a) >= its successor
        if ((g + 1) * 64u < pend) { const uint32_t c1 = coarse[g + 1]; h = c1 < h ? c1 : h; }   // ... and <= the successor of the 64th anchor after it
    } else if (h - l > 8192u) {
        // the answer is rarely far: FRAGMENT_LENGTH bases hold a few hundred to a thousand anchors even between Gb-scale genomes, so one
        // probe 8 192 anchors on usually cuts a 27-step search over a 155 M-anchor pair to 13 steps
        const uint32_t far = l + 8192u; const uint64_t kf = ((uint64_t)anc[far].w << 32) | anc[far].x; if (kf > key) h = far; else l = far + 1;
    }
    while (l < h) { uint32_t mid = (l + h) >> 1; uint64_t k2 = ((uint64_t)anc[mid].w << 32) | anc[mid].x; if (k2 <= key) l = mid + 1; else h = mid; }
    nxt[COARSE == 1 ? t : a] = l;
}

// Alternative for a few very large pairs (Gb-scale genomes), where one wave walking 50 000 heads is the critical
// path: nxt[] for every anchor in parallel, then one wave per pair follows nxt[] from the pair's first anchor. The walk is serial, so the wave stages a
// 4 096-entry window of nxt[] in LDS with one round of coalesced loads and lane 0 hops inside it.
constexpr int HOP_WIN = 4096;
__global__ __launch_bounds__(64) void chunk_hops_kernel(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ nxt, const uint32_t* __restrict__ cbase,
                                                         uint32_t n_pairs, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks,
                                                         uint32_t* __restrict__ err) {
    __shared__ uint32_t s_win[HOP_WIN];
    __shared__ uint32_t s_h, s_n;
    const uint32_t p = blockIdx.x;
    if (p >= n_pairs) return;
    const int lane = threadIdx.x;
    const uint32_t row0 = cbase[p], max_chunks = cbase[p + 1] - row0;
    const uint32_t pend = pstart[p + 1];
    uint32_t h = pstart[p], n = 0;
    if (pend - h < MIN_ANCHORS) { if (lane == 0) n_chunks[p] = 0; return; }   // as in chunk_heads_kernel
    while (h < pend) {
        const uint32_t w0 = h, wn = pend - w0 < (uint32_t)HOP_WIN ? pend - w0 : (uint32_t)HOP_WIN;
        for (uint32_t i = lane; i < wn; i += 64) s_win[i] = nxt[w0 + i];
        lds_wave_sync();
        if (lane == 0) {
            while (h < pend && h - w0 < wn) {
                uint32_t e = s_win[h - w0];
                if (n < max_chunks) chunks[(size_t)row0 + n] = make_uint2(h, e); else atomicOr(err, 1u);
                n++; h = e;
            }
            s_h = h; s_n = n;
        }
        lds_wave_sync();
        h = s_h; n = s_n;
    }
    if (lane == 0) n_chunks[p] = n < max_chunks ? n : max_chunks;
}

// Gb-scale pairs: one wave chasing 150 000 heads of a 3 Gb pair is 80 ms of pure latency. A chunk never spans two contigs, so
// the first anchor of every contig is a head whatever came before: HOP_SLICES waves per pair each walk the contigs of their
// slice, once to count their chunks (the rows of the table must stay dense and in order) and once more to write them.
constexpr int HOP_SLICES = 32;
__device__ __forceinline__ uint32_t first_anchor_of_contig(const uint4* __restrict__ anc, uint32_t a, uint32_t b, uint32_t c) {
    while (a < b) { const uint32_t mid = (a + b) >> 1; if (anc[mid].w < c) a = mid + 1; else b = mid; }
    return a;
}
// pass 0: the walk - every slice writes its rows into a scratch table, at the place they would have if every earlier contig held as many
// chunks as its length allows (last seed position / (FRAGMENT_LENGTH + 1) + 1: the bound the table's rows are sized by), and leaves its
// count in slice_cnt; pass 1: the slices' rows copied to their dense places, 64 rows per step. (Walking twice - count, then write - was
// 4.1 ms per 3 Gb pair each time.)
__device__ __forceinline__ uint32_t contig_row_bound(const PairDesc& P, uint32_t c) {
    const uint32_t a = P.q_contig_start[c], b = P.q_contig_start[c + 1];
    return a < b ? P.q_seed_pos_base[b - 1] / (FRAGMENT_LENGTH + 1u) + 1u : 0u;      // a contig without seeds has no anchors and no chunk
}
__global__ __launch_bounds__(64) void chunk_hops_sliced_kernel(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ nxt, const uint4* __restrict__ anc,
                                                                const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ cbase, uint32_t n_pairs,
                                                                uint32_t* __restrict__ slice_cnt, int pass, uint2* __restrict__ scratch, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks,
                                                                uint32_t* __restrict__ err) {
    __shared__ uint32_t s_win[HOP_WIN];
    __shared__ uint32_t s_h, s_n;
    const uint32_t p = blockIdx.x, w = blockIdx.y;
    const int lane = threadIdx.x;
    const uint32_t a = pstart[p], b = pstart[p + 1];
    const PairDesc& P = pairs[p];
    const uint32_t nc = P.q_nc;
    const uint32_t c_lo = (uint32_t)((uint64_t)nc * w / HOP_SLICES), c_hi = (uint32_t)((uint64_t)nc * (w + 1) / HOP_SLICES);
    const bool dead = b - a < MIN_ANCHORS;
    if (pass == 0 && (dead || c_lo == c_hi)) { if (lane == 0) slice_cnt[p * HOP_SLICES + w] = 0; return; }
    if (pass == 1 && dead) { if (lane == 0 && w == 0) n_chunks[p] = 0; return; }
    const uint32_t row0 = cbase[p], max_chunks = cbase[p + 1] - row0;
    uint32_t ub = 0;      // rows the contigs before this slice can hold at most: where the slice's scratch rows start
    for (uint32_t c = lane; c < c_lo; c += 64) ub += contig_row_bound(P, c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ub += __shfl_xor(ub, o);
    if (pass == 1) {
        uint32_t off = 0, total = 0;
        for (int j = 0; j < HOP_SLICES; j++) { const uint32_t c = slice_cnt[p * HOP_SLICES + j]; if ((uint32_t)j < w) off += c; total += c; }
        if (lane == 0 && w == 0) { n_chunks[p] = total < max_chunks ? total : max_chunks; if (total > max_chunks) atomicOr(err, 1u); }
        const uint32_t n = slice_cnt[p * HOP_SLICES + w];
        for (uint32_t i = lane; i < n; i += 64) if (off + i < max_chunks && ub + i < max_chunks) chunks[(size_t)row0 + off + i] = scratch[(size_t)row0 + ub + i];
        return;
    }
    uint32_t h = first_anchor_of_contig(anc, a, b, c_lo);
    const uint32_t hend = c_hi >= nc ? b : first_anchor_of_contig(anc, a, b, c_hi);
    uint32_t n = 0;
    while (h < hend) {
        const uint32_t w0 = h, wn = hend - w0 < (uint32_t)HOP_WIN ? hend - w0 : (uint32_t)HOP_WIN;
        for (uint32_t i = lane; i < wn; i += 64) s_win[i] = nxt[w0 + i];
        lds_wave_sync();
        if (lane == 0) {
            while (h < hend && h - w0 < wn) {
                const uint32_t e = s_win[h - w0];
                if (ub + n < max_chunks) scratch[(size_t)row0 + ub + n] = make_uint2(h, e); else atomicOr(err, 1u);
                n++; h = e;
            }
            s_h = h; s_n = n;
        }
        lds_wave_sync();
        h = s_h; n = s_n;
    }
    if (lane == 0) slice_cnt[p * HOP_SLICES + w] = n;
}

// ---- the chunk table of Gb-scale pairs in ITEM space ---------------------------------------------------------------------------------------------
// anchor_next_kernel finds, for EVERY anchor, the first anchor past its 20 kb window: 155 M searches of 6-7 probes per 3 Gb pair into 16-byte records at
// random (58 GB of 64-byte lines per 8-genome step, profiles/r4/pmc_kernels.json) - for a table of 150 000 heads. But a chunk boundary is a property of the
// QUERY's seed positions, and an anchor-bearing (pair, query seed) item already knows where its anchors start (the scan's offsets): the walk from head to head
// hops over ITEMS; a chunk's row is (offset of the head item, offset of the successor item) - items without anchors have the offset of the next one, so an item
// has anchors iff its offset differs from the next item's.
// HOP_SLICES waves per pair, each over whole contigs (a contig's first anchor is a head whatever came before). The wave stages a window of the query's seed
// positions and of the items' offsets in LDS and walks it TOGETHER: the next head = first item with anchors (64 items per ballot), its successor = first seed more
// than FRAGMENT_LENGTH past it (64 probes 16 seeds apart, then the 16 between). Round 4 searched the successor of every item beforehand (item_next_kernel: 24 M
// gallops per pair for 150 000 heads, 9.4 ms per 8 x 3 Gb step) and walked the result with one lane.
// Pass 0 of chunk_hops_sliced_kernel over items (pass 1 - the copy of the slices' rows to their dense places - is that kernel's own).
__global__ __launch_bounds__(64) void chunk_hops_items_kernel(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ aoff,
                                                               const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase, const uint32_t* __restrict__ cbase,
                                                               uint32_t n_pairs, uint32_t* __restrict__ slice_cnt, uint2* __restrict__ scratch, uint32_t* __restrict__ err) {
    __shared__ __attribute__((aligned(16))) uint32_t s_pos[HOP_WIN], s_ao[HOP_WIN + 4];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));      // (a window starts at any seed: 4-byte aligned 16-byte loads)
    const uint32_t p = blockIdx.x, w = blockIdx.y;
    const int lane = threadIdx.x;
    const PairDesc& P = pairs[p];
    const uint32_t nc = P.q_nc;
    const uint32_t c_lo = (uint32_t)((uint64_t)nc * w / HOP_SLICES), c_hi = (uint32_t)((uint64_t)nc * (w + 1) / HOP_SLICES);
    if (pstart[p + 1] - pstart[p] < MIN_ANCHORS || c_lo == c_hi) { if (lane == 0) slice_cnt[p * HOP_SLICES + w] = 0; return; }
    const uint32_t row0 = cbase[p], max_chunks = cbase[p + 1] - row0;
    uint32_t ub = 0;      // rows the contigs before this slice can hold at most: where the slice's scratch rows start
    for (uint32_t c = lane; c < c_lo; c += 64) ub += contig_row_bound(P, c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ub += __shfl_xor(ub, o);
    const uint32_t seed0 = (uint32_t)(P.q_pos - P.q_seed_pos_base);      // the query's first seed in its store: q_contig_start holds store offsets
    const uint32_t base = sbase[p];
    const uint32_t* __restrict__ qpos = P.q_pos;
    uint32_t n = 0;
    constexpr uint32_t STRIDE = 16;
    for (uint32_t c = c_lo; c < c_hi; c++) {
        uint32_t h = P.q_contig_start[c] - seed0;                  // (seed numbers of the query = item numbers of the pair less `base`)
        const uint32_t hend = P.q_contig_start[c + 1] - seed0;
        uint32_t guess = 0;      // seeds the previous chunk spanned: the next one's successor is looked for around there first
        while (h < hend) {
            const uint32_t w0 = h, wn = hend - w0 < (uint32_t)HOP_WIN ? hend - w0 : (uint32_t)HOP_WIN;
            lds_wave_sync();
            {   // the window: sixteen 16-byte loads per lane in flight (one dword at a time, a window cost eight dependent round trips: more than walking it)
                const uint32_t full = wn & ~3u;
                const uint32_t* __restrict__ gp = qpos + w0; const uint32_t* __restrict__ ga = aoff + base + w0;
                for (uint32_t i0 = 0; i0 < full; i0 += 2048) {
                    u32x4 vp[8], va[8];
#pragma unroll
                    for (int r = 0; r < 8; r++) { const uint32_t x = i0 + (uint32_t)r * 256u + (uint32_t)lane * 4u; if (x < full) { vp[r] = *(const u32x4*)(gp + x); va[r] = *(const u32x4*)(ga + x); } }
#pragma unroll
                    for (int r = 0; r < 8; r++) { const uint32_t x = i0 + (uint32_t)r * 256u + (uint32_t)lane * 4u; if (x < full) { *(uint4*)&s_pos[x] = make_uint4(vp[r].x, vp[r].y, vp[r].z, vp[r].w); *(uint4*)&s_ao[x] = make_uint4(va[r].x, va[r].y, va[r].z, va[r].w); } }
                }
                if (lane < 4) { const uint32_t x = full + (uint32_t)lane; if (x < wn) s_pos[x] = gp[x]; if (x <= wn) s_ao[x] = ga[x]; }      // (the items' offsets hold one entry past the last item)
            }
            lds_wave_sync();
            // (a hop is a chain of dependent LDS round trips: the head's offsets and position are read with its predecessor's closing offset - two round trips per chunk)
            uint32_t i = 0, ao_i = s_ao[0], ao_n = s_ao[1], pos_i = s_pos[0];
            const uint64_t pos_last = s_pos[wn - 1];
            for (;;) {
                // the next head: first item of the window at or after i with anchors (nearly always item i itself)
                bool found = i < wn && ao_n != ao_i;
                if (!found) {
                    while (i < wn) {
                        const uint32_t x = i + lane;
                        const unsigned long long m = __ballot(x < wn && s_ao[x + 1] != s_ao[x]);
                        if (m) { i += (uint32_t)__ffsll((long long)m) - 1u; found = true; break; }
                        i += 64;
                    }
                    if (!found) { h = w0 + wn; break; }
                    ao_i = s_ao[i]; pos_i = s_pos[i];
                }
                const uint64_t target = (uint64_t)pos_i + FRAGMENT_LENGTH;
                uint32_t e = 0xFFFFFFFFu;      // first item of the window past the head's fragment; wn = none in the window
                if (pos_last <= target) e = wn;
                else {
                    if (guess > 32u && i + guess + 32u <= wn) {      // 64 consecutive seeds around where the last chunk ended
                        const uint32_t x0 = i + guess - 32u;
                        const unsigned long long m = __ballot((uint64_t)s_pos[x0 + lane] > target);
                        if (m && !(m & 1ull)) e = x0 + (uint32_t)__ffsll((long long)m) - 1u;
                    }
                    if (e == 0xFFFFFFFFu) {
                        uint32_t lo = i + 1;      // s_pos[lo - 1] <= target
                        for (;;) {
                            const uint32_t x = lo + (uint32_t)lane * STRIDE + (STRIDE - 1);      // last seed of the lane's group
                            const unsigned long long m = __ballot(x >= wn || (uint64_t)s_pos[x] > target);      // (true from some lane on: positions ascend)
                            if (m) { lo += ((uint32_t)__ffsll((long long)m) - 1u) * STRIDE; break; }
                            lo += 64 * STRIDE;
                        }
                        const uint32_t x = lo + (uint32_t)lane;
                        const unsigned long long m = __ballot(lane < (int)STRIDE && x < wn && (uint64_t)s_pos[x] > target);
                        e = lo + (uint32_t)__ffsll((long long)m) - 1u;      // (m != 0: the group's last seed, or the window's, is past the target)
                    }
                    guess = e - i;
                }
                uint32_t a1;
                if (e < wn) { a1 = s_ao[e]; ao_n = s_ao[e + 1]; pos_i = s_pos[e]; }
                else if (w0 + wn == hend) a1 = s_ao[wn];          // (the contig's end closes its last chunk)
                else if (i > 0) { h = w0 + i; break; }            // the successor lies beyond the window: stage again from this head
                else {      // more than HOP_WIN seeds inside one fragment (c < 5): search the rest of the contig in global memory
                    uint32_t lo = w0 + wn, hi = hend;
                    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)qpos[mid] <= target) lo = mid + 1; else hi = mid; }
                    a1 = aoff[base + lo];
                    if (lane == 0) { if (ub + n < max_chunks) scratch[(size_t)row0 + ub + n] = make_uint2(ao_i, a1); else atomicOr(err, 1u); }
                    n++; h = lo;
                    break;
                }
                if (lane == 0) { if (ub + n < max_chunks) scratch[(size_t)row0 + ub + n] = make_uint2(ao_i, a1); else atomicOr(err, 1u); }
                n++;
                i = e; ao_i = a1;
                if (e >= wn) { h = w0 + wn; break; }
            }
        }
    }
    if (lane == 0) slice_cnt[p * HOP_SLICES + w] = n;
}

// ------------------------------------------------------------------ chaining
struct ChunkOut { uint32_t anchors, seeds, n_intervals, n_cand; uint32_t left, right; uint64_t cov_q; };
// One field of the 32-byte candidate-chain records {score, q0, q1, r0, r1, anchors, ref contig, state}: element i of the field at p[8 i]. The fields were eight arrays of
// their own until round 5: a chunk holds one or two candidates, so the selection read one 64-byte line PER FIELD per chunk - 1.0-1.9 kB of HBM traffic per candidate
// (profiles/r5/pmc_kernels.json) for 32 bytes of content, at 5 TB/s: its whole run time. As records a chunk's candidates are one line.
template <class T> struct Strided {
    T* p;
    __host__ __device__ __forceinline__ T& operator[](size_t i) const { return p[i * 8]; }
};

struct ChainArgs {
    const uint4* anc;      // anchors, array of (q pos, r pos, ref contig << 1 | reverse_match, q contig): a lane's chunk is one contiguous run of 16-byte records
    const uint2* chunks; const uint32_t* n_chunks; const uint32_t* cbase; uint32_t n_pairs, n_rows;
    const uint32_t* row_pair;   // pair of every row of the chunk table
    const uint32_t* row_order;  // rows by chunk length, longest first, rows without a chunk last (null: table order) - the DP kernels that put several chunks in a wave
    const PairDesc* pairs;
    ChunkOut* out;
    // serial-path scratch, one entry per anchor
    int32_t* sc_f; uint32_t *sc_ptr, *sc_root, *sc_depth, *sc_best;
    Strided<int32_t> c_score; Strided<uint32_t> c_q0, c_q1, c_r0, c_r1, c_n, c_state, c_rc;   // candidate chains, chunk s writes at [s, s + n_cand): fields of 32-byte records
    uint32_t two_c; int band; int force_serial; int lane_dp;
    uint32_t cap;      // anchors the arrays hold (chunk_seeds_kernel's bound on what a chunk row may point at)
    int dp_prune;      // the lane / quad DP kernels score the far part of the band only where it could win ($PSK_DP_PRUNE=0: always)
    uint32_t* ovf_list; uint32_t* ovf_count;   // rows the lane kernel hands to the wave kernel (more than LANE_TREES qualifying chain trees, >= 16 384 anchors)
    uint32_t* stats;   // [1] chunks / [3] pairs that took a serial fallback (rare paths only: a counter every wave bumps
                       // serialises the whole launch on one L2 address)
};

constexpr int RING = 128;   // power of two > CHAIN_BAND
constexpr int RMAX = 256;   // chain trees per chunk handled in LDS
constexpr int CHAIN_WAVES = 4;


// number of query seeds on contig qc with pos in [lo, hi]
__device__ uint32_t seeds_between(const PairDesc& P, uint32_t qc, uint32_t lo, uint32_t hi) {
    const uint32_t* __restrict__ pos = P.q_seed_pos_base;
    uint32_t a = P.q_contig_start[qc], b = P.q_contig_start[qc + 1];
    uint32_t l = a, r = b;
    while (l < r) { uint32_t m = (l + r) >> 1; if (pos[m] < lo) l = m + 1; else r = m; }
    uint32_t first = l; r = b;
    if (first + 256u < b && pos[first + 256u] > hi) r = first + 256u;      // (a chunk spans FRAGMENT_LENGTH bases: ~160 seeds at c = 125 - eight probes instead of sixteen)
    while (l < r) { uint32_t m = (l + r) >> 1; if (pos[m] <= hi) l = m + 1; else r = m; }
    return l - first;
}

// Serial restatement of the oracle's per-chunk body, run by ONE lane on global scratch. Used for
// chunks the LDS path cannot hold (many chain trees / candidates) and as an in-GPU cross-check.
__device__ uint32_t chain_chunk_serial(const ChainArgs& A, uint32_t s, uint32_t e) {
    for (uint32_t x = s; x < e; x++) {
        int32_t bs = ANCHOR_SCORE2; uint32_t bp = x;
        uint32_t qx = A.anc[x].x, rx = A.anc[x].y, mx = A.anc[x].z;
        for (uint32_t y = x; y-- > s && x - y <= (uint32_t)A.band;) {
            if (A.anc[y].z != mx) continue;
            int64_t dq = (int64_t)qx - (int64_t)A.anc[y].x;
            if (dq > BP_CHAIN_BAND) break;
            int64_t dr = (mx & 1) ? (int64_t)A.anc[y].y - (int64_t)rx : (int64_t)rx - (int64_t)A.anc[y].y;
            if (dq <= 0 || dr <= 0) continue;
            int64_t gap = dq > dr ? dq - dr : dr - dq;
            if (gap > MAX_GAP_LENGTH) continue;
            int32_t sc = A.sc_f[y] + ANCHOR_SCORE2 - (int32_t)gap;
            if (sc > bs) { bs = sc; bp = y; }
        }
        A.sc_f[x] = bs;
        if (bp == x) { A.sc_root[x] = x; A.sc_depth[x] = 1; }
        else { A.sc_root[x] = A.sc_root[bp]; A.sc_depth[x] = A.sc_depth[bp] + 1; }
        A.sc_best[x] = 0xFFFFFFFFu;
    }
    for (uint32_t x = s; x < e; x++) { uint32_t rt = A.sc_root[x]; uint32_t b = A.sc_best[rt]; if (b == 0xFFFFFFFFu || A.sc_f[x] > A.sc_f[b]) A.sc_best[rt] = x; }
    uint32_t nc = 0;
    for (uint32_t x = s; x < e; x++) {
        if (A.sc_root[x] != x) continue;
        uint32_t b = A.sc_best[x];
        if (A.sc_depth[b] < MIN_ANCHORS || A.sc_f[b] < MIN_SCORE2) continue;
        uint32_t ra = A.anc[x].y, rb = A.anc[b].y;
        A.c_score[s + nc] = A.sc_f[b]; A.c_q0[s + nc] = A.anc[x].x; A.c_q1[s + nc] = A.anc[b].x;
        A.c_r0[s + nc] = ra < rb ? ra : rb; A.c_r1[s + nc] = ra < rb ? rb : ra; A.c_n[s + nc] = A.sc_depth[b];
        A.c_rc[s + nc] = A.anc[x].z >> 1;
        nc++;
    }
    return nc;
}


// ---- lane-per-chunk DP ---------------------------------------------------------------------------------------
// chain_chunk's DP step is a 64-lane affair for a band of ~20 predecessors, and the kernel is VALU-issue bound
// (profiles/r1d_overlap.md). Here ONE LANE owns one chunk: the last LANE_N anchors (q, r, ref contig|strand, f) live
// in registers as a shift register, every (anchor, predecessor) pair is ~25 branch-free instructions with no
// cross-lane traffic, and 64 chunks advance per wave step. Per anchor it leaves f, the tree id and the depth in
// sc_f / sc_root / sc_depth; chain_chunk_kernel then only aggregates the trees and emits candidates.
// The register file is laid out for the band: LANE_N >= band (band = 2500/c: 20 at c = 125).
constexpr int LANE_N = 24;          // predecessors held per lane (multiple of 4)
constexpr int LANE_WAVES = 2;
constexpr int LANE_TREES = 4;        // qualifying chain trees per chunk kept in registers
#ifndef LANE_NEAR_N
#define LANE_NEAR_N 3
#endif
constexpr int LANE_NEAR = LANE_NEAR_N;    // predecessors that are always scored; the rest of the band only where it could win. >= 3: the far loop reads the register window only, so it must start behind the step's own four anchors. Measured on the 10 000 x 10 000 step's DP: 8: 147, 6: 136, 5: 131, 4: 126, 3: 122 ms

// XT: further tree slots per lane in LDS (0, or LANE_XTREES for Gb-scale pairs: there a seed has ~6 chance 15-mer matches beside the
// true one, the band of 20 ANCHORS reaches back only ~3 seeds, a true chain breaks wherever three seeds in a row do not match and a
// chunk holds 5-15 qualifying trees - with four slots most chunks went to the wave-per-chunk kernel, 7.7 of 10 ms per 3 Gb pair)
constexpr int LANE_XTREES = 12;
template <int W, int XT>      // window depth: the band rounded up to a multiple of four (20 at c = 125; 24 covers c >= 105)
__device__ __forceinline__ void chain_lane_body(const ChainArgs& A, const uint32_t rows_per_wave) {
    __shared__ uint32_t s_rd[LANE_WAVES][32][64];     // tree id << 14 | depth of the last 32 anchors, per lane
    __shared__ unsigned long long s_xk[LANE_WAVES][XT ? XT : 1][XT ? 64 : 1];     // slots 4 .. 4 + XT - 1: best anchor key
    __shared__ uint32_t s_xr[LANE_WAVES][XT ? XT : 1][XT ? 64 : 1];               // ... and root
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t slot_i = (blockIdx.x * LANE_WAVES + wave) * rows_per_wave + lane;
    const uint32_t slot = A.row_order && (uint32_t)lane < rows_per_wave && slot_i < A.n_rows ? A.row_order[slot_i] : slot_i;      // (rows by chunk length: row_len_kernel)
    uint32_t s = 0, e = 0;
    bool mine = false, real = false;     // real: a row of the chunk table that holds a chunk; mine: this lane chains it
    const uint32_t pair = A.row_pair[slot < A.n_rows ? slot : A.n_rows - 1];
    if ((uint32_t)lane < rows_per_wave && slot_i < A.n_rows) {
        if (slot - A.cbase[pair] < A.n_chunks[pair]) {
            const uint2 se = A.chunks[slot];
            s = se.x; e = se.y;
            real = true;
            mine = e > s && e - s < 16384;
        }
    }
    // LD = 2: a lane asks for 128 contiguous bytes - eight anchors, a whole cache line - per load and walks them as two steps of four: with 64 bytes per
    // load the other half of every line was fetched again a step later (the kernel's counters: 1.8 x its anchors' bytes, and once the far part of the band
    // is rarely scored that traffic, not the instruction count, is what the kernel waits for)
    constexpr int LD = XT ? 1 : 2;
    const uint32_t s_al = s & ~(4u * LD - 1u);
    const uint32_t len = mine ? e - s_al : 0;          // steps this lane takes part in (the first s - s_al are idle)
    LanePred P[W];
#pragma unroll
    for (int i = 0; i < W; i++) { P[i].q1 = 0; P[i].u = 0; P[i].m = 0xFFFFFFFFu; P[i].f1 = -1; }      // (score - 1 of an EMPTY entry: below every real one, see the far bound)
    // Chain trees that can yield a candidate, at most LANE_TREES per chunk, keyed by the local index of their ROOT
    // anchor. A tree gets a slot when its first anchor with score >= MIN_SCORE2 appears (such an anchor has depth >= 3,
    // and lower-scoring anchors can never be the tree's best once one exists); the many single-anchor trees of
    // spurious matches never take one. Slot: best anchor key f<<28 | (16383 - local index)<<14 | depth, its (q, r).
    unsigned long long bk[LANE_TREES];
    uint32_t sroot[LANE_TREES];      // (the best anchor's q and r are read back from the anchor array at the end: its index is in the key)
#pragma unroll
    for (int j = 0; j < LANE_TREES; j++) { bk[j] = 0; sroot[j] = 0xFFFFFFFFu; }
    uint32_t S = 0;
    bool ovf = false;
    uint32_t (*rd)[64] = s_rd[wave];      // root index << 14 | depth of the last 32 anchors
    const int band = A.band;
    // The FAR part of the band - predecessors more than LANE_NEAR anchors back - is scored only where it could win (the rule of chain_quad_deep_kernel): a
    // predecessor y scores f[y] + ANCHOR_SCORE2 - gap <= f[y] + ANCHOR_SCORE2 and equal scores go to the NEARER one, so an anchor whose best near score
    // reaches the largest far f + ANCHOR_SCORE2 is done; and an anchor whose diagonal is not within MAX_GAP_LENGTH of any far entry's (far_diag: one bit
    // per 1024 diagonals mod 32, two bits per entry, rebuilt every 16 steps) has no far predecessor at all - the chance match off the chain. The wave
    // decides: one lane that needs the far part has all 64 score it (same results). Not for the Gb-scale kernel (XT: most anchors there are chance matches).
    const bool prune = XT == 0 && A.dp_prune != 0;
    constexpr int NR = LANE_NEAR;
    uint32_t far_diag = 0;
    for (uint32_t tb = 0; __any(tb < len); tb += 4 * LD) {
      uint4 an[4 * LD];
#pragma unroll
      for (int i = 0; i < 4 * LD; i++) an[i] = make_uint4(0, 0, 0, 0);
      if (tb < len) {
#pragma unroll
          for (int i = 0; i < 4 * LD; i++) an[i] = A.anc[s_al + tb + i];      // (the array ends in 64 spare records)
      }
#pragma unroll
      for (int h = 0; h < LD; h++) {
        const uint32_t t0 = tb + 4u * h, x0 = s_al + t0;
        const uint32_t qs[4] = {an[4 * h].x, an[4 * h + 1].x, an[4 * h + 2].x, an[4 * h + 3].x}, rs[4] = {an[4 * h].y, an[4 * h + 1].y, an[4 * h + 2].y, an[4 * h + 3].y},
                       ms[4] = {an[4 * h].z, an[4 * h + 1].z, an[4 * h + 2].z, an[4 * h + 3].z};
        LanePred nw[4];
        int32_t ftop[4] = {-1, -1, -1, -1};      // largest f - 1 among the far entries of the step's anchor u: P[NR - u .. W - 1]
        if (prune) {
            if ((t0 & 63u) == 0) {
                far_diag = 0;
#pragma unroll
                for (int i = NR - 3; i < W; i++) far_diag |= __builtin_amdgcn_alignbit(3u, 3u, 32u - (((P[i].u - (uint32_t)MAX_GAP_LENGTH) >> 10) & 31u));
            } else {
#pragma unroll
                for (int i = NR - 3; i <= NR; i++) far_diag |= __builtin_amdgcn_alignbit(3u, 3u, 32u - (((P[i].u - (uint32_t)MAX_GAP_LENGTH) >> 10) & 31u));      // the four that turned far
            }
            // (only entries within BP_CHAIN_BAND of the step's FIRST anchor count - the later ones lie further on: where anchors are sparse, pairs 10 % apart,
            // a chain that broke at a long gap leaves its high scores in the window for twenty anchors, out of reach but above everything the new chain has)
            const uint32_t q0 = qs[0] + 1u;
            int32_t m = -1;
#pragma unroll
            for (int i = NR; i < W; i++) { const int32_t f = q0 - P[i].q1 <= (uint32_t)BP_CHAIN_BAND ? P[i].f1 : -1; m = f > m ? f : m; }
            ftop[0] = m;
#pragma unroll
            for (int u = 1; u < 4; u++) { const int32_t f = q0 - P[NR - u].q1 <= (uint32_t)BP_CHAIN_BAND ? P[NR - u].f1 : -1; m = f > m ? f : m; ftop[u] = m; }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t x = x0 + u, t = t0 + u;
            const bool act = x >= s && x < e && mine;
            const uint32_t qx = qs[u], rx = rs[u], mx = ms[u];
            const uint32_t ux = lane_diag(qx, rx, 0u - (mx & 1u));
            int32_t best = 0;
#pragma unroll
            for (int d = 1; d <= NR; d++) {
                if (d <= band) {
                    const int32_t k = d <= u ? lane_eval2(qx, ux, mx, nw[u - d], d) : lane_eval2(qx, ux, mx, P[d - 1 - u], d);
                    best = k > best ? k : best;
                }
            }
            // (... and the far entries lie further back on the query than the nearest of them: none is within BP_CHAIN_BAND if that one is not - sparse anchors,
            // pairs 10 % apart, restart their chains every few anchors and would otherwise ask for the far part each time)
            if (!prune || __any(act && ftop[u] >= 0 && (best >> 7) < ftop[u] + 1 + ANCHOR_SCORE2 && ((far_diag >> ((ux >> 10) & 31u)) & 1u) && qx + 1u - P[NR - u].q1 <= (uint32_t)BP_CHAIN_BAND)) {
#pragma unroll
                for (int d = NR + 1; d <= W; d++) {
                    if (d <= band) {
                        const int32_t k = lane_eval2(qx, ux, mx, P[d - 1 - u], d);
                        best = k > best ? k : best;
                    }
                }
            }
            int32_t f = ANCHOR_SCORE2; uint32_t ridx = x - s, dep = 1;
            if (best > 0) {
                f = best >> 7;
                const uint32_t v = rd[(t - (127u - ((uint32_t)best & 127u))) & 31u][lane];
                ridx = v >> 14; dep = (v & 16383u) + 1;
            }
            rd[t & 31u][lane] = (ridx << 14) | dep;
            nw[u].q1 = qx + 1u; nw[u].u = ux; nw[u].m = act ? mx : 0xFFFFFFFFu; nw[u].f1 = act ? f - 1 : -1;
            if (act && f >= MIN_SCORE2) {
                const unsigned long long k64 = ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - (x - s)) << 14) | dep;
                bool found = false;
#pragma unroll
                for (int j = 0; j < LANE_TREES; j++) {
                    const bool hit = sroot[j] == ridx;
                    found = found || hit;
                    if (hit && k64 > bk[j]) bk[j] = k64;
                }
                if (XT && !found && S > (uint32_t)LANE_TREES) {      // the LDS slots (a lane's own column: no other lane touches it)
                    const uint32_t nx = S - LANE_TREES < (uint32_t)XT ? S - LANE_TREES : (uint32_t)XT;
                    for (uint32_t j = 0; j < nx; j++)
                        if (s_xr[wave][j][lane] == ridx) { found = true; if (k64 > s_xk[wave][j][lane]) s_xk[wave][j][lane] = k64; break; }
                }
                if (!found) {
                    if (S >= (uint32_t)(LANE_TREES + XT)) ovf = true;
                    else if (XT && S >= (uint32_t)LANE_TREES) { s_xr[wave][S - LANE_TREES][lane] = ridx; s_xk[wave][S - LANE_TREES][lane] = k64; }
#pragma unroll
                    for (int j = 0; j < LANE_TREES; j++) if (S == (uint32_t)j) { sroot[j] = ridx; bk[j] = k64; }
                    S++;
                }
            }
        }
        // shift the register window by four anchors
#pragma unroll
        for (int i = W - 1; i >= 4; i--) P[i] = P[i - 4];
        P[0] = nw[3]; P[1] = nw[2]; P[2] = nw[1]; P[3] = nw[0];
      }
    }
    if ((uint32_t)lane < rows_per_wave && slot < A.n_rows && real) {
        if (mine && !ovf) {
            // candidates in ROOT order (slots were taken in order of first qualifying anchor): pick the smallest root left
            uint32_t nc = 0, last = 0;
            for (uint32_t c = 0; c < S; c++) {
                uint32_t pick = 0xFFFFFFFFu; unsigned long long k = 0;
#pragma unroll
                for (int j = 0; j < LANE_TREES; j++)
                    if (sroot[j] != 0xFFFFFFFFu && (c == 0 || sroot[j] > last) && sroot[j] < pick) { pick = sroot[j]; k = bk[j]; }
                if (XT && S > (uint32_t)LANE_TREES)
                    for (uint32_t j = 0; j < S - LANE_TREES; j++) {
                        const uint32_t rt = s_xr[wave][j][lane];
                        if ((c == 0 || rt > last) && rt < pick) { pick = rt; k = s_xk[wave][j][lane]; }
                    }
                last = pick;
                const uint32_t xb = s + (16383u - (uint32_t)((k >> 14) & 16383u));      // the tree's best anchor
                const uint32_t q1 = A.anc[xb].x, rb = A.anc[xb].y;
                const uint32_t xr = s + pick, ra = A.anc[xr].y, o = s + nc;
                A.c_score[o] = (int32_t)(uint32_t)(k >> 28); A.c_q0[o] = A.anc[xr].x; A.c_q1[o] = q1;
                A.c_r0[o] = ra < rb ? ra : rb; A.c_r1[o] = ra < rb ? rb : ra;
                A.c_n[o] = (uint32_t)(k & 16383u); A.c_rc[o] = A.anc[xr].z >> 1;
                nc++;
            }
            ChunkOut o{};
            o.n_cand = nc; o.left = 0xFFFFFFFFu; o.right = 0;
            A.out[slot] = o;
        } else {
            A.ovf_list[atomicAdd(A.ovf_count, 1u)] = slot;       // rare: the wave kernel redoes this chunk
        }
    }
}

// W = 20 fits three waves per SIMD (168 registers; the 24-deep window needs 192 and runs two): the kernel is VALU-issue bound and
// a third wave fills issue slots that two leave empty
__global__ __launch_bounds__(64 * LANE_WAVES) __attribute__((amdgpu_waves_per_eu(3, 8))) void chain_lane20_kernel(ChainArgs A, uint32_t rows_per_wave) { chain_lane_body<20, 0>(A, rows_per_wave); }
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_lane20x_kernel(ChainArgs A, uint32_t rows_per_wave) { chain_lane_body<20, LANE_XTREES>(A, rows_per_wave); }
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_lane_kernel(ChainArgs A, uint32_t rows_per_wave) { chain_lane_body<LANE_N, 0>(A, rows_per_wave); }

// ---- four lanes per chunk, for launches too small to fill the chip with one lane per chunk -----------------
// (the headline search: 100 pairs = 22 k chunks). Lane j of a quad owns the anchors whose index is j mod 4: ownership
// never moves, so the 24-deep window becomes four 6-deep ones with static register indices, each lane scores 6
// predecessors per anchor instead of 24, and two quad DPP exchanges give all four the best key. The step's dependent
// instruction chain - what a lone wave per SIMD is bound by - is ~2.4 x shorter; throughput per chunk is lower, so
// the one-lane kernel stays for big launches.
__device__ __forceinline__ uint32_t lane_eval_d(uint32_t qx, uint32_t ux, uint32_t mx, const LaneAnchor& y, uint32_t d, int band) {
    const int32_t dq = (int32_t)(qx - y.q);
    const int32_t t = (int32_t)(ux - y.u), nt = (int32_t)(y.u - ux);
    const int32_t dr = dq - t;
    const int32_t gap = t > nt ? t : nt;
    const int32_t scp = y.f - gap;
    const uint32_t z = y.m ^ mx;
    const uint32_t bad = (uint32_t)(dq - 1) | (uint32_t)(BP_CHAIN_BAND - dq) | (uint32_t)(dr - 1) | (uint32_t)(MAX_GAP_LENGTH - gap) |
                         (uint32_t)(scp - 1) | z | (0u - z) | (uint32_t)(band - (int32_t)d);
    const uint32_t ok = (uint32_t)((int32_t)~bad >> 31);
    return ((((uint32_t)scp << 7) + (((uint32_t)ANCHOR_SCORE2 << 7) | 127u) - d)) & ok;
}

constexpr int QUAD_N = LANE_N / 4;
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_quad_kernel(ChainArgs A) {
    __shared__ uint32_t s_rd[LANE_WAVES][32][16];     // root index << 14 | depth of the last 32 anchors, per quad
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, quad = lane >> 2;
    const uint32_t j = lane & 3;
    const uint32_t slot_i = (blockIdx.x * LANE_WAVES + wave) * 16 + quad;
    const uint32_t slot = A.row_order && slot_i < A.n_rows ? A.row_order[slot_i] : slot_i;
    uint32_t s = 0, e = 0;
    bool mine = false, real = false;
    const uint32_t pair = A.row_pair[slot < A.n_rows ? slot : A.n_rows - 1];
    if (slot < A.n_rows && slot - A.cbase[pair] < A.n_chunks[pair]) {
        const uint2 se = A.chunks[slot];
        s = se.x; e = se.y;
        real = true;
        mine = e > s && e - s < 16384;
    }
    const uint32_t s_al = s & ~3u;
    const uint32_t len = mine ? e - s_al : 0;
    // window: entry i = the anchor 4 i before this lane's latest one (scalar arrays: a struct array with conditional
    // whole-struct moves ends up in scratch memory)
    uint32_t Wq[QUAD_N], Wr[QUAD_N], Wm[QUAD_N]; int32_t Wf[QUAD_N];
#pragma unroll
    for (int i = 0; i < QUAD_N; i++) { Wq[i] = 0; Wr[i] = 0; Wm[i] = 0xFFFFFFFFu; Wf[i] = 0; }
    unsigned long long bk[LANE_TREES];
    uint32_t sroot[LANE_TREES], bq[LANE_TREES], br[LANE_TREES];
#pragma unroll
    for (int k = 0; k < LANE_TREES; k++) { bk[k] = 0; sroot[k] = 0xFFFFFFFFu; bq[k] = br[k] = 0; }
    uint32_t S = 0;
    bool ovf = false;
    uint32_t (*rd)[16] = s_rd[wave];
    const int band = A.band;
    for (uint32_t t0 = 0; __any(t0 < len); t0 += 4) {
        const uint32_t x0 = s_al + t0;
        uint4 an0 = make_uint4(0, 0, 0, 0), an1 = an0, an2 = an0, an3 = an0;
        if (t0 < len) { an0 = A.anc[x0]; an1 = A.anc[x0 + 1]; an2 = A.anc[x0 + 2]; an3 = A.anc[x0 + 3]; }      // 64 contiguous bytes per lane
        const uint32_t qs[4] = {an0.x, an1.x, an2.x, an3.x}, rs[4] = {an0.y, an1.y, an2.y, an3.y}, ms[4] = {an0.z, an1.z, an2.z, an3.z};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t x = x0 + u, t = t0 + u;
            const bool act = x >= s && x < e && mine;
            const uint32_t qx = qs[u], rx = rs[u], mx = ms[u];
            const uint32_t ux = lane_diag(qx, rx, 0u - (mx & 1u));
            // this lane's latest anchor sits d0 = ((u - j) mod 4, 4 if 0) before x
            const uint32_t d0 = ((((uint32_t)u - j) - 1u) & 3u) + 1u;
            uint32_t best = 0;
#pragma unroll
            for (int i = 0; i < QUAD_N; i++) {
                LaneAnchor y; y.q = Wq[i]; y.u = Wr[i]; y.m = Wm[i]; y.f = Wf[i];
                const uint32_t k = lane_eval_d(qx, ux, mx, y, d0 + 4u * i, band);
                best = k > best ? k : best;
            }
            {   // all four lanes of the quad get the maximum
                uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)best, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                best = o > best ? o : best;
                o = (uint32_t)__builtin_amdgcn_mov_dpp((int)best, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
                best = o > best ? o : best;
            }
            int32_t f = ANCHOR_SCORE2; uint32_t ridx = x - s, dep = 1;
            if (best) {
                f = (int32_t)(best >> 7);
                const uint32_t v = rd[(t - (127u - (best & 127u))) & 31u][quad];
                ridx = v >> 14; dep = (v & 16383u) + 1;
            }
            rd[t & 31u][quad] = (ridx << 14) | dep;          // four lanes, one value
            const bool own = j == (uint32_t)u;               // x is 4-aligned at u = 0, so anchor x belongs to lane u
#pragma unroll
            for (int i = QUAD_N - 1; i >= 1; i--) { Wq[i] = own ? Wq[i - 1] : Wq[i]; Wr[i] = own ? Wr[i - 1] : Wr[i]; Wm[i] = own ? Wm[i - 1] : Wm[i]; Wf[i] = own ? Wf[i - 1] : Wf[i]; }
            Wq[0] = own ? qx : Wq[0]; Wr[0] = own ? ux : Wr[0]; Wm[0] = own ? (act ? mx : 0xFFFFFFFFu) : Wm[0]; Wf[0] = own ? f : Wf[0];   // Wr holds diagonals
            if (act && f >= MIN_SCORE2) {
                const unsigned long long k64 = ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - (x - s)) << 14) | dep;
                bool found = false;
#pragma unroll
                for (int k = 0; k < LANE_TREES; k++) {
                    const bool hit = sroot[k] == ridx;
                    found = found || hit;
                    if (hit && k64 > bk[k]) { bk[k] = k64; bq[k] = qx; br[k] = rx; }
                }
                if (!found) {
                    if (S >= (uint32_t)LANE_TREES) ovf = true;
#pragma unroll
                    for (int k = 0; k < LANE_TREES; k++) if (S == (uint32_t)k) { sroot[k] = ridx; bk[k] = k64; bq[k] = qx; br[k] = rx; }
                    S++;
                }
            }
        }
    }
    if (j == 0 && slot < A.n_rows && real) {
        if (mine && !ovf) {
            uint32_t nc = 0, last = 0;
            for (uint32_t c = 0; c < S; c++) {
                uint32_t pick = 0xFFFFFFFFu; unsigned long long k = 0; uint32_t q1 = 0, rb = 0;
#pragma unroll
                for (int i = 0; i < LANE_TREES; i++)
                    if (sroot[i] != 0xFFFFFFFFu && (c == 0 || sroot[i] > last) && sroot[i] < pick) { pick = sroot[i]; k = bk[i]; q1 = bq[i]; rb = br[i]; }
                last = pick;
                const uint32_t xr = s + pick, ra = A.anc[xr].y, o = s + nc;
                A.c_score[o] = (int32_t)(uint32_t)(k >> 28); A.c_q0[o] = A.anc[xr].x; A.c_q1[o] = q1;
                A.c_r0[o] = ra < rb ? ra : rb; A.c_r1[o] = ra < rb ? rb : ra;
                A.c_n[o] = (uint32_t)(k & 16383u); A.c_rc[o] = A.anc[xr].z >> 1;
                nc++;
            }
            ChunkOut o{};
            o.n_cand = nc; o.left = 0xFFFFFFFFu; o.right = 0;
            A.out[slot] = o;
        } else {
            A.ovf_list[atomicAdd(A.ovf_count, 1u)] = slot;
        }
    }
}

// ---- four lanes per chunk with DEEP windows: the lane DP for bands beyond its register window ---------------------------------------
// c = 30 (metagenome mode) means a band of 83 anchors: no lane holds 83 predecessors, and the wave-per-chunk kernel spends ~150 SIMD
// cycles per anchor on it. Here a quad shares the band: lane j owns the anchors whose index is j mod 4 (as in chain_quad_kernel) and
// keeps its last QD of them - 4 x 21 = 84 - in the lane kernel's form (q + 1, diagonal, contig | strand, score - 1). Per anchor a
// lane scores QD predecessors with the sign-bit step of lane_eval2 (the distance's lane-dependent part, j, is added to the lane's
// best key after its maximum: it is the same for all of a lane's candidates), two quad DPP exchanges give all four lanes the best key.
// The window moves ONCE per four anchors, by plain register renaming: within a step a lane's newest own anchor is a separate entry
// that either joins the candidates (u > j) or not, one select per field (v_cndmask is the slowest VALU instruction: the shifting
// window of chain_quad_kernel would cost 84 of them per anchor). 16 chunks per wave step: ~66 SIMD cycles per anchor.
constexpr int QD = 21;            // own anchors per lane: bands up to 4 * QD = 84
constexpr int QD_NEAR = 2;        // entries per lane that are always scored (with the step's own anchors: the quad's last 8-11); the others only when they could win. Measured on the
                                  // 100 000 x 5 000 step: 5 -> 48.1 ms, 3 -> 44.5, 2 -> 42.4, 1 -> 41.1 (the far pass becomes more frequent as the near part shrinks)
constexpr int QD_RING = 128;      // root / depth ring per quad (power of two > 4 * QD + 3)
__device__ __forceinline__ int32_t quad_eval(uint32_t qx, uint32_t ux, uint32_t mx, uint32_t yq1, uint32_t yu, uint32_t ym, int32_t yf1, int32_t dpj, int32_t bj) {
    // dpj = the distance of the two anchors PLUS j (a compile-time number for the window entries, one select for the extra entry); bj = band + j
    const int32_t a = (int32_t)(qx - yq1);
    const int32_t t = (int32_t)(ux - yu), nt = (int32_t)(yu - ux);
    const int32_t gap = t > nt ? t : nt;
    const int32_t b = a - t;
    const int32_t s1 = yf1 - gap;
    const uint32_t z = ym ^ mx;
    const uint32_t bad = (uint32_t)a | (uint32_t)(BP_CHAIN_BAND - 1 - a) | (uint32_t)b | (uint32_t)(MAX_GAP_LENGTH - gap) | (uint32_t)s1 | z | (0u - z) | (uint32_t)(bj - dpj);
    const uint32_t key = ((uint32_t)s1 << 7) + (((((uint32_t)ANCHOR_SCORE2 + 1u) << 7) | 127u) - (uint32_t)dpj);      // + j after the lane's maximum
    return (int32_t)(key | (bad & 0x80000000u));
}
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_quad_deep_kernel(ChainArgs A) {      // (255 registers, two waves per SIMD: capped at 168 it spills 83 dwords per lane)
    __shared__ uint32_t s_rd[LANE_WAVES][QD_RING][16];     // root index << 14 | depth of the last QD_RING anchors, per quad
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, quad = lane >> 2;
    const int32_t j = lane & 3;
    const uint32_t slot_i = (blockIdx.x * LANE_WAVES + wave) * 16 + quad;
    const uint32_t slot = A.row_order && slot_i < A.n_rows ? A.row_order[slot_i] : slot_i;
    uint32_t s = 0, e = 0;
    bool mine = false, real = false;
    const uint32_t pair = A.row_pair[slot < A.n_rows ? slot : A.n_rows - 1];
    if (slot < A.n_rows && slot - A.cbase[pair] < A.n_chunks[pair]) {
        const uint2 se = A.chunks[slot];
        s = se.x; e = se.y;
        real = true;
        mine = e > s && e - s < 16384;
    }
    const uint32_t s_al = s & ~3u;
    const uint32_t len = mine ? e - s_al : 0;
    uint32_t Wq[QD], Wu[QD], Wm[QD]; int32_t Wf[QD];      // entry i = this lane's anchor 4 (i + 1) - (u - j) ... before x: its (i + 1)-th latest of EARLIER steps
#pragma unroll
    for (int i = 0; i < QD; i++) { Wq[i] = 0; Wu[i] = 0; Wm[i] = 0xFFFFFFFFu; Wf[i] = -1; }      // (score - 1 of an EMPTY entry: below every real one, see the far bound)
    // the chunk's qualifying chain trees (at most LANE_TREES = 4, as in the lane kernel): the quad's four lanes see the same anchor, root and key, so each keeps
    // ONE slot - lane j the j-th tree to qualify - instead of all four keeping all four (a compare and three selects per anchor and lane instead of four times that);
    // a quad-wide OR tells whether the root already has a slot, lane 0 collects the four at the end
    static_assert(LANE_TREES == 4, "one tree slot per lane of the quad");
    unsigned long long my_bk = 0;
    uint32_t my_root = 0xFFFFFFFFu, my_bq = 0, my_br = 0;
    uint32_t S = 0;
    bool ovf = false;
    uint32_t (*rd)[16] = s_rd[wave];
    const int32_t bj = A.band + j;
    const bool prune = A.dp_prune != 0;
    uint32_t far_diag = 0;
    for (uint32_t t0 = 0; __any(t0 < len); t0 += 4) {
        const uint32_t x0 = s_al + t0;
        uint4 an0 = make_uint4(0, 0, 0, 0), an1 = an0, an2 = an0, an3 = an0;
        if (t0 < len) { an0 = A.anc[x0]; an1 = A.anc[x0 + 1]; an2 = A.anc[x0 + 2]; an3 = A.anc[x0 + 3]; }      // 64 contiguous bytes per lane
        const uint32_t qs[4] = {an0.x, an1.x, an2.x, an3.x}, rs[4] = {an0.y, an1.y, an2.y, an3.y}, ms[4] = {an0.z, an1.z, an2.z, an3.z};
        uint32_t nq = 0, nu = 0, nm = 0xFFFFFFFFu; int32_t nf = -1;      // this lane's own anchor of the step (from u = j on)
        // The FAR part of the window - a lane's entries QD_NEAR .. QD - 1: the quad's anchors more than 4 QD_NEAR + 3 back - can only win with a score above the
        // best near one: a predecessor y scores f[y] + ANCHOR_SCORE2 - gap <= f[y] + ANCHOR_SCORE2, and on equal scores the NEARER one is taken. far_top =
        // the largest f - 1 among the far entries of the quad (-1: all empty), one pass per step (the window does not move within a step). Along a chain f
        // grows by ~ANCHOR_SCORE2 per anchor, so the nearest predecessors nearly always beat that bound and the far three quarters of the band are not scored
        // at all; the decision is taken per WAVE (an anchor off its chunk's chain - no near predecessor - has all sixteen quads score everything: same
        // results, nothing skipped). $PSK_DP_PRUNE=0: every entry always (tests, A/B).
        // ... and only for an anchor whose diagonal is within MAX_GAP_LENGTH of a far entry's. far_diag: one bit per 1024 diagonals (mod 32): an entry on diagonal
        // d sets the two bits that cover d - MAX_GAP_LENGTH .. d + MAX_GAP_LENGTH; an anchor whose own bit is clear has no predecessor in the far part. That
        // is the chance match off the chunk's chain (k-mers are seeds by content: ~1 % of a query's seeds also sit somewhere else in a 5 Mb reference): no
        // near predecessor either, but no reason to score 60 entries that cannot hold one. The bits of the entry that turns far are added every step and
        // the set is rebuilt every 16 steps (bits of entries that left linger until then: a few more anchors pass the test, none fewer).
        int32_t far_top = -1;
        if (prune) {
            if ((t0 & 63u) == 0) {
                far_diag = 0;
#pragma unroll
                for (int i = QD_NEAR; i < QD; i++) far_diag |= __builtin_amdgcn_alignbit(3u, 3u, 32u - (((Wu[i] - (uint32_t)MAX_GAP_LENGTH) >> 10) & 31u));
            } else far_diag |= __builtin_amdgcn_alignbit(3u, 3u, 32u - (((Wu[QD_NEAR] - (uint32_t)MAX_GAP_LENGTH) >> 10) & 31u));
#pragma unroll
            for (int i = QD_NEAR; i < QD; i++) far_top = Wf[i] > far_top ? Wf[i] : far_top;
            int32_t o = __builtin_amdgcn_mov_dpp(far_top, 0xB1, 0xF, 0xF, true); far_top = o > far_top ? o : far_top;
            o = __builtin_amdgcn_mov_dpp(far_top, 0x4E, 0xF, 0xF, true); far_top = o > far_top ? o : far_top;
        }
        uint32_t fd = far_diag;      // the quad's
        fd |= (uint32_t)__builtin_amdgcn_mov_dpp((int)fd, 0xB1, 0xF, 0xF, true); fd |= (uint32_t)__builtin_amdgcn_mov_dpp((int)fd, 0x4E, 0xF, 0xF, true);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t x = x0 + u, t = t0 + u;
            const bool act = x >= s && x < e && mine;
            const uint32_t qx = qs[u], rx = rs[u], mx = ms[u];
            const uint32_t ux = lane_diag(qx, rx, 0u - (mx & 1u));
            int32_t best = 0;
#pragma unroll
            for (int i = 0; i < QD_NEAR; i++) {      // own anchors of earlier steps: distance u - j + 4 (i + 1)
                const int32_t k = quad_eval(qx, ux, mx, Wq[i], Wu[i], Wm[i], Wf[i], u + 4 * (i + 1), bj);
                best = k > best ? k : best;
            }
            {   // the one candidate that depends on the lane: its own anchor of THIS step (u > j, distance u - j) or its oldest (distance u - j + 4 QD)
                const bool late = u > j;
                const int32_t k = quad_eval(qx, ux, mx, late ? nq : Wq[QD - 1], late ? nu : Wu[QD - 1], late ? nm : Wm[QD - 1], late ? nf : Wf[QD - 1], late ? u : u + 4 * QD, bj);
                best = k > best ? k : best;
            }
            bool far = !prune;
            if (prune) {      // does any quad of the wave still need its far entries? (best, before the lane's + j: score << 7 | low bits)
                int32_t nb = best;
                int32_t o = __builtin_amdgcn_mov_dpp(nb, 0xB1, 0xF, 0xF, true); nb = o > nb ? o : nb;
                o = __builtin_amdgcn_mov_dpp(nb, 0x4E, 0xF, 0xF, true); nb = o > nb ? o : nb;
                far = __any(act && far_top >= 0 && (nb >> 7) < far_top + 1 + ANCHOR_SCORE2 && ((fd >> ((ux >> 10) & 31u)) & 1u) && qx + 1u - Wq[QD_NEAR] <= (uint32_t)BP_CHAIN_BAND);      // (a quad past its chunk's end has no say; a lane's far entries lie at or behind its nearest one)
            }
            if (far) {
#pragma unroll
                for (int i = QD_NEAR; i < QD - 1; i++) {
                    const int32_t k = quad_eval(qx, ux, mx, Wq[i], Wu[i], Wm[i], Wf[i], u + 4 * (i + 1), bj);
                    best = k > best ? k : best;
                }
            }
            if (best > 0) best += j;      // the lane-dependent part of 127 - distance
            {   // all four lanes of the quad get the maximum
                int32_t o = __builtin_amdgcn_mov_dpp(best, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                best = o > best ? o : best;
                o = __builtin_amdgcn_mov_dpp(best, 0x4E, 0xF, 0xF, true);            // quad_perm [2,3,0,1]
                best = o > best ? o : best;
            }
            int32_t f = ANCHOR_SCORE2; uint32_t ridx = x - s, dep = 1;
            if (best > 0) {
                f = best >> 7;
                const uint32_t v = rd[(t - (127u - ((uint32_t)best & 127u))) & (uint32_t)(QD_RING - 1)][quad];
                ridx = v >> 14; dep = (v & 16383u) + 1;
            }
            rd[t & (uint32_t)(QD_RING - 1)][quad] = (ridx << 14) | dep;          // four lanes, one value
            if (j == u) { nq = qx + 1u; nu = ux; nm = act ? mx : 0xFFFFFFFFu; nf = act ? f - 1 : -1; }      // x is 4-aligned at u = 0: anchor x belongs to lane u
            if (act && f >= MIN_SCORE2) {
                const unsigned long long k64 = ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - (x - s)) << 14) | dep;
                const bool hit = my_root == ridx;
                if (hit && k64 > my_bk) { my_bk = k64; my_bq = qx; my_br = rx; }
                int fnd = hit ? 1 : 0;      // over the quad
                fnd |= __builtin_amdgcn_mov_dpp(fnd, 0xB1, 0xF, 0xF, true); fnd |= __builtin_amdgcn_mov_dpp(fnd, 0x4E, 0xF, 0xF, true);
                if (!fnd) {
                    if (S >= (uint32_t)LANE_TREES) ovf = true;
                    if (S == (uint32_t)j) { my_root = ridx; my_bk = k64; my_bq = qx; my_br = rx; }
                    S++;
                }
            }
        }
        // the window moves by one own anchor per step
#pragma unroll
        for (int i = QD - 1; i >= 1; i--) { Wq[i] = Wq[i - 1]; Wu[i] = Wu[i - 1]; Wm[i] = Wm[i - 1]; Wf[i] = Wf[i - 1]; }
        Wq[0] = nq; Wu[0] = nu; Wm[0] = nm; Wf[0] = nf;
    }
    unsigned long long bk[LANE_TREES];
    uint32_t sroot[LANE_TREES], bq[LANE_TREES], br[LANE_TREES];
#pragma unroll
    for (int k = 0; k < LANE_TREES; k++) {      // lane k of the quad holds slot k
        const int src = (lane & ~3) + k;
        sroot[k] = (uint32_t)__shfl((int)my_root, src); bq[k] = (uint32_t)__shfl((int)my_bq, src); br[k] = (uint32_t)__shfl((int)my_br, src);
        bk[k] = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(my_bk >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)my_bk, src);
    }
    if (j == 0 && slot < A.n_rows && real) {
        if (mine && !ovf) {
            uint32_t nc = 0, last = 0;
            for (uint32_t c = 0; c < S; c++) {
                uint32_t pick = 0xFFFFFFFFu; unsigned long long k = 0; uint32_t q1 = 0, rb = 0;
#pragma unroll
                for (int i = 0; i < LANE_TREES; i++)
                    if (sroot[i] != 0xFFFFFFFFu && (c == 0 || sroot[i] > last) && sroot[i] < pick) { pick = sroot[i]; k = bk[i]; q1 = bq[i]; rb = br[i]; }
                last = pick;
                const uint32_t xr = s + pick, ra = A.anc[xr].y, o = s + nc;
                A.c_score[o] = (int32_t)(uint32_t)(k >> 28); A.c_q0[o] = A.anc[xr].x; A.c_q1[o] = q1;
                A.c_r0[o] = ra < rb ? ra : rb; A.c_r1[o] = ra < rb ? rb : ra;
                A.c_n[o] = (uint32_t)(k & 16383u); A.c_rc[o] = A.anc[xr].z >> 1;
                nc++;
            }
            ChunkOut o{};
            o.n_cand = nc; o.left = 0xFFFFFFFFu; o.right = 0;
            A.out[slot] = o;
        } else {
            A.ovf_list[atomicAdd(A.ovf_count, 1u)] = slot;
        }
    }
}

// maximum over the wave, uniform result: four DPP steps leave every row of 16 lanes with its maximum, four lane reads finish it
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); v = o > v ? o : v;      // quad_perm [1,0,3,2]
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); v = o > v ? o : v;               // quad_perm [2,3,0,1]
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, true); v = o > v ? o : v;              // row_ror:4
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true); v = o > v ? o : v;              // row_ror:8
    const uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// The wave-per-chunk chaining of ONE row of the chunk table (one wavefront): DP over an LDS ring, per-tree bests,
// candidate emission. Shared arrays are the calling wave's slices.
// (the kernel's throughput follows the waves a CU holds, and those follow this struct: the candidate staging shares the ring's
// space - the ring is dead once the DP is through - and the roots' indices ride in the unused top bits of the per-tree best keys)
struct ChainWaveLds {
    union {
        uint32_t ring[6][RING];        // qp, rp, rm, f, root id, depth
        uint32_t cand[7][64];          // score, q0, q1, r0, r1, nanch, ref contig (after the DP)
    };
    unsigned long long best[RMAX];     // root's local index << 49 | f<<28 | (16383-local idx)<<14 | depth
};
static_assert(sizeof(uint32_t) * 7 * 64 <= sizeof(uint32_t) * 6 * RING, "candidate staging fits the ring");

__device__ void chain_row_candidates(const ChainArgs& A, uint32_t s, uint32_t e, ChunkOut* op, ChainWaveLds& L, int lane, uint32_t R, bool fast);

__device__ void chain_chunk_row(const ChainArgs& A, uint32_t slot, ChainWaveLds& L, int lane) {
    const uint2 se = A.chunks[slot];
    const uint32_t s = se.x, e = se.y, n = e - s;
    ChunkOut* op = &A.out[slot];
    uint32_t (*ring)[RING] = L.ring;
    unsigned long long* s_best_w = L.best;
    bool fast = !A.force_serial && n < 16384;
    uint32_t R = 0;
    if (fast) {
        for (uint32_t base = s; base < e && fast; base += 64) {
            const uint32_t idx = base + lane;
            const bool have = idx < e;
            const uint4 my_a = have ? A.anc[idx] : make_uint4(0, 0, 0, 0);
            const uint32_t my_qp = my_a.x, my_rp = my_a.y, my_rm = my_a.z;
            const uint32_t cnt = e - base < 64 ? e - base : 64;
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t x = base + j;
                const uint32_t qx = __builtin_amdgcn_readlane(my_qp, j), rx = __builtin_amdgcn_readlane(my_rp, j),
                               mx = __builtin_amdgcn_readlane(my_rm, j);
                const uint32_t avail = x - s;   // anchors before x in the chunk
                uint32_t key = 0;
                // the band may need two sweeps of 64 predecessors; the second only if the 65th is still in bp range
                int sweeps = 1;
                if (avail > 64 && A.band > 64 && qx - ring[0][(x - 65) & (RING - 1)] <= (uint32_t)BP_CHAIN_BAND) sweeps = 2;
                for (int sw = 0; sw < sweeps; sw++) {
                    const uint32_t dist = lane + 1 + 64 * sw;
                    if (dist <= avail && dist <= (uint32_t)A.band) {
                        const uint32_t sl = (x - dist) & (RING - 1);
                        const uint32_t qy = ring[0][sl], ry = ring[1][sl], my = ring[2][sl];
                        const int32_t fy = (int32_t)ring[3][sl];
                        const int32_t dq = (int32_t)(qx - qy);
                        const int32_t dr = (mx & 1) ? (int32_t)(ry - rx) : (int32_t)(rx - ry);
                        const int32_t gap = dq > dr ? dq - dr : dr - dq;
                        const int32_t sc = fy + ANCHOR_SCORE2 - gap;
                        if (my == mx && dq > 0 && dq <= BP_CHAIN_BAND && dr > 0 && gap <= MAX_GAP_LENGTH && sc > ANCHOR_SCORE2) {
                            uint32_t k2 = ((uint32_t)sc << 7) | (127u - dist);   // max score, then nearest predecessor
                            key = k2 > key ? k2 : key;
                        }
                    }
                }
                uint32_t best = wave_max_u32(key);
                int32_t f = ANCHOR_SCORE2; uint32_t rid, dep;
                if (best) {
                    f = (int32_t)(best >> 7);
                    const uint32_t sl = (x - (127u - (best & 127u))) & (RING - 1);
                    rid = ring[4][sl]; dep = ring[5][sl] + 1;
                } else {
                    rid = R++; dep = 1;
                    if (rid >= RMAX) { fast = false; break; }
                    if (lane == 0) s_best_w[rid] = (unsigned long long)avail << 49;      // the root's index; any real key of the tree compares above it
                }
                if (lane == 0) {
                    const uint32_t sl = x & (RING - 1);
                    ring[0][sl] = qx; ring[1][sl] = rx; ring[2][sl] = mx; ring[3][sl] = (uint32_t)f; ring[4][sl] = rid; ring[5][sl] = dep;
                    const unsigned long long old = s_best_w[rid];
                    const unsigned long long k64 = (old & ~((1ull << 49) - 1)) | ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - avail) << 14) | dep;
                    if (k64 > old) s_best_w[rid] = k64;
                }
                lds_wave_sync();
            }
        }
    }
    chain_row_candidates(A, s, e, op, L, lane, R, fast);
}

// the chunk's candidate chains out of the per-tree bests in L.best[0 .. R) (fast), or the lane-serial path over global scratch (!fast)
__device__ void chain_row_candidates(const ChainArgs& A, uint32_t s, uint32_t e, ChunkOut* op, ChainWaveLds& L, int lane, uint32_t R, bool fast) {
    unsigned long long* s_best_w = L.best; uint32_t (*s_cand_w)[64] = L.cand;
    uint32_t C = 0;
    if (fast) {
        // candidates: one per chain tree whose best anchor passes the thresholds, in root order
        for (uint32_t r0 = 0; r0 < R && fast; r0 += 64) {
            const uint32_t r = r0 + lane;
            bool qual = false; uint32_t f = 0, lx = 0, dep = 0, rootx = 0;
            if (r < R) {
                unsigned long long bk = s_best_w[r];
                rootx = (uint32_t)(bk >> 49); bk &= (1ull << 49) - 1;
                f = (uint32_t)(bk >> 28); lx = 16383u - (uint32_t)((bk >> 14) & 16383u); dep = (uint32_t)(bk & 16383u);
                qual = dep >= MIN_ANCHORS && (int32_t)f >= MIN_SCORE2;
            }
            unsigned long long bal = __ballot(qual);
            uint32_t ci = C + __popcll(bal & ((1ull << lane) - 1));
            C += __popcll(bal);
            if (C > 64) { fast = false; break; }
            if (qual) {
                uint32_t xr = s + rootx, xb = s + lx;
                uint32_t ra = A.anc[xr].y, rb = A.anc[xb].y;
                s_cand_w[0][ci] = f; s_cand_w[1][ci] = A.anc[xr].x; s_cand_w[2][ci] = A.anc[xb].x;
                s_cand_w[3][ci] = ra < rb ? ra : rb; s_cand_w[4][ci] = ra < rb ? rb : ra; s_cand_w[5][ci] = dep;
                s_cand_w[6][ci] = A.anc[xr].z >> 1;
            }
        }
    }
    if (!fast) {   // lane-serial path writes its candidates straight to the global arrays
        if (lane == 0) {
            C = chain_chunk_serial(A, s, e);
            atomicAdd(&A.stats[1], 1u);
        }
    } else {
        lds_wave_sync();
        if ((uint32_t)lane < C) {
            A.c_score[s + lane] = (int32_t)s_cand_w[0][lane]; A.c_q0[s + lane] = s_cand_w[1][lane]; A.c_q1[s + lane] = s_cand_w[2][lane];
            A.c_r0[s + lane] = s_cand_w[3][lane]; A.c_r1[s + lane] = s_cand_w[4][lane]; A.c_n[s + lane] = s_cand_w[5][lane];
            A.c_rc[s + lane] = s_cand_w[6][lane];
        }
    }
    if (lane == 0) {
        ChunkOut o{};
        o.n_cand = C; o.left = 0xFFFFFFFFu; o.right = 0;
        *op = o;
    }
}

// every row of the chunk table (used when the lane kernel does not run: band > LANE_N, PSK_CHAIN_LANE=0, serial cross-check)
__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_chunk_kernel(ChainArgs A) {
    __shared__ ChainWaveLds s_lds[CHAIN_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * CHAIN_WAVES + wave;   // row of the chunk table
    const uint32_t pair = A.row_pair[slot < A.n_rows ? slot : A.n_rows - 1];
    if (slot >= A.n_rows) return;
    if (slot - A.cbase[pair] >= A.n_chunks[pair]) return;
    chain_chunk_row(A, slot, s_lds[wave], lane);
}

// only the rows the lane kernel listed (fixed grid, waves loop over the list: its length is known on the device only)
__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_chunk_list_kernel(ChainArgs A) {
    __shared__ ChainWaveLds s_lds[CHAIN_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_list = *A.ovf_count, n_waves = gridDim.x * CHAIN_WAVES;
    for (uint32_t k = blockIdx.x * CHAIN_WAVES + wave; k < n_list; k += n_waves) {
        chain_chunk_row(A, A.ovf_list[k], s_lds[wave], lane);
        lds_wave_sync();
    }
}

// Rows of the chunk table by chunk length. A wave of the lane / quad DP kernels runs until the LONGEST of its chunks is through, and rows without a chunk
// (a pair has as many rows as its query could have chunks) sit between the others: in table order a metagenome batch spends 2.5 x the lane-instructions
// its anchors need (profiles/r3/r3q_pmc_sq_meta.txt: 3 530 per anchor at 84 predecessors x 17). key = length (0: no chunk), sorted descending with the row
// number as the value: equal lengths share waves, the long chunks start first, the empty rows end up in waves that exit at once.
__global__ __launch_bounds__(256) void row_len_kernel(const uint2* __restrict__ chunks, const uint32_t* __restrict__ n_chunks, const uint32_t* __restrict__ cbase,
                                                      const uint32_t* __restrict__ row_pair, uint32_t n_rows, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t p = row_pair[r];
    uint32_t len = 0;
    if (r - cbase[p] < n_chunks[p]) { const uint2 se = chunks[r]; len = se.y - se.x; len = (len + 7u) >> 3; len = len < 255u ? len : 255u; }
    key[r] = len; val[r] = r;      // eight bits: ONE pass of the radix sort - lanes of a wave want chunks of similar length, not of equal length (classes of eight anchors; 2 040 and more share the last)
}

// ---- wave-per-chunk DP with the look-back window in REGISTERS (launches of few rows) ----------------------------
// A launch of a few hundred rows (one Database.query of a contig: one or two chunks per shortlisted reference) is as slow as its
// longest chunk, and per anchor the kernels above are a chain of LDS round trips (ring read -> score -> wave maximum -> tree id read ->
// ring write -> wait: ~1 400 cycles) or, four lanes per chunk, 21 predecessors one after the other (~2 000 cycles): 220-300 us for the
// 330 anchors of a 10 kb contig at c = 30. Here the window of 64 * S anchors is spread over the wave's registers - the anchor with
// chunk-local index a lives in lane a & 63, register set (a >> 6) % S - every lane scores the S predecessors it holds, one wave maximum
// picks the winner, two lane reads fetch its tree and depth, and the per-tree bests sit in registers too (tree t: lane t & 63, register
// t >> 6). No LDS and no wait inside the loop. Same keys, same tie-break (nearest predecessor), same tree numbering as chain_chunk_row.
// six registers take wave-uniform values in ONE lane (v_writelane_b32; this compiler has no builtin for it; on gfx9 the lane select sits in M0 when
// the value is a scalar register too: one constant-bus operand per instruction)
__device__ __forceinline__ void write_lane6(uint32_t lane, uint32_t& v0, uint32_t a0, uint32_t& v1, uint32_t a1, uint32_t& v2, uint32_t a2, uint32_t& v3, uint32_t a3,
                                            uint32_t& v4, uint32_t a4, uint32_t& v5, uint32_t a5) {
    uint32_t keep;      // (M0 is the compiler's own: handed back as it was)
    asm volatile("s_mov_b32 %6, m0\n\ts_mov_b32 m0, %7\n\ts_nop 0\n\tv_writelane_b32 %0, %8, m0\n\tv_writelane_b32 %1, %9, m0\n\tv_writelane_b32 %2, %10, m0\n\t"
                 "v_writelane_b32 %3, %11, m0\n\tv_writelane_b32 %4, %12, m0\n\tv_writelane_b32 %5, %13, m0\n\ts_mov_b32 m0, %6"
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "=&s"(keep)
                 : "s"(lane), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5));
}
struct RegWin { uint32_t q1, u, m; int32_t f1; uint32_t id, dp; };      // one window slot per lane: q + 1, diagonal, ref contig | strand, score - 1 (lane_eval2's form), tree, depth
// maximum over the wave as a scalar: the four row steps of wave_max_u32, then the rows are folded into the last one (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3) and lane 63 is read - 6 DPP steps + 1 lane read where four lane reads and their scalar maxima cost 13 instructions
__device__ __forceinline__ uint32_t wave_max_scalar(uint32_t v) {
    uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); v = o > v ? o : v;      // quad_perm [1,0,3,2]
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); v = o > v ? o : v;               // quad_perm [2,3,0,1]
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x124, 0xF, 0xF, true); v = o > v ? o : v;              // row_ror:4
    o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true); v = o > v ? o : v;              // row_ror:8
    // (written out: from the builtin the compiler makes a copy, a v_mov_dpp and a v_max for each of the two steps; the no-ops are the DPP read-after-write wait states)
    asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                 "v_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// one block of up to 64 anchors (chunk-local indices base - s ...): its anchors take register set T (compile time: no branch per anchor)
template <int S, int T>
__device__ __forceinline__ void chain_reg_block(const ChainArgs& A, ChainWaveLds& L, uint32_t* s_root, const int lane, const uint32_t s, const uint32_t e, const uint32_t base,
                                                const uint32_t band, RegWin& w0, RegWin& w1, uint32_t& R, bool& over) {
    constexpr uint32_t WMASK = 64u * S - 1u;
    RegWin& wt = T ? w1 : w0;
    const uint32_t idx = base + lane;
    const uint4 my_a = idx < e ? A.anc[idx] : make_uint4(0, 0, 0, 0);
    const uint32_t cnt = e - base < 64 ? e - base : 64;
    for (uint32_t j = 0; j < cnt; j++) {
        const uint32_t qx = __builtin_amdgcn_readlane(my_a.x, j), rx = __builtin_amdgcn_readlane(my_a.y, j), mx = __builtin_amdgcn_readlane(my_a.z, j);
        const uint32_t avail = base + j - s;   // anchors before this one in the chunk = its chunk-local index
        const uint32_t ux = lane_diag(qx, rx, 0u - (mx & 1u));
        // every lane scores the predecessor(s) it holds: lane_eval2's key, negative when not chainable or outside the band
        const uint32_t d0 = (avail - (uint32_t)lane) & WMASK;      // 0: the slot this anchor is about to take
        int32_t key = lane_eval2(qx, ux, mx, LanePred{w0.q1, w0.u, w0.m, w0.f1}, (int)d0) | (int32_t)(((band - d0) | (d0 - 1u)) & 0x80000000u);
        key = key > 0 ? key : 0;
        if (S > 1) {
            const uint32_t d1 = (d0 - 64u) & WMASK;
            const int32_t k1 = lane_eval2(qx, ux, mx, LanePred{w1.q1, w1.u, w1.m, w1.f1}, (int)d1) | (int32_t)(((band - d1) | (d1 - 1u)) & 0x80000000u);
            key = k1 > key ? k1 : key;
        }
        const uint32_t best = wave_max_scalar((uint32_t)key);
        int32_t f = ANCHOR_SCORE2; uint32_t rid, dep;
        if (best) {
            f = (int32_t)(best >> 7);
            const uint32_t ps = (avail - (127u - (best & 127u))) & WMASK;
            // (a lane read per register set and a scalar choice: picking the register set first turns into an indexed array in scratch)
            rid = __builtin_amdgcn_readlane(w0.id, ps & 63u); dep = __builtin_amdgcn_readlane(w0.dp, ps & 63u);
            if (S > 1) {
                const uint32_t rid1 = __builtin_amdgcn_readlane(w1.id, ps & 63u), dep1 = __builtin_amdgcn_readlane(w1.dp, ps & 63u);
                if (ps >> 6) { rid = rid1; dep = dep1; }
            }
            dep++;
        } else {
            rid = R++; dep = 1;
            if (rid >= RMAX) { over = true; rid = 0; }      // more trees than the LDS tables hold: the block runs to its end (results discarded), the lane-serial path takes the chunk
            else if (lane == 0) s_root[rid] = avail;
        }
        // the anchor takes its slot: everything about it is wave-uniform, six lane writes
        uint32_t f1 = (uint32_t)wt.f1;
        write_lane6(j, wt.q1, qx + 1u, wt.u, ux, wt.m, mx, f1, (uint32_t)(f - 1), wt.id, rid, wt.dp, dep);      // (lane = chunk-local index & 63 = j: blocks start at multiples of 64)
        wt.f1 = (int32_t)f1;
    }
    // the block's anchors now sit one per lane in register set T: their keys go to their trees' bests together
    // (the maximum over a tree's anchors of f << 28 | (16383 - index) << 14 | depth, as chain_chunk_row keeps it anchor by anchor)
    if (!over && (uint32_t)lane < cnt)
        atomicMax(&L.best[wt.id], ((unsigned long long)(uint32_t)(wt.f1 + 1) << 28) | ((unsigned long long)(16383u - (base - s + (uint32_t)lane)) << 14) | wt.dp);
}

template <int S>
__device__ void chain_chunk_row_reg(const ChainArgs& A, uint32_t slot, ChainWaveLds& L, int lane) {
    static_assert(S == 1 || S == 2, "one or two window slots per lane");
    const uint2 se = A.chunks[slot];
    const uint32_t s = se.x, e = se.y, n = e - s;
    ChunkOut* op = &A.out[slot];
    bool fast = !A.force_serial && n < 16384;
    uint32_t R = 0;
    RegWin w0{0, 0, 0xFFFFFFFFu, 0, 0, 0}, w1 = w0;      // (m = all ones: a slot nothing was written to matches no anchor)
    uint32_t* s_root = &L.ring[0][0];                    // chunk-local index of every tree's root (the ring itself is not used here)
    static_assert(RMAX <= 6 * RING, "root table fits the ring's space");
    const uint32_t band = (uint32_t)A.band;
    if (fast) {
#pragma unroll
        for (int u = 0; u < RMAX / 64; u++) L.best[lane + 64 * u] = 0;
        lds_wave_sync();
        bool over = false;
        for (uint32_t base = s; base < e && !over; base += 64u * S) {      // anchor a lives in lane a & 63 of register set (a >> 6) % S: blocks alternate between the sets
            chain_reg_block<S, 0>(A, L, s_root, lane, s, e, base, band, w0, w1, R, over);
            if (S > 1 && base + 64u < e && !over) chain_reg_block<S, 1>(A, L, s_root, lane, s, e, base + 64u, band, w0, w1, R, over);
        }
        fast = !over;
    }
    if (fast) {
        lds_wave_sync();
#pragma unroll
        for (int u = 0; u < RMAX / 64; u++) if ((uint32_t)lane + 64u * u < R) L.best[lane + 64 * u] |= (unsigned long long)s_root[lane + 64 * u] << 49;      // the root's index rides in the top bits
        lds_wave_sync();
    }
    chain_row_candidates(A, s, e, op, L, lane, R, fast);
}

template <int S>
__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_wave_reg_kernel(ChainArgs A) {
    __shared__ ChainWaveLds s_lds[CHAIN_WAVES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;      // (told to be uniform: the row's bounds, the loop counters and the tree count live in scalar registers)
    const uint32_t slot = blockIdx.x * CHAIN_WAVES + wave;   // row of the chunk table
    const uint32_t pair = A.row_pair[slot < A.n_rows ? slot : A.n_rows - 1];
    if (slot >= A.n_rows) return;
    if (slot - A.cbase[pair] >= A.n_chunks[pair]) return;
    chain_chunk_row_reg<S>(A, slot, s_lds[wave], lane);
}

// ------------------------------------------------------------------ chain selection (per pair)
// Greedy over ALL candidate chains of a pair by (score desc, generation order): a chain is kept unless it
// overlaps a kept chain on the query (same chunk) or on the reference (same ref contig). One wave per pair:
// candidates staged in LDS, bitonic sort of (score, ~order) keys, kept list scanned 64 entries at a time.
constexpr int CMAX = 1024;

struct SelArgs {
    const uint2* chunks; const uint32_t* n_chunks; const uint32_t* cbase; uint32_t n_pairs;
    Strided<int32_t> c_score; Strided<uint32_t> c_q0, c_q1, c_r0, c_r1, c_n, c_rc, c_state;
    ChunkOut* out; uint32_t two_c; int force_serial; uint32_t* stats;
    const uint32_t* live; const uint32_t* n_live;      // pairs that have a chunk table (every other pair has no candidate chain)
    uint32_t* rest_list; uint32_t* rest_count;         // select_tiny_kernel: the live pairs it did NOT take (what the wave kernel still has to visit)
    uint32_t* big_list; uint32_t* big_count;           // pairs with more than CMAX candidates, for select_big_kernel
    int tiny_done;                                     // pairs of at most TINY_ROWS chunks and TINY_CANDS candidates were selected by select_tiny_kernel
};
constexpr uint32_t TINY_ROWS = 4, TINY_CANDS = 8;

__device__ __forceinline__ void sel_commit(const SelArgs& S, uint32_t row, uint32_t q0, uint32_t q1, uint32_t n) {
    ChunkOut* o = &S.out[row];
    atomicAdd(&o->anchors, n); atomicAdd(&o->n_intervals, 1u);
    atomicMin(&o->left, q0); atomicMax(&o->right, q1);
    atomicAdd((unsigned long long*)&o->cov_q, (unsigned long long)(q1 - q0) + 1 + S.two_c);
}

// lane-serial O(C^2) selection on global memory: for pairs with more than CMAX candidates
__device__ void select_serial(const SelArgs& S, uint32_t row0, uint32_t nrows) {
    for (uint32_t r = 0; r < nrows; r++) { uint32_t s = S.chunks[row0 + r].x, nc = S.out[row0 + r].n_cand; for (uint32_t i = 0; i < nc; i++) S.c_state[s + i] = 0; }
    for (;;) {
        int32_t best = -1; uint32_t brow = 0, bslot = 0;
        for (uint32_t r = 0; r < nrows; r++) {            // generation order: rows, then slots; strict > keeps the earliest
            uint32_t s = S.chunks[row0 + r].x, nc = S.out[row0 + r].n_cand;
            for (uint32_t i = 0; i < nc; i++) if (S.c_state[s + i] == 0 && S.c_score[s + i] > best) { best = S.c_score[s + i]; brow = r; bslot = s + i; }
        }
        if (best < 0) break;
        bool ok = true;
        for (uint32_t r = 0; r < nrows && ok; r++) {
            uint32_t s = S.chunks[row0 + r].x, nc = S.out[row0 + r].n_cand;
            for (uint32_t i = 0; i < nc && ok; i++) if (S.c_state[s + i] == 1) {
                uint32_t j = s + i;
                if (r == brow && !(S.c_q1[bslot] < S.c_q0[j] || S.c_q0[bslot] > S.c_q1[j])) ok = false;
                else if (S.c_rc[bslot] == S.c_rc[j] && !(S.c_r1[bslot] < S.c_r0[j] || S.c_r0[bslot] > S.c_r1[j])) ok = false;
            }
        }
        S.c_state[bslot] = ok ? 1 : 2;
        if (ok) sel_commit(S, row0 + brow, S.c_q0[bslot], S.c_q1[bslot], S.c_n[bslot]);
    }
}

// Two instantiations: CSMALL candidates (25 KB of LDS: six waves per CU) for the bulk - a 5 Mb pair has ~300 candidate chains -,
// which passes the pairs that do not fit to the CMAX one (51 KB: three waves per CU), which passes on to select_big_kernel. With
// ~13 000 pairs per launch the kernel's time is residency (12.8 -> 9.4 ms per 10^5 pairs); with 3 000 it was each wave's own
// chain of LDS round trips and the smaller instantiation gained nothing.
constexpr int CSMALL = 512;
template <int CM>
__device__ void select_pair(const SelArgs& S, const uint32_t p, uint32_t* __restrict__ over_list, uint32_t* __restrict__ over_count, const bool first_tier) {
    __shared__ uint32_t l_q0[CM], l_q1[CM], l_r0[CM], l_r1[CM], l_rc[CM], l_row[CM], l_n[CM];
    __shared__ unsigned long long l_key[CM];     // (ref contig, r0) keys of the reference-order sort, then the priority keys of the conflicted candidates
    __shared__ uint32_t l_sc[CM];                // candidate scores; afterwards (first half) the kept list of the greedy
    __shared__ uint16_t l_ord[CM];               // conflicted candidates by priority rank
    __shared__ uint16_t l_idx[CM];               // payload of the reference-order sort, then the conflicted list
    uint16_t* const l_kept = (uint16_t*)l_sc;    // (the scores are dead once the conflicted candidates are in priority order)
    __shared__ uint8_t l_conf[CM];
    const int lane = threadIdx.x;
    const uint32_t row0 = S.cbase[p], nrows = S.n_chunks[p];
    // candidates in generation order (rows, then slots)
    uint32_t C = 0;
    for (uint32_t r0 = 0; r0 < nrows; r0 += 64) {
        uint32_t r = r0 + lane;
        uint32_t cnt = r < nrows ? S.out[row0 + r].n_cand : 0;
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        uint32_t off = C + incl - cnt;
        uint32_t tot = __shfl(incl, 63);
        if (!S.force_serial && C + tot <= (uint32_t)CM && cnt) {
            uint32_t s = S.chunks[row0 + r].x;
            for (uint32_t i = 0; i < cnt; i++) {
                l_sc[off + i] = (uint32_t)S.c_score[s + i];
                l_q0[off + i] = S.c_q0[s + i]; l_q1[off + i] = S.c_q1[s + i];
                l_r0[off + i] = S.c_r0[s + i]; l_r1[off + i] = S.c_r1[s + i]; l_rc[off + i] = S.c_rc[s + i];
                l_row[off + i] = r; l_n[off + i] = S.c_n[s + i];
            }
        }
        C += tot;
        if (C > (uint32_t)CM && !S.force_serial) break;      // does not fit this tier whatever follows (a Gb-scale pair has 150 000 rows to count otherwise)
    }
    if (C == 0) return;
    if (S.tiny_done && nrows <= TINY_ROWS && C <= TINY_CANDS) return;      // select_tiny_kernel took it
    if (S.force_serial) {   // cross-check path: O(C^2) by one lane (run by the first tier only)
        if (first_tier && lane == 0) { select_serial(S, row0, nrows); atomicAdd(&S.stats[3], 1u); }
        return;
    }
    if (C > (uint32_t)CM) { if (lane == 0) over_list[atomicAdd(over_count, 1u)] = p; return; }   // the next tier takes the pairs that do not fit in this one's LDS (an append per such pair)
    uint32_t P = 64; while (P < C) P <<= 1;
    lds_wave_sync();
    // ---- which candidates overlap ANY other candidate? Only those need the sequential greedy - and only those need to be in priority
    // order: a candidate that overlaps nothing is kept whatever its rank (the commits add and take minima / maxima: any order), so the one
    // full-length sort of a pair is the reference-order one (two of them were 2/3 of this kernel's time at ~300 candidates per 5 Mb pair) ----
    // query side: chunk mates are neighbours in generation order
    for (uint32_t i = lane; i < C; i += 64) {
        const uint32_t row = l_row[i], q0 = l_q0[i], q1 = l_q1[i];
        bool cf = false;
        for (uint32_t j = i; j-- > 0 && l_row[j] == row;) if (!(q1 < l_q0[j] || q0 > l_q1[j])) cf = true;
        for (uint32_t j = i + 1; j < C && l_row[j] == row; j++) if (!(q1 < l_q0[j] || q0 > l_q1[j])) cf = true;
        l_conf[i] = cf;
    }
    // reference side: sort by (ref contig, r0); u overlaps an earlier one iff the running max of r1 reaches r0[u],
    // a later one iff the next r0 is <= r1[u]
    for (uint32_t i = lane; i < P; i += 64) { l_key[i] = i < C ? (((unsigned long long)l_rc[i] << 32) | l_r0[i]) : ~0ull; l_idx[i] = (uint16_t)i; }
    lds_wave_sync();
    // bitonic network over P keys with their payload: a lane owns compare-exchange pairs q = lane, lane + 64, ... of every stage (pair q: the index with a
    // zero inserted at the stride's bit, and its partner) and reads ALL of its pairs before it compares and writes - the stage is one LDS round trip
    // instead of one per pair (the loop over indices t that skipped half of them waited for each pair's reads in turn)
    constexpr int NPAIR = CM / 128;
    for (uint32_t kk = 2; kk <= P; kk <<= 1)
        for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
            unsigned long long ka[NPAIR], kb[NPAIR]; uint16_t ia[NPAIR], ib[NPAIR]; uint32_t ta[NPAIR];
#pragma unroll
            for (int u = 0; u < NPAIR; u++) {
                const uint32_t q = (uint32_t)lane + 64u * u;
                const uint32_t t = ((q & ~(jj - 1u)) << 1) | (q & (jj - 1u));
                ta[u] = t;
                if (q < P / 2) { ka[u] = l_key[t]; kb[u] = l_key[t | jj]; ia[u] = l_idx[t]; ib[u] = l_idx[t | jj]; }
            }
#pragma unroll
            for (int u = 0; u < NPAIR; u++) {
                const uint32_t q = (uint32_t)lane + 64u * u, t = ta[u];
                if (q < P / 2) {
                    const bool asc = (t & kk) == 0;
                    if ((ka[u] > kb[u]) == asc) { l_key[t] = kb[u]; l_key[t | jj] = ka[u]; l_idx[t] = ib[u]; l_idx[t | jj] = ia[u]; }
                }
            }
            lds_wave_sync();
        }
    {
        uint32_t carry_rc = 0xFFFFFFFFu, carry_max = 0;      // segmented inclusive max-scan of r1 in reference order; the running maximum BEFORE u decides the backward overlap
        for (uint32_t u0 = 0; u0 < C; u0 += 64) {
            const uint32_t u = u0 + lane;
            const bool in = u < C;
            const unsigned long long ku = in ? l_key[u] : 0ull;
            const uint32_t rc = in ? (uint32_t)(ku >> 32) : 0xFFFFFFFEu, r0 = (uint32_t)ku;
            const uint32_t i = in ? l_idx[u] : 0;
            const uint32_t r1 = in ? l_r1[i] : 0;
            uint32_t v = r1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { uint32_t pv = __shfl_up(v, o), prc = __shfl_up(rc, o); if (lane >= o && prc == rc) v = pv > v ? pv : v; }
            if (rc == carry_rc) v = carry_max > v ? carry_max : v;
            // the inclusive maximum of the element before u (lane 0: the carry), if it is of the same ref contig
            uint32_t pv = __shfl_up(v, 1), prc = __shfl_up(rc, 1);
            if (lane == 0) { pv = carry_max; prc = carry_rc; }
            bool cf = false;
            if (in && u > 0 && prc == rc && pv >= r0) cf = true;
            if (in && u + 1 < C) { const unsigned long long kn = l_key[u + 1]; if ((uint32_t)(kn >> 32) == rc && (uint32_t)kn <= r1) cf = true; }
            if (cf) l_conf[i] = 1;
            carry_rc = __shfl(rc, 63); carry_max = __shfl(v, 63);
        }
    }
    lds_wave_sync();
    // ---- candidates that overlap nothing are kept outright; the others are listed ----
    uint32_t ncf = 0;
    for (uint32_t t0 = 0; t0 < C; t0 += 64) {
        const uint32_t i = t0 + lane;
        const bool cf = i < C && l_conf[i];
        if (i < C && !cf) {
            // (a chunk's only candidate - nine chunks in ten - fills the row's record with one 32-byte store: six L2 atomics otherwise, 1 800 per pair)
            const uint32_t row = l_row[i];
            const bool alone = (i == 0 || l_row[i - 1] != row) && (i + 1 == C || l_row[i + 1] != row);
            if (alone) {
                const uint32_t q0 = l_q0[i], q1 = l_q1[i];
                ChunkOut o{};
                o.anchors = l_n[i]; o.n_intervals = 1; o.n_cand = 1; o.left = q0; o.right = q1; o.cov_q = (uint64_t)(q1 - q0) + 1 + S.two_c;
                S.out[row0 + row] = o;
            } else sel_commit(S, row0 + row, l_q0[i], l_q1[i], l_n[i]);
        }
        unsigned long long bal = __ballot(cf);
        if (cf) l_idx[ncf + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = (uint16_t)i;   // l_idx is free again: the conflicted list
        ncf += (uint32_t)__popcll(bal);
    }
    lds_wave_sync();
    if (ncf == 0) return;
    // ---- the conflicted ones in priority order: (score desc, generation order asc), keys distinct ----
    for (uint32_t t = lane; t < ncf; t += 64) { const uint32_t i = l_idx[t]; l_key[t] = ((unsigned long long)l_sc[i] << 32) | (0xFFFFFFFFu - i); }
    lds_wave_sync();
    if (ncf <= 128u) {      // few: every key counts the keys above it (two LDS broadcast reads per comparison round, no exchange steps)
        for (uint32_t t = lane; t < ncf; t += 64) {
            const unsigned long long my = l_key[t];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < ncf; j++) rank += l_key[j] > my ? 1u : 0u;
            l_ord[rank] = l_idx[t];
        }
    } else {                // many (repeat-rich pairs): bitonic sort, descending, of the padded list
        uint32_t P2 = 64; while (P2 < ncf) P2 <<= 1;
        for (uint32_t t = ncf + lane; t < P2; t += 64) l_key[t] = 0ull;
        lds_wave_sync();
        for (uint32_t kk = 2; kk <= P2; kk <<= 1)
            for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                for (uint32_t t = lane; t < P2; t += 64) {
                    uint32_t ixj = t ^ jj;
                    if (ixj > t) {
                        unsigned long long a = l_key[t], b = l_key[ixj];
                        bool desc = (t & kk) == 0;
                        if ((a < b) == desc) { l_key[t] = b; l_key[ixj] = a; }
                    }
                }
                lds_wave_sync();
            }
        for (uint32_t t = lane; t < ncf; t += 64) l_ord[t] = (uint16_t)(0xFFFFFFFFu - (uint32_t)l_key[t]);
    }
    lds_wave_sync();
    uint32_t nk = 0;
    for (uint32_t t = 0; t < ncf; t++) {
        const uint32_t i = l_ord[t];
        const uint32_t q0 = l_q0[i], q1 = l_q1[i], r0 = l_r0[i], r1 = l_r1[i], rc = l_rc[i], row = l_row[i];
        bool ov = false;
        for (uint32_t j = lane; j < nk; j += 64) {
            const uint32_t k2 = l_kept[j];
            if (l_row[k2] == row && !(q1 < l_q0[k2] || q0 > l_q1[k2])) ov = true;
            else if (l_rc[k2] == rc && !(r1 < l_r0[k2] || r0 > l_r1[k2])) ov = true;
        }
        if (__ballot(ov) == 0) {
            if (lane == 0) { l_kept[nk] = (uint16_t)i; sel_commit(S, row0 + row, q0, q1, l_n[i]); }
            nk++;
            lds_wave_sync();
        }
    }
}

// one wave per LIVE pair (pairs without a chunk table - every rescued short contig against an unrelated reference - never reach
// the selection): a fixed grid walks the device-side list, so a batch of 10^6 pairs of which 10^5 are live does not schedule
// 10^6 workgroups of 51 KB of LDS each to find that out
// Contig pairs: one to three chunks, a handful of candidate chains. A wave that stages them in LDS and runs a 64-key bitonic sort spends
// ~40 us on what is a comparison or two: here ONE LANE takes the pair and runs the serial greedy (the definition the parallel selection
// is checked against) on its few candidates in place (64 -> 23 ms per 10^7 live contig pairs of the metagenome step).
__global__ __launch_bounds__(256) void select_tiny_kernel(SelArgs S) {
    const uint32_t n = *S.n_live;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    bool rest = false; uint32_t p = 0;
    if (k < n) {
        p = S.live[k];
        const uint32_t row0 = S.cbase[p], nrows = S.n_chunks[p];
        if (nrows != 0) {
            uint32_t C = 0;
            if (nrows <= TINY_ROWS) for (uint32_t r = 0; r < nrows; r++) C += S.out[row0 + r].n_cand;
            if (nrows > TINY_ROWS || C > TINY_CANDS) rest = true;
            else if (C != 0) select_serial(S, row0, nrows);
        }
    }
    // the pairs left for the wave kernel, listed (one append per wave): walking all 17 M live pairs of a metagenome step again only to find the
    // few with more candidates cost that kernel 18 ms
    const unsigned long long bal = __ballot(rest);
    if (bal) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == __ffsll((long long)bal) - 1) base = atomicAdd(S.rest_count, (uint32_t)__popcll(bal));
        base = __shfl(base, __ffsll((long long)bal) - 1);
        if (rest) S.rest_list[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = p;
    }
}

__global__ __launch_bounds__(64) void select_kernel(SelArgs S, uint32_t* __restrict__ mid_list, uint32_t* __restrict__ mid_count) {
    const uint32_t* list = S.tiny_done ? S.rest_list : S.live;      // after select_tiny_kernel: only what it left
    const uint32_t n = S.tiny_done ? *S.rest_count : (S.live ? *S.n_live : S.n_pairs);      // small launches skip the list: every pair is visited
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        select_pair<CSMALL>(S, list ? list[k] : k, mid_list, mid_count, true);
        lds_wave_sync();
    }
}
// second tier: the pairs with more than CSMALL candidates
__global__ __launch_bounds__(64) void select_mid_kernel(SelArgs S, const uint32_t* __restrict__ mid_list, const uint32_t* __restrict__ mid_count) {
    const uint32_t n = *mid_count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        select_pair<CMAX>(S, mid_list[k], S.big_list, S.big_count, false);
        lds_wave_sync();
    }
}

// ---- pairs with more than CMAX candidate chains (genomes beyond ~10 Mb): the same algorithm on global scratch, by a GROUP of
// G workgroups of 1024 threads per pair (G = 1 for up to BIG_SOLO candidates; for Gb-scale pairs the cooperative kernel below
// gives every pair 8..128 workgroups that meet at a counter barrier between phases). Scratch is indexed from the pair's first
// anchor: a pair with n anchors has at most n/3 candidates, and the padded sort length stays below n.
struct BigArgs {
    SelArgs S; const uint32_t* pstart;
    unsigned long long* key;   // sort keys
    uint32_t *slot, *crow;     // candidate j -> global candidate slot, chunk row (generation order)
    uint32_t *idx, *pm, *pm2;  // payload of the reference-order sort; running max of r1 (double buffer)
    uint32_t *ord, *clist;
    uint8_t* conf;
    uint32_t *huge_list, *huge_count;   // pairs with more than BIG_SOLO candidates, listed by the solo kernel for the cooperative one
    uint32_t *ctr, *parts;              // per group: barrier counter; 2 x BIG_GMAX partial sums of the two ordered compactions
    uint32_t solo;                      // BIG_SOLO ($PSK_BIG_SOLO in tests)
};
constexpr int BIG_T = 1024;
constexpr uint32_t BIG_TILE = 4096;       // keys of one LDS-staged sort tile
constexpr uint32_t BIG_SOLO = 32768;      // up to here one workgroup per pair: a barrier between workgroups costs more than it divides
constexpr uint32_t BIG_GMAX = 128;        // workgroups of the cooperative launch (co-resident: one per CU, at most two launches per CU pair of lanes)
constexpr uint32_t BIG_GROUPS = 16;       // pairs in flight in the cooperative launch

struct BigGrp { uint32_t G, gr, epoch; uint32_t* ctr; uint32_t* part_a; uint32_t* part_b; };

// all G workgroups of the group arrive; global writes made before are visible to every member after (the recipe of a grid-wide
// sync: workgroup barrier, agent-scope release by one thread, counter, agent-scope acquire, workgroup barrier)
__device__ __forceinline__ void grp_sync(BigGrp& g) {
    __syncthreads();
    if (g.G > 1) {
        g.epoch++;
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(g.ctr, 1u);
            const uint32_t target = g.epoch * g.G;
            while (__hip_atomic_load(g.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
}

// Bitonic sort of P (a power of two >= 1024) keys in global memory by the group. Compare-exchange distances below BIG_TILE stay
// inside an aligned tile of BIG_TILE keys, so those passes run on a tile staged in LDS (one global round trip per tile and stage
// instead of one per pass: 36 instead of 190 sweeps over the array for 2^19 keys); only the longer distances sweep global memory.
__device__ void big_bitonic(unsigned long long* key, uint32_t* pay, uint32_t P, bool descending, BigGrp& g, unsigned long long* t_key, uint32_t* t_pay) {
    const uint32_t tile = P < BIG_TILE ? P : BIG_TILE;
    auto tile_passes = [&](uint32_t kk_first, uint32_t kk_last) {
        for (uint32_t b = g.gr * tile; b < P; b += g.G * tile) {
            for (uint32_t t = threadIdx.x; t < tile; t += BIG_T) { t_key[t] = key[b + t]; if (pay) t_pay[t] = pay[b + t]; }
            __syncthreads();
            for (uint32_t kk = kk_first; kk <= kk_last; kk <<= 1)
                for (uint32_t jj = (kk >> 1) < tile ? (kk >> 1) : (tile >> 1); jj > 0; jj >>= 1) {
                    for (uint32_t c = threadIdx.x; c < (tile >> 1); c += BIG_T) {
                        const uint32_t t = ((c & ~(jj - 1)) << 1) | (c & (jj - 1)), u = t | jj;
                        const unsigned long long a = t_key[t], v = t_key[u];
                        const bool up = (((b + t) & kk) == 0) != descending;
                        if ((a > v) == up) {
                            t_key[t] = v; t_key[u] = a;
                            if (pay) { const uint32_t pa = t_pay[t]; t_pay[t] = t_pay[u]; t_pay[u] = pa; }
                        }
                    }
                    __syncthreads();
                }
            for (uint32_t t = threadIdx.x; t < tile; t += BIG_T) { key[b + t] = t_key[t]; if (pay) pay[b + t] = t_pay[t]; }
            __syncthreads();
        }
        grp_sync(g);
    };
    tile_passes(2, tile);                                   // every tile sorted (direction by its position)
    for (uint32_t kk = tile << 1; kk <= P; kk <<= 1) {
        for (uint32_t jj = kk >> 1; jj >= tile; jj >>= 1) {
            for (uint32_t c = g.gr * BIG_T + threadIdx.x; c < (P >> 1); c += g.G * BIG_T) {
                const uint32_t t = ((c & ~(jj - 1)) << 1) | (c & (jj - 1)), u = t | jj;
                const unsigned long long a = key[t], v = key[u];
                const bool up = ((t & kk) == 0) != descending;
                if ((a > v) == up) {
                    key[t] = v; key[u] = a;
                    if (pay) { const uint32_t pa = pay[t]; pay[t] = pay[u]; pay[u] = pa; }
                }
            }
            grp_sync(g);
        }
        tile_passes(kk, kk);
    }
}

// inclusive scan of one value per thread over the workgroup (s_scan: BIG_T words)
__device__ __forceinline__ uint32_t big_block_scan(uint32_t v, uint32_t* s_scan) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t x = __shfl_up(v, o); if (lane >= (uint32_t)o) v += x; }
    __syncthreads();
    if (lane == 63) s_scan[w] = v;
    __syncthreads();
    if (w == 0) {
        uint32_t x = lane < BIG_T / 64 ? s_scan[lane] : 0;
#pragma unroll
        for (int o = 1; o < BIG_T / 64; o <<= 1) { uint32_t y = __shfl_up(x, o); if (lane >= (uint32_t)o) x += y; }
        if (lane < BIG_T / 64) s_scan[lane] = x;
    }
    __syncthreads();
    return v + (w ? s_scan[w - 1] : 0);
}

// number of candidate chains of a pair (sum of its rows' counts), by one workgroup
__device__ uint32_t big_count_candidates(const SelArgs& S, uint32_t p, uint32_t* s_scan) {
    const uint32_t row0 = S.cbase[p], nrows = S.n_chunks[p];
    uint32_t c = 0;
    for (uint32_t r = threadIdx.x; r < nrows; r += BIG_T) c += S.out[row0 + r].n_cand;
    const uint32_t incl = big_block_scan(c, s_scan);
    __syncthreads();
    if (threadIdx.x == BIG_T - 1) s_scan[BIG_T / 64] = incl;
    __syncthreads();
    const uint32_t tot = s_scan[BIG_T / 64];
    __syncthreads();
    return tot;
}

__device__ void select_big_pair(const BigArgs& B, const uint32_t p, BigGrp& g, unsigned char* s_raw, uint32_t* s_scan) {
    __shared__ uint32_t s_carry;
    const SelArgs& S = B.S;
    const uint32_t tid = threadIdx.x, G = g.G, gr = g.gr;
    const uint32_t gt = gr * BIG_T + tid, GT = G * BIG_T;          // this thread in the group, threads of the group
    const uint32_t row0 = S.cbase[p], nrows = S.n_chunks[p];
    const uint32_t base = B.pstart[p];
    unsigned long long* key = B.key + base; uint32_t* slot = B.slot + base; uint32_t* crow = B.crow + base;
    uint32_t* idx = B.idx + base; uint32_t* pm = B.pm + base; uint32_t* pm2 = B.pm2 + base;
    uint32_t* ord = B.ord + base; uint32_t* clist = B.clist + base; uint8_t* conf = B.conf + base;
    unsigned long long* t_key = (unsigned long long*)s_raw; uint32_t* t_pay = (uint32_t*)(s_raw + 8 * BIG_TILE);
    __syncthreads();
    // ---- candidates in generation order: every workgroup takes a span of rows; spans are stitched by the partial sums ----
    const uint32_t rspan = (((nrows + G - 1) / G) + BIG_T - 1) / BIG_T * BIG_T;
    const uint32_t ra = gr * rspan < nrows ? gr * rspan : nrows, rb = ra + rspan < nrows ? ra + rspan : nrows;
    {
        uint32_t c = 0;
        for (uint32_t r = ra + tid; r < rb; r += BIG_T) c += S.out[row0 + r].n_cand;
        const uint32_t incl = big_block_scan(c, s_scan);
        if (tid == BIG_T - 1) g.part_a[gr] = incl;
    }
    grp_sync(g);
    uint32_t C = 0, pre = 0;
    for (uint32_t w = 0; w < G; w++) { const uint32_t v = g.part_a[w]; if (w < gr) pre += v; C += v; }
    if (tid == 0) s_carry = pre;
    __syncthreads();
    for (uint32_t r0 = ra; r0 < rb; r0 += BIG_T) {
        const uint32_t r = r0 + tid;
        const uint32_t cnt = r < rb ? S.out[row0 + r].n_cand : 0;
        const uint32_t incl = big_block_scan(cnt, s_scan);
        const uint32_t off = s_carry + incl - cnt;
        if (cnt) { const uint32_t sl = S.chunks[row0 + r].x; for (uint32_t i = 0; i < cnt; i++) { slot[off + i] = sl + i; crow[off + i] = r; } }
        __syncthreads();
        if (tid == BIG_T - 1) s_carry += incl;
        __syncthreads();
    }
    uint32_t P = 1024; while (P < C) P <<= 1;
    grp_sync(g);
    // (only the candidates that overlap another one need to be in priority order: the one full-length sort is the reference-order one, as in select_pair)
    // ---- conflicts: chunk mates on the query ----
    for (uint32_t j = gt; j < C; j += GT) {
        const uint32_t row = crow[j], q0 = S.c_q0[slot[j]], q1 = S.c_q1[slot[j]];
        bool cf = false;
        for (uint32_t v = j; v-- > 0 && crow[v] == row;) if (!(q1 < S.c_q0[slot[v]] || q0 > S.c_q1[slot[v]])) cf = true;
        for (uint32_t v = j + 1; v < C && crow[v] == row; v++) if (!(q1 < S.c_q0[slot[v]] || q0 > S.c_q1[slot[v]])) cf = true;
        conf[j] = cf;
    }
    grp_sync(g);
    // ---- conflicts on the reference: order by (ref contig, r0), running max of r1 by doubling ----
    for (uint32_t j = gt; j < P; j += GT) { key[j] = j < C ? (((unsigned long long)S.c_rc[slot[j]] << 32) | S.c_r0[slot[j]]) : ~0ull; idx[j] = j; }
    grp_sync(g);
    big_bitonic(key, idx, P, false, g, t_key, t_pay);
    for (uint32_t u = gt; u < C; u += GT) pm[u] = S.c_r1[slot[idx[u]]];
    grp_sync(g);
    uint32_t* src = pm; uint32_t* dst = pm2;
    for (uint32_t o = 1; o < C; o <<= 1) {
        for (uint32_t u = gt; u < C; u += GT) {
            uint32_t v = src[u];
            if (u >= o && (uint32_t)(key[u - o] >> 32) == (uint32_t)(key[u] >> 32)) { uint32_t w = src[u - o]; v = w > v ? w : v; }
            dst[u] = v;
        }
        grp_sync(g);
        uint32_t* t2 = src; src = dst; dst = t2;
    }
    for (uint32_t u = gt; u < C; u += GT) {
        const uint32_t j = idx[u];
        const uint32_t rc = (uint32_t)(key[u] >> 32), r0 = (uint32_t)key[u], r1 = S.c_r1[slot[j]];
        bool cf = false;
        if (u > 0 && (uint32_t)(key[u - 1] >> 32) == rc && src[u - 1] >= r0) cf = true;
        if (u + 1 < C && (uint32_t)(key[u + 1] >> 32) == rc && (uint32_t)key[u + 1] <= r1) cf = true;
        if (cf) conf[j] = 1;
    }
    grp_sync(g);
    // ---- unconflicted chains are kept; conflicted ones listed (spans of candidates, stitched as above), then put in priority order ----
    const uint32_t tspan = (((C + G - 1) / G) + BIG_T - 1) / BIG_T * BIG_T;
    const uint32_t ta = gr * tspan < C ? gr * tspan : C, tb = ta + tspan < C ? ta + tspan : C;
    {
        uint32_t c = 0;
        for (uint32_t t = ta + tid; t < tb; t += BIG_T) {
            const uint32_t j = t;
            if (conf[j]) c++;
            else {
                const uint32_t sl = slot[j], row = crow[j];
                if ((j == 0 || crow[j - 1] != row) && (j + 1 == C || crow[j + 1] != row)) {      // the chunk's only candidate: the row's record in one store (select_pair)
                    const uint32_t q0 = S.c_q0[sl], q1 = S.c_q1[sl];
                    ChunkOut o{};
                    o.anchors = S.c_n[sl]; o.n_intervals = 1; o.n_cand = 1; o.left = q0; o.right = q1; o.cov_q = (uint64_t)(q1 - q0) + 1 + S.two_c;
                    S.out[row0 + row] = o;
                } else sel_commit(S, row0 + row, S.c_q0[sl], S.c_q1[sl], S.c_n[sl]);
            }
        }
        const uint32_t incl = big_block_scan(c, s_scan);
        if (tid == BIG_T - 1) g.part_b[gr] = incl;
    }
    grp_sync(g);
    uint32_t ncf = 0; pre = 0;
    for (uint32_t w = 0; w < G; w++) { const uint32_t v = g.part_b[w]; if (w < gr) pre += v; ncf += v; }
    if (tid == 0) s_carry = pre;
    __syncthreads();
    for (uint32_t t0 = ta; t0 < tb; t0 += BIG_T) {
        const uint32_t t = t0 + tid;
        const uint32_t j = t < tb ? t : 0;
        const uint32_t cf = (t < tb && conf[j]) ? 1u : 0u;
        const uint32_t incl = big_block_scan(cf, s_scan);
        if (cf) clist[s_carry + incl - 1] = j;
        __syncthreads();
        if (tid == BIG_T - 1) s_carry += incl;
        __syncthreads();
    }
    grp_sync(g);
    if (ncf) {      // (score desc, generation order asc) over the conflicted ones only
        uint32_t P2 = 1024; while (P2 < ncf) P2 <<= 1;
        for (uint32_t t = gt; t < P2; t += GT) { const uint32_t j = t < ncf ? clist[t] : 0; key[t] = t < ncf ? (((unsigned long long)(uint32_t)S.c_score[slot[j]] << 32) | (0xFFFFFFFFu - j)) : 0ull; }
        grp_sync(g);
        big_bitonic(key, nullptr, P2, true, g, t_key, t_pay);
        for (uint32_t t = gt; t < ncf; t += GT) clist[t] = 0xFFFFFFFFu - (uint32_t)key[t];
        grp_sync(g);
    }
    if (gr != 0) return;
    // ---- greedy over the conflicted chains in priority order, by the group's first workgroup: candidates staged 1024 at a time
    // in LDS, kept chains in LDS (the first KL) and in compact global arrays (pm, pm2, idx, ord and the key array are free now) ----
    constexpr uint32_t KL = 512;
    uint32_t* L = (uint32_t*)s_raw;
    uint32_t *g_q0 = L, *g_q1 = L + BIG_T, *g_r0 = L + 2 * BIG_T, *g_r1 = L + 3 * BIG_T, *g_rc = L + 4 * BIG_T, *g_row = L + 5 * BIG_T, *g_n = L + 6 * BIG_T;
    uint32_t* K = L + 7 * BIG_T;
    uint32_t *l_q0 = K, *l_q1 = K + KL, *l_r0 = K + 2 * KL, *l_r1 = K + 3 * KL, *l_rc = K + 4 * KL, *l_row = K + 5 * KL;
    uint32_t *k_q0 = pm, *k_q1 = pm2, *k_r0 = idx, *k_r1 = ord, *k_rc = (uint32_t*)key, *k_row = (uint32_t*)key + C;
    // 64 candidates at a time (a Gb-scale pair has ~1 800 conflicted chains of 150 000 - 500 000 candidates; one candidate per round of the whole workgroup
    // was 0.8 us each, 1.5 of the kernel's 4 ms): every wave tests all 64 against its share of the kept list, then the first wave settles the 64 among
    // themselves - lane i knows which EARLIER candidates of the block it overlaps, and a 64-step scan over a uniform mask of the accepted ones replays the
    // sequential rule exactly (kept iff no overlap with anything kept before it, in priority order).
    __shared__ unsigned long long s_ov[BIG_T / 64];
    __shared__ uint32_t s_nk;
    const int lane = tid & 63, wave = tid >> 6;
    uint32_t nk = 0;
    for (uint32_t t0 = 0; t0 < ncf; t0 += BIG_T) {
        const uint32_t nb = ncf - t0 < (uint32_t)BIG_T ? ncf - t0 : (uint32_t)BIG_T;
        __syncthreads();
        if (tid < nb) {
            const uint32_t j = clist[t0 + tid], sl = slot[j];
            g_q0[tid] = S.c_q0[sl]; g_q1[tid] = S.c_q1[sl]; g_r0[tid] = S.c_r0[sl]; g_r1[tid] = S.c_r1[sl]; g_rc[tid] = S.c_rc[sl];
            g_row[tid] = crow[j]; g_n[tid] = S.c_n[sl];
        }
        __syncthreads();
        for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
            const uint32_t bn = nb - b0 < 64u ? nb - b0 : 64u;
            const bool have = (uint32_t)lane < bn;
            const uint32_t ci = b0 + (have ? lane : 0);
            const uint32_t q0 = g_q0[ci], q1 = g_q1[ci], r0 = g_r0[ci], r1 = g_r1[ci], rc = g_rc[ci], row = g_row[ci];
            bool ov = false;
            for (uint32_t v = wave; v < nk; v += BIG_T / 64) {      // (v is the wave's: the kept entry is read once and broadcast)
                uint32_t a0, a1, b0r, b1r, bc, brow;
                if (v < KL) { a0 = l_q0[v]; a1 = l_q1[v]; b0r = l_r0[v]; b1r = l_r1[v]; bc = l_rc[v]; brow = l_row[v]; }
                else { a0 = k_q0[v]; a1 = k_q1[v]; b0r = k_r0[v]; b1r = k_r1[v]; bc = k_rc[v]; brow = k_row[v]; }
                if (brow == row && !(q1 < a0 || q0 > a1)) ov = true;
                else if (bc == rc && !(r1 < b0r || r0 > b1r)) ov = true;
            }
            const unsigned long long wov = __ballot(ov && have);
            if (lane == 0) s_ov[wave] = wov;
            __syncthreads();
            if (wave == 0) {
                unsigned long long dead = 0;
#pragma unroll
                for (int w = 0; w < BIG_T / 64; w++) dead |= s_ov[w];
                unsigned long long mine = 0;      // earlier candidates of the block this one overlaps
                for (uint32_t j = 0; j < bn; j++) {
                    const uint32_t cj = b0 + j;
                    const uint32_t a0 = g_q0[cj], a1 = g_q1[cj], b0r = g_r0[cj], b1r = g_r1[cj], bc = g_rc[cj], brow = g_row[cj];
                    const bool hit = (brow == row && !(q1 < a0 || q0 > a1)) || (bc == rc && !(r1 < b0r || r0 > b1r));
                    if (hit && j < (uint32_t)lane) mine |= 1ull << j;
                }
                const bool alive = have && !((dead >> lane) & 1ull);
                unsigned long long accepted = 0;
                for (uint32_t i = 0; i < bn; i++) {
                    const unsigned long long okm = __ballot(alive && (mine & accepted) == 0);
                    if ((okm >> i) & 1ull) accepted |= 1ull << i;
                }
                if ((accepted >> lane) & 1ull) {
                    const uint32_t at = nk + (uint32_t)__popcll(accepted & ((1ull << lane) - 1ull));
                    if (at < KL) { l_q0[at] = q0; l_q1[at] = q1; l_r0[at] = r0; l_r1[at] = r1; l_rc[at] = rc; l_row[at] = row; }
                    else { k_q0[at] = q0; k_q1[at] = q1; k_r0[at] = r0; k_r1[at] = r1; k_rc[at] = rc; k_row[at] = row; }
                    sel_commit(S, row0 + row, q0, q1, g_n[ci]);
                }
                if (lane == 0) s_nk = nk + (uint32_t)__popcll(accepted);
                __threadfence_block();
            }
            __syncthreads();
            nk = s_nk;
        }
    }
    if (tid == 0) atomicAdd(&S.stats[3], 1u);
}

// the pairs select_kernel listed (more than CMAX candidates): a small fixed grid walks the list, one workgroup per pair; pairs
// with more than BIG_SOLO candidates are passed on to the cooperative kernel
__global__ __launch_bounds__(BIG_T) void select_big_kernel(BigArgs B) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[12 * BIG_TILE];
    __shared__ uint32_t s_scan[BIG_T / 64 + 1];
    if (B.S.force_serial) return;
    const uint32_t n = *B.S.big_count;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const uint32_t p = B.S.big_list[k];
        if (big_count_candidates(B.S, p, s_scan) > B.solo) { if (threadIdx.x == 0) B.huge_list[atomicAdd(B.huge_count, 1u)] = p; continue; }
        BigGrp g{1, 0, 0, nullptr, B.parts, B.parts + BIG_GMAX};
        g.part_a = B.parts + (size_t)(BIG_GROUPS + blockIdx.x) * 2 * BIG_GMAX; g.part_b = g.part_a + BIG_GMAX;
        select_big_pair(B, p, g, s_raw, s_scan);
        __syncthreads();
    }
}

// Gb-scale pairs: the launch's workgroups (all co-resident: at most BIG_GMAX, one per CU) split into min(16, pairs) groups, each
// group takes every groups-th listed pair. Workgroups of a group sit on as few XCDs as possible (workgroup b runs on XCD b % 8).
__global__ __launch_bounds__(BIG_T) void select_huge_kernel(BigArgs B) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[12 * BIG_TILE];
    __shared__ uint32_t s_scan[BIG_T / 64 + 1];
    const uint32_t n = *B.huge_count;
    if (n == 0) return;
    const uint32_t NB = gridDim.x;                  // a power of two, >= 8 (or 1)
    uint32_t groups = 1; while (groups < n && groups < BIG_GROUPS && groups < NB) groups <<= 1;
    const uint32_t G = NB / groups;
    uint32_t gid, gr;
    if (NB < 8) { gid = blockIdx.x / G; gr = blockIdx.x % G; }
    else {
        const uint32_t xcd = blockIdx.x & 7, sl = blockIdx.x >> 3, per_xcd = NB >> 3;
        if (groups >= 8) { const uint32_t gpx = groups >> 3; gid = xcd * gpx + sl / G; gr = sl % G; }
        else { gid = xcd % groups; gr = (xcd / groups) * per_xcd + sl; }
    }
    BigGrp g{G, gr, 0, B.ctr + gid, B.parts + (size_t)gid * 2 * BIG_GMAX, B.parts + (size_t)gid * 2 * BIG_GMAX + BIG_GMAX};
    for (uint32_t k = gid; k < n; k += groups) {
        select_big_pair(B, B.huge_list[k], g, s_raw, s_scan);
        __syncthreads();
    }
}

// seeds of the query between the leftmost and rightmost kept anchor of every chunk
__global__ __launch_bounds__(256) void chunk_seeds_kernel(ChainArgs A) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pair = A.row_pair[row < A.n_rows ? row : A.n_rows - 1];
    if (row >= A.n_rows) return;
    if (row - A.cbase[pair] >= A.n_chunks[pair]) return;
    ChunkOut* o = &A.out[row];
    if (!o->n_intervals) return;
    // (an attempt that is going to be rerun - a pair that outgrew its room in the one-walk index join, an anchor total beyond the capacity - leaves rows whose anchors
    // were never written: what they point at is a previous batch's, and this kernel is the one that uses an anchor's content as an INDEX. Nothing is read through it
    // unchecked: found by a 480-seed fuzz sweep as a memory fault that needed seventeen earlier cases' leftovers in the scratch arrays)
    const uint2 ch0 = A.chunks[row];
    if (ch0.x > ch0.y || ch0.y > A.cap) return;
    const uint32_t nc = A.pairs[pair].q_nc;
    const uint32_t qc = nc == 1u ? 0u : A.anc[ch0.x].w;      // (a one-contig query - most complete bacterial genomes - needs no look at the chunk's anchors: a cold 64-byte line per row)
    if (qc < nc) o->seeds = seeds_between(A.pairs[pair], qc, o->left, o->right);
}

// ------------------------------------------------------------------ per-pair ANI / AF
struct ReduceArgs {
    const ChunkOut* chunks; const uint32_t* n_chunks; const uint32_t* cbase;
    const uint32_t* pstart; const PairDesc* pairs;
    const uint32_t* pcnt;   // anchors per pair where pstart does not say (the one-pass index join: pstart = item offsets); null: pstart[p + 1] - pstart[p]
    const uint2* pair_qr;   // (query, reference) of every pair: travels with the hit (reserved, ref_index)
    const uint32_t* live; const uint32_t* n_live;   // pairs with a chunk table
    int small_done;                                 // chunk tables of <= 64 rows are reduced by pair_reduce_small_kernel
    int wave_done;                                  // ... and those of 65 .. 64 RW_PER rows by pair_reduce_wave_kernel
    int tiny_done;                                  // ... and those of 1 .. 4 rows (contig pairs, mean ANI) by pair_reduce_tiny_kernel, a lane per pair
    int k, median, robust; double min_af;
    psk_hit* hits;
    double* big_vals;   // 2 * rows(+pad) doubles per launch: sort space for pairs with more than RED_CAP chunk values
};
constexpr int RED_CAP = 4096;   // chunk ANI values sortable in LDS (genomes up to ~80 Mb at 20 kb chunks)

// pairs without a chunk table (fewer than MIN_ANCHORS anchors: every rescued short contig against an unrelated reference): one
// empty record each, one lane per pair
__global__ __launch_bounds__(256) void pair_empty_kernel(ReduceArgs R, uint32_t n_pairs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs || R.n_chunks[p] != 0) return;
    psk_hit h{};
    h.ani = -1.0f; h.ani_raw = -1.0f;
    h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
    h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
    R.hits[p] = h;
}

// Two instantiations, like the selection: CAP = RED_SMALL for the bulk (12 KB of LDS instead of 49: residency: 4.1 -> 3.4 ms per 10^5
// pairs of 5 Mb genomes), CAP = RED_CAP for the pairs with more chunk rows than that (and, beyond RED_CAP values, the global sort).
constexpr int RED_SMALL = 1024;
// groups of 64 chunk rows one wave reduces (pair_reduce_wave_kernel)
constexpr int RW_PER = 8;      // 512 rows: a 5 Mb genome has ~250 chunks, and the pairs just past 256 rows took a workgroup each (15 of the 19 ms of reduction per 10 000 x 10 000 step)
template <int CAP>
__device__ void pair_reduce_pair(const ReduceArgs& R, const uint32_t p) {
    __shared__ double s_v[CAP];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_acc[5];
    const uint32_t nc = R.n_chunks[p];
    if (CAP == RED_SMALL ? nc > (uint32_t)RED_SMALL : nc <= (uint32_t)RED_SMALL) return;      // the other instantiation's pair
    if (R.small_done && nc != 0 && nc <= 64) return;      // pair_reduce_small_kernel took it
    if (R.wave_done && nc > 64 && nc <= 64u * RW_PER) return;      // pair_reduce_wave_kernel took it
    if (nc == 0) {      // only reached when the launch visits every pair (no live list): the empty record of pair_empty_kernel
        if (threadIdx.x == 0) {
            psk_hit h{};
            h.ani = -1.0f; h.ani_raw = -1.0f;
            h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
            h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
            R.hits[p] = h;
        }
        return;
    }
    const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
    if (threadIdx.x == 0) { s_n = 0; for (int i = 0; i < 5; i++) s_acc[i] = 0; }
    __syncthreads();
    // integer totals (order-free) and the number of chunks that kept a chain
    unsigned long long t_cq = 0, t_cr = 0, t_a = 0, t_s = 0, t_i = 0; uint32_t t_m = 0;
    for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) { const uint32_t ni = co[i].n_intervals; t_cq += co[i].cov_q; t_cr += co[i].cov_q; t_a += co[i].anchors; t_s += ni ? co[i].seeds : 0; t_i += ni; t_m += ni != 0; }
    atomicAdd(&s_acc[0], t_cq); atomicAdd(&s_acc[1], t_cr); atomicAdd(&s_acc[2], t_a); atomicAdd(&s_acc[3], t_s); atomicAdd(&s_acc[4], t_i);
    atomicAdd(&s_n, t_m);
    __syncthreads();
    // their rows compacted in chunk order (the oracle's summation order), 256 rows per step: ballot ranks inside a wave, the four
    // wave totals through LDS. Not needed beyond RED_CAP values (those pairs never index s_idx).
    __shared__ uint32_t s_idx[CAP];
    __shared__ uint32_t s_wt[2][4];
    if (s_n <= (uint32_t)CAP) {
        uint32_t run = 0;
        const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (uint32_t i0 = 0, it = 0; i0 < nc; i0 += 256, it++) {
            const uint32_t i = i0 + threadIdx.x;
            const bool f = i < nc && co[i].n_intervals != 0;
            const unsigned long long bal = __ballot(f);
            if (lane == 0) s_wt[it & 1][w] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, tot = 0;
            for (uint32_t x = 0; x < 4; x++) { const uint32_t c = s_wt[it & 1][x]; if (x < w) before += c; tot += c; }
            if (f) s_idx[run + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = i;
            run += tot;
        }
    }
    __syncthreads();
    const uint32_t m = s_n;
    psk_hit h{};
    h.ani = -1.0f;
    const bool overflow = m > (uint32_t)CAP;
    double mean_serial = 0;   // only thread 0 uses it
    if (!overflow) {
        for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) {
            const ChunkOut c = co[s_idx[j]];
            double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
            if (ratio > 1.0) ratio = 1.0;
            s_v[j] = pow(ratio, 1.0 / (double)R.k);
        }
        __syncthreads();
        if (R.median || R.robust) {   // bitonic sort of s_v[0..m) padded with +inf
            uint32_t P = 1; while (P < m) P <<= 1;
            for (uint32_t j = m + threadIdx.x; j < P; j += blockDim.x) s_v[j] = INFINITY;
            __syncthreads();
            for (uint32_t kk = 2; kk <= P; kk <<= 1)
                for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (uint32_t t = threadIdx.x; t < P; t += blockDim.x) {
                        uint32_t ixj = t ^ jj;
                        if (ixj > t) {
                            double a = s_v[t], b = s_v[ixj];
                            bool up = (t & kk) == 0;
                            if ((a > b) == up) { s_v[t] = b; s_v[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
        }
    } else if (R.median || R.robust) {   // very long genomes: sort the chunk values in global scratch
        double* gv = R.big_vals + 2 * (size_t)R.cbase[p] + 1024 * (size_t)p;
        uint32_t P = 1024; while (P < nc) P <<= 1;
        for (uint32_t i = threadIdx.x; i < P; i += blockDim.x) {
            double v = INFINITY;
            if (i < nc && co[i].n_intervals) {
                double ratio = (double)co[i].anchors / (double)(co[i].seeds > 1 ? co[i].seeds - 1 : 1); if (ratio > 1.0) ratio = 1.0;
                v = pow(ratio, 1.0 / (double)R.k);
            }
            gv[i] = v;
        }
        __syncthreads();
        for (uint32_t kk = 2; kk <= P; kk <<= 1)
            for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
                for (uint32_t t = threadIdx.x; t < P; t += blockDim.x) {
                    uint32_t ixj = t ^ jj;
                    if (ixj > t) {
                        double a = gv[t], b = gv[ixj];
                        bool up = (t & kk) == 0;
                        if ((a > b) == up) { gv[t] = b; gv[ixj] = a; }
                    }
                }
                __syncthreads();
            }
    }
    __syncthreads();
    // mean and sample standard deviation of ALL chunk values (feature of the learned-ANI regression; also the mean of
    // pairs beyond RED_CAP chunks): two block-parallel passes, fixed thread -> element mapping (deterministic)
    __shared__ double s_red[8];
    auto block_sum = [&](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        return s_red[0] + s_red[1] + s_red[2] + s_red[3];
    };
    auto chunk_val = [&](uint32_t i) {
        double ratio = (double)co[i].anchors / (double)(co[i].seeds > 1 ? co[i].seeds - 1 : 1); if (ratio > 1.0) ratio = 1.0;
        return pow(ratio, 1.0 / (double)R.k);
    };
    double part = 0;
    if (!overflow) { for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) part += s_v[j]; }
    else { for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) if (co[i].n_intervals) part += chunk_val(i); }
    const double mean_all = m ? block_sum(part) / (double)m : 0.0;
    part = 0;
    if (!overflow) { for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) { const double d = s_v[j] - mean_all; part += d * d; } }
    else { for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) if (co[i].n_intervals) { const double d = chunk_val(i) - mean_all; part += d * d; } }
    const double ssq = block_sum(part);
    const double std_all = m > 1 ? sqrt(ssq / (double)(m - 1)) : 0.0;
    mean_serial = mean_all;
    if (threadIdx.x == 0) {
        h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
        h.n_chunks = m; h.n_intervals = (uint32_t)s_acc[4];
        h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
        h.covered_query = s_acc[0]; h.covered_ref = s_acc[1]; h.sum_chain_anchors = s_acc[2]; h.sum_chunk_seeds = s_acc[3];
        if (m > 0) {
            double ani;
            bool ok = true;
            if (overflow && (R.median || R.robust)) {
                const double* gv = R.big_vals + 2 * (size_t)R.cbase[p] + 1024 * (size_t)p;
                if (R.median) ani = gv[m / 2];
                else {
                    uint32_t lo = 0, hi = m;
                    if (m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                    double sum = 0; for (uint32_t i = lo; i < hi; i++) sum += gv[i];
                    ani = sum / (double)(hi - lo);
                }
            }
            else if (overflow) ani = mean_serial;
            else if (R.median) ani = s_v[m / 2];
            else {
                uint32_t lo = 0, hi = m;
                if (R.robust && m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                double sum = 0; for (uint32_t i = lo; i < hi; i++) sum += s_v[i];
                ani = sum / (double)(hi - lo);
            }
            double afq = (double)s_acc[0] / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
            double afr = (double)s_acc[0] / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
            h.af_query = (float)afq; h.af_ref = (float)afr;
            if (ok && (afq >= R.min_af || afr >= R.min_af)) h.ani = (float)ani;
            h.ani_raw = h.ani; h.ani_std = (float)std_all;
        }
        R.hits[p] = h;
    }
}
// Pairs whose chunk table has at most 64 rows (short contigs: 1-3 chunks) - ONE WAVE per pair, a lane per chunk, shuffles instead
// of LDS and workgroup barriers; four independent pairs per workgroup. Same arithmetic and summation order as pair_reduce_pair
// (values in chunk order for the mean, ascending for median / trimmed mean).
__global__ __launch_bounds__(256) void pair_reduce_small_kernel(ReduceArgs R, uint32_t n_pairs) {
    __shared__ double s_sorted[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    for (uint32_t k = blockIdx.x * 4 + wave; k < n; k += gridDim.x * 4) {
        const uint32_t p = R.live ? R.live[k] : k;
        const uint32_t nc = R.n_chunks[p];
        if (nc == 0 && !R.live) {                         // a launch without the live list (few pairs): the empty record here, as pair_empty_kernel writes it
            if (lane == 0) {
                psk_hit h{};
                h.ani = -1.0f; h.ani_raw = -1.0f;
                h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
                h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
                R.hits[p] = h;
            }
            continue;
        }
        if (nc == 0 || nc > 64) continue;                 // empty records / larger tables: the other kernels
        if (R.tiny_done && nc <= 4) continue;             // pair_reduce_tiny_kernel took it
        const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
        ChunkOut c{};
        if ((uint32_t)lane < nc) c = co[lane];
        const bool valid = (uint32_t)lane < nc && c.n_intervals != 0;
        unsigned long long t_cq = c.cov_q, t_a = c.anchors, t_s = valid ? c.seeds : 0, t_i = c.n_intervals;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_cq += __shfl_xor(t_cq, o); t_a += __shfl_xor(t_a, o); t_s += __shfl_xor(t_s, o); t_i += __shfl_xor(t_i, o); }
        const unsigned long long vm = __ballot(valid);
        const uint32_t m = (uint32_t)__popcll(vm);
        double v = 0.0;
        if (valid) {
            double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
            if (ratio > 1.0) ratio = 1.0;
            v = pow(ratio, 1.0 / (double)R.k);
        }
        // compact the values in chunk order (position = number of valid lanes below)
        const uint32_t pos = (uint32_t)__popcll(vm & ((1ull << lane) - 1));
        double* sv = s_sorted[wave];
        lds_wave_sync();
        if (valid) sv[pos] = v;
        lds_wave_sync();
        // mean and sample standard deviation of all values
        double sum_all = 0;
        for (uint32_t j = 0; j < m; j++) sum_all += sv[j];                 // chunk order, like the serial sum of the big path
        const double mean_all = m ? sum_all / (double)m : 0.0;
        double dev = valid ? (v - mean_all) * (v - mean_all) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dev += __shfl_xor(dev, o);
        const double std_all = m > 1 ? sqrt(dev / (double)(m - 1)) : 0.0;
        double ani = mean_all;
        if ((R.median || R.robust) && m) {
            // ascending order: rank = values below + equal values at lower positions
            uint32_t rank = 0;
            const double mine = (uint32_t)lane < m ? sv[lane] : 0.0;
            for (uint32_t j = 0; j < m; j++) { const double o = sv[j]; rank += (o < mine) || (o == mine && j < (uint32_t)lane); }
            lds_wave_sync();
            if ((uint32_t)lane < m) sv[rank] = mine;
            lds_wave_sync();
            if (R.median) ani = sv[m / 2];
            else {
                uint32_t lo = 0, hi = m;
                if (m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                double sum = 0; for (uint32_t j = lo; j < hi; j++) sum += sv[j];
                ani = sum / (double)(hi - lo);
            }
        }
        if (lane == 0) {
            psk_hit h{};
            h.ani = -1.0f;
            h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
            h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
            h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
            h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
            if (m > 0) {
                double afq = (double)t_cq / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
                double afr = (double)t_cq / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
                h.af_query = (float)afq; h.af_ref = (float)afr;
                if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)ani;
                h.ani_raw = h.ani; h.ani_std = (float)std_all;
            }
            R.hits[p] = h;
        }
    }
}

// Contig pairs: one to three chunks. A wave per pair leaves 61 lanes idle for 17 M pairs per metagenome step (15 ms); here ONE LANE reduces a pair of up to four chunk
// rows (mean ANI only: median / trimmed mean stay with the wave kernel). Same values in the same order as pair_reduce_small_kernel: the mean as the sequential sum
// in chunk order, the squared deviations added the way that kernel's shuffle tree adds lanes 0..3: (d0 + d2) + (d1 + d3).
__global__ __launch_bounds__(256) void pair_reduce_tiny_kernel(ReduceArgs R, uint32_t n_pairs) {
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const uint32_t p = R.live ? R.live[k] : k;
    const uint32_t nc = R.n_chunks[p];
    if (nc == 0 || nc > 4) return;
    const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
    unsigned long long t_cq = 0, t_a = 0, t_s = 0, t_i = 0;
    double v[4] = {0.0, 0.0, 0.0, 0.0}; bool valid[4] = {false, false, false, false};
    uint32_t m = 0;
    double sum_all = 0;
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) if (r < nc) {
        const ChunkOut c = co[r];
        valid[r] = c.n_intervals != 0;
        t_cq += c.cov_q; t_a += c.anchors; t_s += valid[r] ? c.seeds : 0; t_i += c.n_intervals;
        if (valid[r]) {
            double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
            if (ratio > 1.0) ratio = 1.0;
            v[r] = pow(ratio, 1.0 / (double)R.k);
            sum_all += v[r];
            m++;
        }
    }
    const double mean_all = m ? sum_all / (double)m : 0.0;
    double d[4];
#pragma unroll
    for (int r = 0; r < 4; r++) d[r] = valid[r] ? (v[r] - mean_all) * (v[r] - mean_all) : 0.0;
    const double dev = (d[0] + d[2]) + (d[1] + d[3]);
    const double std_all = m > 1 ? sqrt(dev / (double)(m - 1)) : 0.0;
    psk_hit h{};
    h.ani = -1.0f;
    h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
    h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
    h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
    h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
    if (m > 0) {
        double afq = (double)t_cq / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
        double afr = (double)t_cq / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
        h.af_query = (float)afq; h.af_ref = (float)afr;
        if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)mean_all;
        h.ani_raw = h.ani; h.ani_std = (float)std_all;
    }
    R.hits[p] = h;
}

// Pairs whose chunk table has 65 .. 64 RW_PER rows (a pair of 5 Mb genomes: ~170-300 chunks): ONE WAVE per pair, RW_PER rows per lane, four independent pairs per
// workgroup, no workgroup barrier - the workgroup-per-pair kernel spends its time in a dozen barriers and a one-thread sum over LDS while 255 threads
// wait (33 ns per pair of a 10^6-pair batch). Same arithmetic in the same order as pair_reduce_pair: the chunk values compacted in chunk order, mean and
// deviation sums as that kernel's 256 threads form them (one value per thread, a shuffle tree per 64, the four trees added in order), the ANI mean as the
// sequential sum in chunk order - here over lane reads of registers -, median / trimmed mean over an ascending order.
__global__ __launch_bounds__(256) void pair_reduce_wave_kernel(ReduceArgs R, uint32_t n_pairs) {
    __shared__ double s_val[4][64 * RW_PER];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    double* sv = s_val[wave];
    for (uint32_t k = blockIdx.x * 4 + wave; k < n; k += gridDim.x * 4) {
        const uint32_t p = R.live ? R.live[k] : k;
        const uint32_t nc = R.n_chunks[p];
        if (nc <= 64 || nc > 64u * RW_PER) continue;      // the one-wave-one-row kernel / the workgroup kernel
        const ChunkOut* co = R.chunks + (size_t)R.cbase[p];
        unsigned long long t_cq = 0, t_a = 0, t_s = 0, t_i = 0;
        double v[RW_PER]; bool valid[RW_PER];
#pragma unroll
        for (int g = 0; g < RW_PER; g++) {
            const uint32_t r = 64u * g + (uint32_t)lane;
            ChunkOut c{};
            if (r < nc) c = co[r];
            valid[g] = r < nc && c.n_intervals != 0;
            t_cq += c.cov_q; t_a += c.anchors; t_s += valid[g] ? c.seeds : 0; t_i += c.n_intervals;
            v[g] = 0.0;
            if (valid[g]) {
                double ratio = (double)c.anchors / (double)(c.seeds > 1 ? c.seeds - 1 : 1);   // end seeds are anchors by construction
                if (ratio > 1.0) ratio = 1.0;
                v[g] = pow(ratio, 1.0 / (double)R.k);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { t_cq += __shfl_xor(t_cq, o); t_a += __shfl_xor(t_a, o); t_s += __shfl_xor(t_s, o); t_i += __shfl_xor(t_i, o); }
        // the values of the chunks that kept a chain, compacted in chunk order
        lds_wave_sync();
        uint32_t m = 0;
#pragma unroll
        for (int g = 0; g < RW_PER; g++) {
            const unsigned long long vm = __ballot(valid[g]);
            if (valid[g]) sv[m + (uint32_t)__popcll(vm & ((1ull << lane) - 1))] = v[g];
            m += (uint32_t)__popcll(vm);
        }
        lds_wave_sync();
        double cv[RW_PER];      // compacted value j sits where thread j of the workgroup kernel has it: lane j & 63 of group j >> 6
#pragma unroll
        for (int g = 0; g < RW_PER; g++) cv[g] = 64u * g + (uint32_t)lane < m ? sv[64 * g + lane] : 0.0;
        auto tree4 = [&](const double* x) {      // block_sum of pair_reduce_pair: a shuffle tree per 64 threads, the four results added in order
            // (thread t of that kernel's 256 adds elements t and t + 256 before the tree: groups g and g + 4 here)
            double t[4];
#pragma unroll
            for (int g = 0; g < 4; g++) { double y = 0.0 + x[g]; y += x[g + 4]; for (int o = 32; o > 0; o >>= 1) y += __shfl_xor(y, o); t[g] = y; }
            return t[0] + t[1] + t[2] + t[3];
        };
        const double mean_all = m ? tree4(cv) / (double)m : 0.0;
        double dv[RW_PER];
#pragma unroll
        for (int g = 0; g < RW_PER; g++) { const double d = cv[g] - mean_all; dv[g] = 64u * g + (uint32_t)lane < m ? 0.0 + d * d : 0.0; }
        const double ssq = tree4(dv);
        const double std_all = m > 1 ? sqrt(ssq / (double)(m - 1)) : 0.0;
        double ani = 0.0;
        if (m) {
            if (R.median || R.robust) {
                // ascending order: rank = values below + equal values at lower positions
                uint32_t rank[RW_PER];
#pragma unroll
                for (int g = 0; g < RW_PER; g++) rank[g] = 0;
                for (uint32_t j = 0; j < m; j++) {
                    const double o = sv[j];
#pragma unroll
                    for (int g = 0; g < RW_PER; g++) rank[g] += (o < cv[g]) || (o == cv[g] && j < 64u * g + (uint32_t)lane);
                }
                lds_wave_sync();
#pragma unroll
                for (int g = 0; g < RW_PER; g++) if (64u * g + (uint32_t)lane < m) sv[rank[g]] = cv[g];
                lds_wave_sync();
                if (R.median) ani = sv[m / 2];
                else {
                    uint32_t lo = 0, hi = m;
                    if (m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                    double sum = 0; for (uint32_t j = lo; j < hi; j++) sum += sv[j];
                    ani = sum / (double)(hi - lo);
                }
            } else {
                // the sequential sum in chunk order, over lane reads (a row without a chain contributes an exact + 0.0)
                double sum = 0;
#pragma unroll
                for (int g = 0; g < RW_PER; g++) {
                    const uint32_t lo32 = (uint32_t)__double2loint(v[g]), hi32 = (uint32_t)__double2hiint(v[g]);
                    const uint32_t cnt = nc > 64u * g ? (nc - 64u * g < 64u ? nc - 64u * g : 64u) : 0u;
                    for (uint32_t l = 0; l < cnt; l++)
                        sum += __hiloint2double((int)__builtin_amdgcn_readlane(hi32, l), (int)__builtin_amdgcn_readlane(lo32, l));
                }
                ani = sum / (double)m;
            }
        }
        if (lane == 0) {
            psk_hit h{};
            h.ani = -1.0f;
            h.ref_index = R.pair_qr[p].y; h.reserved = R.pair_qr[p].x;
            h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
            h.n_anchors = R.pcnt ? R.pcnt[p] : R.pstart[p + 1] - R.pstart[p];
            h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
            if (m > 0) {
                double afq = (double)t_cq / (double)R.pairs[p].q_total_len; if (afq > 1) afq = 1;
                double afr = (double)t_cq / (double)R.pairs[p].r_total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
                h.af_query = (float)afq; h.af_ref = (float)afr;
                if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)ani;
                h.ani_raw = h.ani; h.ani_std = (float)std_all;
            }
            R.hits[p] = h;
        }
    }
}

__global__ __launch_bounds__(256) void pair_reduce_kernel(ReduceArgs R, uint32_t n_pairs) {      // one workgroup per LIVE pair, fixed grid over the list
    const uint32_t n = R.live ? *R.n_live : n_pairs;        // small launches skip the list: every pair is visited
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        pair_reduce_pair<RED_SMALL>(R, R.live ? R.live[k] : k);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void pair_reduce_large_kernel(ReduceArgs R, uint32_t n_pairs) {      // the pairs with more than RED_SMALL chunk rows
    const uint32_t n = R.live ? *R.n_live : n_pairs;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        pair_reduce_pair<RED_CAP>(R, R.live ? R.live[k] : k);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void pair_ref_keys_kernel(const uint2* __restrict__ pair_qr, uint32_t n_pairs, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) { keys[p] = pair_qr[p].y; ids[p] = p; }
}

struct IsLivePair { const uint32_t* nch; __host__ __device__ bool operator()(const uint32_t& p) const { return nch[p] != 0; } };

// ------------------------------------------------------------------ host orchestration
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// unindexed_ok: the sketch is described with its seed count and chunk-table rows although it has no k-mer index (rounds that join through the
// database-wide seed index: no kernel of theirs reads a per-sketch index)
static SketchDesc make_desc(const psk_sketch* s, bool unindexed_ok = false) {
    SketchDesc d{};
    const bool ix = s->idx != nullptr;
    d.key = ix ? s->idx->km32 + s->idx_off : nullptr; d.pms = ix ? s->idx->pms + s->idx_off : nullptr;
    d.perm = ix ? s->idx->perm + s->idx_off : nullptr; d.bucket = ix ? s->idx->bucket + s->idx_boff : nullptr;
    d.bshift = ix ? s->idx_bshift : 0; d.n = (ix || (unindexed_ok && s->store)) ? (uint32_t)s->n_seeds : 0;
    d.pos = s->store ? s->store->seed_pos + s->seed_off : nullptr; d.meta = s->store ? s->store->seed_meta + s->seed_off : nullptr;
    d.kmer = s->store ? s->store->seed_kmer + s->seed_off : nullptr;
    d.seed_pos_base = s->store ? s->store->seed_pos : nullptr;
    d.contig_start = s->store ? s->store->contig_seed_start + s->contig_off : nullptr;
    d.total_len = s->total_len; d.n_contigs = (uint32_t)s->contig_len.size();
    uint64_t rows = 0;      // chunk heads on one contig are more than FRAGMENT_LENGTH apart
    if (d.n) for (uint32_t len : s->contig_len) rows += (uint64_t)len / (FRAGMENT_LENGTH + 1) + 1;
    d.rows = (uint32_t)std::min<uint64_t>(rows, 0xFFFFFFFFu);
    s->len_quantiles(d.lenq);
    d.tab = (ix && s->ptab) ? (const ProbeLine*)s->ptab->base + s->ptab_off : nullptr; d.tab_lines = (ix && s->ptab) ? s->ptab_lines : 0;
    return d;
}

__device__ __forceinline__ PairDesc combine_desc(const SketchDesc& Q, const SketchDesc& R) {
    PairDesc P;
    P.r_key = R.key; P.r_pms = R.pms; P.r_n = R.n; P.r_bucket = R.bucket; P.r_bshift = R.bshift; P.r_tab = R.tab; P.r_tab_lines = R.tab_lines;
    P.q_n = Q.n; P.q_key = Q.key; P.q_perm = Q.perm; P.q_pos = Q.pos; P.q_meta = Q.meta; P.q_kmer = Q.kmer; P.q_nc = Q.n_contigs; P.pad_ = 0;
    P.q_seed_pos_base = Q.seed_pos_base; P.q_contig_start = Q.contig_start;
    P.q_total_len = Q.total_len; P.r_total_len = R.total_len;
    return P;
}

// pairs from an explicit (query desc, ref desc) index list; sbase / cbase come from the host
__global__ __launch_bounds__(256) void pair_build_list_kernel(const uint2* __restrict__ qr, const SketchDesc* __restrict__ qd, const SketchDesc* __restrict__ rd,
                                                              uint32_t n_pairs, PairDesc* __restrict__ pairs) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) pairs[p] = combine_desc(qd[qr[p].x], rd[qr[p].y]);
}

// Device-side shortlist: one workgroup per batch entry walks its query's row of the pass matrix and turns the passing
// references with rank in [rank_lo, rank_hi) into pairs. Every pair of one query has the same item and row count, so the
// item / row offsets follow from the rank: no scan, no pass[] on the host (lib.rs:617-637 + 640-645 in one kernel).
// (struct BatchQ: slice_join.h)
__global__ __launch_bounds__(256) void pair_build_rows_kernel(const BatchQ* __restrict__ bq, const uint8_t* __restrict__ pass, uint32_t n_refs,
                                                              const SketchDesc* __restrict__ qd, const SketchDesc* __restrict__ rd,
                                                              PairDesc* __restrict__ pairs, uint32_t* __restrict__ sbase, uint32_t* __restrict__ cbase,
                                                              uint2* __restrict__ pair_qr, uint32_t n_pairs, uint32_t n_items, uint32_t n_rows) {
    __shared__ uint32_t s_w[4];
    const BatchQ B = bq[blockIdx.x];
    const SketchDesc Q = qd[B.q];
    const uint8_t* __restrict__ row = pass + (size_t)B.q * n_refs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t running = 0;
    for (uint32_t base = 0; base < n_refs && running < B.rank_hi; base += 256) {
        const uint32_t r = base + threadIdx.x;
        const bool f = r < n_refs && row[r];
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += s_w[w];
        const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        const uint32_t rank = running + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1));
        if (f && rank >= B.rank_lo && rank < B.rank_hi) {
            const uint32_t j = rank - B.rank_lo, slot = B.pair_off + j;
            pairs[slot] = combine_desc(Q, rd[r]);
            sbase[slot] = B.item_off + j * Q.n; cbase[slot] = B.row_off + j * Q.rows;
            pair_qr[slot] = make_uint2(B.q, r);
        }
        running += tot;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { sbase[n_pairs] = n_items; cbase[n_pairs] = n_rows; }
}

// rows of the pass matrix: per-query pass counts and per-reference "passed somewhere" flags (what the host needs to plan batches
// and to index the references that will be chained); with duplicate names, a passing entry first moves to the name's last sketch
__global__ __launch_bounds__(256) void pass_canon_kernel(uint8_t* __restrict__ pass, uint32_t n_refs, const uint32_t* __restrict__ canon) {
    uint8_t* row = pass + (size_t)blockIdx.x * n_refs;
    for (uint32_t r = threadIdx.x; r < n_refs; r += blockDim.x) if (row[r] && canon[r] != r) { row[canon[r]] = 1; row[r] = 0; }   // canon[r] > r and canon[canon[r]] == canon[r]
}
// ... and (row_blocks) the number of 2^BSI_BLOG-reference blocks that hold one of the query's passing references: what the slice join's plan looks at
static_assert((1 << BSI_BLOG) == 256, "pass_count_kernel: one sweep of its 256 threads = one block of references");
__global__ __launch_bounds__(256) void pass_count_kernel(const uint8_t* __restrict__ pass, uint32_t n_refs, uint32_t* __restrict__ row_count, uint8_t* __restrict__ col_flag, uint32_t* __restrict__ row_blocks) {
    __shared__ uint32_t s_c[4], s_any;
    const uint8_t* row = pass + (size_t)blockIdx.x * n_refs;
    uint32_t c = 0, blocks = 0;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    for (uint32_t r0 = 0; r0 < n_refs; r0 += 256) {
        const uint32_t r = r0 + threadIdx.x;
        const bool f = r < n_refs && row[r];
        if (f) { c++; col_flag[r] = 1; s_any = 1; }
        __syncthreads();
        blocks += s_any;
        __syncthreads();
        if (threadIdx.x == 0) s_any = 0;
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { row_count[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3]; row_blocks[blockIdx.x] = blocks; }
}

// ------------------------------------------------------------------ join of MANY SMALL pairs through the database-wide seed index (psk_db::gsi_*)
// The probe join visits one 64-byte table line per (pair, query seed): 5.7 G lines for 100 000 contigs against 5 000 references, 68 bytes of HBM traffic per
// 16 algorithmic ones, although the batch holds only 33 M distinct query seeds. Here ONE lookup per query seed returns the seed's matches in EVERY reference
// (a contiguous run of the index, sorted by reference, contig, position), and the wave that owns the query deals them to the query's pairs:
//   * one wave per batch entry (a query and a rank range of at most GSI_PMAX of its passing references);
//   * the query's row of the pass matrix becomes a bitset + per-word prefix counts in LDS: reference -> rank -> pair of the entry, two LDS reads;
//   * seeds are taken in (contig, position) order, a run's entries in index order, and every pair has a cursor in LDS: the anchors of a pair come out in
//     (q contig, q pos, r contig, r pos) order with no sort. A reference that holds the k-mer several times sits in consecutive lanes: ballot arithmetic gives
//     every lane its place in the group, the group's last lane moves the cursor.
// COUNT pass: the cursors' final values are the pairs' anchor counts (-> scan -> pstart). EMIT pass: the same walk writes the 16-byte anchors.
// The item records, their scan and the per-item emit of the other joins do not exist here.
// GSI_PMAX (slice_join.h) = 256      // (an entry's LDS: 4 B (count) / 20 B (emit) per pair; a query with more passing references is walked by several entries - cheap for the short contigs that have them)
struct GsiJoinArgs {
    const BatchQ* bq; const uint8_t* pass; uint32_t n_refs; const SketchDesc* qd;
    const uint32_t* g_key; const unsigned long long* g_val; const uint32_t* g_bucket; int g_shift;
    // b_blocks > 0: the index is ALSO there in blocks of 2^BSI_BLOG references with a bucket table of b_nb1 entries each (psk_db::bsi_*): a wave whose query has its passing
    // references in at most b_max of them walks those blocks (a run of the database-wide index holds ~1 % of all genomes by chance), any other wave the database-wide index
    const uint32_t* b_key; const unsigned long long* b_val; const uint32_t* b_bucket; int b_shift; uint32_t b_nb1, b_blocks, b_max;
    uint32_t* pair_cnt; const uint32_t* pstart; uint4* anc; uint32_t cap; uint32_t* err;
    uint32_t p_cap;      // most pairs any entry of the batch holds, rounded up: what the cursor arrays in LDS are sized for (<= GSI_PMAX)
    uint2* chunks; uint32_t* n_chunks;      // EMIT: the pairs' chunk tables, written by the same walk (rows at entry.row_off + slot * query rows)
    // ONE PASS (no COUNT pass, no scan): pstart holds the pairs' ITEM offsets, stretched (gsi_room_kernel: room for nine anchors per eight query seeds and
    // eight more), the walk leaves every pair's count in pair_cnt and adds the batch's total to *total; a pair that would need more room (a reference that
    // holds the query's k-mers several times over) raises err bit 2 and the batch is rerun with the two passes
    int onepass; unsigned long long* total;
    int stage;      // EMIT: anchors leave in pairs of 32 bytes (an even-indexed anchor waits in LDS for its neighbour); 0: every anchor its own 16-byte store ($PSK_GSI_STAGE=0)
};
template <bool EMIT>
__global__ __launch_bounds__(64) void gsi_join_kernel(GsiJoinArgs A) {
    extern __shared__ unsigned long long s_gsi[];
    const uint32_t nw = (A.n_refs + 63u) / 64u;
    uint4* s_line = (uint4*)s_gsi;      // EMIT with A.stage: the even-indexed anchor every pair holds back (16 B per pair, at the front: 16-byte aligned)
    unsigned long long* s_bits = s_gsi + ((EMIT && A.stage) ? 2u * A.p_cap : 0u);
    uint32_t* s_pref = (uint32_t*)(s_bits + nw);
    uint32_t* s_cur = s_pref + ((nw + 1u) & ~1u);
    uint32_t* s_ps = s_cur + A.p_cap;       // EMIT: first anchor of every pair of the entry (GSI_DEAD: fewer than MIN_ANCHORS anchors - it cannot chain: no anchors, no chunk table),
    uint32_t* s_hq = s_ps + A.p_cap;        //       query position, anchor index and (q contig << 16 | rows so far) of the chunk being filled
    uint32_t* s_hi = s_hq + A.p_cap;
    uint32_t* s_hc = s_hi + A.p_cap;
    constexpr uint32_t GSI_DEAD = 0xFFFFFFFFu;
    const int lane = threadIdx.x;
    const BatchQ B = A.bq[blockIdx.x];
    const uint32_t P = B.rank_hi - B.rank_lo;
    {   // pass row -> bitset + prefix counts
        const uint8_t* __restrict__ row = A.pass + (size_t)B.q * A.n_refs;
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < nw; w0 += 4) {
            uint8_t f[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t r = (w0 + u) * 64u + (uint32_t)lane; f[u] = r < A.n_refs ? row[r] : (uint8_t)0; }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const unsigned long long m = __ballot(f[u] != 0);
                if (w0 + u < nw && lane == 0) { s_bits[w0 + u] = m; s_pref[w0 + u] = run; }
                run += (uint32_t)__popcll(m);
            }
        }
        for (uint32_t j = lane; j < P; j += 64) { s_cur[j] = 0; if (EMIT) { const uint32_t a = A.pstart[B.pair_off + j], z = A.pstart[B.pair_off + j + 1]; s_ps[j] = !A.onepass && z - a < MIN_ANCHORS ? GSI_DEAD : a; s_hc[j] = 0; } }
    }
    lds_wave_sync();
    const SketchDesc Q = A.qd[B.q];
    const uint32_t nq = Q.n;
    // The walk is a chain of dependent round trips (k-mers -> bucket bounds -> index entries -> LDS) and a launch holds only a few waves per SIMD (one per entry):
    // everything is requested ahead. A batch of 64 seeds has its k-mers loaded two batches ahead and its bucket bounds one batch ahead; its runs are cut into
    // STEPS of 64 index entries, numbered through the batch (a prefix sum over the lanes' step counts), and the entries of step t + GSI_AHEAD are requested
    // before step t is dealt out - the second and third step of a long run (a k-mer that a whole family of references holds) included.
    constexpr uint32_t GSI_AHEAD = 4;
    unsigned long long visited = 0;      // index entries in the runs this lane's seeds found (psk_ctx_join_work)
    // (blocked index: one walk of the query's seeds per block that holds a passing reference - a pair's reference sits in ONE block, so its anchors keep their order)
    unsigned long long masks[4] = {1ull, 0ull, 0ull, 0ull};      // blocks to walk, 64 per word (the database-wide index: "block" 0)
    bool blocked = false;
    if (A.b_blocks && A.b_blocks <= 256u) {
        uint32_t n_with = 0; unsigned long long mk[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (uint32_t g = 0; g < 4; g++) {
            const uint32_t bk = g * 64u + (uint32_t)lane;
            unsigned long long any = 0;
            if (bk < A.b_blocks) {
                const uint32_t w0 = bk << (BSI_BLOG - 6), w1 = (w0 + (1u << (BSI_BLOG - 6))) < nw ? w0 + (1u << (BSI_BLOG - 6)) : nw;
                for (uint32_t w = w0; w < w1; w++) any |= s_bits[w];
            }
            mk[g] = __ballot(any != 0);
            n_with += (uint32_t)__popcll(mk[g]);
        }
        if (n_with <= A.b_max) { blocked = true; masks[0] = mk[0]; masks[1] = mk[1]; masks[2] = mk[2]; masks[3] = mk[3]; }
    }
    const uint32_t* __restrict__ x_key = blocked ? A.b_key : A.g_key; const unsigned long long* __restrict__ x_val = blocked ? A.b_val : A.g_val;
    const int x_shift = blocked ? A.b_shift : A.g_shift;
#pragma unroll 1
    for (uint32_t blk0 = 0; blk0 < 256u; blk0 += 64) {
    unsigned long long blk_mask = masks[blk0 >> 6];
    while (blk_mask) {
    const uint32_t blk = blk0 + (uint32_t)__ffsll((long long)blk_mask) - 1u;
    blk_mask &= blk_mask - 1ull;
    const uint32_t* __restrict__ bkt = blocked ? A.b_bucket + (size_t)blk * A.b_nb1 : A.g_bucket;
    uint32_t km1 = (uint32_t)lane < nq ? Q.kmer[lane] : 0u, km2 = 64u + (uint32_t)lane < nq ? Q.kmer[64 + lane] : 0u;
    uint32_t lo1 = 0, hi1 = 0;
    if ((uint32_t)lane < nq) { const uint32_t b = km1 >> x_shift; lo1 = bkt[b]; hi1 = bkt[b + 1]; }
    for (uint32_t c0 = 0; c0 < nq; c0 += 64) {
        const uint32_t i = c0 + (uint32_t)lane;
        const uint32_t km = km1, lo = lo1, hi = hi1;
        km1 = km2; lo1 = 0; hi1 = 0;
        if (i + 64u < nq) { const uint32_t b = km1 >> x_shift; lo1 = bkt[b]; hi1 = bkt[b + 1]; }
        km2 = i + 128u < nq ? Q.kmer[i + 128u] : 0u;
        uint32_t qp = 0, qm = 0;
        if (EMIT && i < nq) { qp = Q.pos[i]; qm = Q.meta[i]; }
        const uint32_t nst = (hi - lo + 63u) >> 6;
        visited += hi - lo;
        uint32_t pre = nst;      // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(pre, o); if (lane >= o) pre += y; }
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
        pre -= nst;
        uint32_t ns[GSI_AHEAD], nx[GSI_AHEAD], nh[GSI_AHEAD], nk[GSI_AHEAD]; unsigned long long nv[GSI_AHEAD];
#define GSI_FETCH(t, u) do { \
            ns[u] = 0; nx[u] = 0; nh[u] = 0; nk[u] = 0xFFFFFFFFu; nv[u] = 0ull; \
            if ((t) < T) { \
                const unsigned long long own = __ballot(nst != 0 && pre <= (t)); \
                ns[u] = 63u - (uint32_t)__clzll((long long)own); \
                nx[u] = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)ns[u]) + 64u * ((t) - (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)ns[u])); \
                nh[u] = (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)ns[u]); \
                if (nx[u] + (uint32_t)lane < nh[u]) { nk[u] = x_key[nx[u] + lane]; nv[u] = x_val[nx[u] + lane]; } \
            } } while (0)
#pragma unroll
        for (uint32_t u = 0; u < GSI_AHEAD; u++) GSI_FETCH(u, u);
        for (uint32_t t0 = 0; t0 < T; t0 += GSI_AHEAD) {
            uint32_t cs[GSI_AHEAD], cx[GSI_AHEAD], ch[GSI_AHEAD], ck[GSI_AHEAD]; unsigned long long cv[GSI_AHEAD];
#pragma unroll
            for (uint32_t u = 0; u < GSI_AHEAD; u++) { cs[u] = ns[u]; cx[u] = nx[u]; ch[u] = nh[u]; ck[u] = nk[u]; cv[u] = nv[u]; }
#pragma unroll
            for (uint32_t u = 0; u < GSI_AHEAD; u++) GSI_FETCH(t0 + GSI_AHEAD + u, u);
#pragma unroll
            for (uint32_t u = 0; u < GSI_AHEAD; u++) {
                if (t0 + u >= T) break;
                const uint32_t s = cs[u], shi = ch[u], x = cx[u] + (uint32_t)lane, k = ck[u]; const unsigned long long v = cv[u];
                const uint32_t skm = (uint32_t)__builtin_amdgcn_readlane((int)km, (int)s);
                const uint32_t sqp = EMIT ? (uint32_t)__builtin_amdgcn_readlane((int)qp, (int)s) : 0u, sqm = EMIT ? (uint32_t)__builtin_amdgcn_readlane((int)qm, (int)s) : 0u;
                const bool match = x < shi && k == skm;
                if (!__any(match)) continue;
                const uint32_t ref = (uint32_t)(v >> 48), w = ref >> 6, bpos = ref & 63u;
                uint32_t slot = 0xFFFFFFFFu;
                if (match) {
                    const unsigned long long bits = s_bits[w];
                    const uint32_t rk = s_pref[w] + (uint32_t)__popcll(bits & ((1ull << bpos) - 1ull));
                    if (((bits >> bpos) & 1ull) && rk >= B.rank_lo && rk < B.rank_hi) slot = rk - B.rank_lo;
                    if (EMIT && slot != 0xFFFFFFFFu && s_ps[slot] == GSI_DEAD) slot = 0xFFFFFFFFu;
                }
                const bool valid = slot != 0xFFFFFFFFu;
                if (!__any(valid)) continue;
                if (!EMIT) {      // the count pass needs no order: one LDS atomic per anchor
                    if (valid) atomicAdd(&s_cur[slot], 1u);
                    continue;
                }
                const uint32_t prev = __shfl_up(slot, 1);
                const bool same = valid && lane > 0 && prev == slot;                       // not the first lane of its (seed, reference) group
                // a reference that holds the k-mer ONCE (nearly always) is a group of one lane: every valid lane of the step then has a slot of its own, reads and moves
                // its cursor itself, and nothing has to be ordered between lanes (the LDS takes a wave's operations in issue order, step after step)
                const bool dup = __ballot(same) != 0;
                bool last = valid; uint32_t j = 0;
                if (dup) {
                    const uint32_t next = __shfl_down(slot, 1);
                    last = valid && !(lane < 63 && next == slot);
                    const unsigned long long starts = __ballot(valid && !same);
                    const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                    j = valid ? (uint32_t)lane - (63u - (uint32_t)__clzll((long long)upto)) : 0u;
                }
                const uint32_t base = valid ? s_cur[slot] : 0u;
                // Anchors leave in PAIRS (A.stage): a scattered 16-byte store costs 32 bytes of HBM write traffic (profiles/r4/r4k_pmc_calibration.md), so an anchor
                // with an even index waits in LDS (one 16-byte slot per pair) for its odd neighbour, and the lane that brings that one writes both: one 32-byte granule.
                // Inside a group of several lanes (a reference holding the k-mer several times) neighbours go out directly; only a group's last even anchor waits.
                bool hold = false; uint4 av = make_uint4(0, 0, 0, 0);
                if (EMIT && valid) {
                    const unsigned long long dst = (unsigned long long)s_ps[slot] + base + j;
                    if (A.onepass && base + j >= nq + (nq >> 3) + 8u) atomicOr(A.err, 4u);      // the pair's room (gsi_room_kernel) is used up
                    else if (dst < A.cap) {
                        const uint32_t rmeta = (uint32_t)((((v >> 33) & 0x7FFFull) << 1) | (v & 1ull));      // ref contig << 1 | (fwd < rc)
                        av = make_uint4(sqp, (uint32_t)(v >> 1), (rmeta & ~1u) | ((rmeta ^ sqm) & 1u), sqm >> 1);
                        const uint32_t d32 = (uint32_t)dst;
                        if (!A.stage) A.anc[dst] = av;
                        else if (d32 & 1u) {      // odd: out it goes - with its even neighbour from LDS when that one is the pair's own and is not the lane before this one
                            if (!same && d32 > s_ps[slot]) A.anc[d32 - 1u] = s_line[slot];
                            A.anc[d32] = av;
                        } else if (last) hold = true;      // even and the group's last: waits (written to LDS below, after the step's reads of the slots)
                        else A.anc[d32] = av;                // even with its odd neighbour in the next lane: both go out directly
                    } else atomicOr(A.err, 2u);
                    // chunk table: a chunk = the pair's anchors of one query contig within FRAGMENT_LENGTH of its first anchor (chunk_heads_kernel's rule), decided
                    // by the first lane of the (seed, reference) group - one group per pair and step
                    if (!same) {
                        const uint32_t idx = s_ps[slot] + base, qc = sqm >> 1, hc = s_hc[slot];
                        if (base == 0) { s_hq[slot] = sqp; s_hi[slot] = idx; s_hc[slot] = qc << 16; }
                        else if ((hc >> 16) != qc || (unsigned long long)sqp > (unsigned long long)s_hq[slot] + FRAGMENT_LENGTH) {
                            const uint32_t rows = hc & 0xFFFFu;
                            if (rows < Q.rows && rows < 0xFFFFu) A.chunks[(size_t)B.row_off + (size_t)slot * Q.rows + rows] = make_uint2(s_hi[slot], idx < A.cap ? idx : A.cap); else atomicOr(A.err, 1u);
                            s_hq[slot] = sqp; s_hi[slot] = idx; s_hc[slot] = (qc << 16) | (rows + 1u);
                        }
                    }
                }
                if (dup) lds_wave_sync();
                if (EMIT && hold) s_line[slot] = av;
                if (last) s_cur[slot] = base + j + 1u;
                if (dup) lds_wave_sync();
            }
        }
#undef GSI_FETCH
    }
    }      // the group's blocks that hold a passing reference
    }      // groups of 64 blocks
    lds_wave_sync();
    if (EMIT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
        if (lane == 0 && visited) atomicAdd((unsigned long long*)(A.err + 18), visited);
    }
    if (!EMIT) for (uint32_t j = lane; j < P; j += 64) A.pair_cnt[B.pair_off + j] = s_cur[j];
    else {
        unsigned long long sum = 0;
        for (uint32_t j = lane; j < P; j += 64) {      // the last chunk of every pair, and its row count
            uint32_t rows = 0;
            const uint32_t n = s_cur[j];
            if (A.stage && n && s_ps[j] != GSI_DEAD && !(A.onepass && n > nq + (nq >> 3) + 8u)) {      // an even last anchor is still waiting for a neighbour that never came
                const unsigned long long e = (unsigned long long)s_ps[j] + n - 1u;
                if (!(e & 1ull) && e < A.cap) A.anc[e] = s_line[j];
            }
            if (s_ps[j] != GSI_DEAD && n && !(A.onepass && n < MIN_ANCHORS)) {      // (fewer than MIN_ANCHORS anchors: no chain, no chunk table - the rows written on the way are not counted)
                rows = s_hc[j] & 0xFFFFu;
                const unsigned long long e = (unsigned long long)s_ps[j] + n;
                if (rows < Q.rows) { A.chunks[(size_t)B.row_off + (size_t)j * Q.rows + rows] = make_uint2(s_hi[j], e < A.cap ? (uint32_t)e : A.cap); rows++; } else atomicOr(A.err, 1u);
            }
            A.n_chunks[B.pair_off + j] = rows;
            if (A.onepass) { A.pair_cnt[B.pair_off + j] = n; sum += n; }
        }
        if (A.onepass) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
            if (lane == 0 && sum) atomicAdd(A.total, sum);
        }
    }
}
// ONE-PASS index join: pair p's anchors start at sbase[p] * 9 / 8 + 8 p - its (pair, query seed) items' offset, stretched: room for one anchor per query seed, an
// eighth more and eight (a contig that IS part of the reference matches with every seed, and ~1 % of a 5 Mb reference's k-mers sit in it twice)
__global__ __launch_bounds__(256) void gsi_room_kernel(const uint32_t* __restrict__ sbase, uint32_t n_pairs, uint32_t* __restrict__ pstart) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p <= n_pairs) { const uint32_t a = sbase[p]; pstart[p] = a + (a >> 3) + 8u * p; }
}
__global__ void gsi_total_kernel(const unsigned long long* __restrict__ poff, uint32_t n_pairs, unsigned long long* __restrict__ total64) { *total64 = poff[n_pairs]; }

struct HitPasses { __host__ __device__ bool operator()(const psk_hit& h) const { return h.ani > 0.1f; } };   // lib.rs:654

// measurement only (psk_ctx_set_timing): out[0] += candidate chains, out[1] += chunk-table rows that hold a chunk
__global__ __launch_bounds__(256) void work_rows_kernel(const ChunkOut* __restrict__ cout, const uint32_t* __restrict__ n_chunks, const uint32_t* __restrict__ cbase,
                                                        const uint32_t* __restrict__ row_pair, uint32_t n_rows, unsigned long long* __restrict__ out) {
    // (a fixed grid, one pair of atomics per WORKGROUP: an atomic per wave on two addresses serialised 110 000 waves - 2.5 ms per batch)
    __shared__ unsigned long long s_c[4], s_l[4];
    unsigned long long c = 0, live = 0;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        const uint32_t p = row_pair[r];
        if (r - cbase[p] < n_chunks[p]) { c += cout[r].n_cand; live++; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o); live += __shfl_xor(live, o); }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = c; s_l[threadIdx.x >> 6] = live; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c = s_c[0] + s_c[1] + s_c[2] + s_c[3]; live = s_l[0] + s_l[1] + s_l[2] + s_l[3];
        if (live) { atomicAdd(&out[0], c); atomicAdd(&out[1], live); }
    }
}

constexpr size_t CHAIN_ANCHOR_WORDS = 14;      // u32 per anchor in Lane::q_d: the 16-byte record, the successor array, the serial DP's back-pointers, the 32-byte candidate record
// device arrays of one chain launch sequence, carved from ctx->q_b
struct ChainBufs {
    PairDesc* pairs; uint32_t *sbase, *cbase, *pstart; uint2* lbcnt; uint32_t* aoff; uint32_t* nch; uint2* chunks; ChunkOut* cout;
    psk_hit* hits; psk_hit* hits_sel; uint32_t* misc; uint32_t* ovf; unsigned long long* bsum; uint2* pair_qr; BatchQ* bq;
    uint32_t *blk_pair, *row_pair, *live, *big_list, *huge_list;
    uint32_t gi, gi_sum;      // 256-item tiles; entries of bsum
    unsigned long long* total;      // the 64-bit anchor total: misc[16..17], so that status, total and the hits behind them cross in one copy
    uint32_t rows_pair_max = 0xFFFFFFFFu;      // most chunk-table rows any pair of the batch can have (the host knows its queries): which reduce kernels have work
    // join through the database-wide seed index (gsi_join_kernel): the index, the pass matrix the pairs came from and the batch's entries; g_key null: not available
    const uint32_t* g_key = nullptr; const unsigned long long* g_val = nullptr; const uint32_t* g_bucket = nullptr; int g_shift = 0;
    uint32_t g_nb1 = 0, g_blocks = 0;      // (the slice join's index comes in blocks of references: psk_db::bsi_*)
    const uint32_t* b_key = nullptr; const unsigned long long* b_val = nullptr; const uint32_t* b_bucket = nullptr; int b_shift = 0; uint32_t b_nb1 = 0, b_blocks = 0, b_max = 0;      // the contig join: the blocked index beside the database-wide one
    const uint8_t* d_pass = nullptr; uint32_t n_refs = 0, n_bq = 0, p_cap = 0;
    // mid-sized pairs (all-vs-all of genomes): the index join by (query, slice) waves (slice_join.hip); the batch's wave table, record offsets and per-record arrays
    bool gsi_slice = false; const uint2* gsl_tab = nullptr; uint32_t gsl_n_tab = 0; const uint2* gsl_ebase = nullptr; uint32_t *gsl_cnt = nullptr, *gsl_bm = nullptr, *gsl_un = nullptr, gsl_n_slices = 0; uint4* gsl_rec = nullptr;
    bool gsi_onepass = false;      // the index join without its COUNT pass (GsiJoinArgs::onepass): asked for by the caller, which reruns the batch without it when err bit 2 comes back
};
static psk_status chain_layout(Lane* ctx, size_t n_pairs, size_t n_items, size_t n_rows, size_t n_bq, ChainBufs* L) {
    const size_t gi = (n_items + 255) / 256, gi_sum = std::max(gi, (n_pairs + 3) / 4);
    size_t o_pairs = 0, o_sbase = al256(o_pairs + sizeof(PairDesc) * n_pairs), o_cbase = al256(o_sbase + 4 * (n_pairs + 1)),
           o_pstart = al256(o_cbase + 4 * (n_pairs + 1)), o_lb = al256(o_pstart + 4 * (n_pairs + 1)),
           o_aoff = al256(o_lb + 8 * (n_items + 1)), o_nch = al256(o_aoff + 4 * (n_items + 1)),
           o_chunks = al256(o_nch + 4 * n_pairs), o_cout = al256(o_chunks + sizeof(uint2) * n_rows),
           o_misc = al256(o_cout + sizeof(ChunkOut) * n_rows), o_hits = o_misc + 256, o_sel = al256(o_hits + sizeof(psk_hit) * n_pairs),      // (misc | hits: one copy takes both)
           o_ovf = al256(o_sel + sizeof(psk_hit) * n_pairs), o_bsum = al256(o_ovf + 4 * n_rows),
           o_qr = al256(o_bsum + 8 * (gi_sum + 1)), o_bq = al256(o_qr + 8 * n_pairs), o_bp = al256(o_bq + sizeof(BatchQ) * (n_bq + 1)),
           o_rp = al256(o_bp + 4 * (gi + 1)), o_live = al256(o_rp + 4 * (n_rows + 1)), o_big = al256(o_live + 4 * (n_pairs + 1)),
           o_huge = al256(o_big + 4 * (n_pairs + 1)), o_end = o_huge + 4 * (n_pairs + 1);
    PSK_TRY(ctx->q_b.reserve(o_end));
    char* B = (char*)ctx->q_b.p;
    L->pairs = (PairDesc*)(B + o_pairs); L->sbase = (uint32_t*)(B + o_sbase); L->cbase = (uint32_t*)(B + o_cbase); L->pstart = (uint32_t*)(B + o_pstart);
    L->lbcnt = (uint2*)(B + o_lb); L->aoff = (uint32_t*)(B + o_aoff); L->nch = (uint32_t*)(B + o_nch); L->chunks = (uint2*)(B + o_chunks);
    L->cout = (ChunkOut*)(B + o_cout); L->hits = (psk_hit*)(B + o_hits); L->hits_sel = (psk_hit*)(B + o_sel); L->misc = (uint32_t*)(B + o_misc);
    L->ovf = (uint32_t*)(B + o_ovf); L->bsum = (unsigned long long*)(B + o_bsum); L->pair_qr = (uint2*)(B + o_qr); L->bq = (BatchQ*)(B + o_bq);
    L->blk_pair = (uint32_t*)(B + o_bp); L->row_pair = (uint32_t*)(B + o_rp); L->live = (uint32_t*)(B + o_live); L->big_list = (uint32_t*)(B + o_big); L->huge_list = (uint32_t*)(B + o_huge);
    L->gi = (uint32_t)gi; L->gi_sum = (uint32_t)gi_sum;
    L->total = (unsigned long long*)(L->misc + 16);
    return PSK_OK;
}

// Everything between "pairs / sbase / cbase are on the device" and "hits are on the device": no host synchronisation.
// Anchor arrays are sized optimistically (cap anchors); the 64-bit anchor total travels back with the hits and the caller
// reruns the batch with a larger capacity if it did not fit (emit and every later kernel stay inside cap).
static psk_status chain_run(Lane* ctx, const ChainBufs& L, uint32_t n_pairs, size_t n_items, size_t n_rows, const psk_params& prm,
                            const psk_query_opts* o, const SketchDesc* d_qd, const SketchDesc* d_rd, uint64_t cap, bool wide, bool probe_ok = false) {
    hipStream_t st = ctx->stream;
    const int force_serial = getenv("PSK_CHAIN_SERIAL") != nullptr;
    // misc[0..15] status / counts, misc[11] pairs for select_huge_kernel, misc[16..17] the 64-bit anchor total, misc[32..47] the group barriers: zeroed by pair_table_kernel
    const uint32_t gi = L.gi;
    hipLaunchKernelGGL(pair_table_kernel, dim3((uint32_t)(((size_t)gi + n_rows + 255) / 256)), dim3(256), 0, st, L.sbase, L.cbase, n_pairs, gi, (uint32_t)n_items, (uint32_t)n_rows, L.blk_pair, L.row_pair,
                       L.misc, L.lbcnt + n_items);
    ctx->t_begin(K_ANCHOR);
    if (wide) hipLaunchKernelGGL(anchor_count_kernel, dim3(gi), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, L.lbcnt, L.bsum, L.blk_pair);
    const uint32_t gi4 = (gi + JT - 1) / JT;
    uint32_t n_sum = gi;
    const char* jp_env = getenv("PSK_JOIN_PAIRS");      // "1" / "0" force / forbid the pair-major join (tests, A/B)
    const bool gsi_join = !wide && L.g_key && L.d_pass && L.n_bq && L.n_refs <= 65536u;      // (the PLAN decides - PSK_GSI_JOIN is read there, once per round: the round's sketches carry no k-mer index to fall back on)      // (every batch of a round that was planned for it: its sketches carry no k-mer index)
    const bool join_pairs = !wide && (gsi_join || (jp_env ? jp_env[0] == '1' : (n_pairs >= 16384 && n_items / n_pairs < 2048)));
    const bool gsl = gsi_join && L.gsi_slice;      // one wave per (query, slice of its seeds): count walk -> scan over the pairs -> heads -> emit walk
    const bool gsi_one = gsi_join && !gsl && L.gsi_onepass && cap >= n_items + n_items / 8 + 8 * ((size_t)n_pairs + 1);      // (gsi_room_kernel's layout fits)
    bool probe_local = false;
    GsiJoinArgs GA{};
    GslArgs GL{};
    const size_t gsi_lds_row = 8 * (size_t)((L.n_refs + 63) / 64) + 4 * (size_t)((((L.n_refs + 63) / 64) + 1) & ~1u), gsi_lds_count = gsi_lds_row + 4 * (size_t)L.p_cap, gsi_lds_emit = gsi_lds_row + 4 * (size_t)L.p_cap * 5;
    if (gsi_join) {
        GA.bq = L.bq; GA.pass = L.d_pass; GA.n_refs = L.n_refs; GA.qd = d_qd; GA.g_key = L.g_key; GA.g_val = L.g_val; GA.g_bucket = L.g_bucket; GA.g_shift = L.g_shift;
        GA.b_key = L.b_key; GA.b_val = L.b_val; GA.b_bucket = L.b_bucket; GA.b_shift = L.b_shift; GA.b_nb1 = L.b_nb1; GA.b_blocks = L.b_blocks; GA.b_max = L.b_max;
        GA.pair_cnt = L.big_list; GA.pstart = L.pstart; GA.cap = (uint32_t)cap; GA.err = L.misc; GA.p_cap = L.p_cap;
        if (gsl) {
            GL.bq = L.bq; GL.n_entries = L.n_bq; GL.tab = L.gsl_tab; GL.n_tab = L.gsl_n_tab; GL.ebase = L.gsl_ebase; GL.pass = L.d_pass; GL.n_refs = L.n_refs; GL.qd = d_qd;
            GL.g_key = L.g_key; GL.g_val = L.g_val; GL.g_bucket = L.g_bucket; GL.g_shift = L.g_shift; GL.g_nb1 = L.g_nb1; GL.g_blocks = L.g_blocks; GL.cnt = L.gsl_cnt; GL.rec = L.gsl_rec; GL.bm = L.gsl_bm; GL.un = L.gsl_un; GL.n_slices = L.gsl_n_slices;
            GL.pair_cnt = L.big_list; GL.pstart = L.pstart; GL.cap = (uint32_t)cap; GL.err = L.misc; GL.p_cap = L.p_cap; GL.chunks = L.chunks; GL.n_chunks = L.nch;
            { const char* e = getenv("PSK_GSL_STAGE"); GL.stage = e ? atoi(e) : 1; }      // (A/B: every anchor its own 16-byte store)
            PSK_HIP(hipMemsetAsync(L.big_list, 0, 4 * ((size_t)n_pairs + 1), st));      // the slices of a pair add their counts
            PSK_TRY(gsl_count_launch(GL, st));
        }
        else if (!gsi_one) hipLaunchKernelGGL(gsi_join_kernel<false>, dim3(L.n_bq), dim3(64), gsi_lds_count, st, GA);
        probe_local = true;      // (the scan over the pairs' counts below is the probe join's)
    }
    else if (join_pairs) {
        // pair ids sorted by reference index (a pair's reference = pair_qr[p].y): one radix sort of n_pairs small keys
        size_t ts = 0;
        uint32_t* keys_in = L.big_list;                     // free until select runs
        uint32_t* vals_in = L.live;                         // free until the live list is built
        uint32_t* keys_out = (uint32_t*)L.hits_sel;         // free until the hits are selected
        uint32_t* order = keys_out + n_pairs;
        hipLaunchKernelGGL(pair_ref_keys_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, L.pair_qr, n_pairs, keys_in, vals_in);
        PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, ts, keys_in, keys_out, vals_in, order, (int)n_pairs, 0, 32, st));
        PSK_TRY(ctx->q_g.reserve(ts + 256));
        PSK_HIP(hipcub::DeviceRadixSort::SortPairs(ctx->q_g.p, ts, keys_in, keys_out, vals_in, order, (int)n_pairs, 0, 32, st));
        const uint32_t nb = (n_pairs + 3) / 4;
        // every reference of the batch carries a probe table (ensure_probe): one line read per lookup instead of the index's chain of reads
        // (the probe join leaves offsets within the pair + pair totals: PSK_PROBE_LOCAL=0 keeps the scan over all items; tests, A/B)
        static const bool pl_off = getenv("PSK_PROBE_LOCAL") && getenv("PSK_PROBE_LOCAL")[0] == '0';
        probe_local = probe_ok && !pl_off;
        if (probe_ok) hipLaunchKernelGGL(anchor_join_probe_kernel, dim3(nb), dim3(256), 0, st, L.pairs, L.sbase, order, n_pairs, L.lbcnt, L.bsum, L.misc + 5,
                                         probe_local ? L.aoff : (uint32_t*)nullptr, probe_local ? L.big_list : (uint32_t*)nullptr);      // (big_list: free until select runs)
        else hipLaunchKernelGGL(anchor_join_pairs_kernel, dim3(nb), dim3(256), 0, st, L.pairs, L.sbase, order, n_pairs, L.lbcnt, L.bsum, L.misc + 5);
        n_sum = nb;
    }
    // many mid-sized pairs (all-vs-all): the join counts every pair's anchors, one workgroup per pair then emits with a running offset
    // (anchor_emit_pairs_kernel) instead of a scan over all items; PSK_EMIT_PAIRS=1 / 0 force / forbid it (tests, A/B)
    const char* ep_env = getenv("PSK_EMIT_PAIRS");
    const bool emit_pairs = !wide && !join_pairs && n_items >= 2 * ((size_t)n_pairs + 1) &&      // (its 64-bit pair offsets live in the per-item offsets array)
                            (ep_env ? ep_env[0] == '1' : (n_pairs >= 1024 && n_items / n_pairs >= 1024 && n_items / n_pairs <= (1u << 17)));
    uint32_t* pair_cnt = L.live;      // free until the live list is built
    if (emit_pairs) PSK_HIP(hipMemsetAsync(pair_cnt, 0, 4 * ((size_t)n_pairs + 1), st));
    // workgroups of one pair per XCD turn (0 = contiguous eighths of the grid; PSK_XCD_GROUP overrides): see xcd_group_block_id
    static const int xg_env = getenv("PSK_XCD_GROUP") ? atoi(getenv("PSK_XCD_GROUP")) : -1;
    const uint32_t xcd_group = xg_env >= 0 ? (uint32_t)xg_env : (n_pairs >= 64 ? (uint32_t)std::min<size_t>(4096, std::max<size_t>(1, 4 * (n_items / n_pairs) / (JT * 256))) : 0u);      // four pairs per turn (measured: 1 pair 38.5, 2: 37.4, 4 and more: 36.8 ms of join per 10^5 pairs; contiguous eighths: 44.0)
    if (!wide && !join_pairs) { hipLaunchKernelGGL(anchor_join4_kernel, dim3(gi4), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, gi, L.lbcnt, L.bsum, L.misc + 5, L.blk_pair, emit_pairs ? pair_cnt : (uint32_t*)nullptr, xcd_group); n_sum = gi4; }
    ctx->t_end();
    size_t tmp = 0, tmp2 = 0;
    hipcub::TransformInputIterator<uint32_t, CountOf, const uint2*> cnt_it(L.lbcnt, CountOf());
    hipcub::TransformInputIterator<uint32_t, PackedCount, const uint2*> pcnt_it(L.lbcnt, PackedCount());
    PSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, cnt_it, L.aoff, (int)(n_items + 1), st));
    PSK_HIP(hipcub::DeviceReduce::Sum(nullptr, tmp2, L.bsum, L.total, (int)L.gi_sum, st));
    size_t tmp3 = 0;
    PSK_HIP(hipcub::DeviceSelect::If(nullptr, tmp3, L.hits, L.hits_sel, L.misc + 12, (int)n_pairs, HitPasses(), st));
    PSK_TRY(ctx->q_c.reserve(std::max(tmp, std::max(tmp2, tmp3))));
    unsigned long long* poff = (unsigned long long*)L.aoff;      // emit_pairs: 64-bit prefix of the pairs' counts (the per-item offsets array is not used then)
    hipcub::TransformInputIterator<unsigned long long, Widen, const uint32_t*> pc_it(pair_cnt, Widen());
    if (gsi_one) {      // the pairs' anchors start where their items do; the emit walk counts
        hipLaunchKernelGGL(gsi_room_kernel, dim3((n_pairs + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t*)L.sbase, n_pairs, L.pstart);
        PSK_HIP(hipMemsetAsync(L.total, 0, 8, st));
    }
    else if (emit_pairs) {
        size_t tmp4 = 0;
        PSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp4, pc_it, poff, (int)(n_pairs + 1), st));
        PSK_TRY(ctx->q_c.reserve(std::max(tmp4, std::max(tmp, std::max(tmp2, tmp3)))));
        PSK_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->q_c.p, tmp4, pc_it, poff, (int)(n_pairs + 1), st));
        hipLaunchKernelGGL(pair_start64_kernel, dim3((n_pairs + 1 + 255) / 256), dim3(256), 0, st, poff, n_pairs, L.pstart, (uint32_t)cap, (const uint32_t*)(L.misc + 5));
    }
    else if (probe_local) {      // the pairs' totals (anchor_join_probe_kernel) -> 64-bit prefix -> pstart; the items carry their offsets within the pair
        hipcub::TransformInputIterator<unsigned long long, Widen, const uint32_t*> pl_it(L.big_list, Widen());
        size_t tmp5 = 0;
        PSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp5, pl_it, (unsigned long long*)nullptr, (int)(n_pairs + 1), st));
        const size_t o_poff = al256(tmp5 + 256);
        PSK_TRY(ctx->q_g.reserve(o_poff + 8 * ((size_t)n_pairs + 2) + 256));
        unsigned long long* pl_off64 = (unsigned long long*)((char*)ctx->q_g.p + o_poff);
        PSK_HIP(hipMemsetAsync(L.big_list + n_pairs, 0, 4, st));      // the scan reads n_pairs + 1 counts
        PSK_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->q_g.p, tmp5, pl_it, pl_off64, (int)(n_pairs + 1), st));
        hipLaunchKernelGGL(pair_start64_kernel, dim3((n_pairs + 1 + 255) / 256), dim3(256), 0, st, (const unsigned long long*)pl_off64, n_pairs, L.pstart, (uint32_t)cap, (const uint32_t*)(L.misc + 5));
        if (gsi_join) hipLaunchKernelGGL(gsi_total_kernel, dim3(1), dim3(1), 0, st, (const unsigned long long*)pl_off64, n_pairs, L.total);
    }
    else if (wide) PSK_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->q_c.p, tmp, cnt_it, L.aoff, (int)(n_items + 1), st));
    else PSK_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->q_c.p, tmp, pcnt_it, L.aoff, (int)(n_items + 1), st));
    const bool small_sum = !gsi_join && n_sum <= 16384;
    if (!small_sum && !gsi_join) PSK_HIP(hipcub::DeviceReduce::Sum(ctx->q_c.p, tmp2, L.bsum, L.total, (int)n_sum, st));      // 64-bit total, beside the 32-bit offsets
    if ((!emit_pairs && !probe_local) || small_sum)
        hipLaunchKernelGGL(pair_start_kernel, dim3((emit_pairs || probe_local) ? 1u : (n_pairs + 1 + 255) / 256), dim3(256), 0, st, (emit_pairs || probe_local) ? (const uint32_t*)nullptr : L.aoff, L.sbase, n_pairs, L.pstart, (uint32_t)cap,
                           L.bsum, small_sum ? n_sum : 0u, L.total, (const uint32_t*)(L.misc + 5));
    hipLaunchKernelGGL(pair_guard_kernel, dim3((n_pairs + 1 + 255) / 256), dim3(256), 0, st, (const uint32_t*)(L.misc + 5), (const unsigned long long*)L.total, (unsigned long long)cap, L.pstart, n_pairs);
    // ---- anchors + candidates: 12 arrays of u32 per anchor (CHAIN_ANCHOR_WORDS); the lane-serial DP's four per-anchor arrays - a fallback that runs inside the DP
    // kernels - borrow the selection's scratch, which nothing touches before the DP is through (they had four arrays of their own: 13 of the 67 GB a batch of
    // 3 Gb pairs asks for, and a cold pass pays ~25 ms per GB it is handed)
    const size_t na = ((size_t)cap + 64 + 63) & ~(size_t)63;     // multiple of 64: every per-anchor array stays 256-byte aligned (16-byte loads in the lane kernels)
    PSK_TRY(ctx->q_d.reserve(4 * na * CHAIN_ANCHOR_WORDS));
    PSK_TRY(ctx->q_e.reserve(na * (8 + 4 * 7 + 1) + 512 + 4 * 2 * BIG_GMAX * (BIG_GROUPS + 64)));   // select_big_kernel / select_huge_kernel scratch
    uint32_t* D = (uint32_t*)ctx->q_d.p;
    uint32_t* E4 = (uint32_t*)ctx->q_e.p;   // (37 bytes per anchor: room for the serial DP's 16)
    uint4* anc = (uint4*)D;                 // the first four u32 arrays' worth of space: one 16-byte record per anchor
    uint32_t* a_nxt = D + 4 * na;
    ChainArgs A{};
    A.anc = anc;
    A.sc_ptr = D + 5 * na;
    A.sc_f = (int32_t*)E4; A.sc_root = E4 + na; A.sc_depth = E4 + 2 * na; A.sc_best = E4 + 3 * na;
    uint32_t* CAND = D + 6 * na;      // 8 words per anchor slot: one 32-byte record per candidate chain
    A.c_score.p = (int32_t*)CAND; A.c_q0.p = CAND + 1; A.c_q1.p = CAND + 2; A.c_r0.p = CAND + 3; A.c_r1.p = CAND + 4; A.c_n.p = CAND + 5; A.c_rc.p = CAND + 6; A.c_state.p = CAND + 7;
    A.chunks = L.chunks; A.n_chunks = L.nch; A.cbase = L.cbase; A.n_pairs = n_pairs; A.n_rows = (uint32_t)n_rows;
    A.row_pair = L.row_pair;
    A.pairs = L.pairs;
    A.out = L.cout; A.two_c = 2u * (uint32_t)prm.c; A.force_serial = force_serial; A.stats = L.misc + 1;
    A.band = std::max(1, std::min(MAX_CHAIN_BAND, BP_CHAIN_BAND / (int)prm.c));
    A.cap = (uint32_t)cap;
    { const char* e = getenv("PSK_DP_PRUNE"); A.dp_prune = e && e[0] == '0' ? 0 : 1; }      // (read per call: tests switch it within a process)
    // the per-pair emit also writes the chunk table unless the pointer-chase builder is asked for (PSK_CHUNK_HOPS) or PSK_EMIT_HEADS=0
    const char* hops_env = getenv("PSK_CHUNK_HOPS");
    const bool use_hops = gsi_join ? false : hops_env ? hops_env[0] != '0' : ((n_pairs < 1024 && n_items / n_pairs > 4096) || n_items / n_pairs > (1u << 20));      // (few pairs of a contig's few hundred seeds: one wave per pair walks its heads - one launch instead of two)
    static const bool emit_heads_off = getenv("PSK_EMIT_HEADS") && getenv("PSK_EMIT_HEADS")[0] == '0';
    const bool emit_heads = emit_pairs && !use_hops && !emit_heads_off;
    // Gb-scale pairs: the walk in ITEM space where the join left per-item offsets (chunk_hops_items_kernel); PSK_HOPS_ITEMS=1 / 0 force / forbid (tests, A/B)
    const char* hi_env = getenv("PSK_HOPS_ITEMS");
    const bool hops_items = use_hops && !gsl && !emit_pairs && !join_pairs && !getenv("PSK_HOPS_UNSLICED") && n_items <= 0x7FFFFFFFull &&
                            (hi_env ? hi_env[0] == '1' : n_items / n_pairs > (1u << 20));
    ctx->t_begin(K_ANCHOR_EMIT);      // anchors out of the join's records + the chunk table
    if (hops_items) {
        // ... on the lane's SIDE stream, beside the emit: the walk is a few hundred waves each waiting on its own chain of LDS round trips (24 contigs x 11 pairs: 4 ms
        // per batch with three quarters of the chip idle) and reads only the items' offsets and the query's positions; the emit fills the chip's memory pipes
        const size_t o_scr = al256(4 * (size_t)n_pairs * HOP_SLICES + 256);
        PSK_TRY(ctx->q_g.reserve(o_scr + sizeof(uint2) * n_rows + 256));
        uint32_t* slice_cnt = (uint32_t*)ctx->q_g.p;
        uint2* scratch_rows = (uint2*)((char*)ctx->q_g.p + o_scr);
        hipStream_t sd = nullptr;
        PSK_TRY(ctx->side_lane(&sd));
        PSK_HIP(hipEventRecord(ctx->side_fork, st));
        PSK_HIP(hipStreamWaitEvent(sd, ctx->side_fork, 0));
        hipLaunchKernelGGL(chunk_hops_items_kernel, dim3(n_pairs, HOP_SLICES), dim3(64), 0, sd, L.pstart, (const uint32_t*)L.aoff, L.pairs, L.sbase, L.cbase, n_pairs, slice_cnt, scratch_rows, L.misc);
        hipLaunchKernelGGL(chunk_hops_sliced_kernel, dim3(n_pairs, HOP_SLICES), dim3(64), 0, sd, L.pstart, a_nxt, anc, L.pairs, L.cbase, n_pairs, slice_cnt, 1, scratch_rows, L.chunks, L.nch, L.misc);
        PSK_HIP(hipEventRecord(ctx->side_join, sd));
    }
    if (gsl) { GL.anc = anc; PSK_TRY(gsl_heads_launch(GL, st)); PSK_TRY(gsl_emit_launch(GL, st)); }
    else if (gsi_join) { GA.anc = anc; GA.chunks = L.chunks; GA.n_chunks = L.nch; GA.onepass = gsi_one ? 1 : 0; GA.total = L.total; if (gsi_one) GA.pair_cnt = L.aoff;      /* (the per-item offsets array: not used by this join) */
                    { const char* e = getenv("PSK_GSI_STAGE"); GA.stage = e && e[0] == '0' ? 0 : 1; }      // (read per batch: tests switch it within a process)
                    hipLaunchKernelGGL(gsi_join_kernel<true>, dim3(L.n_bq), dim3(64), gsi_lds_emit + (GA.stage ? 16 * (size_t)L.p_cap : 0), st, GA); }
    else if (wide) hipLaunchKernelGGL(anchor_emit_kernel, dim3(gi), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, L.lbcnt, L.aoff, anc, (uint32_t)cap, L.misc, L.blk_pair);
    else if (emit_pairs) hipLaunchKernelGGL(anchor_emit_pairs_kernel, dim3(n_pairs), dim3(EP_T), 0, st, L.pairs, L.sbase, n_pairs, L.lbcnt, poff, anc, (uint32_t)cap, L.misc,
                                            L.cbase, emit_heads ? L.chunks : (uint2*)nullptr, L.nch);
    else {
        // k-mers with many matches (Gb-scale pairs): anchor-major emit; PSK_EMIT_EXPAND=1 / 0 force / forbid (tests, A/B)
        const char* ex_env = getenv("PSK_EMIT_EXPAND");
        const bool expand = ex_env ? ex_env[0] == '1' : n_items / n_pairs > (1u << 20);
        if (expand) hipLaunchKernelGGL(anchor_emit_expand_kernel, dim3(gi), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, L.lbcnt, L.aoff, anc, (uint32_t)cap, L.misc, L.blk_pair);
        else hipLaunchKernelGGL(anchor_emit_packed4_kernel, dim3(gi4), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, gi, L.lbcnt, L.aoff, anc, (uint32_t)cap, L.misc, L.blk_pair,
                                probe_local ? (const uint32_t*)L.pstart : (const uint32_t*)nullptr);
    }
    // few pairs (one wave each cannot fill the chip) or huge ones: nxt[] for every anchor in parallel + pointer chase
    if (hops_items) PSK_HIP(hipStreamWaitEvent(st, ctx->side_join, 0));      // (the chunk table of Gb-scale pairs was built beside the emit: above)
    else if (use_hops) {
        if (n_items / n_pairs > (1u << 20)) {      // Gb-scale: every 64th anchor first (into the spare per-anchor array after a_nxt), then all of them between those
            uint32_t* coarse = E4;                  // sc_f's space: the serial path is not running yet
            hipLaunchKernelGGL(anchor_next_kernel<1>, dim3((uint32_t)((cap / 64 + 1 + 255) / 256)), dim3(256), 0, st, anc, L.pstart, n_pairs, (const uint32_t*)nullptr, coarse);
            hipLaunchKernelGGL(anchor_next_kernel<2>, dim3((uint32_t)((cap + 255) / 256)), dim3(256), 0, st, anc, L.pstart, n_pairs, (const uint32_t*)coarse, a_nxt);
        } else
        hipLaunchKernelGGL(anchor_next_kernel<0>, dim3((uint32_t)((cap + 255) / 256)), dim3(256), 0, st, anc, L.pstart, n_pairs, (const uint32_t*)nullptr, a_nxt);
        if (n_items / n_pairs > (1u << 20) && !getenv("PSK_HOPS_UNSLICED")) {      // Gb-scale pairs: HOP_SLICES waves per pair, count then write
            const size_t o_scr = al256(4 * (size_t)n_pairs * HOP_SLICES + 256);
            PSK_TRY(ctx->q_g.reserve(o_scr + sizeof(uint2) * n_rows + 256));
            uint32_t* slice_cnt = (uint32_t*)ctx->q_g.p;
            uint2* scratch_rows = (uint2*)((char*)ctx->q_g.p + o_scr);
            for (int pass = 0; pass < 2; pass++)
                hipLaunchKernelGGL(chunk_hops_sliced_kernel, dim3(n_pairs, HOP_SLICES), dim3(64), 0, st, L.pstart, a_nxt, anc, L.pairs, L.cbase, n_pairs, slice_cnt, pass, scratch_rows, L.chunks, L.nch, L.misc);
        } else
        hipLaunchKernelGGL(chunk_hops_kernel, dim3(n_pairs), dim3(64), 0, st, L.pstart, a_nxt, L.cbase, n_pairs, L.chunks, L.nch, L.misc);
    } else if (!emit_heads && !gsi_join)
        hipLaunchKernelGGL(chunk_heads_kernel, dim3(n_pairs), dim3(64), 0, st, L.pstart, anc, L.cbase, n_pairs, L.chunks, L.nch, L.misc);
    ctx->t_end();
    ctx->t_begin(K_CHAIN_CHUNK);
    // rows by chunk length for the DP kernels that put several chunks in one wave (row_len_kernel; PSK_ROW_SORT=0: table order)
    auto order_rows = [&]() -> psk_status {
        static const bool rs_off = getenv("PSK_ROW_SORT") && getenv("PSK_ROW_SORT")[0] == '0';
        if (rs_off || n_rows < 4096) return PSK_OK;
        size_t ts = 0;
        PSK_HIP(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, ts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n_rows, 0, 8, st));
        const size_t ob = al256(4 * n_rows);
        PSK_TRY(ctx->q_g.reserve(4 * ob + ts + 256));
        uint32_t* k_in = (uint32_t*)ctx->q_g.p; uint32_t* v_in = (uint32_t*)((char*)ctx->q_g.p + ob); uint32_t* k_out = (uint32_t*)((char*)ctx->q_g.p + 2 * ob); uint32_t* v_out = (uint32_t*)((char*)ctx->q_g.p + 3 * ob);
        hipLaunchKernelGGL(row_len_kernel, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, L.chunks, L.nch, L.cbase, L.row_pair, (uint32_t)n_rows, k_in, v_in);
        PSK_HIP(hipcub::DeviceRadixSort::SortPairsDescending((char*)ctx->q_g.p + 4 * ob, ts, (const uint32_t*)k_in, k_out, (const uint32_t*)v_in, v_out, (int)n_rows, 0, 8, st));
        A.row_order = v_out;
        return PSK_OK;
    };
    {   // lane-per-chunk DP when the band fits its register window (PSK_CHAIN_LANE=0 keeps the wave-per-chunk DP)
        const char* le = getenv("PSK_CHAIN_LANE");
        A.lane_dp = !force_serial && A.band <= LANE_N && !(le && le[0] == '0');
        if (A.lane_dp) {
            PSK_TRY(order_rows());
            // few rows: spread them over more waves (idle lanes cost nothing on an under-filled chip)
            uint32_t rpw = 64;
            while (rpw > 16 && n_rows / rpw < 512) rpw >>= 1;
            if (le && atoi(le) >= 8) rpw = (uint32_t)std::min(64, atoi(le));
            const uint32_t waves = (uint32_t)((n_rows + rpw - 1) / rpw);
            A.ovf_list = L.ovf; A.ovf_count = L.misc + 8;      // misc was zeroed above
            const bool quad = le && le[0] == 'q' ? true : (le && atoi(le) >= 8 ? false : n_rows < 32 * 1024);
            if (quad) {   // small launch: four lanes per chunk, 16 chunks per wave
                const uint32_t qw = (uint32_t)((n_rows + 15) / 16);
                hipLaunchKernelGGL(chain_quad_kernel, dim3((qw + LANE_WAVES - 1) / LANE_WAVES), dim3(64 * LANE_WAVES), 0, st, A);
            } else {
                // Gb-scale pairs: sixteen tree slots per chunk (twelve of them in LDS); PSK_LANE_XTREES=1 / 0 force / forbid (tests, A/B)
                const char* xt_env = getenv("PSK_LANE_XTREES");
                const bool xtrees = xt_env ? xt_env[0] == '1' : n_items / n_pairs > (1u << 20);
                if (A.band <= 20 && xtrees) hipLaunchKernelGGL(chain_lane20x_kernel, dim3((waves + LANE_WAVES - 1) / LANE_WAVES), dim3(64 * LANE_WAVES), 0, st, A, rpw);
                else if (A.band <= 20) hipLaunchKernelGGL(chain_lane20_kernel, dim3((waves + LANE_WAVES - 1) / LANE_WAVES), dim3(64 * LANE_WAVES), 0, st, A, rpw);
                else hipLaunchKernelGGL(chain_lane_kernel, dim3((waves + LANE_WAVES - 1) / LANE_WAVES), dim3(64 * LANE_WAVES), 0, st, A, rpw);
            }
            // the few chunks it passes on (more than LANE_TREES trees, >= 16 384 anchors): wave kernel over the list
            const uint32_t lw = (uint32_t)std::min<size_t>((n_rows + CHAIN_WAVES - 1) / CHAIN_WAVES, 2048);
            hipLaunchKernelGGL(chain_chunk_list_kernel, dim3(lw), dim3(64 * CHAIN_WAVES), 0, st, A);
        }
    }
    // bands beyond the lane kernel's window (c < 105; metagenome mode c = 30: 83): four lanes per chunk with 21-deep windows, its leftovers to
    // the wave-per-chunk kernel's list form; PSK_CHAIN_QUAD_DEEP=0 keeps the wave-per-chunk kernel for every chunk (tests, A/B)
    static const bool qd_off = getenv("PSK_CHAIN_QUAD_DEEP") && getenv("PSK_CHAIN_QUAD_DEEP")[0] == '0';
    // a launch of few rows is as slow as its longest chunk: one wave per row with the window in registers (PSK_CHAIN_WAVE_REG=1 / 0 force / forbid: tests, A/B)
    const char* wr_env = getenv("PSK_CHAIN_WAVE_REG");
    const bool wave_reg = !A.lane_dp && !force_serial && A.band < 128 && (wr_env ? wr_env[0] == '1' : n_rows <= 2048);      // (one wave per SIMD up to 1 024 rows: 0.29 us per anchor of the longest chunk; the four-lanes-per-chunk kernel needs 0.9 us but takes 16 rows per wave)
    if (wave_reg) {
        const dim3 g((uint32_t)((n_rows + CHAIN_WAVES - 1) / CHAIN_WAVES)), b(64 * CHAIN_WAVES);
        if (A.band < 64) hipLaunchKernelGGL(chain_wave_reg_kernel<1>, g, b, 0, st, A);
        else hipLaunchKernelGGL(chain_wave_reg_kernel<2>, g, b, 0, st, A);
    }
    const bool quad_deep = !wave_reg && !A.lane_dp && !force_serial && !qd_off && A.band <= 4 * QD && !(getenv("PSK_CHAIN_LANE") && getenv("PSK_CHAIN_LANE")[0] == '0');
    if (quad_deep) {
        A.ovf_list = L.ovf; A.ovf_count = L.misc + 8;      // misc was zeroed above
        A.lane_dp = 1;                                     // (chain_chunk_list_kernel walks the list)
        PSK_TRY(order_rows());
        const uint32_t qw = (uint32_t)((n_rows + 15) / 16);
        hipLaunchKernelGGL(chain_quad_deep_kernel, dim3((qw + LANE_WAVES - 1) / LANE_WAVES), dim3(64 * LANE_WAVES), 0, st, A);
        const uint32_t lw = (uint32_t)std::min<size_t>((n_rows + CHAIN_WAVES - 1) / CHAIN_WAVES, 2048);
        hipLaunchKernelGGL(chain_chunk_list_kernel, dim3(lw), dim3(64 * CHAIN_WAVES), 0, st, A);
    }
    if (!A.lane_dp && !wave_reg)
    hipLaunchKernelGGL(chain_chunk_kernel, dim3((uint32_t)((n_rows + CHAIN_WAVES - 1) / CHAIN_WAVES)), dim3(64 * CHAIN_WAVES), 0, st, A);
    ctx->t_end();
    SelArgs SA{};
    SA.chunks = L.chunks; SA.n_chunks = L.nch; SA.cbase = L.cbase; SA.n_pairs = n_pairs;
    SA.c_score = A.c_score; SA.c_q0 = A.c_q0; SA.c_q1 = A.c_q1; SA.c_r0 = A.c_r0; SA.c_r1 = A.c_r1; SA.c_n = A.c_n; SA.c_rc = A.c_rc; SA.c_state = A.c_state;
    SA.out = L.cout; SA.two_c = A.two_c; SA.force_serial = force_serial; SA.stats = L.misc + 1;
    // the pairs that have a chunk table, in pair order (misc[9] = their number, misc[10] = pairs listed for select_big_kernel)
    const bool use_live = n_pairs > 4096;      // below that the list costs more launches than it saves workgroups
    SA.live = use_live ? L.live : nullptr; SA.n_live = L.misc + 9; SA.big_list = L.big_list; SA.big_count = L.misc + 10;
    if (use_live) {
        hipcub::CountingInputIterator<uint32_t> ids(0);
        size_t tl = 0;
        PSK_HIP(hipcub::DeviceSelect::If(nullptr, tl, ids, L.live, L.misc + 9, (int)n_pairs, IsLivePair{L.nch}, st));
        PSK_TRY(ctx->q_c.reserve(std::max(tl, std::max(tmp, std::max(tmp2, tmp3)))));
        PSK_HIP(hipcub::DeviceSelect::If(ctx->q_c.p, tl, ids, L.live, L.misc + 9, (int)n_pairs, IsLivePair{L.nch}, st));
    }
    ctx->t_begin(K_SELECT);
    {   // batches of pairs with short chunk tables (contigs): one lane per pair first; PSK_SELECT_TINY=0 leaves every pair to the wave kernels
        static const bool tiny_off = getenv("PSK_SELECT_TINY") && getenv("PSK_SELECT_TINY")[0] == '0';
        SA.tiny_done = use_live && !force_serial && !tiny_off && n_rows / n_pairs < 16;
        SA.rest_list = (uint32_t*)L.hits_sel; SA.rest_count = L.misc + 13;      // (hits_sel: free until the hits are selected; misc was zeroed by pair_table_kernel)
        if (SA.tiny_done) hipLaunchKernelGGL(select_tiny_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, SA);
    }
    // (the list of the second tier lives in huge_list: select_big_kernel only writes that after select_mid_kernel has read it)
    hipLaunchKernelGGL(select_kernel, dim3(std::min<uint32_t>(n_pairs, 16384u)), dim3(64), 0, st, SA, L.huge_list, L.misc + 14);
    hipLaunchKernelGGL(select_mid_kernel, dim3(std::min<uint32_t>(n_pairs, 768u)), dim3(64), 0, st, SA, (const uint32_t*)L.huge_list, (const uint32_t*)(L.misc + 14));
    {   // pairs whose candidates do not fit the LDS kernel (large genomes); workgroups of small pairs exit at once
        if (!force_serial) {
            BigArgs BA{};
            BA.S = SA; BA.pstart = L.pstart;
            char* E = (char*)ctx->q_e.p;
            BA.key = (unsigned long long*)E; uint32_t* U = (uint32_t*)(E + 8 * na);
            BA.slot = U; BA.crow = U + na; BA.idx = U + 2 * na; BA.pm = U + 3 * na; BA.pm2 = U + 4 * na; BA.ord = U + 5 * na; BA.clist = U + 6 * na;
            BA.conf = (uint8_t*)(U + 7 * na);
            BA.parts = (uint32_t*)(E + (((size_t)na * (8 + 4 * 7 + 1) + 255) & ~(size_t)255));
            BA.huge_list = L.huge_list; BA.huge_count = L.misc + 11; BA.ctr = L.misc + 32;
            static const uint32_t solo = getenv("PSK_BIG_SOLO") ? (uint32_t)std::max(atoi(getenv("PSK_BIG_SOLO")), CMAX) : BIG_SOLO;
            BA.solo = solo;
            hipLaunchKernelGGL(select_big_kernel, dim3(std::min<uint32_t>(n_pairs, 64u)), dim3(BIG_T), 0, st, BA);
            // the cooperative launch only where a pair can have more than BIG_SOLO candidates (a candidate needs 3 anchors; there are
            // at most as many anchors as the capacity). Its workgroups spin at barriers, so all of them must be resident at once, and a
            // CU holds two of them: 512 in all. A batch of Gb-scale pairs takes the device's one full-size launch (BIG_GMAX workgroups,
            // under huge_mu until the batch's synchronisation); any other batch - where such a pair is an exception - a share of the rest
            // that stays safe if every lane launched at once.
            if (na / 3 > solo) {
                // co-resident slots of this kernel on THIS device (occupancy query, once): a partition with fewer CUs, a CU mask or
                // a different LDS budget changes it, and a workgroup that cannot become resident would be waited on forever
                static std::mutex slots_mu;
                static std::unordered_map<int, uint32_t> slots_of;
                uint32_t slots;
                {
                    std::lock_guard<std::mutex> lk(slots_mu);
                    auto it = slots_of.find(ctx->device);
                    if (it == slots_of.end()) {
                        int per_cu = 0, cus = 0;
                        PSK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, select_huge_kernel, BIG_T, 0));
                        PSK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
                        if (const char* e = getenv("PSK_HUGE_SLOTS")) { per_cu = 1; cus = std::max(0, atoi(e)); }      // tests: pretend a smaller device
                        it = slots_of.emplace(ctx->device, (uint32_t)std::max(0, per_cu) * (uint32_t)std::max(0, cus)).first;
                    }
                    slots = it->second;
                }
                uint32_t nb = BIG_GMAX;
                const char* hm_env = getenv("PSK_HUGE_MIN_SEEDS");      // (tests: batches of small pairs take the full-size launch and its mutex too)
                if (n_items / n_pairs > (hm_env ? strtoull(hm_env, nullptr, 10) : (1ull << 20))) {      // the device's one full-size launch: the largest power of two that is resident at once
                    ctx->huge_acquire();
                    while (nb > 1 && nb > slots) nb >>= 1;
                } else {                                     // a share of what the full-size launch leaves, safe if every lane launched at once
                    const uint32_t rest = slots > BIG_GMAX ? slots - BIG_GMAX : 0;
                    while (nb > 1 && (size_t)nb * ctx->dev->max_lanes > rest) nb >>= 1;
                }
                if (nb < 8) nb = 1;      // one workgroup per pair: no barrier between workgroups, nothing to wait for
                hipLaunchKernelGGL(select_huge_kernel, dim3(nb), dim3(BIG_T), 0, st, BA);
            }
        }
    }
    hipLaunchKernelGGL(chunk_seeds_kernel, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, A);      // (inside the selection's timer: the seeds between a chunk's outermost kept anchors - 3.9 % of the 10 000 x 10 000 step that no timer held)
    ctx->t_end();
    ReduceArgs R{};
    R.chunks = L.cout; R.n_chunks = L.nch; R.cbase = L.cbase; R.pstart = L.pstart; R.pcnt = gsi_one ? L.aoff : nullptr; R.pairs = L.pairs; R.pair_qr = L.pair_qr;
    R.k = prm.k; R.median = o->median; R.robust = o->robust;
    R.min_af = o->min_aligned_frac > 0 ? o->min_aligned_frac : 0.15; R.hits = L.hits;
    if (o->median || o->robust) {
        PSK_TRY(ctx->q_f.reserve(sizeof(double) * (2 * n_rows + 1024 * (size_t)n_pairs + 1024)));
        R.big_vals = (double*)ctx->q_f.p;
    }
    ctx->t_begin(K_PAIR_REDUCE);
    R.live = use_live ? L.live : nullptr; R.n_live = L.misc + 9;
    if (use_live) hipLaunchKernelGGL(pair_empty_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, R, n_pairs);
    // many pairs with short chunk tables (contigs): one wave per pair first; the workgroup-per-pair kernel then only sees the long tables
    const char* rs_env = getenv("PSK_REDUCE_SMALL");
    const bool no_small = rs_env && rs_env[0] == '0';
    // tables of 65 .. 512 rows (pairs of ~5 Mb genomes: ~250): one wave per pair, eight rows per lane; PSK_REDUCE_WAVE=0 leaves them to the workgroup kernel (tests, A/B)
    const bool no_wave = getenv("PSK_REDUCE_WAVE") && getenv("PSK_REDUCE_WAVE")[0] == '0';
    R.wave_done = !no_wave && !no_small && L.rows_pair_max > 64u && n_rows / n_pairs <= 64u * RW_PER;
    // tables of <= 64 rows (contigs; the short pairs beside the others): one wave per pair, a row per lane (also without the live list: the few pairs of one contig's query)
    R.small_done = !no_small && (n_rows / n_pairs < 16 || R.wave_done);
    // contig batches with the mean ANI: pairs of up to four chunk rows by one lane each first (PSK_REDUCE_TINY=0: by a wave each)
    static const bool no_tiny = getenv("PSK_REDUCE_TINY") && getenv("PSK_REDUCE_TINY")[0] == '0';
    R.tiny_done = R.small_done && !no_tiny && !o->median && !o->robust && n_rows / n_pairs < 16;
    if (R.tiny_done) hipLaunchKernelGGL(pair_reduce_tiny_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, R, n_pairs);
    if (R.small_done) hipLaunchKernelGGL(pair_reduce_small_kernel, dim3(std::min<uint32_t>((n_pairs + 3) / 4, 8192u)), dim3(256), 0, st, R, n_pairs);
    if (R.wave_done) hipLaunchKernelGGL(pair_reduce_wave_kernel, dim3(std::min<uint32_t>((n_pairs + 3) / 4, 16384u)), dim3(256), 0, st, R, n_pairs);
    // (when no pair of the batch can have more rows than the wave kernels take - contigs have 1-3 chunks, 5 Mb genomes ~250 - the two
    // workgroup-per-pair kernels would only walk the pairs to find that out: 24 ms per 17 M contig pairs)
    if (!((R.small_done && L.rows_pair_max <= 64u) || (R.small_done && R.wave_done && L.rows_pair_max <= 64u * RW_PER)))
        hipLaunchKernelGGL(pair_reduce_kernel, dim3(std::min<uint32_t>(n_pairs, 8192u)), dim3(256), 0, st, R, n_pairs);
    if (n_rows > (size_t)RED_SMALL && L.rows_pair_max > (uint32_t)RED_SMALL) hipLaunchKernelGGL(pair_reduce_large_kernel, dim3(std::min<uint32_t>(n_pairs, 512u)), dim3(256), 0, st, R, n_pairs);      // some pair may have more than RED_SMALL rows: a small grid walks the list for them
    ctx->t_end();
    // measurement (timers on): candidate chains and live chunk-table rows of the batch, for the selection's and the reduce's byte counts
    if (ctx->dev->timing) hipLaunchKernelGGL(work_rows_kernel, dim3((uint32_t)std::min<size_t>((n_rows + 255) / 256, 2048)), dim3(256), 0, st, (const ChunkOut*)L.cout, (const uint32_t*)L.nch, (const uint32_t*)L.cbase, (const uint32_t*)L.row_pair, (uint32_t)n_rows, (unsigned long long*)(L.misc + 20));
    // learned-ANI regression (lib.rs:611-614): explicit request, or the default rule c >= 70 && !median when a model is given
    const bool learned = o->model && (o->learned_ani == 1 || (o->learned_ani == -1 && prm.c >= 70 && !o->median));
    if (learned) learned_apply_launch(o->model, L.hits, L.pair_qr, d_qd, d_rd, n_pairs, st);
    return PSK_OK;
}

static uint64_t anchor_cap_for(Lane* ctx, size_t n_items, bool sparse = false, bool gb_scale = false) {
    const uint64_t have = ctx->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) > 128 ? ctx->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) - 128 : 0;     // anchors the per-anchor arrays already hold
    // non-repetitive genomes: at most ~one anchor per query seed; contigs against a whole database (sparse): a third of the (pair, seed) items match
    // (100 bytes of scratch per anchor: 2^30 items would reserve 136 GB otherwise; a batch that does not fit is rerun with the true total)
    // Gb-scale pairs: a seed has ~6.5 matches (chance 15-mer hits in 3 Gb beside the true one) - sized for that at once: the first batch used to overflow, and its
    // second attempt freed 7 GB to allocate 40 GB, which takes 1.5 s when the driver is still clearing memory a previous process released (profiles/r3/r3y_50x_alloc_trace.txt)
    const uint64_t want = gb_scale ? (uint64_t)n_items * 7 + 65536 : sparse ? (uint64_t)n_items / 2 + 65536 : (uint64_t)n_items + n_items / 4 + 65536;
    return std::min<uint64_t>(std::max(have, want), 0x7FFFFF00ull);
}

// outcome of a launch sequence, read back with the hits
struct ChainTail { uint32_t misc[16]; unsigned long long total64, visited, cands, rows; };      // (misc[16..23]: the anchor total; index entries the join visited; with the timers on, candidates and live chunk rows)
static bool join_wide_default() { const char* e = getenv("PSK_JOIN"); return e && !strcmp(e, "wide"); }
static psk_status chain_check(const ChainTail& T, uint32_t n_pairs, uint64_t* cap, bool* wide, bool* retry) {
    *retry = false;
    if (!*wide && T.misc[5]) { *wide = true; *retry = true; return PSK_OK; }   // a count or contig number the packed join format cannot hold: rerun in the wide format
    if (T.total64 >= 0x7FFFFFF0ull) {   // the 32-bit offsets wrapped (or would not fit the per-anchor arrays): the caller splits the batch
        psk_set_error("%u pair(s) yield %llu anchors, more than one launch takes (2^31)%s", n_pairs, T.total64, n_pairs > 1 ? "" : ": the pair is too repetitive to chain");
        return PSK_ELIMIT;
    }
    if (T.total64 > *cap) { *cap = std::min<uint64_t>(T.total64 + T.total64 / 8 + 65536, 0x7FFFFF00ull); *retry = true; return PSK_OK; }
    if (T.misc[0] & 1u) { psk_set_error("internal: chunk table overflow"); return PSK_EHIP; }
    return PSK_OK;
}

struct HostPair { const psk_sketch* r; const psk_sketch* q; };

// one launch sequence over an explicit list of (ref, query) pairs; out[p] in pair order
static psk_status chain_batch(Lane* ctx, const HostPair* hp, uint32_t n_pairs, const psk_query_opts* o, psk_hit* out) {
    hipStream_t st = ctx->stream;
    // descriptor table: one entry per distinct sketch
    std::unordered_map<const psk_sketch*, uint32_t> slot;
    std::vector<SketchDesc> descs;
    std::vector<uint2> qr(n_pairs);
    std::vector<uint32_t> h_sbase(n_pairs + 1), h_cbase(n_pairs + 1);
    uint64_t items = 0, rows = 0;
    auto desc_of = [&](const psk_sketch* s) { auto it = slot.find(s); if (it != slot.end()) return it->second; uint32_t i = (uint32_t)descs.size(); slot.emplace(s, i); descs.push_back(make_desc(s)); return i; };
    for (uint32_t p = 0; p < n_pairs; p++) {
        const uint32_t qi = desc_of(hp[p].q), ri = desc_of(hp[p].r);
        qr[p] = make_uint2(qi, ri);
        h_sbase[p] = (uint32_t)items; h_cbase[p] = (uint32_t)rows;
        items += descs[qi].n; rows += descs[qi].rows;
    }
    h_sbase[n_pairs] = (uint32_t)items; h_cbase[n_pairs] = (uint32_t)rows;
    if (items == 0 || rows == 0) {
        for (uint32_t p = 0; p < n_pairs; p++) { out[p] = psk_hit{}; out[p].ani = -1.0f; out[p].ani_raw = -1.0f; }
        return PSK_OK;
    }
    if (items >= 0xFFFFFF00ull || rows >= 0xFFFFFF00ull) { psk_set_error("batch of %u pairs exceeds the per-launch limits", n_pairs); return PSK_ELIMIT; }
    ChainBufs L;
    PSK_TRY(chain_layout(ctx, n_pairs, (size_t)items, (size_t)rows, 0, &L));
    PSK_TRY(ctx->q_h.reserve(al256(sizeof(SketchDesc) * descs.size()) + 256));
    SketchDesc* d_desc = (SketchDesc*)ctx->q_h.p;
    PSK_HIP(hipMemcpyAsync(d_desc, descs.data(), sizeof(SketchDesc) * descs.size(), hipMemcpyHostToDevice, st));
    PSK_HIP(hipMemcpyAsync(L.pair_qr, qr.data(), 8 * (size_t)n_pairs, hipMemcpyHostToDevice, st));
    PSK_HIP(hipMemcpyAsync(L.sbase, h_sbase.data(), 4 * (size_t)(n_pairs + 1), hipMemcpyHostToDevice, st));
    PSK_HIP(hipMemcpyAsync(L.cbase, h_cbase.data(), 4 * (size_t)(n_pairs + 1), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(pair_build_list_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, L.pair_qr, d_desc, d_desc, n_pairs, L.pairs);
    void* hpin;
    PSK_TRY(ctx->pinned(sizeof(psk_hit) * n_pairs + 512, &hpin));
    ChainTail* T = (ChainTail*)hpin; psk_hit* h_hits = (psk_hit*)((char*)hpin + 256);
    uint64_t cap = anchor_cap_for(ctx, (size_t)items);
    bool wide = join_wide_default();
    for (int attempt = 0;; attempt++) {
        PSK_TRY(chain_run(ctx, L, n_pairs, (size_t)items, (size_t)rows, hp[0].q->params, o, d_desc, d_desc, cap, wide));
        PSK_HIP(hipMemcpyAsync(T, L.misc, 256 + sizeof(psk_hit) * n_pairs, hipMemcpyDeviceToHost, st));      // status words, anchor total and the hits behind them: one copy (ChainTail mirrors misc[0..17])
        PSK_HIP(hipStreamSynchronize(st));      // the ONE synchronisation of a launch sequence (also keeps the host staging above alive)
        ctx->huge_release();
        bool retry;
        PSK_TRY(chain_check(*T, n_pairs, &cap, &wide, &retry));
        if (!retry) { ctx->dev->w_pairs += n_pairs; ctx->dev->w_items += items; ctx->dev->w_anchors += T->total64; ctx->dev->w_cands += T->cands; ctx->dev->w_rows += T->rows; break; }
        if (attempt >= 3) { psk_set_error("internal: anchor capacity did not converge"); return PSK_EHIP; }
    }
    for (uint32_t p = 0; p < n_pairs; p++) { out[p] = h_hits[p]; out[p].reserved = 0; }
    return PSK_OK;
}

// chain an arbitrary list of (ref, query) pairs; out[i] belongs to pair i (ref_index is left to the caller)
psk_status chain_pairs_impl(Lane* ctx, const psk_sketch* const* refs, const psk_sketch* const* queries, uint32_t n,
                            const psk_query_opts* o, psk_hit* out) {
    if (!ctx || !o || (n && (!refs || !queries || !out))) { psk_set_error("chain: NULL argument"); return PSK_EINVAL; }
    if (o->learned_ani == 1 && !o->model) { psk_set_error("learned ANI requested but no regression model is loaded (skani's GBDT weights are embedded in the skani crate; supply them with psk_model_load_file)"); return PSK_ENOMODEL; }
    if (o->model && o->model->ctx != ctx->dev) { psk_set_error("the regression model belongs to another context"); return PSK_EINVAL; }
    for (uint32_t i = 0; i < n; i++) {
        if (!refs[i] || !queries[i]) { psk_set_error("chain: NULL sketch in pair %u", i); return PSK_EINVAL; }
        if (!queries[i]->has_seeds) { psk_set_error("query sketch was built with seed=False; it cannot be chained"); return PSK_EINVAL; }
        if (!refs[i]->has_seeds) { psk_set_error("reference of pair %u was sketched with seed=False; it cannot be chained", i); return PSK_EINVAL; }
        if (refs[i]->params.k != queries[i]->params.k || refs[i]->params.c != queries[i]->params.c) { psk_set_error("pair %u: reference and query were sketched with different parameters", i); return PSK_EINVAL; }
    }
    {   // one index build for every sketch of the call that lacks one (references and queries together)
        std::vector<const psk_sketch*> all(refs, refs + n);
        all.insert(all.end(), queries, queries + n);
        PSK_TRY(ensure_index(ctx, all.data(), (uint32_t)all.size()));
    }
    // bound one launch: lb/cnt/aoff cost 12 B per (pair, query seed); anchors ~64 B each
    const uint64_t MAX_ITEMS = 1ull << 27; const uint32_t MAX_PAIRS = 1u << 18;
    std::vector<HostPair> hp;
    uint32_t b = 0;
    while (b < n) {
        hp.clear();
        uint64_t items = 0; uint32_t e = b;
        while (e < n && hp.size() < MAX_PAIRS && (hp.empty() || items + queries[e]->n_seeds <= MAX_ITEMS)) { hp.push_back({refs[e], queries[e]}); items += queries[e]->n_seeds; e++; }
        psk_status rc = chain_batch(ctx, hp.data(), (uint32_t)hp.size(), o, out + b);
        if (rc == PSK_ELIMIT && hp.size() > 1) {   // too many anchors for 32-bit offsets: halve the batch until single pairs
            std::vector<HostPair> todo(hp);
            size_t step = (todo.size() + 1) / 2;
            for (size_t s0 = 0; s0 < todo.size();) {
                const size_t n1 = std::min(step, todo.size() - s0);
                rc = chain_batch(ctx, todo.data() + s0, (uint32_t)n1, o, out + b + s0);
                if (rc == PSK_ELIMIT && n1 > 1) { step = (n1 + 1) / 2; continue; }
                if (rc != PSK_OK) return rc;
                s0 += n1;
            }
            rc = PSK_OK;
        }
        PSK_TRY(rc);
        b = e;
    }
    return PSK_OK;
}

psk_status chain_impl(Lane* ctx, const psk_sketch* const* refs, uint32_t n_refs, const psk_sketch* q,
                      const psk_query_opts* o, psk_hit* out) {
    if (!q) { psk_set_error("chain: NULL query"); return PSK_EINVAL; }
    std::vector<const psk_sketch*> qs(n_refs, q);
    PSK_TRY(chain_pairs_impl(ctx, refs, qs.data(), n_refs, o, out));
    for (uint32_t i = 0; i < n_refs; i++) out[i].ref_index = i;
    return PSK_OK;
}

// ------------------------------------------------------------------ Database.query / query_many (lib.rs:569-659)
// n_queries x (screen every reference, chain the shortlist, keep ani > 0.1). The pass matrix, the shortlist, the pair
// table and the ani > 0.1 filter all stay on the device; the host sees, per round, the per-query pass counts and the
// per-reference flags (ONE synchronisation: it sizes the batches and indexes the references about to be chained), and per
// batch of up to 2^20 pairs the surviving hits (ONE synchronisation).
// ------------------------------------------------------------------ seed prefilter of rescued queries
// A contig with fewer than SMALL_MARKER_COUNT markers passes the marker screen against EVERY reference (lib.rs:617-630), but a pair
// with fewer than MIN_ANCHORS shared seeds cannot form a chain and never produces a hit. For a batch of such contigs the exact
// anchor count of every (contig, reference) pair is cheap the other way round: the contigs' seed k-mers (a few hundred thousand
// entries) are cut into k-mer slices that fit an LDS hash table, and every reference's k-mer-sorted index - whose entries of one
// slice are contiguous - streams past the table of its slice. Pairs below MIN_ANCHORS are then taken out of the pass matrix, so
// that the join, which probes a 1.3 MB reference index once per (pair, query seed), only sees the pairs that can chain (metagenome
// with rescue: 10.7 M pairs -> ~1 M; the join was 92 of 231 ms). Counts are exact: one per (query seed, reference seed) of equal k-mer.
constexpr uint32_t PF_SLOTS = 8192;           // LDS hash slots per slice (48 KB: 4-byte k-mer + 2-byte query each; three workgroups per CU)
constexpr uint32_t PF_MAX_FILL = 4096;
constexpr int PF_T = 512;
constexpr uint32_t PF_EMPTY = 0xFFFFFFFFu;    // no k-mer of k <= 15 (30 bits)

__global__ __launch_bounds__(256) void pref_gather_kernel(const SketchDesc* __restrict__ qd, const uint32_t* __restrict__ rq, const uint32_t* __restrict__ eoff,
                                                          const uint32_t* __restrict__ qn, uint32_t n_resc, uint32_t* __restrict__ e_key, uint32_t* __restrict__ e_qid) {
    const uint32_t j = blockIdx.x;
    if (j >= n_resc) return;
    const uint32_t* __restrict__ km = qd[rq[j]].kmer;
    const uint32_t o = eoff[j], nn = qn[j];
    for (uint32_t i = threadIdx.x; i < nn; i += blockDim.x) { e_key[o + i] = km[i]; e_qid[o + i] = j; }
}

// workgroup (slice, reference chunk): table of the slice in LDS, then the slice's stretch of every reference of the chunk
__global__ __launch_bounds__(PF_T) void pref_count_kernel(const uint32_t* __restrict__ e_key, const uint32_t* __restrict__ e_qid, uint32_t n_entries,
                                                         uint32_t slice_shift, const SketchDesc* __restrict__ rd, uint32_t n_refs, uint32_t refs_per_chunk,
                                                         uint32_t* __restrict__ cnt, uint32_t* __restrict__ overflow) {
    __shared__ uint32_t t_key[PF_SLOTS];
    __shared__ uint16_t t_qid[PF_SLOTS];
    __shared__ uint32_t s_lo, s_hi;
    const uint32_t slice = blockIdx.x, chunk = blockIdx.y;
    const uint32_t k_lo = slice << slice_shift;
    const uint64_t k_hi64 = ((uint64_t)(slice + 1)) << slice_shift;      // exclusive
    for (uint32_t i = threadIdx.x; i < PF_SLOTS; i += blockDim.x) t_key[i] = PF_EMPTY;
    if (threadIdx.x == 0) {      // the slice's entries in the k-mer-sorted table
        uint32_t a = 0, b = n_entries;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if (e_key[mid] < k_lo) a = mid + 1; else b = mid; }
        s_lo = a; b = n_entries;
        while (a < b) { const uint32_t mid = (a + b) >> 1; if ((uint64_t)e_key[mid] < k_hi64) a = mid + 1; else b = mid; }
        s_hi = a;
    }
    __syncthreads();
    const uint32_t lo = s_lo, hi = s_hi;
    if (hi == lo) return;
    if (hi - lo > PF_MAX_FILL) { if (threadIdx.x == 0) atomicOr(overflow, 1u); return; }      // skewed k-mers: the caller leaves the pass matrix as it is
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t km = e_key[i];
        uint32_t slot = (km * 2654435761u) >> 19;      // 13 bits
        while (atomicCAS(&t_key[slot], PF_EMPTY, km) != PF_EMPTY) slot = (slot + 1) & (PF_SLOTS - 1);
        t_qid[slot] = (uint16_t)e_qid[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r0 = chunk * refs_per_chunk, r1 = r0 + refs_per_chunk < n_refs ? r0 + refs_per_chunk : n_refs;
    for (uint32_t r = r0 + wave; r < r1; r += PF_T / 64) {      // a wave per reference
        const SketchDesc& R = rd[r];
        const uint32_t rn = R.n;
        const uint32_t* __restrict__ key = R.key;
        if (rn == 0 || key == nullptr) continue;
        uint32_t first, last;
        if (slice_shift >= R.bshift) {      // slice boundaries are bucket boundaries: two reads of the bucket table
            const uint32_t sh = slice_shift - R.bshift;
            first = R.bucket[slice << sh]; last = R.bucket[(slice + 1) << sh];
        } else {
            uint32_t a = 0, b = rn;
            while (a < b) { const uint32_t mid = (a + b) >> 1; if (key[mid] < k_lo) a = mid + 1; else b = mid; }
            first = a; b = rn;
            while (a < b) { const uint32_t mid = (a + b) >> 1; if ((uint64_t)key[mid] < k_hi64) a = mid + 1; else b = mid; }
            last = a;
        }
        for (uint32_t i = first + lane; i < last; i += 64) {
            const uint32_t km = key[i];
            uint32_t slot = (km * 2654435761u) >> 19;
            for (;;) {
                const uint32_t k2 = t_key[slot];
                if (k2 == PF_EMPTY) break;
                if (k2 == km) atomicAdd(&cnt[(size_t)t_qid[slot] * n_refs + r], 1u);
                slot = (slot + 1) & (PF_SLOTS - 1);
            }
        }
    }
}

__global__ __launch_bounds__(256) void pref_apply_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ rq, uint32_t n_resc, uint32_t n_refs,
                                                         const uint32_t* __restrict__ overflow, uint8_t* __restrict__ pass) {
    if (*overflow) return;
    const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (size_t)n_resc * n_refs) return;
    const uint32_t j = (uint32_t)(cell / n_refs), r = (uint32_t)(cell % n_refs);
    if (cnt[cell] < MIN_ANCHORS) pass[(size_t)rq[j] * n_refs + r] = 0;
}

// how many references carry a k-mer index (low word) / a probe table (high word): the descriptor table is stale when this moves
// Seed prefilter of rescued queries through the database-wide seed index: the exact anchor count of (query, every reference) is one lookup per query seed
// (pref_count_kernel streams every reference's own k-mer index past LDS tables of the queries' k-mers: it needs those indexes, a gather and a radix sort).
// One workgroup per rescued query, a 16-bit counter per reference in LDS (saturating at MIN_ANCHORS); pairs below MIN_ANCHORS leave the pass matrix.
__global__ __launch_bounds__(256) void gsi_prefilter_kernel(const SketchDesc* __restrict__ qd, const uint32_t* __restrict__ rq, uint32_t n_refs,
                                                            const uint32_t* __restrict__ g_key, const unsigned long long* __restrict__ g_val, const uint32_t* __restrict__ g_bucket, int g_shift,
                                                            uint8_t* __restrict__ pass) {
    extern __shared__ uint32_t s_pc[];      // two 16-bit counters per word
    const uint32_t q = rq[blockIdx.x];
    const SketchDesc Q = qd[q];
    const uint32_t nwd = (n_refs + 1u) / 2u;
    for (uint32_t i = threadIdx.x; i < nwd; i += blockDim.x) s_pc[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    constexpr int U = 2;
    for (uint32_t i0 = wave * U; i0 < Q.n; i0 += 4 * U) {
        uint32_t km[U], lo[U], hi[U];
#pragma unroll
        for (int u = 0; u < U; u++) km[u] = i0 + u < Q.n ? Q.kmer[i0 + u] : 0u;
#pragma unroll
        for (int u = 0; u < U; u++) { lo[u] = 0; hi[u] = 0; if (i0 + u < Q.n) { const uint32_t b = km[u] >> g_shift; lo[u] = g_bucket[b]; hi[u] = g_bucket[b + 1]; } }
#pragma unroll
        for (int u = 0; u < U; u++)
            for (uint32_t x = lo[u] + lane; x < hi[u]; x += 64u)
                if (g_key[x] == km[u]) {
                    const uint32_t ref = (uint32_t)(g_val[x] >> 48), sh = (ref & 1u) * 16u;
                    if (((s_pc[ref >> 1] >> sh) & 0xFFFFu) < MIN_ANCHORS) atomicAdd(&s_pc[ref >> 1], 1u << sh);      // (at most MIN_ANCHORS - 1 + 256 concurrent adds: no carry into the neighbour)
                }
    }
    __syncthreads();
    uint8_t* row = pass + (size_t)q * n_refs;
    for (uint32_t r = threadIdx.x; r < n_refs; r += blockDim.x) if (((s_pc[r >> 1] >> ((r & 1u) * 16u)) & 0xFFFFu) < MIN_ANCHORS) row[r] = 0;
}

static uint64_t index_stamp(const psk_db* db) {
    uint64_t v = 0;
    for (const psk_sketch* r : db->refs) v += (uint64_t)(r->idx != nullptr) + ((uint64_t)(r->ptab != nullptr) << 32);
    return v;
}
static psk_status refresh_ref_descs(Lane* ctx, psk_db* db) {
    const uint32_t n = (uint32_t)db->refs.size();
    const uint64_t indexed = index_stamp(db);
    if (!db->desc_dirty && db->desc_indexed == indexed && db->desc_n == n) return PSK_OK;
    std::vector<SketchDesc>& h = db->h_refdesc;     // stays alive until the copy has drained (every query ends with a synchronisation)
    h.resize(n);
    for (uint32_t i = 0; i < n; i++) h[i] = make_desc(db->refs[i]);
    PSK_TRY(db->d_refdesc.reserve(ctx->dev, sizeof(SketchDesc) * (size_t)n + 256));
    PSK_HIP(hipMemcpyAsync(db->d_refdesc.p, h.data(), sizeof(SketchDesc) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    db->desc_dirty = false; db->desc_indexed = indexed; db->desc_n = n;
    return PSK_OK;
}

static psk_status build_gsi(Lane* ctx, psk_db* db);
static psk_status build_bsi(Lane* ctx, psk_db* db);
// The record a query call hands back: psk_hit (the reference's three numbers, the reference's index and every chaining integer behind them: parity tests) or
// psk_hit_min (hit.rs:77-104's fields in 20 bytes: what crosses PCIe - and xGMI - when nobody asked for the integers: 9.5 M hits of a metagenome step are 763 MB
// as psk_hit). The chain stage writes psk_hit per pair on the device either way; the ani > 0.1 selection (lib.rs:654) converts on its way out.
template <class H> struct HitRec;
template <> struct HitRec<psk_hit> {
    __host__ __device__ static psk_hit from_raw(const psk_hit& r) { return r; }
    static uint32_t local_query(const psk_hit& h) { return h.reserved; }      // pair_reduce left the round-local query index there
    static void finish(psk_hit& h, uint32_t) { h.reserved = 0; }
};
template <> struct HitRec<psk_hit_min> {
    __host__ __device__ static psk_hit_min from_raw(const psk_hit& r) { psk_hit_min m; m.ani = r.ani; m.af_query = r.af_query; m.af_ref = r.af_ref; m.ref_index = r.ref_index; m.query = r.reserved | (r.learned ? 0x80000000u : 0u); return m; }
    static uint32_t local_query(const psk_hit_min& h) { return h.query & 0x7FFFFFFFu; }
    static void finish(psk_hit_min& h, uint32_t q) { h.query = (h.query & 0x80000000u) | q; }      // the query's index within the call
};
template <class H> struct ToRec { __host__ __device__ H operator()(const psk_hit& r) const { return HitRec<H>::from_raw(r); } };
template <class H> struct RecPasses { __host__ __device__ bool operator()(const H& h) const { return h.ani > 0.1f; } };   // lib.rs:654

template <class H>
static psk_status query_many_t(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o,
                               HitListT<H>& all, uint64_t* offsets) {
    hipStream_t st = ctx->stream;
    offsets[0] = 0;
    if (o->learned_ani == 1 && !o->model) { psk_set_error("learned ANI requested but no regression model is loaded"); return PSK_ENOMODEL; }
    if (o->model && o->model->ctx != ctx->dev) { psk_set_error("the regression model belongs to another context"); return PSK_EINVAL; }
    // Locking: the call holds the database SHARED while it computes, so queries from several host threads overlap on their
    // lanes; whatever rebuilds device state other lanes may be reading (marker table, inverted index, the references' k-mer
    // indexes and descriptor table) is done EXCLUSIVELY, synchronised before the lock is handed back.
    std::shared_lock<std::shared_mutex> sh(db->rw);
    const uint32_t n = (uint32_t)db->refs.size();
    auto exclusive = [&](auto&& fn) -> psk_status {
        sh.unlock();
        psk_status rc;
        {
            std::unique_lock<std::shared_mutex> ex(db->rw);
            rc = db->refs.size() == n ? fn() : PSK_EINVAL;
            if (rc == PSK_OK && hipStreamSynchronize(st) != hipSuccess) rc = PSK_EHIP;
        }
        sh.lock();
        if (db->refs.size() != n) { psk_set_error("the database was modified while it was being queried"); return PSK_EINVAL; }
        return rc;
    };
    for (uint32_t i = 0; i < n_queries; i++) if (!queries[i]) { psk_set_error("query_many: NULL query %u", i); return PSK_EINVAL; }
    if (n == 0) { for (uint32_t i = 0; i < n_queries; i++) offsets[i + 1] = 0; return PSK_OK; }
    const double screen_val = o->cutoff != 0.0 ? o->cutoff : 0.80;   // lib.rs:603-609
    // queries per round (pass matrix <= 1 GiB). A round costs ~3.5 ms of host work with the GPU idle (its screen set-up, the shortlist, the last batch's hits):
    // 65 536 queries per round instead of 16 384 is 2 rounds instead of 7 for 100 000 contigs (metagenome step 420 -> 384 ms); PSK_ROUND_QUERIES overrides (tests, A/B)
    static const uint32_t qb_env = getenv("PSK_ROUND_QUERIES") ? (uint32_t)std::max(1, atoi(getenv("PSK_ROUND_QUERIES"))) : 0u;
    const uint32_t QB = std::max<uint32_t>(1, std::min<uint32_t>(qb_env ? qb_env : 65536u, (uint32_t)((1ull << 30) / n)));
    {
        const char* force = getenv("PSK_SCREEN");
        const bool want_inv = force ? !strcmp(force, "inv") : ((uint64_t)n * std::min(QB, n_queries) >= (1ull << 18));
        if (db->tables_dirty || (want_inv && db->inv_dirty) || (db->has_dups && db->canon_dirty))
            PSK_TRY(exclusive([&]() -> psk_status {
                PSK_TRY(upload_marker_table(ctx, db));
                if (want_inv) PSK_TRY(build_inverted(ctx, db));
                if (db->has_dups && db->canon_dirty) {
                    PSK_TRY(db->d_canon.reserve(ctx->dev, 4 * (size_t)n));
                    PSK_HIP(hipMemcpyAsync(db->d_canon.p, db->canon.data(), 4 * (size_t)n, hipMemcpyHostToDevice, st));
                    db->canon_dirty = false;
                }
                return PSK_OK;
            }));
    }
    std::vector<uint32_t> h_cnt; std::vector<uint8_t> h_flag;
    std::vector<SketchDesc> h_qd;
    int64_t h_qd_gsi_round = -1;      // the round (its first query) whose descriptors h_qd holds in the seed-index form (make_desc(.., true))
    std::vector<BatchQ> bqs;
    std::vector<uint2> gsl_tab, gsl_ebase; std::vector<uint32_t> gsl_qn;      // slice join: a batch's wave table (host copies live until the batch's synchronisation)
    std::unique_ptr<LaneGuard> lane2;      // the second lane of rounds that keep two batches in flight (taken at the first such round, held to the end of the call)
    for (uint32_t b = 0; b < n_queries; b += QB) {
        const uint32_t m = std::min(QB, n_queries - b);
        // ---- screen: pass matrix on the device, counts + flags to the host
        const size_t o_pass = 0, o_cnt = al256((size_t)m * n), o_flag = al256(o_cnt + 8 * (size_t)m), o_end = o_flag + n;      // (d_cnt: the queries' pass counts, then their counts of index blocks with a passing reference)
        PSK_TRY(ctx->q_i.reserve(o_end + 256));
        uint8_t* d_pass = (uint8_t*)ctx->q_i.p + o_pass; uint32_t* d_cnt = (uint32_t*)((char*)ctx->q_i.p + o_cnt); uint8_t* d_flag = (uint8_t*)ctx->q_i.p + o_flag;
        // a single query (psk_query: one contig against the database) is as slow as its chain of waits: its k-mer index is launched here,
        // ahead of the screen, and not waited for - one host synchronisation fewer per call
        // (a database that has not been queried yet indexes its references and the query in ONE launch further down: the headline step)
        if (n_queries == 1 && !db->desc_dirty && db->desc_n == n && queries[0]->has_seeds && queries[0]->store && queries[0]->n_seeds && !queries[0]->idx &&
            queries[0]->params.k == db->params.k && queries[0]->params.c == db->params.c)
            PSK_TRY(ensure_index(ctx, queries, 1, true));
        ScreenStaging keep;
        PSK_TRY(screen_many_device(ctx, db, queries + b, m, screen_val, !o->faster_small, d_pass, keep));
        if (db->has_dups) hipLaunchKernelGGL(pass_canon_kernel, dim3(m), dim3(256), 0, st, d_pass, n, (const uint32_t*)db->d_canon.p);
        // ---- rescued short queries: exact anchor counts against every reference, pairs that cannot chain leave the pass matrix
        PoolScratch pf_buf;      // lives until the round's synchronisations are through, like the host arrays the copies read
        std::vector<uint32_t> rq, eoff, qn;
        {
            const char* pf_env = getenv("PSK_PREFILTER");      // "0": never; "1": whatever the number of pairs (tests)
            const bool pf_off = pf_env && pf_env[0] == '0', pf_force = pf_env && pf_env[0] == '1';
            uint64_t E = 0;
            if (!o->faster_small && !pf_off && db->params.k <= 15)
                for (uint32_t i = 0; i < m; i++) {
                    const psk_sketch* q = queries[b + i];
                    if (q->has_seeds && q->store && q->n_seeds && q->n_seeds <= 4096 && q->n_markers < SMALL_MARKER_COUNT &&
                        q->params.k == db->params.k && q->params.c == db->params.c && rq.size() < 65535) {
                        rq.push_back(i); eoff.push_back((uint32_t)E); qn.push_back((uint32_t)q->n_seeds); E += q->n_seeds;
                    }
                }
            bool refs_ok = !rq.empty() && ((uint64_t)rq.size() * n >= (pf_force ? 1ull : (1ull << 20))) && (uint64_t)rq.size() * n * 4 <= (1ull << 31);
            if (refs_ok) for (const psk_sketch* rs : db->refs) if (!rs->has_seeds || rs->params.k != db->params.k || rs->params.c != db->params.c) { refs_ok = false; break; }
            // through the database-wide seed index where the database can have one (no per-reference index, no gather, no sort); PSK_GSI_JOIN=0: the per-reference path
            const bool gsi_pf_off = getenv("PSK_GSI_JOIN") && getenv("PSK_GSI_JOIN")[0] == '0';      // (read per call: tests switch it within a process)
            if (refs_ok && !gsi_pf_off && n <= 65536u && !join_wide_default()) {
                if (db->gsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_gsi(ctx, db); }));
                if (db->gsi_state == 1) {
                    const uint32_t nr = (uint32_t)rq.size();
                    h_qd.resize(m);
                    for (uint32_t i = 0; i < m; i++) h_qd[i] = make_desc(queries[b + i], true);
                    h_qd_gsi_round = (int64_t)b;      // (the round's descriptors for the seed-index paths: built once, see below)
                    const size_t o_qd = al256(4 * (size_t)nr), o_endp = o_qd + sizeof(SketchDesc) * (size_t)m;
                    PSK_TRY(pf_buf.reserve(ctx->dev, o_endp + 256));
                    char* Bp = (char*)pf_buf.p;
                    PSK_HIP(hipMemcpyAsync(Bp, rq.data(), 4 * (size_t)nr, hipMemcpyHostToDevice, st));
                    PSK_HIP(hipMemcpyAsync(Bp + o_qd, h_qd.data(), sizeof(SketchDesc) * (size_t)m, hipMemcpyHostToDevice, st));
                    static std::once_flag pf_once; static hipError_t pf_rc = hipSuccess;
                    std::call_once(pf_once, [] { pf_rc = hipFuncSetAttribute((const void*)gsi_prefilter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536 + 16); });
                    PSK_HIP(pf_rc);
                    hipLaunchKernelGGL(gsi_prefilter_kernel, dim3(nr), dim3(256), 4 * (size_t)((n + 1) / 2), st, (const SketchDesc*)(Bp + o_qd), (const uint32_t*)Bp, n,
                                       (const uint32_t*)db->gsi_key.p, (const unsigned long long*)db->gsi_val.p, (const uint32_t*)db->gsi_bucket.p, db->gsi_shift, d_pass);
                    refs_ok = false;      // (done)
                }
            }
            if (refs_ok) {
                bool all_idx = !db->desc_dirty && db->desc_n == n;
                for (const psk_sketch* rs : db->refs) if (!rs->idx && rs->n_seeds && rs->store) all_idx = false;
                if (!all_idx || index_stamp(db) != db->desc_indexed)
                    PSK_TRY(exclusive([&]() -> psk_status {
                        std::vector<const psk_sketch*> all_refs(db->refs.begin(), db->refs.end());
                        PSK_TRY(ensure_index(ctx, all_refs.data(), (uint32_t)all_refs.size()));
                        return refresh_ref_descs(ctx, db);
                    }));
                const uint32_t nr = (uint32_t)rq.size();
                // slices: a power of two with ~2 048 entries each (the table takes 4 096)
                uint32_t slices = 1; while ((uint64_t)slices * 2048 < E && slices < (1u << 16)) slices <<= 1;
                const uint32_t kbits = 2u * (uint32_t)db->params.k;
                uint32_t lg = 0; while ((1u << lg) < slices) lg++;
                if (lg > kbits) { lg = kbits; slices = 1u << lg; }
                const uint32_t slice_shift = kbits - lg;
                h_qd.resize(m);
                for (uint32_t i = 0; i < m; i++) h_qd[i] = make_desc(queries[b + i]);
                h_qd_gsi_round = -1;
                size_t ts = 0;
                PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, ts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)E, 0, (int)kbits, st));
                const size_t o_rq = 0, o_eoff = al256(4 * (size_t)nr), o_qn = al256(o_eoff + 4 * (size_t)nr), o_qd = al256(o_qn + 4 * (size_t)nr),
                             o_k0 = al256(o_qd + sizeof(SketchDesc) * (size_t)m), o_v0 = al256(o_k0 + 4 * E), o_k1 = al256(o_v0 + 4 * E), o_v1 = al256(o_k1 + 4 * E),
                             o_cnt = al256(o_v1 + 4 * E), o_ovf = al256(o_cnt + 4 * (size_t)nr * n), o_tmp = al256(o_ovf + 4), o_endp = o_tmp + ts + 256;
                PSK_TRY(pf_buf.reserve(ctx->dev, o_endp));
                char* Bp = (char*)pf_buf.p;
                uint32_t *d_rq = (uint32_t*)(Bp + o_rq), *d_eoff = (uint32_t*)(Bp + o_eoff), *d_qn = (uint32_t*)(Bp + o_qn);
                SketchDesc* d_pqd = (SketchDesc*)(Bp + o_qd);
                uint32_t *k0 = (uint32_t*)(Bp + o_k0), *v0 = (uint32_t*)(Bp + o_v0), *k1 = (uint32_t*)(Bp + o_k1), *v1 = (uint32_t*)(Bp + o_v1);
                uint32_t *d_pcnt = (uint32_t*)(Bp + o_cnt), *d_ovf = (uint32_t*)(Bp + o_ovf);
                PSK_HIP(hipMemcpyAsync(d_rq, rq.data(), 4 * (size_t)nr, hipMemcpyHostToDevice, st));
                PSK_HIP(hipMemcpyAsync(d_eoff, eoff.data(), 4 * (size_t)nr, hipMemcpyHostToDevice, st));
                PSK_HIP(hipMemcpyAsync(d_qn, qn.data(), 4 * (size_t)nr, hipMemcpyHostToDevice, st));
                PSK_HIP(hipMemcpyAsync(d_pqd, h_qd.data(), sizeof(SketchDesc) * (size_t)m, hipMemcpyHostToDevice, st));
                PSK_HIP(hipMemsetAsync(d_pcnt, 0, 4 * (size_t)nr * n + 256 + 4, st));      // counts and (behind them) the overflow flag
                hipLaunchKernelGGL(pref_gather_kernel, dim3(nr), dim3(256), 0, st, d_pqd, d_rq, d_eoff, d_qn, nr, k0, v0);
                PSK_HIP(hipcub::DeviceRadixSort::SortPairs(Bp + o_tmp, ts, (const uint32_t*)k0, k1, (const uint32_t*)v0, v1, (int)E, 0, (int)kbits, st));
                const uint32_t chunks = std::max<uint32_t>(1, std::min<uint32_t>(n, 4096 / slices));      // ~4 096 workgroups in all
                const uint32_t rpc = (n + chunks - 1) / chunks;
                hipLaunchKernelGGL(pref_count_kernel, dim3(slices, (n + rpc - 1) / rpc), dim3(PF_T), 0, st, (const uint32_t*)k1, (const uint32_t*)v1, (uint32_t)E, slice_shift,
                                   (const SketchDesc*)db->d_refdesc.p, n, rpc, d_pcnt, d_ovf);
                hipLaunchKernelGGL(pref_apply_kernel, dim3((uint32_t)(((size_t)nr * n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_pcnt, (const uint32_t*)d_rq, nr, n, (const uint32_t*)d_ovf, d_pass);
            }
        }
        void* hpin;
        uint64_t round_blocks = 0;      // over the round's queries: index blocks (2^BSI_BLOG references each) that hold a passing reference
        if ((size_t)m * n <= 65536) {     // a handful of queries: the pass rows themselves cross (<= 64 kB), counted on the host (two launches fewer)
            PSK_TRY(ctx->pinned((size_t)m * n + 64, &hpin));
            PSK_HIP(hipMemcpyAsync(hpin, d_pass, (size_t)m * n, hipMemcpyDeviceToHost, st));
            PSK_HIP(hipStreamSynchronize(st));
            h_cnt.assign(m, 0); h_flag.assign(n, 0);
            const uint8_t* hp = (const uint8_t*)hpin;
            for (uint32_t i = 0; i < m; i++) {
                uint32_t last_blk = 0xFFFFFFFFu;
                for (uint32_t r = 0; r < n; r++) if (hp[(size_t)i * n + r]) { h_cnt[i]++; h_flag[r] = 1; if ((r >> BSI_BLOG) != last_blk) { last_blk = r >> BSI_BLOG; round_blocks++; } }
            }
        } else {
            PSK_HIP(hipMemsetAsync(d_flag, 0, n, st));
            hipLaunchKernelGGL(pass_count_kernel, dim3(m), dim3(256), 0, st, d_pass, n, d_cnt, d_flag, d_cnt + m);
            PSK_TRY(ctx->pinned(8 * (size_t)m + n + 64, &hpin));
            PSK_HIP(hipMemcpyAsync(hpin, d_cnt, 8 * (size_t)m, hipMemcpyDeviceToHost, st));
            PSK_HIP(hipMemcpyAsync((char*)hpin + 8 * (size_t)m, d_flag, n, hipMemcpyDeviceToHost, st));
            PSK_HIP(hipStreamSynchronize(st));
            h_cnt.assign((uint32_t*)hpin, (uint32_t*)hpin + m);
            for (uint32_t i = 0; i < m; i++) round_blocks += ((const uint32_t*)hpin)[m + i];
            h_flag.assign((uint8_t*)hpin + 8 * (size_t)m, (uint8_t*)hpin + 8 * (size_t)m + n);
        }
        // ---- the references and queries about to be chained: validate, index, describe
        std::vector<const psk_sketch*> need;
        for (uint32_t r = 0; r < n; r++) if (h_flag[r]) {
            const psk_sketch* rs = db->refs[r];
            if (!rs->has_seeds) { psk_set_error("reference %u ('%s') was sketched with seed=False; it cannot be chained", r, db->names[r].c_str()); return PSK_EINVAL; }
            need.push_back(rs);
        }
        const size_t n_need_refs = need.size();
        uint64_t round_pairs = 0;
        for (uint32_t i = 0; i < m; i++) if (h_cnt[i]) {
            const psk_sketch* q = queries[b + i];
            if (!q->has_seeds) { psk_set_error("query sketch was built with seed=False; it cannot be chained"); return PSK_EINVAL; }
            if (q->params.k != db->params.k || q->params.c != db->params.c) { psk_set_error("query %u and the database were sketched with different parameters", b + i); return PSK_EINVAL; }
            need.push_back(q);
            round_pairs += h_cnt[i];
        }
        for (const psk_sketch* rs : need) if (rs->params.k != db->params.k || rs->params.c != db->params.c) { psk_set_error("a reference and the database were sketched with different parameters"); return PSK_EINVAL; }
        if (round_pairs == 0) { for (uint32_t i = 0; i < m; i++) offsets[b + i + 1] = offsets[b + i]; continue; }
        // Rounds of many SMALL pairs (metagenome contigs) do not merge-join through the sketches' own k-mer indexes: they go through the database-wide seed
        // index (one lookup per query SEED finds its matches in every reference: gsi_join_kernel; no per-sketch index is read, so none is built for such a
        // round - neither for the references nor for the round's 65 536 contigs) or, where the database cannot have one, through per-reference probe tables
        // (one 64-byte line per (pair, seed)). PSK_PROBE=0 never, =1 whatever the round's shape; PSK_GSI_JOIN=0: the probe tables (tests, A/B)
        bool round_probe = false, round_gsi = false, want_small = false, round_slice = false, round_bsi = false;
        uint64_t round_items = 0;      // (pair, query seed) items of the round
        for (uint32_t i = 0; i < m; i++) round_items += (uint64_t)h_cnt[i] * queries[b + i]->n_seeds;
        static const double max_blocks_join = getenv("PSK_GSL_MAX_BLOCKS") ? atof(getenv("PSK_GSL_MAX_BLOCKS")) : 4.0;
        {
            const char* pb_env = getenv("PSK_PROBE");
            const bool pb_off = pb_env && pb_env[0] == '0', pb_force = pb_env && pb_env[0] == '1';
            want_small = !pb_off && (pb_force || (round_pairs >= 16384 && round_items / round_pairs < 2048));
            const bool gsi_join_off = getenv("PSK_GSI_JOIN") && getenv("PSK_GSI_JOIN")[0] == '0';      // (read per round: tests switch it within a process; chain_run follows the plan)
            // Rounds of many MID-SIZED pairs (all-vs-all of ~5 Mb genomes: every query passes against its family) go through the same index by (query, slice) waves
            // (slice_join.hip) instead of one merge join per pair: one lookup per query SEED where the per-pair join makes one per (pair, seed). PSK_GSI_SLICE=0 never,
            // =1 whatever the round's shape (tests, A/B)
            const char* sl_env = getenv("PSK_GSI_SLICE");      // (read per round: tests switch it within a process)
            const bool sl_off = sl_env && sl_env[0] == '0', sl_force = sl_env && sl_env[0] == '1';
            const bool want_slice = !want_small && !sl_off && (sl_force || (round_pairs >= 2048 && round_items / round_pairs >= 2048 && round_items / round_pairs <= (1u << 18)));
            static const double max_blocks = getenv("PSK_GSL_MAX_BLOCKS") ? atof(getenv("PSK_GSL_MAX_BLOCKS")) : 4.0;
            uint64_t q_with = 0; for (uint32_t i = 0; i < m; i++) q_with += h_cnt[i] != 0;
            const bool few_blocks = (double)round_blocks <= max_blocks * (double)std::max<uint64_t>(q_with, 1);
            if (want_small && !gsi_join_off && n <= 65536u && !join_wide_default()) {
                // contigs: through the index in blocks of references when their passing references sit in few of them (a contig's relatives - what the marker screen and the
                // prefilter of rescued contigs leave), through the database-wide index otherwise (a rescued contig against EVERY reference: one walk instead of one per block)
                static const bool bsi_small_off = getenv("PSK_BSI_SMALL") && getenv("PSK_BSI_SMALL")[0] == '0';      // (tests, A/B)
                if (db->gsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_gsi(ctx, db); }));
                round_gsi = db->gsi_state == 1;
                if (round_gsi && !bsi_small_off) {      // (every wave of the join chooses by its own query: both indexes are handed over)
                    if (db->bsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_bsi(ctx, db); }));
                    round_bsi = db->bsi_state == 1;
                }
            }
            // (the slice join walks, per query, the index BLOCKS that hold one of its passing references: worth it while those are few - relatives that sit next to each
            // other in the database; a query whose references are scattered over many blocks would walk its seeds once per block: PSK_GSL_MAX_BLOCKS, default 4 on average)
            if (want_slice && !gsi_join_off && n <= 65536u && !join_wide_default() && (sl_force || few_blocks)) {
                if (db->bsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_bsi(ctx, db); }));
                round_gsi = round_slice = round_bsi = db->bsi_state == 1;
            }
        }
        if (round_gsi) {
            round_probe = !round_slice;      // (the round's batches are sized for small pairs)
            if (db->desc_dirty || db->desc_n != n) PSK_TRY(exclusive([&]() -> psk_status { return refresh_ref_descs(ctx, db); }));
        } else {
        {   // references first (shared state: exclusive), then this call's own query sketches
            bool refs_stale = db->desc_dirty || db->desc_n != n;
            refs_stale = refs_stale || index_stamp(db) != db->desc_indexed;
            for (size_t i = 0; i < n_need_refs && !refs_stale; i++) refs_stale = !need[i]->idx && need[i]->n_seeds && need[i]->store;
            if (refs_stale)     // one index launch for the references AND this call's queries (a fresh database: the headline step)
                PSK_TRY(exclusive([&]() -> psk_status {
                    PSK_TRY(ensure_index(ctx, need.data(), (uint32_t)need.size()));
                    return refresh_ref_descs(ctx, db);
                }));
            else if (need.size() > n_need_refs) PSK_TRY(ensure_index(ctx, need.data() + n_need_refs, (uint32_t)(need.size() - n_need_refs)));
        }
        if (want_small) {      // probe tables: built once per reference, like the k-mer index, for the references about to be chained
            round_probe = true;
            bool missing = false;
            for (size_t i = 0; i < n_need_refs; i++) {
                if (need[i]->n_seeds < 64 || need[i]->n_seeds > (1u << 22)) { round_probe = false; break; }
                missing = missing || !need[i]->ptab;
            }
            if (round_probe && missing)
                PSK_TRY(exclusive([&]() -> psk_status {
                    PSK_TRY(ensure_probe(ctx, need.data(), (uint32_t)n_need_refs));
                    return refresh_ref_descs(ctx, db);
                }));
        }
        }
        if (!(round_gsi && h_qd_gsi_round == (int64_t)b && h_qd.size() == m)) {      // (65 536 descriptors: ~2 ms of pointer chasing with the GPU idle - the prefilter of this round made the same ones)
            h_qd.resize(m);
            for (uint32_t i = 0; i < m; i++) h_qd[i] = make_desc(queries[b + i], round_gsi);
        }
        PSK_TRY(ctx->q_h.reserve(sizeof(SketchDesc) * (size_t)m + 256));
        SketchDesc* d_qd = (SketchDesc*)ctx->q_h.p;
        PSK_HIP(hipMemcpyAsync(d_qd, h_qd.data(), sizeof(SketchDesc) * (size_t)m, hipMemcpyHostToDevice, st));
        // ---- batches: consecutive (query, rank range) entries under the per-launch limits
        // 2^29 query seeds per batch (about 25 GB of scratch for 5 Mb genomes; the per-pair latency chains of chunk_heads / select /
        // pair_reduce and the batch's synchronisation are spread over four times the pairs of 2^27: all-vs-all 185 -> 169 ms);
        // 2^27 when a query is Gb-scale (~6 anchors per seed from chance 15-mer matches: 2^27 seeds already carry 14 GB of anchors).
        // A batch whose scratch cannot be allocated is planned again at a quarter of the size.
        static const int items_env = getenv("PSK_BATCH_ITEMS_LOG2") ? std::min(31, std::max(16, atoi(getenv("PSK_BATCH_ITEMS_LOG2")))) : 0;
        int items_log2 = items_env ? items_env : 29;
        if (!items_env) for (uint32_t i = 0; i < m; i++) if (h_qd[i].n > (1u << 20)) { items_log2 = 27; break; }
        if (!items_env && items_log2 == 27) {
            // ... 2^28 where the device has the room (7 anchors per seed x 93 bytes of per-anchor arrays, with the buffers' growth slack: 204 GiB; the lane's own arrays count as room): eleven 3 Gb pairs
            // per batch instead of five - the per-pair chains of the chunk walk, the selection's group barriers and the reduction overlap across twice the pairs
            // (8 x 3 Gb: 234 -> 211 ms per step). 2^29 would pass the 2^31 anchors one launch sequence addresses.
            size_t free_b = 0, total_b = 0;
            const size_t need = (size_t)212 << 30;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + ctx->q_d.cap + ctx->q_e.cap <= need && total_b > need) {
                psk_trim_idle_lanes(ctx->dev, ctx);      // (an earlier all-vs-all left 47-94 GB of scratch on each of its two lanes)
                if (lane2) { Scratch* big[] = {&lane2->lane->q_b, &lane2->lane->q_c, &lane2->lane->q_d, &lane2->lane->q_e, &lane2->lane->q_g, &lane2->lane->q_j, &lane2->lane->q_sel}; (void)hipStreamSynchronize(lane2->lane->stream); for (Scratch* s : big) s->release(); }
                (void)hipMemGetInfo(&free_b, &total_b);
            }
            if (free_b + ctx->q_d.cap + ctx->q_e.cap > need) items_log2 = 28;
        }
        // Rounds of many small pairs (contigs): up to 2^22 pairs and 2^30 seeds per batch. The probe join visits a batch's pairs reference by reference, and a line of a
        // reference's table is probed about once per 2^20 pairs of a 5 000-reference database: with twice the pairs every line is probed twice while it is still
        // cached (join 142 -> 124 ms per 100 000 contigs). PSK_BATCH_PAIRS_LOG2 overrides (tests, A/B).
        static const int pairs_env = getenv("PSK_BATCH_PAIRS_LOG2") ? std::min(24, std::max(10, atoi(getenv("PSK_BATCH_PAIRS_LOG2")))) : 0;
        if (!items_env && items_log2 == 29 && round_probe) items_log2 = 30;
        // ... and rounds of mid-sized pairs joined by (query, slice) waves: a launch of 10 000 waves is three waves deep on the chip and its last third runs half empty;
        // 2^30 seeds (268 genomes of 5 Mb and their ~27 000 pairs) per batch: 860 -> 796 ms per 10 000 x 10 000 step (2^28: 969)
        if (!items_env && items_log2 == 29 && round_slice) items_log2 = 30;
        uint64_t max_items = 1ull << items_log2, max_pairs = 1ull << (pairs_env ? pairs_env : (round_probe ? 22 : 21)), max_rows = 1ull << 26;      // (2^22 pairs: 363 -> 353 ms per 100 000 contigs)
        {   // the one-pass index join lays a batch's anchors out at 9/8 of its items (gsi_room_kernel) where about two thirds of that are used: three quarters of the
            // items per batch keep the per-anchor arrays (100 bytes per slot) near what the two passes reserved
            const bool one_off = getenv("PSK_GSI_ONEPASS") && getenv("PSK_GSI_ONEPASS")[0] == '0';
            if (round_gsi && !round_slice && !one_off && !items_env && max_items == (1ull << 30)) max_items = 3ull << 28;
        }
        uint32_t qi = 0, rank = 0;      // next (query, rank) to chain
        std::vector<uint64_t> q_hits(m, 0);            // hits per query of the round
        // The hits of a batch are appended to the result (and counted per query) while the NEXT batch runs on the GPU: two halves of one
        // pinned staging buffer, sized once for the round so that it never moves while a half is still unread.
        const size_t half_pairs = (size_t)std::min<uint64_t>(round_pairs, max_pairs);
        const size_t half_bytes = al256(sizeof(psk_hit) * half_pairs + 512);
        void* hpin2 = nullptr;
        PSK_TRY(ctx->pinned(2 * half_bytes, &hpin2));
        {   // one allocation for the round's hits (untouched pages are free). Later rounds: the hits so far say how many the whole call will
            // bring - growing the list round by round copied everything gathered before, 14, 27, 39, ... ms with the GPU idle (a third of the
            // metagenome step: profiles/r3/r3i_metagenome_gaps.txt)
            size_t want = all.n + (size_t)std::min<uint64_t>(round_pairs, 1ull << 26);
            // (the extrapolations only where the list has to grow anyway: a second round whose own pairs still fit must not move 6 M hits - 50 ms - because
            // its estimate of the whole call came out 5 % above the first round's)
            if (want > all.cap) {
                if (b == 0 && n_queries > m) want = std::max(want, (size_t)std::min<double>((double)round_pairs * ((double)n_queries / (double)m) * 1.05, (double)(1ull << 27)));      // every pair yields at most one hit
                if (b > 0 && all.n) want = std::max(want, (size_t)((double)all.n * ((double)n_queries / (double)b) * 1.1) + 4096);
            }
            if (round_pairs > 4096 && !all.reserve(want)) { psk_set_error("out of host memory"); return PSK_ENOMEM; }
        }
        int parity = 0;
        const H* pend_hits = nullptr; uint32_t pend_n = 0;
        bool pend_copy = false;      // the pending hits are still crossing on the copy stream
        hipStream_t cst = nullptr;
        // every way out of the round (an error return between two batches included) waits for a copy that is still crossing: the lane's pinned staging and the
        // selection halves it reads go back to the next caller with the lane
        struct CopyDrain { bool& pend; hipStream_t& s; ~CopyDrain() { if (pend && s) (void)hipStreamSynchronize(s); } } copy_drain{pend_copy, cst};
        const size_t sel_half = al256(sizeof(H) * half_pairs + 256);
        auto consume = [&]() -> psk_status {
            if (!pend_n) return PSK_OK;
            if (pend_copy) { PSK_HIP(hipStreamSynchronize(cst)); pend_copy = false; }
            const size_t old = all.n;
            if (!all.reserve_for(pend_n)) { psk_set_error("out of host memory"); return PSK_ENOMEM; }
            // pair_reduce left the round-local query index in `reserved`: counted per query, then cleared. A large batch (130 MB of records per metagenome batch)
            // is moved by a few threads: after the LAST batch of a round nothing is left to hide the move behind (12 ms of a 214 ms step with the GPU idle)
            auto move = [&](size_t lo, size_t hi, bool shared) {
                memcpy(all.p + old + lo, pend_hits + lo, sizeof(H) * (hi - lo));
                for (size_t i = lo; i < hi; i++) {
                    H& h = all.p[old + i];
                    const uint32_t lq = HitRec<H>::local_query(h);
                    if (shared) __atomic_fetch_add(&q_hits[lq], 1u, __ATOMIC_RELAXED); else q_hits[lq]++;      // (slices meet inside a query's hits)
                    HitRec<H>::finish(h, b + lq);
                }
            };
            static const unsigned move_threads = [] { const char* e = getenv("PSK_HIT_THREADS"); const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
                                                      return e ? (unsigned)std::max(1, atoi(e)) : std::min(8u, std::max(1u, hw / 8)); }();
            const unsigned nt = (size_t)pend_n * sizeof(H) >= ((size_t)16 << 20) ? move_threads : 1u;
            if (nt <= 1) move(0, pend_n, false);
            else {
                std::vector<std::thread> th;
                const size_t per = ((size_t)pend_n + nt - 1) / nt;
                size_t started = per;      // records [0, started) have a mover (this thread takes the first slice and whatever no helper could be started for)
                try {
                    th.reserve(nt);
                    for (unsigned t = 1; t < nt; t++) {
                        const size_t lo = std::min<size_t>(pend_n, t * per), hi = std::min<size_t>(pend_n, lo + per);
                        if (hi > lo) { th.emplace_back(move, lo, hi, true); started = hi; }
                    }
                } catch (...) {}      // (no thread to be had: the rest is moved here - nothing may leave this function as an exception, its callers are extern "C")
                move(0, std::min<size_t>(pend_n, per), true);
                if (started < pend_n) move(std::max(started, std::min<size_t>(pend_n, per)), pend_n, true);
                for (std::thread& t : th) t.join();
            }
            all.n += pend_n;
            pend_n = 0;
            return PSK_OK;
        };
        // ---- two batches in flight (rounds of mid-sized pairs joined by (query, slice) waves) ------------------------------------------------------------
        // A batch is a chain of kernels with different appetites - the index walks wait on memory at an occupancy their LDS sets, the DP on instruction issue -
        // and every one of them ends in a tail that leaves the chip half empty. Batches are independent: alternate ones run on a SECOND lane (its own stream,
        // scratch and pinned staging), each driven by a helper thread, at half the seeds per batch, and the two chains fill each other's gaps (10 000 x 10 000:
        // the query 587 -> 520 ms with two callers of half the queries each; profiles/r5/r5_ablation.md). This thread plans the batches and appends their hits in
        // batch order; the helpers touch nothing of the database's lock. A batch that does not fit its lane (memory, too repetitive) ends the mode: the loop
        // below takes over from that batch's first pair at a quarter of the size. PSK_PIPELINE=1 / 0 force / forbid (tests, A/B).
        {
            const int pipe_env = getenv("PSK_PIPELINE") ? atoi(getenv("PSK_PIPELINE")) : -1;      // (read per round: bench.py takes its kernel table from a step run as one chain)
            const bool pipe_want = round_slice && pipe_env != 0 && (pipe_env == 1 || round_items >= (4ull << 29));
            if (pipe_want && !lane2) {
                lane2.reset(new (std::nothrow) LaneGuard(ctx->dev, true));
                if (lane2 && !lane2->lane) lane2.reset();      // every lane is taken (other callers): one chain
            }
            if (pipe_want && lane2) {
                if (!items_env) max_items = 1ull << 29;
                struct PipeJob {
                    uint32_t q0 = 0, r0 = 0, q1 = 0, r1 = 0;
                    std::vector<BatchQ> bqs; std::vector<uint2> tab, ebase; std::vector<uint32_t> qn;      // (host copies feed asynchronous copies: alive until the job is reaped)
                    uint64_t pairs = 0, items = 0, rows = 0, rows_pair_max = 0;
                    psk_status rc = PSK_OK; bool refit = false; char err[512] = "";
                    std::vector<H> hits; uint64_t anchors = 0, cands = 0, wrows = 0, visited = 0, lookups = 0;
                };
                PipeJob job[2];
                std::thread th[2];
                bool live[2] = {false, false};
                Lane* lanes[2] = {ctx, lane2->lane};
                PSK_HIP(hipStreamSynchronize(st));      // the round's tables (pass matrix, query descriptors) are complete before the other stream reads them
                const SketchDesc* d_rd = (const SketchDesc*)db->d_refdesc.p;
                auto exec = [&](Lane* ln, PipeJob* Jp) {
                    PipeJob& J = *Jp;
                    (void)hipSetDevice(ln->device);
                    J.refit = false; J.hits.clear();
                    try {
                    J.rc = [&]() -> psk_status {
                        hipStream_t s2 = ln->stream;
                        const uint32_t n_pairs = (uint32_t)J.pairs;
                        ChainBufs L;
                        psk_status lrc = chain_layout(ln, n_pairs, (size_t)J.items, (size_t)J.rows, J.bqs.size(), &L);
                        if (lrc == PSK_ENOMEM) { J.refit = true; return PSK_OK; }
                        PSK_TRY(lrc);
                        L.rows_pair_max = (uint32_t)std::min<uint64_t>(J.rows_pair_max, 0xFFFFFFFFu);
                        L.g_key = (const uint32_t*)db->bsi_key.p; L.g_val = (const unsigned long long*)db->bsi_val.p; L.g_bucket = (const uint32_t*)db->bsi_bucket.p; L.g_shift = db->bsi_shift; L.g_nb1 = db->bsi_nb1; L.g_blocks = db->bsi_blocks;
                        L.d_pass = d_pass; L.n_refs = n; L.n_bq = (uint32_t)J.bqs.size();
                        uint32_t pm = 1; for (const BatchQ& e : J.bqs) pm = std::max(pm, e.rank_hi - e.rank_lo);
                        L.p_cap = (pm + 15u) & ~15u;
                        L.gsi_onepass = false;
                        J.qn.resize(J.bqs.size());
                        for (size_t e = 0; e < J.bqs.size(); e++) J.qn[e] = h_qd[J.bqs[e].q].n;
                        uint64_t n_rec = 0, n_sl = 0;
                        gsl_make_tab(J.bqs.data(), J.bqs.size(), J.qn.data(), J.tab, J.ebase, &n_rec, &n_sl);
                        if (n_rec >= 0x7FFFFF00ull) { psk_set_error("internal: %llu (pair, slice) records in one batch", (unsigned long long)n_rec); return PSK_ELIMIT; }
                        const size_t o_tab = 0, o_eb = al256(o_tab + 8 * J.tab.size()), o_cnt = al256(o_eb + 8 * J.ebase.size()), o_rec = al256(o_cnt + 4 * (size_t)n_rec),
                                     o_bm = al256(o_rec + 16 * (size_t)n_rec), o_un = al256(o_bm + 4 * (size_t)GSL_WORDS * (size_t)n_rec), o_endj = o_un + 4 * (size_t)GSL_WORDS * (size_t)n_sl;
                        lrc = ln->q_j.reserve(o_endj + 256);
                        if (lrc == PSK_ENOMEM) { J.refit = true; return PSK_OK; }
                        PSK_TRY(lrc);
                        char* Jb = (char*)ln->q_j.p;
                        PSK_HIP(hipMemcpyAsync(Jb + o_tab, J.tab.data(), 8 * J.tab.size(), hipMemcpyHostToDevice, s2));
                        PSK_HIP(hipMemcpyAsync(Jb + o_eb, J.ebase.data(), 8 * J.ebase.size(), hipMemcpyHostToDevice, s2));
                        L.gsi_slice = true; L.gsl_tab = (const uint2*)(Jb + o_tab); L.gsl_n_tab = (uint32_t)J.tab.size(); L.gsl_ebase = (const uint2*)(Jb + o_eb); L.gsl_un = (uint32_t*)(Jb + o_un); L.gsl_n_slices = (uint32_t)n_sl;
                        L.gsl_cnt = (uint32_t*)(Jb + o_cnt); L.gsl_rec = (uint4*)(Jb + o_rec); L.gsl_bm = (uint32_t*)(Jb + o_bm);
                        PSK_HIP(hipMemcpyAsync(L.bq, J.bqs.data(), sizeof(BatchQ) * J.bqs.size(), hipMemcpyHostToDevice, s2));
                        hipLaunchKernelGGL(pair_build_rows_kernel, dim3((uint32_t)J.bqs.size()), dim3(256), 0, s2, L.bq, d_pass, n, d_qd, d_rd, L.pairs, L.sbase, L.cbase, L.pair_qr, n_pairs, (uint32_t)J.items, (uint32_t)J.rows);
                        const bool host_filter = n_pairs <= 4096;
                        H* d_sel = nullptr;
                        if (!host_filter) { lrc = ln->q_sel.reserve(al256(sizeof(H) * (size_t)n_pairs + 256)); if (lrc == PSK_ENOMEM) { J.refit = true; return PSK_OK; } PSK_TRY(lrc); d_sel = (H*)ln->q_sel.p; }
                        void* hp = nullptr;
                        PSK_TRY(ln->pinned(al256(sizeof(psk_hit) * (size_t)n_pairs + 512), &hp));
                        ChainTail* T = (ChainTail*)hp; H* h_sel = (H*)((char*)hp + 256);
                        uint64_t cap = anchor_cap_for(ln, (size_t)J.items, false, false);
                        cap = std::min<uint64_t>(cap, std::max<uint64_t>(ln->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) > 128 ? ln->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) - 128 : 0, (uint64_t)J.items / 4 * 3 + 65536));
                        bool wide = join_wide_default();
                        for (int attempt = 0;; attempt++) {
                            psk_status rrc = chain_run(ln, L, n_pairs, (size_t)J.items, (size_t)J.rows, db->params, o, d_qd, d_rd, cap, wide, false);
                            if (rrc == PSK_ENOMEM) { (void)hipStreamSynchronize(s2); J.refit = true; return PSK_OK; }
                            PSK_TRY(rrc);
                            if (!host_filter) {
                                size_t tmp3 = 0;
                                hipcub::TransformInputIterator<H, ToRec<H>, const psk_hit*> rec_it(L.hits, ToRec<H>());
                                PSK_HIP(hipcub::DeviceSelect::If(nullptr, tmp3, rec_it, d_sel, L.misc + 12, (int)n_pairs, RecPasses<H>(), s2));
                                PSK_TRY(ln->q_c.reserve(tmp3));
                                PSK_HIP(hipcub::DeviceSelect::If(ln->q_c.p, tmp3, rec_it, d_sel, L.misc + 12, (int)n_pairs, RecPasses<H>(), s2));
                                PSK_HIP(hipMemcpyAsync(T, L.misc, sizeof(ChainTail), hipMemcpyDeviceToHost, s2));
                            } else PSK_HIP(hipMemcpyAsync(T, L.misc, 256 + sizeof(psk_hit) * (size_t)n_pairs, hipMemcpyDeviceToHost, s2));
                            PSK_HIP(hipStreamSynchronize(s2));
                            bool retry, was_wide = wide;
                            psk_status rc = chain_check(*T, n_pairs, &cap, &wide, &retry);
                            if ((rc == PSK_ELIMIT && n_pairs > 1) || wide != was_wide) { J.refit = true; return PSK_OK; }      // (the one-chain loop knows what to do with these)
                            PSK_TRY(rc);
                            if (!retry) break;
                            if (attempt >= 3) { psk_set_error("internal: anchor capacity did not converge"); return PSK_EHIP; }
                        }
                        uint32_t n_sel = T->misc[12];
                        if (host_filter) {
                            const psk_hit* raw = (const psk_hit*)((char*)hp + 256);
                            for (uint32_t i = 0; i < n_pairs; i++) { const psk_hit r = raw[i]; if (r.ani > 0.1f) J.hits.push_back(HitRec<H>::from_raw(r)); }
                        } else if (n_sel) {
                            PSK_HIP(hipMemcpyAsync(h_sel, d_sel, sizeof(H) * (size_t)n_sel, hipMemcpyDeviceToHost, s2));
                            PSK_HIP(hipStreamSynchronize(s2));
                            J.hits.assign(h_sel, h_sel + n_sel);
                        }
                        J.anchors = T->total64; J.cands = T->cands; J.wrows = T->rows; J.visited = T->visited;
                        J.lookups = 0; for (const BatchQ& e : J.bqs) J.lookups += h_qd[e.q].n;
                        return PSK_OK;
                    }();
                    } catch (...) { psk_set_error("out of host memory"); J.rc = PSK_ENOMEM; }      // (nothing may leave a helper thread as an exception)
                    if (J.rc != PSK_OK) snprintf(J.err, sizeof J.err, "%s", psk_last_error());
                    if (J.rc != PSK_OK || J.refit) (void)hipStreamSynchronize(ln->stream);      // (whatever was enqueued reads the job's host tables and the lane's scratch)
                    ln->huge_release();      // a batch with a Gb-scale pair took the device's group-selection mutex inside chain_run: it goes back with the batch, whatever its outcome - the other lane's next such batch waits for it
                };
                bool refit = false;
                struct JoinAll { std::thread* t; ~JoinAll() { for (int i = 0; i < 2; i++) if (t[i].joinable()) t[i].join(); } } join_all{th};      // (no way out of this block leaves a helper running; declared after everything the helpers reach by reference)
                auto reap = [&](int sl) -> psk_status {
                    if (th[sl].joinable()) th[sl].join();
                    live[sl] = false;
                    PipeJob& J = job[sl];
                    if (J.rc != PSK_OK) { psk_set_error("%s", J.err); return J.rc; }
                    if (J.refit) { if (!refit) { refit = true; qi = J.q0; rank = J.r0; } return PSK_OK; }
                    if (refit) return PSK_OK;      // a batch after one that is going to be run again: so is this one
                    const size_t nh = J.hits.size();
                    if (nh) {
                        if (!all.reserve_for(nh)) { psk_set_error("out of host memory"); return PSK_ENOMEM; }
                        memcpy(all.p + all.n, J.hits.data(), sizeof(H) * nh);
                        for (size_t i = 0; i < nh; i++) { H& h = all.p[all.n + i]; const uint32_t lq = HitRec<H>::local_query(h); q_hits[lq]++; HitRec<H>::finish(h, b + lq); }
                        all.n += nh;
                    }
                    ctx->dev->w_pairs += J.pairs; ctx->dev->w_items += J.items; ctx->dev->w_anchors += J.anchors; ctx->dev->w_cands += J.cands; ctx->dev->w_rows += J.wrows;
                    ctx->dev->w_lookups += J.lookups; ctx->dev->w_visited += J.visited;
                    return PSK_OK;
                };
                uint32_t k = 0;
                for (;;) {
                    while (qi < m && rank >= h_cnt[qi]) { qi++; rank = 0; }
                    if (qi >= m || refit) break;
                    const int sl = (int)(k & 1u);
                    if (live[sl]) { PSK_TRY(reap(sl)); if (refit) break; }
                    PipeJob& J = job[sl];
                    J.bqs.clear(); J.pairs = J.items = J.rows = J.rows_pair_max = 0;
                    J.q0 = qi; J.r0 = rank;
                    uint32_t pq = qi, pr = rank;
                    while (pq < m) {      // (the plan of the loop below)
                        const uint32_t left = h_cnt[pq] - pr;
                        if (left == 0) { pq++; pr = 0; continue; }
                        const uint64_t qn = h_qd[pq].n, qrows = h_qd[pq].rows;
                        uint64_t take = std::min<uint64_t>(left, max_pairs - J.pairs);
                        take = std::min<uint64_t>(take, GSI_PMAX);
                        if (qn) take = std::min<uint64_t>(take, (max_items - J.items) / qn);
                        if (qrows) take = std::min<uint64_t>(take, (max_rows - J.rows) / qrows);
                        if (take == 0) { if (J.pairs == 0) take = 1; else break; }
                        J.bqs.push_back(BatchQ{pq, pr, pr + (uint32_t)take, (uint32_t)J.pairs, (uint32_t)J.items, (uint32_t)J.rows});
                        J.rows_pair_max = std::max<uint64_t>(J.rows_pair_max, qrows);
                        J.pairs += take; J.items += take * qn; J.rows += take * qrows;
                        pr += (uint32_t)take;
                        if (J.pairs >= max_pairs || J.items >= max_items || J.rows >= max_rows) break;
                    }
                    J.q1 = pq; J.r1 = pr;
                    if (J.items >= 0xFFFFFF00ull || J.rows >= 0xFFFFFF00ull) { psk_set_error("a single pair exceeds the per-launch limits (%llu query seeds)", (unsigned long long)J.items); return PSK_ELIMIT; }
                    qi = pq; rank = pr;
                    if (J.items == 0 || J.rows == 0) continue;      // nothing to chain (queries without seeds): no hits
                    try { th[sl] = std::thread(exec, lanes[sl], &J); }
                    catch (...) { exec(lanes[sl], &J); }      // (no thread to be had: the batch runs here)
                    live[sl] = true;
                    k++;
                }
                for (int i = 0; i < 2; i++) { const int sl = (int)((k + (uint32_t)i) & 1u); if (live[sl]) PSK_TRY(reap(sl)); }
                if (refit) max_items = std::max<uint64_t>(1, max_items / 4);
            }
        }
        while (qi < m) {
            if (rank >= h_cnt[qi]) { qi++; rank = 0; continue; }
            // plan one batch from (qi, rank)
            bqs.clear();
            uint64_t pairs = 0, items = 0, rows = 0, rows_pair_max = 0;
            uint32_t pq = qi, pr = rank;
            while (pq < m) {
                const uint32_t left = h_cnt[pq] - pr;
                if (left == 0) { pq++; pr = 0; continue; }
                const uint64_t qn = h_qd[pq].n, qrows = h_qd[pq].rows;
                uint64_t take = std::min<uint64_t>(left, max_pairs - pairs);
                if (round_gsi) take = std::min<uint64_t>(take, GSI_PMAX);      // (an entry = one wave of the index join: its pairs' cursors sit in LDS)
                if (qn) take = std::min<uint64_t>(take, (max_items - items) / qn);
                if (qrows) take = std::min<uint64_t>(take, (max_rows - rows) / qrows);
                if (take == 0) { if (pairs == 0) take = 1; else break; }      // a single pair always goes through (chain_check refuses what cannot fit)
                bqs.push_back(BatchQ{pq, pr, pr + (uint32_t)take, (uint32_t)pairs, (uint32_t)items, (uint32_t)rows});
                rows_pair_max = std::max<uint64_t>(rows_pair_max, qrows);
                pairs += take; items += take * qn; rows += take * qrows;
                pr += (uint32_t)take;
                if (pairs >= max_pairs || items >= max_items || rows >= max_rows) break;
            }
            const uint32_t n_pairs = (uint32_t)pairs;
            if (items >= 0xFFFFFF00ull || rows >= 0xFFFFFF00ull) { psk_set_error("a single pair exceeds the per-launch limits (%llu query seeds)", (unsigned long long)items); return PSK_ELIMIT; }
            uint32_t n_sel = 0;
            H* h_sel = nullptr;
            if (items == 0 || rows == 0) {
                // nothing to chain (queries without seeds): no hits
            } else {
                ChainBufs L;
                psk_status lrc = chain_layout(ctx, n_pairs, (size_t)items, (size_t)rows, bqs.size(), &L);
                L.rows_pair_max = (uint32_t)std::min<uint64_t>(rows_pair_max, 0xFFFFFFFFu);
                if (round_gsi) {
                    if (round_slice) { L.g_key = (const uint32_t*)db->bsi_key.p; L.g_val = (const unsigned long long*)db->bsi_val.p; L.g_bucket = (const uint32_t*)db->bsi_bucket.p; L.g_shift = db->bsi_shift; L.g_nb1 = db->bsi_nb1; L.g_blocks = db->bsi_blocks; }
                    else {
                        L.g_key = (const uint32_t*)db->gsi_key.p; L.g_val = (const unsigned long long*)db->gsi_val.p; L.g_bucket = (const uint32_t*)db->gsi_bucket.p; L.g_shift = db->gsi_shift;
                        if (round_bsi) { L.b_key = (const uint32_t*)db->bsi_key.p; L.b_val = (const unsigned long long*)db->bsi_val.p; L.b_bucket = (const uint32_t*)db->bsi_bucket.p; L.b_shift = db->bsi_shift; L.b_nb1 = db->bsi_nb1; L.b_blocks = db->bsi_blocks; L.b_max = (uint32_t)std::max(1.0, max_blocks_join); }
                    }
                    L.d_pass = d_pass; L.n_refs = n; L.n_bq = (uint32_t)bqs.size();
                    uint32_t pm = 1; for (const BatchQ& e : bqs) pm = std::max(pm, e.rank_hi - e.rank_lo);
                    L.p_cap = round_slice ? (pm + 15u) & ~15u : (pm + 63u) & ~63u;      // (the slice join's LDS arrays are indexed by pair alone: no need for whole waves of them)
                    const bool one_off = getenv("PSK_GSI_ONEPASS") && getenv("PSK_GSI_ONEPASS")[0] == '0';      // tests, A/B: count pass + scan + emit pass
                    L.gsi_onepass = !one_off && !round_slice;
                    if (round_slice && lrc == PSK_OK) {      // wave table + per-(pair, slice) records of the batch
                        gsl_qn.resize(bqs.size());
                        for (size_t e = 0; e < bqs.size(); e++) gsl_qn[e] = h_qd[bqs[e].q].n;
                        uint64_t n_rec = 0, n_sl = 0;
                        gsl_make_tab(bqs.data(), bqs.size(), gsl_qn.data(), gsl_tab, gsl_ebase, &n_rec, &n_sl);
                        if (n_rec >= 0x7FFFFF00ull) { psk_set_error("internal: %llu (pair, slice) records in one batch", (unsigned long long)n_rec); return PSK_ELIMIT; }
                        const size_t o_tab = 0, o_eb = al256(o_tab + 8 * gsl_tab.size()), o_cnt = al256(o_eb + 8 * gsl_ebase.size()), o_rec = al256(o_cnt + 4 * (size_t)n_rec),
                                     o_bm = al256(o_rec + 16 * (size_t)n_rec), o_un = al256(o_bm + 4 * (size_t)GSL_WORDS * (size_t)n_rec), o_endj = o_un + 4 * (size_t)GSL_WORDS * (size_t)n_sl;
                        lrc = ctx->q_j.reserve(o_endj + 256);
                        if (lrc == PSK_OK) {
                            char* J = (char*)ctx->q_j.p;
                            PSK_HIP(hipMemcpyAsync(J + o_tab, gsl_tab.data(), 8 * gsl_tab.size(), hipMemcpyHostToDevice, st));
                            PSK_HIP(hipMemcpyAsync(J + o_eb, gsl_ebase.data(), 8 * gsl_ebase.size(), hipMemcpyHostToDevice, st));
                            L.gsi_slice = true; L.gsl_tab = (const uint2*)(J + o_tab); L.gsl_n_tab = (uint32_t)gsl_tab.size(); L.gsl_ebase = (const uint2*)(J + o_eb); L.gsl_un = (uint32_t*)(J + o_un); L.gsl_n_slices = (uint32_t)n_sl;
                            L.gsl_cnt = (uint32_t*)(J + o_cnt); L.gsl_rec = (uint4*)(J + o_rec); L.gsl_bm = (uint32_t*)(J + o_bm);
                        }
                    }
                }
                if (lrc == PSK_ENOMEM && n_pairs > 1 && max_items > (1ull << 22)) { max_items >>= 2; continue; }
                PSK_TRY(lrc);
                PSK_HIP(hipMemcpyAsync(L.bq, bqs.data(), sizeof(BatchQ) * bqs.size(), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(pair_build_rows_kernel, dim3((uint32_t)bqs.size()), dim3(256), 0, st, L.bq, d_pass, n, d_qd, (const SketchDesc*)db->d_refdesc.p,
                                   L.pairs, L.sbase, L.cbase, L.pair_qr, n_pairs, (uint32_t)items, (uint32_t)rows);
                // a small batch: status and every record in one copy; a large one: the status alone is waited for, the selected hits then cross on the copy
                // stream WHILE THE NEXT BATCH COMPUTES (600 MB per metagenome step: 15 ms of copies that kept the compute queues idle), out of one of two
                // device halves so that the next batch's selection does not write what is still being read
                const bool host_filter = n_pairs <= 4096;
                H* d_sel = nullptr;
                if (!host_filter) {
                    PSK_TRY(ctx->copy_lane(&cst));
                    PSK_TRY(ctx->q_sel.reserve(2 * sel_half));
                    d_sel = (H*)((char*)ctx->q_sel.p + (parity ? sel_half : 0));
                }
                if (sizeof(psk_hit) * (size_t)n_pairs + 512 > half_bytes) { psk_set_error("internal: batch larger than its staging half"); return PSK_EHIP; }
                hpin = (char*)hpin2 + (parity ? half_bytes : 0);
                ChainTail* T = (ChainTail*)hpin; h_sel = (H*)((char*)hpin + 256);
                uint64_t cap = anchor_cap_for(ctx, (size_t)items, round_probe, items / n_pairs > (1u << 20));
                // (pairs of one family: (1 - d)^15 of a query's seeds match, half of them over the divergences met - three quarters of the items is room enough,
                // and a batch that needs more is rerun with the count walk's total)
                if (round_slice) cap = std::min<uint64_t>(cap, std::max<uint64_t>(ctx->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) > 128 ? ctx->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) - 128 : 0, (uint64_t)items / 4 * 3 + 65536));
                if (L.gsi_onepass) cap = std::min<uint64_t>(std::max<uint64_t>(cap, items + items / 8 + 8 * ((uint64_t)n_pairs + 1) + 64), 0x7FFFFF00ull);      // gsi_room_kernel's layout
                bool too_big = false, wide = join_wide_default();
                static const bool trace_batch = getenv("PSK_TRACE_BATCH") != nullptr;      // diagnostics: host wall clock of every batch (launching, waiting)
                for (int attempt = 0;; attempt++) {
                    struct timespec tb0{}, tb1{}, tb2{};
                    if (trace_batch) clock_gettime(CLOCK_MONOTONIC, &tb0);
                    psk_status rrc = chain_run(ctx, L, n_pairs, (size_t)items, (size_t)rows, db->params, o, d_qd, (const SketchDesc*)db->d_refdesc.p, cap, wide, round_probe && !round_gsi);
                    if (trace_batch) clock_gettime(CLOCK_MONOTONIC, &tb1);
                    if (rrc == PSK_ENOMEM && n_pairs > 1 && max_items > (1ull << 22)) { (void)hipStreamSynchronize(st); ctx->huge_release(); too_big = true; break; }
                    PSK_TRY(rrc);
                    if (!host_filter) {      // (the ani > 0.1 filter of a small batch runs on the host: three launches fewer)
                        size_t tmp3 = 0;
                        hipcub::TransformInputIterator<H, ToRec<H>, const psk_hit*> rec_it(L.hits, ToRec<H>());      // (the record that crosses is made here)
                        PSK_HIP(hipcub::DeviceSelect::If(nullptr, tmp3, rec_it, d_sel, L.misc + 12, (int)n_pairs, RecPasses<H>(), st));
                        PSK_TRY(ctx->q_c.reserve(tmp3));
                        PSK_HIP(hipcub::DeviceSelect::If(ctx->q_c.p, tmp3, rec_it, d_sel, L.misc + 12, (int)n_pairs, RecPasses<H>(), st));   // order-preserving: hits stay in (query, ref) order
                    }
                    if (host_filter) PSK_HIP(hipMemcpyAsync(T, L.misc, 256 + sizeof(psk_hit) * (size_t)n_pairs, hipMemcpyDeviceToHost, st));      // status, anchor total, hits: one copy
                    else PSK_HIP(hipMemcpyAsync(T, L.misc, sizeof(ChainTail), hipMemcpyDeviceToHost, st));
                    PSK_TRY(consume());                     // the previous batch's hits, while this one runs
                    PSK_HIP(hipStreamSynchronize(st));      // the ONE synchronisation of a batch
                    if (trace_batch) {
                        clock_gettime(CLOCK_MONOTONIC, &tb2);
                        static struct timespec last{};
                        const double gap = last.tv_sec ? (tb0.tv_sec - last.tv_sec) * 1e3 + (tb0.tv_nsec - last.tv_nsec) / 1e6 : 0.0;
                        fprintf(stderr, "[psk batch] pairs %u items %llu attempt %d: since last %.1f ms, launch %.1f ms, wait %.1f ms\n", n_pairs, (unsigned long long)items, attempt, gap,
                                (tb1.tv_sec - tb0.tv_sec) * 1e3 + (tb1.tv_nsec - tb0.tv_nsec) / 1e6, (tb2.tv_sec - tb1.tv_sec) * 1e3 + (tb2.tv_nsec - tb1.tv_nsec) / 1e6);
                        last = tb2;
                    }
                    ctx->huge_release();
                    bool retry;
                    psk_status rc = chain_check(*T, n_pairs, &cap, &wide, &retry);
                    if (rc == PSK_ELIMIT && n_pairs > 1) { too_big = true; break; }
                    PSK_TRY(rc);
                    if (!retry && L.gsi_onepass && (T->misc[0] & 4u)) { L.gsi_onepass = false; retry = true; }      // a pair with more anchors than query seeds: with the count pass
                    if (!retry) {
                        ctx->dev->w_pairs += n_pairs; ctx->dev->w_items += items; ctx->dev->w_anchors += T->total64; ctx->dev->w_cands += T->cands; ctx->dev->w_rows += T->rows;
                        if (round_gsi) { uint64_t lk = 0; for (const BatchQ& e : bqs) lk += h_qd[e.q].n; ctx->dev->w_lookups += lk; ctx->dev->w_visited += T->visited; }
                        break;
                    }
                    if (attempt >= 3) { psk_set_error("internal: anchor capacity did not converge"); return PSK_EHIP; }
                }
                if (too_big) { max_items = std::max<uint64_t>(1, items / 4); max_pairs = std::max<uint64_t>(1, pairs / 4); continue; }   // repeat-rich: plan smaller batches from the same position
                n_sel = T->misc[12];
                if (n_pairs <= 4096) {      // host-side filter of a small batch (lib.rs:654), order kept; the raw records become H where they stand (H is no larger)
                    const psk_hit* raw = (const psk_hit*)((char*)hpin + 256);
                    uint32_t w = 0;
                    for (uint32_t i = 0; i < n_pairs; i++) { const psk_hit r = raw[i]; if (r.ani > 0.1f) h_sel[w++] = HitRec<H>::from_raw(r); }
                    n_sel = w;
                }
                else if (n_sel) {
                    PSK_HIP(hipMemcpyAsync(h_sel, d_sel, sizeof(H) * (size_t)n_sel, hipMemcpyDeviceToHost, cst));      // (the batch is complete: its one synchronisation is behind us)
                    pend_copy = true;
                }
            }
            // hits arrive in (query, ref) order; they join the result during the next batch (or after the last one)
            if (n_sel) { pend_hits = h_sel; pend_n = n_sel; parity ^= 1; }
            qi = pq; rank = pr;
        }
        PSK_TRY(consume());
        for (uint32_t i = 0; i < m; i++) offsets[b + i + 1] = offsets[b + i] + q_hits[i];
    }
    return PSK_OK;
}
psk_status query_many_impl(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o, HitList& all, uint64_t* offsets) {
    return query_many_t<psk_hit>(ctx, db, queries, n_queries, o, all, offsets);
}
psk_status query_many_min_impl(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t n_queries, const psk_query_opts* o, HitListMin& all, uint64_t* offsets) {
    return query_many_t<psk_hit_min>(ctx, db, queries, n_queries, o, all, offsets);
}


// ------------------------------------------------------------------ database-wide seed index (psk_db::gsi_*)
struct GsiSeg { const uint32_t* kmer; const uint64_t* pm; uint32_t n, off; };
__global__ __launch_bounds__(256) void gsi_gather_kernel(const GsiSeg* __restrict__ segs, uint32_t* __restrict__ key, unsigned long long* __restrict__ val) {
    const GsiSeg sg = segs[blockIdx.y];
    const unsigned long long ref = (unsigned long long)blockIdx.y << 48;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += gridDim.x * blockDim.x) {
        const unsigned long long pm = sg.pm[i];
        const uint32_t meta = (uint32_t)pm;      // contig << 1 | (fwd < rc)
        key[sg.off + i] = sg.kmer[i];
        val[sg.off + i] = ref | ((unsigned long long)(meta >> 1) << 33) | ((pm >> 32) << 1) | (meta & 1u);
    }
}
// bucket[b] = first entry whose k-mer >> shift is >= b (b = 0 .. nb)
__global__ __launch_bounds__(256) void gsi_bucket_kernel(const uint32_t* __restrict__ key, uint32_t n, int shift, uint32_t nb, uint32_t* __restrict__ bucket) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = key[i] >> shift;
    const uint32_t from = i ? (key[i - 1] >> shift) + 1u : 0u;
    for (uint32_t x = from; x <= b; x++) bucket[x] = i;
    if (i == n - 1) for (uint32_t x = b + 1; x <= nb; x++) bucket[x] = n;
}
// called with the database locked exclusively; leaves gsi_state 1 (built) or 2 (this database cannot have one)
static psk_status build_gsi(Lane* ctx, psk_db* db) {
    if (db->gsi_state) return PSK_OK;
    static const bool off = getenv("PSK_GSI") && getenv("PSK_GSI")[0] == '0';
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)db->refs.size();
    db->gsi_state = 2;
    if (off || n == 0 || n > 65536u || db->params.k > 16) return PSK_OK;
    std::vector<GsiSeg> segs(n);
    uint64_t N = 0; uint32_t maxn = 0;
    for (uint32_t i = 0; i < n; i++) {
        const psk_sketch* r = db->refs[i];
        if (!r->has_seeds || r->contig_len.size() > 32768u || r->params.k != db->params.k || r->params.c != db->params.c) return PSK_OK;
        const uint32_t ns = r->store ? (uint32_t)r->n_seeds : 0u;
        segs[i] = GsiSeg{ns ? r->store->seed_kmer + r->seed_off : nullptr, ns ? r->store->seed_pm + r->seed_off : nullptr, ns, (uint32_t)N};
        N += ns; maxn = std::max(maxn, ns);
        if (N >= 0x7FFFFF00ull) return PSK_OK;
    }
    if (N == 0) return PSK_OK;
    const int kbits = 2 * db->params.k;
    int bits = 4; while (bits < 26 && (8ull << bits) < N) bits++;      // ~8 entries per bucket
    if (bits > kbits) bits = kbits;
    const uint32_t nb = 1u << bits;
    size_t ts = 0;
    PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, ts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)N, 0, kbits, st));
    PoolScratch tmp;      // unsorted copies + sort scratch + segment table: back to the pool when the build is done
    const size_t o_k = 0, o_v = al256(4 * (size_t)N), o_t = al256(o_v + 8 * (size_t)N), o_s = al256(o_t + ts), o_end = o_s + sizeof(GsiSeg) * (size_t)n;
    psk_status rc = tmp.reserve(ctx->dev, o_end + 256);
    if (rc == PSK_OK) rc = db->gsi_key.reserve(ctx->dev, 4 * (size_t)N + 256);
    if (rc == PSK_OK) rc = db->gsi_val.reserve(ctx->dev, 8 * (size_t)N + 256);
    if (rc == PSK_OK) rc = db->gsi_bucket.reserve(ctx->dev, 4 * ((size_t)nb + 2));
    if (rc != PSK_OK) { db->gsi_key.release(); db->gsi_val.release(); db->gsi_bucket.release(); return rc == PSK_ENOMEM ? PSK_OK : rc; }      // no room: the paths that would use it take their other route
    char* T = (char*)tmp.p;
    auto fail = [&](hipError_t e, const char* what) -> psk_status {      // an OPTIONAL index: a failed build leaves no buffer behind, state 2, and does not fail the caller's query
        (void)hipStreamSynchronize(st);
        db->gsi_key.release(); db->gsi_val.release(); db->gsi_bucket.release();
        psk_set_error("%s: %s", what, hipGetErrorString(e));
        return PSK_OK;
    };
    hipError_t e = hipMemcpyAsync(T + o_s, segs.data(), sizeof(GsiSeg) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(e, "gsi: upload");
    hipLaunchKernelGGL(gsi_gather_kernel, dim3(std::max(1u, std::min(64u, (maxn + 4095u) / 4096u)), n), dim3(256), 0, st, (const GsiSeg*)(T + o_s), (uint32_t*)(T + o_k), (unsigned long long*)(T + o_v));
    e = hipcub::DeviceRadixSort::SortPairs(T + o_t, ts, (const uint32_t*)(T + o_k), (uint32_t*)db->gsi_key.p, (const unsigned long long*)(T + o_v), (unsigned long long*)db->gsi_val.p, (int)N, 0, kbits, st);
    if (e != hipSuccess) return fail(e, "gsi: sort");
    hipLaunchKernelGGL(gsi_bucket_kernel, dim3((uint32_t)((N + 255) / 256)), dim3(256), 0, st, (const uint32_t*)db->gsi_key.p, (uint32_t)N, kbits - bits, nb, (uint32_t*)db->gsi_bucket.p);
    e = hipStreamSynchronize(st);      // (segs and tmp die with this frame)
    if (e != hipSuccess) return fail(e, "gsi: build");
    db->gsi_n = N; db->gsi_shift = kbits - bits;
    db->gsi_state = 1;
    return PSK_OK;
}

// ------------------------------------------------------------------ the seed index in blocks of 2^BSI_BLOG references (psk_db::bsi_*: what the slice join walks)
// bucket table of one block: bucket[b] = base + first entry of the block whose k-mer >> shift is >= b (b = 0 .. nb); an empty block: every entry = base
__global__ __launch_bounds__(256) void bsi_bucket_kernel(const uint32_t* __restrict__ key, uint32_t n, int shift, uint32_t nb, uint32_t* __restrict__ bucket, uint32_t base) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (n == 0) { for (uint32_t x = i; x <= nb; x += gridDim.x * 256u) bucket[x] = base; return; }
    if (i >= n) return;
    const uint32_t b = key[i] >> shift;
    const uint32_t from = i ? (key[i - 1] >> shift) + 1u : 0u;
    for (uint32_t x = from; x <= b; x++) bucket[x] = base + i;
    if (i == n - 1) for (uint32_t x = b + 1; x <= nb; x++) bucket[x] = base + n;
}
// called with the database locked exclusively; leaves bsi_state 1 (built) or 2 (this database cannot have one)
static psk_status build_bsi(Lane* ctx, psk_db* db) {
    if (db->bsi_state) return PSK_OK;
    static const bool off = getenv("PSK_GSI") && getenv("PSK_GSI")[0] == '0';
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)db->refs.size();
    db->bsi_state = 2;
    if (off || n == 0 || n > 65536u || db->params.k > 16) return PSK_OK;
    std::vector<GsiSeg> segs(n);
    uint64_t N = 0; uint32_t maxn = 0;
    for (uint32_t i = 0; i < n; i++) {
        const psk_sketch* r = db->refs[i];
        if (!r->has_seeds || r->contig_len.size() > 32768u || r->params.k != db->params.k || r->params.c != db->params.c) return PSK_OK;
        const uint32_t ns = r->store ? (uint32_t)r->n_seeds : 0u;
        segs[i] = GsiSeg{ns ? r->store->seed_kmer + r->seed_off : nullptr, ns ? r->store->seed_pm + r->seed_off : nullptr, ns, (uint32_t)N};
        N += ns; maxn = std::max(maxn, ns);
        if (N >= 0x7FFFFF00ull) return PSK_OK;
    }
    if (N == 0) return PSK_OK;
    const uint32_t n_blocks = (n + (1u << BSI_BLOG) - 1) >> BSI_BLOG;
    uint64_t max_block = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint32_t r0 = b << BSI_BLOG, r1 = std::min<uint32_t>(n, (b + 1) << BSI_BLOG);
        max_block = std::max<uint64_t>(max_block, (uint64_t)segs[r1 - 1].off + segs[r1 - 1].n - segs[r0].off);
    }
    const int kbits = 2 * db->params.k;
    int bits = 4; while (bits < 24 && (8ull << bits) < max_block) bits++;      // ~8 entries per bucket of the largest block
    if (bits > kbits) bits = kbits;
    const uint32_t nb = 1u << bits;
    if ((uint64_t)n_blocks * (nb + 1) >= 0x7FFFFF00ull) return PSK_OK;
    size_t ts = 0;
    PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, ts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)std::min<uint64_t>(max_block, 0x7FFFFFFFull), 0, kbits, st));
    PoolScratch tmp;      // unsorted copies + sort scratch + segment table: back to the pool when the build is done
    const size_t o_k = 0, o_v = al256(4 * (size_t)N), o_t = al256(o_v + 8 * (size_t)N), o_s = al256(o_t + ts), o_end = o_s + sizeof(GsiSeg) * (size_t)n;
    psk_status rc = tmp.reserve(ctx->dev, o_end + 256);
    if (rc == PSK_OK) rc = db->bsi_key.reserve(ctx->dev, 4 * (size_t)N + 256);
    if (rc == PSK_OK) rc = db->bsi_val.reserve(ctx->dev, 8 * (size_t)N + 256);
    if (rc == PSK_OK) rc = db->bsi_bucket.reserve(ctx->dev, 4 * ((size_t)n_blocks * (nb + 1) + 2));
    if (rc != PSK_OK) { db->bsi_key.release(); db->bsi_val.release(); db->bsi_bucket.release(); if (rc == PSK_ENOMEM) return PSK_OK; return rc; }      // no room: the join takes its other route
    char* T = (char*)tmp.p;
    auto fail = [&](hipError_t e, const char* what) -> psk_status {      // (an optional index: a failed build leaves nothing behind and does not fail the query)
        (void)hipStreamSynchronize(st);
        db->bsi_key.release(); db->bsi_val.release(); db->bsi_bucket.release();
        psk_set_error("%s: %s", what, hipGetErrorString(e));
        return PSK_OK;
    };
    hipError_t e = hipMemcpyAsync(T + o_s, segs.data(), sizeof(GsiSeg) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(e, "bsi: upload");
    hipLaunchKernelGGL(gsi_gather_kernel, dim3(std::max(1u, std::min(64u, (maxn + 4095u) / 4096u)), n), dim3(256), 0, st, (const GsiSeg*)(T + o_s), (uint32_t*)(T + o_k), (unsigned long long*)(T + o_v));
    for (uint32_t b = 0; b < n_blocks; b++) {      // one stable sort per block: within a k-mer the entries stay in (reference, contig, position) order
        const uint32_t r0 = b << BSI_BLOG, r1 = std::min<uint32_t>(n, (b + 1) << BSI_BLOG);
        const size_t o = segs[r0].off; const uint32_t cnt = (uint32_t)((uint64_t)segs[r1 - 1].off + segs[r1 - 1].n - o);
        if (cnt) {
            size_t ts_b = ts;
            e = hipcub::DeviceRadixSort::SortPairs(T + o_t, ts_b, (const uint32_t*)(T + o_k) + o, (uint32_t*)db->bsi_key.p + o, (const unsigned long long*)(T + o_v) + o, (unsigned long long*)db->bsi_val.p + o, (int)cnt, 0, kbits, st);
            if (e != hipSuccess) return fail(e, "bsi: sort");
        }
        hipLaunchKernelGGL(bsi_bucket_kernel, dim3(std::max(1u, (cnt + 255u) / 256u)), dim3(256), 0, st, (const uint32_t*)db->bsi_key.p + o, cnt, kbits - bits, nb, (uint32_t*)db->bsi_bucket.p + (size_t)b * (nb + 1), (uint32_t)o);
    }
    e = hipStreamSynchronize(st);      // (segs and tmp die with this frame)
    if (e != hipSuccess) return fail(e, "bsi: build");
    db->bsi_n = N; db->bsi_shift = kbits - bits; db->bsi_nb1 = nb + 1; db->bsi_blocks = n_blocks;
    db->bsi_state = 1;
    return PSK_OK;
}

// ------------------------------------------------------------------ what the one-launch-sequence query (small_query.hip) reads on the device
psk_status small_query_prepare(Lane* ctx, psk_db* db, std::shared_lock<std::shared_mutex>& sh, bool* ok) {
    *ok = false;
    const int state = db->small_state.load(std::memory_order_acquire);
    if (state == 1) { *ok = true; return PSK_OK; }
    if (state == 2) return PSK_OK;
    const uint32_t n = (uint32_t)db->refs.size();
    if (n == 0 || n > SQ_MAX_REFS) return PSK_OK;      // (not recorded: the database may grow into / out of the range)
    for (const psk_sketch* r : db->refs)
        if (!r->has_seeds || !r->store || r->params.k != db->params.k || r->params.c != db->params.c) { db->small_state = 2; return PSK_OK; }
    sh.unlock();
    psk_status rc = PSK_OK;
    {
        std::unique_lock<std::shared_mutex> ex(db->rw);
        if (db->refs.size() == n && db->small_state.load() == 0) {
            auto build = [&]() -> psk_status {
                PSK_TRY(upload_marker_table(ctx, db));
                PSK_TRY(build_inverted(ctx, db));
                if (db->has_dups && db->canon_dirty) {
                    PSK_TRY(db->d_canon.reserve(ctx->dev, 4 * (size_t)n));
                    PSK_HIP(hipMemcpyAsync(db->d_canon.p, db->canon.data(), 4 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
                    db->canon_dirty = false;
                }
                std::vector<const psk_sketch*> all_refs(db->refs.begin(), db->refs.end());
                PSK_TRY(ensure_index(ctx, all_refs.data(), (uint32_t)all_refs.size()));
                PSK_TRY(refresh_ref_descs(ctx, db));
                PSK_HIP(hipStreamSynchronize(ctx->stream));
                PSK_TRY(build_gsi(ctx, db));      // (optional: without it a rescued contig is chained against every reference)
                return PSK_OK;
            };
            rc = build();
            if (rc == PSK_OK) db->small_state.store(1, std::memory_order_release);
        }
    }
    sh.lock();
    PSK_TRY(rc);
    *ok = db->small_state.load(std::memory_order_acquire) == 1 && db->refs.size() == n;
    return PSK_OK;
}
