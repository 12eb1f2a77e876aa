// The marker screen (check_markers_quickly, lib.rs:617-637): per-pair kernels, the inverted marker index, the many-query screen.
#include "query_parts.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>

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

psk_status upload_marker_table(Lane* ctx, psk_db* db) {
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

psk_status build_inverted(Lane* ctx, psk_db* db) {
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
psk_status screen_many_device(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t nq, double screen_val, int rescue_small,
                                     uint8_t* d_pass, ScreenStaging& keep, const Switches& sw) {
    const uint32_t n = (uint32_t)db->refs.size();
    if (n == 0 || nq == 0) return PSK_OK;
    hipStream_t st = ctx->stream;
    PSK_TRY(upload_marker_table(ctx, db));
    const double thresh = pow(screen_val, (double)K_MARKER);
    // small jobs: one workgroup per (ref, query). Large jobs: inverted index + count matrix.
    const char* force = sw.screen.get();    // "inv" / "brute" for tests
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
        if (use_inv && n <= INV_LDS_REFS && !sw.screen_global.get()) {
            static bool lds_attr = false;
            if (!lds_attr) {
                PSK_HIP(hipFuncSetAttribute((const void*)inv_screen_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * INV_LDS_REFS)));
                PSK_HIP(hipFuncSetAttribute((const void*)inv_screen_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * INV_LDS_REFS)));
                lds_attr = true;
            }
            const bool wave_off = sw.screen_wave.get() && sw.screen_wave.get()[0] == '0';      // "0": one lane per marker, binary search over the whole index (A/B, tests)
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
