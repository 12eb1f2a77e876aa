// The database's seed indexes (psk_db::gsi_*, bsi_*) and what the one-launch-sequence query (small_query.hip) reads on the device.
#include "query_parts.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>

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
psk_status build_gsi(Lane* ctx, psk_db* db) {
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
psk_status build_bsi(Lane* ctx, psk_db* db) {
    if (db->bsi_state) return PSK_OK;
    static const bool off = getenv("PSK_GSI") && getenv("PSK_GSI")[0] == '0';
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)db->refs.size();
    db->bsi_state = 2;
    if (off || n == 0 || db->params.k > 16) return PSK_OK;
    // one segment per reference, offsets WITHIN its block; the blocks' own offsets are 64-bit (no bound on the references or the seeds of the database but memory)
    const uint32_t n_blocks = (n + (1u << BSI_BLOG) - 1) >> BSI_BLOG;
    std::vector<GsiSeg> segs(n);
    std::vector<uint64_t> base(n_blocks + 1, 0);
    uint64_t N = 0, max_block = 0; uint32_t maxn = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint32_t r0 = b << BSI_BLOG, r1 = std::min<uint32_t>(n, (b + 1) << BSI_BLOG);
        uint64_t in_block = 0;
        for (uint32_t i = r0; i < r1; i++) {
            const psk_sketch* r = db->refs[i];
            if (!r->has_seeds || r->contig_len.size() > 32768u || r->params.k != db->params.k || r->params.c != db->params.c) return PSK_OK;
            const uint32_t ns = r->store ? (uint32_t)r->n_seeds : 0u;
            if (in_block + ns >= 0x7FFFFF00ull) return PSK_OK;      // (a block is one radix sort: 256 references of more than 8 M seeds each - 1 Gb at c = 125 - have none)
            segs[i] = GsiSeg{ns ? r->store->seed_kmer + r->seed_off : nullptr, ns ? r->store->seed_pm + r->seed_off : nullptr, ns, (uint32_t)in_block};
            in_block += ns; maxn = std::max(maxn, ns);
        }
        base[b] = N; N += in_block; max_block = std::max(max_block, in_block);
    }
    base[n_blocks] = N;
    if (N == 0) return PSK_OK;
    const int kbits = 2 * db->params.k;
    int bits = 4; while (bits < 24 && (8ull << bits) < max_block) bits++;      // ~8 entries per bucket of the largest block
    if (bits > kbits) bits = kbits;
    const uint32_t nb = 1u << bits;
    size_t ts = 0;
    PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, ts, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)max_block, 0, kbits, st));
    PoolScratch tmp;      // ONE block's unsorted copy + sort scratch, and the segment table: back to the pool when the build is done
    const size_t o_k = 0, o_v = al256(4 * (size_t)max_block), o_t = al256(o_v + 8 * (size_t)max_block), o_s = al256(o_t + ts), o_end = o_s + sizeof(GsiSeg) * (size_t)n;
    psk_status rc = tmp.reserve(ctx->dev, o_end + 256);
    if (rc == PSK_OK) rc = db->bsi_key.reserve(ctx->dev, 4 * (size_t)N + 256);
    if (rc == PSK_OK) rc = db->bsi_val.reserve(ctx->dev, 8 * (size_t)N + 256);
    if (rc == PSK_OK) rc = db->bsi_bucket.reserve(ctx->dev, 4 * ((size_t)n_blocks * (nb + 1) + 2));
    if (rc == PSK_OK) rc = db->bsi_base.reserve(ctx->dev, 8 * ((size_t)n_blocks + 1) + 256);
    auto drop = [&]() { db->bsi_key.release(); db->bsi_val.release(); db->bsi_bucket.release(); db->bsi_base.release(); };
    if (rc != PSK_OK) { drop(); if (rc == PSK_ENOMEM) return PSK_OK; return rc; }      // no room: the join takes its other route
    char* T = (char*)tmp.p;
    auto fail = [&](hipError_t e, const char* what) -> psk_status {      // (an optional index: a failed build leaves nothing behind and does not fail the query)
        (void)hipStreamSynchronize(st);
        drop();
        psk_set_error("%s: %s", what, hipGetErrorString(e));
        return PSK_OK;
    };
    hipError_t e = hipMemcpyAsync(T + o_s, segs.data(), sizeof(GsiSeg) * (size_t)n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(db->bsi_base.p, base.data(), 8 * ((size_t)n_blocks + 1), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return fail(e, "bsi: upload");
    for (uint32_t b = 0; b < n_blocks; b++) {      // per block: gather (the kernel's y index = the reference's id WITHIN the block), one stable sort - within a k-mer the entries stay in (reference, contig, position) order -, the bucket table
        const uint32_t r0 = b << BSI_BLOG, r1 = std::min<uint32_t>(n, (b + 1) << BSI_BLOG);
        const size_t o = (size_t)base[b]; const uint32_t cnt = (uint32_t)(base[b + 1] - base[b]);
        if (cnt) {
            hipLaunchKernelGGL(gsi_gather_kernel, dim3(std::max(1u, std::min(64u, (maxn + 4095u) / 4096u)), r1 - r0), dim3(256), 0, st, (const GsiSeg*)(T + o_s) + r0, (uint32_t*)(T + o_k), (unsigned long long*)(T + o_v));
            size_t ts_b = ts;
            e = hipcub::DeviceRadixSort::SortPairs(T + o_t, ts_b, (const uint32_t*)(T + o_k), (uint32_t*)db->bsi_key.p + o, (const unsigned long long*)(T + o_v), (unsigned long long*)db->bsi_val.p + o, (int)cnt, 0, kbits, st);
            if (e != hipSuccess) return fail(e, "bsi: sort");
        }
        hipLaunchKernelGGL(bsi_bucket_kernel, dim3(std::max(1u, (cnt + 255u) / 256u)), dim3(256), 0, st, (const uint32_t*)db->bsi_key.p + o, cnt, kbits - bits, nb, (uint32_t*)db->bsi_bucket.p + (size_t)b * (nb + 1), 0u);
    }
    e = hipStreamSynchronize(st);      // (segs, base and tmp die with this frame)
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
