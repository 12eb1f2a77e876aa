// Database.query / query_many (lib.rs:549-660): screen rounds, the plan (which join, which batches), two batches in flight, hit selection.
#include "chain_stages.h"
#include "query_parts.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <deque>
#include <thread>

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

// The same prefilter where the database has no database-wide index (more than 65 536 references: its entries carry 16-bit reference ids): one wave per (rescued query, block
// of the blocked index), the block's 256 counters in LDS, a lane per query seed.
__global__ __launch_bounds__(64) void bsi_prefilter_kernel(const SketchDesc* __restrict__ qd, const uint32_t* __restrict__ rq, uint32_t n_refs,
                                                           const uint32_t* __restrict__ b_key, const unsigned long long* __restrict__ b_val, const uint32_t* __restrict__ b_bucket,
                                                           const unsigned long long* __restrict__ b_base, int b_shift, uint32_t b_nb1, uint32_t blk0, uint8_t* __restrict__ pass) {
    __shared__ uint32_t s_c[1u << BSI_BLOG];
    const uint32_t q = rq[blockIdx.x], blk = blk0 + blockIdx.y, lane = threadIdx.x;
    const SketchDesc Q = qd[q];
    for (uint32_t i = lane; i < (1u << BSI_BLOG); i += 64u) s_c[i] = 0;
    lds_wave_sync();
    const unsigned long long base = b_base[blk];
    const uint32_t* __restrict__ key = b_key + base; const unsigned long long* __restrict__ val = b_val + base; const uint32_t* __restrict__ bkt = b_bucket + (size_t)blk * b_nb1;
    for (uint32_t i = lane; i < Q.n; i += 64u) {
        const uint32_t km = Q.kmer[i], b = km >> b_shift;
        for (uint32_t x = bkt[b], hi = bkt[b + 1]; x < hi; x++)
            if (key[x] == km) { const uint32_t ref = (uint32_t)(val[x] >> 48) & ((1u << BSI_BLOG) - 1u); if (s_c[ref] < MIN_ANCHORS) atomicAdd(&s_c[ref], 1u); }
    }
    lds_wave_sync();
    uint8_t* row = pass + (size_t)q * n_refs + ((size_t)blk << BSI_BLOG);
    for (uint32_t r = lane; r < (1u << BSI_BLOG); r += 64u) if (((size_t)blk << BSI_BLOG) + r < n_refs && s_c[r] < MIN_ANCHORS) row[r] = 0;
}

uint64_t index_stamp(const psk_db* db) {
    uint64_t v = 0;
    for (const psk_sketch* r : db->refs) v += (uint64_t)(r->idx != nullptr) + ((uint64_t)(r->ptab != nullptr) << 32);
    return v;
}
psk_status refresh_ref_descs(Lane* ctx, psk_db* db) {
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
    const Switches sw = Switches::read();      // ($PSK_*: once per call; the helper threads of the two-lane mode read this copy)
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
    const uint32_t qb_env = sw.round_queries.get() ? (uint32_t)std::max(1, atoi(sw.round_queries.get())) : 0u;
    const uint32_t QB = std::max<uint32_t>(1, std::min<uint32_t>(qb_env ? qb_env : 65536u, (uint32_t)((1ull << 30) / n)));
    {
        const char* force = sw.screen.get();
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
        PSK_TRY(screen_many_device(ctx, db, queries + b, m, screen_val, !o->faster_small, d_pass, keep, sw));
        if (db->has_dups) hipLaunchKernelGGL(pass_canon_kernel, dim3(m), dim3(256), 0, st, d_pass, n, (const uint32_t*)db->d_canon.p);
        // ---- rescued short queries: exact anchor counts against every reference, pairs that cannot chain leave the pass matrix
        PoolScratch pf_buf;      // lives until the round's synchronisations are through, like the host arrays the copies read
        std::vector<uint32_t> rq, eoff, qn;
        {
            const char* pf_env = sw.prefilter.get();      // "0": never; "1": whatever the number of pairs (tests)
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
            const bool gsi_pf_off = sw.gsi_join.get() && sw.gsi_join.get()[0] == '0';      // (read per call: tests switch it within a process)
            if (refs_ok && !gsi_pf_off && !sw.join_wide()) {
                if (db->gsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_gsi(ctx, db); }));
                if (db->gsi_state != 1 && db->bsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_bsi(ctx, db); }));      // (no database-wide index - more than 65 536 references -: the blocked one)
                if (db->gsi_state == 1 || db->bsi_state == 1) {
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
                    if (db->gsi_state == 1)
                    hipLaunchKernelGGL(gsi_prefilter_kernel, dim3(nr), dim3(256), 4 * (size_t)((n + 1) / 2), st, (const SketchDesc*)(Bp + o_qd), (const uint32_t*)Bp, n,
                                       (const uint32_t*)db->gsi_key.p, (const unsigned long long*)db->gsi_val.p, (const uint32_t*)db->gsi_bucket.p, db->gsi_shift, d_pass);
                    else
                    for (uint32_t b0 = 0; b0 < db->bsi_blocks; b0 += 32768u)      // (grid.y holds 65 535)
                        hipLaunchKernelGGL(bsi_prefilter_kernel, dim3(nr, std::min<uint32_t>(32768u, db->bsi_blocks - b0)), dim3(64), 0, st, (const SketchDesc*)(Bp + o_qd), (const uint32_t*)Bp, n,
                                           (const uint32_t*)db->bsi_key.p, (const unsigned long long*)db->bsi_val.p, (const uint32_t*)db->bsi_bucket.p,
                                           (const unsigned long long*)db->bsi_base.p, db->bsi_shift, db->bsi_nb1, b0, d_pass);
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
        const double max_blocks_join = sw.gsl_max_blocks.get() ? atof(sw.gsl_max_blocks.get()) : 4.0;
        {
            const char* pb_env = sw.probe.get();
            const bool pb_off = pb_env && pb_env[0] == '0', pb_force = pb_env && pb_env[0] == '1';
            want_small = !pb_off && (pb_force || (round_pairs >= 16384 && round_items / round_pairs < 2048));
            const bool gsi_join_off = sw.gsi_join.get() && sw.gsi_join.get()[0] == '0';      // (read per round: tests switch it within a process; chain_run follows the plan)
            // Rounds of many MID-SIZED pairs (all-vs-all of ~5 Mb genomes: every query passes against its family) go through the same index by (query, slice) waves
            // (slice_join.hip) instead of one merge join per pair: one lookup per query SEED where the per-pair join makes one per (pair, seed). PSK_GSI_SLICE=0 never,
            // =1 whatever the round's shape (tests, A/B)
            const char* sl_env = sw.gsi_slice.get();      // (read per round: tests switch it within a process)
            const bool sl_off = sl_env && sl_env[0] == '0', sl_force = sl_env && sl_env[0] == '1';
            const bool want_slice = !want_small && !sl_off && (sl_force || (round_pairs >= 2048 && round_items / round_pairs >= 2048 && round_items / round_pairs <= (1u << 18)));
            const double max_blocks = sw.gsl_max_blocks.get() ? atof(sw.gsl_max_blocks.get()) : 4.0;
            uint64_t q_with = 0; for (uint32_t i = 0; i < m; i++) q_with += h_cnt[i] != 0;
            const bool few_blocks = (double)round_blocks <= max_blocks * (double)std::max<uint64_t>(q_with, 1);
            if (want_small && !gsi_join_off && !sw.join_wide()) {
                // contigs: through the index in blocks of references when their passing references sit in few of them (a contig's relatives - what the marker screen and the
                // prefilter of rescued contigs leave), through the database-wide index otherwise (a rescued contig against EVERY reference: one walk instead of one per block)
                const bool bsi_small_off = sw.bsi_small.get() && sw.bsi_small.get()[0] == '0';      // (tests, A/B)
                if (db->gsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_gsi(ctx, db); }));
                round_gsi = db->gsi_state == 1;
                if (!bsi_small_off || !round_gsi) {      // (every wave of the join chooses by its own query: both indexes are handed over; a database beyond the database-wide index's 65 536 references has the blocked one alone)
                    if (db->bsi_state == 0) PSK_TRY(exclusive([&]() -> psk_status { return build_bsi(ctx, db); }));
                    round_bsi = db->bsi_state == 1;
                    round_gsi = round_gsi || round_bsi;
                }
            }
            // (the slice join walks, per query, the index BLOCKS that hold one of its passing references: worth it while those are few - relatives that sit next to each
            // other in the database; a query whose references are scattered over many blocks would walk its seeds once per block: PSK_GSL_MAX_BLOCKS, default 4 on average)
            if (want_slice && !gsi_join_off && !sw.join_wide() && (sl_force || few_blocks)) {
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
        const int items_env = sw.batch_items_log2.get() ? std::min(31, std::max(16, atoi(sw.batch_items_log2.get()))) : 0;
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
                if (lane2) { Scratch* big[] = {&lane2->lane->q_b, &lane2->lane->q_c, &lane2->lane->q_d, &lane2->lane->q_e, &lane2->lane->q_g, &lane2->lane->q_j, &lane2->lane->q_k, &lane2->lane->q_sel}; (void)hipStreamSynchronize(lane2->lane->stream); for (Scratch* s : big) s->release(); }
                (void)hipMemGetInfo(&free_b, &total_b);
            }
            if (free_b + ctx->q_d.cap + ctx->q_e.cap > need) items_log2 = 28;
        }
        // Rounds of many small pairs (contigs): up to 2^22 pairs and 2^30 seeds per batch. The probe join visits a batch's pairs reference by reference, and a line of a
        // reference's table is probed about once per 2^20 pairs of a 5 000-reference database: with twice the pairs every line is probed twice while it is still
        // cached (join 142 -> 124 ms per 100 000 contigs). PSK_BATCH_PAIRS_LOG2 overrides (tests, A/B).
        const int pairs_env = sw.batch_pairs_log2.get() ? std::min(24, std::max(10, atoi(sw.batch_pairs_log2.get()))) : 0;
        if (!items_env && items_log2 == 29 && round_probe) items_log2 = 30;
        // ... and rounds of mid-sized pairs joined by (query, slice) waves: a launch of 10 000 waves is three waves deep on the chip and its last third runs half empty;
        // 2^30 seeds (268 genomes of 5 Mb and their ~27 000 pairs) per batch: 860 -> 796 ms per 10 000 x 10 000 step (2^28: 969)
        if (!items_env && items_log2 == 29 && round_slice) items_log2 = 30;
        uint64_t max_items = 1ull << items_log2, max_pairs = 1ull << (pairs_env ? pairs_env : (round_probe ? 22 : 21)), max_rows = 1ull << 26;      // (2^22 pairs: 363 -> 353 ms per 100 000 contigs)
        {   // the one-pass index join lays a batch's anchors out at 9/8 of its items (gsi_room_kernel) where about two thirds of that are used: three quarters of the
            // items per batch keep the per-anchor arrays (100 bytes per slot) near what the two passes reserved
            const bool one_off = sw.gsi_onepass.get() && sw.gsi_onepass.get()[0] == '0';
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
            const int pipe_env = sw.pipeline.get() ? atoi(sw.pipeline.get()) : -1;      // (read per round: bench.py takes its kernel table from a step run as one chain)
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
                        L.g_key = (const uint32_t*)db->bsi_key.p; L.g_val = (const unsigned long long*)db->bsi_val.p; L.g_bucket = (const uint32_t*)db->bsi_bucket.p; L.g_shift = db->bsi_shift; L.g_nb1 = db->bsi_nb1; L.g_blocks = db->bsi_blocks; L.g_base = (const unsigned long long*)db->bsi_base.p;
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
                        bool wide = sw.join_wide();
                        for (int attempt = 0;; attempt++) {
                            psk_status rrc = chain_run(ln, L, n_pairs, (size_t)J.items, (size_t)J.rows, db->params, o, d_qd, d_rd, cap, wide, sw, false);
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
                    if (round_slice) { L.g_key = (const uint32_t*)db->bsi_key.p; L.g_val = (const unsigned long long*)db->bsi_val.p; L.g_bucket = (const uint32_t*)db->bsi_bucket.p; L.g_shift = db->bsi_shift; L.g_nb1 = db->bsi_nb1; L.g_blocks = db->bsi_blocks; L.g_base = (const unsigned long long*)db->bsi_base.p; }
                    else {
                        if (db->gsi_state == 1) { L.g_key = (const uint32_t*)db->gsi_key.p; L.g_val = (const unsigned long long*)db->gsi_val.p; L.g_bucket = (const uint32_t*)db->gsi_bucket.p; L.g_shift = db->gsi_shift; }
                        if (round_bsi) { L.b_key = (const uint32_t*)db->bsi_key.p; L.b_val = (const unsigned long long*)db->bsi_val.p; L.b_bucket = (const uint32_t*)db->bsi_bucket.p; L.b_shift = db->bsi_shift; L.b_nb1 = db->bsi_nb1; L.b_blocks = db->bsi_blocks; L.b_base = (const unsigned long long*)db->bsi_base.p; L.b_max = (uint32_t)std::max(1.0, max_blocks_join); }
                    }
                    L.d_pass = d_pass; L.n_refs = n; L.n_bq = (uint32_t)bqs.size();
                    uint32_t pm = 1; for (const BatchQ& e : bqs) pm = std::max(pm, e.rank_hi - e.rank_lo);
                    L.p_cap = round_slice ? (pm + 15u) & ~15u : (pm + 63u) & ~63u;      // (the slice join's LDS arrays are indexed by pair alone: no need for whole waves of them)
                    const bool one_off = sw.gsi_onepass.get() && sw.gsi_onepass.get()[0] == '0';      // tests, A/B: count pass + scan + emit pass
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
                bool too_big = false, wide = sw.join_wide();
                static const bool trace_batch = getenv("PSK_TRACE_BATCH") != nullptr;      // diagnostics: host wall clock of every batch (launching, waiting)
                for (int attempt = 0;; attempt++) {
                    struct timespec tb0{}, tb1{}, tb2{};
                    if (trace_batch) clock_gettime(CLOCK_MONOTONIC, &tb0);
                    psk_status rrc = chain_run(ctx, L, n_pairs, (size_t)items, (size_t)rows, db->params, o, d_qd, (const SketchDesc*)db->d_refdesc.p, cap, wide, sw, round_probe && !round_gsi);
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
