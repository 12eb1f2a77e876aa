// Marker screen, seed-index lookup, chunked chaining and ANI/AF reduction on gfx950.
// Replaces, for Database::query (/root/reference/src/pyskani/_skani/lib.rs:569-659):
//   skani::screen::check_markers_quickly   (call site lib.rs:623-628)  -> screen_kernel
//   skani::chain::chain_seeds              (call site lib.rs:652-653)  -> anchor_* / chunk_* /
//                                                                        chain_chunk / pair_reduce
// Semantics are normative in oracle/skani_oracle.c (orc_screen / orc_chain).
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <algorithm>

// ------------------------------------------------------------------ screen
struct MarkerSet { const uint64_t* p; uint32_t n; uint32_t pad; };

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

psk_status screen_impl(psk_db* db, const psk_sketch* q, double screen_val, int rescue_small, uint8_t* pass, uint32_t* shared) {
    psk_ctx* ctx = db->ctx;
    const uint32_t n = (uint32_t)db->refs.size();
    if (n == 0) return PSK_OK;
    hipStream_t st = ctx->stream;
    if (db->tables_dirty) {
        std::vector<MarkerSet> h(n);
        for (uint32_t i = 0; i < n; i++) {
            const psk_sketch* r = db->refs[i];
            h[i].p = r->store ? r->store->markers + r->marker_off : nullptr;
            h[i].n = (uint32_t)r->n_markers; h[i].pad = 0;
        }
        PSK_TRY(db->d_marker_ptr.reserve(sizeof(MarkerSet) * n));
        PSK_HIP(hipMemcpyAsync(db->d_marker_ptr.p, h.data(), sizeof(MarkerSet) * n, hipMemcpyHostToDevice, st));
        PSK_HIP(hipStreamSynchronize(st));
        db->tables_dirty = false;
    }
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

// ------------------------------------------------------------------ anchors
struct RefIndex { const uint64_t* key; const uint64_t* pm; uint32_t n; uint32_t pad; };   // key = slot<<32 | kmer

// one thread per (pair, query seed): range of equal k-mers in the ref index
__global__ __launch_bounds__(256) void anchor_count_kernel(const RefIndex* __restrict__ refs, const uint32_t* __restrict__ q_kmer,
                                                           uint32_t nq, uint32_t* __restrict__ lb_out, uint32_t* __restrict__ cnt_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const RefIndex r = refs[blockIdx.y];
    uint32_t km = q_kmer[i];
    uint32_t lo = 0, hi = r.n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if ((uint32_t)r.key[mid] < km) lo = mid + 1; else hi = mid; }
    uint32_t cnt = 0;
    if (lo < r.n && (uint32_t)r.key[lo] == km) {
        uint32_t step = 1;
        while (lo + step < r.n && (uint32_t)r.key[lo + step] == km) step <<= 1;
        uint32_t a = lo + (step >> 1), b = lo + step < r.n ? lo + step : r.n;   // kmer[a]==km, kmer[b]!=km or b==n
        while (a + 1 < b) { uint32_t mid = (a + b) >> 1; if ((uint32_t)r.key[mid] == km) a = mid; else b = mid; }
        cnt = b - lo;
    }
    size_t o = (size_t)blockIdx.y * nq + i;
    lb_out[o] = lo; cnt_out[o] = cnt;
}

__global__ __launch_bounds__(256) void anchor_emit_kernel(const RefIndex* __restrict__ refs, const uint32_t* __restrict__ q_pos,
                                                          const uint32_t* __restrict__ q_meta, uint32_t nq,
                                                          const uint32_t* __restrict__ lb, const uint32_t* __restrict__ cnt,
                                                          const uint32_t* __restrict__ aoff,
                                                          uint32_t* __restrict__ a_qp, uint32_t* __restrict__ a_qc,
                                                          uint32_t* __restrict__ a_rp, uint32_t* __restrict__ a_rm) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    size_t o = (size_t)blockIdx.y * nq + i;
    uint32_t c = cnt[o];
    if (c == 0) return;
    const RefIndex r = refs[blockIdx.y];
    uint32_t l = lb[o], dst = aoff[o];
    uint32_t qp = q_pos[i], qm = q_meta[i];
    for (uint32_t j = 0; j < c; j++) {
        uint64_t pm = r.pm[l + j];
        uint32_t rmeta = (uint32_t)pm;
        a_qp[dst + j] = qp; a_qc[dst + j] = qm >> 1;
        a_rp[dst + j] = (uint32_t)(pm >> 32);
        a_rm[dst + j] = (rmeta & ~1u) | ((rmeta ^ qm) & 1u);   // ref contig << 1 | reverse_match
    }
}

// nxt[a] = first anchor of the same pair that starts a new chunk if a chunk starts at a
__global__ __launch_bounds__(256) void anchor_next_kernel(const uint32_t* __restrict__ a_qp, const uint32_t* __restrict__ a_qc,
                                                          const uint32_t* __restrict__ aoff, uint32_t nq, uint32_t n_pairs,
                                                          uint32_t total, uint32_t* __restrict__ nxt) {
    uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= total) return;
    uint32_t lo = 0, hi = n_pairs - 1;   // pair p owns [aoff[p*nq], aoff[(p+1)*nq])
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (aoff[(size_t)mid * nq] <= a) lo = mid; else hi = mid - 1; }
    uint32_t pend = aoff[(size_t)(lo + 1) * nq];
    uint64_t key = ((uint64_t)a_qc[a] << 32) + (uint64_t)a_qp[a] + FRAGMENT_LENGTH;   // first b with (qc,qp) > key
    uint32_t l = a + 1, h = pend;
    while (l < h) { uint32_t mid = (l + h) >> 1; uint64_t k2 = ((uint64_t)a_qc[mid] << 32) | a_qp[mid]; if (k2 <= key) l = mid + 1; else h = mid; }
    nxt[a] = l;
}

__global__ void chunk_heads_kernel(const uint32_t* __restrict__ aoff, const uint32_t* __restrict__ nxt, uint32_t nq,
                                   uint32_t n_pairs, uint32_t max_chunks, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks,
                                   uint32_t* __restrict__ err) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    uint32_t h = aoff[(size_t)p * nq], pend = aoff[(size_t)(p + 1) * nq], n = 0;
    while (h < pend) {
        uint32_t e = nxt[h];
        if (n < max_chunks) chunks[(size_t)p * max_chunks + n] = make_uint2(h, e); else atomicOr(err, 1u);
        n++; h = e;
    }
    n_chunks[p] = n < max_chunks ? n : max_chunks;
}

// ------------------------------------------------------------------ chaining
struct ChunkOut { uint32_t anchors, seeds, n_intervals, flags; uint64_t cov_q, cov_r; };

struct ChainArgs {
    const uint32_t *a_qp, *a_qc, *a_rp, *a_rm;
    const uint2* chunks; const uint32_t* n_chunks; uint32_t max_chunks, n_pairs;
    const uint32_t* q_seed_pos;      // store base
    const uint32_t* q_contig_start;  // query's slice of contig_seed_start (global seed offsets)
    ChunkOut* out;
    // serial-path scratch, one entry per anchor
    int32_t* sc_f; uint32_t *sc_ptr, *sc_root, *sc_depth, *sc_best;
    int32_t* c_score; uint32_t *c_q0, *c_q1, *c_r0, *c_r1, *c_n, *c_state;
    uint32_t two_c; int force_serial;
    uint32_t* stats;   // [0] fast chunks, [1] serial chunks
};

constexpr int RING = 128;   // power of two > CHAIN_BAND
constexpr int RMAX = 256;   // chain trees per chunk handled in LDS
constexpr int CHAIN_WAVES = 4;

__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// number of query seeds on contig qc with pos in [lo, hi]
__device__ uint32_t seeds_between(const ChainArgs& A, uint32_t qc, uint32_t lo, uint32_t hi) {
    uint32_t a = A.q_contig_start[qc], b = A.q_contig_start[qc + 1];
    uint32_t l = a, r = b;
    while (l < r) { uint32_t m = (l + r) >> 1; if (A.q_seed_pos[m] < lo) l = m + 1; else r = m; }
    uint32_t first = l; r = b;
    while (l < r) { uint32_t m = (l + r) >> 1; if (A.q_seed_pos[m] <= hi) l = m + 1; else r = m; }
    return l - first;
}

// Serial restatement of the oracle's per-chunk body, run by ONE lane on global scratch. Used for
// chunks the LDS path cannot hold (many chain trees / candidates) and as an in-GPU cross-check.
__device__ void chain_chunk_serial(const ChainArgs& A, uint32_t s, uint32_t e, ChunkOut& o) {
    for (uint32_t x = s; x < e; x++) {
        int32_t bs = ANCHOR_SCORE; uint32_t bp = x;
        uint32_t qx = A.a_qp[x], rx = A.a_rp[x], mx = A.a_rm[x];
        for (uint32_t y = x; y-- > s && x - y <= (uint32_t)CHAIN_BAND;) {
            if (A.a_rm[y] != mx) continue;
            int64_t dq = (int64_t)qx - (int64_t)A.a_qp[y];
            if (dq > BP_CHAIN_BAND) break;
            int64_t dr = (mx & 1) ? (int64_t)A.a_rp[y] - (int64_t)rx : (int64_t)rx - (int64_t)A.a_rp[y];
            if (dq <= 0 || dr <= 0) continue;
            int64_t gap = dq > dr ? dq - dr : dr - dq;
            if (gap > MAX_GAP_LENGTH) continue;
            int32_t sc = A.sc_f[y] + ANCHOR_SCORE - (int32_t)gap;
            if (sc > bs) { bs = sc; bp = y; }
        }
        A.sc_f[x] = bs; A.sc_ptr[x] = bp;
        if (bp == x) { A.sc_root[x] = x; A.sc_depth[x] = 1; }
        else { A.sc_root[x] = A.sc_root[bp]; A.sc_depth[x] = A.sc_depth[bp] + 1; }
        A.sc_best[x] = 0xFFFFFFFFu;
    }
    for (uint32_t x = s; x < e; x++) { uint32_t rt = A.sc_root[x]; uint32_t b = A.sc_best[rt]; if (b == 0xFFFFFFFFu || A.sc_f[x] > A.sc_f[b]) A.sc_best[rt] = x; }
    uint32_t nc = 0;
    for (uint32_t x = s; x < e; x++) {
        if (A.sc_root[x] != x) continue;
        uint32_t b = A.sc_best[x];
        if (A.sc_depth[b] < MIN_ANCHORS || A.sc_f[b] < MIN_SCORE) continue;
        uint32_t ra = A.a_rp[x], rb = A.a_rp[b];
        A.c_score[s + nc] = A.sc_f[b]; A.c_q0[s + nc] = A.a_qp[x]; A.c_q1[s + nc] = A.a_qp[b];
        A.c_r0[s + nc] = ra < rb ? ra : rb; A.c_r1[s + nc] = ra < rb ? rb : ra; A.c_n[s + nc] = A.sc_depth[b]; A.c_state[s + nc] = 0;
        nc++;
    }
    // greedy selection: repeatedly take the pending candidate with the highest score (lowest order on ties)
    uint32_t anch = 0, left = 0xFFFFFFFFu, right = 0, nk = 0; uint64_t cq = 0, cr = 0;
    for (uint32_t it = 0; it < nc; it++) {
        int32_t best = -1; uint32_t bi = 0;
        for (uint32_t i = 0; i < nc; i++) if (A.c_state[s + i] == 0 && A.c_score[s + i] > best) { best = A.c_score[s + i]; bi = i; }
        bool ok = true;
        for (uint32_t j = 0; j < nc && ok; j++) if (A.c_state[s + j] == 1 && !(A.c_q1[s + bi] < A.c_q0[s + j] || A.c_q0[s + bi] > A.c_q1[s + j])) ok = false;
        A.c_state[s + bi] = ok ? 1 : 2;
        if (ok) {
            uint32_t q0 = A.c_q0[s + bi], q1 = A.c_q1[s + bi];
            anch += A.c_n[s + bi]; nk++;
            if (q0 < left) left = q0;
            if (q1 > right) right = q1;
            cq += (uint64_t)(q1 - q0) + 1 + A.two_c;
            cr += (uint64_t)(A.c_r1[s + bi] - A.c_r0[s + bi]) + 1 + A.two_c;
        }
    }
    o.anchors = anch; o.n_intervals = nk; o.cov_q = cq; o.cov_r = cr; o.flags = 1;
    o.seeds = nk ? seeds_between(A, A.a_qc[s], left, right) : 0;
}

__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_chunk_kernel(ChainArgs A) {
    typedef hipcub::WarpReduce<uint32_t, 64> WR;
    __shared__ typename WR::TempStorage s_wr[CHAIN_WAVES];
    __shared__ uint32_t s_ring[CHAIN_WAVES][6][RING];        // qp, rp, rm, f, root id, depth
    __shared__ unsigned long long s_best[CHAIN_WAVES][RMAX]; // f<<28 | (16383-local idx)<<14 | depth
    __shared__ uint32_t s_rootx[CHAIN_WAVES][RMAX];          // local index of each tree's root anchor
    __shared__ uint32_t s_cand[CHAIN_WAVES][6][64];          // score, q0, q1, r0, r1, nanch
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t slot = blockIdx.x * CHAIN_WAVES + wave;
    const uint32_t pair = slot / A.max_chunks, ck = slot % A.max_chunks;
    if (pair >= A.n_pairs || ck >= A.n_chunks[pair]) return;
    const uint2 se = A.chunks[(size_t)pair * A.max_chunks + ck];
    const uint32_t s = se.x, e = se.y, n = e - s;
    ChunkOut* op = &A.out[(size_t)pair * A.max_chunks + ck];
    uint32_t (*ring)[RING] = s_ring[wave];
    bool fast = !A.force_serial && n < 16384;
    uint32_t R = 0;
    if (fast) {
        for (uint32_t base = s; base < e && fast; base += 64) {
            const uint32_t idx = base + lane;
            const bool have = idx < e;
            const uint32_t my_qp = have ? A.a_qp[idx] : 0, my_rp = have ? A.a_rp[idx] : 0, my_rm = have ? A.a_rm[idx] : 0;
            const uint32_t cnt = e - base < 64 ? e - base : 64;
            for (uint32_t j = 0; j < cnt; j++) {
                const uint32_t x = base + j;
                const uint32_t qx = __builtin_amdgcn_readlane(my_qp, j), rx = __builtin_amdgcn_readlane(my_rp, j),
                               mx = __builtin_amdgcn_readlane(my_rm, j);
                const uint32_t avail = x - s;   // anchors before x in the chunk
                uint32_t key = 0;
                // the band may need two sweeps of 64 predecessors; the second only if the 65th is still in bp range
                int sweeps = 1;
                if (avail > 64 && qx - ring[0][(x - 65) & (RING - 1)] <= (uint32_t)BP_CHAIN_BAND) sweeps = 2;
                for (int sw = 0; sw < sweeps; sw++) {
                    const uint32_t dist = lane + 1 + 64 * sw;
                    if (dist <= avail && dist <= (uint32_t)CHAIN_BAND) {
                        const uint32_t sl = (x - dist) & (RING - 1);
                        const uint32_t qy = ring[0][sl], ry = ring[1][sl], my = ring[2][sl];
                        const int32_t fy = (int32_t)ring[3][sl];
                        const int32_t dq = (int32_t)(qx - qy);
                        const int32_t dr = (mx & 1) ? (int32_t)(ry - rx) : (int32_t)(rx - ry);
                        const int32_t gap = dq > dr ? dq - dr : dr - dq;
                        const int32_t sc = fy + ANCHOR_SCORE - gap;
                        if (my == mx && dq > 0 && dq <= BP_CHAIN_BAND && dr > 0 && gap <= MAX_GAP_LENGTH && sc > ANCHOR_SCORE) {
                            uint32_t k2 = ((uint32_t)sc << 7) | (127u - dist);   // max score, then nearest predecessor
                            key = k2 > key ? k2 : key;
                        }
                    }
                }
                uint32_t best = WR(s_wr[wave]).Reduce(key, hipcub::Max());
                best = __builtin_amdgcn_readfirstlane(best);
                int32_t f = ANCHOR_SCORE; uint32_t rid, dep;
                if (best) {
                    f = (int32_t)(best >> 7);
                    const uint32_t sl = (x - (127u - (best & 127u))) & (RING - 1);
                    rid = ring[4][sl]; dep = ring[5][sl] + 1;
                } else {
                    rid = R++; dep = 1;
                    if (rid >= RMAX) { fast = false; break; }
                    if (lane == 0) { s_rootx[wave][rid] = avail; s_best[wave][rid] = 0; }
                }
                if (lane == 0) {
                    const uint32_t sl = x & (RING - 1);
                    ring[0][sl] = qx; ring[1][sl] = rx; ring[2][sl] = mx; ring[3][sl] = (uint32_t)f; ring[4][sl] = rid; ring[5][sl] = dep;
                    unsigned long long k64 = ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - avail) << 14) | dep;
                    if (k64 > s_best[wave][rid]) s_best[wave][rid] = k64;
                }
                lds_wave_sync();
            }
        }
    }
    uint32_t C = 0;
    if (fast) {
        // candidates: one per chain tree whose best anchor passes the thresholds, in root order
        for (uint32_t r0 = 0; r0 < R && fast; r0 += 64) {
            const uint32_t r = r0 + lane;
            bool qual = false; uint32_t f = 0, lx = 0, dep = 0;
            if (r < R) {
                unsigned long long bk = s_best[wave][r];
                f = (uint32_t)(bk >> 28); lx = 16383u - (uint32_t)((bk >> 14) & 16383u); dep = (uint32_t)(bk & 16383u);
                qual = dep >= MIN_ANCHORS && (int32_t)f >= MIN_SCORE;
            }
            unsigned long long bal = __ballot(qual);
            uint32_t ci = C + __popcll(bal & ((1ull << lane) - 1));
            C += __popcll(bal);
            if (C > 64) { fast = false; break; }
            if (qual) {
                uint32_t xr = s + s_rootx[wave][r], xb = s + lx;
                uint32_t ra = A.a_rp[xr], rb = A.a_rp[xb];
                s_cand[wave][0][ci] = f; s_cand[wave][1][ci] = A.a_qp[xr]; s_cand[wave][2][ci] = A.a_qp[xb];
                s_cand[wave][3][ci] = ra < rb ? ra : rb; s_cand[wave][4][ci] = ra < rb ? rb : ra; s_cand[wave][5][ci] = dep;
            }
        }
    }
    if (!fast) {
        if (lane == 0) {
            ChunkOut o{};
            chain_chunk_serial(A, s, e, o);
            *op = o;
            atomicAdd(&A.stats[1], 1u);
        }
        return;
    }
    lds_wave_sync();
    const bool mine = (uint32_t)lane < C;
    const uint32_t c_sc = mine ? s_cand[wave][0][lane] : 0, c_q0 = mine ? s_cand[wave][1][lane] : 0, c_q1 = mine ? s_cand[wave][2][lane] : 0,
                   c_r0 = mine ? s_cand[wave][3][lane] : 0, c_r1 = mine ? s_cand[wave][4][lane] : 0, c_n = mine ? s_cand[wave][5][lane] : 0;
    // greedy non-overlapping selection by (score desc, order asc); one candidate per lane
    unsigned long long pending = C >= 64 ? ~0ull : ((1ull << C) - 1), keptm = 0;
    while (pending) {
        uint32_t key = ((pending >> lane) & 1) ? ((c_sc << 6) | (63u - lane)) : 0;   // score < 2^19, order < 64
        uint32_t best = WR(s_wr[wave]).Reduce(key, hipcub::Max());
        best = __builtin_amdgcn_readfirstlane(best);
        const uint32_t w = 63u - (best & 63u);
        const uint32_t wq0 = __builtin_amdgcn_readlane(c_q0, w), wq1 = __builtin_amdgcn_readlane(c_q1, w);
        const bool ov = ((keptm >> lane) & 1) && !(wq1 < c_q0 || wq0 > c_q1);
        if (__ballot(ov) == 0) keptm |= 1ull << w;
        pending &= ~(1ull << w);
    }
    const bool kept = (keptm >> lane) & 1;
    // per-chunk totals
    uint32_t anch = kept ? c_n : 0, left = kept ? c_q0 : 0xFFFFFFFFu, right = kept ? c_q1 : 0;
    unsigned long long cq = kept ? (unsigned long long)(c_q1 - c_q0) + 1 + A.two_c : 0, cr = kept ? (unsigned long long)(c_r1 - c_r0) + 1 + A.two_c : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        anch += __shfl_xor(anch, o);
        uint32_t l2 = __shfl_xor(left, o), r2 = __shfl_xor(right, o);
        left = l2 < left ? l2 : left; right = r2 > right ? r2 : right;
        cq += __shfl_xor(cq, o); cr += __shfl_xor(cr, o);
    }
    if (lane == 0) {
        ChunkOut o{};
        o.anchors = anch; o.n_intervals = (uint32_t)__popcll(keptm); o.cov_q = cq; o.cov_r = cr; o.flags = 0;
        o.seeds = o.n_intervals ? seeds_between(A, A.a_qc[s], left, right) : 0;
        *op = o;
        atomicAdd(&A.stats[0], 1u);
    }
}

// ------------------------------------------------------------------ per-pair ANI / AF
struct ReduceArgs {
    const ChunkOut* chunks; const uint32_t* n_chunks; uint32_t max_chunks;
    const uint32_t* aoff; uint32_t nq;
    const uint64_t* ref_total_len;   // per pair
    uint64_t q_total_len;
    int k, median, robust; double min_af;
    psk_hit* hits;
};
constexpr int RED_CAP = 4096;   // chunk ANI values sortable in LDS (genomes up to ~80 Mb at 20 kb chunks)

__global__ __launch_bounds__(256) void pair_reduce_kernel(ReduceArgs R) {
    __shared__ double s_v[RED_CAP];
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_acc[5];
    const uint32_t p = blockIdx.x;
    const uint32_t nc = R.n_chunks[p];
    const ChunkOut* co = R.chunks + (size_t)p * R.max_chunks;
    if (threadIdx.x == 0) { s_n = 0; for (int i = 0; i < 5; i++) s_acc[i] = 0; }
    __syncthreads();
    // integer totals (order-free)
    unsigned long long t_cq = 0, t_cr = 0, t_a = 0, t_s = 0, t_i = 0;
    for (uint32_t i = threadIdx.x; i < nc; i += blockDim.x) { t_cq += co[i].cov_q; t_cr += co[i].cov_r; t_a += co[i].anchors; t_s += co[i].seeds; t_i += co[i].n_intervals; }
    atomicAdd(&s_acc[0], t_cq); atomicAdd(&s_acc[1], t_cr); atomicAdd(&s_acc[2], t_a); atomicAdd(&s_acc[3], t_s); atomicAdd(&s_acc[4], t_i);
    // chunk ANI values, compacted in chunk order (serial prefix by thread 0 keeps the oracle's summation order)
    __shared__ uint32_t s_idx[RED_CAP];
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (uint32_t i = 0; i < nc; i++) if (co[i].n_intervals) { if (m < RED_CAP) s_idx[m] = i; m++; }
        s_n = m;
    }
    __syncthreads();
    const uint32_t m = s_n;
    psk_hit h{};
    h.ani = -1.0f;
    const bool overflow = m > RED_CAP;
    double mean_serial = 0;   // only thread 0 uses it
    if (!overflow) {
        for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) {
            const ChunkOut c = co[s_idx[j]];
            double ratio = (double)c.anchors / (double)c.seeds;
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
    } else if (threadIdx.x == 0 && !(R.median || R.robust)) {   // very long genomes: stream the mean in chunk order
        double sum = 0; uint32_t cnt = 0;
        for (uint32_t i = 0; i < nc; i++) if (co[i].n_intervals) {
            double ratio = (double)co[i].anchors / (double)co[i].seeds; if (ratio > 1.0) ratio = 1.0;
            sum += pow(ratio, 1.0 / (double)R.k); cnt++;
        }
        mean_serial = sum / (double)cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        h.ref_index = p;
        h.n_chunks = m; h.n_intervals = (uint32_t)s_acc[4];
        h.n_anchors = R.aoff[(size_t)(p + 1) * R.nq] - R.aoff[(size_t)p * R.nq];
        h.covered_query = s_acc[0]; h.covered_ref = s_acc[1]; h.sum_chain_anchors = s_acc[2]; h.sum_chunk_seeds = s_acc[3];
        if (m > 0) {
            double ani;
            bool ok = true;
            if (overflow) { if (R.median || R.robust) { ok = false; ani = -2.0; } else ani = mean_serial; }
            else if (R.median) ani = s_v[m / 2];
            else {
                uint32_t lo = 0, hi = m;
                if (R.robust && m - 2 * (m / 10) > 0) { lo = m / 10; hi = m - m / 10; }
                double sum = 0; for (uint32_t i = lo; i < hi; i++) sum += s_v[i];
                ani = sum / (double)(hi - lo);
            }
            double afq = (double)s_acc[0] / (double)R.q_total_len; if (afq > 1) afq = 1;
            double afr = (double)s_acc[1] / (double)R.ref_total_len[p]; if (afr > 1) afr = 1;
            h.af_query = (float)afq; h.af_ref = (float)afr;
            if (!ok) h.ani = -2.0f;
            else if (afq >= R.min_af || afr >= R.min_af) h.ani = (float)ani;
        }
        R.hits[p] = h;
    }
}

// ------------------------------------------------------------------ host orchestration
static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static psk_status chain_batch(psk_ctx* ctx, const psk_sketch* const* refs, uint32_t n_pairs, const psk_sketch* q,
                              const psk_query_opts* o, psk_hit* out) {
    hipStream_t st = ctx->stream;
    const uint32_t nq = (uint32_t)q->n_seeds;
    const int force_serial = getenv("PSK_CHAIN_SERIAL") != nullptr;
    // max chunks per pair: chunk heads on one contig are more than FRAGMENT_LENGTH apart
    uint64_t max_chunks64 = 0;
    for (uint32_t len : q->contig_len) max_chunks64 += (uint64_t)len / (FRAGMENT_LENGTH + 1) + 1;
    const uint32_t max_chunks = (uint32_t)max_chunks64;
    if (nq == 0 || max_chunks == 0) {
        for (uint32_t p = 0; p < n_pairs; p++) { out[p] = psk_hit{}; out[p].ani = -1.0f; }
        return PSK_OK;
    }
    // ---- tables ----
    std::vector<RefIndex> h_refs(n_pairs);
    std::vector<uint64_t> h_rlen(n_pairs);
    for (uint32_t p = 0; p < n_pairs; p++) {
        const psk_sketch* r = refs[p];
        h_refs[p].key = r->idx ? r->idx->key + r->idx_off : nullptr;
        h_refs[p].pm = r->idx ? r->idx->pm + r->idx_off : nullptr;
        h_refs[p].n = (uint32_t)r->n_seeds; h_refs[p].pad = 0;
        h_rlen[p] = r->total_len;
    }
    const size_t npq = (size_t)n_pairs * nq;
    size_t o_refs = 0, o_rlen = al256(o_refs + sizeof(RefIndex) * n_pairs), o_lb = al256(o_rlen + 8 * (size_t)n_pairs),
           o_cnt = al256(o_lb + 4 * npq), o_aoff = al256(o_cnt + 4 * (npq + 1)), o_nch = al256(o_aoff + 4 * (npq + 1)),
           o_chunks = al256(o_nch + 4 * (size_t)n_pairs), o_cout = al256(o_chunks + sizeof(uint2) * (size_t)n_pairs * max_chunks),
           o_hits = al256(o_cout + sizeof(ChunkOut) * (size_t)n_pairs * max_chunks), o_misc = al256(o_hits + sizeof(psk_hit) * n_pairs),
           o_end = o_misc + 64;
    PSK_TRY(ctx->q_b.reserve(o_end));
    char* B = (char*)ctx->q_b.p;
    RefIndex* d_refs = (RefIndex*)(B + o_refs); uint64_t* d_rlen = (uint64_t*)(B + o_rlen);
    uint32_t* d_lb = (uint32_t*)(B + o_lb); uint32_t* d_cnt = (uint32_t*)(B + o_cnt); uint32_t* d_aoff = (uint32_t*)(B + o_aoff);
    uint32_t* d_nch = (uint32_t*)(B + o_nch); uint2* d_chunks = (uint2*)(B + o_chunks); ChunkOut* d_cout = (ChunkOut*)(B + o_cout);
    psk_hit* d_hits = (psk_hit*)(B + o_hits); uint32_t* d_misc = (uint32_t*)(B + o_misc);   // [0] err, [1..2] stats
    PSK_HIP(hipMemcpyAsync(d_refs, h_refs.data(), sizeof(RefIndex) * n_pairs, hipMemcpyHostToDevice, st));
    PSK_HIP(hipMemcpyAsync(d_rlen, h_rlen.data(), 8 * (size_t)n_pairs, hipMemcpyHostToDevice, st));
    PSK_HIP(hipMemsetAsync(d_misc, 0, 64, st));
    PSK_HIP(hipMemsetAsync(d_cnt + npq, 0, 4, st));
    const uint32_t* q_kmer = q->store->seed_kmer + q->seed_off;
    const uint32_t* q_pos = q->store->seed_pos + q->seed_off;
    const uint32_t* q_meta = q->store->seed_meta + q->seed_off;
    dim3 g2((nq + 255) / 256, n_pairs);
    ctx->t_begin(K_ANCHOR);
    hipLaunchKernelGGL(anchor_count_kernel, g2, dim3(256), 0, st, d_refs, q_kmer, nq, d_lb, d_cnt);
    ctx->t_end();
    size_t tmp = 0;
    PSK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, d_cnt, d_aoff, (int)(npq + 1), st));
    PSK_TRY(ctx->q_c.reserve(tmp));
    PSK_HIP(hipcub::DeviceScan::ExclusiveSum(ctx->q_c.p, tmp, d_cnt, d_aoff, (int)(npq + 1), st));
    void* hp;
    PSK_TRY(ctx->pinned(sizeof(psk_hit) * n_pairs + 256, &hp));
    uint32_t* h_small = (uint32_t*)hp;
    PSK_HIP(hipMemcpyAsync(h_small, d_aoff + npq, 4, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipStreamSynchronize(st));
    const uint32_t total = h_small[0];
    // ---- anchors + serial-path scratch: 4 + 5 + 7 arrays of u32 per anchor ----
    const size_t na = (size_t)total + 64;
    PSK_TRY(ctx->q_d.reserve(4 * na * 16));
    uint32_t* D = (uint32_t*)ctx->q_d.p;
    uint32_t *a_qp = D, *a_qc = D + na, *a_rp = D + 2 * na, *a_rm = D + 3 * na, *a_nxt = D + 4 * na;
    ChainArgs A{};
    A.a_qp = a_qp; A.a_qc = a_qc; A.a_rp = a_rp; A.a_rm = a_rm;
    A.sc_f = (int32_t*)(D + 5 * na); A.sc_ptr = D + 6 * na; A.sc_root = D + 7 * na; A.sc_depth = D + 8 * na; A.sc_best = D + 9 * na;
    A.c_score = (int32_t*)(D + 10 * na); A.c_q0 = D + 11 * na; A.c_q1 = D + 12 * na; A.c_r0 = D + 13 * na; A.c_r1 = D + 14 * na; A.c_n = D + 15 * na;
    A.c_state = a_nxt;   // nxt is dead once the chunk table exists
    A.chunks = d_chunks; A.n_chunks = d_nch; A.max_chunks = max_chunks; A.n_pairs = n_pairs;
    A.q_seed_pos = q->store->seed_pos; A.q_contig_start = q->store->contig_seed_start + q->contig_off;
    A.out = d_cout; A.two_c = 2u * (uint32_t)q->params.c; A.force_serial = force_serial; A.stats = d_misc + 1;
    if (total > 0) {
        hipLaunchKernelGGL(anchor_emit_kernel, g2, dim3(256), 0, st, d_refs, q_pos, q_meta, nq, d_lb, d_cnt, d_aoff, a_qp, a_qc, a_rp, a_rm);
        hipLaunchKernelGGL(anchor_next_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a_qp, a_qc, d_aoff, nq, n_pairs, total, a_nxt);
    }
    hipLaunchKernelGGL(chunk_heads_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, st, d_aoff, a_nxt, nq, n_pairs, max_chunks, d_chunks, d_nch, d_misc);
    const uint32_t slots = n_pairs * max_chunks;
    ctx->t_begin(K_CHAIN_CHUNK);
    hipLaunchKernelGGL(chain_chunk_kernel, dim3((slots + CHAIN_WAVES - 1) / CHAIN_WAVES), dim3(64 * CHAIN_WAVES), 0, st, A);
    ctx->t_end();
    ReduceArgs R{};
    R.chunks = d_cout; R.n_chunks = d_nch; R.max_chunks = max_chunks; R.aoff = d_aoff; R.nq = nq; R.ref_total_len = d_rlen;
    R.q_total_len = q->total_len; R.k = q->params.k; R.median = o->median; R.robust = o->robust;
    R.min_af = o->min_aligned_frac > 0 ? o->min_aligned_frac : 0.15; R.hits = d_hits;
    ctx->t_begin(K_PAIR_REDUCE);
    hipLaunchKernelGGL(pair_reduce_kernel, dim3(n_pairs), dim3(256), 0, st, R);
    ctx->t_end();
    psk_hit* h_hits = (psk_hit*)((char*)hp + 256);
    PSK_HIP(hipMemcpyAsync(h_hits, d_hits, sizeof(psk_hit) * n_pairs, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipMemcpyAsync(h_small, d_misc, 16, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipStreamSynchronize(st));
    if (h_small[0]) { psk_set_error("internal: chunk table overflow"); return PSK_EHIP; }
    for (uint32_t p = 0; p < n_pairs; p++) {
        out[p] = h_hits[p];
        if (out[p].ani == -2.0f) { psk_set_error("median/robust ANI needs <= %d chunks per pair (genome too long)", RED_CAP); return PSK_ELIMIT; }
    }
    return PSK_OK;
}

psk_status chain_impl(psk_ctx* ctx, const psk_sketch* const* refs, uint32_t n_refs, const psk_sketch* q,
                      const psk_query_opts* o, psk_hit* out) {
    if (!ctx || !q || !o || (!refs && n_refs) || (!out && n_refs)) { psk_set_error("chain: NULL argument"); return PSK_EINVAL; }
    if (o->learned_ani == 1) { psk_set_error("learned ANI requested but no regression model is loaded (skani's GBDT weights are not redistributable here)"); return PSK_ENOMODEL; }
    if (!q->has_seeds) { psk_set_error("query sketch was built with seed=False; it cannot be chained"); return PSK_EINVAL; }
    for (uint32_t i = 0; i < n_refs; i++) {
        if (!refs[i] || !refs[i]->has_seeds) { psk_set_error("reference %u was sketched with seed=False; it cannot be chained", i); return PSK_EINVAL; }
        if (refs[i]->params.k != q->params.k || refs[i]->params.c != q->params.c) { psk_set_error("reference %u and query were sketched with different parameters", i); return PSK_EINVAL; }
    }
    PSK_TRY(ensure_index(ctx, refs, n_refs));
    // bound the scratch of one launch: lb/cnt/aoff cost 12 B per (pair, query seed)
    const uint64_t nq = q->n_seeds ? q->n_seeds : 1;
    uint32_t per = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(2048, (1ull << 31) / (nq * 2)));
    for (uint32_t b = 0; b < n_refs; b += per) {
        uint32_t nb = std::min(per, n_refs - b);
        PSK_TRY(chain_batch(ctx, refs + b, nb, q, o, out + b));
        for (uint32_t i = 0; i < nb; i++) out[b + i].ref_index = b + i;
    }
    return PSK_OK;
}
