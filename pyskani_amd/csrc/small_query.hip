// Database.query() of ONE small genome from host bytes in one launch sequence and one synchronisation
// (the call SURVEY.md 8(d) config 4 specifies: a contig per query; /root/reference/src/pyskani/_skani/lib.rs:549-660 with the
// `_sketch` call of lib.rs:571 inside it).
//
// The general path (psk_sketch_host + query_many_impl) is built for batches: for one 10 kb contig it is ~26 dependent launches of
// 4-5 us kernels and three host round trips, with the GPU ~95 % idle (profiles/r3/r3q_query_timeline.txt). Here the whole call is
//   ONE upload     contig table + tile tables + ASCII (pinned staging)
//   sketch_scan    (sketch.hip, unchanged)
//   sketch_emit    with the tile offsets formed by the emit waves themselves
//   sq_screen      ONE workgroup: raw markers -> sorted distinct set (LDS bitonic), inverted-index screen with the per-reference
//                  counters in LDS, pass rule, duplicate-name rule, ORDERED shortlist
//   sq_chain       one workgroup per shortlisted reference, everything in LDS: index join -> anchors in (q contig, q pos, r contig,
//                  r pos) order -> chunk table -> banded DP (block sweep, below) -> candidate chains -> greedy selection -> per-chunk
//                  identities -> ANI / AF record
//   ONE download   status block + hit records
// Sizes the host does not know (seeds, markers, shortlist length) stay on the device: every kernel reads them from the status block.
// A call that exceeds a capacity (SQ_* in common.h) raises a flag and is rerun on the general path. Results are the general path's,
// bit for bit (tests/test_gpu_small_query.py: against the oracle and against PSK_SMALL_QUERY=0).
#include "common.h"
#include "chain_dev.h"
#include <cmath>
#include <cstddef>
#include <algorithm>

namespace {

constexpr int SQ_SCREEN_T = 1024;
constexpr uint32_t SQ_TREES = 128;       // chain trees per chunk (per wave) in LDS

struct SqScreenArgs {
    const uint64_t* mstage; const uint32_t* toff; const uint32_t* tmc; uint32_t n_tiles;
    SmallQHead* head; uint64_t* markers_out;
    const MarkerSet* refs; const uint64_t* inv_key; const uint32_t* inv_val; const uint32_t* inv_bucket; int inv_shift; uint32_t inv_n;
    uint32_t n_refs; double thresh; int rescue_small; const uint32_t* canon;
    uint32_t* shortlist;
    // database-wide seed index (null: none) and the query's seed k-mers: the seed prefilter of a rescued contig
    const uint32_t* gsi_key; const unsigned long long* gsi_val; const uint32_t* gsi_bucket; int gsi_shift; const uint32_t* q_kmer;
};

// exclusive scan over the workgroup's threads (blockDim = SQ_SCREEN_T), total to every thread
__device__ __forceinline__ uint32_t sq_block_scan(uint32_t v, uint32_t* s_w, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc, o); if (lane >= o) inc += x; }
    __syncthreads();
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t before = 0, tot = 0;
    for (int w = 0; w < SQ_SCREEN_T / 64; w++) { const uint32_t c = s_w[w]; if (w < wave) before += c; tot += c; }
    *total = tot;
    return before + inc - v;
}

__global__ __launch_bounds__(SQ_SCREEN_T) void sq_screen_kernel(SqScreenArgs A) {
    __shared__ unsigned long long s_m[SQ_MARKERS];      // raw markers, sorted in place
    __shared__ unsigned long long s_u[SQ_MARKERS];      // the distinct ones
    __shared__ uint32_t s_tsrc[SQ_MAX_TILES], s_tdst[SQ_MAX_TILES + 1];
    __shared__ uint32_t s_w[SQ_SCREEN_T / 64];
    extern __shared__ uint32_t s_count[];               // one counter per reference
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    SmallQHead* H = A.head;
    if (H->flags & SQ_F_SEEDS) { if (tid == 0) { H->n_short = 0; H->n_markers = 0; H->n_markers_raw = 0; } return; }      // the emit waves wrote nothing reliable
    // ---- the tiles' raw markers, gathered in tile order ----
    uint32_t n_raw;
    {
        const uint32_t c = tid < A.n_tiles ? A.tmc[tid] : 0u;
        const uint32_t ex = sq_block_scan(c, s_w, &n_raw);
        if (tid < A.n_tiles) { s_tsrc[tid] = A.toff[tid]; s_tdst[tid] = ex; }
        if (tid == 0) s_tdst[A.n_tiles] = n_raw;
    }
    if (n_raw > SQ_MARKERS) { if (tid == 0) { atomicOr(&H->flags, SQ_F_MARKERS); H->n_short = 0; H->n_markers = 0; H->n_markers_raw = n_raw; } return; }
    __syncthreads();
    uint32_t P = 2; while (P < n_raw) P <<= 1;
    for (uint32_t d = tid; d < P; d += SQ_SCREEN_T) {
        unsigned long long v = ~0ull;
        if (d < n_raw) {
            uint32_t lo = 0, hi = A.n_tiles;                          // last tile q with s_tdst[q] <= d
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_tdst[mid] <= d) lo = mid; else hi = mid; }
            v = A.mstage[s_tsrc[lo] + (d - s_tdst[lo])];
        }
        s_m[d] = v;
    }
    __syncthreads();
    // ---- bitonic sort of P <= 2 048 keys: at most one compare-exchange pair per thread and stage ----
    for (uint32_t kk = 2; kk <= P; kk <<= 1)
        for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
            if (tid < P / 2) {
                const uint32_t t = ((tid & ~(jj - 1u)) << 1) | (tid & (jj - 1u));
                const unsigned long long a = s_m[t], b = s_m[t | jj];
                const bool asc = (t & kk) == 0;
                if ((a > b) == asc) { s_m[t] = b; s_m[t | jj] = a; }
            }
            __syncthreads();
        }
    // ---- distinct values, in order ----
    uint32_t n_mark;
    {
        const uint32_t i0 = 2 * tid;
        const bool f0 = i0 < n_raw && (i0 == 0 || s_m[i0] != s_m[i0 - 1]), f1 = i0 + 1 < n_raw && s_m[i0 + 1] != s_m[i0];
        uint32_t rank = sq_block_scan((uint32_t)f0 + (uint32_t)f1, s_w, &n_mark);
        if (f0) { s_u[rank] = s_m[i0]; A.markers_out[rank] = s_m[i0]; rank++; }
        if (f1) { s_u[rank] = s_m[i0 + 1]; A.markers_out[rank] = s_m[i0 + 1]; }
    }
    for (uint32_t r = tid; r < A.n_refs; r += SQ_SCREEN_T) s_count[r] = 0;
    __syncthreads();
    // ---- shared markers per reference through the inverted index: a wave per marker, four markers in flight (inv_screen_wave_kernel's lookup) ----
    if (A.inv_n) {
        constexpr int U = 4;
        for (uint32_t i0 = wave * U; i0 < n_mark; i0 += (SQ_SCREEN_T / 64) * U) {
            unsigned long long m[U]; uint32_t lo[U], hi[U];
#pragma unroll
            for (int u = 0; u < U; u++) m[u] = i0 + u < n_mark ? s_u[i0 + u] : ~0ull;
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t b = (uint32_t)(m[u] >> A.inv_shift); lo[u] = 0; hi[u] = 0; if (i0 + u < n_mark) { lo[u] = A.inv_bucket[b]; hi[u] = A.inv_bucket[b + 1]; } }
            unsigned long long k[U];
#pragma unroll
            for (int u = 0; u < U; u++) k[u] = lo[u] + lane < hi[u] ? A.inv_key[lo[u] + lane] : ~0ull;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (lo[u] + lane < hi[u] && k[u] == m[u]) atomicAdd(&s_count[A.inv_val[lo[u] + lane]], 1u);
                for (uint32_t x = lo[u] + 64u + lane; x < hi[u]; x += 64u) if (A.inv_key[x] == m[u]) atomicAdd(&s_count[A.inv_val[x]], 1u);
            }
        }
    }
    __syncthreads();
    // ---- check_markers_quickly's rule (lib.rs:623-628), then the shortlist of NAMES (lib.rs:629-637: a passing entry stands for its name's last sketch) ----
    for (uint32_t r = tid; r < A.n_refs; r += SQ_SCREEN_T) {
        const uint32_t b = A.refs[r].n, small = n_mark < b ? n_mark : b;
        uint32_t ok;
        if (A.rescue_small && small < SMALL_MARKER_COUNT) ok = 1;
        else if (small == 0) ok = 0;
        else ok = ((double)s_count[r] / (double)small) > A.thresh;
        s_count[r] = ok;
    }
    __syncthreads();
    if (A.canon) {
        for (uint32_t r = tid; r < A.n_refs; r += SQ_SCREEN_T) { const uint32_t cr = A.canon[r]; if (s_count[r] && cr != r) { s_count[cr] = 1; s_count[r] = 0; } }      // canon[r] > r, canon[canon[r]] == canon[r]
        __syncthreads();
    }
    // ---- a RESCUED contig (fewer than 20 markers, lib.rs:538-541) has passed against every reference, yet a pair with fewer than MIN_ANCHORS shared
    // seeds cannot chain and never yields a hit: the exact anchor count of every reference, one lookup per query seed in the database-wide seed index
    // (an anchor = a (query seed, reference seed) pair of equal k-mer), takes those pairs off the shortlist - 5 000 chain workgroups become ~100 ----
    if (A.gsi_key && A.rescue_small && n_mark < SMALL_MARKER_COUNT) {
        const uint32_t nq = H->n_seeds;
        constexpr int U = 4;
        for (uint32_t i0 = wave * U; i0 < nq; i0 += (SQ_SCREEN_T / 64) * U) {
            uint32_t km[U], lo[U], hi[U];
#pragma unroll
            for (int u = 0; u < U; u++) km[u] = i0 + u < nq ? A.q_kmer[i0 + u] : 0u;
#pragma unroll
            for (int u = 0; u < U; u++) { lo[u] = 0; hi[u] = 0; if (i0 + u < nq) { const uint32_t b = km[u] >> A.gsi_shift; lo[u] = A.gsi_bucket[b]; hi[u] = A.gsi_bucket[b + 1]; } }
#pragma unroll
            for (int u = 0; u < U; u++)
                for (uint32_t x = lo[u] + lane; x < hi[u]; x += 64u) if (A.gsi_key[x] == km[u]) atomicAdd(&s_count[(uint32_t)(A.gsi_val[x] >> 48)], 2u);      // (bit 0: the pass flag)
        }
        __syncthreads();
        for (uint32_t r = tid; r < A.n_refs; r += SQ_SCREEN_T) { const uint32_t v = s_count[r]; s_count[r] = (v & 1u) && (v >> 1) >= MIN_ANCHORS ? 1u : 0u; }
        __syncthreads();
    }
    uint32_t run = 0;
    for (uint32_t base = 0; base < A.n_refs; base += SQ_SCREEN_T) {
        const uint32_t r = base + tid;
        const uint32_t f = r < A.n_refs && s_count[r] ? 1u : 0u;
        uint32_t tot;
        const uint32_t ex = sq_block_scan(f, s_w, &tot);
        if (f) A.shortlist[run + ex] = r;
        run += tot;
    }
    if (tid == 0) { H->n_markers_raw = n_raw; H->n_markers = n_mark; H->n_short = run; }
}

// ------------------------------------------------------------------ the fused chain kernel
struct SqChainArgs {
    const SmallQHead* head; SmallQHead* head_w; const uint32_t* shortlist;
    const uint32_t* q_kmer; const uint32_t* q_pos; const uint32_t* q_meta;
    const SketchDesc* rd;
    psk_hit* hits;
    unsigned long long q_total_len; uint32_t n_desc;
    uint32_t band, two_c; int k, median, robust; double min_af;
    unsigned long long* prof;      // diagnostics ($PSK_SQ_PROFILE): 100 MHz timestamps of the first pair's phases
    int no_team;                   // $PSK_SQ_TEAM=0: every chunk by one wave (tests, A/B)
    uint4* host_out;               // pinned host block [status | hits]: written by the LAST workgroup to finish (no download command); null: the host copies
};

constexpr int SQ_CHAIN_T = 256;
__device__ __forceinline__ uint32_t sq_rl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
struct SqWin { uint32_t q1, u, m; int32_t f1; uint32_t id, dp; };      // one block of 64 anchors, an anchor per lane: q + 1, diagonal, ref contig | strand, score - 1 (lane_eval2's form), tree, depth

// Banded chaining DP of one chunk by ONE wave, 64 anchors at a time ("block sweep"). The recurrence is sequential in the anchors, and a wave that takes them one
// by one (chain_wave_reg_kernel) pays a wave-wide maximum and a handful of lane reads per anchor: ~500 cycles each. Here an anchor sits in its own lane and the
// PREDECESSORS are what is stepped through:
//   (1) predecessors in earlier blocks - their scores are final - are broadcast one per step out of the two previous blocks' registers; the steps are
//       independent of each other;
//   (2) inside the block, step j broadcasts anchor j - final once steps 0 .. j-1 are through - to the lanes after it. The only value on the dependent chain
//       is anchor j's score (one lane read, a scalar max and a subtraction feeding lane_eval2's last four instructions).
// (3) Tree and depth follow from the chosen predecessors by pointer doubling inside the block (six shuffle rounds) instead of one lane read per anchor.
// Same keys (score, then nearest predecessor), same tree numbering (roots in anchor order) as chain_chunk_row: the candidates are identical.
__device__ void sq_chain_chunk(const uint4* __restrict__ s_anc, const uint32_t s, const uint32_t e, const uint32_t band, unsigned long long* __restrict__ best,
                               uint32_t* __restrict__ root, const int lane, uint32_t& R_out, bool& over) {
    SqWin w0{0, 0, 0xFFFFFFFFu, 0, 0, 0}, w1 = w0;
    uint32_t R = 0;
    for (uint32_t B = s; B < e; B += 64) {
        const uint32_t cnt = __builtin_amdgcn_readfirstlane(e - B < 64u ? e - B : 64u);
        const bool have = (uint32_t)lane < cnt;
        const uint4 a = have ? s_anc[B + lane] : make_uint4(0u, 0u, 0xFFFFFFFEu, 0u);
        const uint32_t qx = a.x, mx = a.z, ux = lane_diag(qx, a.y, 0u - (mx & 1u)), q1 = qx + 1u;
        int32_t key = 0;
        const uint32_t avail = __builtin_amdgcn_readfirstlane(B - s);                     // anchors before the block (a multiple of 64)
        const uint32_t nst = avail < band ? avail : band;
        // predecessors more than BP_CHAIN_BAND bases before the block's FIRST anchor serve no lane of it: positions ascend with the index, so the
        // ones in reach are the last lanes of the previous blocks - counted once per block, not tested per step
        const uint32_t q_first = sq_rl(qx, 0);
        const uint32_t reach0 = (uint32_t)__popcll(__ballot((int32_t)(q_first - w0.q1) < BP_CHAIN_BAND && w0.m != 0xFFFFFFFFu));
        const uint32_t reach1 = (uint32_t)__popcll(__ballot((int32_t)(q_first - w1.q1) < BP_CHAIN_BAND && w1.m != 0xFFFFFFFFu));
        const uint32_t n0 = __builtin_amdgcn_readfirstlane(nst < 64u ? (nst < reach0 ? nst : reach0) : reach0);
        const uint32_t n1 = __builtin_amdgcn_readfirstlane(reach0 < 64u || nst <= 64u ? 64u : (nst < 64u + reach1 ? nst : 64u + reach1));
        for (uint32_t t = 0; t < n0; t++) {                                                // (1) the block before this one: its lane 63 - t is at distance lane + 1 + t
            const uint32_t pl = 63u - t;
            const LanePred y{sq_rl(w0.q1, pl), sq_rl(w0.u, pl), sq_rl(w0.m, pl), (int32_t)sq_rl((uint32_t)w0.f1, pl)};
            const uint32_t d = (uint32_t)lane + 1u + t;
            const int32_t kx = lane_eval2(qx, ux, mx, y, (int)(d & 127u)) | (int32_t)((band - d) & 0x80000000u);
            key = kx > key ? kx : key;
        }
        for (uint32_t t = 64; t < n1; t++) {                                               //     and the one before that (bands beyond 64 anchors)
            const uint32_t pl = 127u - t;
            const LanePred y{sq_rl(w1.q1, pl), sq_rl(w1.u, pl), sq_rl(w1.m, pl), (int32_t)sq_rl((uint32_t)w1.f1, pl)};
            const uint32_t d = (uint32_t)lane + 1u + t;
            const int32_t kx = lane_eval2(qx, ux, mx, y, (int)(d & 127u)) | (int32_t)((band - d) & 0x80000000u);
            key = kx > key ? kx : key;
        }
        for (uint32_t j = 0; j + 1 < cnt; j++) {                                           // (2) the sweep
            const int32_t kj = (int32_t)sq_rl((uint32_t)key, j);
            const int32_t fj1 = (kj > 0 ? (kj >> 7) : ANCHOR_SCORE2) - 1;
            const LanePred y{sq_rl(q1, j), sq_rl(ux, j), sq_rl(mx, j), fj1};
            const uint32_t d = (uint32_t)lane - j;                                          // 1 .. band for the lanes this predecessor can serve
            const int32_t kx = lane_eval2(qx, ux, mx, y, (int)(d & 127u)) | (int32_t)(((band - d) | (d - 1u)) & 0x80000000u);
            key = kx > key ? kx : key;
        }
        // (3) score, tree and depth of every anchor of the block
        const bool isroot = key <= 0;
        const int32_t f = isroot ? ANCHOR_SCORE2 : (key >> 7);
        const uint32_t dd = 127u - ((uint32_t)key & 127u);                                  // distance of the chosen predecessor
        const unsigned long long rb = __ballot(have && isroot);
        const bool outside = !isroot && dd > (uint32_t)lane;                                // the predecessor sits in an earlier block
        const uint32_t back = dd - (uint32_t)lane, sl = (64u - back) & 63u;                  // ... `back` anchors before this block: lane 64 - back of the last one, 128 - back of the one before
        const uint32_t id0 = __shfl(w0.id, (int)sl), dp0 = __shfl(w0.dp, (int)sl), id1 = __shfl(w1.id, (int)sl), dp1 = __shfl(w1.dp, (int)sl);
        uint32_t t_id = 0, t_dp = 0, ptr = (uint32_t)lane, dist = 0;
        if (isroot) { t_id = R + (uint32_t)__popcll(rb & ((1ull << lane) - 1)); t_dp = 1; }
        else if (outside) { t_id = back <= 64u ? id0 : id1; t_dp = (back <= 64u ? dp0 : dp1) + 1u; }
        else { ptr = (uint32_t)lane - dd; dist = 1; }
#pragma unroll
        for (int r = 0; r < 6; r++) { const uint32_t np = __shfl(ptr, (int)ptr), nd = __shfl(dist, (int)ptr); dist += nd; ptr = np; }
        const uint32_t id = __shfl(t_id, (int)ptr), dp = __shfl(t_dp, (int)ptr) + dist;
        R += (uint32_t)__popcll(rb);
        if (R > SQ_TREES) { over = true; break; }
        if (have && isroot) root[t_id] = avail + (uint32_t)lane;
        if (have) atomicMax(&best[id], ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - (avail + (uint32_t)lane)) << 14) | dp);
        w1 = w0;
        w0 = SqWin{q1, ux, mx, f - 1, id, dp};
    }
    R_out = R;
}


// The same DP by a TEAM of T = 2 or 4 waves of the workgroup (pairs of one or two chunks leave three or two of its waves - and of the CU's SIMDs - idle).
// Part (1) is what a team can share: its steps are independent, so wave tw takes steps tw, tw + T, ... and leaves its partial keys in LDS; the team's
// first wave merges them and runs (2) and (3) alone. The state of the last 128 anchors (what (1) reads and what an anchor's tree follows from) lives in an
// LDS ring, one 16-byte broadcast read per step where the single-wave version reads four lanes. Every wave of the WORKGROUP calls this for the same number
// of blocks (`n_blocks`: the longest chunk of the round; a team without a block left only keeps the barriers company).
struct SqRing { uint4 p[128]; uint32_t id[128], dp[128]; };      // (q + 1, diagonal, ref contig | strand, score - 1), tree, depth by chunk-local index & 127
__device__ void sq_chain_team(const uint4* __restrict__ s_anc, const uint32_t s, const uint32_t e, const bool active, const uint32_t n_blocks, const uint32_t band,
                              SqRing& ring, int32_t (*__restrict__ pk)[64], const uint32_t T, const uint32_t tw, unsigned long long* __restrict__ best,
                              uint32_t* __restrict__ root, const int lane, uint32_t& R_out, bool& over) {
    uint32_t R = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        const uint32_t B = s + 64u * b;
        const bool act = active && !over && B < e;
        uint32_t cnt = 0, qx = 0, mx = 0xFFFFFFFEu, ux = 0, q1 = 0;
        const uint32_t avail = 64u * b;
        bool have = false;
        if (act) {
            cnt = __builtin_amdgcn_readfirstlane(e - B < 64u ? e - B : 64u);
            have = (uint32_t)lane < cnt;
            const uint4 a = have ? s_anc[B + lane] : make_uint4(0u, 0u, 0xFFFFFFFEu, 0u);
            qx = a.x; mx = a.z; ux = lane_diag(qx, a.y, 0u - (mx & 1u)); q1 = qx + 1u;
            const uint32_t nst = avail < band ? avail : band;
            // steps in reach of the block's first anchor (positions ascend with the index: the predecessors in reach are the nearest ones)
            const uint32_t q_first = sq_rl(qx, 0);
            const uint32_t t0 = (uint32_t)lane, t1 = (uint32_t)lane + 64u;
            const bool in0 = t0 < nst && (int32_t)(q_first - ring.p[(avail - 1u - t0) & 127u].x) < BP_CHAIN_BAND;
            const bool in1 = t1 < nst && (int32_t)(q_first - ring.p[(avail - 1u - t1) & 127u].x) < BP_CHAIN_BAND;
            const uint32_t n_steps = (uint32_t)__popcll(__ballot(in0)) + (uint32_t)__popcll(__ballot(in1));
            int32_t key = 0;
            for (uint32_t t = tw; t < n_steps; t += T) {                                       // (1) this wave's share of the earlier blocks' anchors
                const uint4 yp = ring.p[(avail - 1u - t) & 127u];
                const LanePred y{yp.x, yp.y, yp.z, (int32_t)yp.w};
                const uint32_t d = (uint32_t)lane + 1u + t;
                const int32_t kx = lane_eval2(qx, ux, mx, y, (int)(d & 127u)) | (int32_t)((band - d) & 0x80000000u);
                key = kx > key ? kx : key;
            }
            pk[tw][lane] = key;
        }
        __syncthreads();
        if (act && tw == 0) {
            int32_t key = pk[0][lane];
            for (uint32_t w = 1; w < T; w++) { const int32_t o = pk[w][lane]; key = o > key ? o : key; }
            for (uint32_t j = 0; j + 1 < cnt; j++) {                                           // (2) the sweep
                const int32_t kj = (int32_t)sq_rl((uint32_t)key, j);
                const int32_t fj1 = (kj > 0 ? (kj >> 7) : ANCHOR_SCORE2) - 1;
                const LanePred y{sq_rl(q1, j), sq_rl(ux, j), sq_rl(mx, j), fj1};
                const uint32_t d = (uint32_t)lane - j;
                const int32_t kx = lane_eval2(qx, ux, mx, y, (int)(d & 127u)) | (int32_t)(((band - d) | (d - 1u)) & 0x80000000u);
                key = kx > key ? kx : key;
            }
            const bool isroot = key <= 0;                                                     // (3) score, tree and depth
            const int32_t f = isroot ? ANCHOR_SCORE2 : (key >> 7);
            const uint32_t dd = 127u - ((uint32_t)key & 127u);
            const unsigned long long rb = __ballot(have && isroot);
            const bool outside = !isroot && dd > (uint32_t)lane;
            const uint32_t pidx = (avail + (uint32_t)lane - dd) & 127u;
            const uint32_t o_id = ring.id[pidx], o_dp = ring.dp[pidx];
            uint32_t t_id = 0, t_dp = 0, ptr = (uint32_t)lane, dist = 0;
            if (isroot) { t_id = R + (uint32_t)__popcll(rb & ((1ull << lane) - 1)); t_dp = 1; }
            else if (outside) { t_id = o_id; t_dp = o_dp + 1u; }
            else { ptr = (uint32_t)lane - dd; dist = 1; }
#pragma unroll
            for (int r = 0; r < 6; r++) { const uint32_t np = __shfl(ptr, (int)ptr), nd = __shfl(dist, (int)ptr); dist += nd; ptr = np; }
            const uint32_t id = __shfl(t_id, (int)ptr), dp = __shfl(t_dp, (int)ptr) + dist;
            R += (uint32_t)__popcll(rb);
            if (R > SQ_TREES) over = true;
            else {
                if (have && isroot) root[t_id] = avail + (uint32_t)lane;
                if (have) {
                    atomicMax(&best[id], ((unsigned long long)(uint32_t)f << 28) | ((unsigned long long)(16383u - (avail + (uint32_t)lane)) << 14) | dp);
                    const uint32_t me = (avail + (uint32_t)lane) & 127u;
                    ring.p[me] = make_uint4(q1, ux, mx, (uint32_t)(f - 1)); ring.id[me] = id; ring.dp[me] = dp;
                }
            }
        }
        __syncthreads();
    }
    R_out = R;
}

__global__ __launch_bounds__(SQ_CHAIN_T) void sq_chain_kernel(SqChainArgs A) {
    __shared__ uint4 s_anc[SQ_SEEDS];                       // anchors of the pair: (q pos, r pos, ref contig << 1 | reverse_match, q contig << 16 | index of the query seed)
    __shared__ unsigned long long s_best[SQ_CHAIN_T / 64][SQ_TREES];
    __shared__ uint32_t s_root[SQ_CHAIN_T / 64][SQ_TREES];
    __shared__ uint32_t c_sc[SQ_CANDS], c_q0[SQ_CANDS], c_q1[SQ_CANDS], c_r0[SQ_CANDS], c_r1[SQ_CANDS], c_rc[SQ_CANDS], c_row[SQ_CANDS], c_n[SQ_CANDS], c_ord[SQ_CANDS], c_i0[SQ_CANDS], c_i1[SQ_CANDS];
    __shared__ uint16_t s_prio[SQ_CANDS], s_kept[SQ_CANDS];
    __shared__ uint2 s_chunk[SQ_ROWS];
    __shared__ uint32_t r_anch[SQ_ROWS], r_nint[SQ_ROWS], r_left[SQ_ROWS], r_right[SQ_ROWS];
    __shared__ unsigned long long r_cov[SQ_ROWS];
    __shared__ double s_val[64];
    __shared__ uint32_t s_wt[4][4];
    __shared__ uint32_t s_nrows, s_ncand, s_over;
    __shared__ SqRing s_ring[2];
    __shared__ int32_t s_pk[SQ_CHAIN_T / 64][64];
    const uint32_t tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SmallQHead* H = A.head;
    const uint32_t nq = H->n_seeds;
    const uint32_t n_short = (H->flags & (SQ_F_SEEDS | SQ_F_MARKERS)) ? 0u : H->n_short;
    for (uint32_t pi = blockIdx.x; pi < n_short; pi += gridDim.x) {
        const uint32_t ref = A.shortlist[pi];
        const SketchDesc R = A.rd[ref];
        if (tid == 0) { s_ncand = 0; s_over = 0; s_nrows = 0; }
        __syncthreads();
        if (A.prof && pi == 0 && tid == 0) A.prof[0] = wall_clock64();
        // ---- A. join: every query seed looked up in the reference's k-mer index, anchors written in seed order (already (q contig, q pos, r contig, r pos) order) ----
        uint32_t run = 0;
        bool a_over = false;
        constexpr int U = 4;
        for (uint32_t base = 0; base < nq; base += SQ_CHAIN_T * U) {
            uint32_t iq[U], km[U], lo[U], hi[U], cnt[U], qpos[U], qmeta[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; u++) { iq[u] = base + u * SQ_CHAIN_T + tid; ok[u] = iq[u] < nq && R.n != 0; km[u] = ok[u] ? A.q_kmer[iq[u]] : 0u; qpos[u] = ok[u] ? A.q_pos[iq[u]] : 0u; qmeta[u] = ok[u] ? A.q_meta[iq[u]] : 0u; }
#pragma unroll
            for (int u = 0; u < U; u++) { lo[u] = 0; hi[u] = 0; if (ok[u]) { const uint32_t bk = km[u] >> R.bshift; lo[u] = R.bucket[bk]; hi[u] = R.bucket[bk + 1]; } }
            for (;;) {      // lower bound of km in [lo, hi), the four seeds of a thread in lockstep
                bool any = false;
                uint32_t mid[U], kv[U];
#pragma unroll
                for (int u = 0; u < U; u++) { mid[u] = (lo[u] + hi[u]) >> 1; any = any || lo[u] < hi[u]; }
                if (!__any(any)) break;
#pragma unroll
                for (int u = 0; u < U; u++) kv[u] = lo[u] < hi[u] ? R.key[mid[u]] : 0u;
#pragma unroll
                for (int u = 0; u < U; u++) if (lo[u] < hi[u]) { if (kv[u] < km[u]) lo[u] = mid[u] + 1; else hi[u] = mid[u]; }
            }
            uint32_t k0[U], k1[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                k0[u] = (ok[u] && lo[u] < R.n) ? R.key[lo[u]] : 0xFFFFFFFFu;
                k1[u] = (ok[u] && lo[u] + 1 < R.n) ? R.key[lo[u] + 1] : 0xFFFFFFFFu;
            }
            unsigned long long pm[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                cnt[u] = 0;
                if (ok[u] && k0[u] == km[u]) {
                    cnt[u] = 1;
                    if (k1[u] == km[u]) {      // a repeat (rare): gallop for the end of the run
                        uint32_t step = 2;
                        while (lo[u] + step < R.n && R.key[lo[u] + step] == km[u]) step <<= 1;
                        uint32_t a2 = lo[u] + (step >> 1), b2 = lo[u] + step < R.n ? lo[u] + step : R.n;
                        while (a2 + 1 < b2) { const uint32_t m2 = (a2 + b2) >> 1; if (R.key[m2] == km[u]) a2 = m2; else b2 = m2; }
                        cnt[u] = b2 - lo[u];
                    }
                }
                pm[u] = cnt[u] ? R.pms[lo[u]] : 0ull;
            }
            // anchor offsets: seed order = (u, thread) order within the round
            uint32_t inc[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                inc[u] = cnt[u];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t x = __shfl_up(inc[u], o); if (lane >= o) inc[u] += x; }
                if (lane == 63) s_wt[u][wave] = inc[u];
            }
            __syncthreads();
            uint32_t before = run;
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint32_t off = before;
                for (int w = 0; w < 4; w++) { const uint32_t c = s_wt[u][w]; if (w < wave) off += c; before += c; }
                off += inc[u] - cnt[u];
                if (cnt[u]) {
                    if (off + cnt[u] > SQ_SEEDS) a_over = true;
                    else {
                        const uint32_t qp = qpos[u], qm = qmeta[u];
                        for (uint32_t j = 0; j < cnt[u]; j++) {
                            const unsigned long long p2 = j ? R.pms[lo[u] + j] : pm[u];
                            const uint32_t rmeta = (uint32_t)p2;
                            s_anc[off + j] = make_uint4(qp, (uint32_t)(p2 >> 32), (rmeta & ~1u) | ((rmeta ^ qm) & 1u), ((qm >> 1) << 16) | iq[u]);
                        }
                    }
                }
            }
            run = before;
            __syncthreads();
        }
        if (a_over) s_over = 1;
        const uint32_t na = run;      // (the same in every thread)
        __syncthreads();
        bool skip = s_over != 0 || na < MIN_ANCHORS;
        if (A.prof && pi == 0 && tid == 0) A.prof[1] = wall_clock64();
        // ---- B. chunk table (wave 0): a chunk = the anchors of one query contig within FRAGMENT_LENGTH of its first anchor ----
        if (!skip && wave == 0) {
            uint32_t h = 0, n = 0;
            while (h < na) {
                const uint4 ah = s_anc[h];
                const unsigned long long limit = (((unsigned long long)(ah.w >> 16) << 32) | ah.x) + FRAGMENT_LENGTH;
                uint32_t sp = h + 1, b = na;
                while (sp < na) {
                    const uint32_t idx = sp + (uint32_t)lane;
                    bool past = false;
                    if (idx < na) { const uint4 ai = s_anc[idx]; past = ((((unsigned long long)(ai.w >> 16) << 32) | ai.x) > limit); }
                    const unsigned long long bal = __ballot(past);
                    if (bal) { b = sp + (uint32_t)__ffsll((long long)bal) - 1u; break; }
                    sp += 64;
                }
                if (n < SQ_ROWS) { if (lane == 0) { s_chunk[n] = make_uint2(h, b); r_anch[n] = 0; r_nint[n] = 0; r_left[n] = 0xFFFFFFFFu; r_right[n] = 0; r_cov[n] = 0ull; } }
                n++; h = b;
            }
            if (lane == 0) { s_nrows = n <= SQ_ROWS ? n : SQ_ROWS; if (n > SQ_ROWS) s_over = 1; }
        }
        __syncthreads();
        const uint32_t nrows = s_nrows;
        skip = skip || s_over != 0;
        if (A.prof && pi == 0 && tid == 0) A.prof[2] = wall_clock64();
        // ---- C. the chunks' DP; candidate chains of the pair in LDS. Three chunks or more: a wave per chunk. One or two (most contigs): a team of four or two waves per chunk ----
        auto emit_candidates = [&](const uint32_t row, const uint2 se, const unsigned long long* best, const uint32_t* root, const uint32_t Rn) {
            uint32_t crow = 0;
            for (uint32_t r0 = 0; r0 < Rn; r0 += 64) {      // one candidate per chain tree whose best anchor passes the thresholds, in root order
                const uint32_t r = r0 + (uint32_t)lane;
                bool qual = false; uint32_t f = 0, lx = 0, dep = 0;
                if (r < Rn) {
                    const unsigned long long bk = best[r];
                    f = (uint32_t)(bk >> 28); lx = 16383u - (uint32_t)((bk >> 14) & 16383u); dep = (uint32_t)(bk & 16383u);
                    qual = dep >= MIN_ANCHORS && (int32_t)f >= MIN_SCORE2;
                }
                const unsigned long long bal = __ballot(qual);
                if (!bal) continue;
                uint32_t slot0 = 0;
                if (lane == 0) slot0 = atomicAdd(&s_ncand, (uint32_t)__popcll(bal));
                slot0 = __shfl(slot0, 0);
                const uint32_t rk = (uint32_t)__popcll(bal & ((1ull << lane) - 1));
                if (slot0 + (uint32_t)__popcll(bal) > SQ_CANDS) { if (lane == 0) s_over = 1; break; }
                if (qual) {
                    const uint32_t ci = slot0 + rk;
                    const uint4 ar = s_anc[se.x + root[r]], ab = s_anc[se.x + lx];
                    c_sc[ci] = f; c_q0[ci] = ar.x; c_q1[ci] = ab.x; c_r0[ci] = ar.y < ab.y ? ar.y : ab.y; c_r1[ci] = ar.y < ab.y ? ab.y : ar.y;
                    c_n[ci] = dep; c_rc[ci] = ar.z >> 1; c_i0[ci] = ar.w & 0xFFFFu; c_i1[ci] = ab.w & 0xFFFFu; c_row[ci] = row; c_ord[ci] = (row << 16) | (crow + rk);      // generation order: rows, then trees
                }
                crow += (uint32_t)__popcll(bal);
            }
        };
        if (!skip && nrows <= 2 && !A.no_team) {
            const uint32_t T = nrows == 1 ? 4u : 2u, team = (uint32_t)wave / T, tw = (uint32_t)wave % T;
            const bool active = team < nrows;
            const uint2 se = active ? s_chunk[team] : make_uint2(0u, 0u);
            uint32_t n_blocks = 0;
            for (uint32_t r = 0; r < nrows; r++) { const uint2 x = s_chunk[r]; const uint32_t nb = (x.y - x.x + 63u) / 64u; n_blocks = nb > n_blocks ? nb : n_blocks; }
            unsigned long long* best = s_best[wave]; uint32_t* root = s_root[wave];      // (the team's first wave's tables)
            if (tw == 0) for (uint32_t i = lane; i < SQ_TREES; i += 64) best[i] = 0ull;
            uint32_t Rn = 0; bool over = false;
            sq_chain_team(s_anc, se.x, se.y, active, n_blocks, A.band, s_ring[team & 1u], &s_pk[team * T], T, tw, best, root, lane, Rn, over);
            if (tw == 0 && active) {
                lds_wave_sync();
                if (over) { if (lane == 0) s_over = 1; }
                else emit_candidates(team, se, best, root, Rn);
            }
        }
        else if (!skip) {
            for (uint32_t row = (uint32_t)wave; row < nrows; row += SQ_CHAIN_T / 64) {
                const uint2 se = s_chunk[row];
                unsigned long long* best = s_best[wave]; uint32_t* root = s_root[wave];
                for (uint32_t i = lane; i < SQ_TREES; i += 64) best[i] = 0ull;
                lds_wave_sync();
                uint32_t Rn = 0; bool over = false;
                sq_chain_chunk(s_anc, se.x, se.y, A.band, best, root, lane, Rn, over);
                lds_wave_sync();
                if (over) { if (lane == 0) s_over = 1; continue; }
                emit_candidates(row, se, best, root, Rn);
            }
        }
        __syncthreads();
        skip = skip || s_over != 0;
        const uint32_t C = skip ? 0u : (s_ncand < SQ_CANDS ? s_ncand : SQ_CANDS);
        if (A.prof && pi == 0 && tid == 0) A.prof[3] = wall_clock64();
        // ---- D. greedy selection over ALL candidates of the pair by (score desc, generation order): kept unless it overlaps a kept chain on the
        //         query (same chunk) or on the reference (same ref contig) ----
        if (tid < C) {
            const unsigned long long my = ((unsigned long long)c_sc[tid] << 32) | (0xFFFFFFFFu - c_ord[tid]);
            uint32_t rank = 0;
            for (uint32_t j = 0; j < C; j++) rank += ((((unsigned long long)c_sc[j] << 32) | (0xFFFFFFFFu - c_ord[j])) > my) ? 1u : 0u;
            s_prio[rank] = (uint16_t)tid;
        }
        __syncthreads();
        if (wave == 0 && C) {
            uint32_t nk = 0;
            for (uint32_t t = 0; t < C; t++) {
                const uint32_t i = s_prio[t];
                const uint32_t q0 = c_q0[i], q1 = c_q1[i], r0 = c_r0[i], r1 = c_r1[i], rc = c_rc[i], row = c_row[i];
                bool ov = false;
                for (uint32_t j = lane; j < nk; j += 64) {
                    const uint32_t k2 = s_kept[j];
                    if (c_row[k2] == row && !(q1 < c_q0[k2] || q0 > c_q1[k2])) ov = true;
                    else if (c_rc[k2] == rc && !(r1 < c_r0[k2] || r0 > c_r1[k2])) ov = true;
                }
                if (__ballot(ov) == 0) {
                    if (lane == 0) {
                        s_kept[nk] = (uint16_t)i;
                        r_anch[row] += c_n[i]; r_nint[row] += 1u;
                        const uint32_t i0 = c_i0[i], i1 = c_i1[i];      // the chains' end SEEDS: the query seeds between the leftmost and the rightmost kept anchor are an index difference
                        r_left[row] = i0 < r_left[row] ? i0 : r_left[row]; r_right[row] = i1 > r_right[row] ? i1 : r_right[row];
                        r_cov[row] += (unsigned long long)(q1 - q0) + 1ull + A.two_c;
                    }
                    nk++;
                    lds_wave_sync();
                }
            }
        }
        __syncthreads();
        // ---- E. per-chunk identities and the pair's record (wave 0; the arithmetic and summation order of pair_reduce_small_kernel) ----
        if (wave == 0) {
            psk_hit h{};
            h.ani = -1.0f; h.ani_raw = -1.0f;
            h.ref_index = ref; h.reserved = 0;
            h.n_anchors = na;
            if (!skip && nrows) {
                const bool inrow = (uint32_t)lane < nrows;
                const uint32_t ni = inrow ? r_nint[lane] : 0u, an = inrow ? r_anch[lane] : 0u;
                const bool valid = inrow && ni != 0;
                uint32_t seeds = 0;
                if (valid) seeds = r_right[lane] - r_left[lane] + 1u;      // seeds of the chunk's contig with position in [leftmost, rightmost kept anchor] (orc_chain's seeds_between)
                unsigned long long t_cq = inrow ? r_cov[lane] : 0ull, t_a = an, t_s = valid ? seeds : 0, t_i = ni;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { t_cq += __shfl_xor(t_cq, o); t_a += __shfl_xor(t_a, o); t_s += __shfl_xor(t_s, o); t_i += __shfl_xor(t_i, o); }
                const unsigned long long vm = __ballot(valid);
                const uint32_t m = (uint32_t)__popcll(vm);
                double v = 0.0;
                if (valid) {
                    double ratio = (double)an / (double)(seeds > 1 ? seeds - 1 : 1);   // end seeds are anchors by construction
                    if (ratio > 1.0) ratio = 1.0;
                    v = pow(ratio, 1.0 / (double)A.k);
                }
                const uint32_t pos = (uint32_t)__popcll(vm & ((1ull << lane) - 1));
                lds_wave_sync();
                if (valid) s_val[pos] = v;
                lds_wave_sync();
                double sum_all = 0;
                for (uint32_t j = 0; j < m; j++) sum_all += s_val[j];                 // chunk order
                const double mean_all = m ? sum_all / (double)m : 0.0;
                double dev = valid ? (v - mean_all) * (v - mean_all) : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) dev += __shfl_xor(dev, o);
                const double std_all = m > 1 ? sqrt(dev / (double)(m - 1)) : 0.0;
                double ani = mean_all;
                if ((A.median || A.robust) && m) {
                    uint32_t rank = 0;
                    const double mine = (uint32_t)lane < m ? s_val[lane] : 0.0;
                    for (uint32_t j = 0; j < m; j++) { const double o = s_val[j]; rank += (o < mine) || (o == mine && j < (uint32_t)lane); }
                    lds_wave_sync();
                    if ((uint32_t)lane < m) s_val[rank] = mine;
                    lds_wave_sync();
                    if (A.median) ani = s_val[m / 2];
                    else {
                        uint32_t lo2 = 0, hi2 = m;
                        if (m - 2 * (m / 10) > 0) { lo2 = m / 10; hi2 = m - m / 10; }
                        double sum = 0; for (uint32_t j = lo2; j < hi2; j++) sum += s_val[j];
                        ani = sum / (double)(hi2 - lo2);
                    }
                }
                h.n_chunks = m; h.n_intervals = (uint32_t)t_i;
                h.covered_query = t_cq; h.covered_ref = t_cq; h.sum_chain_anchors = t_a; h.sum_chunk_seeds = t_s;
                if (m > 0) {
                    double afq = (double)t_cq / (double)A.q_total_len; if (afq > 1) afq = 1;
                    double afr = (double)t_cq / (double)R.total_len; if (afr > 1) afr = 1;   // one covered-bases count serves both
                    h.af_query = (float)afq; h.af_ref = (float)afr;
                    if (afq >= A.min_af || afr >= A.min_af) h.ani = (float)ani;
                    h.ani_raw = h.ani; h.ani_std = (float)std_all;
                }
            }
            if (lane == 0) {
                if (h.ani > 0.1f) A.hits[atomicAdd(&A.head_w->n_hits, 1u)] = h;      // lib.rs:654: only these cross to the host (in arrival order; the host sorts the few of them by reference)
                atomicAdd(&A.head_w->n_anchors, (unsigned long long)na);
                if (s_over) atomicOr(&A.head_w->flags, SQ_F_PAIR);
            }
            if (A.prof && pi == 0 && tid == 0) { A.prof[4] = wall_clock64(); A.prof[5] = na; A.prof[6] = nrows; }
        }
        __syncthreads();
    }
    // ---- the last workgroup to finish hands the status block and the hits to the host: plain stores into pinned memory, no download command in the stream ----
    if (A.host_out) {
        __shared__ uint32_t s_last;
        __syncthreads();
        if (tid == 0) { __threadfence(); s_last = atomicAdd(&A.head_w->done, 1u) == gridDim.x - 1u; }
        __syncthreads();
        if (s_last) {
            __threadfence();
            const uint32_t nh = __hip_atomic_load(&A.head_w->n_hits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint4* src = (const uint4*)A.head_w;
            const uint32_t n16 = (uint32_t)((sizeof(SmallQHead) + sizeof(psk_hit) * (size_t)nh) / 16);
            for (uint32_t i = tid; i < n16; i += SQ_CHAIN_T) A.host_out[i] = src[i];
        }
    }
}

inline size_t al256q(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

psk_status query_host_small(Lane* ctx, psk_db* db, const uint8_t* const* contigs, const uint64_t* lens, uint32_t n_contigs, const psk_query_opts* o,
                            HitList& all, bool* done) {
    *done = false;
    static const bool off = getenv("PSK_SMALL_QUERY") && getenv("PSK_SMALL_QUERY")[0] == '0';
    if (off) return PSK_OK;
    const psk_params prm = db->params;
    if (prm.k < 1 || prm.k > 16 || prm.c < 1 || prm.marker_c < 1) return PSK_OK;
    // the regression model runs as its own launch over a pair table the host knows: those calls take the general path
    if (o->model && (o->learned_ani == 1 || (o->learned_ani == -1 && prm.c >= 70 && !o->median))) return PSK_OK;
    if (o->learned_ani == 1 && !o->model) return PSK_OK;      // (the general path reports PSK_ENOMODEL)
    // ---- the query's shape: kept contigs (lib.rs:156), tiles, expected counts against the capacities ----
    uint32_t n_desc = 0, n_tiles = 0; uint64_t bases = 0, rows = 0, ascii_bytes = 32;
    for (uint32_t i = 0; i < n_contigs; i++) {
        const uint64_t L = lens[i];
        if (L < MIN_LENGTH_CONTIG) continue;
        if (L > (uint64_t)SQ_MAX_TILES * TILE_BASES) return PSK_OK;
        n_desc++; n_tiles += (uint32_t)((L + TILE_BASES - 1) / TILE_BASES); bases += L; rows += L / (FRAGMENT_LENGTH + 1) + 1;
        ascii_bytes += (L + 15 + 16) & ~15ull;
        if (n_desc >= SQ_MAX_DESC || n_tiles > SQ_MAX_TILES) return PSK_OK;      // (>=: the emit wave fills coff[0 .. n_desc] one entry per lane - 64 kept contigs would need a 65th lane for the total: ADVICE r4)
    }
    if (n_desc == 0 || rows > SQ_ROWS) return PSK_OK;
    auto cap_of = [](double expect) { return expect * 1.1 + 6.0 * sqrt(expect) + 64.0; };
    if (cap_of((double)bases / (double)prm.c) > (double)SQ_SEEDS || cap_of((double)bases / (double)prm.marker_c) > (double)SQ_MARKERS) return PSK_OK;
    std::shared_lock<std::shared_mutex> sh(db->rw);
    bool ok = false;
    PSK_TRY(small_query_prepare(ctx, db, sh, &ok));
    if (!ok) return PSK_OK;
    const uint32_t n_refs = (uint32_t)db->refs.size();
    hipStream_t st = ctx->stream;
    // ---- layout: [in: contig table | first tiles | tile -> contig | tile info | ASCII] [work arrays] [out: status block | hits] ----
    const size_t i_desc = 0, i_cft = al256q(sizeof(ContigDesc) * SQ_MAX_DESC), i_tci = al256q(i_cft + 4 * (SQ_MAX_DESC + 1)), i_tinfo = al256q(i_tci + 4 * SQ_MAX_TILES),
                 i_ascii = al256q(i_tinfo + sizeof(uint4) * SQ_MAX_TILES), i_end = al256q(i_ascii + (size_t)SQ_MAX_TILES * TILE_BASES + 32 * (SQ_MAX_DESC + 2));
    const size_t w_packed = i_end, w_mask = al256q(w_packed + 4 * ((size_t)SQ_MAX_TILES * TILE_WORDS + 8)), w_cnt = al256q(w_mask + 8 * (size_t)SQ_MAX_TILES * TILE_MASKS),
                 w_toff = w_cnt + 4 * (SQ_MAX_TILES + 1), w_tmc = w_toff + 4 * (SQ_MAX_TILES + 1), w_kmer = al256q(w_tmc + 4 * (SQ_MAX_TILES + 1)), w_pos = w_kmer + 4 * (size_t)SQ_SEEDS,
                 w_meta = w_pos + 4 * (size_t)SQ_SEEDS, w_pm = w_meta + 4 * (size_t)SQ_SEEDS, w_mstage = w_pm + 8 * (size_t)SQ_SEEDS, w_mout = w_mstage + 8 * (size_t)(SQ_SEEDS + 1),
                 w_short = al256q(w_mout + 8 * (size_t)SQ_MARKERS), w_out = al256q(w_short + 4 * (size_t)n_refs), w_end = w_out + sizeof(SmallQHead) + sizeof(psk_hit) * (size_t)n_refs;
    PSK_TRY(ctx->q_small.reserve(w_end + 256));
    char* D = (char*)ctx->q_small.p;
    const size_t in_bytes = i_ascii + ascii_bytes;
    const size_t out_first = sizeof(SmallQHead) + sizeof(psk_hit) * (size_t)std::min<uint32_t>(n_refs, SQ_HITS_FIRST);
    void* hp;
    PSK_TRY(ctx->pinned(al256q(in_bytes) + sizeof(SmallQHead) + sizeof(psk_hit) * (size_t)n_refs + 256, &hp));
    char* Hin = (char*)hp; char* Hout = Hin + al256q(in_bytes);
    ContigDesc* h_desc = (ContigDesc*)(Hin + i_desc); uint32_t* h_cft = (uint32_t*)(Hin + i_cft); uint32_t* h_tci = (uint32_t*)(Hin + i_tci); uint4* h_tinfo = (uint4*)(Hin + i_tinfo);
    uint64_t total_len = 0;
    {
        uint32_t d = 0, tile = 0; uint64_t off = 32;      // (sketch_scan reads the 32 bytes before a tile that does not start its contig: never before the first)
        memset(Hin + i_ascii, 0, 32);
        for (uint32_t i = 0; i < n_contigs; i++) {
            const uint64_t L = lens[i];
            if (L < MIN_LENGTH_CONTIG) continue;
            ContigDesc cd{};
            cd.byte_off = off; cd.len = (uint32_t)L; cd.first_tile = tile; cd.genome = 0; cd.contig_index = d;
            h_desc[d] = cd; h_cft[d] = tile;
            const uint32_t nt = (uint32_t)((L + TILE_BASES - 1) / TILE_BASES);
            for (uint32_t t = 0; t < nt; t++) { h_tci[tile + t] = d; h_tinfo[tile + t] = make_uint4(tile, 0u, d, d); }
            memcpy(Hin + i_ascii + off, contigs[i], L);
            const uint64_t padded = (L + 15 + 16) & ~15ull;
            memset(Hin + i_ascii + off + L, 0, padded - L);
            off += padded; tile += nt; d++; total_len += L;
        }
        h_cft[n_desc] = n_tiles;
    }
    // PSK_SQ_ZEROCOPY=0: the input crosses with an upload command and the results with a download command (A/B); default: the kernels read the pinned
    // input block in place (a contig is tens of kilobytes) and the chain kernel's last workgroup writes the results into pinned memory - two commands fewer in the stream
    static const bool zc = !(getenv("PSK_SQ_ZEROCOPY") && getenv("PSK_SQ_ZEROCOPY")[0] == '0');
    char* In = zc ? Hin : D;
    if (!zc) PSK_HIP(hipMemcpyAsync(D, Hin, in_bytes, hipMemcpyHostToDevice, st));
    SmallQHead* d_head = (SmallQHead*)(D + w_out);
    psk_hit* d_hits = (psk_hit*)(D + w_out + sizeof(SmallQHead));
    SmallQSketch S{};
    S.d_bases = (const uint8_t*)(In + i_ascii); S.d_desc = (const ContigDesc*)(In + i_desc); S.d_tci = (const uint32_t*)(In + i_tci); S.d_tinfo = (const uint4*)(In + i_tinfo);
    S.d_cft = (const uint32_t*)(In + i_cft); S.n_desc = n_desc; S.n_tiles = n_tiles;
    S.d_packed = (uint32_t*)(D + w_packed); S.d_mask = (uint64_t*)(D + w_mask); S.d_cnt = (uint32_t*)(D + w_cnt); S.d_toff = (uint32_t*)(D + w_toff); S.d_tmc = (uint32_t*)(D + w_tmc);
    S.seed_kmer = (uint32_t*)(D + w_kmer); S.seed_pos = (uint32_t*)(D + w_pos); S.seed_meta = (uint32_t*)(D + w_meta); S.seed_pm = (uint64_t*)(D + w_pm); S.mstage = (uint64_t*)(D + w_mstage);
    S.head = d_head;
    PSK_TRY(small_query_sketch_enqueue(ctx, &prm, S, st));
    SqScreenArgs SA{};
    SA.mstage = S.mstage; SA.toff = S.d_toff; SA.tmc = S.d_tmc; SA.n_tiles = n_tiles; SA.head = d_head; SA.markers_out = (uint64_t*)(D + w_mout);
    SA.refs = (const MarkerSet*)db->d_marker_ptr.p; SA.inv_key = (const uint64_t*)db->inv_key.p; SA.inv_val = (const uint32_t*)db->inv_ref.p; SA.inv_bucket = (const uint32_t*)db->inv_bucket.p;
    SA.inv_shift = 2 * K_MARKER - db->inv_bits; SA.inv_n = (uint32_t)db->inv_n; SA.n_refs = n_refs;
    SA.thresh = pow(o->cutoff != 0.0 ? o->cutoff : 0.80, (double)K_MARKER); SA.rescue_small = !o->faster_small;      // lib.rs:597, 603-609
    SA.canon = db->has_dups ? (const uint32_t*)db->d_canon.p : nullptr;
    SA.shortlist = (uint32_t*)(D + w_short);
    static const bool pf_off = getenv("PSK_SQ_PREFILTER") && getenv("PSK_SQ_PREFILTER")[0] == '0';      // tests, A/B
    if (db->gsi_state == 1 && !pf_off) { SA.gsi_key = (const uint32_t*)db->gsi_key.p; SA.gsi_val = (const unsigned long long*)db->gsi_val.p; SA.gsi_bucket = (const uint32_t*)db->gsi_bucket.p; SA.gsi_shift = db->gsi_shift; }
    SA.q_kmer = S.seed_kmer;
    static std::once_flag lds_once;
    static hipError_t lds_rc = hipSuccess;
    std::call_once(lds_once, [] { lds_rc = hipFuncSetAttribute((const void*)sq_screen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * SQ_MAX_REFS)); });
    PSK_HIP(lds_rc);
    ctx->t_begin(K_SCREEN);
    // (the status block in pinned memory is the device's to fill: until it has, it says "rerun on the general path" - a launch that never ran must not leave
    // the PREVIOUS call's status and hits to be read as this call's: ADVICE r4)
    if (zc) ((SmallQHead*)Hout)->flags = 0xFFFFFFFFu;
    hipLaunchKernelGGL(sq_screen_kernel, dim3(1), dim3(SQ_SCREEN_T), 4 * (size_t)n_refs, st, SA);
    PSK_HIP(hipGetLastError());
    ctx->t_end();
    SqChainArgs CA{};
    CA.head = d_head; CA.head_w = d_head; CA.shortlist = SA.shortlist; CA.q_kmer = S.seed_kmer; CA.q_pos = S.seed_pos; CA.q_meta = S.seed_meta;
    CA.rd = (const SketchDesc*)db->d_refdesc.p; CA.hits = d_hits; CA.q_total_len = total_len; CA.n_desc = n_desc;
    CA.band = (uint32_t)std::max(1, std::min(MAX_CHAIN_BAND, BP_CHAIN_BAND / (int)prm.c)); CA.two_c = 2u * (uint32_t)prm.c;
    CA.k = prm.k; CA.median = o->median; CA.robust = o->robust; CA.min_af = o->min_aligned_frac > 0 ? o->min_aligned_frac : 0.15;
    static const bool prof_on = getenv("PSK_SQ_PROFILE") != nullptr;
    const size_t prof_off = (offsetof(SmallQHead, pad1) + 7) & ~(size_t)7;      // (seven 8-byte words inside the status block's padding)
    CA.prof = prof_on ? (unsigned long long*)((char*)d_head + prof_off) : nullptr;
    CA.host_out = zc ? (uint4*)Hout : nullptr;
    static const bool no_team = getenv("PSK_SQ_TEAM") && getenv("PSK_SQ_TEAM")[0] == '0';
    CA.no_team = no_team;
    ctx->t_begin(K_CHAIN_CHUNK);
    // The shortlist's length is known on the device only, and a workgroup of this kernel holds 77 KB of LDS: a grid sized for the worst case (every
    // reference) is a thousand workgroups that queue for LDS only to find nothing to do - from eight host threads they held the rate at 20 k queries/s
    // where 128-256 workgroups give 31 k (profiles/r4/r4i_grid.txt). The grid follows the LAST call's shortlist on this lane (neighbouring queries of a
    // workload have similar shortlists); a longer one is walked in several rounds by the same workgroups. PSK_SQ_GRID fixes it (A/B).
    static const uint32_t grid_env = getenv("PSK_SQ_GRID") ? (uint32_t)std::max(1, atoi(getenv("PSK_SQ_GRID"))) : 0u;
    const uint32_t grid = grid_env ? grid_env : std::max(64u, std::min(1024u, ctx->sq_last_short + ctx->sq_last_short / 4 + 16u));
    hipLaunchKernelGGL(sq_chain_kernel, dim3(std::min<uint32_t>(n_refs, grid)), dim3(SQ_CHAIN_T), 0, st, CA);
    PSK_HIP(hipGetLastError());
    ctx->t_end();
    if (!zc) PSK_HIP(hipMemcpyAsync(Hout, d_head, out_first, hipMemcpyDeviceToHost, st));
    PSK_HIP(hipStreamSynchronize(st));      // the ONE synchronisation of the call
    const SmallQHead* hh = (const SmallQHead*)Hout;
    if (hh->flags) { ctx->dev->sq_rerun++; return PSK_OK; }            // a capacity was exceeded: the general path sizes everything from the counts
    const uint32_t n_short = hh->n_short, nh = hh->n_hits;
    ctx->sq_last_short = n_short;
    if (n_short > n_refs || nh > n_short) { psk_set_error("internal: shortlist longer than the database"); return PSK_EHIP; }
    if (!zc && nh > SQ_HITS_FIRST) {          // more hits than cross with the status block: the rest of the records
        PSK_HIP(hipMemcpyAsync(Hout + out_first, (const char*)d_head + out_first, sizeof(psk_hit) * (size_t)(nh - SQ_HITS_FIRST), hipMemcpyDeviceToHost, st));
        PSK_HIP(hipStreamSynchronize(st));
    }
    psk_hit* hits = (psk_hit*)(Hout + sizeof(SmallQHead));
    std::sort(hits, hits + nh, [](const psk_hit& a, const psk_hit& b) { return a.ref_index < b.ref_index; });      // ascending insertion index, like the general path
    if (!all.reserve(all.n + std::max<uint32_t>(nh, 1u))) { psk_set_error("out of host memory"); return PSK_ENOMEM; }
    memcpy(all.p + all.n, hits, sizeof(psk_hit) * (size_t)nh);
    all.n += nh;
    if (prof_on) {
        const unsigned long long* P = (const unsigned long long*)(Hout + prof_off);
        fprintf(stderr, "[psk small query] seeds %u markers %u shortlist %u hits %u | first pair: anchors %llu rows %llu, join %.2f us, chunk table %.2f, DP %.2f, select+reduce %.2f\n",
                hh->n_seeds, hh->n_markers, n_short, nh, P[5], P[6], (P[1] - P[0]) / 100.0, (P[2] - P[1]) / 100.0, (P[3] - P[2]) / 100.0, (P[4] - P[3]) / 100.0);
    }
    ctx->dev->w_pairs += n_short; ctx->dev->w_items += (uint64_t)n_short * hh->n_seeds; ctx->dev->w_anchors += hh->n_anchors;
    ctx->dev->sq_taken++;
    *done = true;
    return PSK_OK;
}
