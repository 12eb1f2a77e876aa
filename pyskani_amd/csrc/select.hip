// Chain stage 3: greedy selection of non-overlapping candidate chains per pair, the seeds between every chunk's outermost kept anchors.
#include "chain_stages.h"

// ------------------------------------------------------------------ chain selection (per pair)
// Greedy over ALL candidate chains of a pair by (score desc, generation order): a chain is kept unless it
// overlaps a kept chain on the query (same chunk) or on the reference (same ref contig). One wave per pair:
// candidates staged in LDS, bitonic sort of (score, ~order) keys, kept list scanned 64 entries at a time.


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
#ifdef SEL_TRACE
    unsigned long long t_ph[8]; int n_ph = 0;
#define SEL_STAMP() do { if (gr == 0 && tid == 0 && n_ph < 8) t_ph[n_ph++] = wall_clock64(); } while (0)
#else
#define SEL_STAMP() do {} while (0)
#endif
    SEL_STAMP();
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
    SEL_STAMP();      // (1: candidates listed, query-side conflicts)
    // ---- conflicts on the reference: order by (ref contig, r0), running max of r1 by doubling ----
    for (uint32_t j = gt; j < P; j += GT) { key[j] = j < C ? (((unsigned long long)S.c_rc[slot[j]] << 32) | S.c_r0[slot[j]]) : ~0ull; idx[j] = j; }
    grp_sync(g);
    big_bitonic(key, idx, P, false, g, t_key, t_pay);
    SEL_STAMP();      // (2: the reference-order sort)
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
    SEL_STAMP();      // (3: running maximum + reference-side conflicts)
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
    SEL_STAMP();      // (4: commits of the unconflicted, list + sort of the conflicted)
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
    SEL_STAMP();      // (5: greedy)
#ifdef SEL_TRACE
    if (tid == 0 && G > 1) printf("SEL_TRACE pair %u C %u ncf %u G %u us: list %.0f sort %.0f max %.0f commit %.0f greedy %.0f\n", p, C, ncf, G, (t_ph[1] - t_ph[0]) * 0.01, (t_ph[2] - t_ph[1]) * 0.01, (t_ph[3] - t_ph[2]) * 0.01, (t_ph[4] - t_ph[3]) * 0.01, (t_ph[5] - t_ph[4]) * 0.01);
#endif
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
        const uint32_t n_cand = big_count_candidates(B.S, p, s_scan);
        if (n_cand > B.solo) { if (threadIdx.x == 0) { const uint32_t at = atomicAdd(B.huge_count, 1u); B.huge_list[at] = p; if (at < BIG_GROUPS) B.huge_c[at] = n_cand; } continue; }
        BigGrp g{1, 0, 0, nullptr, B.parts, B.parts + BIG_GMAX};
        g.part_a = B.parts + (size_t)(BIG_GROUPS + blockIdx.x) * 2 * BIG_GMAX; g.part_b = g.part_a + BIG_GMAX;
        select_big_pair(B, p, g, s_raw, s_scan);
        __syncthreads();
    }
}

// Gb-scale pairs: the launch's workgroups (all co-resident: at most BIG_GMAX, one per CU) split into groups, one per listed pair; a group's workgroups sit on as few XCDs as
// possible (workgroup b runs on XCD b % 8). Up to BIG_GROUPS pairs (a batch of eleven 3 Gb pairs): the workgroups are dealt BY WEIGHT - a pair's candidates, counted by the solo
// kernel - because the launch lasts as long as its largest pair (150 000 to 480 000 candidates in one batch of the 8 x 3 Gb step: with sixteen workgroups each, the largest
// pair's sort, running maximum and commits took 3.8 of the launch's 4.8 ms while the groups of the small pairs had long finished; the phases' times: profiles/r6/r6_ablation.md).
// More pairs than that: equal groups, each taking every groups-th pair. Any group size gives the same result (select_big_pair).
__global__ __launch_bounds__(BIG_T) void select_huge_kernel(BigArgs B) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[12 * BIG_TILE];
    __shared__ uint32_t s_scan[BIG_T / 64 + 1];
    const uint32_t n = *B.huge_count;
    if (n == 0) return;
    const uint32_t NB = gridDim.x;                  // a power of two, >= 8 (or 1)
    if (n <= BIG_GROUPS && NB >= 8 && NB >= 2 * n) {
        const uint32_t l = (blockIdx.x & 7u) * (NB >> 3) + (blockIdx.x >> 3);      // consecutive logical workgroups share an XCD
        unsigned long long total = 0;
        for (uint32_t k = 0; k < n; k++) total += B.huge_c[k];
        uint32_t start = 0, gid = 0, G = 1; unsigned long long cum = 0;
        for (uint32_t k = 0; k < n; k++) {      // group k = logical workgroups [start, end): its share of the candidates, at least one workgroup, room left for the pairs behind it
            cum += B.huge_c[k];
            uint32_t end = k + 1 == n ? NB : (uint32_t)((unsigned long long)NB * cum / (total ? total : 1ull));
            if (end < start + 1u) end = start + 1u;
            if (end > NB - (n - 1u - k)) end = NB - (n - 1u - k);
            if (l >= start && l < end) { gid = k; G = end - start; break; }
            start = end;
        }
        BigGrp g{G, l - start, 0, B.ctr + gid, B.parts + (size_t)gid * 2 * BIG_GMAX, B.parts + (size_t)gid * 2 * BIG_GMAX + BIG_GMAX};
        select_big_pair(B, B.huge_list[gid], g, s_raw, s_scan);
        return;
    }
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
