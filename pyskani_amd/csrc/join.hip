// Chain stage 1: the anchors of every pair of a batch - every (query seed, reference seed) pair of equal k-mer, in (q contig, q pos, r contig, r pos) order - and the pairs' chunk tables.
#include "chain_stages.h"

// ------------------------------------------------------------------ anchors

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
// The join kernels are bound by the LATENCY of their chain of dependent loads (pair table -> pair descriptor -> query k-mer ->
// bucket table -> reference k-mers -> reference position) and by instruction issue, at a wave residency the register file
// already caps (profiles/r2/r2e_pmc_join_kernels_sq.txt: 79 % of residency waiting, 7 waves per SIMD). The *4 variants put
// JT = 4 tiles of 256 items through every stage TOGETHER - four independent chains in flight per wave instead of one - and
// amortise the pair lookup over 1 024 items (emit: 37.7 -> 26.1 ms, join: 40.7 -> 38.6 ms per 10^5 pairs).
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
// ... and because the workgroup sees the pair's items in (contig, position) order anyway, it also builds the pair's CHUNK TABLE
// (chunk_heads_kernel's rows: a chunk runs from its head anchor to the first anchor more than FRAGMENT_LENGTH further on the
// query): every lane leaves its items' keys and in-wave offsets in LDS, and once the round's anchors are written wave 0 steps from
// head to head through the round's keys with 64-wide compares - the separate pass over all anchors (16 B each) that
// chunk_heads_kernel makes is gone.
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
template <bool EMIT>
__global__ __launch_bounds__(64) void gsi_join_kernel(GsiJoinArgs A) {
    extern __shared__ unsigned long long s_gsi[];
    // Which index this wave walks. The entry's block table (gsl_blocks_kernel) lists the blocks that hold one of its pairs' references: at most b_max of them - or no
    // database-wide index to turn to (databases beyond its 16-bit reference ids) - and the wave walks those blocks, with the pass bits of ONE block (four words) in LDS at a
    // time; otherwise the database-wide index, with the query's whole row of the pass matrix as a bitset.
    const uint32_t n_blk = A.b_blocks ? A.blk_cnt[blockIdx.x] : 0u;
    const bool blocked = A.b_blocks && (n_blk <= A.b_max || !A.g_key);
    const uint32_t nw = blocked ? 4u : (A.n_refs + 63u) / 64u;      // words of the bitset in use (the launch's LDS holds nw_lds)
    // LDS of the wave. EMIT: [with A.stage: the even-indexed anchor every pair holds back, 16 B per pair][the pairs' STATE, one 16-byte record each: {anchors so far, first
    // anchor of the pair (GSI_DEAD: fewer than MIN_ANCHORS anchors - it cannot chain: no anchors, no chunk table), query position of the open chunk's head, q contig << 16 | rows
    // so far}: ONE read per anchor in the walk's chain of dependent LDS round trips, where four arrays took three trips][the open chunk's first anchor per pair][the pass bitset
    // and its prefix counts]. COUNT: [bitset][prefix counts][the pairs' anchor counts].
    uint4* s_line = (uint4*)s_gsi;
    uint4* s_st = s_line + ((EMIT && A.stage) ? A.p_cap : 0u);
    uint32_t* s_hi = (uint32_t*)(s_st + (EMIT ? A.p_cap : 0u));
    unsigned long long* s_bits = (unsigned long long*)(s_hi + (EMIT ? A.p_cap : 0u));
    uint32_t* s_pref = (uint32_t*)(s_bits + A.nw_lds);
    uint32_t* s_cur = s_pref + ((A.nw_lds + 1u) & ~1u);      // (COUNT only)
    constexpr uint32_t GSI_DEAD = 0xFFFFFFFFu;
    const int lane = threadIdx.x;
    const BatchQ B = A.bq[blockIdx.x];
    const uint32_t P = B.rank_hi - B.rank_lo;
    {   // pass row -> bitset + prefix counts (the database-wide index; a wave that walks blocks fills four words per block from its table)
        const uint8_t* __restrict__ row = A.pass + (size_t)B.q * A.n_refs;
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < nw && !blocked; w0 += 4) {
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
        for (uint32_t j = lane; j < P; j += 64) {
            if (EMIT) { const uint32_t a = A.pstart[B.pair_off + j], z = A.pstart[B.pair_off + j + 1]; s_st[j] = make_uint4(0u, !A.onepass && z - a < MIN_ANCHORS ? GSI_DEAD : a, 0u, 0u); }
            else s_cur[j] = 0;
        }
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
    // (blocked index: one walk of the query's seeds per listed block - a pair's reference sits in ONE block, so its anchors keep their order)
    const int x_shift = blocked ? A.b_shift : A.g_shift;
    const uint32_t n_it = blocked ? n_blk : 1u;
    const uint32_t* __restrict__ brow = A.blk_tab + (size_t)blockIdx.x * A.blk_cap * GSL_BT_WORDS;
#pragma unroll 1
    for (uint32_t bi = 0; bi < n_it; bi++, brow += GSL_BT_WORDS) {
    {
    const uint32_t* __restrict__ x_key = A.g_key; const unsigned long long* __restrict__ x_val = A.g_val; const uint32_t* __restrict__ bkt = A.g_bucket;
    if (blocked) {
        const uint32_t blk = brow[8];
        const unsigned long long x_base = ((unsigned long long)brow[11] << 32) | brow[10];
        x_key = A.b_key + x_base; x_val = A.b_val + x_base; bkt = A.b_bucket + (size_t)blk * A.b_nb1;
        lds_wave_sync();      // (the previous block's last lookups are through)
        if (lane < 4) {
            uint32_t run = brow[9];
#pragma unroll
            for (int u = 0; u < 6; u++) { const uint32_t w = brow[u]; if (u < 2 * lane) run += (uint32_t)__popc(w); }
            s_bits[lane] = ((unsigned long long)brow[2 * lane + 1] << 32) | brow[2 * lane]; s_pref[lane] = run;
        }
        lds_wave_sync();
    }
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
                const uint32_t ref = blocked ? (uint32_t)(v >> 48) & 255u : (uint32_t)(v >> 48), w = ref >> 6, bpos = ref & 63u;      // (a block's entries carry the reference's id within the block)
                uint32_t slot = 0xFFFFFFFFu;
                uint4 st = make_uint4(0u, 0u, 0u, 0u);      // EMIT: the pair's state, read once
                if (match) {
                    const unsigned long long bits = s_bits[w];
                    const uint32_t rk = s_pref[w] + (uint32_t)__popcll(bits & ((1ull << bpos) - 1ull));
                    if (((bits >> bpos) & 1ull) && rk >= B.rank_lo && rk < B.rank_hi) slot = rk - B.rank_lo;
                    if (EMIT && slot != 0xFFFFFFFFu) { st = s_st[slot]; if (st.y == GSI_DEAD) slot = 0xFFFFFFFFu; }
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
                const uint32_t base = valid ? st.x : 0u;
                // Anchors leave in PAIRS (A.stage): a scattered 16-byte store costs 32 bytes of HBM write traffic (profiles/r4/r4k_pmc_calibration.md), so an anchor
                // with an even index waits in LDS (one 16-byte slot per pair) for its odd neighbour, and the lane that brings that one writes both: one 32-byte granule.
                // Inside a group of several lanes (a reference holding the k-mer several times) neighbours go out directly; only a group's last even anchor waits.
                bool hold = false; uint4 av = make_uint4(0, 0, 0, 0);
                if (EMIT && valid) {
                    const unsigned long long dst = (unsigned long long)st.y + base + j;
                    if (A.onepass && base + j >= nq + (nq >> 3) + 8u) atomicOr(A.err, 4u);      // the pair's room (gsi_room_kernel) is used up
                    else if (dst < A.cap) {
                        const uint32_t rmeta = (uint32_t)((((v >> 33) & 0x7FFFull) << 1) | (v & 1ull));      // ref contig << 1 | (fwd < rc)
                        av = make_uint4(sqp, (uint32_t)(v >> 1), (rmeta & ~1u) | ((rmeta ^ sqm) & 1u), sqm >> 1);
                        const uint32_t d32 = (uint32_t)dst;
                        if (!A.stage) A.anc[dst] = av;
                        else if (d32 & 1u) {      // odd: out it goes - with its even neighbour from LDS when that one is the pair's own and is not the lane before this one
                            if (!same && d32 > st.y) A.anc[d32 - 1u] = s_line[slot];
                            A.anc[d32] = av;
                        } else if (last) hold = true;      // even and the group's last: waits (written to LDS below, after the step's reads of the slots)
                        else A.anc[d32] = av;                // even with its odd neighbour in the next lane: both go out directly
                    } else atomicOr(A.err, 2u);
                    // chunk table: a chunk = the pair's anchors of one query contig within FRAGMENT_LENGTH of its first anchor (chunk_heads_kernel's rule), decided
                    // by the first lane of the (seed, reference) group - one group per pair and step
                    if (!same) {
                        const uint32_t idx = st.y + base, qc = sqm >> 1, hc = st.w;
                        uint32_t* stw = (uint32_t*)&s_st[slot];
                        if (base == 0) { stw[2] = sqp; s_hi[slot] = idx; stw[3] = qc << 16; }
                        else if ((hc >> 16) != qc || (unsigned long long)sqp > (unsigned long long)st.z + FRAGMENT_LENGTH) {
                            const uint32_t rows = hc & 0xFFFFu;
                            if (rows < Q.rows && rows < 0xFFFFu) A.chunks[(size_t)B.row_off + (size_t)slot * Q.rows + rows] = make_uint2(s_hi[slot], idx < A.cap ? idx : A.cap); else atomicOr(A.err, 1u);
                            stw[2] = sqp; s_hi[slot] = idx; stw[3] = (qc << 16) | (rows + 1u);
                        }
                    }
                }
                if (dup) lds_wave_sync();
                if (EMIT && hold) s_line[slot] = av;
                if (last) { if (EMIT) ((uint32_t*)&s_st[slot])[0] = base + j + 1u; else s_cur[slot] = base + j + 1u; }
                if (dup) lds_wave_sync();
            }
        }
#undef GSI_FETCH
    }
    }
    }      // the listed blocks (or the one walk of the database-wide index)
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
            const uint4 fin = s_st[j];
            const uint32_t n = fin.x;
            if (A.stage && n && fin.y != GSI_DEAD && !(A.onepass && n > nq + (nq >> 3) + 8u)) {      // an even last anchor is still waiting for a neighbour that never came
                const unsigned long long e = (unsigned long long)fin.y + n - 1u;
                if (!(e & 1ull) && e < A.cap) A.anc[e] = s_line[j];
            }
            if (fin.y != GSI_DEAD && n && !(A.onepass && n < MIN_ANCHORS)) {      // (fewer than MIN_ANCHORS anchors: no chain, no chunk table - the rows written on the way are not counted)
                rows = fin.w & 0xFFFFu;
                const unsigned long long e = (unsigned long long)fin.y + n;
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

// (launched from chain.hip)
template __global__ void anchor_next_kernel<0>(const uint4* __restrict__ anc,
                                                          const uint32_t* __restrict__ pstart, uint32_t n_pairs,
                                                          const uint32_t* __restrict__ coarse, uint32_t* __restrict__ nxt);
template __global__ void anchor_next_kernel<1>(const uint4* __restrict__ anc,
                                                          const uint32_t* __restrict__ pstart, uint32_t n_pairs,
                                                          const uint32_t* __restrict__ coarse, uint32_t* __restrict__ nxt);
template __global__ void anchor_next_kernel<2>(const uint4* __restrict__ anc,
                                                          const uint32_t* __restrict__ pstart, uint32_t n_pairs,
                                                          const uint32_t* __restrict__ coarse, uint32_t* __restrict__ nxt);

// (launched from chain.hip)
template __global__ void gsi_join_kernel<false>(GsiJoinArgs A);
template __global__ void gsi_join_kernel<true>(GsiJoinArgs A);
