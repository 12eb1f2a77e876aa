// Chain stage 2: banded chaining DP over every chunk's anchors, one candidate chain per DP tree.
#include "chain_stages.h"

// ------------------------------------------------------------------ chaining





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

// XT: further tree slots per lane in LDS (0, or LANE_XTREES for Gb-scale pairs: there a seed has ~6 chance 15-mer matches beside the
// true one, the band of 20 ANCHORS reaches back only ~3 seeds, a true chain breaks wherever three seeds in a row do not match and a
// chunk holds 5-15 qualifying trees - with four slots most chunks went to the wave-per-chunk kernel, 7.7 of 10 ms per 3 Gb pair)
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


// (launched from chain.hip)
template __global__ void chain_wave_reg_kernel<1>(ChainArgs A);
template __global__ void chain_wave_reg_kernel<2>(ChainArgs A);
