// Seed-index join of batches of MID-SIZED pairs (all-vs-all of ~5 Mb genomes): the anchors of lib.rs:640-657's chain_seeds calls,
// for every pair of a batch at once, through the seed index in blocks of 256 references (psk_db::bsi_*: a block's seeds sorted by k-mer,
// within a k-mer by reference, contig, position; block-local reference ids, 64-bit block offsets: no bound on the database).
//
// The per-pair join (anchor_join4_kernel + anchor_emit_pairs_kernel) looks every query seed up once per PAIR: 4 x 10^10 (pair, seed)
// lookups and as many 8-byte records written in k-mer order and read back in position order for 10 000 x 10 000 genomes, where the
// batch holds 4 x 10^8 distinct query seeds. Here ONE lookup per query seed returns the seed's matches in every reference (a
// contiguous run of the index), and the wave deals them to the query's pairs, as gsi_join_kernel does for contigs. What is different
// for genomes: a query has 40 000 seeds - one wave per query would be 10 000 dependent chains of 2 M index entries each - so a query
// is cut into SLICES of GSL_SEEDS seeds, one wave per (query, slice), and a pair's anchors, which must end up contiguous and in
// (q contig, q pos, r contig, r pos) order, get their places from a COUNT walk:
//
//   gsl_blocks_kernel       per batch ENTRY (a query and <= 256 of its passing references): the index blocks that hold one of those references, with their pass bits
//   gsl_walk_kernel<false>  COUNT: per (pair, slice) the number of anchors and a bitmap of the slice's seeds that have one
//   (scan over the pairs' totals -> pstart: chain.hip)
//   gsl_heads_kernel        per pair, slice after slice: first anchor of every (pair, slice), and - from the bitmaps - which seeds head
//                           a CHUNK (a chunk = the anchors of one query contig within FRAGMENT_LENGTH of its first anchor - a property of
//                           the query's seed positions, so no anchor is read for it): the rows before, and the head open at, every slice
//   gsl_walk_kernel<true>   EMIT: the same walk writes the 16-byte anchors, every pair's through a 64-byte line staged in LDS, so
//                           that they leave as whole lines (a scattered 16-byte store costs 32 bytes of HBM write traffic), and the chunk
//                           table's rows as it meets the heads
//
// Bit-exactness: the anchors and chunk tables are the ones anchor_emit_pairs_kernel / chunk_heads_kernel produce (same order, same
// rows); everything downstream (DP, selection, reduce) is unchanged.
#include "slice_join.h"
#include "chain_dev.h"

void gsl_make_tab(const BatchQ* bq, size_t n_entries, const uint32_t* q_seeds, std::vector<uint2>& tab, std::vector<uint2>& ebase, uint64_t* n_records, uint64_t* n_slices) {
    tab.clear(); ebase.resize(n_entries);
    uint64_t rec = 0, nsl_all = 0; uint32_t max_sl = 0;
    for (size_t e = 0; e < n_entries; e++) {
        const uint32_t nsl = (q_seeds[e] + GSL_SEEDS - 1) / GSL_SEEDS;
        ebase[e] = make_uint2((uint32_t)rec, (uint32_t)nsl_all);
        rec += (uint64_t)nsl * (bq[e].rank_hi - bq[e].rank_lo);
        nsl_all += nsl;
        max_sl = std::max(max_sl, nsl);
    }
    *n_records = rec; *n_slices = nsl_all;
    // (a query's slices one after the other. Slice-major orders, and one that kept a slice position of every query on ONE XCD so that relatives would find each
    // other's index runs in its L2, were measured on the 10 000 x 10 000 step: 846 and 860 ms against 843 - the index is 4.8 GB, a run is read from HBM whoever read it last)
    (void)max_sl;
    for (size_t e = 0; e < n_entries; e++) for (uint32_t sl = 0; sl < (q_seeds[e] + GSL_SEEDS - 1) / GSL_SEEDS; sl++) tab.push_back(make_uint2((uint32_t)e, sl));
}

// LDS of one walk wave, the 16-byte units first: [EMIT: 4 x p_cap staged anchors][EMIT: lim1 per pair: p_cap x 8][EMIT: (cursor, first anchor) per pair, the step's lines]
// [the walked block's pass row as eight (32 bits, prefix count) entries]
// [COUNT: cursors: p_cap x 4; anchor-seed bitmaps, GSL_WORDS rows of p_cap + 1 words | EMIT: (cursor, first anchor) per pair: p_cap x 8; rows so far per pair: p_cap x 4]
// anchors per staged line of the emit walk (GSL_LINE_N = 4: whole 64-byte lines; 2: 32-byte halves - half the LDS per wave, two more waves per SIMD)
#ifndef GSL_LINE_N
#define GSL_LINE_N 4
#endif
constexpr uint32_t GSL_LW = GSL_LINE_N, GSL_LM = GSL_LW - 1u, GSL_LS = GSL_LW == 8 ? 3u : GSL_LW == 4 ? 2u : 1u;
static_assert(GSL_LW == 8 || GSL_LW == 4 || GSL_LW == 2, "a staged line holds eight, four or two anchors");
static size_t gsl_walk_lds(const GslArgs& A, bool emit) {
    const size_t nw = (A.n_refs + 63) / 64;
    (void)nw;
    return (emit ? (16 * GSL_LW + 8 + 8) * (size_t)A.p_cap + 8 * 64 : 0) + 64 + 4 * (size_t)A.p_cap + (emit ? 0 : 4 * (size_t)GSL_WORDS * (A.p_cap + 1));
}

// (streaming - nontemporal - stores of the anchors were measured on the 10 000 x 10 000 step: the walk takes the same time and the DP kernel that reads the anchors next 178 instead of
// 165 ms: plain stores; profiles/r5/r5_ablation.md)
template <bool EMIT, bool STAGE>
__global__ __launch_bounds__(64) void gsl_walk_kernel(GslArgs A) {
    extern __shared__ uint4 s_gsl[];
    const int lane = threadIdx.x;
    const uint2 te = A.tab[blockIdx.x];
    if (te.x == 0xFFFFFFFFu) return;
    if (EMIT && (A.err[5] || *(const unsigned long long*)(A.err + 16) > A.cap)) return;      // the attempt is rerun whatever it writes (pair_guard_kernel)
    const BatchQ B = A.bq[te.x];
    const SketchDesc Q = A.qd[B.q];
    const uint32_t sb = te.y * GSL_SEEDS;
    if (sb >= Q.n) return;
    const uint32_t se = Q.n - sb > GSL_SEEDS ? sb + GSL_SEEDS : Q.n;
    const uint32_t P = B.rank_hi - B.rank_lo, pc = A.p_cap;
    const uint2 eb = A.ebase[te.x];
    const uint32_t rec0 = eb.x + te.y * P;
    uint4* s_line = s_gsl;                                                                  // EMIT: slot t of pair j at [t * pc + j]
    unsigned long long* s_lim = (unsigned long long*)(s_gsl + (EMIT ? GSL_LW * pc : 0u));       // EMIT: lim1 of the pair's open chunk
    uint2* s_cs = (uint2*)(s_lim + (EMIT ? pc : 0u));                                       // EMIT: (cursor, first anchor of the (pair, slice)): one 8-byte read
    uint2* s_fl = s_cs + (EMIT ? pc : 0u);                                                  // EMIT: the step's complete lines (pair | first slot << 16, first anchor of the line)
    uint2* s_bp = s_fl + (EMIT ? 64u : 0u);                                                 // the block being walked: its 256 references of the query's pass row, 32 per entry: (bits, passing references before them)
    uint32_t* s_cur = (uint32_t*)(s_bp + 8u);                                               // COUNT: anchors so far; EMIT: chunk-table rows of the pair so far
    uint32_t* s_rows = s_cur;
    uint32_t* s_bm = s_cur + pc;                                                            // COUNT: word w of pair j at [w * (pc + 1) + j]
    // The index comes in blocks of 2^BSI_BLOG references: only the blocks that hold a reference of one of the entry's pairs are walked (a run of the whole database's index
    // holds ~1 % of ALL genomes by chance - see psk_db::bsi_*). A pair's reference sits in one block, so its anchors still come out in seed order. Which blocks those are,
    // and their 256 references of the query's pass row, comes from the entry's block table (gsl_blocks_kernel: one sweep of the row per ENTRY, not per wave).
    {
        if (EMIT) for (uint32_t j = lane; j < P; j += 64) { const uint4 v = A.rec[rec0 + j]; s_cs[j] = make_uint2(v.x, v.x); s_rows[j] = v.y; s_lim[j] = ((unsigned long long)v.w << 32) | v.z; }
        else {
            for (uint32_t j = lane; j < P; j += 64) s_cur[j] = 0;
            for (uint32_t x = lane; x < GSL_WORDS * (pc + 1u); x += 64) s_bm[x] = 0;
        }
    }
    // EMIT: the slice's seeds that head a chunk of SOME pair of the entry (heads kernel), word w in lane w: the chunk-table code only runs in those seeds' steps
    uint32_t unw = 0;
    if (EMIT && lane < (int)GSL_WORDS) unw = A.un[(size_t)(eb.y + te.y) * GSL_WORDS + lane];
    lds_wave_sync();
    // The walk (gsi_join_kernel's): a batch of 64 seeds has its k-mers loaded two batches ahead and its bucket bounds one batch ahead; its runs are cut into STEPS
    // of 64 index entries, numbered through the batch, and the entries of step t + GSL_AHEAD are requested before step t is dealt out.
    constexpr uint32_t GSL_AHEAD = 4;
    unsigned long long visited = 0;      // COUNT: index entries in the runs this lane's seeds found (psk_ctx_join_work)
    const uint32_t n_blk = A.blk_cnt[te.x];
    const uint32_t* __restrict__ brow = A.blk_tab + (size_t)te.x * A.blk_cap * GSL_BT_WORDS;
#pragma unroll 1
    for (uint32_t bi = 0; bi < n_blk; bi++, brow += GSL_BT_WORDS) {
    {
    const uint32_t blk = brow[8];
    const unsigned long long x_base = ((unsigned long long)brow[11] << 32) | brow[10];
    const uint32_t* __restrict__ x_key = A.g_key + x_base; const unsigned long long* __restrict__ x_val = A.g_val + x_base;
    {   // the block's 256 references of the pass row -> eight (32 bits, passing references before them) entries: reference -> pair of the entry in ONE 8-byte LDS read
        lds_wave_sync();      // (the previous block's last lookups are through)
        if (lane < 8) {
            uint32_t run = brow[9];
#pragma unroll
            for (int u = 0; u < 7; u++) { const uint32_t w = brow[u]; if (u < lane) run += (uint32_t)__popc(w); }
            s_bp[lane] = make_uint2(brow[lane], run);
        }
        lds_wave_sync();
    }
    const uint32_t* __restrict__ bkt = A.g_bucket + (size_t)blk * A.g_nb1;
    uint32_t km1 = sb + (uint32_t)lane < se ? Q.kmer[sb + lane] : 0u, km2 = sb + 64u + (uint32_t)lane < se ? Q.kmer[sb + 64u + lane] : 0u;
    uint32_t lo1 = 0, hi1 = 0;
    if (sb + (uint32_t)lane < se) { const uint32_t b = km1 >> A.g_shift; lo1 = bkt[b]; hi1 = bkt[b + 1]; }
    for (uint32_t c0 = sb; c0 < se; c0 += 64) {
        const uint32_t i = c0 + (uint32_t)lane;
        const uint32_t km = km1, lo = lo1, hi = hi1;
        km1 = km2; lo1 = 0; hi1 = 0;
        if (i + 64u < se) { const uint32_t b = km1 >> A.g_shift; lo1 = bkt[b]; hi1 = bkt[b + 1]; }
        km2 = i + 128u < se ? Q.kmer[i + 128u] : 0u;
        uint32_t qp = 0, qm = 0;
        if (EMIT && i < se) { qp = Q.pos[i]; qm = Q.meta[i]; }
        const uint32_t nst = (hi - lo + 63u) >> 6;
        if (!EMIT) visited += hi - lo;
        uint32_t pre = nst;      // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(pre, o); if (lane >= o) pre += y; }
        const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)pre, 63);
        pre -= nst;
        uint32_t ns[GSL_AHEAD], nx[GSL_AHEAD], nh[GSL_AHEAD], nk[GSL_AHEAD]; unsigned long long nv[GSL_AHEAD];
#define GSL_FETCH(t, u) do { \
            ns[u] = 0; nx[u] = 0; nh[u] = 0; nk[u] = 0xFFFFFFFFu; nv[u] = 0ull; \
            if ((t) < T) { \
                const unsigned long long own = __ballot(nst != 0 && pre <= (t)); \
                ns[u] = 63u - (uint32_t)__clzll((long long)own); \
                nx[u] = (uint32_t)__builtin_amdgcn_readlane((int)lo, (int)ns[u]) + 64u * ((t) - (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)ns[u])); \
                nh[u] = (uint32_t)__builtin_amdgcn_readlane((int)hi, (int)ns[u]); \
                if (nx[u] + (uint32_t)lane < nh[u]) { nk[u] = x_key[nx[u] + lane]; nv[u] = x_val[nx[u] + lane]; } \
            } } while (0)
#pragma unroll
        for (uint32_t u = 0; u < GSL_AHEAD; u++) GSL_FETCH(u, u);
        for (uint32_t t0 = 0; t0 < T; t0 += GSL_AHEAD) {
            uint32_t cs[GSL_AHEAD], cx[GSL_AHEAD], ch[GSL_AHEAD], ck[GSL_AHEAD]; unsigned long long cv[GSL_AHEAD];
#pragma unroll
            for (uint32_t u = 0; u < GSL_AHEAD; u++) { cs[u] = ns[u]; cx[u] = nx[u]; ch[u] = nh[u]; ck[u] = nk[u]; cv[u] = nv[u]; }
#pragma unroll
            for (uint32_t u = 0; u < GSL_AHEAD; u++) GSL_FETCH(t0 + GSL_AHEAD + u, u);
            // the pairs of the group's entries first, all four steps' LDS reads in flight together (they depend on nothing but the entries), then the steps one by one:
            // what is sequential - a pair's cursor - is only in the second part
            uint32_t slots[GSL_AHEAD]; uint2 bps[GSL_AHEAD];
#pragma unroll
            for (uint32_t u = 0; u < GSL_AHEAD; u++) bps[u] = s_bp[(uint32_t)(cv[u] >> 53) & 7u];      // (every lane reads: a lane without an entry has v = 0 - the block's first word; an entry's reference is of this block)
#pragma unroll
            for (uint32_t u = 0; u < GSL_AHEAD; u++) {
                const uint32_t skm = (uint32_t)__builtin_amdgcn_readlane((int)km, (int)cs[u]);
                const uint32_t b = (uint32_t)(cv[u] >> 48) & 31u, rk = bps[u].y + (uint32_t)__popc(bps[u].x & ~(~0u << b)) - B.rank_lo;
                const bool ok = cx[u] + (uint32_t)lane < ch[u] && ck[u] == skm && ((bps[u].x >> b) & 1u) && rk < P;      // (steps past the batch's last: nk = 0xFFFFFFFF is no k-mer, nh = 0)
                slots[u] = ok ? rk : 0xFFFFFFFFu;
            }
#pragma unroll
            for (uint32_t u = 0; u < GSL_AHEAD; u++) {
                if (t0 + u >= T) break;
                const uint32_t s = cs[u]; const unsigned long long v = cv[u];
                const uint32_t slot = slots[u];
                const bool valid = slot != 0xFFFFFFFFu;
                if (!__any(valid)) continue;
                const uint32_t jl = c0 - sb + s;      // the step's seed within the slice
                if (!EMIT) {      // no order needed: one LDS atomic per anchor, one for the seed's bit
                    if (valid) { atomicAdd(&s_cur[slot], 1u); atomicOr(&s_bm[(jl >> 5) * (pc + 1u) + slot], 1u << (jl & 31u)); }
                    continue;
                }
                const uint32_t sqp = (uint32_t)__builtin_amdgcn_readlane((int)qp, (int)s), sqm = (uint32_t)__builtin_amdgcn_readlane((int)qm, (int)s);
                const uint32_t rmeta = (uint32_t)((((v >> 33) & 0x7FFFull) << 1) | (v & 1ull));      // ref contig << 1 | (fwd < rc)
                const uint4 av = make_uint4(sqp, (uint32_t)(v >> 1), (rmeta & ~1u) | ((rmeta ^ sqm) & 1u), sqm >> 1);
                const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)slot, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);      // (lane 0: no lane before it)
                const bool same = valid && prev == slot;      // not the first lane of its (seed, reference) group
                const bool dup = __ballot(same) != 0;
                const uint2 cs = s_cs[valid ? slot : 0u];      // (every lane reads - no branch around the read)
                const uint32_t base = cs.x;
                // chunk table: the group's first lane opens a new chunk when the seed's key is beyond the reach of the pair's open head (chunk_heads_kernel's rule; a
                // group cut by a step boundary meets its own head again: not beyond) - the new row's first anchor and, with it, the end of the row before
                if (((uint32_t)__builtin_amdgcn_readlane((int)unw, (int)(jl >> 5)) >> (jl & 31u)) & 1u) {
                    if (valid && !same) {
                        const unsigned long long key1 = (((unsigned long long)(sqm >> 1) << 32) | sqp) + 1ull;
                        if (key1 > s_lim[slot]) {
                            const uint32_t row = s_rows[slot], idxc = base < A.cap ? base : A.cap;
                            if (row < Q.rows) {
                                uint2* r = A.chunks + ((size_t)B.row_off + (size_t)slot * Q.rows + row);
                                r->x = idxc;
                                if (row > 0) r[-1].y = idxc;
                            } else atomicOr(A.err, 1u);
                            s_rows[slot] = row + 1u; s_lim[slot] = key1 + FRAGMENT_LENGTH;
                        }
                    }
                }
                if (!dup) {      // every valid lane has a pair of its own (a reference holds a k-mer once, nearly always): nothing to order between lanes
                    // (no capacity test on this path: the count walk's total was held against the capacity before this kernel started - see the guard at its top)
                    if (!STAGE) { if (valid) { s_cs[slot].x = base + 1u; A.anc[base] = av; } continue; }
                    const uint32_t t3 = base & GSL_LM;
                    if (valid) { s_cs[slot].x = base + 1u; s_line[t3 * pc + slot] = av; }
                    // complete lines leave TOGETHER: four consecutive lanes write one pair's 64 bytes (one request per line where a lane writing its own line makes four)
                    const bool fl = valid && t3 == GSL_LM;
                    const unsigned long long fm = __ballot(fl);
                    if (fm) {
                        if (fl) {      // (from the pair's first anchor in this slice on - what is before belongs to the previous slice's wave)
                            const uint32_t st0 = cs.y, lb = base & ~GSL_LM, ft = lb >= st0 ? 0u : st0 - lb;
                            s_fl[__popcll(fm & ((1ull << lane) - 1ull))] = make_uint2(slot | (ft << 16), lb);
                        }
                        lds_wave_sync();
                        const uint32_t nf = (uint32_t)__popcll(fm), t = (uint32_t)lane & GSL_LM;
                        for (uint32_t g0 = 0; g0 < nf; g0 += 64u / GSL_LW) {
                            const uint32_t r2 = g0 + ((uint32_t)lane >> GSL_LS);
                            if (r2 < nf) { const uint2 f = s_fl[r2]; if (t >= (f.x >> 16)) A.anc[f.y + t] = s_line[t * pc + (f.x & 0xFFFFu)]; }
                        }
                    }
                    continue;
                }
                const uint32_t st0 = cs.y;
                // a reference that holds the k-mer several times sits in consecutive lanes: the group's anchors take consecutive places
                const uint32_t next = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)slot, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);      // (lane 63: no lane after it)
                const bool last = valid && next != slot;
                const unsigned long long starts = __ballot(valid && !same);
                const unsigned long long upto = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
                const uint32_t jj = valid ? (uint32_t)lane - (63u - (uint32_t)__clzll((long long)upto)) : 0u;
                const unsigned long long ends = __ballot(last);
                const uint32_t tail = valid ? (uint32_t)__ffsll((long long)(ends >> lane)) - 1u : 0u;      // lanes of the group after this one
                const uint32_t d = base + jj, e = d + tail, lastline = e >> GSL_LS;
                if (!STAGE) {
                    if (valid) { if (d < A.cap) A.anc[d] = av; else atomicOr(A.err, 2u); }
                    lds_wave_sync();
                    if (last) s_cs[slot].x = e + 1u;
                    lds_wave_sync();
                    continue;
                }
                // (A) a group that runs past the line being staged: its first lane sends out what the line holds so far
                if (valid && jj == 0 && (base >> GSL_LS) != lastline && e < A.cap) {
                    const uint32_t f0 = (base & ~GSL_LM) > st0 ? (base & ~GSL_LM) : st0;
                    for (uint32_t t = f0; t < base; t++) A.anc[t] = s_line[(t & GSL_LM) * pc + slot];
                }
                lds_wave_sync();
                // (B) anchors of the group's last line are staged, the others complete their lines and go out directly
                if (valid) {
                    if ((d >> GSL_LS) == lastline) s_line[(d & GSL_LM) * pc + slot] = av;
                    else if (e < A.cap) A.anc[d] = av;
                }
                lds_wave_sync();
                // (C) the group's last lane moves the cursor and sends the line out if the group completed it
                if (last) {
                    if (e >= A.cap) atomicOr(A.err, 2u);
                    else if ((e & GSL_LM) == GSL_LM) {
                        const uint32_t f0 = (e & ~GSL_LM) > st0 ? (e & ~GSL_LM) : st0;
                        for (uint32_t t = f0; t <= e; t++) A.anc[t] = s_line[(t & GSL_LM) * pc + slot];
                    }
                    s_cs[slot].x = e + 1u;
                }
                lds_wave_sync();
            }
        }
#undef GSL_FETCH
    }
    }
    }      // the entry's index blocks
    lds_wave_sync();
    if (!EMIT) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o);
        if (lane == 0 && visited) atomicAdd((unsigned long long*)(A.err + 18), visited);
        for (uint32_t j = lane; j < P; j += 64) {
            const uint32_t c = s_cur[j];
            A.cnt[rec0 + j] = c;
            if (c) atomicAdd(&A.pair_cnt[B.pair_off + j], c);
        }
        for (uint32_t x = lane; x < P * GSL_WORDS; x += 64) { const uint32_t j = x / GSL_WORDS, w = x % GSL_WORDS; A.bm[(size_t)(rec0 + j) * GSL_WORDS + w] = s_bm[w * (pc + 1u) + j]; }
    } else if (STAGE) {
        for (uint32_t j = lane; j < P; j += 64) {      // what is left in the lines: the pairs' last anchors of the slice
            const uint32_t c = s_cs[j].x, st0 = s_cs[j].y;
            if (c > st0 && (c & GSL_LM) && c <= A.cap) {
                const uint32_t f0 = ((c - 1u) & ~GSL_LM) > st0 ? ((c - 1u) & ~GSL_LM) : st0;
                for (uint32_t t = f0; t < c; t++) A.anc[t] = s_line[(t & GSL_LM) * pc + j];
            }
        }
    }
}

// Per pair, slice after slice: the first anchor of every (pair, slice) and the state of the pair's chunk table at the slice's start. One wave per (entry, 64 of its
// pairs), a lane per pair; the slice's seed keys (q contig << 32 | q pos) and the lanes' bitmaps sit in LDS, the next slice's are loaded while this one is walked. A
// chunk head = the first seed WITH an anchor whose key is beyond the previous head's + FRAGMENT_LENGTH (chunk_heads_kernel's rule on anchors: an anchor carries its
// seed's key): which seeds are heads follows from the bitmaps alone; the emit walk, which knows the anchors' indices, writes the rows.
constexpr uint32_t GSL_HK = GSL_SEEDS / 64;      // keys per lane
__global__ __launch_bounds__(64) void gsl_heads_kernel(GslArgs A) {
    __shared__ unsigned long long s_key[GSL_SEEDS];
    __shared__ uint32_t s_bm[64 * (GSL_WORDS + 1)];
    __shared__ uint32_t s_un[GSL_WORDS];      // the slice's seeds that head a chunk of some pair of the group
    const int lane = threadIdx.x;
    const uint32_t e = blockIdx.x;
    const BatchQ B = A.bq[e];
    const uint32_t P = B.rank_hi - B.rank_lo, j = blockIdx.y * 64u + (uint32_t)lane;
    if (blockIdx.y * 64u >= P) return;
    const SketchDesc Q = A.qd[B.q];
    const bool act = j < P;
    const uint32_t nsl = (Q.n + GSL_SEEDS - 1u) / GSL_SEEDS;
    const uint32_t p = B.pair_off + j;
    const uint2 eb = A.ebase[e];
    const bool guard = A.err[5] || *(const unsigned long long*)(A.err + 16) > A.cap;      // the attempt is rerun: no chunk tables
    uint32_t cursor = act ? A.pstart[p] : 0u;
    const uint32_t pend = act ? A.pstart[p + 1] : 0u;
    const bool live = act && !guard && pend - cursor >= MIN_ANCHORS;      // fewer: no chain can form, no chunk table
    uint32_t rows = 0; unsigned long long lim1 = live ? 0ull : ~0ull;      // (a pair without a chunk table: no key is beyond the limit, the emit walk opens no chunk)
    uint32_t* my_bm = s_bm + lane * (GSL_WORDS + 1);
    // a slice's inputs - the seeds' positions and contigs, the (pair, slice)'s anchor count and bitmap - are requested TWO slices ahead (two register sets, taken in turn): a
    // lane's walk over its pair's ~160 slices is a chain of these loads, and one slice of look-ahead left every step waiting for most of a memory round trip
    struct HSlice { uint32_t kp[GSL_HK], kmt[GSL_HK]; uint4 bw[GSL_WORDS / 4]; uint32_t c; };
    HSlice S0, S1;
    auto load_slice = [&](HSlice& X, uint32_t sl) {
        const uint32_t sb = sl * GSL_SEEDS;
#pragma unroll
        for (uint32_t u = 0; u < GSL_HK; u++) { const uint32_t i = sb + u * 64u + (uint32_t)lane; X.kp[u] = 0; X.kmt[u] = 0; if (i < Q.n) { X.kp[u] = Q.pos[i]; X.kmt[u] = Q.meta[i]; } }
        const size_t r = (size_t)eb.x + (size_t)sl * P + j;
        X.c = 0;
#pragma unroll
        for (uint32_t u = 0; u < GSL_WORDS / 4; u++) X.bw[u] = make_uint4(0, 0, 0, 0);
        if (act) {
            X.c = A.cnt[r];
            const uint4* __restrict__ src = (const uint4*)(A.bm + r * GSL_WORDS);
#pragma unroll
            for (uint32_t u = 0; u < GSL_WORDS / 4; u++) X.bw[u] = src[u];
        }
    };
    auto step = [&](HSlice& X, uint32_t sl) {
        const uint32_t sb = sl * GSL_SEEDS, ns = Q.n - sb > GSL_SEEDS ? GSL_SEEDS : Q.n - sb;
        const size_t r = (size_t)eb.x + (size_t)sl * P + j;
        lds_wave_sync();
        if (lane < (int)GSL_WORDS) s_un[lane] = 0;
#pragma unroll
        for (uint32_t u = 0; u < GSL_HK; u++) s_key[u * 64u + lane] = (((unsigned long long)(X.kmt[u] >> 1) << 32) | X.kp[u]) + 1ull;
#pragma unroll
        for (uint32_t u = 0; u < GSL_WORDS / 4; u++) { my_bm[4 * u] = X.bw[u].x; my_bm[4 * u + 1] = X.bw[u].y; my_bm[4 * u + 2] = X.bw[u].z; my_bm[4 * u + 3] = X.bw[u].w; }
        const uint32_t c = X.c;
        if (sl + 2 < nsl) load_slice(X, sl + 2);
        lds_wave_sync();
        if (act) {
            A.rec[r] = make_uint4(cursor, rows, (uint32_t)lim1, (uint32_t)(lim1 >> 32));
            if (live && c) {
                uint32_t from = 0;
                for (;;) {
                    {   // first seed of the slice at or after `from` whose key is beyond the open head's reach (keys ascend; lim1 = 0: the pair's first anchor)
                        uint32_t lo = from, hi = ns;
                        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (s_key[mid] > lim1) hi = mid; else lo = mid + 1; }
                        from = lo;
                    }
                    if (from >= ns) break;
                    uint32_t w = from >> 5, x = my_bm[w] & (~0u << (from & 31u));
                    while (!x && ++w < GSL_WORDS) x = my_bm[w];
                    if (!x) break;
                    const uint32_t jl = w * 32u + (uint32_t)__ffs((int)x) - 1u;      // ... that has an anchor: a head
                    lim1 = s_key[jl] + FRAGMENT_LENGTH; rows++;
                    atomicOr(&s_un[w], 1u << (jl & 31u));
                    from = jl + 1u;
                }
            }
            cursor += c;
        }
        lds_wave_sync();
        if (lane < (int)GSL_WORDS && s_un[lane]) atomicOr(&A.un[(size_t)(eb.y + sl) * GSL_WORDS + lane], s_un[lane]);
    };
    load_slice(S0, 0);
    if (nsl > 1) load_slice(S1, 1);
    for (uint32_t sl = 0; sl < nsl; sl += 2) {
        step(S0, sl);
        if (sl + 1 < nsl) step(S1, sl + 1);
    }
    if (act) {
        if (live && rows) {      // the last row's end (its first anchor comes from the emit walk)
            if (rows - 1u < Q.rows) A.chunks[(size_t)B.row_off + (size_t)j * Q.rows + rows - 1u].y = pend < A.cap ? pend : A.cap; else atomicOr(A.err, 1u);
        }
        A.n_chunks[p] = live ? (rows < Q.rows ? rows : Q.rows) : 0u;
    }
}

// The block table of every entry of a batch: one wave per entry sweeps its query's row of the pass matrix ONCE (the walks did, per wave: 160 slices of a 5 Mb query each
// read the whole row) and lists the index blocks that hold a reference of one of the entry's pairs - a block whose passing references' ranks [run, run + n) meet the entry's
// [rank_lo, rank_hi) - each with its 256 pass bits, the passing references before it and where its index entries start. No bound on the number of blocks.
__global__ __launch_bounds__(64) void gsl_blocks_kernel(const BatchQ* __restrict__ bq, uint32_t n_entries, const uint8_t* __restrict__ pass, uint32_t n_refs, uint32_t g_blocks,
                                                        const unsigned long long* __restrict__ g_base, uint32_t* __restrict__ blk_tab, uint32_t* __restrict__ blk_cnt, uint32_t blk_cap) {
    static_assert(BSI_BLOG == 8, "four ballots of 64 references = one index block");
    const uint32_t e = blockIdx.x;
    if (e >= n_entries) return;
    const int lane = threadIdx.x;
    const BatchQ B = bq[e];
    const uint8_t* __restrict__ row = pass + (size_t)B.q * n_refs;
    uint32_t* __restrict__ out = blk_tab + (size_t)e * blk_cap * GSL_BT_WORDS;
    uint32_t run = 0, cnt = 0;
    for (uint32_t blk = 0; blk < g_blocks && run < B.rank_hi; blk++) {
        uint8_t f[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t r = (blk * 4u + u) * 64u + (uint32_t)lane; f[u] = r < n_refs ? row[r] : (uint8_t)0; }
        unsigned long long m[4]; uint32_t n_here = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) { m[u] = __ballot(f[u] != 0); n_here += (uint32_t)__popcll(m[u]); }
        if (n_here && run + n_here > B.rank_lo) {
            if (cnt < blk_cap) {
                uint32_t w = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) { if (lane == 2 * u) w = (uint32_t)m[u]; if (lane == 2 * u + 1) w = (uint32_t)(m[u] >> 32); }
                const unsigned long long base = g_base[blk];
                if (lane == 8) w = blk; if (lane == 9) w = run; if (lane == 10) w = (uint32_t)base; if (lane == 11) w = (uint32_t)(base >> 32);
                if (lane < (int)GSL_BT_WORDS) out[(size_t)cnt * GSL_BT_WORDS + lane] = w;
            }
            cnt++;
        }
        run += n_here;
    }
    if (lane == 0) blk_cnt[e] = cnt < blk_cap ? cnt : blk_cap;      // (an entry of P pairs has references in at most P blocks and blk_cap >= min(blocks, the batch's largest P): never cut)
}
psk_status gsl_blocks_launch(const BatchQ* bq, uint32_t n_entries, const uint8_t* pass, uint32_t n_refs, uint32_t g_blocks, const unsigned long long* g_base,
                             uint32_t* blk_tab, uint32_t* blk_cnt, uint32_t blk_cap, hipStream_t st) {
    if (!n_entries) return PSK_OK;
    hipLaunchKernelGGL(gsl_blocks_kernel, dim3(n_entries), dim3(64), 0, st, bq, n_entries, pass, n_refs, g_blocks, g_base, blk_tab, blk_cnt, blk_cap);
    PSK_HIP(hipGetLastError());
    return PSK_OK;
}
psk_status gsl_count_launch(const GslArgs& A, hipStream_t st) {
    hipLaunchKernelGGL((gsl_walk_kernel<false, false>), dim3(A.n_tab), dim3(64), gsl_walk_lds(A, false), st, A);
    PSK_HIP(hipGetLastError());
    return PSK_OK;
}
psk_status gsl_heads_launch(const GslArgs& A, hipStream_t st) {
    PSK_HIP(hipMemsetAsync(A.un, 0, 4 * (size_t)GSL_WORDS * A.n_slices, st));
    hipLaunchKernelGGL(gsl_heads_kernel, dim3(A.n_entries, (A.p_cap + 63u) / 64u), dim3(64), 0, st, A);
    PSK_HIP(hipGetLastError());
    return PSK_OK;
}
psk_status gsl_emit_launch(const GslArgs& A, hipStream_t st) {
    if (A.stage) hipLaunchKernelGGL((gsl_walk_kernel<true, true>), dim3(A.n_tab), dim3(64), gsl_walk_lds(A, true), st, A);
    else hipLaunchKernelGGL((gsl_walk_kernel<true, false>), dim3(A.n_tab), dim3(64), gsl_walk_lds(A, true), st, A);
    PSK_HIP(hipGetLastError());
    return PSK_OK;
}
