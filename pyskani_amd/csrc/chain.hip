// The chain stage's host side: pair tables, scratch layout, the launch sequence over a batch of pairs (chain_run), capacity retries, psk_chain's entry points.
#include "chain_stages.h"
#include "query_parts.h"
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <algorithm>
#include <unordered_map>

// ------------------------------------------------------------------ host orchestration

// unindexed_ok: the sketch is described with its seed count and chunk-table rows although it has no k-mer index (rounds that join through the
// database-wide seed index: no kernel of theirs reads a per-sketch index)
SketchDesc make_desc(const psk_sketch* s, bool unindexed_ok) {
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

psk_status chain_layout(Lane* ctx, size_t n_pairs, size_t n_items, size_t n_rows, size_t n_bq, ChainBufs* L) {
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
psk_status chain_run(Lane* ctx, const ChainBufs& L, uint32_t n_pairs, size_t n_items, size_t n_rows, const psk_params& prm,
                     const psk_query_opts* o, const SketchDesc* d_qd, const SketchDesc* d_rd, uint64_t cap, bool wide, const Switches& sw, bool probe_ok) {
    hipStream_t st = ctx->stream;
    const int force_serial = sw.chain_serial.get() != nullptr;
    // misc[0..15] status / counts, misc[11] pairs for select_huge_kernel, misc[16..17] the 64-bit anchor total, misc[32..47] the group barriers: zeroed by pair_table_kernel
    const uint32_t gi = L.gi;
    hipLaunchKernelGGL(pair_table_kernel, dim3((uint32_t)(((size_t)gi + n_rows + 255) / 256)), dim3(256), 0, st, L.sbase, L.cbase, n_pairs, gi, (uint32_t)n_items, (uint32_t)n_rows, L.blk_pair, L.row_pair,
                       L.misc, L.lbcnt + n_items);
    ctx->t_begin(K_ANCHOR);
    if (wide) hipLaunchKernelGGL(anchor_count_kernel, dim3(gi), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, L.lbcnt, L.bsum, L.blk_pair);
    const uint32_t gi4 = (gi + JT - 1) / JT;
    uint32_t n_sum = gi;
    const char* jp_env = sw.join_pairs.get();      // "1" / "0" force / forbid the pair-major join (tests, A/B)
    const bool gsi_join = !wide && (L.g_key || L.b_key) && L.d_pass && L.n_bq;      // (the PLAN decides - PSK_GSI_JOIN is read there, once per round: the round's sketches carry no k-mer index to fall back on)      // (every batch of a round that was planned for it: its sketches carry no k-mer index)
    const bool join_pairs = !wide && (gsi_join || (jp_env ? jp_env[0] == '1' : (n_pairs >= 16384 && n_items / n_pairs < 2048)));
    const bool gsl = gsi_join && L.gsi_slice;      // one wave per (query, slice of its seeds): count walk -> scan over the pairs -> heads -> emit walk
    const bool gsi_one = gsi_join && !gsl && L.gsi_onepass && cap >= n_items + n_items / 8 + 8 * ((size_t)n_pairs + 1);      // (gsi_room_kernel's layout fits)
    bool probe_local = false;
    GsiJoinArgs GA{};
    GslArgs GL{};
    // (the contig join's pass bitset in LDS: the query's whole row where a wave may walk the database-wide index, the four words of one block where only blocks are walked)
    const uint32_t gsi_nw = (L.g_key && !L.gsi_slice) ? std::max(4u, (L.n_refs + 63u) / 64u) : 4u;      // (never fewer than one block's four words: a wave of the same launch may walk blocks)
    const size_t gsi_lds_row = 8 * (size_t)gsi_nw + 4 * (size_t)((gsi_nw + 1u) & ~1u), gsi_lds_count = gsi_lds_row + 4 * (size_t)L.p_cap, gsi_lds_emit = gsi_lds_row + 4 * (size_t)L.p_cap * 5;
    if (gsi_join) {
        GA.bq = L.bq; GA.pass = L.d_pass; GA.n_refs = L.n_refs; GA.qd = d_qd; GA.g_key = L.g_key; GA.g_val = L.g_val; GA.g_bucket = L.g_bucket; GA.g_shift = L.g_shift;
        GA.b_key = L.b_key; GA.b_val = L.b_val; GA.b_bucket = L.b_bucket; GA.b_shift = L.b_shift; GA.b_nb1 = L.b_nb1; GA.b_blocks = L.b_blocks; GA.b_max = L.b_max;
        GA.pair_cnt = L.big_list; GA.pstart = L.pstart; GA.cap = (uint32_t)cap; GA.err = L.misc; GA.p_cap = L.p_cap; GA.nw_lds = gsi_nw;
        // the entries' block tables: which blocks of the blocked index hold a reference of an entry's pairs (one sweep of the query's pass row per entry; both index joins read them)
        const uint32_t t_blocks = gsl ? L.g_blocks : L.b_blocks;
        const unsigned long long* t_base = gsl ? L.g_base : L.b_base;
        const uint32_t bcap = std::max(1u, std::min<uint32_t>(t_blocks, std::min<uint32_t>(GSI_PMAX, std::max(1u, L.p_cap))));      // (an entry of P pairs has references in at most P blocks; p_cap >= every entry's P)
        uint32_t* d_btab = nullptr; uint32_t* d_bcnt = nullptr;
        if (t_blocks) {
            const size_t o_bc = al256(4 * (size_t)L.n_bq * bcap * GSL_BT_WORDS);
            PSK_TRY(ctx->q_k.reserve(o_bc + 4 * (size_t)L.n_bq + 256));
            d_btab = (uint32_t*)ctx->q_k.p; d_bcnt = (uint32_t*)((char*)ctx->q_k.p + o_bc);
            PSK_TRY(gsl_blocks_launch(L.bq, L.n_bq, L.d_pass, L.n_refs, t_blocks, t_base, d_btab, d_bcnt, bcap, st));
        }
        GA.blk_tab = d_btab; GA.blk_cnt = d_bcnt; GA.blk_cap = bcap;
        if (gsl) {
            GL.bq = L.bq; GL.n_entries = L.n_bq; GL.tab = L.gsl_tab; GL.n_tab = L.gsl_n_tab; GL.ebase = L.gsl_ebase; GL.pass = L.d_pass; GL.n_refs = L.n_refs; GL.qd = d_qd;
            GL.g_key = L.g_key; GL.g_val = L.g_val; GL.g_bucket = L.g_bucket; GL.g_shift = L.g_shift; GL.g_nb1 = L.g_nb1; GL.g_blocks = L.g_blocks; GL.g_base = L.g_base; GL.blk_tab = d_btab; GL.blk_cnt = d_bcnt; GL.blk_cap = bcap; GL.cnt = L.gsl_cnt; GL.rec = L.gsl_rec; GL.bm = L.gsl_bm; GL.un = L.gsl_un; GL.n_slices = L.gsl_n_slices;
            GL.pair_cnt = L.big_list; GL.pstart = L.pstart; GL.cap = (uint32_t)cap; GL.err = L.misc; GL.p_cap = L.p_cap; GL.chunks = L.chunks; GL.n_chunks = L.nch;
            { const char* e = sw.gsl_stage.get(); GL.stage = e ? atoi(e) : 1; }      // (A/B: every anchor its own 16-byte store)
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
        const bool pl_off = sw.probe_local.get() && sw.probe_local.get()[0] == '0';
        probe_local = probe_ok && !pl_off;
        if (probe_ok) hipLaunchKernelGGL(anchor_join_probe_kernel, dim3(nb), dim3(256), 0, st, L.pairs, L.sbase, order, n_pairs, L.lbcnt, L.bsum, L.misc + 5,
                                         probe_local ? L.aoff : (uint32_t*)nullptr, probe_local ? L.big_list : (uint32_t*)nullptr);      // (big_list: free until select runs)
        else hipLaunchKernelGGL(anchor_join_pairs_kernel, dim3(nb), dim3(256), 0, st, L.pairs, L.sbase, order, n_pairs, L.lbcnt, L.bsum, L.misc + 5);
        n_sum = nb;
    }
    // many mid-sized pairs (all-vs-all): the join counts every pair's anchors, one workgroup per pair then emits with a running offset
    // (anchor_emit_pairs_kernel) instead of a scan over all items; PSK_EMIT_PAIRS=1 / 0 force / forbid it (tests, A/B)
    const char* ep_env = sw.emit_pairs.get();
    const bool emit_pairs = !wide && !join_pairs && n_items >= 2 * ((size_t)n_pairs + 1) &&      // (its 64-bit pair offsets live in the per-item offsets array)
                            (ep_env ? ep_env[0] == '1' : (n_pairs >= 1024 && n_items / n_pairs >= 1024 && n_items / n_pairs <= (1u << 17)));
    uint32_t* pair_cnt = L.live;      // free until the live list is built
    if (emit_pairs) PSK_HIP(hipMemsetAsync(pair_cnt, 0, 4 * ((size_t)n_pairs + 1), st));
    // workgroups of one pair per XCD turn (0 = contiguous eighths of the grid; PSK_XCD_GROUP overrides): see xcd_group_block_id
    const int xg_env = sw.xcd_group.get() ? atoi(sw.xcd_group.get()) : -1;
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
    PSK_TRY(ctx->q_e.reserve(na * (8 + 4 * 7 + 1) + 512 + 4 * 2 * BIG_GMAX * (BIG_GROUPS + 64) + 4 * BIG_GROUPS));   // select_big_kernel / select_huge_kernel scratch
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
    { const char* e = sw.dp_prune.get(); A.dp_prune = e && e[0] == '0' ? 0 : 1; }      // (read per call: tests switch it within a process)
    // the per-pair emit also writes the chunk table unless the pointer-chase builder is asked for (PSK_CHUNK_HOPS) or PSK_EMIT_HEADS=0
    const char* hops_env = sw.chunk_hops.get();
    const bool use_hops = gsi_join ? false : hops_env ? hops_env[0] != '0' : ((n_pairs < 1024 && n_items / n_pairs > 4096) || n_items / n_pairs > (1u << 20));      // (few pairs of a contig's few hundred seeds: one wave per pair walks its heads - one launch instead of two)
    const bool emit_heads_off = sw.emit_heads.get() && sw.emit_heads.get()[0] == '0';
    const bool emit_heads = emit_pairs && !use_hops && !emit_heads_off;
    // Gb-scale pairs: the walk in ITEM space where the join left per-item offsets (chunk_hops_items_kernel); PSK_HOPS_ITEMS=1 / 0 force / forbid (tests, A/B)
    const char* hi_env = sw.hops_items.get();
    const bool hops_items = use_hops && !gsl && !emit_pairs && !join_pairs && !sw.hops_unsliced.get() && n_items <= 0x7FFFFFFFull &&
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
                    { const char* e = sw.gsi_stage.get(); GA.stage = e && e[0] == '0' ? 0 : 1; }      // (read per batch: tests switch it within a process)
                    hipLaunchKernelGGL(gsi_join_kernel<true>, dim3(L.n_bq), dim3(64), gsi_lds_emit + (GA.stage ? 16 * (size_t)L.p_cap : 0), st, GA); }
    else if (wide) hipLaunchKernelGGL(anchor_emit_kernel, dim3(gi), dim3(256), 0, st, L.pairs, L.sbase, n_pairs, (uint32_t)n_items, L.lbcnt, L.aoff, anc, (uint32_t)cap, L.misc, L.blk_pair);
    else if (emit_pairs) hipLaunchKernelGGL(anchor_emit_pairs_kernel, dim3(n_pairs), dim3(EP_T), 0, st, L.pairs, L.sbase, n_pairs, L.lbcnt, poff, anc, (uint32_t)cap, L.misc,
                                            L.cbase, emit_heads ? L.chunks : (uint2*)nullptr, L.nch);
    else {
        // k-mers with many matches (Gb-scale pairs): anchor-major emit; PSK_EMIT_EXPAND=1 / 0 force / forbid (tests, A/B)
        const char* ex_env = sw.emit_expand.get();
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
        if (n_items / n_pairs > (1u << 20) && !sw.hops_unsliced.get()) {      // Gb-scale pairs: HOP_SLICES waves per pair, count then write
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
        const bool rs_off = sw.row_sort.get() && sw.row_sort.get()[0] == '0';
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
        const char* le = sw.chain_lane.get();
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
                const char* xt_env = sw.lane_xtrees.get();
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
    const bool qd_off = sw.chain_quad_deep.get() && sw.chain_quad_deep.get()[0] == '0';
    // a launch of few rows is as slow as its longest chunk: one wave per row with the window in registers (PSK_CHAIN_WAVE_REG=1 / 0 force / forbid: tests, A/B)
    const char* wr_env = sw.chain_wave_reg.get();
    const bool wave_reg = !A.lane_dp && !force_serial && A.band < 128 && (wr_env ? wr_env[0] == '1' : n_rows <= 2048);      // (one wave per SIMD up to 1 024 rows: 0.29 us per anchor of the longest chunk; the four-lanes-per-chunk kernel needs 0.9 us but takes 16 rows per wave)
    if (wave_reg) {
        const dim3 g((uint32_t)((n_rows + CHAIN_WAVES - 1) / CHAIN_WAVES)), b(64 * CHAIN_WAVES);
        if (A.band < 64) hipLaunchKernelGGL(chain_wave_reg_kernel<1>, g, b, 0, st, A);
        else hipLaunchKernelGGL(chain_wave_reg_kernel<2>, g, b, 0, st, A);
    }
    const bool quad_deep = !wave_reg && !A.lane_dp && !force_serial && !qd_off && A.band <= 4 * QD && !(sw.chain_lane.get() && sw.chain_lane.get()[0] == '0');
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
        const bool tiny_off = sw.select_tiny.get() && sw.select_tiny.get()[0] == '0';
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
            BA.huge_list = L.huge_list; BA.huge_count = L.misc + 11; BA.ctr = L.misc + 32; BA.huge_c = BA.parts + (size_t)2 * BIG_GMAX * (BIG_GROUPS + 64);
            const uint32_t solo = sw.big_solo.get() ? (uint32_t)std::max(atoi(sw.big_solo.get()), CMAX) : BIG_SOLO;
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
                        if (const char* e = sw.huge_slots.get()) { per_cu = 1; cus = std::max(0, atoi(e)); }      // tests: pretend a smaller device
                        it = slots_of.emplace(ctx->device, (uint32_t)std::max(0, per_cu) * (uint32_t)std::max(0, cus)).first;
                    }
                    slots = it->second;
                }
                uint32_t nb = BIG_GMAX;
                const char* hm_env = sw.huge_min_seeds.get();      // (tests: batches of small pairs take the full-size launch and its mutex too)
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
    const char* rs_env = sw.reduce_small.get();
    const bool no_small = rs_env && rs_env[0] == '0';
    // tables of 65 .. 512 rows (pairs of ~5 Mb genomes: ~250): one wave per pair, eight rows per lane; PSK_REDUCE_WAVE=0 leaves them to the workgroup kernel (tests, A/B)
    const bool no_wave = sw.reduce_wave.get() && sw.reduce_wave.get()[0] == '0';
    R.wave_done = !no_wave && !no_small && L.rows_pair_max > 64u && n_rows / n_pairs <= 64u * RW_PER;
    // tables of <= 64 rows (contigs; the short pairs beside the others): one wave per pair, a row per lane (also without the live list: the few pairs of one contig's query)
    R.small_done = !no_small && (n_rows / n_pairs < 16 || R.wave_done);
    // contig batches with the mean ANI: pairs of up to four chunk rows by one lane each first (PSK_REDUCE_TINY=0: by a wave each)
    const bool no_tiny = sw.reduce_tiny.get() && sw.reduce_tiny.get()[0] == '0';
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

uint64_t anchor_cap_for(Lane* ctx, size_t n_items, bool sparse, bool gb_scale) {
    const uint64_t have = ctx->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) > 128 ? ctx->q_d.cap / (4 * CHAIN_ANCHOR_WORDS) - 128 : 0;     // anchors the per-anchor arrays already hold
    // non-repetitive genomes: at most ~one anchor per query seed; contigs against a whole database (sparse): a third of the (pair, seed) items match
    // (100 bytes of scratch per anchor: 2^30 items would reserve 136 GB otherwise; a batch that does not fit is rerun with the true total)
    // Gb-scale pairs: a seed has ~6.5 matches (chance 15-mer hits in 3 Gb beside the true one) - sized for that at once: the first batch used to overflow, and its
    // second attempt freed 7 GB to allocate 40 GB, which takes 1.5 s when the driver is still clearing memory a previous process released (profiles/r3/r3y_50x_alloc_trace.txt)
    const uint64_t want = gb_scale ? (uint64_t)n_items * 7 + 65536 : sparse ? (uint64_t)n_items / 2 + 65536 : (uint64_t)n_items + n_items / 4 + 65536;
    return std::min<uint64_t>(std::max(have, want), 0x7FFFFF00ull);
}

Switches Switches::read() {
    Switches s;
    auto take = [](SwitchVal& v, const char* name) {
        const char* e = getenv(name);
        v.set = e != nullptr; v.text[0] = 0;
        if (e) { strncpy(v.text, e, sizeof v.text - 1); v.text[sizeof v.text - 1] = 0; }
    };
#define X(field, name) take(s.field, name);
    PSK_SWITCHES(X)
#undef X
    return s;
}
psk_status chain_check(const ChainTail& T, uint32_t n_pairs, uint64_t* cap, bool* wide, bool* retry) {
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


// one launch sequence over an explicit list of (ref, query) pairs; out[p] in pair order
psk_status chain_batch(Lane* ctx, const HostPair* hp, uint32_t n_pairs, const psk_query_opts* o, psk_hit* out, const Switches& sw) {
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
    bool wide = sw.join_wide();
    for (int attempt = 0;; attempt++) {
        PSK_TRY(chain_run(ctx, L, n_pairs, (size_t)items, (size_t)rows, hp[0].q->params, o, d_desc, d_desc, cap, wide, sw));
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
    const Switches sw = Switches::read();      // ($PSK_*: once per call)
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
        psk_status rc = chain_batch(ctx, hp.data(), (uint32_t)hp.size(), o, out + b, sw);
        if (rc == PSK_ELIMIT && hp.size() > 1) {   // too many anchors for 32-bit offsets: halve the batch until single pairs
            std::vector<HostPair> todo(hp);
            size_t step = (todo.size() + 1) / 2;
            for (size_t s0 = 0; s0 < todo.size();) {
                const size_t n1 = std::min(step, todo.size() - s0);
                rc = chain_batch(ctx, todo.data() + s0, (uint32_t)n1, o, out + b + s0, sw);
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
