// The chain stage (lib.rs:640-657: chain_seeds of the query against every shortlisted reference) as launch sequences over BATCHES of pairs: what its stage files share.
//   join.hip     anchors of every pair (merge / probe / seed-index joins), their emit, the chunk tables
//   dp.hip       banded chaining DP per chunk, candidate chains
//   select.hip   greedy selection of non-overlapping chains per pair
//   reduce.hip   per-pair ANI / aligned fractions
//   chain.hip    the host side: scratch layout, the launch sequence (chain_run), retries
// Kernels are declared here and launched from chain.hip; every stage file holds its own device helpers (no relocatable device code).
#pragma once
#include "common.h"
#include "chain_dev.h"
#include "slice_join.h"

// One (reference, query) pair of a launch. Pairs may mix queries (query_many / all-vs-all).
struct PairDesc {
    const uint32_t* r_key; const uint64_t* r_pms;                             // ref index slice: k-mers ascending; r_pms = the seeds' pos<<32|meta in the same order
    const uint32_t* q_key; const uint32_t* q_perm;                            // query index slice: the join walks the query in k-mer order
    const uint32_t* q_pos; const uint32_t* q_meta;                            // query seeds, (contig,pos) order
    const uint32_t* q_kmer;                                                   // their k-mers, same order
    uint32_t q_nc, pad_;                                                      // kept contigs of the query
    const uint32_t* q_seed_pos_base;   // base of the query's store (q_contig_start holds offsets into it)
    const uint32_t* q_contig_start;
    uint64_t q_total_len, r_total_len;
    uint32_t r_n, q_n;
    const uint32_t* r_bucket; uint32_t r_bshift;                              // ref index bucket table (IndexStore::bucket)
    uint32_t r_tab_lines; const ProbeLine* r_tab;                             // ref probe table (null until built: ensure_probe)
};
// sbase[p] = first (pair, query seed) item of pair p in lb/cnt/aoff; cbase[p] = first row of pair p in the chunk table


__device__ __forceinline__ uint32_t find_le(const uint32_t* __restrict__ base, uint32_t n, uint32_t x) {
    uint32_t lo = 0, hi = n - 1;   // largest p in [0,n) with base[p] <= x
    while (lo < hi) { uint32_t mid = (lo + hi + 1) >> 1; if (base[mid] <= x) lo = mid; else hi = mid - 1; }
    return lo;
}

// pair of item x given the pair of the workgroup's first item: a short forward walk (a pair usually holds far more items
// than a workgroup has threads; pairs without items are stepped over)
__device__ __forceinline__ uint32_t pair_from_hint(const uint32_t* __restrict__ base, uint32_t n, uint32_t x, uint32_t p) {
    while (p + 1 < n && base[p + 1] <= x) p++;
    return p;
}

struct CountOf { __host__ __device__ uint32_t operator()(const uint2& v) const { return v.y; } };

constexpr int JOIN_WIN = 256;

struct PackedCount { __host__ __device__ uint32_t operator()(const uint2& v) const { return v.y >> 24; } };

constexpr int JT = 4;

struct Widen { __host__ __device__ unsigned long long operator()(const uint32_t& v) const { return v; } };

constexpr int EP_W = 1;                 // waves per workgroup (EP_T threads, JT x EP_T items per round). Measured per 10^5 pairs of the all-vs-all step: 8 waves 160 ms, 4: 139.6, 2: 136.6, 1: 134.5 - the fewer waves wait at the round's barrier for wave 0's walk from head to head, the better

constexpr int EP_T = 64 * EP_W;

constexpr int HEAD_WIN = 2048;

constexpr int HOP_WIN = 4096;

constexpr int HOP_SLICES = 32;

struct ChunkOut { uint32_t anchors, seeds, n_intervals, n_cand; uint32_t left, right; uint64_t cov_q; };

// One field of the 32-byte candidate-chain records {score, q0, q1, r0, r1, anchors, ref contig, state}: element i of the field at p[8 i]. The fields were eight arrays of
// their own until round 5: a chunk holds one or two candidates, so the selection read one 64-byte line PER FIELD per chunk - 1.0-1.9 kB of HBM traffic per candidate
// (profiles/r5/pmc_kernels.json) for 32 bytes of content, at 5 TB/s: its whole run time. As records a chunk's candidates are one line.
template <class T> struct Strided {
    T* p;
    __host__ __device__ __forceinline__ T& operator[](size_t i) const { return p[i * 8]; }
};

struct ChainArgs {
    const uint4* anc;      // anchors, array of (q pos, r pos, ref contig << 1 | reverse_match, q contig): a lane's chunk is one contiguous run of 16-byte records
    const uint2* chunks; const uint32_t* n_chunks; const uint32_t* cbase; uint32_t n_pairs, n_rows;
    const uint32_t* row_pair;   // pair of every row of the chunk table
    const uint32_t* row_order;  // rows by chunk length, longest first, rows without a chunk last (null: table order) - the DP kernels that put several chunks in a wave
    const PairDesc* pairs;
    ChunkOut* out;
    // serial-path scratch, one entry per anchor
    int32_t* sc_f; uint32_t *sc_ptr, *sc_root, *sc_depth, *sc_best;
    Strided<int32_t> c_score; Strided<uint32_t> c_q0, c_q1, c_r0, c_r1, c_n, c_state, c_rc;   // candidate chains, chunk s writes at [s, s + n_cand): fields of 32-byte records
    uint32_t two_c; int band; int force_serial; int lane_dp;
    uint32_t cap;      // anchors the arrays hold (chunk_seeds_kernel's bound on what a chunk row may point at)
    int dp_prune;      // the lane / quad DP kernels score the far part of the band only where it could win ($PSK_DP_PRUNE=0: always)
    uint32_t* ovf_list; uint32_t* ovf_count;   // rows the lane kernel hands to the wave kernel (more than LANE_TREES qualifying chain trees, >= 16 384 anchors)
    uint32_t* stats;   // [1] chunks / [3] pairs that took a serial fallback (rare paths only: a counter every wave bumps
                       // serialises the whole launch on one L2 address)
};

constexpr int RING = 128;   // power of two > CHAIN_BAND

constexpr int RMAX = 256;   // chain trees per chunk handled in LDS

constexpr int CHAIN_WAVES = 4;

constexpr int LANE_N = 24;          // predecessors held per lane (multiple of 4)

constexpr int LANE_WAVES = 2;

constexpr int LANE_TREES = 4;        // qualifying chain trees per chunk kept in registers

constexpr int LANE_XTREES = 12;

constexpr int QUAD_N = LANE_N / 4;

constexpr int QD = 21;            // own anchors per lane: bands up to 4 * QD = 84

constexpr int QD_NEAR = 2;        // entries per lane that are always scored (with the step's own anchors: the quad's last 8-11); the others only when they could win. Measured on the
                                  // 100 000 x 5 000 step: 5 -> 48.1 ms, 3 -> 44.5, 2 -> 42.4, 1 -> 41.1 (the far pass becomes more frequent as the near part shrinks)

constexpr int QD_RING = 128;      // root / depth ring per quad (power of two > 4 * QD + 3)

constexpr int CMAX = 1024;

struct SelArgs {
    const uint2* chunks; const uint32_t* n_chunks; const uint32_t* cbase; uint32_t n_pairs;
    Strided<int32_t> c_score; Strided<uint32_t> c_q0, c_q1, c_r0, c_r1, c_n, c_rc, c_state;
    ChunkOut* out; uint32_t two_c; int force_serial; uint32_t* stats;
    const uint32_t* live; const uint32_t* n_live;      // pairs that have a chunk table (every other pair has no candidate chain)
    uint32_t* rest_list; uint32_t* rest_count;         // select_tiny_kernel: the live pairs it did NOT take (what the wave kernel still has to visit)
    uint32_t* big_list; uint32_t* big_count;           // pairs with more than CMAX candidates, for select_big_kernel
    int tiny_done;                                     // pairs of at most TINY_ROWS chunks and TINY_CANDS candidates were selected by select_tiny_kernel
};

constexpr uint32_t TINY_ROWS = 4, TINY_CANDS = 8;

constexpr int CSMALL = 512;

struct BigArgs {
    SelArgs S; const uint32_t* pstart;
    unsigned long long* key;   // sort keys
    uint32_t *slot, *crow;     // candidate j -> global candidate slot, chunk row (generation order)
    uint32_t *idx, *pm, *pm2;  // payload of the reference-order sort; running max of r1 (double buffer)
    uint32_t *ord, *clist;
    uint8_t* conf;
    uint32_t *huge_list, *huge_count;   // pairs with more than BIG_SOLO candidates, listed by the solo kernel for the cooperative one
    uint32_t* huge_c;                   // ... and the candidates of the first BIG_GROUPS of them: the cooperative launch deals its workgroups by these weights
    uint32_t *ctr, *parts;              // per group: barrier counter; 2 x BIG_GMAX partial sums of the two ordered compactions
    uint32_t solo;                      // BIG_SOLO ($PSK_BIG_SOLO in tests)
};

constexpr int BIG_T = 1024;

constexpr uint32_t BIG_TILE = 4096;       // keys of one LDS-staged sort tile

constexpr uint32_t BIG_SOLO = 32768;      // up to here one workgroup per pair: a barrier between workgroups costs more than it divides

#ifndef BIG_GMAX_N
#define BIG_GMAX_N 256
#endif
constexpr uint32_t BIG_GMAX = BIG_GMAX_N;        // workgroups of the cooperative launch (co-resident: one per CU of the 512 slots the chip has for them; 128 -> 256: the group selection of an 8 x 3 Gb step 19.8 -> 14.6 ms, the sweeps of its sorts are spread over twice the workgroups)

constexpr uint32_t BIG_GROUPS = 16;       // pairs in flight in the cooperative launch

struct ReduceArgs {
    const ChunkOut* chunks; const uint32_t* n_chunks; const uint32_t* cbase;
    const uint32_t* pstart; const PairDesc* pairs;
    const uint32_t* pcnt;   // anchors per pair where pstart does not say (the one-pass index join: pstart = item offsets); null: pstart[p + 1] - pstart[p]
    const uint2* pair_qr;   // (query, reference) of every pair: travels with the hit (reserved, ref_index)
    const uint32_t* live; const uint32_t* n_live;   // pairs with a chunk table
    int small_done;                                 // chunk tables of <= 64 rows are reduced by pair_reduce_small_kernel
    int wave_done;                                  // ... and those of 65 .. 64 RW_PER rows by pair_reduce_wave_kernel
    int tiny_done;                                  // ... and those of 1 .. 4 rows (contig pairs, mean ANI) by pair_reduce_tiny_kernel, a lane per pair
    int k, median, robust; double min_af;
    psk_hit* hits;
    double* big_vals;   // 2 * rows(+pad) doubles per launch: sort space for pairs with more than RED_CAP chunk values
};

constexpr int RED_CAP = 4096;   // chunk ANI values sortable in LDS (genomes up to ~80 Mb at 20 kb chunks)

constexpr int RED_SMALL = 1024;

// groups of 64 chunk rows one wave reduces (pair_reduce_wave_kernel)
constexpr int RW_PER = 8;      // 512 rows: a 5 Mb genome has ~250 chunks, and the pairs just past 256 rows took a workgroup each (15 of the 19 ms of reduction per 10 000 x 10 000 step)

struct IsLivePair { const uint32_t* nch; __host__ __device__ bool operator()(const uint32_t& p) const { return nch[p] != 0; } };

struct GsiJoinArgs {
    const BatchQ* bq; const uint8_t* pass; uint32_t n_refs; const SketchDesc* qd;
    const uint32_t* g_key; const unsigned long long* g_val; const uint32_t* g_bucket; int g_shift;
    // b_blocks > 0: the index is ALSO there in blocks of 2^BSI_BLOG references with a bucket table of b_nb1 entries each (psk_db::bsi_*): a wave whose query has its passing
    // references in at most b_max of them walks those blocks (a run of the database-wide index holds ~1 % of all genomes by chance), any other wave the database-wide index
    const uint32_t* b_key; const unsigned long long* b_val; const uint32_t* b_bucket; int b_shift; uint32_t b_nb1, b_blocks, b_max;
    const uint32_t* blk_tab; const uint32_t* blk_cnt; uint32_t blk_cap;      // per entry: the blocks that hold one of its pairs' references (slice_join.h: gsl_blocks_kernel; rows of GSL_BT_WORDS words)
    uint32_t nw_lds;      // 64-reference words of the pass bitset the launch's LDS holds: the whole row where the database-wide index may be walked, four (one block) otherwise
    uint32_t* pair_cnt; const uint32_t* pstart; uint4* anc; uint32_t cap; uint32_t* err;
    uint32_t p_cap;      // most pairs any entry of the batch holds, rounded up: what the cursor arrays in LDS are sized for (<= GSI_PMAX)
    uint2* chunks; uint32_t* n_chunks;      // EMIT: the pairs' chunk tables, written by the same walk (rows at entry.row_off + slot * query rows)
    // ONE PASS (no COUNT pass, no scan): pstart holds the pairs' ITEM offsets, stretched (gsi_room_kernel: room for nine anchors per eight query seeds and
    // eight more), the walk leaves every pair's count in pair_cnt and adds the batch's total to *total; a pair that would need more room (a reference that
    // holds the query's k-mers several times over) raises err bit 2 and the batch is rerun with the two passes
    int onepass; unsigned long long* total;
    int stage;      // EMIT: anchors leave in pairs of 32 bytes (an even-indexed anchor waits in LDS for its neighbour); 0: every anchor its own 16-byte store ($PSK_GSI_STAGE=0)
};

#ifndef LANE_NEAR_N
#define LANE_NEAR_N 3
#endif
constexpr int LANE_NEAR = LANE_NEAR_N;    // predecessors that are always scored; the rest of the band only where it could win. >= 3: the far loop reads the register window only, so it must start behind the step's own four anchors. Measured on the 10 000 x 10 000 step's DP: 8: 147, 6: 136, 5: 131, 4: 126, 3: 122 ms

// ------------------------------------------------------------------ kernels (defined in join.hip / dp.hip / select.hip / reduce.hip)
__global__ __launch_bounds__(256) void pair_table_kernel(const uint32_t* __restrict__ sbase, const uint32_t* __restrict__ cbase, uint32_t n,
                                                         uint32_t n_tiles, uint32_t n_items, uint32_t n_rows,
                                                         uint32_t* __restrict__ blk_pair, uint32_t* __restrict__ row_pair, uint32_t* __restrict__ misc, uint2* __restrict__ lb_tail);
__global__ __launch_bounds__(256) void anchor_count_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                           uint32_t n_pairs, uint32_t n_items,
                                                           uint2* __restrict__ lbcnt_out, unsigned long long* __restrict__ block_sum,
                                                           const uint32_t* __restrict__ blk_pair);
__global__ __launch_bounds__(256) void anchor_join4_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                           uint32_t n_pairs, uint32_t n_items, uint32_t n_tiles,
                                                           uint2* __restrict__ item_out, unsigned long long* __restrict__ block_sum,
                                                           uint32_t* __restrict__ need_wide, const uint32_t* __restrict__ blk_pair,
                                                           uint32_t* __restrict__ pair_cnt, uint32_t xcd_group);
__global__ __launch_bounds__(256) void anchor_join_pairs_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                const uint32_t* __restrict__ order, uint32_t n_pairs,
                                                                uint2* __restrict__ item_out, unsigned long long* __restrict__ block_sum,
                                                                uint32_t* __restrict__ need_wide);
__global__ __launch_bounds__(256) void anchor_join_probe_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                const uint32_t* __restrict__ order, uint32_t n_pairs,
                                                                uint2* __restrict__ item_out, unsigned long long* __restrict__ block_sum,
                                                                uint32_t* __restrict__ need_wide, uint32_t* __restrict__ aoff_local, uint32_t* __restrict__ pair_cnt);
__global__ __launch_bounds__(256) void anchor_emit_packed4_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                  uint32_t n_pairs, uint32_t n_items, uint32_t n_tiles,
                                                                  const uint2* __restrict__ item, const uint32_t* __restrict__ aoff,
                                                                  uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                                  const uint32_t* __restrict__ blk_pair, const uint32_t* __restrict__ pstart_local);
__global__ __launch_bounds__(256) void anchor_emit_expand_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                                 uint32_t n_pairs, uint32_t n_items,
                                                                 const uint2* __restrict__ item, const uint32_t* __restrict__ aoff,
                                                                 uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                                 const uint32_t* __restrict__ blk_pair);
__global__ __launch_bounds__(256) void anchor_emit_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase,
                                                          uint32_t n_pairs, uint32_t n_items,
                                                          const uint2* __restrict__ lbcnt,
                                                          const uint32_t* __restrict__ aoff,
                                                          uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                          const uint32_t* __restrict__ blk_pair);
__global__ __launch_bounds__(EP_T) __attribute__((amdgpu_waves_per_eu(5, 8))) void anchor_emit_pairs_kernel(const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase, uint32_t n_pairs,
                                                                const uint2* __restrict__ item, const unsigned long long* __restrict__ poff,
                                                                uint4* __restrict__ anc, uint32_t cap, uint32_t* __restrict__ err,
                                                                const uint32_t* __restrict__ cbase, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks);
__global__ __launch_bounds__(256) void pair_start64_kernel(const unsigned long long* __restrict__ poff, uint32_t n_pairs, uint32_t* __restrict__ pstart, uint32_t cap,
                                                           const uint32_t* __restrict__ need_wide);
__global__ __launch_bounds__(256) void pair_guard_kernel(const uint32_t* __restrict__ need_wide, const unsigned long long* __restrict__ total64, unsigned long long cap,
                                                         uint32_t* __restrict__ pstart, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_start_kernel(const uint32_t* __restrict__ aoff, const uint32_t* __restrict__ sbase, uint32_t n_pairs, uint32_t* __restrict__ pstart, uint32_t cap,
                                                         const unsigned long long* __restrict__ bsum, uint32_t n_sum, unsigned long long* __restrict__ total64,
                                                         const uint32_t* __restrict__ need_wide);
__global__ __launch_bounds__(64) void chunk_heads_kernel(const uint32_t* __restrict__ pstart, const uint4* __restrict__ anc,
                                                         const uint32_t* __restrict__ cbase, uint32_t n_pairs, uint2* __restrict__ chunks,
                                                         uint32_t* __restrict__ n_chunks, uint32_t* __restrict__ err);
template <int COARSE>
__global__ __launch_bounds__(256) void anchor_next_kernel(const uint4* __restrict__ anc,
                                                          const uint32_t* __restrict__ pstart, uint32_t n_pairs,
                                                          const uint32_t* __restrict__ coarse, uint32_t* __restrict__ nxt);
__global__ __launch_bounds__(64) void chunk_hops_kernel(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ nxt, const uint32_t* __restrict__ cbase,
                                                         uint32_t n_pairs, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks,
                                                         uint32_t* __restrict__ err);
__global__ __launch_bounds__(64) void chunk_hops_sliced_kernel(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ nxt, const uint4* __restrict__ anc,
                                                                const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ cbase, uint32_t n_pairs,
                                                                uint32_t* __restrict__ slice_cnt, int pass, uint2* __restrict__ scratch, uint2* __restrict__ chunks, uint32_t* __restrict__ n_chunks,
                                                                uint32_t* __restrict__ err);
__global__ __launch_bounds__(64) void chunk_hops_items_kernel(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ aoff,
                                                               const PairDesc* __restrict__ pairs, const uint32_t* __restrict__ sbase, const uint32_t* __restrict__ cbase,
                                                               uint32_t n_pairs, uint32_t* __restrict__ slice_cnt, uint2* __restrict__ scratch, uint32_t* __restrict__ err);
__global__ __launch_bounds__(64 * LANE_WAVES) __attribute__((amdgpu_waves_per_eu(3, 8))) void chain_lane20_kernel(ChainArgs A, uint32_t rows_per_wave);
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_lane20x_kernel(ChainArgs A, uint32_t rows_per_wave);
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_lane_kernel(ChainArgs A, uint32_t rows_per_wave);
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_quad_kernel(ChainArgs A);
__global__ __launch_bounds__(64 * LANE_WAVES) void chain_quad_deep_kernel(ChainArgs A);
__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_chunk_kernel(ChainArgs A);
__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_chunk_list_kernel(ChainArgs A);
__global__ __launch_bounds__(256) void row_len_kernel(const uint2* __restrict__ chunks, const uint32_t* __restrict__ n_chunks, const uint32_t* __restrict__ cbase,
                                                      const uint32_t* __restrict__ row_pair, uint32_t n_rows, uint32_t* __restrict__ key, uint32_t* __restrict__ val);
template <int S>
__global__ __launch_bounds__(64 * CHAIN_WAVES) void chain_wave_reg_kernel(ChainArgs A);
__global__ __launch_bounds__(256) void select_tiny_kernel(SelArgs S);
__global__ __launch_bounds__(64) void select_kernel(SelArgs S, uint32_t* __restrict__ mid_list, uint32_t* __restrict__ mid_count);
__global__ __launch_bounds__(64) void select_mid_kernel(SelArgs S, const uint32_t* __restrict__ mid_list, const uint32_t* __restrict__ mid_count);
__global__ __launch_bounds__(BIG_T) void select_big_kernel(BigArgs B);
__global__ __launch_bounds__(BIG_T) void select_huge_kernel(BigArgs B);
__global__ __launch_bounds__(256) void chunk_seeds_kernel(ChainArgs A);
__global__ __launch_bounds__(256) void pair_empty_kernel(ReduceArgs R, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_reduce_small_kernel(ReduceArgs R, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_reduce_tiny_kernel(ReduceArgs R, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_reduce_wave_kernel(ReduceArgs R, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_reduce_kernel(ReduceArgs R, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_reduce_large_kernel(ReduceArgs R, uint32_t n_pairs);
__global__ __launch_bounds__(256) void pair_ref_keys_kernel(const uint2* __restrict__ pair_qr, uint32_t n_pairs, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids);
template <bool EMIT>
__global__ __launch_bounds__(64) void gsi_join_kernel(GsiJoinArgs A);
