// The join of batches of MID-SIZED pairs (all-vs-all of ~5 Mb genomes) through the database-wide seed index, one wave per
// (query, slice of GSL_SEEDS query seeds): slice_join.hip. Shared with query_many.hip / chain.hip, which plan the batches and launches the rest
// of the chain stage (lib.rs:640-657 is the loop this replaces: chain_seeds of the query against every shortlisted reference).
#pragma once
#include "common.h"

// one entry of a batch: query q against its passing references of rank [rank_lo, rank_hi); the entry's pairs, (pair, query seed)
// items and chunk-table rows start at pair_off / item_off / row_off (chain.hip: pair_build_rows_kernel)
struct BatchQ { uint32_t q, rank_lo, rank_hi, pair_off, item_off, row_off; };
constexpr uint32_t GSI_PMAX = 256;      // most pairs of one entry: their cursors (and staged anchor lines) sit in one wave's LDS

#ifndef GSL_SEEDS_N
#define GSL_SEEDS_N 256
#endif
constexpr uint32_t GSL_SEEDS = GSL_SEEDS_N;    // query seeds per slice = per wave (~64 kb of a genome at c = 125: about three chunks)
constexpr uint32_t GSL_WORDS = GSL_SEEDS / 32;
// One row of an entry's block table: the block's 256 references of the query's row of the pass matrix as eight 32-bit words, the block, the query's passing references
// before it, the block's first index entry (64 bits). An entry holds at most GSI_PMAX pairs, so at most GSI_PMAX blocks hold one of their references.
constexpr uint32_t GSL_BT_WORDS = 12;
struct GslArgs {
    const BatchQ* bq; uint32_t n_entries;
    const uint2* tab; uint32_t n_tab;       // wave -> (entry, slice)
    const uint2* ebase;                     // entry -> (x: its first (pair, slice) record, y: its first slice); record of (entry e, slice s, pair j) = ebase[e].x + s * pairs(e) + j
    uint32_t* un; uint32_t n_slices;        // per (entry, slice) GSL_WORDS words: the seeds that head a chunk of some pair of the entry (heads kernel; zeroed by its launcher)
    const uint8_t* pass; uint32_t n_refs; const SketchDesc* qd;
    const uint32_t* g_key; const unsigned long long* g_val; const uint32_t* g_bucket; int g_shift;      // psk_db::bsi_*: the seed index in blocks of 2^BSI_BLOG references
    const unsigned long long* g_base;       // first entry of every block (the bucket tables hold offsets within their block)
    uint32_t g_nb1, g_blocks;               // bucket-table entries per block; blocks
    uint32_t* blk_tab; uint32_t* blk_cnt; uint32_t blk_cap;      // per entry: the index blocks that hold one of its pairs' references (gsl_blocks_kernel): blk_cnt[e] rows of GSL_BT_WORDS words at blk_tab[e * blk_cap * GSL_BT_WORDS]
    uint32_t* cnt;       // per record: anchors of the (pair, slice) (count walk)
    uint4* rec;          // per record, from the heads kernel: {first anchor of the (pair, slice), chunk-table rows of the pair before the slice, lim1 lo, lim1 hi} -
                         // lim1 = (key of the chunk head open at the slice's start) + 1 + FRAGMENT_LENGTH, 0 before the pair's first anchor (key = q contig << 32 | q pos)
    uint32_t* bm;        // per record GSL_WORDS words: the slice's seeds that have an anchor with the pair (count walk)
    uint32_t* pair_cnt;  // per pair: anchors (count walk, atomics over the slices)
    const uint32_t* pstart; uint4* anc; uint32_t cap; uint32_t* err;      // err = the launch sequence's status words (bit 0: chunk table overflow, bit 1: capacity; [5]: rerun wide; [16..17]: 64-bit anchor total)
    uint32_t p_cap;      // most pairs any entry of the batch holds, rounded up to 64
    uint2* chunks; uint32_t* n_chunks;
    int stage;           // emit walk: 1 = a pair's anchors leave as whole 64-byte lines staged in LDS, 0 = every anchor its own 16-byte store (A/B)
};

// wave table of a batch: (entry, slice) for every slice of every entry's query, a query's slices one after the other; ebase: per entry (first record, first slice)
void gsl_make_tab(const BatchQ* bq, size_t n_entries, const uint32_t* q_seeds /* per entry */, std::vector<uint2>& tab, std::vector<uint2>& ebase, uint64_t* n_records, uint64_t* n_slices);
// the block tables of a batch's entries (one wave per entry sweeps its query's row of the pass matrix once; the walks - slice_join.hip's and join.hip's gsi_join_kernel - read the rows)
psk_status gsl_blocks_launch(const BatchQ* bq, uint32_t n_entries, const uint8_t* pass, uint32_t n_refs, uint32_t g_blocks, const unsigned long long* g_base,
                             uint32_t* blk_tab, uint32_t* blk_cnt, uint32_t blk_cap, hipStream_t st);
psk_status gsl_count_launch(const GslArgs& A, hipStream_t st);
psk_status gsl_heads_launch(const GslArgs& A, hipStream_t st);
psk_status gsl_emit_launch(const GslArgs& A, hipStream_t st);
