// What the translation units of the query path (screen.hip, chain.hip, query_many.hip, seed_index.hip) share on the HOST side. The chain stage's device-side
// structures and kernels are in chain_stages.h.
#pragma once
#include "chain_stages.h"
#include <deque>
#include <vector>
#include <cstring>

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- the test / A-B switches of the query path ($PSK_*; profiles/r6/paths.md maps each to the test that covers it). A call (psk_query*, psk_chain*, psk_screen) reads
// them ONCE, when it begins, into this struct and hands it down: every decision of the call sees one consistent set, a test (or bench.py: PSK_PIPELINE) may flip them
// between two calls of one process, and no kernel-launching code touches the environment. A value is kept as the text the environment held (get(): nullptr = not set).
struct SwitchVal {
    char text[24]; bool set;
    const char* get() const { return set ? text : nullptr; }
};
#define PSK_SWITCHES(X) \
    X(chain_serial, "PSK_CHAIN_SERIAL") \
    X(join_pairs, "PSK_JOIN_PAIRS") \
    X(gsl_stage, "PSK_GSL_STAGE") \
    X(probe_local, "PSK_PROBE_LOCAL") \
    X(emit_pairs, "PSK_EMIT_PAIRS") \
    X(xcd_group, "PSK_XCD_GROUP") \
    X(dp_prune, "PSK_DP_PRUNE") \
    X(chunk_hops, "PSK_CHUNK_HOPS") \
    X(emit_heads, "PSK_EMIT_HEADS") \
    X(hops_items, "PSK_HOPS_ITEMS") \
    X(hops_unsliced, "PSK_HOPS_UNSLICED") \
    X(gsi_stage, "PSK_GSI_STAGE") \
    X(emit_expand, "PSK_EMIT_EXPAND") \
    X(row_sort, "PSK_ROW_SORT") \
    X(chain_lane, "PSK_CHAIN_LANE") \
    X(lane_xtrees, "PSK_LANE_XTREES") \
    X(chain_quad_deep, "PSK_CHAIN_QUAD_DEEP") \
    X(chain_wave_reg, "PSK_CHAIN_WAVE_REG") \
    X(select_tiny, "PSK_SELECT_TINY") \
    X(big_solo, "PSK_BIG_SOLO") \
    X(huge_slots, "PSK_HUGE_SLOTS") \
    X(huge_min_seeds, "PSK_HUGE_MIN_SEEDS") \
    X(reduce_small, "PSK_REDUCE_SMALL") \
    X(reduce_wave, "PSK_REDUCE_WAVE") \
    X(reduce_tiny, "PSK_REDUCE_TINY") \
    X(join, "PSK_JOIN") \
    X(round_queries, "PSK_ROUND_QUERIES") \
    X(screen, "PSK_SCREEN") \
    X(prefilter, "PSK_PREFILTER") \
    X(gsi_join, "PSK_GSI_JOIN") \
    X(gsl_max_blocks, "PSK_GSL_MAX_BLOCKS") \
    X(probe, "PSK_PROBE") \
    X(gsi_slice, "PSK_GSI_SLICE") \
    X(bsi_small, "PSK_BSI_SMALL") \
    X(batch_items_log2, "PSK_BATCH_ITEMS_LOG2") \
    X(batch_pairs_log2, "PSK_BATCH_PAIRS_LOG2") \
    X(gsi_onepass, "PSK_GSI_ONEPASS") \
    X(pipeline, "PSK_PIPELINE") \
    X(screen_global, "PSK_SCREEN_GLOBAL") \
    X(screen_wave, "PSK_SCREEN_WAVE")
struct Switches {
#define X(field, name) SwitchVal field;
    PSK_SWITCHES(X)
#undef X
    static Switches read();                  // (chain.hip)
    bool join_wide() const { const char* e = join.get(); return e && !strcmp(e, "wide"); }      // $PSK_JOIN=wide: every join in the wide item format (the fallback of counts the packed format cannot hold)
};

// ---- screen.hip: the marker screen (lib.rs:617-637)
// Screens nq queries against every reference of the db; the pass matrix [nq][n_refs] STAYS ON THE DEVICE (d_pass, caller-owned).
// `keep` holds the host staging of the async uploads until the caller's next stream synchronisation.
struct ScreenStaging { std::deque<std::vector<MarkerSet>> hq; std::deque<std::vector<uint32_t>> qoff; };
psk_status upload_marker_table(Lane* ctx, psk_db* db);
psk_status build_inverted(Lane* ctx, psk_db* db);
psk_status screen_many_device(Lane* ctx, psk_db* db, const psk_sketch* const* queries, uint32_t nq, double screen_val, int rescue_small, uint8_t* d_pass, ScreenStaging& keep, const Switches& sw);

// ---- chain.hip: one launch sequence over a batch of pairs (lib.rs:640-657)
SketchDesc make_desc(const psk_sketch* s, bool unindexed_ok = false);
__global__ __launch_bounds__(256) void pair_build_rows_kernel(const BatchQ* __restrict__ bq, const uint8_t* __restrict__ pass, uint32_t n_refs,
                                                              const SketchDesc* __restrict__ qd, const SketchDesc* __restrict__ rd,
                                                              PairDesc* __restrict__ pairs, uint32_t* __restrict__ sbase, uint32_t* __restrict__ cbase,
                                                              uint2* __restrict__ pair_qr, uint32_t n_pairs, uint32_t n_items, uint32_t n_rows);
__global__ __launch_bounds__(256) void pass_canon_kernel(uint8_t* __restrict__ pass, uint32_t n_refs, const uint32_t* __restrict__ canon);
__global__ __launch_bounds__(256) void pass_count_kernel(const uint8_t* __restrict__ pass, uint32_t n_refs, uint32_t* __restrict__ row_count, uint8_t* __restrict__ col_flag, uint32_t* __restrict__ row_blocks);
constexpr size_t CHAIN_ANCHOR_WORDS = 14;      // u32 per anchor in Lane::q_d: the 16-byte record, the successor array, the serial DP's back-pointers, the 32-byte candidate record
// device arrays of one chain launch sequence, carved from ctx->q_b
struct ChainBufs {
    PairDesc* pairs; uint32_t *sbase, *cbase, *pstart; uint2* lbcnt; uint32_t* aoff; uint32_t* nch; uint2* chunks; ChunkOut* cout;
    psk_hit* hits; psk_hit* hits_sel; uint32_t* misc; uint32_t* ovf; unsigned long long* bsum; uint2* pair_qr; BatchQ* bq;
    uint32_t *blk_pair, *row_pair, *live, *big_list, *huge_list;
    uint32_t gi, gi_sum;      // 256-item tiles; entries of bsum
    unsigned long long* total;      // the 64-bit anchor total: misc[16..17], so that status, total and the hits behind them cross in one copy
    uint32_t rows_pair_max = 0xFFFFFFFFu;      // most chunk-table rows any pair of the batch can have (the host knows its queries): which reduce kernels have work
    // join through the database-wide seed index (gsi_join_kernel): the index, the pass matrix the pairs came from and the batch's entries; g_key null: not available
    const uint32_t* g_key = nullptr; const unsigned long long* g_val = nullptr; const uint32_t* g_bucket = nullptr; int g_shift = 0;
    uint32_t g_nb1 = 0, g_blocks = 0; const unsigned long long* g_base = nullptr;      // (the slice join's index comes in blocks of references: psk_db::bsi_*; g_base: the blocks' first entries)
    const uint32_t* b_key = nullptr; const unsigned long long* b_val = nullptr; const uint32_t* b_bucket = nullptr; int b_shift = 0; uint32_t b_nb1 = 0, b_blocks = 0, b_max = 0; const unsigned long long* b_base = nullptr;      // the contig join: the blocked index beside the database-wide one
    const uint8_t* d_pass = nullptr; uint32_t n_refs = 0, n_bq = 0, p_cap = 0;
    // mid-sized pairs (all-vs-all of genomes): the index join by (query, slice) waves (slice_join.hip); the batch's wave table, record offsets and per-record arrays
    bool gsi_slice = false; const uint2* gsl_tab = nullptr; uint32_t gsl_n_tab = 0; const uint2* gsl_ebase = nullptr; uint32_t *gsl_cnt = nullptr, *gsl_bm = nullptr, *gsl_un = nullptr, gsl_n_slices = 0; uint4* gsl_rec = nullptr;
    bool gsi_onepass = false;      // the index join without its COUNT pass (GsiJoinArgs::onepass): asked for by the caller, which reruns the batch without it when err bit 2 comes back
};
psk_status chain_layout(Lane* ctx, size_t n_pairs, size_t n_items, size_t n_rows, size_t n_bq, ChainBufs* L);
psk_status chain_run(Lane* ctx, const ChainBufs& L, uint32_t n_pairs, size_t n_items, size_t n_rows, const psk_params& prm,
                     const psk_query_opts* o, const SketchDesc* d_qd, const SketchDesc* d_rd, uint64_t cap, bool wide, const Switches& sw, bool probe_ok = false);
uint64_t anchor_cap_for(Lane* ctx, size_t n_items, bool sparse = false, bool gb_scale = false);
// outcome of a launch sequence, read back with the hits
struct ChainTail { uint32_t misc[16]; unsigned long long total64, visited, cands, rows; };      // (misc[16..23]: the anchor total; index entries the join visited; with the timers on, candidates and live chunk rows)
psk_status chain_check(const ChainTail& T, uint32_t n_pairs, uint64_t* cap, bool* wide, bool* retry);
struct HostPair { const psk_sketch* r; const psk_sketch* q; };
psk_status chain_batch(Lane* ctx, const HostPair* hp, uint32_t n_pairs, const psk_query_opts* o, psk_hit* out, const Switches& sw);

// ---- query_many.hip
uint64_t index_stamp(const psk_db* db);
psk_status refresh_ref_descs(Lane* ctx, psk_db* db);

// ---- seed_index.hip: called with the database locked exclusively; leave gsi_state / bsi_state 1 (built) or 2 (this database cannot have one)
psk_status build_gsi(Lane* ctx, psk_db* db);
psk_status build_bsi(Lane* ctx, psk_db* db);
