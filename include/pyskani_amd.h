/* include/pyskani_amd.h — C-ABI of the MI355X-native ANI engine (libpyskani_amd.so).
 *
 * The drop-in boundary for pyskani's Database.sketch()/Database.query() hot path. pyskani
 * has no FFI today: its PyO3 crate calls the Rust crate `skani` directly. Each entry point
 * below replaces one of those call sites (paths are relative to /root/reference):
 *
 *   psk_sketch_host / psk_sketch_batch_device
 *        <- the per-contig loop over skani::seeding::fmh_seeds in Database::_sketch,
 *           src/pyskani/_skani/lib.rs:140-185 (contig filter :156, metadata :157-161,
 *           fmh_seeds :165-171) and skani::types::Sketch::get_markers_only at :495
 *   psk_db_add  <- self.markers.push(marker) + sketches.store(sketch)   lib.rs:501-508
 *   psk_screen  <- skani::screen::check_markers_quickly loop            lib.rs:617-637
 *   psk_chain   <- skani::chain::map_params_from_sketch + chain_seeds   lib.rs:646-653
 *   psk_query   <- the whole allow_threads closure of Database::query   lib.rs:569-659
 *                  (screen_val default :603-609, learned rule :611-614, ani>0.1 filter :654)
 *   psk_hit     <- the fields of skani::types::AniEstResult that Hit exposes, hit.rs:77-104
 *   psk_model_* <- skani::regression::{use_learned_ani, get_model}      lib.rs:611-614 and the
 *                  `&model_opt` argument of map_params_from_sketch      lib.rs:646-651
 *   psk_db_add  also implements the name-keyed store of lib.rs:51-55: queries shortlist NAMES
 *                  (lib.rs:616-637), so a name sketched twice yields one hit, against its last sketch
 *
 * Conventions: plain pointers and sizes, opaque handles, no exceptions cross the ABI.
 * Every function returns a psk_status; psk_last_error() gives the thread-local message.
 * Inputs are borrowed for the duration of the call only. Outputs returned through `**`
 * are library-allocated and released with the matching psk_*_free / psk_free.
 * Threading: every call runs on one of the context's execution lanes (own HIP stream, scratch and pinned
 * staging; at most $PSK_LANES = 8 at a time, further callers wait), so psk_query / psk_screen / psk_chain /
 * psk_sketch_host from different host threads overlap on the device, as `&self` + the released GIL allow in
 * lib.rs:551,569. A database is held shared by queries and exclusively by psk_db_add (`&mut self`, lib.rs:479)
 * and by whatever (re)builds its device tables.
 * A missing GPU is an error (PSK_EHIP) — there is no CPU fallback in this library.
 */
#ifndef PYSKANI_AMD_H
#define PYSKANI_AMD_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PSK_OK = 0,
    PSK_EINVAL = 1,   /* bad argument (k > 16, c == 0, NULL, ...)  -> ValueError          */
    PSK_ENOMEM = 2,   /* host or device allocation failed          -> MemoryError         */
    PSK_EHIP = 3,     /* HIP runtime error / no device             -> RuntimeError        */
    PSK_ENOMODEL = 4, /* learned-ANI requested, no model loaded    -> RuntimeError        */
    PSK_EKEY = 5,     /* unknown reference name                    -> KeyError            */
    PSK_ELIMIT = 6,   /* input exceeds a documented limit          -> OverflowError       */
    PSK_ERCCL = 7     /* RCCL missing or a collective failed       -> RuntimeError        */
} psk_status;

typedef struct psk_ctx psk_ctx;       /* one GPU: device id, stream, scratch arenas         */
typedef struct psk_sketch psk_sketch; /* one genome's seeds+index+markers, resident in HBM  */
typedef struct psk_db psk_db;         /* ordered reference set (insertion order = lib.rs:501) */
typedef struct psk_model psk_model;   /* learned-ANI regression model (gradient-boosted trees), device resident */

/* SketchParams::new(marker_c, c, k, false, false)  lib.rs:416 */
typedef struct {
    int32_t c;        /* compression, default 125        */
    int32_t marker_c; /* marker compression, default 1000 */
    int32_t k;        /* k-mer size, default 15, <= 16    */
} psk_params;

/* kwargs of Database.query, lib.rs:549, and the CommandParams literal lib.rs:573-601 */
typedef struct {
    int32_t learned_ani;     /* -1 = default rule (lib.rs:611-613), 0 = off, 1 = on            */
    int32_t median;          /* lib.rs:583 */
    int32_t robust;          /* lib.rs:582 */
    int32_t faster_small;    /* rescue_small = !faster_small, lib.rs:597 */
    double cutoff;           /* 0 = SEARCH_ANI_CUTOFF_DEFAULT (0.80), lib.rs:603-609 */
    double min_aligned_frac; /* 0 = D_FRAC_COVER_CUTOFF/100 = 0.15, lib.rs:589-590   */
    const psk_model* model;  /* `model_opt` of lib.rs:614/:650. The regression runs when learned_ani == 1, or when
                                learned_ani == -1, c >= 70, !median (lib.rs:611-613) AND a model is given; learned_ani == 1
                                without a model is PSK_ENOMODEL, learned_ani == -1 without one returns the raw chain ANI */
} psk_query_opts;

typedef struct {
    float ani;        /* AniEstResult.ani                   hit.rs:78  */
    float af_query;   /* AniEstResult.align_fraction_query  hit.rs:90  */
    float af_ref;     /* AniEstResult.align_fraction_ref    hit.rs:102 */
    uint32_t ref_index; /* insertion index in the db; name via psk_db_name */
    /* integer intermediates (bit-exact parity checks against the oracle) */
    uint32_t n_chunks, n_intervals;
    uint64_t n_anchors, covered_query, covered_ref, sum_chain_anchors, sum_chunk_seeds;
    float ani_raw;      /* chain ANI before the regression (== ani when learned == 0)      */
    float ani_std;      /* sample standard deviation of the per-chunk ANI estimates        */
    uint32_t learned;   /* 1 iff the regression model produced `ani`                       */
    uint32_t reserved;
} psk_hit;

/* What the reference's Hit holds (hit.rs:77-104: identity, query_fraction, reference_fraction, the two names) and nothing else, in 20 bytes: the record of
 * psk_query_many_min and psk_gather_hits_min - what crosses PCIe and xGMI when the caller did not ask for the chaining integers. */
typedef struct {
    float ani, af_query, af_ref;
    uint32_t ref_index; /* insertion index in the db (sharded runs: the global index) */
    uint32_t query;     /* index of the query within the call (sharded runs: the global index); bit 31: the regression model produced `ani` */
} psk_hit_min;

typedef struct { uint32_t kmer, pos, contig, canon; } psk_seed; /* export record (parity tests) */

const char* psk_last_error(void);
const char* psk_version(void);
/* The C-ABI's revision: raised whenever an entry point's parameters or a structure's layout change (4: psk_sketch_unpack takes the extent of its source buffer as third
 * argument; 5: psk_hit_min / psk_query_many_min / psk_gather_hits_min / psk_ctx_join_work added). A binding compares psk_abi_version() with the PSK_ABI_VERSION it was
 * written against before it calls anything else: an argument list that moved is a memory error, not a link error. */
#define PSK_ABI_VERSION 5
int psk_abi_version(void);
/* Releases an array the library returned (hit lists, gathered lists). Never release such an array with free(): large hit arrays are
 * huge-page blocks the library keeps one of for its next call ($PSK_HIT_CACHE=0: returned to the system at once). */
void psk_free(void* p);

psk_status psk_ctx_create(int device, psk_ctx** out);
void psk_ctx_destroy(psk_ctx* ctx);
psk_status psk_ctx_synchronize(psk_ctx* ctx);
/* Measurement: bracket the named kernels' launches with HIP events on the ctx stream.
 * kernel in {"sketch_scan","sketch_emit","sketch_sort","screen","anchor","chain_chunk",
 * "select","pair_reduce","anchor_emit"}; "reset" clears the accumulators. psk_ctx_timing synchronises the stream. */
psk_status psk_ctx_set_timing(psk_ctx* ctx, int on);
/* Shader clock the device holds under an integer-VALU load: a fixed ~1 ms micro-kernel (v_alignbit_b32 chains, 4 cycles per
 * wave64 instruction on gfx950) timed with HIP events; *mhz = cycles / duration, *ms = the duration (may be NULL). */
psk_status psk_ctx_clock_probe(psk_ctx* ctx, double* mhz, double* ms);
psk_status psk_ctx_timing(psk_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);
/* Work of the chain stage since the last reset: pairs that reached chain_seeds (lib.rs:652-653), (pair, query seed) items joined,
 * anchors emitted. The per-kernel algorithmic bytes of bench.py are counted from these. Any pointer may be NULL. */
psk_status psk_ctx_work(psk_ctx* ctx, uint64_t* pairs, uint64_t* items, uint64_t* anchors, int reset);
/* ... and of the joins that go through the database-wide seed index (metagenome contigs, all-vs-all of genomes): query seeds looked up per walk of the index and index
 * entries in the runs the lookups found (what those kernels' algorithmic bytes are counted from: 12 B per lookup, 12 B per entry, 16 B per anchor); while the timers
 * are on (psk_ctx_set_timing) also the candidate chains the selection read and the chunk-table rows the reduce read. Any pointer may be NULL. */
psk_status psk_ctx_join_work(psk_ctx* ctx, uint64_t* lookups, uint64_t* visited, uint64_t* candidates, uint64_t* rows, int reset);
/* Measurement: psk_query_host calls since the context was created that ran as one launch sequence (`taken`), that exceeded one of its
 * capacities and were rerun on the general path (`rerun`), and that went to the general path at once (`general`, which includes `rerun`). */
psk_status psk_ctx_small_query_stats(psk_ctx* ctx, uint64_t* taken, uint64_t* rerun, uint64_t* general);
/* Host-side 2-bit packing of the ingest pipeline (csrc/pack_host.cpp), exposed for tests: n ASCII bases -> ceil(n / 16) words, the first base in a
 * word's highest two bits; A 0, C 1, G 2, T 3, case-insensitive, every other byte 0 (the codes of the sketch kernels). mode 0: the best
 * implementation the CPU has (AVX-512BW), 1: the scalar one. */
void psk_pack2bit_host(const uint8_t* src, uint64_t n, uint32_t* dst, int mode);
/* device bump allocator for callers that stage genomes in HBM themselves (bench, multi-GPU) */
psk_status psk_device_alloc(psk_ctx* ctx, size_t bytes, void** dptr);
psk_status psk_device_free(psk_ctx* ctx, void* dptr);
psk_status psk_memcpy_h2d(psk_ctx* ctx, void* dst, const void* src, size_t bytes);

/* Sketch one genome from host ASCII contigs. Contigs shorter than 500 are ignored (lib.rs:156). */
psk_status psk_sketch_host(psk_ctx* ctx, const psk_params* p, const uint8_t* const* contigs,
                           const uint64_t* lens, uint32_t n_contigs, int want_seeds,
                           psk_sketch** out);

/* Sketch n_genomes genomes from host ASCII in one call: genome g owns contigs [genome_first_contig[g],
 * genome_first_contig[g+1]) of contigs[]/lens[] (ordinary pageable memory, borrowed for the call). The call is a
 * three-stage pipeline — worker threads copy into pinned staging slots, a copy stream moves the slots over PCIe, the
 * sketch kernels run on the previous sub-batch — so the sustained rate is the host->device link's, not one memcpy
 * thread's ($PSK_INGEST_THREADS, default min(16, cores / 2)). Same result as n_genomes x psk_sketch_host. */
psk_status psk_sketch_many_host(psk_ctx* ctx, const psk_params* p, const uint8_t* const* contigs, const uint64_t* lens,
                                const uint32_t* genome_first_contig, uint32_t n_genomes, int want_seeds,
                                psk_sketch** out);

/* Sketch n_genomes genomes whose ASCII already sits in HBM. d_bases is a device pointer;
 * contig i is bytes [contig_off[i], contig_off[i]+contig_len[i]) of it, contig_off[i] % 16 == 0
 * and the allocation must extend 16 bytes past the last contig. genome g owns contigs
 * [genome_first_contig[g], genome_first_contig[g+1]). out[] receives n_genomes handles. */
psk_status psk_sketch_batch_device(psk_ctx* ctx, const psk_params* p, const uint8_t* d_bases,
                                   const uint64_t* contig_off, const uint64_t* contig_len,
                                   const uint32_t* genome_first_contig, uint32_t n_genomes,
                                   int want_seeds, psk_sketch** out);

void psk_sketch_free(psk_sketch* s);
/* n x psk_sketch_free in one call (a host in an interpreted language pays per call: 100 000 contig sketches) */
void psk_sketch_free_many(psk_sketch* const* sketches, uint32_t n);
psk_status psk_sketch_info(const psk_sketch* s, psk_params* p, uint64_t* n_seeds,
                           uint64_t* n_markers, uint64_t* total_len, uint32_t* n_contigs);
/* copy the sketch back to the host: seeds in (contig,pos) order, markers sorted unique */
psk_status psk_sketch_export(const psk_sketch* s, psk_seed* seeds, uint64_t* markers);

/* kept-contig lengths (n_contigs from psk_sketch_info), needed to serialise a sketch */
psk_status psk_sketch_contig_lens(const psk_sketch* s, uint32_t* lens);
/* Inverse of psk_sketch_export (+ contig lengths): rebuilds a device-resident sketch, e.g. from the records
 * Database.open/load read back (lib.rs:95-122, 249-337). seeds in (contig,pos) order, markers sorted distinct.
 * has_seeds = 0 builds a marker-only sketch (skani::types::Sketch::get_markers_only, lib.rs:495). */
psk_status psk_sketch_import(psk_ctx* ctx, const psk_params* p, const uint32_t* contig_lens, uint32_t n_contigs,
                             const psk_seed* seeds, uint64_t n_seeds, const uint64_t* markers, uint64_t n_markers,
                             int has_seeds, psk_sketch** out);

/* ---- learned-ANI regression (lib.rs:611-614; skani::regression, crate gbdt 0.1.3 at Cargo.lock:1608) ----
 * skani embeds its trained GBDT weights inside the crate, which is not part of the reference tree; a model
 * therefore comes from the caller: either as flat arrays (what a Rust host holding a gbdt::GBDT would pass) or
 * as the serde-JSON text of a gbdt::gradient_boost::GBDT. Inference runs on the GPU, one lane per hit.
 * Tree semantics (gbdt 0.1.3 DecisionTree::predict_one): at an inner node go LEFT iff x[feature] < threshold;
 * a feature equal to PSK_FEATURE_UNKNOWN follows `missing` (-1 left, 0 stop and use this node's value, +1 right);
 * prediction = bias + shrinkage * sum over trees of the reached node's value, accumulated in tree order in f32. */
#define PSK_FEATURE_UNKNOWN (-3.402823466e+38F)   /* gbdt VALUE_TYPE_UNKNOWN = f32::MIN */
typedef struct {
    int32_t feature;    /* index into the model's feature vector */
    float threshold;    /* DTNode.feature_value */
    int32_t left, right;/* node indices relative to the tree's first node */
    float value;        /* DTNode.pred */
    int32_t missing;    /* -1 / 0 / +1 */
    int32_t is_leaf;
    int32_t reserved;
} psk_tree_node;
/* What a model's feature vector is made of. The default vector (features == NULL) is
 * {ANI100, STD100, Q90_QUERY, Q50_QUERY, Q10_QUERY, Q90_REF, Q50_REF, Q10_REF, AVG_CHAIN_LEN}: the fields skani's
 * AniEstResult carries beside the three fractions, as recalled — UNVERIFIED against the absent crate; a model file
 * may name its own order with a top-level "psk_features": ["ani100", "std100", ...] array. */
typedef enum {
    PSK_F_ANI100 = 0,       /* raw chain ANI x 100 */
    PSK_F_STD100 = 1,       /* sample std of the per-chunk ANI estimates x 100 */
    PSK_F_Q90_QUERY = 2, PSK_F_Q50_QUERY = 3, PSK_F_Q10_QUERY = 4,   /* contig-length quantiles: sorted lengths at n*9/10, n/2, n/10 */
    PSK_F_Q90_REF = 5, PSK_F_Q50_REF = 6, PSK_F_Q10_REF = 7,
    PSK_F_AVG_CHAIN_LEN = 8,/* covered bases / kept chains */
    PSK_F_AF_QUERY = 9, PSK_F_AF_REF = 10, PSK_F_N_CHUNKS = 11,
    PSK_F_TOTAL_LEN_QUERY = 12, PSK_F_TOTAL_LEN_REF = 13, PSK_F_N_CONTIGS_QUERY = 14, PSK_F_N_CONTIGS_REF = 15,
    PSK_F_COUNT = 16
} psk_feature;
psk_status psk_model_create(psk_ctx* ctx, const psk_tree_node* nodes, uint64_t n_nodes, const uint32_t* tree_first_node,
                            uint32_t n_trees, float bias, float shrinkage, const int32_t* features, uint32_t n_features,
                            psk_model** out);
psk_status psk_model_load_json(psk_ctx* ctx, const char* json, size_t len, psk_model** out);
psk_status psk_model_load_file(psk_ctx* ctx, const char* path, psk_model** out);
void psk_model_free(psk_model* m);
psk_status psk_model_info(const psk_model* m, uint32_t* n_trees, uint64_t* n_nodes, uint32_t* n_features);
/* evaluate the model on n_rows host rows of n_features floats (GPU evaluation; parity tests and model checks) */
psk_status psk_model_predict(const psk_model* m, const float* rows, uint32_t n_rows, float* out);

/* ---- device-side sketch records: the exchange step of a multi-GPU all-vs-all. A packed sketch is one
 * self-contained, 16-byte aligned byte range of HBM that an RCCL all-gather can move between GPUs as it is;
 * psk_sketch_unpack turns n received records (d_src + offsets[i]) into n sketches sharing one store. Nothing
 * passes through host memory except the 64-byte headers and the contig tables. */
psk_status psk_sketch_pack_size(const psk_sketch* s, uint64_t* bytes);
psk_status psk_sketch_pack(const psk_sketch* s, void* d_dst, uint64_t capacity);
psk_status psk_sketch_unpack(psk_ctx* ctx, const void* d_src, uint64_t capacity, const uint64_t* offsets, uint32_t n, psk_sketch** out);
/* n sketches (of one context) -> n records at d_dst + offsets[i] (16-byte aligned, inside `capacity` bytes): one header upload,
 * one copy launch and one synchronisation for the whole batch */
psk_status psk_sketch_pack_many(const psk_sketch* const* sketches, uint32_t n, void* d_dst, const uint64_t* offsets, uint64_t capacity);

/* ---- multi-GPU exchange. One process per GPU; a psk_comm is this rank's end of an RCCL communicator bound to its
 * context ("one ani_ctx owns its devices and its RCCL communicator", §8b threading row). The path shards by reference
 * (lib.rs:617-657: every (query, ref) pair is independent), so the only collectives are the all-gather of the per-shard hit lists and,
 * for an all-vs-all, the all-gather of the shards' sketches as the query side. librccl.so is loaded at run time: without it every
 * call below returns PSK_ERCCL and the library is otherwise complete (replicas only).
 * Bootstrap: rank 0 calls psk_comm_unique_id, the host program hands the PSK_COMM_ID_BYTES bytes to every rank over any channel
 * it has (MPI, a file, torch.distributed), every rank calls psk_comm_create (collective: returns when all ranks have joined). */
typedef struct psk_comm psk_comm;
#define PSK_COMM_ID_BYTES 128
psk_status psk_comm_unique_id(void* id);
psk_status psk_comm_create(psk_ctx* ctx, int rank, int world, const void* id, psk_comm** out);
void psk_comm_destroy(psk_comm* comm);
/* rank, world, bytes this rank has sent through the communicator and the number of collectives (any may be NULL) */
psk_status psk_comm_info(const psk_comm* comm, int* rank, int* world, uint64_t* bytes_sent, uint64_t* collectives);
/* All-gather of ragged per-shard hit lists (host arrays). Records travel as they are: the caller has put the GLOBAL reference
 * index in ref_index and the global query index in `reserved`. *all (psk_free) = every rank's list in rank order, identical on
 * every rank; counts (world entries, may be NULL) = the ranks' list lengths. */
psk_status psk_gather_hits(psk_comm* comm, const psk_hit* local, uint64_t n_local, psk_hit** all, uint64_t* n_all, uint64_t* counts);
/* ... the same for 20-byte records (the caller has put the global indices in ref_index / query): a quarter of the bytes per link */
psk_status psk_gather_hits_min(psk_comm* comm, const psk_hit_min* local, uint64_t n_local, psk_hit_min** all, uint64_t* n_all, uint64_t* counts);
/* All-gather of device-resident sketches as packed records, HBM -> xGMI -> HBM (no host hop): every rank contributes n sketches
 * and receives everybody's as sketches on ITS GPU. *all (psk_free; each entry psk_sketch_free) = the ranks' sketches in rank
 * order; counts[r] (world entries) = how many came from rank r. */
psk_status psk_gather_sketches(psk_comm* comm, const psk_sketch* const* mine, uint32_t n, psk_sketch*** all, uint32_t* counts);

psk_status psk_db_create(psk_ctx* ctx, const psk_params* p, psk_db** out);
void psk_db_destroy(psk_db* db);
/* takes ownership of s (also on failure) */
psk_status psk_db_add(psk_db* db, const char* name, psk_sketch* s);
/* n x psk_db_add in one call; ownership moves only on PSK_OK */
psk_status psk_db_add_batch(psk_db* db, const char* const* names, psk_sketch* const* sketches, uint32_t n);
uint32_t psk_db_size(const psk_db* db);
const char* psk_db_name(const psk_db* db, uint32_t index);
const psk_sketch* psk_db_sketch(const psk_db* db, uint32_t index);

/* check_markers_quickly(query, ref_i, screen_val, rescue_small) for every ref of the db.
 * pass[i] in {0,1}; shared[i] = |markers(q) ∩ markers(ref_i)| (may be NULL). */
psk_status psk_screen(psk_db* db, const psk_sketch* query, double screen_val, int rescue_small,
                      uint8_t* pass, uint32_t* shared);

/* chain_seeds(ref, query, map_params) for n_refs references against one query. */
psk_status psk_chain(psk_ctx* ctx, const psk_sketch* const* refs, uint32_t n_refs,
                     const psk_sketch* query, const psk_query_opts* o, psk_hit* out);

/* Database.query: screen, chain the shortlist, keep ani > 0.1. hits in ref insertion order. */
psk_status psk_query(psk_db* db, const psk_sketch* query, const psk_query_opts* o,
                     psk_hit** hits, uint64_t* n_hits);

/* Database.query from the caller's host bytes, as the reference's pymethod runs it (lib.rs:549-660: `_sketch` of the query at
 * :571 - not stored - then the screen / chain closure): contigs[i] is lens[i] ASCII bytes, borrowed for the call; contigs shorter
 * than 500 are ignored (lib.rs:156); seed = the `seed` kwarg (lib.rs:553). Same hits as psk_sketch_host + psk_query + psk_sketch_free.
 * A small genome (one or a few contigs, up to ~90 kb at c = 30 / ~380 kb at c = 125) runs as ONE launch sequence with ONE
 * synchronisation (csrc/small_query.hip); $PSK_SMALL_QUERY=0 keeps every call on the general path (tests, A/B). */
psk_status psk_query_host(psk_db* db, const uint8_t* const* contigs, const uint64_t* lens, uint32_t n_contigs, int seed,
                          const psk_query_opts* o, psk_hit** hits, uint64_t* n_hits);

/* n_queries x Database.query against one database (all-vs-all and metagenome-bin workloads, BASELINE
 * configs[2]/[3]); offsets has n_queries+1 entries, hits of query i are hits[offsets[i]..offsets[i+1]). */
psk_status psk_query_many(psk_db* db, const psk_sketch* const* queries, uint32_t n_queries,
                          const psk_query_opts* o, psk_hit** hits, uint64_t* offsets);
/* psk_query_many returning psk_hit_min records: same hits, same order; `query` of a record = the index of its query in `queries`. The chaining integers of
 * psk_hit stay on the device (what a parity test reads through psk_query_many); a metagenome step's 9.5 M hits are 190 MB instead of 763 MB over PCIe. */
psk_status psk_query_many_min(psk_db* db, const psk_sketch* const* queries, uint32_t n_queries,
                              const psk_query_opts* opts, psk_hit_min** hits, uint64_t* offsets);

#ifdef __cplusplus
}
#endif
#endif
