/* oracle/skani_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C, single-threaded CPU restatement of the pyskani hot path
 *   Database.sketch()  (/root/reference/src/pyskani/_skani/lib.rs:140-185, 477-510)
 *   Database.query()   (/root/reference/src/pyskani/_skani/lib.rs:549-660)
 * i.e. of the third-party calls those functions make into crate `skani` v0.3.0
 * (Cargo.lock:1599-1601, commit c57dbe72...; NOT vendored under /root/reference):
 *   skani::seeding::fmh_seeds            (call site lib.rs:165-171)
 *   skani::screen::check_markers_quickly (call site lib.rs:623-628)
 *   skani::chain::map_params_from_sketch (call site lib.rs:646-651)
 *   skani::chain::chain_seeds            (call site lib.rs:652-653)
 *
 * The skani source is absent, so this is a restatement of its published algorithm
 * (FracMinHash seeds -> marker screen -> chunked banded chaining -> per-chunk
 * (anchors/seeds)^(1/k) ANI, chain-length aligned fraction), with every constant
 * that could not be read from a file chosen by the scripted search recorded in
 * oracle/README.md. PINNING STATUS (see oracle/README.md, tests/test_oracle_kat.py):
 *   - aligned fractions and raw ANI (learned_ani=False): match pyskani's KATs
 *     (test_ani.py:28-61) to the reference's own 4 decimals;
 *   - median ANI: within 9e-5 of the KAT (inside BASELINE.json's 1e-4, outside 4 decimals);
 *   - learned-ANI KATs: unreachable (GBDT weights live in the absent crate); the regression STAGE is restated and
 *     runs with a supplied model (orc_model_predict);
 *   - round 2: no natural median / trimmed-mean variant pins the median or robust KAT to 5e-5 (oracle/README.md);
 *   - round 6: nor does any of 1 536 readings of what one value / its denominator is (tools/chunk_unit_sweep.py); over re-drawn seed samples the rule below moves by
 *     3-30 x the KAT tolerance and is consistent with all four reachable KATs (|z| <= 1.52): the seed sample, which the reference does not expose, decides the 4th decimals;
 *   - seed / marker sets: "parity unpinned" (the reference exposes none).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef SKANI_ORACLE_H
#define SKANI_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_K_MARKER 21          /* skani::params::K_MARKER_DNA */
#define ORC_MIN_LENGTH_CONTIG 500 /* skani::params::MIN_LENGTH_CONTIG, used at lib.rs:156 */

typedef struct {
    uint32_t kmer;    /* canonical 2-bit packed k-mer (k <= 16) */
    uint32_t pos;     /* index of the LAST base of the 21-base window */
    uint32_t contig;  /* index among KEPT contigs (lib.rs:146,173) */
    uint32_t canon;   /* 1 iff forward k-mer < reverse complement */
} orc_seed;

typedef struct {
    int c, marker_c, k;
    uint32_t n_contigs;        /* kept contigs */
    uint32_t* contig_len;      /* per kept contig */
    uint64_t total_len;        /* sum over kept contigs (lib.rs:161) */
    uint64_t n_seeds;          /* seeds in (contig, pos) order */
    orc_seed* seeds;
    uint64_t n_markers;        /* sorted, unique canonical 21-mers */
    uint64_t* markers;
    void* kindex;              /* the seeds once more, ordered by (k-mer, contig, pos): the sorted stand-in for the k-mer -> positions map a
                                * skani Sketch carries (built with the sketch, as skani fills its map while seeding; NULL without seeds) */
} orc_sketch;

/* Learned-ANI regression model (skani::regression::get_model, lib.rs:614): gradient-boosted regression trees with
 * the semantics of crate gbdt 0.1.3 (Cargo.lock:1608). skani's trained weights are embedded in the absent crate, so
 * a model is always supplied by the test (synthetic trees) — the oracle restates the EVALUATION, not the weights. */
typedef struct { int32_t feature; float threshold; int32_t left, right; float value; int32_t missing, is_leaf; } orc_node;
typedef struct {
    const orc_node* nodes; const uint32_t* first;   /* first[t]..first[t+1]: nodes of tree t, children relative to first[t] */
    uint32_t n_trees, n_features;
    const int32_t* features;                        /* ORC_F_* id of every position of the feature vector */
    float bias, shrinkage;
} orc_model;
enum { ORC_F_ANI100 = 0, ORC_F_STD100, ORC_F_Q90_QUERY, ORC_F_Q50_QUERY, ORC_F_Q10_QUERY, ORC_F_Q90_REF, ORC_F_Q50_REF, ORC_F_Q10_REF,
       ORC_F_AVG_CHAIN_LEN, ORC_F_AF_QUERY, ORC_F_AF_REF, ORC_F_N_CHUNKS, ORC_F_TOTAL_LEN_QUERY, ORC_F_TOTAL_LEN_REF,
       ORC_F_N_CONTIGS_QUERY, ORC_F_N_CONTIGS_REF, ORC_F_COUNT };
#define ORC_FEATURE_UNKNOWN (-3.402823466e+38F)   /* gbdt VALUE_TYPE_UNKNOWN = f32::MIN */

typedef struct {
    int learned_ani;   /* -1 = default rule (c >= 70 && !median, lib.rs:611-613) when a model is given, 0 = off, 1 = on (needs a model) */
    int median, robust;
    double screen_val; /* 0 -> 0.80 (lib.rs:603-609) */
    int rescue_small;  /* = !faster_small (lib.rs:597) */
    double min_aligned_frac; /* 0.15 (lib.rs:589-590) */
    const orc_model* model;
} orc_query_opts;

typedef struct {
    float ani, af_query, af_ref;
    /* integer intermediates, exposed so the GPU path can be compared bit-exactly */
    uint64_t n_anchors;
    uint32_t n_chunks;       /* chunks that produced an ANI estimate */
    uint32_t n_intervals;    /* kept chains */
    uint64_t covered_query, covered_ref;
    uint64_t sum_chain_anchors, sum_chunk_seeds;
    float ani_raw, ani_std;  /* chain ANI before the regression; sample std of the per-chunk estimates */
    uint32_t learned;
} orc_result;

/* gbdt 0.1.3 GBDT::predict for one row of model->n_features floats */
float orc_model_predict(const orc_model* m, const float* row);

uint64_t orc_mm_hash64(uint64_t key);

/* contigs shorter than ORC_MIN_LENGTH_CONTIG are skipped exactly as lib.rs:155-176 does */
orc_sketch* orc_sketch_new(const uint8_t* const* contigs, const uint64_t* lens, uint32_t n,
                           int c, int marker_c, int k, int want_seeds);
void orc_sketch_free(orc_sketch*);

/* check_markers_quickly(query, ref, screen_val, rescue_small) */
int orc_screen(const orc_sketch* q, const orc_sketch* r, double screen_val, int rescue_small,
               uint64_t* n_shared_out);

/* chain_seeds(ref, query, map_params_from_sketch(ref, ...)) */
int orc_chain(const orc_sketch* ref, const orc_sketch* query, const orc_query_opts* o, orc_result* out);

/* screen + chain of one query against n references in one call (multi-threaded CPU baselines: no interpreter lock inside) */
uint32_t orc_query_refs(const orc_sketch* const* refs, uint32_t n, const orc_sketch* q, const orc_query_opts* o,
                        uint32_t* hit_ref, orc_result* hit_res, uint32_t max_hits);

/* debug dumps for GPU parity tests: per-chunk records of the last orc_chain call on this thread */
typedef struct { uint32_t contig, left, right, anchors, seeds, n_intervals; } orc_chunk_rec;
uint32_t orc_last_chunks(const orc_chunk_rec** recs);

#ifdef __cplusplus
}
#endif
#endif
