/* oracle/skani_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT. See skani_oracle.h for
 * scope, the reference call sites this restates, and the pinning status.           */
#include "skani_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants restated from skani::params (source absent; see oracle/README.md) ---- */
#define FRAGMENT_LENGTH 20000   /* chunk length on the query                         */
#define MAX_GAP_LENGTH 300      /* max |dq - dr| between chained anchors             */
#define MIN_ANCHORS 3
#define BP_CHAIN_BAND 2500      /* look-back in query bases                          */
#define MAX_CHAIN_BAND 100      /* look-back in anchors = clamp(BP_CHAIN_BAND / c, 1, 100) */
/* scores are kept DOUBLED so that the gap cost |dq - dr| / 2 stays integral:         */
#define ANCHOR_SCORE2 40        /* anchor score 20                                   */
#define MIN_SCORE2 90           /* 0.75 * MIN_ANCHORS * ANCHOR_SCORE = 45            */
#define SMALL_MARKER_COUNT 20   /* "less than 20 marker k-mers", lib.rs:538-541      */

/* skani::types::mm_hash64: minimap2's invertible mix, with the first line as the Rust
 * expression `!key.wrapping_add(key << 21)` parses: NOT of the sum.                  */
uint64_t orc_mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}

static uint8_t BYTE_TO_SEQ[256];
static int tab_ready = 0;
static void init_tab(void) {
    if (tab_ready) return;
    memset(BYTE_TO_SEQ, 0, sizeof BYTE_TO_SEQ);
    BYTE_TO_SEQ['C'] = BYTE_TO_SEQ['c'] = 1;
    BYTE_TO_SEQ['G'] = BYTE_TO_SEQ['g'] = 2;
    BYTE_TO_SEQ['T'] = BYTE_TO_SEQ['t'] = 3;
    tab_ready = 1;
}

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}

/* fmh_seeds (lib.rs:165-171) applied by the _sketch driver (lib.rs:140-185). */
static void build_kindex(orc_sketch* s);

orc_sketch* orc_sketch_new(const uint8_t* const* contigs, const uint64_t* lens, uint32_t n,
                           int c, int marker_c, int k, int want_seeds) {
    init_tab();
    if (k < 1 || k > 16 || c < 1 || marker_c < 1) return NULL;
    orc_sketch* s = calloc(1, sizeof *s);
    s->c = c; s->marker_c = marker_c; s->k = k;
    s->contig_len = malloc(sizeof(uint32_t) * (n ? n : 1));
    uint64_t cap_s = 1024, cap_m = 1024;
    s->seeds = malloc(sizeof(orc_seed) * cap_s);
    s->markers = malloc(sizeof(uint64_t) * cap_m);
    const int mk = ORC_K_MARKER;
    const int off_lo = (mk - k) / 2;                 /* seed k-mer is centred in the window */
    const int shift_f = 2 * (mk - k - off_lo), shift_r = 2 * off_lo;
    const uint64_t mmask = (~0ULL) >> (64 - 2 * mk);
    const uint64_t smask = (~0ULL) >> (64 - 2 * k);
    const uint64_t thr = UINT64_MAX / (uint64_t)c, thr_m = UINT64_MAX / (uint64_t)marker_c;
    for (uint32_t ci = 0; ci < n; ci++) {
        const uint8_t* str = contigs[ci];
        uint64_t len = lens[ci];
        if (len < ORC_MIN_LENGTH_CONTIG) continue;       /* lib.rs:156 */
        uint32_t contig_index = s->n_contigs;            /* counts kept contigs, lib.rs:146,173 */
        s->contig_len[s->n_contigs++] = (uint32_t)len;   /* lib.rs:158-160 */
        s->total_len += len;                             /* lib.rs:161 */
        uint64_t f = 0, r = 0;
        for (uint64_t i = 0; i < len; i++) {
            uint64_t b = BYTE_TO_SEQ[str[i]];
            f = ((f << 2) | b) & mmask;
            r = (r >> 2) | ((3 - b) << (2 * (mk - 1)));
            if (i < (uint64_t)mk - 1) continue;
            uint64_t fs = (f >> shift_f) & smask, rs = (r >> shift_r) & smask;
            int canon = fs < rs;
            uint64_t cs = canon ? fs : rs;
            uint64_t h = orc_mm_hash64(cs);
            if (h < thr) {
                if (want_seeds) {
                    if (s->n_seeds == cap_s) { cap_s *= 2; s->seeds = realloc(s->seeds, sizeof(orc_seed) * cap_s); }
                    orc_seed* o = &s->seeds[s->n_seeds++];
                    o->kmer = (uint32_t)cs; o->pos = (uint32_t)i; o->contig = contig_index; o->canon = (uint32_t)canon;
                }
                if (h < thr_m) {
                    if (s->n_markers == cap_m) { cap_m *= 2; s->markers = realloc(s->markers, sizeof(uint64_t) * cap_m); }
                    s->markers[s->n_markers++] = f < r ? f : r;
                }
            }
        }
    }
    /* marker_seeds is a set: sorted + unique is its canonical form */
    qsort(s->markers, s->n_markers, sizeof(uint64_t), cmp_u64);
    uint64_t w = 0;
    for (uint64_t i = 0; i < s->n_markers; i++)
        if (i == 0 || s->markers[i] != s->markers[i - 1]) s->markers[w++] = s->markers[i];
    s->n_markers = w;
    build_kindex(s);
    return s;
}

void orc_sketch_free(orc_sketch* s) {
    if (!s) return;
    free(s->contig_len); free(s->seeds); free(s->markers); free(s->kindex); free(s);
}

/* check_markers_quickly (lib.rs:623-628): containment of the smaller marker set. */
int orc_screen(const orc_sketch* q, const orc_sketch* r, double screen_val, int rescue_small,
               uint64_t* n_shared_out) {
    uint64_t nq = q->n_markers, nr = r->n_markers;
    uint64_t small = nq < nr ? nq : nr;
    uint64_t i = 0, j = 0, shared = 0;
    while (i < nq && j < nr) {
        if (q->markers[i] < r->markers[j]) i++;
        else if (q->markers[i] > r->markers[j]) j++;
        else { shared++; i++; j++; }
    }
    if (n_shared_out) *n_shared_out = shared;
    if (rescue_small && small < SMALL_MARKER_COUNT) return 1;
    if (small == 0) return 0;
    double thresh = pow(screen_val, (double)ORC_K_MARKER);
    return (double)shared / (double)small > thresh;
}

typedef struct { uint32_t qc, qp, rp, rc, rev; } anchor_t;
typedef struct { uint32_t kmer, pos, contig, canon; } kseed_t;

static int cmp_kseed(const void* a, const void* b) {
    const kseed_t* x = a; const kseed_t* y = b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}

/* the reference-side index of chaining: built once per sketch (skani inserts every seed into the sketch's k-mer map while seeding) */
static void build_kindex(orc_sketch* s) {
    s->kindex = NULL;
    if (s->n_seeds == 0) return;
    kseed_t* rs = malloc(sizeof(kseed_t) * s->n_seeds);
    for (uint64_t i = 0; i < s->n_seeds; i++) { rs[i].kmer = s->seeds[i].kmer; rs[i].pos = s->seeds[i].pos; rs[i].contig = s->seeds[i].contig; rs[i].canon = s->seeds[i].canon; }
    qsort(rs, s->n_seeds, sizeof(kseed_t), cmp_kseed);
    s->kindex = rs;
}

typedef struct { int32_t score; uint32_t q0, q1, r0, r1, rc, nanch, order, chunk; } cand_t;
static int cmp_cand(const void* a, const void* b) {      /* score desc, stable by generation order */
    const cand_t* x = a; const cand_t* y = b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->order < y->order ? -1 : x->order > y->order;
}
static int cmp_dbl(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : x > y;
}

static __thread orc_chunk_rec* g_recs = NULL;
static __thread uint32_t g_nrecs = 0;
uint32_t orc_last_chunks(const orc_chunk_rec** recs) { *recs = g_recs; return g_nrecs; }

/* number of query seeds on `contig` with pos in [lo, hi]; seeds are in (contig,pos) order */
static uint32_t seeds_between(const orc_sketch* q, const uint64_t* cstart, uint32_t contig, uint32_t lo, uint32_t hi) {
    uint64_t a = cstart[contig], b = cstart[contig + 1];
    uint64_t l = a, r = b;
    while (l < r) { uint64_t m = (l + r) / 2; if (q->seeds[m].pos < lo) l = m + 1; else r = m; }
    uint64_t first = l; r = b;
    while (l < r) { uint64_t m = (l + r) / 2; if (q->seeds[m].pos <= hi) l = m + 1; else r = m; }
    return (uint32_t)(l - first);
}

/* gbdt 0.1.3 (Cargo.lock:1608) DecisionTree::predict_one / GBDT::predict, SquaredError loss: at an inner node go left
 * iff x[feature] < threshold; an UNKNOWN feature follows `missing` (-1 left, 0 stop at this node, +1 right);
 * prediction = bias + shrinkage * sum_t value(reached node of tree t), accumulated in tree order in f32. */
float orc_model_predict(const orc_model* m, const float* row) {
    float acc = m->bias;
    for (uint32_t t = 0; t < m->n_trees; t++) {
        const orc_node* T = m->nodes + m->first[t];
        uint32_t tn = m->first[t + 1] - m->first[t], i = 0;
        float v = 0.0f;
        for (uint32_t step = 0; step <= tn; step++) {
            const orc_node* nd = &T[i];
            v = nd->value;
            if (nd->is_leaf) break;
            float x = row[nd->feature];
            int go = x == ORC_FEATURE_UNKNOWN ? nd->missing : (x < nd->threshold ? -1 : 1);
            if (go == 0) break;
            i = (uint32_t)(go < 0 ? nd->left : nd->right);
        }
        acc += m->shrinkage * v;
    }
    return acc;
}

static int cmp_u32(const void* a, const void* b) { uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : x > y; }
/* contig-length quantiles {q90, q50, q10}: sorted kept-contig lengths at n*9/10, n/2, n/10 */
static void len_quantiles(const orc_sketch* s, float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    uint32_t n = s->n_contigs;
    if (!n) return;
    uint32_t* v = malloc(sizeof(uint32_t) * n);
    memcpy(v, s->contig_len, sizeof(uint32_t) * n);
    qsort(v, n, sizeof(uint32_t), cmp_u32);
    uint64_t i90 = (uint64_t)n * 9 / 10, i50 = n / 2, i10 = n / 10;
    out[0] = (float)v[i90 < n ? i90 : n - 1]; out[1] = (float)v[i50 < n ? i50 : n - 1]; out[2] = (float)v[i10 < n ? i10 : n - 1];
    free(v);
}

/* chain_seeds (lib.rs:652-653) with the MapParams of map_params_from_sketch (lib.rs:646-651). */
int orc_chain(const orc_sketch* ref, const orc_sketch* query, const orc_query_opts* o, orc_result* out) {
    memset(out, 0, sizeof *out);
    out->ani = -1.0f;
    free(g_recs); g_recs = NULL; g_nrecs = 0;
    if (o->learned_ani == 1 && !o->model) return -2;  /* GBDT weights live inside the absent crate: a model must be supplied */
    const int k = ref->k, c = ref->c;
    uint64_t nq = query->n_seeds, nr = ref->n_seeds;
    if (nq == 0 || nr == 0) return 0;
    /* reference index: seeds ordered by (kmer, contig, pos) = a stable sort by k-mer of the (contig,pos)-ordered seeds — the sorted stand-in for the k-mer map */
    const kseed_t* rs = (const kseed_t*)ref->kindex;      /* built with the sketch (build_kindex) */
    /* anchors: every (query seed, ref seed) pair with equal k-mer; walking query seeds in (contig,pos)
     * order and ref matches in (contig,pos) order yields them sorted by (qc,qp,rc,rp) */
    uint64_t cap = nq + 1024, na = 0;
    anchor_t* A = malloc(sizeof(anchor_t) * cap);
    for (uint64_t i = 0; i < nq; i++) {
        uint32_t km = query->seeds[i].kmer;
        uint64_t l = 0, r = nr;
        while (l < r) { uint64_t m = (l + r) / 2; if (rs[m].kmer < km) l = m + 1; else r = m; }
        for (uint64_t j = l; j < nr && rs[j].kmer == km; j++) {
            if (na == cap) { cap *= 2; A = realloc(A, sizeof(anchor_t) * cap); }
            A[na].qc = query->seeds[i].contig; A[na].qp = query->seeds[i].pos;
            A[na].rp = rs[j].pos; A[na].rc = rs[j].contig; A[na].rev = query->seeds[i].canon != rs[j].canon;
            na++;
        }
    }
    out->n_anchors = na;
    if (na == 0) { free(A); return 0; }
    /* per-contig start offsets of the query seeds */
    uint64_t* cstart = calloc(query->n_contigs + 2, sizeof(uint64_t));
    for (uint64_t i = 0; i < nq; i++) cstart[query->seeds[i].contig + 1]++;
    for (uint32_t i = 0; i < query->n_contigs; i++) cstart[i + 1] += cstart[i];

    int32_t* f = malloc(sizeof(int32_t) * na);
    uint32_t* root = malloc(sizeof(uint32_t) * na);
    uint32_t* depth = malloc(sizeof(uint32_t) * na);
    uint32_t* best = malloc(sizeof(uint32_t) * na);
    cand_t* cands = malloc(sizeof(cand_t) * na);
    int band = BP_CHAIN_BAND / c; if (band < 1) band = 1; if (band > MAX_CHAIN_BAND) band = MAX_CHAIN_BAND;

    /* pass 1: per chunk, banded chaining DP and one candidate chain per DP tree */
    uint32_t nc = 0, n_chunks_all = 0;
    uint64_t chunk_cap = 256; uint32_t* chunk_qc = malloc(sizeof(uint32_t) * chunk_cap);
    uint64_t s = 0;
    while (s < na) {
        /* chunk = run of anchors on one query contig within FRAGMENT_LENGTH of the chunk's first anchor */
        uint64_t e = s; uint64_t endp = (uint64_t)A[s].qp + FRAGMENT_LENGTH;
        while (e < na && A[e].qc == A[s].qc && (uint64_t)A[e].qp <= endp) e++;
        for (uint64_t x = s; x < e; x++) {
            int32_t bs = ANCHOR_SCORE2; uint64_t bp = x;
            for (uint64_t y = x; y-- > s && x - y <= (uint64_t)band;) {
                if (A[y].rc != A[x].rc || A[y].rev != A[x].rev) continue;
                int64_t dq = (int64_t)A[x].qp - (int64_t)A[y].qp;
                if (dq > BP_CHAIN_BAND) break;
                int64_t dr = A[x].rev ? (int64_t)A[y].rp - (int64_t)A[x].rp : (int64_t)A[x].rp - (int64_t)A[y].rp;
                if (dq <= 0 || dr <= 0) continue;
                int64_t gap = dq > dr ? dq - dr : dr - dq;
                if (gap > MAX_GAP_LENGTH) continue;
                int32_t sc = f[y] + ANCHOR_SCORE2 - (int32_t)gap;      /* = 2 * (f + 20 - gap/2) */
                if (sc > bs) { bs = sc; bp = y; }
            }
            f[x] = bs;
            if (bp == x) { root[x] = (uint32_t)x; depth[x] = 1; }
            else { root[x] = root[bp]; depth[x] = depth[bp] + 1; }
        }
        /* the tree's best-scoring anchor (lowest index on ties), backtracked to the root */
        for (uint64_t x = s; x < e; x++) best[x] = UINT32_MAX;
        for (uint64_t x = s; x < e; x++) { uint32_t rt = root[x]; if (best[rt] == UINT32_MAX || f[x] > f[best[rt]]) best[rt] = (uint32_t)x; }
        for (uint64_t x = s; x < e; x++) {
            if (root[x] != x) continue;
            uint32_t b = best[x];
            if (depth[b] < MIN_ANCHORS || f[b] < MIN_SCORE2) continue;
            cand_t* cd = &cands[nc];
            cd->score = f[b]; cd->q0 = A[x].qp; cd->q1 = A[b].qp; cd->nanch = depth[b]; cd->order = nc; cd->chunk = n_chunks_all; cd->rc = A[x].rc;
            cd->r0 = A[x].rp < A[b].rp ? A[x].rp : A[b].rp; cd->r1 = A[x].rp < A[b].rp ? A[b].rp : A[x].rp;
            nc++;
        }
        if (n_chunks_all == chunk_cap) { chunk_cap *= 2; chunk_qc = realloc(chunk_qc, sizeof(uint32_t) * chunk_cap); }
        chunk_qc[n_chunks_all++] = A[s].qc;
        s = e;
    }
    /* pass 2: greedy selection over ALL candidates of the pair by score: a chain is kept unless it overlaps
     * a kept chain on the query (same chunk) or on the reference (same ref contig) */
    qsort(cands, nc, sizeof(cand_t), cmp_cand);
    cand_t* kept = malloc(sizeof(cand_t) * (nc ? nc : 1));
    uint32_t nk = 0;
    for (uint32_t i = 0; i < nc; i++) {
        int ok = 1;
        for (uint32_t j = 0; j < nk && ok; j++) {
            if (cands[i].chunk == kept[j].chunk && !(cands[i].q1 < kept[j].q0 || cands[i].q0 > kept[j].q1)) ok = 0;
            else if (cands[i].rc == kept[j].rc && !(cands[i].r1 < kept[j].r0 || cands[i].r0 > kept[j].r1)) ok = 0;
        }
        if (ok) kept[nk++] = cands[i];
    }
    /* pass 3: per chunk totals */
    uint32_t* c_anch = calloc(n_chunks_all ? n_chunks_all : 1, sizeof(uint32_t));
    uint32_t* c_left = malloc(sizeof(uint32_t) * (n_chunks_all ? n_chunks_all : 1));
    uint32_t* c_right = calloc(n_chunks_all ? n_chunks_all : 1, sizeof(uint32_t));
    uint32_t* c_nint = calloc(n_chunks_all ? n_chunks_all : 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n_chunks_all; i++) c_left[i] = UINT32_MAX;
    for (uint32_t j = 0; j < nk; j++) {
        uint32_t ck = kept[j].chunk;
        c_anch[ck] += kept[j].nanch; c_nint[ck]++;
        if (kept[j].q0 < c_left[ck]) c_left[ck] = kept[j].q0;
        if (kept[j].q1 > c_right[ck]) c_right[ck] = kept[j].q1;
        out->covered_query += (uint64_t)(kept[j].q1 - kept[j].q0) + 1 + 2 * (uint64_t)c;
    }
    out->covered_ref = out->covered_query;      /* one covered-bases count serves both fractions */
    uint64_t rec_cap = 256; g_recs = malloc(sizeof(orc_chunk_rec) * rec_cap);
    uint64_t dcap = 256, nd = 0; double* anis = malloc(sizeof(double) * dcap);
    for (uint32_t ck = 0; ck < n_chunks_all; ck++) {
        if (!c_nint[ck]) continue;
        uint32_t ns = seeds_between(query, cstart, chunk_qc[ck], c_left[ck], c_right[ck]);
        /* the two end seeds are anchors by construction: identity over the ns - 1 seeds after the first */
        uint32_t denom = ns > 1 ? ns - 1 : 1;
        double ratio = (double)c_anch[ck] / (double)denom; if (ratio > 1.0) ratio = 1.0;
        if (nd == dcap) { dcap *= 2; anis = realloc(anis, sizeof(double) * dcap); }
        anis[nd++] = pow(ratio, 1.0 / (double)k);
        if (g_nrecs == rec_cap) { rec_cap *= 2; g_recs = realloc(g_recs, sizeof(orc_chunk_rec) * rec_cap); }
        orc_chunk_rec* rcd = &g_recs[g_nrecs++];
        rcd->contig = chunk_qc[ck]; rcd->left = c_left[ck]; rcd->right = c_right[ck]; rcd->anchors = c_anch[ck]; rcd->seeds = ns; rcd->n_intervals = c_nint[ck];
        out->n_intervals += c_nint[ck]; out->sum_chain_anchors += c_anch[ck]; out->sum_chunk_seeds += ns;
    }
    out->n_chunks = (uint32_t)nd;
    if (nd) {
        double ani;
        if (o->median || o->robust) qsort(anis, nd, sizeof(double), cmp_dbl);
        if (o->median) ani = anis[nd / 2];
        else {
            uint64_t lo = 0, hi = nd;
            if (o->robust && nd - 2 * (nd / 10) > 0) { lo = nd / 10; hi = nd - nd / 10; }
            double sum = 0; for (uint64_t i = lo; i < hi; i++) sum += anis[i];
            ani = sum / (double)(hi - lo);
        }
        double afq = (double)out->covered_query / (double)query->total_len; if (afq > 1) afq = 1;
        double afr = (double)out->covered_query / (double)ref->total_len; if (afr > 1) afr = 1;
        out->af_query = (float)afq; out->af_ref = (float)afr;
        if (afq >= o->min_aligned_frac || afr >= o->min_aligned_frac) out->ani = (float)ani;
        /* sample standard deviation of all chunk estimates (a regression feature) */
        double mean_all = 0, ssq = 0;
        for (uint64_t i = 0; i < nd; i++) mean_all += anis[i];
        mean_all /= (double)nd;
        for (uint64_t i = 0; i < nd; i++) ssq += (anis[i] - mean_all) * (anis[i] - mean_all);
        out->ani_raw = out->ani; out->ani_std = nd > 1 ? (float)sqrt(ssq / (double)(nd - 1)) : 0.0f;
        /* learned ANI (lib.rs:611-614): explicit request, or the default rule c >= 70 && !median, when a model is present */
        int learned = o->model && (o->learned_ani == 1 || (o->learned_ani == -1 && c >= 70 && !o->median));
        if (learned && out->ani > 0.0f) {
            float fm[ORC_F_COUNT], row[64];
            float lq[3], lr[3];
            len_quantiles(query, lq); len_quantiles(ref, lr);
            fm[ORC_F_ANI100] = out->ani_raw * 100.0f; fm[ORC_F_STD100] = out->ani_std * 100.0f;
            fm[ORC_F_Q90_QUERY] = lq[0]; fm[ORC_F_Q50_QUERY] = lq[1]; fm[ORC_F_Q10_QUERY] = lq[2];
            fm[ORC_F_Q90_REF] = lr[0]; fm[ORC_F_Q50_REF] = lr[1]; fm[ORC_F_Q10_REF] = lr[2];
            fm[ORC_F_AVG_CHAIN_LEN] = out->n_intervals ? (float)out->covered_query / (float)out->n_intervals : 0.0f;
            fm[ORC_F_AF_QUERY] = out->af_query; fm[ORC_F_AF_REF] = out->af_ref; fm[ORC_F_N_CHUNKS] = (float)out->n_chunks;
            fm[ORC_F_TOTAL_LEN_QUERY] = (float)query->total_len; fm[ORC_F_TOTAL_LEN_REF] = (float)ref->total_len;
            fm[ORC_F_N_CONTIGS_QUERY] = (float)query->n_contigs; fm[ORC_F_N_CONTIGS_REF] = (float)ref->n_contigs;
            for (uint32_t j = 0; j < o->model->n_features && j < 64; j++) row[j] = fm[o->model->features[j]];
            float pred = orc_model_predict(o->model, row) * 0.01f;
            out->ani = pred < 0.0f ? 0.0f : (pred > 1.0f ? 1.0f : pred);
            out->learned = 1;
        }
    }
    free(A); free(cstart); free(f); free(root); free(depth); free(best); free(cands); free(kept); free(anis);
    free(chunk_qc); free(c_anch); free(c_left); free(c_right); free(c_nint);
    return 0;
}

/* The screen + chain loops of Database.query (lib.rs:617-657) for one query against n references, entirely in C so that a
 * multi-threaded CPU baseline (one query per thread) does not serialise on the Python interpreter lock. Returns the number of
 * hits (ani > 0.1, lib.rs:654); hits_out (may be NULL) receives up to max_hits (reference index, result) records. */
uint32_t orc_query_refs(const orc_sketch* const* refs, uint32_t n, const orc_sketch* q, const orc_query_opts* o,
                        uint32_t* hit_ref, orc_result* hit_res, uint32_t max_hits) {
    double screen_val = o->screen_val > 0 ? o->screen_val : 0.80;
    uint32_t nh = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (!orc_screen(q, refs[i], screen_val, o->rescue_small, NULL)) continue;
        orc_result res;
        if (orc_chain(refs[i], q, o, &res) != 0) continue;
        if (res.ani > 0.1f) {
            if (nh < max_hits) { if (hit_ref) hit_ref[nh] = i; if (hit_res) hit_res[nh] = res; }
            nh++;
        }
    }
    return nh;
}
