/* tools/chunk_unit_sweep.c - TEST INFRASTRUCTURE (hypothesis search, never shipped, never linked into the product).
 *
 * The chaining of oracle/skani_oracle.c (orc_chain passes 1 and 2: anchors, chunks, banded DP, one candidate per DP tree, greedy selection over all candidates of the
 * pair) with two switches the oracle does not have - how the query is cut into chunks, and the DP's constants - and, instead of the oracle's per-chunk totals, the
 * list of KEPT CHAINS. tools/chunk_unit_sweep.py evaluates every reading of "one value of the ANI estimate" over those chains against the reference's known answers
 * (/root/reference/src/pyskani/tests/test_ani.py:28-61). With window_mode 0 and the default constants the kept chains are the oracle's (the script asserts it).  */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t kmer, pos, contig, canon; } seed_t;      /* = orc_seed */
typedef struct { uint32_t qc, qp, rp, rc, rev; } anchor_t;
typedef struct { int32_t score; uint32_t q0, q1, r0, r1, rc, qc, nanch, order, chunk; } chain_t;
typedef struct {
    int32_t window_mode;      /* 0: a chunk = anchors within `fragment` bases of the chunk's FIRST anchor (the oracle); 1: fixed grid, chunk = qpos / fragment */
    int32_t fragment;         /* 20000 */
    int32_t band, bp_band;    /* look-back in anchors (oracle: 2500 / c clamped to [1, 100]) and in query bases (2500) */
    int32_t max_gap;          /* 300 */
    int32_t gap_cost_x4;      /* gap cost per base of |dq - dr|, times 4 (oracle: 0.5 -> 2) */
    int32_t min_anchors;      /* 3 */
    int32_t min_score_x4;     /* 45 -> 180 */
    int32_t ref_overlap;      /* 1: a chain is also rejected when it overlaps a kept chain on the reference (the oracle) */
} sweep_params;

static int cmp_kseed(const void* a, const void* b) {
    const seed_t* x = a; const seed_t* y = b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
static int cmp_chain(const void* a, const void* b) {
    const chain_t* x = a; const chain_t* y = b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;
    return x->order < y->order ? -1 : x->order > y->order;
}

/* query seeds in (contig, pos) order, reference seeds in any order. Returns the number of kept chains written to `out` (capacity cap); *n_chunks = chunks with anchors;
 * chunk_lo[chunk] = first query position of the chunk's window (first anchor or grid start), chunk_qc[chunk] = its query contig (both sized chunk_cap). */
int64_t sweep_chains(const seed_t* q, uint64_t nq, const seed_t* r, uint64_t nr, const sweep_params* P, chain_t* out, uint64_t cap,
                     uint32_t* chunk_lo, uint32_t* chunk_qc, uint64_t chunk_cap, uint64_t* n_chunks_out, uint64_t* n_anchors_out) {
    seed_t* rs = malloc(sizeof(seed_t) * (nr ? nr : 1));
    memcpy(rs, r, sizeof(seed_t) * nr);
    qsort(rs, nr, sizeof(seed_t), cmp_kseed);
    uint64_t acap = nq + 1024, na = 0;
    anchor_t* A = malloc(sizeof(anchor_t) * acap);
    for (uint64_t i = 0; i < nq; i++) {
        uint32_t km = q[i].kmer;
        uint64_t l = 0, h = nr;
        while (l < h) { uint64_t m = (l + h) / 2; if (rs[m].kmer < km) l = m + 1; else h = m; }
        for (uint64_t j = l; j < nr && rs[j].kmer == km; j++) {
            if (na == acap) { acap *= 2; A = realloc(A, sizeof(anchor_t) * acap); }
            A[na].qc = q[i].contig; A[na].qp = q[i].pos; A[na].rp = rs[j].pos; A[na].rc = rs[j].contig; A[na].rev = q[i].canon != rs[j].canon;
            na++;
        }
    }
    *n_anchors_out = na;
    int64_t* f = malloc(sizeof(int64_t) * (na ? na : 1));
    uint32_t* root = malloc(sizeof(uint32_t) * (na ? na : 1));
    uint32_t* depth = malloc(sizeof(uint32_t) * (na ? na : 1));
    uint32_t* best = malloc(sizeof(uint32_t) * (na ? na : 1));
    chain_t* cands = malloc(sizeof(chain_t) * (na ? na : 1));
    const int64_t anchor_score = 80;      /* scores times 4: anchor score 20 */
    uint64_t nc = 0, nchunks = 0, s = 0;
    while (s < na) {
        uint64_t e = s;
        uint32_t lo;
        if (P->window_mode == 0) {
            lo = A[s].qp;
            uint64_t endp = (uint64_t)A[s].qp + (uint64_t)P->fragment;
            while (e < na && A[e].qc == A[s].qc && (uint64_t)A[e].qp <= endp) e++;
        } else {
            uint32_t g = A[s].qp / (uint32_t)P->fragment;
            lo = g * (uint32_t)P->fragment;
            while (e < na && A[e].qc == A[s].qc && A[e].qp / (uint32_t)P->fragment == g) e++;
        }
        for (uint64_t x = s; x < e; x++) {
            int64_t bs = anchor_score; uint64_t bp = x;
            for (uint64_t y = x; y-- > s && x - y <= (uint64_t)P->band;) {
                if (A[y].rc != A[x].rc || A[y].rev != A[x].rev) continue;
                int64_t dq = (int64_t)A[x].qp - (int64_t)A[y].qp;
                if (dq > P->bp_band) break;
                int64_t dr = A[x].rev ? (int64_t)A[y].rp - (int64_t)A[x].rp : (int64_t)A[x].rp - (int64_t)A[y].rp;
                if (dq <= 0 || dr <= 0) continue;
                int64_t gap = dq > dr ? dq - dr : dr - dq;
                if (gap > P->max_gap) continue;
                int64_t sc = f[y] + anchor_score - gap * P->gap_cost_x4;
                if (sc > bs) { bs = sc; bp = y; }
            }
            f[x] = bs;
            if (bp == x) { root[x] = (uint32_t)x; depth[x] = 1; }
            else { root[x] = root[bp]; depth[x] = depth[bp] + 1; }
        }
        for (uint64_t x = s; x < e; x++) best[x] = UINT32_MAX;
        for (uint64_t x = s; x < e; x++) { uint32_t rt = root[x]; if (best[rt] == UINT32_MAX || f[x] > f[best[rt]]) best[rt] = (uint32_t)x; }
        for (uint64_t x = s; x < e; x++) {
            if (root[x] != x) continue;
            uint32_t b = best[x];
            if ((int32_t)depth[b] < P->min_anchors || f[b] < P->min_score_x4) continue;
            chain_t* cd = &cands[nc];
            cd->score = (int32_t)f[b]; cd->q0 = A[x].qp; cd->q1 = A[b].qp; cd->nanch = depth[b]; cd->order = (uint32_t)nc; cd->chunk = (uint32_t)nchunks; cd->rc = A[x].rc; cd->qc = A[x].qc;
            cd->r0 = A[x].rp < A[b].rp ? A[x].rp : A[b].rp; cd->r1 = A[x].rp < A[b].rp ? A[b].rp : A[x].rp;
            nc++;
        }
        if (nchunks < chunk_cap) { chunk_lo[nchunks] = lo; chunk_qc[nchunks] = A[s].qc; }
        nchunks++;
        s = e;
    }
    *n_chunks_out = nchunks;
    qsort(cands, nc, sizeof(chain_t), cmp_chain);
    uint64_t nk = 0;
    for (uint64_t i = 0; i < nc; i++) {
        int ok = 1;
        for (uint64_t j = 0; j < nk && ok; j++) {
            if (cands[i].chunk == out[j].chunk && !(cands[i].q1 < out[j].q0 || cands[i].q0 > out[j].q1)) ok = 0;
            else if (P->ref_overlap && cands[i].rc == out[j].rc && !(cands[i].r1 < out[j].r0 || cands[i].r0 > out[j].r1)) ok = 0;
        }
        if (ok) { if (nk < cap) out[nk] = cands[i]; nk++; }
    }
    free(rs); free(A); free(f); free(root); free(depth); free(best); free(cands);
    return (int64_t)nk;
}

/* The oracle's seeding (orc_sketch_new, one contig) with a SALT xor-ed into the k-mer before it is hashed: salt 0 gives the oracle's seed set (asserted by the script), any
 * other salt an independent FracMinHash sample of the same density. The four known answers are functions of WHICH 1-in-c k-mers are seeds; re-drawing the sample measures how
 * far the numbers move by that alone (tools/chunk_unit_sweep.py --salts). Returns the number of seeds written (<= cap), in position order. */
static uint64_t mm_hash64(uint64_t key) {
    key = ~(key + (key << 21));
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}
int64_t sweep_sketch(const uint8_t* str, uint64_t len, int c, int k, uint64_t salt, seed_t* out, uint64_t cap) {
    static uint8_t T[256]; static int ready = 0;
    if (!ready) { memset(T, 0, 256); T['C'] = T['c'] = 1; T['G'] = T['g'] = 2; T['T'] = T['t'] = 3; ready = 1; }
    const int mk = 21, off_lo = (mk - k) / 2, shift_f = 2 * (mk - k - off_lo), shift_r = 2 * off_lo;
    const uint64_t mmask = (~0ULL) >> (64 - 2 * mk), smask = (~0ULL) >> (64 - 2 * k), thr = UINT64_MAX / (uint64_t)c;
    uint64_t f = 0, r = 0, n = 0;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t b = T[str[i]];
        f = ((f << 2) | b) & mmask;
        r = (r >> 2) | ((3 - b) << (2 * (mk - 1)));
        if (i < (uint64_t)mk - 1) continue;
        uint64_t fs = (f >> shift_f) & smask, rs = (r >> shift_r) & smask;
        int canon = fs < rs;
        uint64_t cs = canon ? fs : rs;
        if (mm_hash64(cs ^ salt) < thr) {
            if (n < cap) { out[n].kmer = (uint32_t)cs; out[n].pos = (uint32_t)i; out[n].contig = 0; out[n].canon = (uint32_t)canon; }
            n++;
        }
    }
    return (int64_t)n;
}
