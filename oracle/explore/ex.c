// Exploration harness (NOT the oracle): switchable restatement used to search the
// hypothesis space of skani v0.3.0 details against the pyskani KATs. Kept for the
// record of how the oracle's constants were chosen; see oracle/README.md.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t kmer; uint32_t pos; uint32_t contig; uint8_t canon; } seed_t;

static inline uint64_t mmh(uint64_t key, int variant) {
    if (variant == 0) key = ~(key + (key << 21));
    else key = (~key) + (key << 21);
    key = key ^ key >> 24;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14;
    key = (key + (key << 2)) + (key << 4);
    key = key ^ key >> 28;
    key = key + (key << 31);
    return key;
}

static uint8_t B2S[256];
static void init_tab(void) {
    memset(B2S, 0, 256);
    B2S['C'] = B2S['c'] = 1; B2S['G'] = B2S['g'] = 2; B2S['T'] = B2S['t'] = 3;
}

// returns number of seeds; seeds in position order
long ex_sketch(const uint8_t* s, long len, int c, int marker_c, int k, int hashvar,
               uint32_t contig, seed_t* out, uint64_t* markers, long* n_markers, int marker_mode) {
    init_tab();
    const int mk = 21;
    if (len < mk) { *n_markers = 0; return 0; }
    uint64_t f = 0, r = 0;
    const uint64_t mmask = (~0ULL) >> (64 - 2 * mk);
    const uint64_t smask = (~0ULL) >> (64 - 2 * k);
    const uint64_t thr = UINT64_MAX / (uint64_t)c, thrm = UINT64_MAX / (uint64_t)marker_c;
    long n = 0, nm = 0;
    for (long i = 0; i < len; i++) {
        uint64_t b = B2S[s[i]];
        f = ((f << 2) | b) & mmask;
        r = (r >> 2) | ((3 - b) << (2 * (mk - 1)));
        if (i < mk - 1) continue;
        uint64_t fs = f & smask;
        uint64_t rs = r >> (2 * (mk - k));
        int canon = fs < rs;
        uint64_t cs = canon ? fs : rs;
        uint64_t h = mmh(cs, hashvar);
        if (h < thr) {
            out[n].kmer = cs; out[n].pos = (uint32_t)i; out[n].contig = contig; out[n].canon = (uint8_t)canon; n++;
            uint64_t cm = f < r ? f : r;
            int keep;
            if (marker_mode == 0) keep = h < thrm;            // seed hash decides
            else keep = mmh(cm, hashvar) < thrm;               // marker hash decides (nested)
            if (keep) markers[nm++] = cm;
        }
    }
    *n_markers = nm;
    return n;
}

typedef struct { uint32_t qc, qp, rp, rc; uint8_t rev; } anchor_t;

static int cmp_seed_kmer(const void* a, const void* b) {
    const seed_t* x = a; const seed_t* y = b;
    if (x->kmer != y->kmer) return x->kmer < y->kmer ? -1 : 1;
    if (x->contig != y->contig) return x->contig < y->contig ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
static int cmp_anchor(const void* a, const void* b) {
    const anchor_t* x = a; const anchor_t* y = b;
    if (x->qc != y->qc) return x->qc < y->qc ? -1 : 1;
    if (x->qp != y->qp) return x->qp < y->qp ? -1 : 1;
    if (x->rp != y->rp) return x->rp < y->rp ? -1 : 1;
    if (x->rc != y->rc) return x->rc < y->rc ? -1 : 1;
    if (x->rev != y->rev) return x->rev < y->rev ? -1 : 1;
    return 0;
}

typedef struct {
    int frag_len;       // 20000
    double max_gap;     // 50
    double anchor_score;// 20
    int min_anchors;    // 3
    int band;           // anchors look-back
    int bp_band;        // 2500?
    double max_lin;     // 5000
    int k;
    int mult_cap;       // 0 = none
    int chunk_mode;     // 0: fixed grid pos/frag ; 1: anchored at first anchor
    int gapcost_mode;   // 0: |dq-dr| ; 1: 0 ; 2: 0.5*|d|
    int chainset_mode;  // 0: best path per union-find set ; 1: set size as anchors
    int require_mono;   // require dr>0
    double gap_w;       // if > 0: gap cost = gap_w * |dq-dr| (overrides gapcost_mode)
} cparams_t;

typedef struct { int chunk; uint32_t qc, q0, q1, rc, r0, r1; int nanch; int nseeds; double score; int rev; int setsize; } interval_t;

static int uf_find(int* p, int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }

// returns number of intervals (all candidate chains, before non-overlap selection)
long ex_chain(const seed_t* qs_in, long nq, const seed_t* rs_in, long nr, const cparams_t* P,
              interval_t* out, long max_out, long* n_anchors_out, anchor_t* anchors_out, int* anchor_chunk_out) {
    seed_t* qs = malloc(sizeof(seed_t) * nq); memcpy(qs, qs_in, sizeof(seed_t) * nq);
    seed_t* rs = malloc(sizeof(seed_t) * nr); memcpy(rs, rs_in, sizeof(seed_t) * nr);
    qsort(qs, nq, sizeof(seed_t), cmp_seed_kmer);
    qsort(rs, nr, sizeof(seed_t), cmp_seed_kmer);
    long cap = 1 << 20, na = 0;
    anchor_t* A = malloc(sizeof(anchor_t) * cap);
    long i = 0, j = 0;
    while (i < nq && j < nr) {
        if (qs[i].kmer < rs[j].kmer) i++;
        else if (qs[i].kmer > rs[j].kmer) j++;
        else {
            long i2 = i, j2 = j;
            while (i2 < nq && qs[i2].kmer == qs[i].kmer) i2++;
            while (j2 < nr && rs[j2].kmer == rs[j].kmer) j2++;
            long mult = (i2 - i) * (j2 - j);
            if (!(P->mult_cap > 0 && mult > P->mult_cap)) {
                for (long a = i; a < i2; a++) for (long b = j; b < j2; b++) {
                    if (na == cap) { cap *= 2; A = realloc(A, sizeof(anchor_t) * cap); }
                    A[na].qc = qs[a].contig; A[na].qp = qs[a].pos; A[na].rp = rs[b].pos; A[na].rc = rs[b].contig;
                    A[na].rev = qs[a].canon != rs[b].canon; na++;
                }
            }
            i = i2; j = j2;
        }
    }
    qsort(A, na, sizeof(anchor_t), cmp_anchor);
    *n_anchors_out = na;
    // chunk assignment
    int* chunk = malloc(sizeof(int) * (na + 1));
    int nch = 0;
    {
        uint32_t curc = 0xffffffff; long endp = 0;
        for (long a = 0; a < na; a++) {
            if (P->chunk_mode == 2) { if (A[a].qc != curc) { curc = A[a].qc; if (a) nch++; } chunk[a] = nch; }
            else if (P->chunk_mode == 0) {
                // fixed grid per contig
                if (A[a].qc != curc) { curc = A[a].qc; endp = P->frag_len; if (a) nch++; }
                while ((long)A[a].qp >= endp) { endp += P->frag_len; nch++; }
                chunk[a] = nch;
            } else {
                if (A[a].qc != curc) { curc = A[a].qc; endp = (long)A[a].qp + P->frag_len; if (a) nch++; }
                else if ((long)A[a].qp > endp) { endp = (long)A[a].qp + P->frag_len; nch++; }
                chunk[a] = nch;
            }
        }
        nch++;
    }
    if (anchors_out) { memcpy(anchors_out, A, sizeof(anchor_t) * na); memcpy(anchor_chunk_out, chunk, sizeof(int) * na); }
    // DP per chunk
    double* f = malloc(sizeof(double) * na);
    int* ptr = malloc(sizeof(int) * na);
    int* uf = malloc(sizeof(int) * na);
    long nout = 0;
    long s = 0;
    while (s < na) {
        long e = s; while (e < na && chunk[e] == chunk[s]) e++;
        for (long x = s; x < e; x++) {
            double best = P->anchor_score; int bp = (int)x;
            for (long y = x - 1; y >= s && x - y <= P->band; y--) {
                if (A[y].rc != A[x].rc || A[y].rev != A[x].rev) continue;
                double dq = (double)A[x].qp - (double)A[y].qp;
                if (P->bp_band > 0 && dq > P->bp_band) break;
                double dr = A[x].rev ? (double)A[y].rp - (double)A[x].rp : (double)A[x].rp - (double)A[y].rp;
                if (P->require_mono && (dr <= 0 || dq <= 0)) continue;
                if (fabs(dr) > P->max_lin || dq > P->max_lin) continue;
                double gap = fabs(dq - dr);
                if (gap > P->max_gap) continue;
                double gc = P->gap_w > 0 ? P->gap_w * gap : (P->gapcost_mode == 0 ? gap : (P->gapcost_mode == 1 ? 0 : 0.5 * gap));
                double sc = f[y] + P->anchor_score - gc;
                if (sc > best) { best = sc; bp = (int)y; }
            }
            f[x] = best; ptr[x] = bp; uf[x] = (int)x;
        }
        for (long x = s; x < e; x++) if (ptr[x] != x) { int a = uf_find(uf, (int)x), b = uf_find(uf, ptr[x]); if (a != b) uf[a] = b; }
        // per set: best score index
        for (long x = s; x < e; x++) {
            int root = uf_find(uf, (int)x);
            (void)root;
        }
        // gather best per root
        // use arrays sized by chunk
        long n = e - s;
        int* besti = malloc(sizeof(int) * n); int* ssz = calloc(n, sizeof(int));
        for (long x = 0; x < n; x++) besti[x] = -1;
        for (long x = s; x < e; x++) {
            int root = uf_find(uf, (int)x) - (int)s; ssz[root]++;
            if (besti[root] < 0 || f[x] > f[besti[root]]) besti[root] = (int)x;
        }
        for (long rr = 0; rr < n; rr++) if (besti[rr] >= 0) {
            int cur = besti[rr]; int cnt = 1; uint32_t qmin = A[cur].qp, qmax = A[cur].qp, rmin = A[cur].rp, rmax = A[cur].rp;
            while (ptr[cur] != cur) { cur = ptr[cur]; cnt++; if (A[cur].qp < qmin) qmin = A[cur].qp; if (A[cur].qp > qmax) qmax = A[cur].qp; if (A[cur].rp < rmin) rmin = A[cur].rp; if (A[cur].rp > rmax) rmax = A[cur].rp; }
            if (nout < max_out) {
                interval_t* o = &out[nout++];
                o->chunk = chunk[s]; o->qc = A[besti[rr]].qc; o->q0 = qmin; o->q1 = qmax; o->rc = A[besti[rr]].rc; o->r0 = rmin; o->r1 = rmax;
                o->nanch = cnt; o->score = f[besti[rr]]; o->rev = A[besti[rr]].rev; o->setsize = ssz[rr]; o->nseeds = 0;
            }
        }
        free(besti); free(ssz);
        s = e;
    }
    free(f); free(ptr); free(uf); free(chunk); free(A); free(qs); free(rs);
    return nout;
}
