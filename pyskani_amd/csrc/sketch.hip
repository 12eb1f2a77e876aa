// FracMinHash sketching on gfx950 — replaces skani::seeding::fmh_seeds as driven by
// Database::_sketch (/root/reference/src/pyskani/_skani/lib.rs:140-185).
//
// Two kernels per batch of genomes:
//   sketch_scan  : ASCII -> 2-bit packed words (kept for the emit pass) + one seed bit per base
//                  (rolling canonical k-mer, mm_hash64, threshold) + per-tile seed counts
//   sketch_emit  : for every set bit rebuild the k-mer from the packed stream and write the seed
//                  record at its scanned offset; seeds whose hash also passes the marker threshold
//                  contribute their canonical 21-mer to the genome's marker list
// followed by rocPRIM plumbing (scan, segmented radix sorts) for the marker set and the
// k-mer-sorted reference index. Semantics are normative in oracle/skani_oracle.c.
#include "common.h"
#include <unordered_set>
#include <hipcub/hipcub.hpp>
#include <algorithm>

// ---- ASCII -> 2-bit, four bytes at a time; every byte that is not ACGT/acgt maps to 0 ----
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {  // 0x80 in each byte of v that is 0
    uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | v | 0x7F7F7F7Fu);
}
__device__ __forceinline__ uint32_t codes4(uint32_t w) {
    uint32_t x = w & 0xDFDFDFDFu;                 // fold case
    uint32_t code = (x >> 1) & 0x03030303u;       // A0 C1 G3 T2
    code ^= (code >> 1) & 0x01010101u;            // A0 C1 G2 T3
    uint32_t ok = zero_bytes(x ^ 0x43434343u) | zero_bytes(x ^ 0x47474747u) | zero_bytes(x ^ 0x54545454u);
    ok >>= 7;
    ok |= ok << 1;
    code &= ok;
    return (code * 0x40100401u) >> 24;            // c0<<6 | c1<<4 | c2<<2 | c3  (first base highest)
}
__device__ __forceinline__ uint32_t pack16_checked(uint4 v) {   // any byte, ~35 instructions per 4 bases
    return (codes4(v.x) << 24) | (codes4(v.y) << 16) | (codes4(v.z) << 8) | codes4(v.w);
}
// Fast path, 7 instructions per 4 bases: bits 1-3 of the case-folded byte index two 8-entry byte tables through
// v_perm_b32, one with the codes (A0 C1 T3 G2) and one with the letters themselves; a byte is ACGT/acgt exactly
// when the looked-up letter equals it. Any other byte in the 16 sends the lane to the checked path.
__device__ __forceinline__ uint32_t codes4_fast(uint32_t w, uint32_t& bad) {
    const uint32_t x = w & 0xDFDFDFDFu;
    const uint32_t idx = (x >> 1) & 0x07070707u;               // A 0, C 1, T 2, G 3
    bad |= __builtin_amdgcn_perm(0xFFFFFFFFu, 0x47544341u, idx) ^ x;
    return __builtin_amdgcn_perm(0u, 0x02030100u, idx) * 0x40100401u;   // top byte = c0<<6 | c1<<4 | c2<<2 | c3
}
__device__ __forceinline__ uint32_t pack16(uint4 v) {
    uint32_t bad = 0;
    const uint32_t px = codes4_fast(v.x, bad), py = codes4_fast(v.y, bad), pz = codes4_fast(v.z, bad), pw = codes4_fast(v.w, bad);
    uint32_t word = __builtin_amdgcn_perm(__builtin_amdgcn_perm(px, py, 0x07030000u), __builtin_amdgcn_perm(pz, pw, 0x00000703u), 0x07060100u);
    asm("" : "+v"(bad));     // keep the four tests one OR-ed word: the compiler otherwise splits them into compares
    if (__builtin_expect(bad != 0, 0)) word = pack16_checked(v);
    return word;
}

__device__ __forceinline__ int find_contig(const ContigDesc* __restrict__ contigs, int n_contigs, uint32_t tile) {
    int lo = 0, hi = n_contigs - 1;  // last contig with first_tile <= tile
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (contigs[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// per tile: contig id and a compact copy of what the emit pass needs {first_tile, genome, contig_index, contig id}
__global__ void tile_contig_kernel(const ContigDesc* __restrict__ contigs, int n_contigs, uint32_t n_tiles, uint32_t* __restrict__ tile_ci,
                                   uint4* __restrict__ tile_info) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int ci = find_contig(contigs, n_contigs, t);
    tile_ci[t] = (uint32_t)ci;
    tile_info[t] = make_uint4(contigs[ci].first_tile, contigs[ci].genome, contigs[ci].contig_index, (uint32_t)ci);
}

// reverse-complement of the 16 bases of a packed word
__device__ __forceinline__ uint32_t rc_word(uint32_t w) {
    uint32_t y = __brev(w);
    y = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
    return ~y;
}

// PACKED: the bases arrive 2-bit packed, tile by tile, in `packed` (the host ingest packs on its worker threads: a genome crosses PCIe as L / 4 bytes):
// phase 1 is a copy into LDS and `ascii` is not read
template <bool PACKED>
__device__ __forceinline__ void sketch_scan_body(
    const uint8_t* __restrict__ ascii, const ContigDesc* __restrict__ contigs, const uint32_t* __restrict__ tile_ci,
    uint32_t* __restrict__ packed, uint64_t* __restrict__ seedmask, uint32_t* __restrict__ tile_count,
    const SketchConsts& C) {
    __shared__ __align__(16) uint32_t s_w[4 + TILE_WORDS];
    __shared__ uint32_t s_cnt[TILE_THREADS / 64];
    const uint32_t tile = blockIdx.x;
    const int tid = threadIdx.x;
    const ContigDesc cd = contigs[tile_ci[tile]];
    const uint32_t pos0 = (tile - cd.first_tile) * TILE_BASES;
    const uint32_t n = min((uint32_t)TILE_BASES, cd.len - pos0);
    const uint8_t* src = ascii + cd.byte_off + pos0;

    // phase 1: pack 16 bases per lane per round, coalesced 16-byte loads
    if (PACKED) {      // four words per thread: one 16-byte load, two 8-byte LDS stores (the tile's words sit two words into s_w, behind the halo)
        const uint4 v = *reinterpret_cast<const uint4*>(packed + (size_t)tile * TILE_WORDS + 4 * tid);
        *reinterpret_cast<uint2*>(&s_w[2 + 4 * tid]) = make_uint2(v.x, v.y);
        *reinterpret_cast<uint2*>(&s_w[4 + 4 * tid]) = make_uint2(v.z, v.w);
    }
    else
#pragma unroll
    for (int r = 0; r < TILE_WORDS / TILE_THREADS; r++) {
        int w = tid + r * TILE_THREADS;
        uint32_t word = 0;
        if ((uint32_t)w * 16 < n) {
            uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)w * 16);
            word = pack16(v);
            uint32_t rem = n - (uint32_t)w * 16;          // bases of this word inside the contig
            if (rem < 16) word &= ~(0xFFFFFFFFu >> (2 * rem));
        }
        s_w[2 + w] = word;
        packed[(size_t)tile * TILE_WORDS + w] = word;
    }
    if (tid < 2) {
        uint32_t word = 0;
        if (pos0 > 0) word = PACKED ? packed[(size_t)tile * TILE_WORDS - 2 + tid] : pack16(*reinterpret_cast<const uint4*>(src - 32 + tid * 16));
        s_w[tid] = word;
    }
    __syncthreads();

    // phase 2: 64 window positions per lane, every k-mer a compile-time funnel shift of two pre-shifted streams.
    // Index space: base i of the lane's 64 sits at index 32+i of its 96-base neighbourhood W0..W5. The seed of
    // window position i starts at W-index 12+i+off_lo; delaying W by delta = 16-off_lo bases (6..14) puts that start
    // at index 28+i of the stream V, whatever k is.
    const uint4 wa = *reinterpret_cast<const uint4*>(&s_w[4 * tid]);       // 16-byte stride: conflict-free b128
    const uint2 wb = *reinterpret_cast<const uint2*>(&s_w[4 * tid + 4]);
    const uint32_t sh = 2 * C.delta;
    uint32_t V[7];
    V[0] = 0;
    V[1] = __funnelshift_r(wa.y, wa.x, sh);
    V[2] = __funnelshift_r(wa.z, wa.y, sh);
    V[3] = __funnelshift_r(wa.w, wa.z, sh);
    V[4] = __funnelshift_r(wb.x, wa.w, sh);
    V[5] = __funnelshift_r(wb.y, wb.x, sh);
    V[6] = __funnelshift_r(0u, wb.y, sh);
    // Q: reverse complement of V's 112 bases, delayed by k-1 bases: the reverse k-mer of position i STARTS at index 83-i
    const uint32_t rsh = 2 * (C.k - 1);
    uint32_t Q[7];
    {
        uint32_t r0 = rc_word(V[6]), r1 = rc_word(V[5]), r2 = rc_word(V[4]), r3 = rc_word(V[3]), r4 = rc_word(V[2]),
                 r5 = rc_word(V[1]), r6 = ~0u;   // V[0] is never read: its mirror only feeds junk bits
        Q[0] = 0;
        Q[1] = __funnelshift_r(r1, r0, rsh);
        Q[2] = __funnelshift_r(r2, r1, rsh);
        Q[3] = __funnelshift_r(r3, r2, rsh);
        Q[4] = __funnelshift_r(r4, r3, rsh);
        Q[5] = __funnelshift_r(r5, r4, rsh);
        Q[6] = __funnelshift_r(r6, r5, rsh);
    }
    // Both strands are taken LEFT-aligned (k-mer in the top 2k bits, following bases below): the smaller word holds
    // the canonical k-mer on top (equal k-mers give the same key either way), one shift right-aligns it.
    const uint32_t kdrop = 32 - 2 * C.k;
    uint32_t mlo = 0, mhi = 0;
    const uint64_t thr = C.thr;
    // a wave whose 4 096 bases lie wholly past the contig's end has nothing to hash (short contigs, last tiles)
    if (64u * (uint32_t)(tid & ~63) < n)
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const int s0 = 28 + i, sw = s0 / 16, so = s0 % 16;
        const uint32_t fl = so == 0 ? V[sw] : __funnelshift_r(V[sw + 1], V[sw], 32 - 2 * so);
        const int t0 = 83 - i, tw = t0 / 16, to = t0 % 16;
        const uint32_t rl = to == 0 ? Q[tw] : __funnelshift_r(Q[tw + 1], Q[tw], 32 - 2 * to);
        const uint32_t key = min(fl, rl) >> kdrop;
        const uint64_t h = mm_hash64_u32(key);
        // mask = mask*2 + (h < thr) as compare + add-with-carry (two instructions; the compiler's select + shift-add is
        // three): the first window position ends up in the HIGHEST bit; reversed below
        if (i < 32) asm("v_cmp_gt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mlo) : "s"(thr), "v"(h) : "vcc");
        else asm("v_cmp_gt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mhi) : "s"(thr), "v"(h) : "vcc");
    }
    mlo = __brev(mlo); mhi = __brev(mhi);
    // windows must lie inside the contig: K_MARKER-1 <= pos < len
    const uint32_t p0 = pos0 + 64u * tid;
    int lo_i = p0 >= (uint32_t)(K_MARKER - 1) ? 0 : (int)(K_MARKER - 1 - p0);
    int hi_i = cd.len > p0 ? (int)min(64u, cd.len - p0) : 0;
    uint64_t vm = 0;
    if (hi_i > lo_i) {
        uint64_t upto_hi = hi_i >= 64 ? ~0ull : ((1ull << hi_i) - 1);
        vm = upto_hi & ~((1ull << lo_i) - 1);
    }
    uint64_t m = (((uint64_t)mhi << 32) | mlo) & vm;
    seedmask[(size_t)tile * TILE_MASKS + tid] = m;
    int cnt = __popcll(m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) tile_count[tile] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

__global__ __launch_bounds__(TILE_THREADS) void sketch_scan_kernel(
    const uint8_t* __restrict__ ascii, const ContigDesc* __restrict__ contigs, const uint32_t* __restrict__ tile_ci,
    uint32_t* __restrict__ packed, uint64_t* __restrict__ seedmask, uint32_t* __restrict__ tile_count, SketchConsts C) {
    sketch_scan_body<false>(ascii, contigs, tile_ci, packed, seedmask, tile_count, C);
}
__global__ __launch_bounds__(TILE_THREADS) void sketch_scan_packed_kernel(
    const ContigDesc* __restrict__ contigs, const uint32_t* __restrict__ tile_ci,
    uint32_t* __restrict__ packed, uint64_t* __restrict__ seedmask, uint32_t* __restrict__ tile_count, SketchConsts C) {
    sketch_scan_body<true>(nullptr, contigs, tile_ci, packed, seedmask, tile_count, C);
}

// n (<= 32) bases starting at contig-relative base `start` of a packed stream, first base highest
__device__ __forceinline__ uint64_t get_bases(const uint32_t* __restrict__ words, uint32_t start, int n) {
    uint32_t first = start >> 4, off = start & 15;
    uint64_t top = ((uint64_t)words[first] << 32) | words[first + 1];
    uint64_t x = off ? ((top << (2 * off)) | ((uint64_t)words[first + 2] >> (32 - 2 * off))) : top;
    return x >> (64 - 2 * n);
}
__device__ __forceinline__ uint64_t revcomp(uint64_t x, int n) {
    uint64_t y = __brevll(x);
    y = ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
    y >>= (64 - 2 * n);
    return y ^ (n == 32 ? ~0ull : ((1ull << (2 * n)) - 1));
}

// One WAVE per tile, one SEED per lane. The wave ranks the tile's seed bits with a shuffle scan (no workgroup
// barrier anywhere), lists the seed positions in LDS, then every lane rebuilds one k-mer from the LDS-staged
// packed tile: all lanes do the same work, and a tile costs ~0.5 k wave-instructions instead of ~2.8 k.
constexpr int EMIT_WAVES = 4;      // tiles per workgroup (independent waves)
constexpr int EMIT_LIST = 1024;    // seed positions listed per pass

// SELF: the tiles' seed offsets are not given - every wave forms them from the tiles' counts (n_tiles <= 64: one count per lane, a shuffle scan), and
// the wave of tile 0 leaves them, the total and the contigs' first seeds for the kernels that follow (the one-launch-sequence query of a small genome)
template <bool SELF>
__device__ __forceinline__ void sketch_emit_body(
    const uint4* __restrict__ tile_info, const uint32_t* __restrict__ packed,
    const uint64_t* __restrict__ seedmask, const uint32_t* __restrict__ tile_off, uint32_t n_tiles,
    uint32_t* __restrict__ seed_kmer, uint32_t* __restrict__ seed_pos, uint32_t* __restrict__ seed_meta,
    uint64_t* __restrict__ seed_pm, uint64_t* __restrict__ marker_stage, uint32_t* __restrict__ tile_mcount,
    const SketchConsts& C, uint32_t seed_cap,
    const uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ toff_out, const uint32_t* __restrict__ cft, uint32_t n_desc, SmallQHead* __restrict__ head) {
    __shared__ __align__(16) uint32_t s_words_all[EMIT_WAVES][TILE_WORDS + 8];   // packed tile + 4 words either side
    __shared__ uint16_t s_list_all[EMIT_WAVES][EMIT_LIST];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * EMIT_WAVES + wv;
    if (tile >= n_tiles) return;
    uint32_t self_off = 0;
    if (SELF) {
        const uint32_t c = (uint32_t)lane < n_tiles ? tile_cnt[lane] : 0u;
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        const uint32_t all = __shfl(inc, 63);
        self_off = __shfl(inc - c, (int)tile);
        if (tile == 0) {
            if ((uint32_t)lane < n_tiles) toff_out[lane] = inc - c;
            if (lane == 0) { toff_out[n_tiles] = all; head->n_seeds = all; head->flags = all > seed_cap ? SQ_F_SEEDS : 0u; head->n_anchors = 0ull; head->n_short = 0u; head->n_hits = 0u; head->done = 0u; }      // (the status block starts from here: no memset)
            const uint32_t ft = (uint32_t)lane <= n_desc ? cft[lane] : 0u;
            const uint32_t ex_at = __shfl(inc - c, (int)(ft < 64u ? ft : 0u));      // (every lane takes part in the shuffle: a lane outside a branch has nothing to give)
            if ((uint32_t)lane <= n_desc) head->coff[lane] = ft < n_tiles ? ex_at : all;
        }
    }
    uint32_t* s_words = s_words_all[wv];
    uint16_t* s_list = s_list_all[wv];
    // four consecutive 64-base stripes per lane
    const ulonglong2* mp = reinterpret_cast<const ulonglong2*>(seedmask + (size_t)tile * TILE_MASKS + 4 * lane);
    const ulonglong2 ma = mp[0], mb = mp[1];
    const unsigned long long m[4] = {ma.x, ma.y, mb.x, mb.y};
    const uint4 ti = tile_info[tile];          // {first_tile, genome, contig_index, contig id}
    const uint32_t t_off = SELF ? self_off : tile_off[tile];
    const uint32_t cnt = (uint32_t)(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    const uint32_t total = __shfl(incl, 63);
    // (seed_cap: arrays that were sized before the count was known - the one-synchronisation path; a tile beyond them writes nothing, the host sees the total)
    if (total == 0 || t_off + total > seed_cap) { if (lane == 0) tile_mcount[tile] = 0; return; }
    const uint32_t excl = incl - cnt;
    uint32_t mrun = 0;   // markers of this tile so far: they go to marker_stage[t_off + 0 ..), no atomics
    const uint32_t pos0 = (tile - ti.x) * TILE_BASES;
    {   // stage the packed tile in LDS with coalesced 16-byte loads; k-mers are rebuilt from LDS
        const uint32_t* gsrc = packed + (size_t)tile * TILE_WORDS;
#pragma unroll
        for (int r = 0; r < TILE_WORDS / 256; r++)
            *reinterpret_cast<uint4*>(&s_words[4 + 4 * (lane + 64 * r)]) = *reinterpret_cast<const uint4*>(gsrc + 4 * (lane + 64 * r));
        if (lane < 4) s_words[lane] = tile > ti.x ? gsrc[lane - 4] : 0u;                     // halo: previous tile of the same contig
        if (lane >= 4 && lane < 8) s_words[TILE_WORDS + lane] = gsrc[TILE_WORDS + lane - 4]; // next words (slot or padding)
    }
    for (uint32_t base = 0; base < total; base += EMIT_LIST) {
        // list the positions (tile-relative) of seeds with rank in [base, base + EMIT_LIST)
        uint32_t off = excl;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            unsigned long long mm = m[r];
            while (mm) {
                int i = __ffsll(mm) - 1;
                mm &= mm - 1;
                if (off >= base && off < base + EMIT_LIST) s_list[off - base] = (uint16_t)((4 * lane + r) * 64 + i);
                off++;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t seg = total - base < (uint32_t)EMIT_LIST ? total - base : (uint32_t)EMIT_LIST;
        for (uint32_t j0 = 0; j0 < seg; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool have = j < seg;
            uint32_t p = 0; bool is_marker = false;
            if (have) {
                const uint32_t pl = s_list[j];                        // tile-relative; +64 keeps the halo index positive
                p = pos0 + pl;                                        // last base of the 21-base window
                uint64_t f = get_bases(s_words, pl + 64 + 1 - C.d - C.k, C.k);
                uint64_t rc = revcomp(f, C.k);
                uint32_t canon = f < rc;
                uint64_t cs = canon ? f : rc;
                uint32_t meta = (ti.z << 1) | canon;
                const uint32_t o = t_off + base + j;
                seed_kmer[o] = (uint32_t)cs;
                seed_pos[o] = p;
                seed_meta[o] = meta;
                seed_pm[o] = ((uint64_t)p << 32) | meta;
                is_marker = mm_hash64(cs) < C.thr_marker;
            }
            // marker slots are tile-local (a tile has at most as many markers as seeds): ballot + running count
            const unsigned long long bal = __ballot(is_marker);
            if (is_marker) {
                uint64_t f21 = get_bases(s_words, (p - pos0) + 64 + 1 - K_MARKER, K_MARKER);
                uint64_t r21 = revcomp(f21, K_MARKER);
                marker_stage[(size_t)t_off + mrun + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = f21 < r21 ? f21 : r21;
            }
            mrun += (uint32_t)__popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (lane == 0) tile_mcount[tile] = mrun;
}

__global__ __launch_bounds__(64 * EMIT_WAVES) void sketch_emit_kernel(
    const uint4* __restrict__ tile_info, const uint32_t* __restrict__ packed,
    const uint64_t* __restrict__ seedmask, const uint32_t* __restrict__ tile_off,
    const uint32_t* __restrict__ genome_seed_off, uint32_t n_tiles,
    uint32_t* __restrict__ seed_kmer, uint32_t* __restrict__ seed_pos, uint32_t* __restrict__ seed_meta,
    uint64_t* __restrict__ seed_pm, uint64_t* __restrict__ marker_stage, uint32_t* __restrict__ tile_mcount,
    SketchConsts C, uint32_t seed_cap) {
    sketch_emit_body<false>(tile_info, packed, seedmask, tile_off, n_tiles, seed_kmer, seed_pos, seed_meta, seed_pm, marker_stage, tile_mcount, C, seed_cap,
                            nullptr, nullptr, nullptr, 0u, nullptr);
}
__global__ __launch_bounds__(64 * EMIT_WAVES) void sketch_emit_self_kernel(
    const uint4* __restrict__ tile_info, const uint32_t* __restrict__ packed, const uint64_t* __restrict__ seedmask, uint32_t n_tiles,
    uint32_t* __restrict__ seed_kmer, uint32_t* __restrict__ seed_pos, uint32_t* __restrict__ seed_meta,
    uint64_t* __restrict__ seed_pm, uint64_t* __restrict__ marker_stage, uint32_t* __restrict__ tile_mcount,
    SketchConsts C, uint32_t seed_cap, const uint32_t* __restrict__ tile_cnt, uint32_t* __restrict__ toff_out, const uint32_t* __restrict__ cft, uint32_t n_desc,
    SmallQHead* __restrict__ head) {
    sketch_emit_body<true>(tile_info, packed, seedmask, nullptr, n_tiles, seed_kmer, seed_pos, seed_meta, seed_pm, marker_stage, tile_mcount, C, seed_cap,
                           tile_cnt, toff_out, cft, n_desc, head);
}

// dense per-genome marker staging: tile t's markers move from marker_stage[tile_off[t] ..) to dense[tile_moff[t] ..)
// tile_info != null: every marker is tagged with its genome's number above the 2 x K_MARKER marker bits, so that ONE device radix
// sort orders all genomes' markers at once (segments stay where they are: the tag is the sort's leading digit)
constexpr int MARKER_BITS = 2 * K_MARKER;
__global__ __launch_bounds__(256) void marker_compact_kernel(const uint64_t* __restrict__ stage, const uint32_t* __restrict__ tile_off,
                                                             const uint32_t* __restrict__ tile_moff, uint32_t n_tiles, uint64_t* __restrict__ dense,
                                                             const uint4* __restrict__ tile_info) {
    const uint32_t tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const uint32_t src = tile_off[tile], dst = tile_moff[tile], n = tile_moff[tile + 1] - dst;
    const uint64_t tag = tile_info ? (uint64_t)tile_info[tile].y << MARKER_BITS : 0ull;
    for (uint32_t i = threadIdx.x & 63; i < n; i += 64) dense[dst + i] = stage[src + i] | tag;
}

// out[i] = tile_off[idx[i]]
__global__ void gather_u32_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, uint32_t* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}
// one block per genome: write the distinct values of the sorted segment, in order, to `uniq` at the same
// segment offset; cnt[g] = number of distinct markers (cnt[n_genomes] must be pre-zeroed by the caller's scan input)
// entries past the batch's last marker (the array is sized by the SEED count, its upper bound) get a tag beyond every genome's: they
// sort to the end and no segment reads them
// (n: the number of entries the sort takes - the raw markers' expected count with a wide margin, not the seed count; more raw markers than that raise *over
// and the caller sorts again at full size)
__global__ __launch_bounds__(256) void marker_pad_kernel(uint64_t* __restrict__ dense, const uint32_t* __restrict__ total, uint32_t n, uint64_t sentinel, uint32_t* __restrict__ over) {
    const uint32_t i = *total + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dense[i] = sentinel;
    if (blockIdx.x == 0 && threadIdx.x == 0 && *total > n) *over = 1u;
}
__global__ __launch_bounds__(256) void marker_unique_kernel(const uint64_t* __restrict__ sorted, uint64_t* __restrict__ uniq,
                                                             const uint32_t* __restrict__ beg, const uint32_t* __restrict__ end,
                                                             uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_w[4], s_base;
    const uint32_t b = beg[blockIdx.x], e = end[blockIdx.x];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (uint32_t i0 = b; i0 < e; i0 += 256) {
        uint32_t i = i0 + threadIdx.x;
        bool f = i < e && (i == b || sorted[i] != sorted[i - 1]);      // (tagged or not: equal within a genome's segment means equal markers)
        unsigned long long bal = __ballot(f);
        if (lane == 0) s_w[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = s_base;
        for (int w = 0; w < wv; w++) off += s_w[w];
        if (f) uniq[b + off + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = sorted[i] & ((1ull << MARKER_BITS) - 1ull);
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_base;
}
template <int NT> __device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave, uint32_t* total);      // (defined below)
// The same for FEW LARGE genomes (3 M markers per 3 Gb genome, one genome per call when the ASCII is streamed): a workgroup per genome walks its segment 256 markers at a
// time with three barriers per step - 9.8 ms per launch whatever the number of genomes, 0.5 s of a 50-genome pass. Here a genome's segment is cut into gridDim.x slices:
// pass 0 counts the distinct markers of every slice, marker_slice_scan_kernel turns the counts into offsets, pass 1 writes.
template <int WRITE>
__global__ __launch_bounds__(256) void marker_unique_sliced_kernel(const uint64_t* __restrict__ sorted, uint64_t* __restrict__ uniq,
                                                                    const uint32_t* __restrict__ beg, const uint32_t* __restrict__ end,
                                                                    uint32_t* __restrict__ slice_cnt) {
    __shared__ uint32_t s_w[4], s_base;
    const uint32_t g = blockIdx.y, S = gridDim.x;
    const uint32_t b = beg[g], e = end[g], len = e - b;
    const uint32_t per = ((len + S - 1) / S + 255u) & ~255u;
    const uint32_t sb = b + blockIdx.x * per < e ? b + blockIdx.x * per : e, se = sb + per < e ? sb + per : e;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = WRITE ? slice_cnt[g * S + blockIdx.x] : 0u;      // pass 1: the slice's offset in the genome's distinct list
    __syncthreads();
    for (uint32_t i0 = sb; i0 < se; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const bool f = i < se && (i == b || sorted[i] != sorted[i - 1]);
        const unsigned long long bal = __ballot(f);
        if (lane == 0) s_w[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (WRITE) {
            uint32_t off = s_base;
            for (int w = 0; w < wv; w++) off += s_w[w];
            if (f) uniq[b + off + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = sorted[i] & ((1ull << MARKER_BITS) - 1ull);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (!WRITE && threadIdx.x == 0) slice_cnt[g * S + blockIdx.x] = s_base;
}
// one workgroup per genome: exclusive scan of its S slice counts in place, the total to cnt[g] (S <= 1024)
__global__ __launch_bounds__(1024) void marker_slice_scan_kernel(uint32_t* __restrict__ slice_cnt, uint32_t S, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_part[1024 / 64 + 1];
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    const uint32_t c = t < S ? slice_cnt[g * S + t] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<1024>(c, s_part, &total);
    if (t < S) slice_cnt[g * S + t] = ex;
    if (t == 0) cnt[g] = total;
}
// moff = exclusive scan of the distinct counts: copy each genome's distinct markers to its dense slot (gridDim.y slices per genome)
__global__ void marker_copy_kernel(const uint64_t* __restrict__ uniq, const uint32_t* __restrict__ beg,
                                   const uint32_t* __restrict__ moff, uint64_t* __restrict__ out) {
    uint32_t b = beg[blockIdx.x], o = moff[blockIdx.x], n = moff[blockIdx.x + 1] - o;
    for (uint32_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += gridDim.y * blockDim.x) out[o + i] = uniq[b + i];
}



// LDS hand-off between lanes of ONE wave: order the ds ops, no workgroup barrier
__device__ __forceinline__ void lds_wave_handoff() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// exclusive prefix of one value per thread over a workgroup of NT threads (two barriers); *total = sum over the workgroup
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave /* NT/64 + 1 words */, uint32_t* total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(incl, o); if ((int)lane >= o) incl += u; }
    __syncthreads();                       // s_wave may still be read from an earlier call
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) { const uint32_t c = s_wave[w]; if ((uint32_t)w < wv) base += c; tot += c; }
    *total = tot;
    return base + incl - v;
}

// ---- marker set of one genome by ONE workgroup: gather the tile-local lists into LDS, bitonic sort, distinct values
// out. Replaces compaction + segmented radix sort + distinct pass (0.30 ms per 1 000 genomes) for genomes with up
// to MB_CAP raw markers (8 Mb at marker_c = 1000); a larger genome raises `overflow` and the batch is redone on the
// segmented-sort path.
constexpr int MB_CAP = 8192;
constexpr int MB_THREADS = 1024;
constexpr int MB_TILES = 1024;     // tiles of one genome the LDS tile table holds (16.7 Mb)
// ONE (the one-synchronisation path of a single small genome): tile_moff holds the tiles' raw marker COUNTS - the workgroup scans them itself -
// the distinct markers go straight to the sketch's array of out_cap entries, and cnt receives {0, distinct} (the marker offsets).
template <bool ONE>
__global__ __launch_bounds__(MB_THREADS) void marker_block_kernel(const uint64_t* __restrict__ stage, const uint32_t* __restrict__ tile_off,
                                                                  const uint32_t* __restrict__ tile_moff, const uint32_t* __restrict__ gft,
                                                                  uint64_t* __restrict__ uniq, uint32_t* __restrict__ cnt, uint32_t* __restrict__ overflow, uint32_t out_cap) {
    __shared__ unsigned long long s_m[MB_CAP];
    __shared__ uint32_t s_cnt[MB_CAP];
    __shared__ uint32_t s_tsrc[MB_TILES], s_tdst[MB_TILES + 1];
    __shared__ uint32_t s_wave[MB_THREADS / 64 + 1];
    const uint32_t g = blockIdx.x, tid = threadIdx.x;
    const uint32_t t0 = gft[g], nt = gft[g + 1] - t0;
    if (nt > (uint32_t)MB_TILES) { if (tid == 0) { atomicOr(overflow, 1u); cnt[g] = 0; if (ONE) cnt[1] = 0; } return; }
    uint32_t m0, n;
    if (ONE) {
        const uint32_t c = tid < nt ? tile_moff[t0 + tid] : 0u;
        const uint32_t ex = block_exclusive_scan<MB_THREADS>(c, s_wave, &n);
        if (tid < nt) { s_tsrc[tid] = tile_off[t0 + tid]; s_tdst[tid] = ex; }
        m0 = 0;
    } else { m0 = tile_moff[t0]; n = tile_moff[t0 + nt] - m0; }
    if (n > (uint32_t)MB_CAP) { if (tid == 0) { atomicOr(overflow, 1u); cnt[g] = 0; if (ONE) cnt[1] = 0; } return; }
    // tile table into LDS with one round of loads, then the markers with at most MB_CAP / MB_THREADS independent loads
    if (!ONE) for (uint32_t q = tid; q < nt; q += MB_THREADS) { s_tsrc[q] = tile_off[t0 + q]; s_tdst[q] = tile_moff[t0 + q] - m0; }
    if (tid == 0) s_tdst[nt] = n;
    __syncthreads();
    constexpr int PER = MB_CAP / MB_THREADS;
    unsigned long long v[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const uint32_t d = tid + r * MB_THREADS;
        v[r] = ~0ull;
        if (d < n) {
            uint32_t lo = 0, hi = nt;                             // last tile q with s_tdst[q] <= d
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_tdst[mid] <= d) lo = mid; else hi = mid; }
            v[r] = stage[s_tsrc[lo] + (d - s_tdst[lo])];
        }
    }
    // Counting sort on the marker's top 13 bits (hash-selected 21-mers are uniform: <= 1 marker per bucket on
    // average), then every bucket is ordered in place. A bitonic network over the same 8 192 slots was bound by LDS
    // bandwidth (91 stages x 128 kB); this touches each marker a handful of times.
    constexpr int MB_SHIFT = 2 * K_MARKER - 13;
#pragma unroll
    for (int r = 0; r < PER; r++) s_cnt[tid * PER + r] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; r++) if (tid + r * MB_THREADS < n) atomicAdd(&s_cnt[(uint32_t)(v[r] >> MB_SHIFT)], 1u);
    __syncthreads();
    uint32_t c8[PER], sum = 0;
#pragma unroll
    for (int r = 0; r < PER; r++) { c8[r] = s_cnt[tid * PER + r]; sum += c8[r]; }
    uint32_t tot_raw;
    uint32_t run = block_exclusive_scan<MB_THREADS>(sum, s_wave, &tot_raw);
#pragma unroll
    for (int r = 0; r < PER; r++) { s_cnt[tid * PER + r] = run; run += c8[r]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PER; r++) if (tid + r * MB_THREADS < n) s_m[atomicAdd(&s_cnt[(uint32_t)(v[r] >> MB_SHIFT)], 1u)] = v[r];
    __syncthreads();
    // the counters are now the bucket ENDS; thread t orders buckets 8t .. 8t+7
#pragma unroll 1
    for (int r = 0; r < PER; r++) {
        const uint32_t b = tid * PER + r;
        const uint32_t lo = b ? s_cnt[b - 1] : 0, m = s_cnt[b] - lo;
        if (m < 2) continue;
        unsigned long long* a = s_m + lo;
        for (uint32_t x = 1; x < m; x++) { const unsigned long long w = a[x]; uint32_t y = x; while (y > 0 && a[y - 1] > w) { a[y] = a[y - 1]; y--; } a[y] = w; }
    }
    __syncthreads();
    // distinct values: each thread looks at 8 consecutive elements, one workgroup scan ranks them
    unsigned long long e[PER + 1];
    const uint32_t i0 = tid * PER;
    e[0] = i0 ? s_m[i0 - 1] : 0;
#pragma unroll
    for (int r = 0; r < PER; r++) e[r + 1] = s_m[i0 + r];
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < PER; r++) mine += (i0 + r < n && (i0 + r == 0 || e[r + 1] != e[r])) ? 1u : 0u;
    uint32_t total;
    uint32_t rank = block_exclusive_scan<MB_THREADS>(mine, s_wave, &total);
    if (ONE && total > out_cap) { if (tid == 0) { atomicOr(overflow, 2u); cnt[0] = 0; cnt[1] = 0; } return; }
#pragma unroll
    for (int r = 0; r < PER; r++) if (i0 + r < n && (i0 + r == 0 || e[r + 1] != e[r])) uniq[m0 + rank++] = e[r + 1];
    if (tid == 0) { if (ONE) { cnt[0] = 0; cnt[1] = total; } else cnt[g] = total; }
}

// One workgroup: exclusive scan of the tiles' seed counts (tile offsets), the 64-bit total, the genome's and the contigs' seed offsets (res: {total lo, total hi,
// goff[0], goff[1]}, coff -> the sketch's contig_seed_start AND the result block), the marker overflow flag cleared. Replaces two device scans, a reduction, two gathers
// and three memsets of the general path when ONE small genome is sketched.
__global__ __launch_bounds__(1024) void sketch_small_offsets_kernel(const uint32_t* __restrict__ cnt, uint32_t n_tiles, uint32_t* __restrict__ toff, const uint32_t* __restrict__ cft, uint32_t n_desc,
                                                                   uint32_t* __restrict__ coff_store, uint32_t* __restrict__ res, uint32_t* __restrict__ res_coff, uint32_t* __restrict__ mflag) {
    __shared__ uint32_t s_wave[1024 / 64 + 1];
    __shared__ uint32_t s_carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_tiles; i0 += 1024) {
        const uint32_t i = i0 + tid, c = i < n_tiles ? cnt[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan<1024>(c, s_wave, &tot);
        const uint32_t carry = s_carry;
        if (i < n_tiles) toff[i] = carry + ex;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    const uint32_t total = s_carry;
    if (tid == 0) { toff[n_tiles] = total; res[0] = total; res[1] = 0; res[2] = 0; res[3] = total; *mflag = 0; }
    __threadfence_block();
    __syncthreads();
    for (uint32_t c = tid; c <= n_desc; c += 1024) { const uint32_t v = c < n_desc ? toff[cft[c]] : total; coff_store[c] = v; res_coff[c] = v; }
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static SketchConsts make_sketch_consts(const psk_params* p) {
    SketchConsts C{};
    C.k = p->k; C.thr = UINT64_MAX / (uint64_t)p->c; C.thr_marker = UINT64_MAX / (uint64_t)p->marker_c;
    C.kmask = p->k == 16 ? 0xFFFFFFFFu : ((1u << (2 * p->k)) - 1u);
    C.rshift = 2 * p->k - 2;
    C.d = K_MARKER - p->k - (K_MARKER - p->k) / 2;
    C.delta = 16 - (K_MARKER - p->k) / 2;
    return C;
}

// The sketch kernels of the one-launch-sequence query (small_query.hip): the tile -> contig tables come from the host with the bases (ONE upload),
// sketch_scan runs as ever, the emit waves form their own offsets. Seeds land in (contig, position) order in A.seed_*, the tiles' raw markers in
// A.mstage at their tiles' seed offsets with the counts in A.d_tmc (the screen workgroup gathers and sorts them), totals in A.head. Nothing is waited for.
psk_status small_query_sketch_enqueue(Lane* ctx, const psk_params* p, const SmallQSketch& A, hipStream_t st) {
    if (A.n_tiles == 0 || A.n_tiles > SQ_MAX_TILES || A.n_desc == 0 || A.n_desc > SQ_MAX_DESC) { psk_set_error("internal: small query beyond its capacities"); return PSK_EINVAL; }
    const SketchConsts C = make_sketch_consts(p);
    ctx->t_begin(K_SKETCH_SCAN, st);
    hipLaunchKernelGGL(sketch_scan_kernel, dim3(A.n_tiles), dim3(TILE_THREADS), 0, st, A.d_bases, A.d_desc, A.d_tci, A.d_packed, A.d_mask, A.d_cnt, C);
    ctx->t_end(st);
    ctx->t_begin(K_SKETCH_EMIT, st);
    hipLaunchKernelGGL(sketch_emit_self_kernel, dim3((A.n_tiles + EMIT_WAVES - 1) / EMIT_WAVES), dim3(64 * EMIT_WAVES), 0, st, A.d_tinfo, (const uint32_t*)A.d_packed, (const uint64_t*)A.d_mask, A.n_tiles,
                       A.seed_kmer, A.seed_pos, A.seed_meta, A.seed_pm, A.mstage, A.d_tmc, C, SQ_SEEDS, (const uint32_t*)A.d_cnt, A.d_toff, A.d_cft, A.n_desc, A.head);
    ctx->t_end(st);
    return PSK_OK;
}
struct ToU64 { __host__ __device__ unsigned long long operator()(uint32_t v) const { return v; } };

// One sub-batch of genomes moving through the sketch pipeline on its own stream. The phases are split
// at the two points where the host must learn a size (total seeds, total distinct markers).
struct SketchJob {
    Lane* ctx; Lane::JobRes* R; hipStream_t st;
    const psk_params* p; const uint8_t* d_bases; int want_seeds;
    const uint32_t* packed_in = nullptr;      // the bases arrive 2-bit packed, tile by tile in this job's tile order (host ingest): d_bases is not read
    uint32_t n_genomes = 0, n_tiles = 0; int n_desc = 0;
    std::vector<ContigDesc> descs;
    std::vector<uint32_t> g_first_desc, g_first_tile;
    std::vector<uint64_t> g_total_len;      // kept bases per genome (the sketch objects are made later: make_objects)
    std::vector<psk_sketch*> sk;
    SketchConsts C{};
    std::shared_ptr<SketchStore> store;
    uint32_t *d_tmc = nullptr, *d_tmoff = nullptr, *d_cnt = nullptr, *d_toff = nullptr, *d_gft = nullptr, *d_cft = nullptr, *d_goff = nullptr, *d_coff = nullptr,
             *d_mcnt = nullptr, *d_sbeg = nullptr, *d_send = nullptr, *d_moff = nullptr, *d_tci = nullptr, *d_packed = nullptr;
    uint4* d_tinfo = nullptr; ContigDesc* d_desc = nullptr; uint64_t* d_mask = nullptr;
    uint64_t *d_mstage = nullptr, *d_msorted = nullptr, *d_mdense = nullptr;
    uint32_t *h_goff = nullptr, *h_coff = nullptr, *h_moff = nullptr;
    unsigned long long *d_total64 = nullptr, *h_total64 = nullptr;
    bool empty = false;

    void drop() { for (auto* x : sk) delete x; sk.clear(); }

#define JHIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { psk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); return _e == hipErrorOutOfMemory ? PSK_ENOMEM : PSK_EHIP; } } while (0)

    // host: contig filter (lib.rs:156) and tile layout
    psk_status prepare(const uint64_t* contig_off, const uint64_t* contig_len, const uint32_t* gfc, uint32_t ng) {
        n_genomes = ng;
        g_first_desc.resize(ng + 1); g_first_tile.resize(ng + 1); sk.assign(ng, nullptr); g_total_len.assign(ng, 0);
        uint64_t n_tiles64 = 0, total_bases = 0;
        for (uint32_t g = 0; g < ng; g++) {
            g_first_desc[g] = (uint32_t)descs.size();
            g_first_tile[g] = (uint32_t)n_tiles64;
            uint32_t kept = 0;
            for (uint32_t ci = gfc[g]; ci < gfc[g + 1]; ci++) {
                uint64_t len = contig_len[ci];
                if (len < MIN_LENGTH_CONTIG) continue;
                if (len > 0xFFFFFFFFull) { psk_set_error("contig longer than 2^32-1 bases"); return PSK_ELIMIT; }
                if (!packed_in && (contig_off[ci] & 15)) { psk_set_error("contig offset not 16-byte aligned"); return PSK_EINVAL; }
                ContigDesc d{};
                d.byte_off = contig_off[ci]; d.len = (uint32_t)len; d.first_tile = (uint32_t)n_tiles64;
                d.genome = g; d.contig_index = kept++;
                descs.push_back(d);
                g_total_len[g] += len;
                n_tiles64 += (len + TILE_BASES - 1) / TILE_BASES;
                total_bases += len;
            }
        }
        g_first_desc[ng] = (uint32_t)descs.size();
        g_first_tile[ng] = (uint32_t)n_tiles64;
        if (n_tiles64 >= (1ull << 31)) { psk_set_error("batch of %llu bases exceeds the per-launch tile limit; split it", (unsigned long long)total_bases); return PSK_ELIMIT; }
        n_tiles = (uint32_t)n_tiles64; n_desc = (int)descs.size();
        empty = n_tiles == 0;
        C = make_sketch_consts(p);
        return PSK_OK;
    }

    // the sketch objects of the batch: made AFTER phase1 has its kernels in flight (100 000 contigs per metagenome step: 4-5 ms of allocations that used to
    // run before the first launch, with the GPU idle)
    void make_objects() {
        for (uint32_t g = 0; g < n_genomes; g++) {
            if (sk[g]) continue;      // (the single-genome path made its object up front)
            psk_sketch* s = new psk_sketch();
            s->ctx = ctx->dev; s->params = *p; s->has_seeds = want_seeds != 0;
            const uint32_t d0 = g_first_desc[g], d1 = g_first_desc[g + 1];
            for (uint32_t d = d0; d < d1; d++) s->contig_len.push_back(descs[d].len);
            s->total_len = g_total_len[g];
            sk[g] = s;
        }
    }

    // tables up, pack + seed bits, tile offsets, per-genome / per-contig seed offsets on their way back
    psk_status phase1(hipEvent_t wait_scan) {
        if (empty) return PSK_OK;
        size_t o_cnt = 0, o_toff = o_cnt + n_tiles + 1, o_gft = o_toff + n_tiles + 1, o_cft = o_gft + n_genomes + 1,
               o_goff = o_cft + n_desc + 1, o_coff = o_goff + n_genomes + 1, o_mcnt = o_coff + n_desc + 1,
               o_sbeg = o_mcnt + n_genomes, o_send = o_sbeg + n_genomes + 1, o_moff = o_send + n_genomes, o_tmc = o_moff + n_genomes + 1,
               o_tmoff = o_tmc + n_tiles + 1, o_end = o_tmoff + n_tiles + 1;
        PSK_TRY(R->s_desc.reserve(sizeof(ContigDesc) * n_desc));
        if (!packed_in) PSK_TRY(R->s_packed.reserve(sizeof(uint32_t) * ((size_t)n_tiles * TILE_WORDS + 8)));
        PSK_TRY(R->s_mask.reserve(sizeof(uint64_t) * (size_t)n_tiles * TILE_MASKS));
        size_t red_tmp = 0;
        hipcub::TransformInputIterator<unsigned long long, ToU64, const uint32_t*> it64_probe((const uint32_t*)nullptr, ToU64());
        JHIP(hipcub::DeviceReduce::Sum(nullptr, red_tmp, it64_probe, (unsigned long long*)nullptr, (int)n_tiles, st));
        const size_t o_tot = align_up(sizeof(uint32_t) * o_end, 256);
        PSK_TRY(R->s_offs.reserve(o_tot + 256 + red_tmp));
        PSK_TRY(R->s_counts.reserve(sizeof(uint4) * ((size_t)n_tiles + 1) + sizeof(uint32_t) * ((size_t)n_tiles + 4)));
        uint32_t* d_offs = (uint32_t*)R->s_offs.p;
        d_total64 = (unsigned long long*)((char*)R->s_offs.p + o_tot);
        d_cnt = d_offs + o_cnt; d_toff = d_offs + o_toff; d_gft = d_offs + o_gft; d_cft = d_offs + o_cft;
        d_goff = d_offs + o_goff; d_coff = d_offs + o_coff; d_mcnt = d_offs + o_mcnt; d_sbeg = d_offs + o_sbeg;
        d_send = d_offs + o_send; d_moff = d_offs + o_moff; d_tmc = d_offs + o_tmc; d_tmoff = d_offs + o_tmoff;
        d_tinfo = (uint4*)R->s_counts.p; d_tci = (uint32_t*)(d_tinfo + n_tiles + 1);
        d_desc = (ContigDesc*)R->s_desc.p; d_packed = packed_in ? const_cast<uint32_t*>(packed_in) : (uint32_t*)R->s_packed.p; d_mask = (uint64_t*)R->s_mask.p;
        size_t hbytes = sizeof(ContigDesc) * n_desc + sizeof(uint32_t) * (n_genomes + 1 + n_desc + 1);
        void* hp;
        PSK_TRY(R->pin(hbytes + sizeof(uint32_t) * (2 * (n_genomes + 1) + n_desc + 1 + 4) + 16, &hp));
        ContigDesc* h_desc = (ContigDesc*)hp;
        uint32_t* h_gft = (uint32_t*)(h_desc + n_desc);
        uint32_t* h_cft = h_gft + n_genomes + 1;
        h_goff = h_cft + n_desc + 1; h_coff = h_goff + n_genomes + 1; h_moff = h_coff + n_desc + 1;
        h_total64 = (unsigned long long*)(((uintptr_t)(h_moff + n_genomes + 1 + 4) + 7) & ~(uintptr_t)7);
        memcpy(h_desc, descs.data(), sizeof(ContigDesc) * n_desc);
        memcpy(h_gft, g_first_tile.data(), sizeof(uint32_t) * (n_genomes + 1));
        for (int i = 0; i < n_desc; i++) h_cft[i] = descs[i].first_tile;
        h_cft[n_desc] = n_tiles;
        JHIP(hipMemcpyAsync(d_desc, h_desc, sizeof(ContigDesc) * n_desc, hipMemcpyHostToDevice, st));
        JHIP(hipMemcpyAsync(d_gft, h_gft, sizeof(uint32_t) * (n_genomes + 1), hipMemcpyHostToDevice, st));
        JHIP(hipMemcpyAsync(d_cft, h_cft, sizeof(uint32_t) * (n_desc + 1), hipMemcpyHostToDevice, st));
        JHIP(hipMemsetAsync(d_cnt + n_tiles, 0, sizeof(uint32_t), st));
        JHIP(hipMemsetAsync(d_tmc + n_tiles, 0, sizeof(uint32_t), st));
        JHIP(hipMemsetAsync(d_moff + n_genomes, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(tile_contig_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, st, d_desc, n_desc, n_tiles, d_tci, d_tinfo);
        if (wait_scan) JHIP(hipStreamWaitEvent(st, wait_scan, 0));   // scans run one after another; emit/sorts fill in beside them
        ctx->t_begin(K_SKETCH_SCAN, st);
        // optional dynamic-LDS ballast caps the scan's residency so that the latency-bound emit/sort kernels of the
        // previous sub-batch find free wave slots beside it (the scan is issue-bound well below 8 waves/SIMD)
        static const int scan_lds = getenv("PSK_SCAN_LDS") ? atoi(getenv("PSK_SCAN_LDS")) : 0;
        if (packed_in) hipLaunchKernelGGL(sketch_scan_packed_kernel, dim3(n_tiles), dim3(TILE_THREADS), scan_lds, st, d_desc, d_tci, d_packed, d_mask, d_cnt, C);
        else hipLaunchKernelGGL(sketch_scan_kernel, dim3(n_tiles), dim3(TILE_THREADS), scan_lds, st, d_bases, d_desc, d_tci, d_packed, d_mask, d_cnt, C);
        ctx->t_end(st);
        JHIP(hipEventRecord(R->scan_done, st));
        size_t tmp_bytes = 0;
        JHIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_cnt, d_toff, (int)(n_tiles + 1), st));
        PSK_TRY(R->s_tmp.reserve(tmp_bytes));
        JHIP(hipcub::DeviceScan::ExclusiveSum(R->s_tmp.p, tmp_bytes, d_cnt, d_toff, (int)(n_tiles + 1), st));
        {   // the offsets are 32-bit: a 64-bit total of the same counts tells phase2 whether they wrapped (low-complexity
            // input can make nearly every base a seed)
            hipcub::TransformInputIterator<unsigned long long, ToU64, const uint32_t*> it64(d_cnt, ToU64());
            size_t tb = red_tmp;
            JHIP(hipcub::DeviceReduce::Sum((char*)R->s_offs.p + o_tot + 256, tb, it64, d_total64, (int)n_tiles, st));
            JHIP(hipMemcpyAsync(h_total64, d_total64, 8, hipMemcpyDeviceToHost, st));
        }
        int n1 = n_genomes + 1, n2 = n_desc + 1;
        hipLaunchKernelGGL(gather_u32_kernel, dim3((n1 + 255) / 256), dim3(256), 0, st, d_toff, d_gft, d_goff, n1);
        hipLaunchKernelGGL(gather_u32_kernel, dim3((n2 + 255) / 256), dim3(256), 0, st, d_toff, d_cft, d_coff, n2);
        JHIP(hipMemcpyAsync(h_goff, d_goff, sizeof(uint32_t) * (n_genomes + 1), hipMemcpyDeviceToHost, st));
        JHIP(hipMemcpyAsync(h_coff, d_coff, sizeof(uint32_t) * (n_desc + 1), hipMemcpyDeviceToHost, st));
        return PSK_OK;
    }

    // total seeds known: allocate the store, emit seed records, sort + unique the marker sets
    psk_status phase2() {
        if (empty) return PSK_OK;
        JHIP(hipStreamSynchronize(st));
        const uint32_t total_seeds = h_goff[n_genomes];
        if (*h_total64 >= 0x7FFFFFF0ull) { psk_set_error("batch yields %llu seeds (>= 2^31 per launch); split it", (unsigned long long)*h_total64); return PSK_ELIMIT; }
        store = std::make_shared<SketchStore>();
        size_t ns = total_seeds;
        size_t b_kmer = 0, b_pos = align_up(b_kmer + 4 * ns, 256), b_meta = align_up(b_pos + 4 * ns, 256),
               b_pm = align_up(b_meta + 4 * ns, 256), b_cstart = align_up(b_pm + 8 * ns, 256), b_end = align_up(b_cstart + 4 * (size_t)(n_desc + 1), 256);
        store->ctx = ctx->dev;
        PSK_TRY(ctx->pool_alloc(b_end, &store->base, &store->bytes));
        char* sb = (char*)store->base;
        store->seed_kmer = (uint32_t*)(sb + b_kmer); store->seed_pos = (uint32_t*)(sb + b_pos); store->seed_meta = (uint32_t*)(sb + b_meta);
        store->seed_pm = (uint64_t*)(sb + b_pm); store->contig_seed_start = (uint32_t*)(sb + b_cstart);
        JHIP(hipMemcpyAsync(store->contig_seed_start, d_coff, sizeof(uint32_t) * (n_desc + 1), hipMemcpyDeviceToDevice, st));
        PSK_TRY(R->s_mark.reserve(sizeof(uint64_t) * (3 * ns + 3)));   // tile-local stage, dense, sorted
        d_mstage = (uint64_t*)R->s_mark.p; d_mdense = d_mstage + ns + 1; d_msorted = d_mdense + ns + 1;
        marker_dense_ready = false; marker_cap_ok = true; marker_capped = false;      // (a job object serves one batch after another)
        ctx->t_begin(K_SKETCH_EMIT, st);
        hipLaunchKernelGGL(sketch_emit_kernel, dim3((n_tiles + EMIT_WAVES - 1) / EMIT_WAVES), dim3(64 * EMIT_WAVES), 0, st, d_tinfo, d_packed, d_mask, d_toff, d_goff, n_tiles,
                           store->seed_kmer, store->seed_pos, store->seed_meta, store->seed_pm, d_mstage, d_tmc, C, 0xFFFFFFFFu);
        ctx->t_end(st);
        ctx->t_begin(K_SKETCH_SORT, st);
        // marker sets: tile-local lists -> dense per-genome segments -> per-genome sort -> distinct values
        size_t tmp_bytes = 0;
        JHIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_tmc, d_tmoff, (int)(n_tiles + 1), st));
        PSK_TRY(R->s_tmp.reserve(tmp_bytes));
        JHIP(hipcub::DeviceScan::ExclusiveSum(R->s_tmp.p, tmp_bytes, d_tmc, d_tmoff, (int)(n_tiles + 1), st));
        hipLaunchKernelGGL(gather_u32_kernel, dim3((n_genomes + 1 + 255) / 256), dim3(256), 0, st, d_tmoff, d_gft, d_sbeg, (int)(n_genomes + 1));   // segment g = [sbeg[g], sbeg[g+1])
        // small genomes (expected raw markers well inside MB_CAP): one workgroup per genome does it all in LDS
        marker_block = getenv("PSK_MARKER_SEGSORT") == nullptr;
        for (uint32_t g = 0; g < n_genomes && marker_block; g++)
            if (g_total_len[g] / (uint64_t)p->marker_c > (uint64_t)(MB_CAP * 3 / 4)) marker_block = false;
        if (marker_block) {
            JHIP(hipMemsetAsync(d_mcnt, 0, sizeof(uint32_t), st));      // overflow flag
            hipLaunchKernelGGL(marker_block_kernel<false>, dim3(n_genomes), dim3(MB_THREADS), 0, st, d_mstage, d_toff, d_tmoff, d_gft, d_mdense, d_moff, d_mcnt, 0xFFFFFFFFu);
            JHIP(hipMemcpyAsync(h_moff + n_genomes + 1, d_mcnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        } else {
            PSK_TRY(marker_segsort());
        }
        PSK_TRY(marker_offsets());
        ctx->t_end(st);
        return PSK_OK;
    }

    bool marker_block = false;
    uint32_t marker_slices = 1;      // slices per genome of the distinct-marker pass (few large genomes)
    bool marker_dense_ready = false;      // d_mdense holds the raw markers (marker_compact_kernel ran; the distinct-marker pass then overwrites the tile-local lists it read)
    bool marker_cap_ok = true, marker_capped = false;      // the tagged sort takes the EXPECTED number of raw markers (x 2 + 65 536) instead of the seed count; capped: this attempt did
    // tile-local lists -> dense per-genome segments -> segmented radix sort -> distinct values (in d_mstage)
    psk_status marker_segsort() {
        const size_t ns = h_goff[n_genomes];
        size_t tmp_bytes = 0;
        // Few genomes with millions of markers each (Gb-scale): a segmented sort hands each segment to too few workgroups (60 ms for
        // 8 x 3 M markers); tagged with the genome number, one device-wide radix sort does all of them (PSK_MARKER_TAGSORT=0: segmented)
        int gbits = 0; while ((1ull << gbits) < (uint64_t)n_genomes + 1) gbits++;      // tags 0 .. n_genomes - 1, and n_genomes for the padding
        static const bool tag_off = getenv("PSK_MARKER_TAGSORT") && getenv("PSK_MARKER_TAGSORT")[0] == '0';
        const bool tagged = !tag_off && MARKER_BITS + gbits <= 64;
        if (!marker_dense_ready) hipLaunchKernelGGL(marker_compact_kernel, dim3((n_tiles + 3) / 4), dim3(256), 0, st, d_mstage, d_toff, d_tmoff, n_tiles, d_mdense, tagged ? (const uint4*)d_tinfo : (const uint4*)nullptr);
        marker_dense_ready = true;      // (the sorts below leave their input as it is)
        marker_capped = false;
        if (ns > 0 && tagged) {
            // The dense array is sized by the SEED count (the raw markers' total is only on the device), but the sort need not be: raw markers number ~ L / marker_c,
            // an eighth of the seeds at the default parameters - sorting 192 M padded keys for 24 M markers was 4.8 ms of the 8 x 3 Gb step's sketching. Twice the
            // expectation + 65 536 entries; a batch that holds more (low-complexity sequence) is flagged by the pad kernel and sorted again at full size (phase3).
            uint64_t expect = 0;
            for (uint32_t g = 0; g < n_genomes; g++) expect += g_total_len[g] / (uint64_t)p->marker_c;
            size_t nm = ns;
            if (marker_cap_ok && 2 * expect + 65536 < (uint64_t)ns) { nm = (size_t)(2 * expect + 65536); marker_capped = true; }
            JHIP(hipMemsetAsync(d_mcnt, 0, sizeof(uint32_t), st));      // overflow flag
            hipLaunchKernelGGL(marker_pad_kernel, dim3((uint32_t)((nm + 255) / 256)), dim3(256), 0, st, d_mdense, d_tmoff + n_tiles, (uint32_t)nm, (uint64_t)n_genomes << MARKER_BITS, d_mcnt);
            JHIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, d_mdense, d_msorted, (int)nm, 0, MARKER_BITS + gbits, st));
            PSK_TRY(R->s_tmp.reserve(tmp_bytes));
            JHIP(hipcub::DeviceRadixSort::SortKeys(R->s_tmp.p, tmp_bytes, d_mdense, d_msorted, (int)nm, 0, MARKER_BITS + gbits, st));
            JHIP(hipMemcpyAsync(h_moff + n_genomes + 1, d_mcnt, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        } else if (ns > 0) {
            JHIP(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, tmp_bytes, d_mdense, d_msorted, (int)ns, (int)n_genomes, d_sbeg, d_sbeg + 1, 0, 2 * K_MARKER, st));
            PSK_TRY(R->s_tmp.reserve(tmp_bytes));
            JHIP(hipcub::DeviceSegmentedRadixSort::SortKeys(R->s_tmp.p, tmp_bytes, d_mdense, d_msorted, (int)ns, (int)n_genomes, d_sbeg, d_sbeg + 1, 0, 2 * K_MARKER, st));
        }
        // few large genomes: their segments in slices (marker_unique_sliced_kernel); many smaller ones: a workgroup each
        uint64_t est = 0;
        for (uint32_t g = 0; g < n_genomes; g++) est = std::max<uint64_t>(est, g_total_len[g] / (uint64_t)p->marker_c);
        marker_slices = (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(1, est / 8192));
        if (const char* e = getenv("PSK_MARKER_SLICES")) marker_slices = (uint32_t)std::max(1, std::min(1024, atoi(e)));      // tests: slices whatever the size
        if (marker_slices > 1 && (uint64_t)marker_slices * n_genomes <= (1u << 20) && !getenv("PSK_MARKER_UNSLICED")) {
            PSK_TRY(R->s_slices.reserve(4 * (size_t)marker_slices * n_genomes + 256));
            uint32_t* d_sc = (uint32_t*)R->s_slices.p;
            hipLaunchKernelGGL(marker_unique_sliced_kernel<0>, dim3(marker_slices, n_genomes), dim3(256), 0, st, d_msorted, d_mstage, d_sbeg, d_sbeg + 1, d_sc);
            hipLaunchKernelGGL(marker_slice_scan_kernel, dim3(n_genomes), dim3(1024), 0, st, d_sc, marker_slices, d_moff);
            hipLaunchKernelGGL(marker_unique_sliced_kernel<1>, dim3(marker_slices, n_genomes), dim3(256), 0, st, d_msorted, d_mstage, d_sbeg, d_sbeg + 1, d_sc);
        } else {
            marker_slices = 1;
            hipLaunchKernelGGL(marker_unique_kernel, dim3(n_genomes), dim3(256), 0, st, d_msorted, d_mstage, d_sbeg, d_sbeg + 1, d_moff);
        }
        return PSK_OK;
    }
    // distinct counts -> offsets, on their way to the host
    psk_status marker_offsets() {
        size_t tmp_bytes = 0;
        JHIP(hipMemsetAsync(d_moff + n_genomes, 0, sizeof(uint32_t), st));
        JHIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_moff, d_moff, (int)(n_genomes + 1), st));
        PSK_TRY(R->s_tmp.reserve(tmp_bytes));
        JHIP(hipcub::DeviceScan::ExclusiveSum(R->s_tmp.p, tmp_bytes, d_moff, d_moff, (int)(n_genomes + 1), st));
        JHIP(hipMemcpyAsync(h_moff, d_moff, sizeof(uint32_t) * (n_genomes + 1), hipMemcpyDeviceToHost, st));
        return PSK_OK;
    }

    // total distinct markers known: dense marker array
    psk_status phase3() {
        if (empty) return PSK_OK;
        JHIP(hipStreamSynchronize(st));
        if (marker_block && h_moff[n_genomes + 1]) {   // a genome held more raw markers than the LDS path takes: redo on the sort path
            marker_block = false;
            h_moff[n_genomes + 1] = 0;
            PSK_TRY(marker_segsort());
            PSK_TRY(marker_offsets());
            JHIP(hipStreamSynchronize(st));
        }
        if (!marker_block && marker_capped && h_moff[n_genomes + 1]) {   // more raw markers than the capped sort took: again, at the seed count
            marker_cap_ok = false;
            h_moff[n_genomes + 1] = 0;
            PSK_TRY(marker_segsort());
            PSK_TRY(marker_offsets());
            JHIP(hipStreamSynchronize(st));
        }
        const uint32_t total_markers = h_moff[n_genomes];
        PSK_TRY(ctx->pool_alloc(sizeof(uint64_t) * ((size_t)total_markers + 1), &store->mbase, &store->mbytes));
        store->markers = (uint64_t*)store->mbase;
        hipLaunchKernelGGL(marker_copy_kernel, dim3(n_genomes, marker_block ? 1u : marker_slices), dim3(256), 0, st, marker_block ? d_mdense : d_mstage, d_sbeg, d_moff, store->markers);
        return PSK_OK;
    }

    // ONE small genome (a contig about to be queried, a genome sketched on its own): the whole pipeline is enqueued against arrays sized from the
    // EXPECTED counts and the host synchronises once, at the end, where the general path stops three times to learn a size (seeds, distinct markers,
    // completion). *done = false: not eligible, or a count did not fit (low-complexity input) - the general path runs instead.
    psk_status run_small(psk_sketch** out, bool* done) {
        *done = false;
        const bool off = getenv("PSK_SKETCH_SMALL") && getenv("PSK_SKETCH_SMALL")[0] == '0';
        if (off || empty || n_genomes != 1 || n_tiles > (uint32_t)MB_TILES || n_desc > 65536) return PSK_OK;
        const uint64_t bases = sk[0]->total_len;
        if (bases / (uint64_t)p->marker_c > (uint64_t)(MB_CAP * 3 / 4)) return PSK_OK;
        // capacity: the expected count (hash-selected: one k-mer in c), a tenth more, six standard deviations of a binomial and a floor
        auto cap_of = [](double expect) { return (uint32_t)(expect * 1.1 + 6.0 * sqrt(expect) + 64.0); };
        const uint32_t ns_cap = cap_of((double)bases / (double)p->c), mk_cap = std::min<uint32_t>((uint32_t)MB_CAP, cap_of((double)bases / (double)p->marker_c));
        // device: [contig descriptors | contig first tiles | genome first tiles {0, n_tiles}] (one upload), work arrays, [result block] (one download)
        const size_t i_desc = 0, i_cft = align_up(sizeof(ContigDesc) * (size_t)n_desc, 16), i_gft = i_cft + 4 * (size_t)(n_desc + 1), i_end = align_up(i_gft + 8, 256);
        const size_t w_cnt = i_end, w_toff = w_cnt + 4 * (size_t)(n_tiles + 1), w_tmc = w_toff + 4 * (size_t)(n_tiles + 1), w_end = align_up(w_tmc + 4 * (size_t)(n_tiles + 1), 256);
        const size_t r_res = w_end, r_moff = r_res + 16, r_flag = r_moff + 8, r_coff = r_flag + 8, r_end = r_coff + 4 * (size_t)(n_desc + 1);
        PSK_TRY(R->s_offs.reserve(r_end + 256));
        PSK_TRY(R->s_packed.reserve(sizeof(uint32_t) * ((size_t)n_tiles * TILE_WORDS + 8)));
        PSK_TRY(R->s_mask.reserve(sizeof(uint64_t) * (size_t)n_tiles * TILE_MASKS));
        PSK_TRY(R->s_counts.reserve(sizeof(uint4) * ((size_t)n_tiles + 1) + sizeof(uint32_t) * ((size_t)n_tiles + 4)));
        PSK_TRY(R->s_mark.reserve(sizeof(uint64_t) * ((size_t)ns_cap + 1)));
        char* B = (char*)R->s_offs.p;
        d_desc = (ContigDesc*)(B + i_desc); d_cft = (uint32_t*)(B + i_cft); d_gft = (uint32_t*)(B + i_gft);
        d_cnt = (uint32_t*)(B + w_cnt); d_toff = (uint32_t*)(B + w_toff); d_tmc = (uint32_t*)(B + w_tmc);
        uint32_t* d_res = (uint32_t*)(B + r_res);
        d_tinfo = (uint4*)R->s_counts.p; d_tci = (uint32_t*)(d_tinfo + n_tiles + 1);
        d_packed = (uint32_t*)R->s_packed.p; d_mask = (uint64_t*)R->s_mask.p; d_mstage = (uint64_t*)R->s_mark.p;
        void* hp;
        PSK_TRY(R->pin(i_end + (r_end - r_res) + 64, &hp));
        char* H = (char*)hp;
        memcpy(H + i_desc, descs.data(), sizeof(ContigDesc) * (size_t)n_desc);
        { uint32_t* h_cft = (uint32_t*)(H + i_cft); for (int i = 0; i < n_desc; i++) h_cft[i] = descs[i].first_tile; h_cft[n_desc] = n_tiles; }
        { uint32_t* h_gft = (uint32_t*)(H + i_gft); h_gft[0] = 0; h_gft[1] = n_tiles; }
        const uint32_t* h_res = (const uint32_t*)(H + i_end);
        // the sketch's arrays, sized by the capacities
        store = std::make_shared<SketchStore>();
        const size_t ns = ns_cap;
        const size_t b_kmer = 0, b_pos = align_up(b_kmer + 4 * ns, 256), b_meta = align_up(b_pos + 4 * ns, 256),
                     b_pm = align_up(b_meta + 4 * ns, 256), b_cstart = align_up(b_pm + 8 * ns, 256), b_end = align_up(b_cstart + 4 * (size_t)(n_desc + 1), 256);
        store->ctx = ctx->dev;
        PSK_TRY(ctx->pool_alloc(b_end, &store->base, &store->bytes));
        char* sb = (char*)store->base;
        store->seed_kmer = (uint32_t*)(sb + b_kmer); store->seed_pos = (uint32_t*)(sb + b_pos); store->seed_meta = (uint32_t*)(sb + b_meta);
        store->seed_pm = (uint64_t*)(sb + b_pm); store->contig_seed_start = (uint32_t*)(sb + b_cstart);
        PSK_TRY(ctx->pool_alloc(sizeof(uint64_t) * ((size_t)mk_cap + 1), &store->mbase, &store->mbytes));
        store->markers = (uint64_t*)store->mbase;
        JHIP(hipMemcpyAsync(B, H, i_end, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(tile_contig_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, st, d_desc, n_desc, n_tiles, d_tci, d_tinfo);
        ctx->t_begin(K_SKETCH_SCAN, st);
        hipLaunchKernelGGL(sketch_scan_kernel, dim3(n_tiles), dim3(TILE_THREADS), 0, st, d_bases, d_desc, d_tci, d_packed, d_mask, d_cnt, C);
        ctx->t_end(st);
        ctx->t_begin(K_SKETCH_EMIT, st);
        hipLaunchKernelGGL(sketch_small_offsets_kernel, dim3(1), dim3(1024), 0, st, (const uint32_t*)d_cnt, n_tiles, d_toff, (const uint32_t*)d_cft, (uint32_t)n_desc,
                           store->contig_seed_start, d_res, (uint32_t*)(B + r_coff), (uint32_t*)(B + r_flag));
        hipLaunchKernelGGL(sketch_emit_kernel, dim3((n_tiles + EMIT_WAVES - 1) / EMIT_WAVES), dim3(64 * EMIT_WAVES), 0, st, d_tinfo, d_packed, d_mask, d_toff, d_res + 2, n_tiles,
                           store->seed_kmer, store->seed_pos, store->seed_meta, store->seed_pm, d_mstage, d_tmc, C, ns_cap);
        ctx->t_end(st);
        ctx->t_begin(K_SKETCH_SORT, st);
        hipLaunchKernelGGL(marker_block_kernel<true>, dim3(1), dim3(MB_THREADS), 0, st, d_mstage, d_toff, d_tmc, d_gft, store->markers, (uint32_t*)(B + r_moff), (uint32_t*)(B + r_flag), mk_cap);
        ctx->t_end(st);
        JHIP(hipMemcpyAsync(H + i_end, B + r_res, r_end - r_res, hipMemcpyDeviceToHost, st));
        JHIP(hipStreamSynchronize(st));
        const uint32_t total = h_res[0], n_mark = h_res[5], flag = h_res[6];
        if (total > ns_cap || flag) { store.reset(); return PSK_OK; }      // a count did not fit: the general path sizes the arrays from the counts
        const uint32_t* h_co = h_res + 8;
        psk_sketch* s = sk[0];
        s->store = store;
        s->seed_off = 0; s->n_seeds = total; s->marker_off = 0; s->n_markers = n_mark; s->contig_off = 0;
        s->contig_seed_start.assign(h_co, h_co + n_desc + 1);
        out[0] = s;
        sk.clear();
        *done = true;
        return PSK_OK;
    }

    psk_status finish(psk_sketch** out) {
        if (empty) {
            for (uint32_t g = 0; g < n_genomes; g++) { sk[g]->contig_seed_start.assign(sk[g]->contig_len.size() + 1, 0); out[g] = sk[g]; }
            sk.clear();
            return PSK_OK;
        }
        JHIP(hipStreamSynchronize(st));
        for (uint32_t g = 0; g < n_genomes; g++) {
            psk_sketch* s = sk[g];
            s->store = store;
            s->seed_off = h_goff[g]; s->n_seeds = h_goff[g + 1] - h_goff[g];
            s->marker_off = h_moff[g]; s->n_markers = h_moff[g + 1] - h_moff[g];
            s->contig_off = g_first_desc[g];
            uint32_t nc = g_first_desc[g + 1] - g_first_desc[g];
            s->contig_seed_start.resize(nc + 1);
            for (uint32_t c = 0; c <= nc; c++) s->contig_seed_start[c] = h_coff[g_first_desc[g] + c] - h_goff[g];
            out[g] = s;
        }
        sk.clear();
        return PSK_OK;
    }
#undef JHIP
};

psk_status sketch_batch_impl(Lane* ctx, const psk_params* p, const uint8_t* d_bases,
                             const uint64_t* contig_off, const uint64_t* contig_len,
                             const uint32_t* genome_first_contig, uint32_t n_genomes,
                             int want_seeds, psk_sketch** out, const uint32_t* d_packed_in) {
    if (!ctx || !p || !out || (!genome_first_contig && n_genomes)) { psk_set_error("sketch: NULL argument"); return PSK_EINVAL; }
    if (p->k < 1 || p->k > 16) { psk_set_error("Value of k > 16 for DNA; not allowed (k=%d)", p->k); return PSK_EINVAL; }
    if (p->c < 1 || p->marker_c < 1) { psk_set_error("compression factors must be >= 1"); return PSK_EINVAL; }
    for (uint32_t g = 0; g < n_genomes; g++) out[g] = nullptr;
    // The pipeline can run as J sub-batches on J streams (PSK_SKETCH_JOBS), sub-batch j+1 scanning while
    // sub-batch j emits and sorts. Measured on MI355X (profiles/r1d_overlap.md) this gains nothing: scan AND
    // emit are both VALU-issue bound, overlapped they slow each other by exactly what they take. Default J = 1.
    uint64_t total = 0;
    std::vector<uint64_t> gb(n_genomes);
    for (uint32_t g = 0; g < n_genomes; g++) {
        uint64_t b = 0;
        for (uint32_t c = genome_first_contig[g]; c < genome_first_contig[g + 1]; c++) b += contig_len[c];
        gb[g] = b; total += b;
    }
    const char* env = getenv("PSK_SKETCH_JOBS");
    uint32_t J = env && !d_packed_in ? (uint32_t)atoi(env) : 1u;      // (packed input is laid out in ONE job's tile order)
    J = std::max(1u, std::min(J, std::min(8u, n_genomes ? n_genomes : 1u)));
    std::vector<uint32_t> cut(J + 1, n_genomes);
    cut[0] = 0;
    { uint64_t acc = 0; uint32_t j = 1; for (uint32_t g = 0; g < n_genomes && j < J; g++) { acc += gb[g]; if (acc >= total * j / J) cut[j++] = g + 1; } }
    std::vector<SketchJob> jobs(J);
    auto abort_all = [&](psk_status rc) {
        for (auto& jb : jobs) if (jb.R) (void)hipStreamSynchronize(jb.R->stream);
        for (auto& jb : jobs) jb.drop();
        for (uint32_t g = 0; g < n_genomes; g++) { delete out[g]; out[g] = nullptr; }
        return rc;
    };
    for (uint32_t j = 0; j < J; j++) {
        SketchJob& jb = jobs[j];
        jb.ctx = ctx; jb.p = p; jb.d_bases = d_bases; jb.want_seeds = want_seeds; jb.packed_in = d_packed_in;
        psk_status rc = ctx->job(j, &jb.R);
        if (rc != PSK_OK) return abort_all(rc);
        jb.st = jb.R->stream;
        rc = jb.prepare(contig_off, contig_len, genome_first_contig + cut[j], cut[j + 1] - cut[j]);
        if (rc != PSK_OK) return abort_all(rc);
    }
    if (J == 1 && n_genomes == 1 && !d_packed_in) {
        bool done = false;
        jobs[0].make_objects();
        psk_status rc = jobs[0].run_small(out, &done);
        if (rc != PSK_OK) return abort_all(rc);
        if (done) return PSK_OK;
    }
    hipEvent_t prev = nullptr;
    for (uint32_t j = 0; j < J; j++) { psk_status rc = jobs[j].phase1(prev); if (rc != PSK_OK) return abort_all(rc); if (!jobs[j].empty) prev = jobs[j].R->scan_done; }
    for (uint32_t j = 0; j < J; j++) jobs[j].make_objects();      // (while the scans run)
    for (uint32_t j = 0; j < J; j++) { psk_status rc = jobs[j].phase2(); if (rc != PSK_OK) return abort_all(rc); }
    for (uint32_t j = 0; j < J; j++) { psk_status rc = jobs[j].phase3(); if (rc != PSK_OK) return abort_all(rc); }
    for (uint32_t j = 0; j < J; j++) { psk_status rc = jobs[j].finish(out + cut[j]); if (rc != PSK_OK) return abort_all(rc); }
    return PSK_OK;
}

// ---- reference index, built on first use: ONE device radix sort of (slot<<32 | kmer) over all the
// sketches that need one; LSD radix sort is stable, so equal k-mers keep their (contig,pos) order ----
struct IdxSeg { const uint32_t* kmer; const uint64_t* pm; uint32_t n; uint32_t out_off; uint32_t bshift, nb, boff; };

__global__ __launch_bounds__(256) void index_gather_kernel(const IdxSeg* __restrict__ segs, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
    const IdxSeg sg = segs[blockIdx.y];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += gridDim.x * blockDim.x) {
        key[sg.out_off + i] = ((uint64_t)blockIdx.y << 32) | sg.kmer[i];
        val[sg.out_off + i] = i;
    }
}

// bucket[boff + b] = first entry of the sketch's sorted slice whose k-mer >> bshift is >= b (b = 0..nb)
__global__ __launch_bounds__(256) void index_bucket_kernel(const IdxSeg* __restrict__ segs, const uint64_t* __restrict__ key, const uint32_t* __restrict__ perm,
                                                           uint32_t total, uint32_t* __restrict__ bucket, uint64_t* __restrict__ pms, uint32_t* __restrict__ km32) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t k = key[i];
    const IdxSeg sg = segs[(uint32_t)(k >> 32)];
    const uint32_t li = i - sg.out_off, b = (uint32_t)k >> sg.bshift;
    pms[i] = sg.pm[perm[i]];
    km32[i] = (uint32_t)k;
    uint32_t from = 0;
    if (li > 0) from = ((uint32_t)key[i - 1] >> sg.bshift) + 1;
    for (uint32_t bb = from; bb <= b; bb++) bucket[sg.boff + bb] = li;
    if (li == sg.n - 1) for (uint32_t bb = b + 1; bb <= sg.nb; bb++) bucket[sg.boff + bb] = sg.n;
}

// Gb-scale sketches (>= IDX_OWN_SORT seeds): every sketch its own sort of (u32 k-mer, u32 position rank) pairs straight from its seed array - 8 bytes per element and
// 2k key bits instead of the group sort's 12 bytes and 33 bits (a slot tag above the k-mer): four passes of 16 B per element instead of five of 24 (8 x 3 Gb: the index
// bracket 15.3 -> 9 ms per step). The values are the same iota for every sketch; the bucket tables and the position | meta copies come from one launch over (entry, sketch).
constexpr uint32_t IDX_OWN_SORT = 1u << 22;
__global__ __launch_bounds__(256) void index_iota_kernel(uint32_t* __restrict__ v, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}
__global__ __launch_bounds__(256) void index_bucket32_kernel(const IdxSeg* __restrict__ segs, const uint32_t* __restrict__ km32, const uint32_t* __restrict__ perm,
                                                             uint32_t* __restrict__ bucket, uint64_t* __restrict__ pms) {
    const IdxSeg sg = segs[blockIdx.y];
    const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= sg.n) return;
    const uint32_t i = sg.out_off + li, k = km32[i], b = k >> sg.bshift;
    pms[i] = sg.pm[perm[i]];
    uint32_t from = 0;
    if (li > 0) from = (km32[i - 1] >> sg.bshift) + 1;
    for (uint32_t bb = from; bb <= b; bb++) bucket[sg.boff + bb] = li;
    if (li == sg.n - 1) for (uint32_t bb = b + 1; bb <= sg.nb; bb++) bucket[sg.boff + bb] = sg.n;
}

// ---- index of one sketch by ONE workgroup (sketches up to IDXB_MAX_SEEDS seeds, i.e. genomes up to ~30 Mb at
// c = 125): counting sort on the k-mer's top bits in LDS (hash-selected k-mers are uniform, ~2-4 per bucket), then
// every bucket is put in (k-mer, seed index) order by one thread. The bucket starts ARE the lookup table. One launch
// replaces gather + 5 radix passes + table build: 100 sketches index in ~30 us instead of ~450.
constexpr int IDXB_THREADS = 1024;
constexpr int IDXB_MAX_LB = 14;                       // 16 384 buckets = 64 kB of LDS counters
constexpr uint32_t IDXB_MAX_SEEDS = 1u << 18;
constexpr int IDXT_MAX_LB = 10;                       // tiny sketches: 1 024 buckets = 4 kB of LDS counters, 256 threads
constexpr uint32_t IDXT_MAX_SEEDS = 1u << 12;

// T threads, 2^LB LDS counters: (1 024, 14) for genomes, (256, 10) for sketches of up to IDXT_MAX_SEEDS seeds (contigs): a workgroup of the
// former takes half a CU whatever the sketch's size - 10 000 contigs indexed in 3.8 ms - the latter fits sixteen to a CU
template <int T, int LB>
__global__ __launch_bounds__(T) void index_block_kernel(const IdxSeg* __restrict__ segs, const IdxSeg one, uint32_t slices, uint64_t* __restrict__ key, uint32_t* __restrict__ perm,
                                                                    uint64_t* __restrict__ pms, uint32_t* __restrict__ bucket, uint32_t* __restrict__ km32) {
    // `slices` workgroups share one sketch: workgroup (seg, sl) owns buckets [B0, B1) = the sl-th part of the bucket
    // space and the index positions its k-mers sort to. Every workgroup streams ALL of the sketch's k-mers (coalesced,
    // cheap) but histograms, scatters, orders and writes only its own part, so the scattered traffic of a sketch
    // is spread over `slices` CUs; the number of k-mers below B0 gives its base position without any grid sync.
    __shared__ uint32_t s_cnt[1 << LB];
    __shared__ uint32_t s_part[T / 64 + 1];
    const IdxSeg sg = segs ? segs[blockIdx.x / slices] : one;      // (a single sketch's descriptor rides in the kernel arguments: no table to upload)
    const uint32_t sl = blockIdx.x % slices;
    const uint32_t tid = threadIdx.x, n = sg.n, nb = sg.nb, sh = sg.bshift;
    const uint32_t B0 = (uint32_t)((uint64_t)nb * sl / slices), B1 = (uint32_t)((uint64_t)nb * (sl + 1) / slices), nbl = B1 - B0;
    uint64_t* __restrict__ K = key + sg.out_off;
    for (uint32_t b = tid; b < nbl; b += T) s_cnt[b] = 0;
    __syncthreads();
    // eight independent loads per thread per round trip, then the dependent LDS atomics
    uint32_t below = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 8 * T) {
        uint32_t km[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const uint32_t i = i0 + tid + r * T; km[r] = i < n ? sg.kmer[i] : 0xFFFFFFFFu; }
#pragma unroll
        for (int r = 0; r < 8; r++) if (i0 + tid + r * T < n) {
            const uint32_t b = km[r] >> sh;
            if (b < B0) below++; else if (b < B1) atomicAdd(&s_cnt[b - B0], 1u);
        }
    }
    uint32_t base;
    block_exclusive_scan<T>(below, s_part, &base);        // base = k-mers that sort before this part
    __syncthreads();
    // exclusive scan of the bucket counts: a run of `per` buckets per thread, then one workgroup scan of the run totals
    const uint32_t per = (nbl + T - 1) / T, b0 = tid * per;
    uint32_t sum = 0;
    for (uint32_t j = 0; j < per; j++) if (b0 + j < nbl) sum += s_cnt[b0 + j];
    uint32_t mine_total;
    uint32_t run = base + block_exclusive_scan<T>(sum, s_part, &mine_total);
    for (uint32_t j = 0; j < per; j++) if (b0 + j < nbl) { const uint32_t c = s_cnt[b0 + j]; s_cnt[b0 + j] = run; run += c; }
    __syncthreads();
    for (uint32_t b = tid; b < nbl; b += T) bucket[sg.boff + B0 + b] = s_cnt[b];
    if (tid == 0 && sl == slices - 1) bucket[sg.boff + nb] = n;
    __syncthreads();
    // scatter (k-mer, seed index) to the bucket's range; the counters become the bucket ENDS
    for (uint32_t i0 = 0; i0 < n; i0 += 8 * T) {
        uint32_t km[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const uint32_t i = i0 + tid + r * T; km[r] = i < n ? sg.kmer[i] : 0xFFFFFFFFu; }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = i0 + tid + r * T, b = km[r] >> sh;
            if (i < n && b >= B0 && b < B1) K[atomicAdd(&s_cnt[b - B0], 1u)] = ((uint64_t)km[r] << 32) | i;
        }
    }
    __threadfence_block();
    __syncthreads();
    // order every bucket by (k-mer, seed index): equal k-mers keep their (contig, pos) order, like a stable sort
    for (uint32_t b = tid; b < nbl; b += T) {
        const uint32_t lo = b ? s_cnt[b - 1] : base, hi = s_cnt[b], m = hi - lo;
        if (m < 2) continue;
        uint64_t* a = K + lo;
        if (m <= 8) {   // the common case: eight independent loads, a sorting network in registers, eight stores
            uint64_t v[8];
#pragma unroll
            for (int x = 0; x < 8; x++) v[x] = (uint32_t)x < m ? a[x] : ~0ull;
#define PSK_CE(p, q) { const uint64_t lo_ = v[p] < v[q] ? v[p] : v[q], hi_ = v[p] < v[q] ? v[q] : v[p]; v[p] = lo_; v[q] = hi_; }
            PSK_CE(0, 1) PSK_CE(2, 3) PSK_CE(4, 5) PSK_CE(6, 7)
            PSK_CE(0, 2) PSK_CE(1, 3) PSK_CE(4, 6) PSK_CE(5, 7)
            PSK_CE(1, 2) PSK_CE(5, 6) PSK_CE(0, 4) PSK_CE(3, 7)
            PSK_CE(1, 5) PSK_CE(2, 6)
            PSK_CE(1, 4) PSK_CE(3, 6)
            PSK_CE(2, 4) PSK_CE(3, 5)
            PSK_CE(3, 4)
#undef PSK_CE
#pragma unroll
            for (int x = 0; x < 8; x++) if ((uint32_t)x < m) a[x] = v[x];
        } else if (m <= 16) {   // rare (Poisson tail of the bucket load): same idea, bitonic network on 16 registers
            uint64_t v[16];
#pragma unroll
            for (int x = 0; x < 16; x++) v[x] = (uint32_t)x < m ? a[x] : ~0ull;
#pragma unroll
            for (int k = 2; k <= 16; k <<= 1)
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
                    for (int x = 0; x < 16; x++) {
                        const int y = x ^ j;
                        if (y > x) {
                            const bool up = (x & k) == 0;
                            const uint64_t lo_ = v[x] < v[y] ? v[x] : v[y], hi_ = v[x] < v[y] ? v[y] : v[x];
                            v[x] = up ? lo_ : hi_; v[y] = up ? hi_ : lo_;
                        }
                    }
#pragma unroll
            for (int x = 0; x < 16; x++) if ((uint32_t)x < m) a[x] = v[x];
        } else if (m <= 32) {
            for (uint32_t x = 1; x < m; x++) { const uint64_t v = a[x]; uint32_t y = x; while (y > 0 && a[y - 1] > v) { a[y] = a[y - 1]; y--; } a[y] = v; }
        } else {   // a repeat family: heap sort in place
            for (uint32_t st0 = m / 2; st0-- > 0;) { uint32_t r = st0; const uint64_t v = a[r]; for (;;) { uint32_t c = 2 * r + 1; if (c >= m) break; if (c + 1 < m && a[c + 1] > a[c]) c++; if (a[c] <= v) break; a[r] = a[c]; r = c; } a[r] = v; }
            for (uint32_t e = m - 1; e > 0; e--) { const uint64_t v = a[e]; a[e] = a[0]; uint32_t r = 0; for (;;) { uint32_t c = 2 * r + 1; if (c >= e) break; if (c + 1 < e && a[c + 1] > a[c]) c++; if (a[c] <= v) break; a[r] = a[c]; r = c; } a[r] = v; }
        }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t pend = base + mine_total;
    for (uint32_t p0 = base; p0 < pend; p0 += 8 * T) {
        uint64_t v[8], pm[8];
#pragma unroll
        for (int r = 0; r < 8; r++) { const uint32_t p = p0 + tid + r * T; v[r] = p < pend ? K[p] : 0; }
#pragma unroll
        for (int r = 0; r < 8; r++) { const uint32_t p = p0 + tid + r * T; pm[r] = p < pend ? sg.pm[(uint32_t)v[r]] : 0; }
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t p = p0 + tid + r * T;
            if (p < pend) { K[p] = ((uint64_t)(blockIdx.x / slices) << 32) | (uint32_t)(v[r] >> 32); km32[sg.out_off + p] = (uint32_t)(v[r] >> 32); perm[sg.out_off + p] = (uint32_t)v[r]; pms[sg.out_off + p] = pm[r]; }
        }
    }
}

// lazy: a single small sketch is indexed without waiting for the kernel (the caller's stream order covers its own use; any other lane finds
// the build's event in the IndexStore and waits for it on its stream). Everything else is complete when the call returns.
psk_status ensure_index(Lane* ctx, const psk_sketch* const* refs, uint32_t n, bool lazy) {
    std::lock_guard<std::mutex> index_lock(ctx->dev->index_mu);   // index builds mutate the sketches they index
    hipStream_t st = ctx->stream;
    std::vector<const psk_sketch*> todo;
    const uint64_t visit = ++ctx->dev->index_visit;      // a sketch listed twice is taken once
    for (uint32_t i = 0; i < n; i++) if (refs[i] && refs[i]->idx && refs[i]->idx->ready && refs[i]->idx->built_on != st)
        PSK_HIP(hipStreamWaitEvent(st, refs[i]->idx->ready, 0));
    for (uint32_t i = 0; i < n; i++) if (refs[i] && !refs[i]->idx && refs[i]->n_seeds && refs[i]->store && refs[i]->visit != visit) {
        refs[i]->visit = visit;
        todo.push_back(refs[i]);
    }
    const uint64_t GROUP = 1ull << 26;   // seeds per sort
    // small sketches first (one workgroup each), large ones after (device radix sort); PSK_INDEX_RADIX=1 forces the latter
    const bool force_radix = getenv("PSK_INDEX_RADIX") != nullptr;
    auto is_small = [&](const psk_sketch* s) { return !force_radix && s->n_seeds <= IDXB_MAX_SEEDS; };
    auto is_tiny = [&](const psk_sketch* s) { return !force_radix && s->n_seeds <= IDXT_MAX_SEEDS; };
    std::stable_partition(todo.begin(), todo.end(), is_small);
    std::stable_partition(todo.begin(), todo.end(), is_tiny);      // tiny ones first, then the other one-workgroup ones, then the radix-sorted ones
    size_t i0 = 0;
    while (i0 < todo.size()) {
        const bool small = is_small(todo[i0]), tiny = is_tiny(todo[i0]);
        size_t i1 = i0; uint64_t T = 0;
        while (i1 < todo.size() && is_small(todo[i1]) == small && is_tiny(todo[i1]) == tiny && i1 - i0 < 65535 && (i1 == i0 || T + todo[i1]->n_seeds <= GROUP)) { T += todo[i1]->n_seeds; i1++; }
        const uint32_t m = (uint32_t)(i1 - i0);
        std::vector<IdxSeg> segs(m);
        uint32_t off = 0, maxn = 0;
        uint64_t boff = 0;
        for (uint32_t j = 0; j < m; j++) {
            const psk_sketch* s = todo[i0 + j];
            // ~4 entries per bucket; k-mers of hash-selected seeds are spread evenly over the 2k-bit space
            int kbits = 2 * s->params.k, lb = 4;
            while (lb < (tiny ? IDXT_MAX_LB : small ? IDXB_MAX_LB : 22) && (1ull << (lb + 2)) < s->n_seeds) lb++;
            if (lb > kbits) lb = kbits;
            segs[j] = IdxSeg{s->store->seed_kmer + s->seed_off, s->store->seed_pm + s->seed_off, (uint32_t)s->n_seeds, off, (uint32_t)(kbits - lb), 1u << lb, (uint32_t)boff};
            off += (uint32_t)s->n_seeds; maxn = std::max(maxn, (uint32_t)s->n_seeds);
            boff += (1ull << lb) + 1;
        }
        auto ix = std::make_shared<IndexStore>();
        ix->ctx = ctx->dev;
        size_t kb = align_up(8 * (size_t)T, 256), vb = align_up(4 * (size_t)T, 256), bb = align_up(4 * (size_t)boff, 256);
        PSK_TRY(ctx->pool_alloc(2 * kb + 2 * vb + bb, &ix->base, &ix->bytes));
        ix->key = (uint64_t*)ix->base; ix->pms = (uint64_t*)((char*)ix->base + kb); ix->perm = (uint32_t*)((char*)ix->base + 2 * kb);
        ix->km32 = (uint32_t*)((char*)ix->base + 2 * kb + vb);
        ix->bucket = (uint32_t*)((char*)ix->base + 2 * kb + 2 * vb);
        const bool by_value = small && m == 1;
        if (!by_value) {
            PSK_TRY(ctx->s_offs.reserve(sizeof(IdxSeg) * m));
            PSK_HIP(hipMemcpyAsync(ctx->s_offs.p, segs.data(), sizeof(IdxSeg) * m, hipMemcpyHostToDevice, st));
        }
        const IdxSeg* d_segs = by_value ? (const IdxSeg*)nullptr : (const IdxSeg*)ctx->s_offs.p;
        if (small) {
            ctx->t_begin(K_SKETCH_SORT);
            // The kernel can split a sketch over several workgroups (slices of the bucket space). Measured on MI355X with
            // 101 sketches: 8 slices are no faster than 1 (0.38 vs 0.33 ms for the whole marker + index phase) - the
            // kernel's time is its chain of dependent round trips, not one CU's scattered traffic. PSK_INDEX_SLICES overrides.
            uint32_t slices = 1;
            if (const char* e = getenv("PSK_INDEX_SLICES")) slices = (uint32_t)std::max(1, std::min(8, atoi(e)));
            if (tiny) hipLaunchKernelGGL((index_block_kernel<256, IDXT_MAX_LB>), dim3(m), dim3(256), 0, st, d_segs, segs[0], 1u, ix->key, ix->perm, ix->pms, ix->bucket, ix->km32);
            else hipLaunchKernelGGL((index_block_kernel<IDXB_THREADS, IDXB_MAX_LB>), dim3(m * slices), dim3(IDXB_THREADS), 0, st, d_segs, segs[0], slices, ix->key, ix->perm, ix->pms, ix->bucket, ix->km32);
            ctx->t_end();
            if (lazy && by_value && todo.size() == 1) {
                PSK_HIP(hipEventCreateWithFlags(&ix->ready, hipEventDisableTiming));
                PSK_HIP(hipEventRecord(ix->ready, st));
                ix->built_on = st;
            } else PSK_HIP(hipStreamSynchronize(st));
            for (uint32_t j = 0; j < m; j++) {
                todo[i0 + j]->idx = ix; todo[i0 + j]->idx_off = segs[j].out_off;
                todo[i0 + j]->idx_boff = segs[j].boff; todo[i0 + j]->idx_bshift = segs[j].bshift;
            }
            i0 = i1;
            continue;
        }
        bool own_sort = getenv("PSK_INDEX_GROUP_SORT") == nullptr;      // (tests, A/B: the group sort for sketches of any size)
        int kb_max = 0;
        for (uint32_t j = 0; j < m; j++) { own_sort = own_sort && segs[j].n >= IDX_OWN_SORT; kb_max = std::max(kb_max, 2 * todo[i0 + j]->params.k); }
        if (own_sort) {
            PSK_TRY(ctx->s_mark.reserve(align_up(4 * (size_t)maxn, 256)));
            uint32_t* v_in = (uint32_t*)ctx->s_mark.p;
            size_t tmp = 0;
            PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)maxn, 0, kb_max, st));
            PSK_TRY(ctx->s_tmp.reserve(tmp));
            ctx->t_begin(K_SKETCH_SORT);
            hipLaunchKernelGGL(index_iota_kernel, dim3(std::min<uint32_t>((maxn + 255) / 256, 4096)), dim3(256), 0, st, v_in, maxn);
            for (uint32_t j = 0; j < m; j++) {
                size_t tj = tmp;
                PSK_HIP(hipcub::DeviceRadixSort::SortPairs(ctx->s_tmp.p, tj, segs[j].kmer, ix->km32 + segs[j].out_off, (const uint32_t*)v_in, ix->perm + segs[j].out_off, (int)segs[j].n, 0, 2 * todo[i0 + j]->params.k, st));
            }
            hipLaunchKernelGGL(index_bucket32_kernel, dim3((maxn + 255) / 256, m), dim3(256), 0, st, (const IdxSeg*)ctx->s_offs.p, (const uint32_t*)ix->km32, (const uint32_t*)ix->perm, ix->bucket, ix->pms);
            ctx->t_end();
        } else {
            PSK_TRY(ctx->s_mark.reserve(kb + vb));
            uint64_t* k_in = (uint64_t*)ctx->s_mark.p; uint32_t* v_in = (uint32_t*)((char*)ctx->s_mark.p + kb);
            ctx->t_begin(K_SKETCH_SORT);
            hipLaunchKernelGGL(index_gather_kernel, dim3(std::min<uint32_t>((maxn + 255) / 256, 2048), m), dim3(256), 0, st,      // (64 workgroups per sketch: 0.74 ms for two 24 M-seed sketches, a quarter of the chip)
                               (const IdxSeg*)ctx->s_offs.p, k_in, v_in);
            int slot_bits = 1; while ((1u << slot_bits) < m) slot_bits++;
            size_t tmp = 0;
            PSK_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, k_in, ix->key, v_in, ix->perm, (int)T, 0, 32 + slot_bits, st));
            PSK_TRY(ctx->s_tmp.reserve(tmp));
            PSK_HIP(hipcub::DeviceRadixSort::SortPairs(ctx->s_tmp.p, tmp, k_in, ix->key, v_in, ix->perm, (int)T, 0, 32 + slot_bits, st));
            hipLaunchKernelGGL(index_bucket_kernel, dim3((uint32_t)((T + 255) / 256)), dim3(256), 0, st, (const IdxSeg*)ctx->s_offs.p, (const uint64_t*)ix->key, (const uint32_t*)ix->perm, (uint32_t)T, ix->bucket, ix->pms, ix->km32);
            ctx->t_end();
        }
        PSK_HIP(hipStreamSynchronize(st));   // segs (host vector) feeds the async copy above
        for (uint32_t j = 0; j < m; j++) {
            todo[i0 + j]->idx = ix; todo[i0 + j]->idx_off = segs[j].out_off;
            todo[i0 + j]->idx_boff = segs[j].boff; todo[i0 + j]->idx_bshift = segs[j].bshift;
        }
        i0 = i1;
    }
    return PSK_OK;
}

// ------------------------------------------------------------------ probe tables (join of batches of many small pairs)
struct ProbeSeg { const uint32_t* key; const uint64_t* pms; ProbeLine* tab; uint32_t n, lines; };
// one thread per index entry; the head of every run of equal k-mers inserts (k-mer, position or - for a run - its first index entry, meta | count << 24)
__global__ __launch_bounds__(256) void probe_build_kernel(const ProbeSeg* __restrict__ segs) {
    const ProbeSeg S = segs[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.n) return;
    const uint32_t km = S.key[i];
    if (i > 0 && S.key[i - 1] == km) return;
    uint32_t cnt = 1;
    while (i + cnt < S.n && cnt < 255u && S.key[i + cnt] == km) cnt++;
    const uint64_t pm = S.pms[i];
    const uint32_t rmeta = (uint32_t)pm;
    // same packing as the join's record: counts >= 255 or reference contig numbers >= 2^23 read as 255 ("rerun in the wide format")
    const uint32_t y = (cnt >= 255u || (rmeta >> 24)) ? ((rmeta & 0xFFFFFFu) | (255u << 24)) : (rmeta | (cnt << 24));
    uint32_t L = probe_line(km, S.lines);
    for (;;) {
        ProbeLine* ln = S.tab + L;
        for (uint32_t s = 0; s < PROBE_SLOTS; s++) {
            if (atomicCAS(&ln->k[s], PROBE_EMPTY, km) == PROBE_EMPTY) { ln->v[s] = make_uint2(cnt > 1 ? i : (uint32_t)(pm >> 32), y); return; }
        }
        L = L + 1 < S.lines ? L + 1 : 0;
    }
}

psk_status ensure_probe(Lane* ctx, const psk_sketch* const* refs, uint32_t n) {
    std::lock_guard<std::mutex> index_lock(ctx->dev->index_mu);
    hipStream_t st = ctx->stream;
    std::vector<const psk_sketch*> todo;
    std::unordered_set<const psk_sketch*> seen;
    for (uint32_t i = 0; i < n; i++)
        if (refs[i] && refs[i]->idx && !refs[i]->ptab && refs[i]->n_seeds >= 64 && refs[i]->n_seeds <= (1u << 22) && seen.insert(refs[i]).second) todo.push_back(refs[i]);
    size_t i0 = 0;
    while (i0 < todo.size()) {
        size_t i1 = i0; uint64_t lines = 0; uint32_t maxn = 0;
        std::vector<ProbeSeg> segs;
        std::vector<uint64_t> loff;
        while (i1 < todo.size() && i1 - i0 < 65535 && lines < (1ull << 28)) {      // <= 16 GB of lines per store
            const psk_sketch* s = todo[i1];
            const uint32_t ln = (uint32_t)((s->n_seeds * 2 + 4) / 5);              // ~2.5 k-mers per line of 5 slots
            loff.push_back(lines);
            segs.push_back(ProbeSeg{s->idx->km32 + s->idx_off, s->idx->pms + s->idx_off, nullptr, (uint32_t)s->n_seeds, ln});
            lines += ln; maxn = std::max(maxn, (uint32_t)s->n_seeds); i1++;
        }
        auto ps = std::make_shared<ProbeStore>();
        ps->ctx = ctx->dev;
        PSK_TRY(ctx->pool_alloc(sizeof(ProbeLine) * (size_t)lines, &ps->base, &ps->bytes));
        for (size_t j = 0; j < segs.size(); j++) segs[j].tab = (ProbeLine*)ps->base + loff[j];
        PSK_HIP(hipMemsetAsync(ps->base, 0xFF, sizeof(ProbeLine) * (size_t)lines, st));
        PSK_TRY(ctx->s_offs.reserve(sizeof(ProbeSeg) * segs.size()));
        PSK_HIP(hipMemcpyAsync(ctx->s_offs.p, segs.data(), sizeof(ProbeSeg) * segs.size(), hipMemcpyHostToDevice, st));
        ctx->t_begin(K_SKETCH_SORT);
        hipLaunchKernelGGL(probe_build_kernel, dim3((maxn + 255) / 256, (uint32_t)segs.size()), dim3(256), 0, st, (const ProbeSeg*)ctx->s_offs.p);
        ctx->t_end();
        PSK_HIP(hipStreamSynchronize(st));      // segs (host vector) feeds the async copy above
        for (size_t j = 0; j < segs.size(); j++) { todo[i0 + j]->ptab = ps; todo[i0 + j]->ptab_off = loff[j]; todo[i0 + j]->ptab_lines = segs[j].lines; }
        i0 = i1;
    }
    return PSK_OK;
}

