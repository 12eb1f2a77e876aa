// Host side of the packed ingest (psk_sketch_many_host): ASCII -> 2-bit on the ingest worker threads, so that a genome crosses PCIe as
// L / 4 bytes instead of L (the boundary of the reference hands over host buffers, lib.rs:485-489: from host memory the path is bound
// by the link, not by the sketch kernels). Same codes as sketch_scan_kernel's phase 1 and the oracle's BYTE_TO_SEQ: A 0, C 1, G 2, T 3,
// case-insensitive, EVERY other byte 0; sixteen bases per 32-bit word, the first base in the highest two bits.
// Plain C++ (no device code): AVX-512BW where the CPU has it (the MI355X hosts: EPYC 9575F), a table-driven scalar loop otherwise.
#include <cstdint>
#include <cstring>
#include <immintrin.h>

namespace {

struct Lut { uint8_t v[256]; Lut() { memset(v, 0, sizeof v); v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } };
const Lut LUT;

inline uint32_t pack16_scalar(const uint8_t* s) {
    uint32_t w = 0;
    for (int i = 0; i < 16; i++) w = (w << 2) | LUT.v[s[i]];
    return w;
}

void pack_scalar(const uint8_t* src, uint64_t n, uint32_t* dst) {
    const uint64_t full = n / 16;
    for (uint64_t i = 0; i < full; i++) dst[i] = pack16_scalar(src + 16 * i);
    const uint32_t rem = (uint32_t)(n & 15);
    if (rem) {
        uint32_t w = 0;
        for (uint32_t i = 0; i < rem; i++) w |= (uint32_t)LUT.v[src[16 * full + i]] << (30 - 2 * i);
        dst[full] = w;
    }
}

__attribute__((target("avx512f,avx512bw"))) void pack_avx512(const uint8_t* src, uint64_t n, uint32_t* dst) {
    const __m512i fold = _mm512_set1_epi8((char)0xDF), three = _mm512_set1_epi8(3), one = _mm512_set1_epi8(1);
    const __m512i cA = _mm512_set1_epi8('A'), cC = _mm512_set1_epi8('C'), cG = _mm512_set1_epi8('G'), cT = _mm512_set1_epi8('T');
    const __m512i m41 = _mm512_set1_epi16(0x0104);            // maddubs: first byte of a pair x 4 + second x 1
    const __m512i m161 = _mm512_set1_epi32(0x00010010);       // madd: first 16-bit x 16 + second x 1
    const __m128i bswap = _mm_set_epi8(12, 13, 14, 15, 8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3);
    const uint64_t blocks = n / 64;
    for (uint64_t b = 0; b < blocks; b++) {
        const __m512i x = _mm512_loadu_si512((const void*)(src + 64 * b));
        const __m512i f = _mm512_and_si512(x, fold);
        const __mmask64 ok = _mm512_cmpeq_epi8_mask(f, cA) | _mm512_cmpeq_epi8_mask(f, cC) | _mm512_cmpeq_epi8_mask(f, cG) | _mm512_cmpeq_epi8_mask(f, cT);
        __m512i c = _mm512_and_si512(_mm512_srli_epi16(f, 1), three);                       // A 0, C 1, G 3, T 2
        c = _mm512_xor_si512(c, _mm512_and_si512(_mm512_srli_epi16(c, 1), one));            // A 0, C 1, G 2, T 3
        c = _mm512_maskz_mov_epi8(ok, c);
        const __m512i p2 = _mm512_maddubs_epi16(c, m41);                                     // 16-bit: c0 * 4 + c1
        const __m512i p4 = _mm512_madd_epi16(p2, m161);                                      // 32-bit: four bases in the low byte
        const __m128i bytes = _mm512_cvtepi32_epi8(p4);                                      // b0 .. b15, b_j = bases 4j .. 4j + 3
        _mm_storeu_si128((__m128i*)(dst + 4 * b), _mm_shuffle_epi8(bytes, bswap));           // word = b0 << 24 | b1 << 16 | b2 << 8 | b3
    }
    if (n & 63) pack_scalar(src + 64 * blocks, n & 63, dst + 4 * blocks);
}

bool have_avx512() {
    static const bool h = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    return h;
}

}  // namespace

// n bases at src -> ceil(n / 16) words at dst (the last word's unused low bits zero). mode: 0 = best available, 1 = scalar (tests)
extern "C" void psk_pack2bit_host(const uint8_t* src, uint64_t n, uint32_t* dst, int mode) {
    if (mode == 0 && have_avx512()) pack_avx512(src, n, dst);
    else pack_scalar(src, n, dst);
}
