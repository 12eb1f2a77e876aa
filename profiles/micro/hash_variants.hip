// Micro-benchmark: throughput of mm_hash64(x) < thr formulations on gfx950 (VALU-bound inner op of sketch_scan).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ uint64_t hashA(uint64_t key) {
    key = ~(key + (key << 21)); key = key ^ key >> 24; key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14; key = (key + (key << 2)) + (key << 4); key = key ^ key >> 28; key = key + (key << 31);
    return key;
}
__device__ __forceinline__ void opq(uint64_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ uint64_t hashC1(uint64_t key) {   // native 64-bit shifts, no multiply folding
    uint64_t t = key << 21; opq(t); key = ~(key + t);
    key = key ^ key >> 24;
    t = key << 3; opq(t); uint64_t u = key << 8; opq(u); key = (key + t) + u;
    key = key ^ key >> 14;
    t = key << 2; opq(t); u = key << 4; opq(u); key = (key + t) + u;
    key = key ^ key >> 28;
    t = key << 31; opq(t); key = key + t;
    return key;
}
__device__ __forceinline__ uint64_t shl(uint64_t k, int s) {   // alignbit-based 64-bit shift by constant
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t h2 = __builtin_amdgcn_alignbit(hi, lo, 32 - s), l2 = lo << s;
    uint64_t r = ((uint64_t)h2 << 32) | l2; opq(r); return r;
}
__device__ __forceinline__ uint64_t xshr(uint64_t k, int s) {
    uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    uint32_t l2 = lo ^ __builtin_amdgcn_alignbit(hi, lo, s), h2 = hi ^ (hi >> s);
    return ((uint64_t)h2 << 32) | l2;
}
__device__ __forceinline__ uint64_t hashC2(uint64_t key) {
    key = ~(key + shl(key, 21));
    key = xshr(key, 24);
    key = (key + shl(key, 3)) + shl(key, 8);
    key = xshr(key, 14);
    key = (key + shl(key, 2)) + shl(key, 4);
    key = xshr(key, 28);
    key = key + shl(key, 31);
    return key;
}
// D: one 32x32->64 multiply-add per constant multiply, high word fixed up with shift-adds; no carries, no 64-bit movs
__device__ __forceinline__ void opq32(uint32_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ uint64_t hashD(uint32_t x) {
    uint64_t r = (uint64_t)x * 0x200001u;                 // key + (key << 21), key < 2^32
    uint32_t lo = ~(uint32_t)r, hi = ~(uint32_t)(r >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= hi >> 24;
    r = (uint64_t)lo * 265u; { uint32_t t = (hi << 8) + hi; opq32(t); hi = (uint32_t)(r >> 32) + (hi << 3) + t; } lo = (uint32_t)r;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 14); hi ^= hi >> 14;
    r = (uint64_t)lo * 21u; { uint32_t t = (hi << 4) + hi; opq32(t); hi = (uint32_t)(r >> 32) + (hi << 2) + t; } lo = (uint32_t)r;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 28); hi ^= hi >> 28;
    r = (uint64_t)lo * 0x80000001u; hi = (uint32_t)(r >> 32) + (hi << 31) + hi; lo = (uint32_t)r;
    return ((uint64_t)hi << 32) | lo;
}
// E: low word by one 32x32->64 multiply, high word by v_mul_lo_u32 + add (no 64-bit addend to assemble)
__device__ __forceinline__ uint64_t hashE(uint32_t x) {
    uint64_t r = (uint64_t)x * 0x200001u;
    uint32_t lo = ~(uint32_t)r, hi = ~(uint32_t)(r >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= hi >> 24;
    r = (uint64_t)lo * 265u; hi = __umul24(hi, 265u) + (__umul24(hi >> 24, 265u) << 24) + (uint32_t)(r >> 32); lo = (uint32_t)r;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 14); hi ^= hi >> 14;
    r = (uint64_t)lo * 21u; hi = __umul24(hi, 21u) + (__umul24(hi >> 24, 21u) << 24) + (uint32_t)(r >> 32); lo = (uint32_t)r;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 28); hi ^= hi >> 28;
    r = (uint64_t)lo * 0x80000001u; hi = (uint32_t)(r >> 32) + (hi << 31) + hi; lo = (uint32_t)r;
    return ((uint64_t)hi << 32) | lo;
}
// F: like the compiler's form but the high word's product is a plain 32-bit multiply
__device__ __forceinline__ uint64_t hashF(uint32_t x) {
    uint64_t r = (uint64_t)x * 0x200001u;
    uint32_t lo = ~(uint32_t)r, hi = ~(uint32_t)(r >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= hi >> 24;
    uint32_t t = hi * 265u; opq32(t); r = (uint64_t)lo * 265u; hi = (uint32_t)(r >> 32) + t; lo = (uint32_t)r;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 14); hi ^= hi >> 14;
    t = hi * 21u; opq32(t); r = (uint64_t)lo * 21u; hi = (uint32_t)(r >> 32) + t; lo = (uint32_t)r;
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 28); hi ^= hi >> 28;
    t = hi * 0x80000001u; opq32(t); r = (uint64_t)lo * 0x80000001u; hi = (uint32_t)(r >> 32) + t; lo = (uint32_t)r;
    return ((uint64_t)hi << 32) | lo;
}

// G: NOT folded into the first xor-shift (P < 2^53), high-word products by v_mul_lo + add, 64-bit shifts for the xor-shifts
__device__ __forceinline__ uint64_t hashG(uint32_t x) {
    uint64_t P = (uint64_t)x * 0x200001u;
    uint32_t lo = (uint32_t)P, hi = (uint32_t)(P >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= 0xFFFFFF00u;         // = ~P ^ (~P >> 24)
    uint64_t r = (uint64_t)lo * 265u; hi = hi * 265u + (uint32_t)(r >> 32);
    uint64_t k = ((uint64_t)hi << 32) | (uint32_t)r; k ^= k >> 14;
    lo = (uint32_t)k; hi = (uint32_t)(k >> 32);
    r = (uint64_t)lo * 21u; hi = hi * 21u + (uint32_t)(r >> 32);
    k = ((uint64_t)hi << 32) | (uint32_t)r; k ^= k >> 28;
    lo = (uint32_t)k; hi = (uint32_t)(k >> 32);
    r = (uint64_t)lo * 0x80000001u; hi = (hi << 31) + (uint32_t)(r >> 32) + hi;
    return ((uint64_t)hi << 32) | (uint32_t)r;
}
// H: G with the x21 and x265 multiplies as v_lshl_add_u64 chains (shift <= 4)
__device__ __forceinline__ uint64_t lsa(uint64_t a, int s, uint64_t b) {
    uint64_t d;
    if (s == 2) asm("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(d) : "v"(a), "v"(b));
    else if (s == 3) asm("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_lshl_add_u64 %0, %1, 4, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint64_t hashH(uint32_t x) {
    uint64_t P = (uint64_t)x * 0x200001u;
    uint32_t lo = (uint32_t)P, hi = (uint32_t)(P >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= 0xFFFFFF00u;
    uint64_t r = (uint64_t)lo * 265u; hi = hi * 265u + (uint32_t)(r >> 32);
    uint64_t k = ((uint64_t)hi << 32) | (uint32_t)r; k ^= k >> 14;
    k = lsa(k, 4, lsa(k, 2, k));                                   // 21 k
    k ^= k >> 28;
    lo = (uint32_t)k; hi = (uint32_t)(k >> 32);
    r = (uint64_t)lo * 0x80000001u; hi = (hi << 31) + (uint32_t)(r >> 32) + hi;
    return ((uint64_t)hi << 32) | (uint32_t)r;
}
// I: G's first step, then the compiler's own 64-bit forms
__device__ __forceinline__ uint64_t hashI(uint32_t x) {
    uint64_t P = (uint64_t)x * 0x200001u;
    uint32_t lo = (uint32_t)P, hi = (uint32_t)(P >> 32);
    lo ^= __builtin_amdgcn_alignbit(hi, lo, 24); hi ^= 0xFFFFFF00u;
    uint64_t key = ((uint64_t)hi << 32) | lo;
    key = (key + (key << 3)) + (key << 8);
    key = key ^ key >> 14; key = (key + (key << 2)) + (key << 4); key = key ^ key >> 28; key = key + (key << 31);
    return key;
}
template <int V> __global__ void bench(uint32_t* out, uint64_t thr, int iters) {
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, cnt = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            x = x * 1664525u + 1013904223u;
            uint64_t k = x & 0x3FFFFFFFu;
            uint64_t h = V == 0 ? hashA(k) : V == 1 ? hashC1(k) : V == 2 ? hashC2(k) : V == 3 ? hashD((uint32_t)k) : V == 4 ? hashE((uint32_t)k) : V == 5 ? hashF((uint32_t)k) : V == 6 ? hashG((uint32_t)k) : V == 7 ? hashH((uint32_t)k) : hashI((uint32_t)k);
            cnt += h < thr;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = cnt;
}
template <int V> __global__ void check(uint64_t* out) { uint64_t k = threadIdx.x * 2654435761u & 0x3FFFFFFF; out[threadIdx.x] = V == 0 ? hashA(k) : V == 1 ? hashC1(k) : V == 2 ? hashC2(k) : V == 3 ? hashD((uint32_t)k) : V == 4 ? hashE((uint32_t)k) : V == 5 ? hashF((uint32_t)k) : V == 6 ? hashG((uint32_t)k) : V == 7 ? hashH((uint32_t)k) : hashI((uint32_t)k); }
template <int V> void run(uint32_t* d, uint64_t* c, const uint64_t* ref, uint64_t* mine) {
    check<V><<<1, 256>>>(c); hipMemcpy(mine, c, 2048, hipMemcpyDeviceToHost);
    int bad = 0; if (ref) for (int i = 0; i < 256; i++) bad += ref[i] != mine[i];
    uint64_t thr = UINT64_MAX / 125; int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a); bench<V><<<2048, 256>>>(d, thr, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double n = 2048.0 * 256 * iters * 16;
        printf("variant %d: %.3f ms  %.1f Ghash/s  mismatches %d\n", V, ms, n / ms / 1e6, bad);
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 2048 * 256);
    uint64_t* c; hipMalloc(&c, 8 * 256);
    uint64_t ref[256], mine[256];
    run<0>(d, c, nullptr, ref);
    run<1>(d, c, ref, mine); run<2>(d, c, ref, mine); run<3>(d, c, ref, mine); run<4>(d, c, ref, mine);
    run<5>(d, c, ref, mine); run<6>(d, c, ref, mine); run<7>(d, c, ref, mine); run<8>(d, c, ref, mine);
    return 0;
}
