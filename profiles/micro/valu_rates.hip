// Micro-benchmark: issue rate of the integer VALU instructions mm_hash64 can be built from, on gfx950.
// Each kernel runs 8 independent dependency chains per lane, 2048 x 256 threads (8 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHAINS 8
#define UNROLL 32
#define BODY32(NAME, ASM)                                                                     \
    __global__ void NAME(uint32_t* out, int iters, uint32_t c) {                                 \
        uint32_t a[CHAINS];                                                                      \
        for (int j = 0; j < CHAINS; j++) a[j] = threadIdx.x * 7u + j + blockIdx.x;               \
        for (int i = 0; i < iters; i++) {                                                        \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                 \
                _Pragma("unroll") for (int j = 0; j < CHAINS; j++) asm volatile(ASM : "+v"(a[j]) : "v"(c)); \
            }                                                                                    \
        }                                                                                        \
        uint32_t s = 0; for (int j = 0; j < CHAINS; j++) s += a[j];                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                          \
    }
#define BODY64(NAME, ASM)                                                                     \
    __global__ void NAME(uint32_t* out, int iters, uint32_t c) {                                 \
        uint64_t a[CHAINS];                                                                      \
        for (int j = 0; j < CHAINS; j++) a[j] = threadIdx.x * 7u + j + blockIdx.x;               \
        for (int i = 0; i < iters; i++) {                                                        \
            _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                 \
                _Pragma("unroll") for (int j = 0; j < CHAINS; j++) asm volatile(ASM : "+v"(a[j]) : "v"(c)); \
            }                                                                                    \
        }                                                                                        \
        uint64_t s = 0; for (int j = 0; j < CHAINS; j++) s += a[j];                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32);         \
    }

BODY32(k_add_u32, "v_add_u32 %0, %0, %1")
BODY32(k_xor_b32, "v_xor_b32 %0, %0, %1")
BODY32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
BODY32(k_add3, "v_add3_u32 %0, %0, %1, %1")
BODY32(k_xor3, "v_bfi_b32 %0, %0, %1, %0")
BODY32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
BODY32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
BODY32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
BODY32(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %1")
BODY32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
BODY32(k_min_u32, "v_min_u32 %0, %0, %1")
BODY32(k_and_or, "v_and_or_b32 %0, %0, %1, %1")
BODY32(k_bfe, "v_bfe_u32 %0, %0, 3, 15")
BODY32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
BODY32(k_cmp_addc, "v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc")
BODY32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
BODY32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
BODY32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %1")
BODY32(k_mad_u16, "v_mad_u16 %0, %0, %1, %1")
BODY32(k_dot4_u8, "v_dot4_u32_u8 %0, %0, %1, %0")
BODY64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
BODY64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0")
BODY64(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
BODY64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %0")
BODY64(k_pk_add_like, "v_pk_mov_b32 %0, %0, %0")


BODY32(k_and_b32, "v_and_b32 %0, %0, %1")
BODY32(k_or_b32, "v_or_b32 %0, %0, %1")
BODY32(k_not_b32, "v_not_b32 %0, %0")
BODY32(k_mov_b32, "v_mov_b32 %0, %0")
BODY32(k_sub_u32, "v_sub_u32 %0, %0, %1")
BODY32(k_lshl_c, "v_lshlrev_b32 %0, 3, %0")
BODY32(k_lshr_c, "v_lshrrev_b32 %0, 3, %0")
BODY32(k_lshl_v, "v_lshlrev_b32 %0, %1, %0")
BODY32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
BODY32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
BODY32(k_cmp32, "v_cmp_lt_u32 vcc, %0, %1")
BODY64(k_cmp64, "v_cmp_lt_u64 vcc, %0, %0")
BODY32(k_max_u32, "v_max_u32 %0, %0, %1")
BODY32(k_xnor, "v_xnor_b32 %0, %0, %1")
BODY32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96")
BODY32(k_perm, "v_perm_b32 %0, %0, %1, %1")
BODY32(k_add_f32, "v_add_f32 %0, %0, %1")
BODY32(k_mul_f32, "v_mul_f32 %0, %0, %1")
BODY32(k_fma_f32, "v_fma_f32 %0, %0, %1, %1")
BODY32(k_fmac_f32, "v_fmac_f32 %0, %1, %1")
BODY64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0")
BODY64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0")
BODY32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
BODY32(k_or3, "v_or3_b32 %0, %0, %1, %1")
BODY32(k_xad, "v_xad_u32 %0, %0, %1, %1")
BODY32(k_add_lshl, "v_add_lshl_u32 %0, %0, %1, 3")
BODY32(k_bfrev, "v_bfrev_b32 %0, %0")
BODY32(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
BODY32(k_ashr_c, "v_ashrrev_i32 %0, 3, %0")
BODY32(k_add_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
BODY32(k_add_dpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

struct Ent { const char* name; void (*fn)(uint32_t*, int, uint32_t); int insts; };
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 2048 * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    Ent ents[] = {
        {"v_add_u32", k_add_u32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_lshl_add_u32", k_lshl_add, 1}, {"v_add3_u32", k_add3, 1},
        {"v_bfi_b32", k_xor3, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1},
        {"v_mad_u32_u24", k_mad_u24, 1}, {"v_mul_u32_u24", k_mul_u24, 1}, {"v_min_u32", k_min_u32, 1}, {"v_and_or_b32", k_and_or, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_add_co_u32", k_addco, 1}, {"v_cmp+v_addc (2 insts)", k_cmp_addc, 2},
        {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_mad_u16", k_mad_u16, 1},
        {"v_dot4_u32_u8", k_dot4_u8, 1},
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1},
        {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_pk_mov_b32", k_pk_add_like, 1},
        {"v_and_b32", k_and_b32, 1}, {"v_or_b32", k_or_b32, 1}, {"v_not_b32", k_not_b32, 1}, {"v_mov_b32", k_mov_b32, 1},
        {"v_sub_u32", k_sub_u32, 1}, {"v_lshlrev_b32 const", k_lshl_c, 1}, {"v_lshrrev_b32 const", k_lshr_c, 1}, {"v_lshlrev_b32 vgpr", k_lshl_v, 1},
        {"v_cndmask_b32 vcc", k_cndmask, 1}, {"v_addc_co_u32", k_addc, 1}, {"v_cmp_lt_u32", k_cmp32, 1}, {"v_cmp_lt_u64", k_cmp64, 1},
        {"v_max_u32", k_max_u32, 1}, {"v_xnor_b32", k_xnor, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_perm_b32", k_perm, 1},
        {"v_add_f32", k_add_f32, 1}, {"v_mul_f32", k_mul_f32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_fmac_f32", k_fmac_f32, 1},
        {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_fma_f64", k_fma_f64, 1},
        {"v_lshl_or_b32", k_lshl_or, 1}, {"v_or3_b32", k_or3, 1}, {"v_xad_u32", k_xad, 1}, {"v_add_lshl_u32", k_add_lshl, 1},
        {"v_bfrev_b32", k_bfrev, 1}, {"v_cvt_f32_u32", k_cvt_f32_u32, 1}, {"v_ashrrev_i32 const", k_ashr_c, 1},
        {"v_add_u32_sdwa", k_add_sdwa, 1}, {"v_add_u32_dpp", k_add_dpp, 1},

    };
    const int iters = 256;
    // reference: clocks via wall time; report lane-ops per second and relative cost vs v_add_u32
    double base = 0;
    for (auto& e : ents) {
        float best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(2048), dim3(256), 0, 0, d, iters, 0x9E3779B1u);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        double n = 2048.0 * 256 * iters * UNROLL * CHAINS * e.insts;
        double rate = n / best / 1e9;   // T lane-insts/s
        if (base == 0) base = rate;
        printf("%-26s %8.3f ms  %7.2f T lane-inst/s  cost %.2f x v_add_u32\n", e.name, best, rate, base / rate);
    }
    return 0;
}
