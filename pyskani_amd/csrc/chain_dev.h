// Device helpers shared by the translation units that chain anchors (join.hip / dp.hip / select.hip / reduce.hip: the batched path; small_query.hip: the
// one-launch-sequence query of a small genome). Header-only: every translation unit gets its own copy (no relocatable device code).
#pragma once
#include "common.h"

struct MarkerSet { const uint64_t* p; uint32_t n; uint32_t pad; };

// LDS hand-off between lanes of ONE wave: order the ds ops, no workgroup barrier
__device__ __forceinline__ void lds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// u = q - r', r' the strand-signed reference position (-r on the reverse strand): the anchor's diagonal. With it the gap of a
// pair is |ux - uy| and dr = dq - (ux - uy): three instructions fewer per (anchor, predecessor) pair than from q and r.
struct LaneAnchor { uint32_t q, u, m; int32_t f; };
__device__ __forceinline__ uint32_t lane_diag(uint32_t qx, uint32_t rx, uint32_t sg) { return qx - ((rx ^ sg) - sg); }

// key of predecessor y for anchor x at distance d, NEGATIVE when y is not chainable; same rule as the wave kernel and the oracle.
// The DP kernel's time is its VALU instruction count (profiles/r3/r3a_chain_lane20_counters.md: 65 % of all issue cycles at three
// waves per SIMD, the rest waits), so the step is written for it: a predecessor is kept as (q + 1, diagonal, contig | strand,
// score - 1) - the two "- 1" of the range tests are paid once per anchor instead of once per pair -, every requirement is a sign
// bit, and the verdict is the key's own sign (one v_and_or) so that the running maximum, taken signed, skips what is not chainable.
// ISA per (anchor, predecessor) pair: 8 v_sub, 3 v_or3, v_xor, v_lshl_add, v_and_or, 2 v_max = 17 instructions / 46 issue cycles;
// the first version had 20 / 70 (its mask came out as v_cmp + v_cndmask, the slowest VALU instruction there is: 32.8 -> 29.3 ms).
struct LanePred { uint32_t q1, u, m; int32_t f1; };
__device__ __forceinline__ int32_t lane_eval2(uint32_t qx, uint32_t ux, uint32_t mx, const LanePred& y, int d) {
    const int32_t a = (int32_t)(qx - y.q1);                               // dq - 1
    const int32_t t = (int32_t)(ux - y.u), nt = (int32_t)(y.u - ux);      // dq - dr (strand -: dr = ry - rx)
    const int32_t gap = t > nt ? t : nt;
    const int32_t b = a - t;                                              // dr - 1
    const int32_t s1 = y.f1 - gap;                                        // score - ANCHOR_SCORE2 - 1
    const uint32_t z = y.m ^ mx;
    // 1 <= dq <= 2500, dr >= 1, gap <= 300, score > 40, same ref contig and strand
    const uint32_t bad = (uint32_t)a | (uint32_t)(BP_CHAIN_BAND - 1 - a) | (uint32_t)b | (uint32_t)(MAX_GAP_LENGTH - gap) | (uint32_t)s1 | z | (0u - z);
    const uint32_t key = ((uint32_t)s1 << 7) + ((((uint32_t)ANCHOR_SCORE2 + 1u) << 7) | (127u - (uint32_t)d));      // scores stay below 2^20 (a chunk holds < 16 384 anchors): the key's sign bit is free
    return (int32_t)(key | (bad & 0x80000000u));
}

