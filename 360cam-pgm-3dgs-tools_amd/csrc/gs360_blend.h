// gs360_blend.h -- the exact-integer bilinear blend of RGB tap rows shared by the gather / staged kernels (gs360_kernels.hip) and the
// source-major kernel (gs360_srcmajor.hip): OpenCV's 8-bit fixed-point bilinear arithmetic, (sum S a b + 512) >> 10.
#pragma once
#include "gs360_kernels.h"

namespace gs360 {

// Integer multiply-adds as v_dot2_i32_i16: v_perm_b32 gathers two tap bytes into a zero-extended 16-bit pair and one
// dot instruction multiplies both by a packed pair of weights and accumulates.  Exact integer arithmetic.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2_i16(uint32_t taps, uint32_t weights, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, taps), __builtin_bit_cast(s16x2, weights), acc, false);
}
// First multiply-add of a chain: the start value (rounding constant) rides in a SCALAR register as the VOP3P form's third operand.
// Left to the compiler a constant start becomes v_mov + the accumulate-in-place VOP2 form -- one more vector instruction per chain.
__device__ __forceinline__ int dot2_i16_from(uint32_t taps, uint32_t weights, int start_uniform) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(taps), "v"(weights), "s"(start_uniform));
    return r;
}
// v_perm_b32(a, b, sel) selector: result = (0, hi, 0, lo) where lo / hi index the bytes of {a (4..7), b (0..3)}
#define GS360_PAIR(lo, hi) (0x0c000c00u | ((uint32_t)(hi) << 16) | (uint32_t)(lo))

// One RGB pixel from its two tap rows (t0 = row iy, t1 = row iy + 1; row bytes: r0 g0 b0 r1 | g1 b1 . .) and the 1/32-pixel phases
// fx, fy in [0, 31]: (sum S a b + 512) >> 10 with a in {32 - fx, fx}, b in {32 - fy, fy}.  The weights of one row, a0 b | (a1 b) << 16,
// are one multiply of the packed horizontal pair (a1 b <= 1024 cannot carry into the upper half).
__device__ __forceinline__ void blend_rgb_rows(const uint2 t0, const uint2 t1, const int fx, const int fy, uint32_t (&out)[3]) {
    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);             // < 2^22
    const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
    out[0] = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1.x, t1.x, GS360_PAIR(0, 3)), wr1,
                                dot2_i16_from(__builtin_amdgcn_perm(t0.x, t0.x, GS360_PAIR(0, 3)), wr0, 512)) >> 10;
    out[1] = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1.y, t1.x, GS360_PAIR(1, 4)), wr1,
                                dot2_i16_from(__builtin_amdgcn_perm(t0.y, t0.x, GS360_PAIR(1, 4)), wr0, 512)) >> 10;
    out[2] = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1.y, t1.x, GS360_PAIR(2, 5)), wr1,
                                dot2_i16_from(__builtin_amdgcn_perm(t0.y, t0.x, GS360_PAIR(2, 5)), wr0, 512)) >> 10;
}

}  // namespace gs360
