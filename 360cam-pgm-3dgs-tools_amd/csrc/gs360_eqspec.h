// gs360_eqspec.h -- EQ-SPEC v1 device helpers shared by the 8-bit kernels (gs360_kernels.hip) and the 16-bit kernels
// (gs360_u16.hip).  Every function is one IEEE binary32 operation per step (compile with -ffp-contract=off); the CPU
// CPU checker restates the same sequence independently.
#pragma once
#include "gs360_kernels.h"

namespace gs360 {

// ------------------------------------------------------------------------------------------------
// EQ-SPEC v1
// ------------------------------------------------------------------------------------------------
#define EQ_T8 0x1.a8279ap-2f
#define EQ_C1 (-0.33333316445350647f)
#define EQ_C2 (0.199985072016716f)
#define EQ_C3 (-0.14244139194488525f)
#define EQ_C4 (0.10597943514585495f)
#define EQ_C5 (-0.06087981536984444f)

// n / d, correctly rounded, for EQ-SPEC's operands: d a NORMAL float in [2^-100, 2^100]; n = 0 or 2^-103 <= |n| <= 2 d; the
// quotient zero or normal.  This is the IEEE sequence the compiler emits for `n / d`
// (v_div_scale x2, v_rcp, Newton step on the reciprocal, quotient + two residual corrections, v_div_fmas, v_div_fixup) with
// the three instructions that are the identity on this domain removed: v_div_scale only rescales when an operand, the
// reciprocal or the quotient is denormal, the numerator is below 2^-103 or the exponents differ by >= 96, and v_div_fixup only
// patches NaN / inf / zero divisors.  In EQ-SPEC d is max(|a|, |b|) or a sum of two magnitudes of ray components (>= 6e-29
// even for the centre pixel of a pole view), the divisor-zero case is replaced by d = 1 before the call, and a NON-ZERO
// numerator is a ray component or a difference of two: pixel coordinates are >= 2.7e-10 or exactly 0, a cancelling
// fma(sin p, yv, cos p) is a multiple of 2^-47 |cos p| with |cos p| >= 2^-32 whenever yv can cancel it (else it is cos p >= 6e-17
// itself, times a polynomial value that is 0 or >= 6e-8 in fisheye mode) -- never below 2^-101.  gs360_selftest_arith checks the two forms against
// each other on 10^9 operand sets of this domain (a numerator of 2^-125 over a divisor of 2^-99 does differ).
// Same bits as `/`, 8 instead of 11 instructions, two divisions per pixel pair.
__device__ __forceinline__ float eq_div(float n, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}

// sqrt(x), correctly rounded, x >= 0 finite.  For x >= 2^-96 this is the generic IEEE sequence (v_sqrt_f32 + the
// two-sided one-ulp correction) without its input scaling for tiny operands and without the inf / zero class test; a
// wavefront in which ANY lane holds a smaller operand (x = 0, or the centre pixel of a pole view: x^2 + b^2 ~ 4e-33)
// takes the generic sequence as a whole (wave-uniform branch), so results are the same bits everywhere.
__device__ __forceinline__ float eq_sqrt_normal(float x) {   // x in [2^-96, 2^96]: no fallback needed
    const float s = __builtin_amdgcn_sqrtf(x);
    const int si = __builtin_bit_cast(int, s);
    const float sd = __builtin_bit_cast(float, si - 1), su = __builtin_bit_cast(float, si + 1);
    const float lo = __builtin_fmaf(-sd, s, x) <= 0.0f ? sd : s;
    return __builtin_fmaf(-su, s, x) > 0.0f ? su : lo;
}
__device__ __forceinline__ float eq_sqrt(float x) {
    if (__builtin_expect(__any(!(x >= 0x1p-96f)), 0)) return __builtin_sqrtf(x);
    const float s = __builtin_amdgcn_sqrtf(x);
    const int si = __builtin_bit_cast(int, s);
    const float sd = __builtin_bit_cast(float, si - 1), su = __builtin_bit_cast(float, si + 1);
    const float lo = __builtin_fmaf(-sd, s, x) <= 0.0f ? sd : s;
    return __builtin_fmaf(-su, s, x) > 0.0f ? su : lo;
}

// atan2(yy, xx) = r0 + K * pi/4 with |r0| <= pi/8 (sign folded into r0), K in [-4, 4].
// XPOS: the caller guarantees xx >= 0 (latitude: xx is a square root), which drops the left-half-plane fix-up.
template <bool XPOS = false>
__device__ __forceinline__ float eq_atan2_red(float yy, float xx, int& K) {
    float ax = __builtin_fabsf(xx), ay = __builtin_fabsf(yy);
    bool steep = ay > ax;
    float mx = steep ? ay : ax, mn = steep ? ax : ay;
    bool big = mn > EQ_T8 * mx;
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float t = eq_div(num, den > 0.0f ? den : 1.0f);   // den == 0 only when num == 0: same value as the spec's guard, no branch
    float z = t * t;
    float p = __builtin_fmaf(EQ_C5, z, EQ_C4);
    p = __builtin_fmaf(p, z, EQ_C3);
    p = __builtin_fmaf(p, z, EQ_C2);
    p = __builtin_fmaf(p, z, EQ_C1);
    float r0 = __builtin_fmaf(p * z, t, t);
    int k = big ? 1 : 0;
    if (steep) { r0 = -r0; k = 2 - k; }
    if constexpr (!XPOS) {
        if (xx < 0.0f) { r0 = -r0; k = 4 - k; }
    }
    if (yy < 0.0f) { r0 = -r0; k = -k; }
    K = k;
    return r0;
}

// Equidistant-fisheye output: S(q) = sin(pi r/2)/r and C(q) = cos(pi r/2) as degree-8 polynomials in q = r^2 on [0, 4]
// (same coefficients and Horner order as the oracle).
static __device__ __constant__ const float kEqFishS[9] = {1.5707963705062866f, -0.6459640860557556f, 0.07969262450933456f,
                                                   -0.004681753925979137f, 0.0001604411081643775f, -3.598792090997449e-06f,
                                                   5.689994608815141e-08f, -6.633614213491512e-10f, 5.326020006968246e-12f};
static __device__ __constant__ const float kEqFishC[9] = {1.0f, -1.2337005138397217f, 0.25366950035095215f, -0.020863480865955353f,
                                                   0.0009192594443447888f, -2.5201432436006144e-05f, 4.708266487796209e-07f,
                                                   -6.321354106830768e-09f, 5.675555858619674e-11f};
__device__ __forceinline__ float eq_poly8(const float* k, float q) {
    float p = k[8];
#pragma unroll
    for (int n = 7; n >= 0; --n) p = __builtin_fmaf(p, q, k[n]);
    return p;
}

// quantised longitude coordinate (1/32 px, wrapped to [0, 32W)), in two steps: the part every view of a yaw ring shares
// (in [-18W, 18W + 32]) and the view's own integer offset x0i32 in [0, 32W) with the single wrap.  Integer addition is
// associative, so the split is the per-view formula bit for bit.
__device__ __forceinline__ int eq_lon_base(float r0, int K, const EqLaunch& L, float x0f32) {
    return (int)__builtin_rintf(__builtin_fmaf(r0, L.kx32, x0f32)) + K * 4 * L.W;
}
__device__ __forceinline__ int eq_lon_wrap(int base, int x0i32, int W32) {
    int sx = base + x0i32;
    if (sx < 0) sx += W32;
    if (sx >= W32) sx -= W32;
    return sx;
}
// The same wrap for the ring-member loop in three operations: with the shared part reduced to [0, 32W) once
// (eq_lon_norm: it lies in (-32W, 32W) because 18W + 32 < 32W for W >= 3), base + x0i32 is in [0, 64W) and the unsigned minimum
// of t and t - 32W is t mod 32W.
__device__ __forceinline__ int eq_lon_norm(int base, int W32) { return base + ((base >> 31) & W32); }
__device__ __forceinline__ int eq_lon_member(int base_norm, int x0i32, int W32) {
    const uint32_t t = (uint32_t)(base_norm + x0i32);
    return (int)min(t, t - (uint32_t)W32);
}
__device__ __forceinline__ int eq_quant_lon(float r0, int K, const EqLaunch& L, const EqView& V) {
    return eq_lon_wrap(eq_lon_base(r0, K, L, V.x0f32), V.x0i32, 32 * L.W);
}

}  // namespace gs360
