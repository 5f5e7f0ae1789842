// gs360_cvremap.h -- cv2.remap's coordinate conversion and its straight-line bilinear sampler (BORDER_CONSTANT), shared by the table
// kernels (gs360_table.hip) and the LDS-staged table kernel (gs360_tablestage.hip).  DF:2001-2008 / DF:1198-1205 are the call sites.
#pragma once
#include "gs360_sampler.h"

namespace gs360 {

__device__ __forceinline__ int cv_round(float v) {  // SSE cvtss2si: half-to-even, indefinite -> INT_MIN
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
    return (int)__builtin_rintf(v);
}
__device__ __forceinline__ int sat_s16(int v) { return min(max(v, -32768), 32767); }

template <int C>
__device__ __forceinline__ void cv_sample_linear(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                 float mx, float my, const uint8_t (&cval)[4], uint32_t (&out)[4]) {
    // Straight-line formulation (single exit): taps are fetched from clamped, always-valid addresses and
    // replaced by the border constant afterwards, exactly reproducing remapBilinear's BORDER_CONSTANT rule.
    int sx = cv_round(mx * 32.0f), sy = cv_round(my * 32.0f);
    int fx = sx & 31, fy = sy & 31;
    int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
    bool outside = ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0;
    uint32_t a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
    uint32_t w00 = a0 * b0, w01 = a1 * b0, w10 = a0 * b1, w11 = a1 * b1;
    bool x0in = (unsigned)ix < (unsigned)W, x1in = (unsigned)(ix + 1) < (unsigned)W;
    bool y0in = (unsigned)iy < (unsigned)H, y1in = (unsigned)(iy + 1) < (unsigned)H;
    int xa = min(max(ix, 0), W - 1), xb = min(max(ix + 1, 0), W - 1);
    int ya = min(max(iy, 0), H - 1), yb = min(max(iy + 1, 0), H - 1);
    const uint8_t* ra = src + (int64_t)ya * stride;
    const uint8_t* rb = src + (int64_t)yb * stride;
    uint32_t s00[4], s01[4], s10[4], s11[4];
    bool wide = false;
    if constexpr (C == 3) wide = x0in && y0in && ix < W - 2 && y1in;  // 8-byte reads stay inside the buffer
    if (wide) {
        uint2 t0 = ld_u64(ra + 3 * xa), t1 = ld_u64(rb + 3 * xa);
        s00[0] = byte_of(t0.x, 0); s00[1] = byte_of(t0.x, 1); s00[2] = byte_of(t0.x, 2);
        s01[0] = byte_of(t0.x, 3); s01[1] = byte_of(t0.y, 0); s01[2] = byte_of(t0.y, 1);
        s10[0] = byte_of(t1.x, 0); s10[1] = byte_of(t1.x, 1); s10[2] = byte_of(t1.x, 2);
        s11[0] = byte_of(t1.x, 3); s11[1] = byte_of(t1.y, 0); s11[2] = byte_of(t1.y, 1);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            s00[c] = ra[xa * C + c]; s01[c] = ra[xb * C + c];
            s10[c] = rb[xa * C + c]; s11[c] = rb[xb * C + c];
        }
    }
    bool in00 = x0in && y0in, in01 = x1in && y0in, in10 = x0in && y1in, in11 = x1in && y1in;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        uint32_t cv = cval[c];
        uint32_t v = blend(in00 ? s00[c] : cv, in01 ? s01[c] : cv, in10 ? s10[c] : cv, in11 ? s11[c] : cv,
                           w00, w01, w10, w11);
        out[c] = outside ? cv : v;
    }
}

}  // namespace gs360
