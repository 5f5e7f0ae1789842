// gs360_kernels.hip -- hand-written gfx950 kernels for the 360PerspCut reprojection hot path.
//
//   eq_views_kernel      equirect -> rectilinear views, analytic in-kernel map (EQ-SPEC v1), replaces the
//                        per-view ffmpeg v360 processes of cli_tools/gs360_360PerspCut.py:310-314.
//   table_remap_kernel   cv2.remap(INTER_LINEAR|INTER_NEAREST, BORDER_CONSTANT) + valid fill, replaces
//                        cli_tools/gs360_DualFisheyeDistortionCalibration.py:2001-2014 / :2031-2043.
//   fe_views_kernel      fused dual-fisheye -> perspective (FE-SPEC v1), DF:1759-1823 evaluated in-kernel.
//
// All three are HBM/L2-bound byte gathers (no contraction -> no MFMA).  Work decomposition: one 32x32
// output tile per 256-thread workgroup, each lane owns 4 horizontally adjacent pixels (12 B of RGB =
// three dword stores), the 8 lanes of a tile row cover 32 px, a wavefront covers 8 tile rows.  Tiles are
// numbered row-major per view and handed to XCDs in contiguous chunks (block b runs on XCD b % 8) so that
// neighbouring tiles -- which share source cache lines -- hit the same per-XCD L2.
//
// Compile with -ffp-contract=off: the float32 specs are defined operation by operation and must match the
// CPU oracle bit for bit; only explicit __builtin_fmaf may fuse.
#include "gs360_kernels.h"

namespace gs360 {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);  // unaligned dword load (gfx950 runs in unaligned access mode)
    return v;
}
__device__ __forceinline__ uint2 ld_u64(const uint8_t* p) {
    uint2 v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
__device__ __forceinline__ uint32_t byte_of(uint32_t v, int k) { return (v >> (8 * k)) & 0xffu; }

// bilinear blend of one channel, weights a0+a1 = 32, b0+b1 = 32  ->  (sum + 512) >> 10
__device__ __forceinline__ uint32_t blend(uint32_t s00, uint32_t s01, uint32_t s10, uint32_t s11,
                                          uint32_t w00, uint32_t w01, uint32_t w10, uint32_t w11) {
    return (s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11 + 512u) >> 10;
}

// store 4 pixels x C channels held as px[p][c] to a row pointer; packed dword stores when possible
template <int C>
__device__ __forceinline__ void store_px4(uint8_t* d, const uint32_t (&px)[4][4], int n_valid, bool aligned4) {
    if (n_valid == 4 && aligned4) {
        if constexpr (C == 3) {
            uint32_t* q = reinterpret_cast<uint32_t*>(d);
            q[0] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
            q[1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
            q[2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
            return;
        } else if constexpr (C == 4) {
            uint4 v;
            v.x = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[0][3] << 24);
            v.y = px[1][0] | (px[1][1] << 8) | (px[1][2] << 16) | (px[1][3] << 24);
            v.z = px[2][0] | (px[2][1] << 8) | (px[2][2] << 16) | (px[2][3] << 24);
            v.w = px[3][0] | (px[3][1] << 8) | (px[3][2] << 16) | (px[3][3] << 24);
            *reinterpret_cast<uint4*>(d) = v;  // aligned4 guarantees 16-B alignment for C == 4 (see host)
            return;
        } else if constexpr (C == 1) {
            *reinterpret_cast<uint32_t*>(d) = px[0][0] | (px[1][0] << 8) | (px[2][0] << 16) | (px[3][0] << 24);
            return;
        }
    }
    for (int p = 0; p < n_valid; ++p)
        for (int c = 0; c < C; ++c) d[p * C + c] = (uint8_t)px[p][c];
}

// ------------------------------------------------------------------------------------------------
// EQ-SPEC v1
// ------------------------------------------------------------------------------------------------
#define EQ_T8 0x1.a8279ap-2f
#define EQ_C1 (-0.33333316445350647f)
#define EQ_C2 (0.199985072016716f)
#define EQ_C3 (-0.14244139194488525f)
#define EQ_C4 (0.10597943514585495f)
#define EQ_C5 (-0.06087981536984444f)

// atan2(yy, xx) = r0 + K * pi/4 with |r0| <= pi/8 (sign folded into r0), K in [-4, 4]
__device__ __forceinline__ float eq_atan2_red(float yy, float xx, int& K) {
    float ax = __builtin_fabsf(xx), ay = __builtin_fabsf(yy);
    bool steep = ay > ax;
    float mx = steep ? ay : ax, mn = steep ? ax : ay;
    bool big = mn > EQ_T8 * mx;
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float t = den > 0.0f ? num / den : 0.0f;
    float z = t * t;
    float p = __builtin_fmaf(EQ_C5, z, EQ_C4);
    p = __builtin_fmaf(p, z, EQ_C3);
    p = __builtin_fmaf(p, z, EQ_C2);
    p = __builtin_fmaf(p, z, EQ_C1);
    float r0 = __builtin_fmaf(p * z, t, t);
    int k = big ? 1 : 0;
    if (steep) { r0 = -r0; k = 2 - k; }
    if (xx < 0.0f) { r0 = -r0; k = 4 - k; }
    if (yy < 0.0f) { r0 = -r0; k = -k; }
    K = k;
    return r0;
}

template <int C>
__device__ __forceinline__ void eq_sample(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                          int sx, int sy, uint32_t (&out)[4]) {
    int fx = sx & 31, ix = sx >> 5;
    int fy = sy & 31, iy = sy >> 5;
    int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
    const uint8_t* r0 = src + (int64_t)y0 * stride;
    const uint8_t* r1 = src + (int64_t)y1 * stride;
    uint32_t a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
    uint32_t w00 = a0 * b0, w01 = a1 * b0, w10 = a0 * b1, w11 = a1 * b1;
    uint32_t s00[4], s01[4], s10[4], s11[4];
    bool wide = false;
    if constexpr (C == 3) wide = ix < W - 2;  // both taps in-row and the 8-byte read stays inside the row
    if (wide) {
        uint2 t0 = ld_u64(r0 + 3 * ix), t1 = ld_u64(r1 + 3 * ix);
        s00[0] = byte_of(t0.x, 0); s00[1] = byte_of(t0.x, 1); s00[2] = byte_of(t0.x, 2);
        s01[0] = byte_of(t0.x, 3); s01[1] = byte_of(t0.y, 0); s01[2] = byte_of(t0.y, 1);
        s10[0] = byte_of(t1.x, 0); s10[1] = byte_of(t1.x, 1); s10[2] = byte_of(t1.x, 2);
        s11[0] = byte_of(t1.x, 3); s11[1] = byte_of(t1.y, 0); s11[2] = byte_of(t1.y, 1);
    } else {
        int ix1 = (ix + 1 == W) ? 0 : ix + 1;  // horizontal wrap
#pragma unroll
        for (int c = 0; c < C; ++c) {
            s00[c] = r0[ix * C + c]; s01[c] = r0[ix1 * C + c];
            s10[c] = r1[ix * C + c]; s11[c] = r1[ix1 * C + c];
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = blend(s00[c], s01[c], s10[c], s11[c], w00, w01, w10, w11);
}

template <int C>
__global__ __launch_bounds__(256) void eq_views_kernel(const EqLaunch L) {
    // XCD-aware tile order: XCD x (= blockIdx % 8) walks tiles [x*chunk, (x+1)*chunk)
    int b = blockIdx.x;
    int t = (b & 7) * L.chunk + (b >> 3);
    if (t >= L.total_tiles) return;
    int f = t / L.tiles_per_frame;
    int r = t - f * L.tiles_per_frame;
    int k = 0;
    while (k + 1 < L.n_views && r >= L.view[k + 1].tile_base) ++k;
    const EqView& V = L.view[k];
    r -= V.tile_base;
    int tile_y = r / V.tiles_x, tile_x = r - tile_y * V.tiles_x;

    int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    int x0 = tile_x * 32 + tx * 4, y = tile_y * 32 + ty;
    if (y >= V.out_h || x0 >= V.out_w) return;

    const uint8_t* __restrict__ src = L.src[f];
    int64_t dstride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * C;
    uint8_t* drow = L.dst[f * L.n_views + k] + (int64_t)y * dstride + (int64_t)x0 * C;

    float yv = (float)(2 * y + 1 - V.out_h) * V.syv;
    float bz = __builtin_fmaf(V.sp, yv, V.cp);    // forward component after pitch
    float cy = __builtin_fmaf(-V.cp, yv, V.sp);   // up component after pitch
    float bb = bz * bz;
    int W32 = 32 * L.W;

    uint32_t px[4][4];
    int n_valid = min(4, V.out_w - x0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p < n_valid) {
            float x = (float)(2 * (x0 + p) + 1 - V.out_w) * V.sxu;
            float h = __builtin_sqrtf(__builtin_fmaf(x, x, bb));
            int Kl, Kt;
            float rl = eq_atan2_red(x, bz, Kl);
            float rt = eq_atan2_red(cy, h, Kt);
            int sx = (int)__builtin_rintf(__builtin_fmaf(rl, L.kx32, V.x0f32)) + V.x0i32 + Kl * 4 * L.W;
            if (sx < 0) sx += W32;
            if (sx >= W32) sx -= W32;
            int sy = L.y0i32 - Kt * 8 * L.H - (int)__builtin_rintf(rt * L.ky32);
            eq_sample<C>(src, L.src_stride, L.W, L.H, sx, sy, px[p]);
        }
    }
    bool aligned4 = (C == 4) ? ((dstride & 15) == 0) : ((dstride & 3) == 0);
    store_px4<C>(drow, px, n_valid, aligned4);
}

// ------------------------------------------------------------------------------------------------
// cv2.remap semantics (shared by the table kernel and the fused fisheye kernel)
// ------------------------------------------------------------------------------------------------
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

template <int C>
__device__ __forceinline__ void cv_sample_nearest(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                  float mx, float my, const uint8_t (&cval)[4], uint32_t (&out)[4]) {
    int ix = sat_s16(cv_round(mx)), iy = sat_s16(cv_round(my));
    bool inside = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
    int xa = min(max(ix, 0), W - 1), ya = min(max(iy, 0), H - 1);
    const uint8_t* s = src + (int64_t)ya * stride + (int64_t)xa * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        uint32_t v = s[c];
        out[c] = inside ? v : (uint32_t)cval[c];
    }
}

template <int C>
__global__ __launch_bounds__(256) void table_remap_kernel(const TableLaunch L, int tiles_x, int total_tiles, int chunk) {
    int b = blockIdx.x;
    int t = (b & 7) * chunk + (b >> 3);
    if (t >= total_tiles) return;
    int tile_y = t / tiles_x, tile_x = t - tile_y * tiles_x;
    int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    int x0 = tile_x * 32 + tx * 4, y = tile_y * 32 + ty;
    if (y >= L.h || x0 >= L.w) return;
    int n_valid = min(4, L.w - x0);
    int64_t o = (int64_t)y * L.w + x0;
    float mx[4], my[4];
    if (n_valid == 4 && (L.w & 3) == 0) {  // 16-B coalesced map reads
        float4 vx = *reinterpret_cast<const float4*>(L.map_x + o);
        float4 vy = *reinterpret_cast<const float4*>(L.map_y + o);
        mx[0] = vx.x; mx[1] = vx.y; mx[2] = vx.z; mx[3] = vx.w;
        my[0] = vy.x; my[1] = vy.y; my[2] = vy.z; my[3] = vy.w;
    } else {
        for (int p = 0; p < 4; ++p) {
            mx[p] = p < n_valid ? L.map_x[o + p] : 0.0f;
            my[p] = p < n_valid ? L.map_y[o + p] : 0.0f;
        }
    }
    uint32_t px[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p < n_valid) {
            if (L.interp == GS360_INTERP_LINEAR) cv_sample_linear<C>(L.src, L.src_stride, L.W, L.H, mx[p], my[p], L.cval, px[p]);
            else cv_sample_nearest<C>(L.src, L.src_stride, L.W, L.H, mx[p], my[p], L.cval, px[p]);
            if (L.valid && !L.valid[o + p]) {
#pragma unroll
                for (int c = 0; c < C; ++c) px[p][c] = (uint32_t)L.fill;
            }
        }
    }
    bool aligned4 = (C == 4) ? ((L.dst_stride & 15) == 0) : ((L.dst_stride & 3) == 0);
    store_px4<C>(L.dst + (int64_t)y * L.dst_stride + (int64_t)x0 * C, px, n_valid, aligned4);
}

// ------------------------------------------------------------------------------------------------
// FE-SPEC v1: fused fisheye -> perspective
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void fe_views_kernel(const FeLaunch L) {
    int b = blockIdx.x;
    int t = (b & 7) * L.chunk + (b >> 3);
    if (t >= L.total_tiles) return;
    int k = 0;
    while (k + 1 < L.n_views && t >= L.view[k + 1].tile_base) ++k;
    const FeView& V = L.view[k];
    int r = t - V.tile_base;
    int tile_y = r / V.tiles_x, tile_x = r - tile_y * V.tiles_x;
    int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
    int x0 = tile_x * 32 + tx * 4, y = tile_y * 32 + ty;
    if (y >= V.out_h || x0 >= V.out_w) return;
    int n_valid = min(4, V.out_w - x0);
    int64_t dstride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * C;

    float yv = (float)(2 * y + 1 - V.out_h) * V.syv;       // ray y = -yv
    float Y = __builtin_fmaf(-V.cp, yv, V.sp);
    float z1 = __builtin_fmaf(V.sp, yv, V.cp);
    float sz1 = V.sy * z1, cz1 = V.cy * z1;
    float n2y = __builtin_fmaf(yv, yv, 1.0f);

    uint32_t px[4][4];
    uint32_t vmask = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (p < n_valid) {
            float x = (float)(2 * (x0 + p) + 1 - V.out_w) * V.sxu;
            float X = __builtin_fmaf(V.cy, x, sz1);
            float Z = __builtin_fmaf(-V.sy, x, cz1);
            float N = __builtin_sqrtf(__builtin_fmaf(x, x, n2y));
            float d = N * (N + Z);
            float s = d > 0.0f ? __builtin_sqrtf(2.0f / d) : 0.0f;
            float xn = X * s, yn = -(Y * s);
            float r2 = __builtin_fmaf(xn, xn, yn * yn);
            float r4 = r2 * r2;
            float radial = __builtin_fmaf(V.k4, r4 * r4, __builtin_fmaf(V.k3, r4 * r2,
                           __builtin_fmaf(V.k2, r4, __builtin_fmaf(V.k1, r2, 1.0f))));
            float xd = xn * radial, yd = yn * radial;
            if (V.tang) {
                float xy = xn * yn;
                xd = __builtin_fmaf(V.tp2, xy, __builtin_fmaf(V.p1, __builtin_fmaf(2.0f * xn, xn, r2), xd));
                yd = __builtin_fmaf(V.tp1, xy, __builtin_fmaf(V.p2, __builtin_fmaf(2.0f * yn, yn, r2), yd));
            }
            float mx = __builtin_fmaf(yd, V.b2, __builtin_fmaf(xd, V.b1, __builtin_fmaf(xd, V.f, V.cx0)));
            float my = __builtin_fmaf(yd, V.f, V.cy0);
            bool ok = (Z >= V.cos_tmax * N) && (mx >= 0.0f) && (mx <= V.wmax) && (my >= 0.0f) && (my <= V.hmax);
            if (L.interp == GS360_INTERP_LINEAR) cv_sample_linear<C>(V.src, L.src_stride, V.W, V.H, mx, my, L.cval, px[p]);
            else cv_sample_nearest<C>(V.src, L.src_stride, V.W, V.H, mx, my, L.cval, px[p]);
            if (!ok && L.mask_outside) {
#pragma unroll
                for (int c = 0; c < C; ++c) px[p][c] = (uint32_t)L.mask_value;
            }
            vmask |= (ok ? 1u : 0u) << (8 * p);
        }
    }
    bool aligned4 = (C == 4) ? ((dstride & 15) == 0) : ((dstride & 3) == 0);
    store_px4<C>(V.dst + (int64_t)y * dstride + (int64_t)x0 * C, px, n_valid, aligned4);
    if (V.valid_out) {
        uint8_t* vo = V.valid_out + (int64_t)y * V.out_w + x0;
        if (n_valid == 4 && (V.out_w & 3) == 0) *reinterpret_cast<uint32_t*>(vo) = vmask;
        else for (int p = 0; p < n_valid; ++p) vo[p] = (uint8_t)((vmask >> (8 * p)) & 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
hipError_t launch_equirect(const EqLaunch& L, int C, hipStream_t s) {
    dim3 grid((unsigned)(L.chunk * 8)), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL(eq_views_kernel<1>, grid, block, 0, s, L); break;
        case 3: hipLaunchKernelGGL(eq_views_kernel<3>, grid, block, 0, s, L); break;
        case 4: hipLaunchKernelGGL(eq_views_kernel<4>, grid, block, 0, s, L); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_table(const TableLaunch& L, int C, hipStream_t s) {
    int tiles_x = (L.w + 31) / 32, tiles_y = (L.h + 31) / 32;
    int total = tiles_x * tiles_y, chunk = (total + 7) / 8;
    dim3 grid((unsigned)(chunk * 8)), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL(table_remap_kernel<1>, grid, block, 0, s, L, tiles_x, total, chunk); break;
        case 3: hipLaunchKernelGGL(table_remap_kernel<3>, grid, block, 0, s, L, tiles_x, total, chunk); break;
        case 4: hipLaunchKernelGGL(table_remap_kernel<4>, grid, block, 0, s, L, tiles_x, total, chunk); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_fisheye(const FeLaunch& L, int C, hipStream_t s) {
    dim3 grid((unsigned)(L.chunk * 8)), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL(fe_views_kernel<1>, grid, block, 0, s, L); break;
        case 3: hipLaunchKernelGGL(fe_views_kernel<3>, grid, block, 0, s, L); break;
        case 4: hipLaunchKernelGGL(fe_views_kernel<4>, grid, block, 0, s, L); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gs360
