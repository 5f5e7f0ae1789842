// gs360_u16.hip -- 16-bit (uint16) variants of the two samplers.
//
//   eq_views_u16_kernel      EQ-SPEC v1 on 16-bit equirect sources: 16-bit stills keep their depth through the reference's
//                            ffmpeg path (PC:327-347 sets no -pix_fmt for PNG/TIFF stills) and > 8-bit videos leave as
//                            rgb48le (PC:343-347).  Same quantised coordinates as the 8-bit kernel, evaluated per pixel
//                            (eq_coord_px); bilinear (sum S a b + 512) >> 10, bicubic with the fixed-point Keys table in
//                            64-bit, columns wrap, rows clamp.
//   table_remap_u16_kernel   cv2.remap on CV_16U sources (DF:735 keeps 16-bit inputs at native depth; DF:2001-2014):
//                            OpenCV routes ushort to its FLOAT-weight samplers -- 2-D weight = cy[k1] * cx[k2] in float32
//                            from the 1-D phase tables, float32 accumulation in OpenCV's expression order, cvRound +
//                            saturate to [0, 65535].  Order (compiled with -ffp-contract=off): window inside the image --
//                            bilinear ((S00 w0 + S01 w1) + S10 w2) + S11 w3; bicubic / lanczos4 sum each window row left to
//                            right and add the row sums row by row; window on the border (BORDER_CONSTANT) -- bilinear replaces
//                            outside taps by the border value, bicubic / lanczos4 start from the border value cv and add
//                            (S - cv) w for every in-image tap in row-major order.
//
// First-cut kernels: one output pixel per lane per row slot, straight-line samplers, 2-byte-element gathers.  They are
// HBM-bound byte gathers like their 8-bit siblings but carry none of the tuned fetch paths yet.
#include "gs360_eqspec.h"

namespace gs360 {

namespace {

__device__ __forceinline__ int cv_round_u16(float v) {
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
    return (int)__builtin_rintf(v);
}
__device__ __forceinline__ int sat_s16_u16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ uint16_t sat_u16(float v) { return (uint16_t)min(max(cv_round_u16(v), 0), 65535); }

template <int C, bool CUBIC>
__global__ __launch_bounds__(64 * kWaves) void eq_views_u16_kernel(const EqLaunch L) {
    int b = blockIdx.x;
    int t = (b & 7) * L.chunk + (b >> 3);
    if (t >= L.total_tiles) return;
    int f = t / L.tiles_per_frame;
    int r = t - f * L.tiles_per_frame;
    int k = 0;
    while (k + 1 < L.n_views && r >= L.view[k + 1].tile_base) ++k;
    const EqView& V = L.view[k];
    r -= V.tile_base;
    const int tile_y = r / V.tiles_x, tile_x = r - tile_y * V.tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = tile_x * kTileW + lane;
    if (i >= V.out_w) return;
    const uint16_t* __restrict__ src = reinterpret_cast<const uint16_t*>(L.src[f]);
    const size_t ss = (size_t)L.src_stride >> 1;                       // elements per source row
    const int64_t dstride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * C * 2;
    uint8_t* dst = L.dst[f * L.n_views + k];
    const int W = L.W, H = L.H;
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s) {
        const int j = tile_y * kTileH + wave * kRowsPerWave + s;
        if (j >= V.out_h) break;
        int sx, sy;
        eq_coord_px(L, V, i, j, sx, sy);
        const int fx = sx & 31, ix = sx >> 5, fy = sy & 31, iy = sy >> 5;
        uint16_t* out = reinterpret_cast<uint16_t*>(dst + (int64_t)j * dstride) + (size_t)i * C;
        if constexpr (!CUBIC) {
            const int ix1 = (ix + 1 == W) ? 0 : ix + 1;
            const int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
            const uint16_t* r0 = src + (size_t)y0 * ss;
            const uint16_t* r1 = src + (size_t)y1 * ss;
            const uint32_t a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
            if constexpr (C == 3) {
                // the two RGB taps of a row are 12 contiguous bytes: one dword-aligned 16-byte read per row (the frame base
                // is 4-byte aligned, device buffers carry 64 bytes of slack) shifted into place, instead of six 2-byte loads
                if (ix1 != 0 && ix + 3 <= W && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)L.src_stride) & 3) == 0) {
                    uint32_t d[2][3];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const uint16_t* p = (rr ? r1 : r0) + (size_t)ix * 3;
                        const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;        // 0 or 2
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(reinterpret_cast<const uint8_t*>(p) - o, 4));
                        const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                        d[rr][0] = __builtin_amdgcn_alignbyte(q1, q0, o);
                        d[rr][1] = __builtin_amdgcn_alignbyte(q2, q1, o);
                        d[rr][2] = __builtin_amdgcn_alignbyte(q3, q2, o);
                    }
                    // samples: R0 G0 | B0 R1 | G1 B1
                    const uint32_t s00[3] = {d[0][0] & 0xffffu, d[0][0] >> 16, d[0][1] & 0xffffu};
                    const uint32_t s01[3] = {d[0][1] >> 16, d[0][2] & 0xffffu, d[0][2] >> 16};
                    const uint32_t s10[3] = {d[1][0] & 0xffffu, d[1][0] >> 16, d[1][1] & 0xffffu};
                    const uint32_t s11[3] = {d[1][1] >> 16, d[1][2] & 0xffffu, d[1][2] >> 16};
#pragma unroll
                    for (int c = 0; c < 3; ++c) out[c] = (uint16_t)(((s00[c] * a0 + s01[c] * a1) * b0 + (s10[c] * a0 + s11[c] * a1) * b1 + 512u) >> 10);
                    continue;
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const uint32_t acc = ((uint32_t)r0[ix * C + c] * a0 + (uint32_t)r0[ix1 * C + c] * a1) * b0 +
                                     ((uint32_t)r1[ix * C + c] * a0 + (uint32_t)r1[ix1 * C + c] * a1) * b1;
                out[c] = (uint16_t)((acc + 512u) >> 10);
            }
        } else {
            const int16_t* wt = L.cubic_tab + (fy * 32 + fx) * 16;
            int cols[4];
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int xx = ix - 1 + kx;
                cols[kx] = xx < 0 ? xx + W : (xx >= W ? xx - W : xx);
            }
            int64_t acc[4] = {0, 0, 0, 0};
            if constexpr (C == 3) {
                // the four RGB taps of a window row are 24 contiguous bytes: seven dwords from the dword boundary below them
                // (dwordx4 + dwordx3) shifted into place, instead of twelve 2-byte loads per row
                if (ix >= 1 && ix + 4 <= W && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)L.src_stride) & 3) == 0) {
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky) {
                        const uint16_t* p = src + (size_t)min(max(iy - 1 + ky, 0), H - 1) * ss + (size_t)(ix - 1) * 3;
                        const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;        // 0 or 2
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(reinterpret_cast<const uint8_t*>(p) - o, 4));
                        uint32_t r[7];
#pragma unroll
                        for (int t = 0; t < 7; ++t) r[t] = q[t];
                        uint32_t d[6];
#pragma unroll
                        for (int t = 0; t < 6; ++t) d[t] = __builtin_amdgcn_alignbyte(r[t + 1], r[t], o);
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const int w = wt[ky * 4 + kx];
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                const int e = kx * 3 + c;                                      // sample index within the row's 12
                                const int v = (int)((e & 1) ? (d[e >> 1] >> 16) : (d[e >> 1] & 0xffffu));
                                acc[c] += (int64_t)(v * w);
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int64_t v = (acc[c] + (1 << 14)) >> 15;
                        out[c] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
                    }
                    continue;
                }
            }
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const uint16_t* row = src + (size_t)min(max(iy - 1 + ky, 0), H - 1) * ss;
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int w = wt[ky * 4 + kx];
#pragma unroll
                    for (int c = 0; c < C; ++c) acc[c] += (int64_t)((int)row[(size_t)cols[kx] * C + c] * w);
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int64_t v = (acc[c] + (1 << 14)) >> 15;
                out[c] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
            }
        }
    }
}

// cv2.remap(CV_16U) -- see the file header.  coef: 32 phases x (2 + 4 + 8) float32 1-D coefficients (linear, cubic, lanczos4).
template <int C>
__global__ __launch_bounds__(64 * kWaves) void table_remap_u16_kernel(const TableLaunch T, const float* __restrict__ coef, const uint16_t c0,
                                                                      const uint16_t c1, const uint16_t c2, const uint16_t c3) {
    const int tiles_x = (T.w + kTileW - 1) / kTileW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = tile_x * kTileW + lane;
    if (x >= T.w) return;
    const uint16_t cval[4] = {c0, c1, c2, c3};
    const uint16_t* __restrict__ src = reinterpret_cast<const uint16_t*>(T.src);
    const size_t ss = (size_t)T.src_stride >> 1;
    const int W = T.W, H = T.H, interp = T.interp;
    const int ks = interp == GS360_INTERP_LINEAR ? 2 : (interp == GS360_INTERP_CUBIC ? 4 : 8);
    const float* tab = coef + (interp == GS360_INTERP_LINEAR ? 0 : (interp == GS360_INTERP_CUBIC ? 64 : 192));
#pragma unroll 1
    for (int s = 0; s < kRowsPerWave; ++s) {
        const int y = tile_y * kTileH + wave * kRowsPerWave + s;
        if (y >= T.h) break;
        const size_t p = (size_t)y * T.w + x;
        uint16_t* out = reinterpret_cast<uint16_t*>(T.dst + (int64_t)y * T.dst_stride) + (size_t)x * C;
        if (T.valid && !T.valid[p]) {
#pragma unroll
            for (int c = 0; c < C; ++c) out[c] = (uint16_t)T.fill;
            continue;
        }
        const float mx = T.map_x[p], my = T.map_y[p];
        if (interp == GS360_INTERP_NEAREST) {
            const int ix = sat_s16_u16(cv_round_u16(mx)), iy = sat_s16_u16(cv_round_u16(my));
            const bool in = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
#pragma unroll
            for (int c = 0; c < C; ++c) out[c] = in ? src[(size_t)iy * ss + (size_t)ix * C + c] : cval[c];
            continue;
        }
        const int sx = cv_round_u16(mx * 32.0f), sy = cv_round_u16(my * 32.0f);
        const int fx = sx & 31, fy = sy & 31;
        const int ix = sat_s16_u16(sx >> 5), iy = sat_s16_u16(sy >> 5);
        const int x0 = ix - (ks / 2 - 1), y0 = iy - (ks / 2 - 1);
        if (x0 >= W || x0 + ks <= 0 || y0 >= H || y0 + ks <= 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) out[c] = cval[c];
            continue;
        }
        const float* cy = tab + fy * ks;
        const float* cx = tab + fx * ks;
        const bool inside = x0 >= 0 && x0 + ks <= W && y0 >= 0 && y0 + ks <= H;
        if constexpr (C == 3) {
            // windows inside the image, RGB: the ks taps of a window row are 6 ks contiguous bytes -> dword-aligned wide reads
            // shifted into place (2-byte loads otherwise); the float32 accumulation order is the one spelled out in the header
            if (inside && interp != GS360_INTERP_LANCZOS4 && x0 + ks + 2 <= W &&
                ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)T.src_stride) & 3) == 0) {
                float sum[3] = {0.f, 0.f, 0.f};
                if (interp == GS360_INTERP_LINEAR) {
                    float v[2][6];
#pragma unroll
                    for (int ky = 0; ky < 2; ++ky) {
                        const uint16_t* pp = src + (size_t)(y0 + ky) * ss + (size_t)x0 * 3;
                        const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(pp) & 3u;
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(reinterpret_cast<const uint8_t*>(pp) - o, 4));
                        const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                        const uint32_t d0 = __builtin_amdgcn_alignbyte(q1, q0, o), d1 = __builtin_amdgcn_alignbyte(q2, q1, o),
                                       d2 = __builtin_amdgcn_alignbyte(q3, q2, o);
                        v[ky][0] = (float)(d0 & 0xffffu); v[ky][1] = (float)(d0 >> 16); v[ky][2] = (float)(d1 & 0xffffu);
                        v[ky][3] = (float)(d1 >> 16); v[ky][4] = (float)(d2 & 0xffffu); v[ky][5] = (float)(d2 >> 16);
                    }
                    const float w00 = cy[0] * cx[0], w01 = cy[0] * cx[1], w10 = cy[1] * cx[0], w11 = cy[1] * cx[1];
#pragma unroll
                    for (int c = 0; c < 3; ++c) sum[c] = v[0][c] * w00 + v[0][3 + c] * w01 + v[1][c] * w10 + v[1][3 + c] * w11;
                } else {
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky) {
                        const uint16_t* pp = src + (size_t)(y0 + ky) * ss + (size_t)x0 * 3;
                        const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(pp) & 3u;
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(reinterpret_cast<const uint8_t*>(pp) - o, 4));
                        uint32_t r[7], d[6];
#pragma unroll
                        for (int t = 0; t < 7; ++t) r[t] = q[t];
#pragma unroll
                        for (int t = 0; t < 6; ++t) d[t] = __builtin_amdgcn_alignbyte(r[t + 1], r[t], o);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float rs = 0.f;
#pragma unroll
                            for (int kx = 0; kx < 4; ++kx) {
                                const int e = kx * 3 + c;
                                const float v = (float)((e & 1) ? (d[e >> 1] >> 16) : (d[e >> 1] & 0xffffu));
                                const float term = v * (cy[ky] * cx[kx]);
                                rs = kx == 0 ? term : rs + term;
                            }
                            sum[c] = ky == 0 ? rs : sum[c] + rs;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) out[c] = sat_u16(sum[c]);
                continue;
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float sum;
            if (interp == GS360_INTERP_LINEAR) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
                    v[q] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? (float)src[(size_t)yy * ss + (size_t)xx * C + c] : (float)cval[c];
                }
                sum = v[0] * (cy[0] * cx[0]) + v[1] * (cy[0] * cx[1]) + v[2] * (cy[1] * cx[0]) + v[3] * (cy[1] * cx[1]);
            } else if (inside) {
                sum = 0.f;
                for (int ky = 0; ky < ks; ++ky) {
                    const uint16_t* row = src + (size_t)(y0 + ky) * ss + (size_t)x0 * C + c;
                    float rs = (float)row[0] * (cy[ky] * cx[0]);
                    for (int kx = 1; kx < ks; ++kx) rs += (float)row[(size_t)kx * C] * (cy[ky] * cx[kx]);
                    sum = (ky == 0 && interp == GS360_INTERP_CUBIC) ? rs : sum + rs;
                }
            } else {
                const float cv = (float)cval[c];
                sum = cv;
                for (int ky = 0; ky < ks; ++ky) {
                    const int yy = y0 + ky;
                    if (yy < 0 || yy >= H) continue;
                    for (int kx = 0; kx < ks; ++kx) {
                        const int xx = x0 + kx;
                        if (xx < 0 || xx >= W) continue;
                        sum += ((float)src[(size_t)yy * ss + (size_t)xx * C + c] - cv) * (cy[ky] * cx[kx]);
                    }
                }
            }
            out[c] = sat_u16(sum);
        }
    }
}

}  // namespace

namespace {
// Self-test of the reduced IEEE sequences of gs360_eqspec.h against the generic ones (`/`, sqrtf: correctly rounded with this
// build's flags) on pseudo-random operands drawn from -- and well beyond -- the operand domains EQ-SPEC / FE-SPEC produce.
__global__ __launch_bounds__(256) void arith_selftest_kernel(uint32_t seed, int iters, unsigned long long* bad) {
    uint32_t s = seed ^ (blockIdx.x * 2654435761u) ^ (threadIdx.x * 40503u + 1u);
    auto next = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        // divisor: any mantissa, exponent in [2^-100, 2^100]; numerator: 0, a fraction of the divisor that stays >= 2^-103
        // (|n| <= d), or 2 (FE-SPEC's 2 / d)
        const uint32_t r0 = next(), r1 = next(), r2 = next();
        const float d = __builtin_bit_cast(float, ((27u + r0 % 201u) << 23) | (r1 & 0x7fffffu));
        const float u = __builtin_bit_cast(float, ((100u + r2 % 28u) << 23) | (next() & 0x7fffffu));     // (2^-27, 2)
        float n = (r2 & 0x80000000u) ? d * fminf(u, 1.0f) : 0.0f;
        if (n != 0.0f && n < 0x1p-103f) n = 0.0f;          // below that v_div_scale rescales the numerator
        if ((r0 >> 28) == 0) n = -n;                                      // EQ-SPEC's (mn - mx) numerators are <= 0
        if ((r0 >> 24) == 0x55) n = d;                                    // quotient exactly 1
        if (eq_div(n, d) != n / d) ++wrong;
        if (d > 0x1p-24f && d < 0x1p24f && eq_div(2.0f, d) != 2.0f / d) ++wrong;
        // square roots: operands in [2^-96, 2^96]
        const float x = __builtin_bit_cast(float, ((31u + r1 % 193u) << 23) | (r0 & 0x7fffffu));
        if (eq_sqrt_normal(x) != __builtin_sqrtf(x)) ++wrong;
        if (eq_sqrt(x) != __builtin_sqrtf(x)) ++wrong;
    }
    // tiny / zero operands take eq_sqrt's wave-uniform fallback
    const float tiny = __builtin_bit_cast(float, (next() % 31u) << 23 | (next() & 0x7fffffu));
    if (eq_sqrt(tiny) != __builtin_sqrtf(tiny)) ++wrong;
    if (eq_sqrt(0.0f) != 0.0f) ++wrong;
    if (wrong) atomicAdd(bad, wrong);
}
}  // namespace

hipError_t launch_arith_selftest(uint32_t seed, int blocks, int iters, unsigned long long* d_bad, hipStream_t s) {
    hipLaunchKernelGGL(arith_selftest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, seed, iters, d_bad);
    return hipGetLastError();
}

hipError_t launch_equirect_u16(const EqLaunch& L, int C, bool cubic, hipStream_t s) {
    dim3 grid((unsigned)(L.chunk * 8)), block(64 * kWaves);
    if (cubic) {
        switch (C) {
            case 1: hipLaunchKernelGGL((eq_views_u16_kernel<1, true>), grid, block, 0, s, L); break;
            case 3: hipLaunchKernelGGL((eq_views_u16_kernel<3, true>), grid, block, 0, s, L); break;
            case 4: hipLaunchKernelGGL((eq_views_u16_kernel<4, true>), grid, block, 0, s, L); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (C) {
            case 1: hipLaunchKernelGGL((eq_views_u16_kernel<1, false>), grid, block, 0, s, L); break;
            case 3: hipLaunchKernelGGL((eq_views_u16_kernel<3, false>), grid, block, 0, s, L); break;
            case 4: hipLaunchKernelGGL((eq_views_u16_kernel<4, false>), grid, block, 0, s, L); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

hipError_t launch_table_u16(const TableLaunch& T, int C, const float* coef, const uint16_t cval[4], hipStream_t s) {
    const int tiles = ((T.w + kTileW - 1) / kTileW) * ((T.h + kTileH - 1) / kTileH);
    if (tiles == 0) return hipSuccess;
    dim3 grid((unsigned)tiles), block(64 * kWaves);
    switch (C) {
        case 1: hipLaunchKernelGGL((table_remap_u16_kernel<1>), grid, block, 0, s, T, coef, cval[0], cval[1], cval[2], cval[3]); break;
        case 3: hipLaunchKernelGGL((table_remap_u16_kernel<3>), grid, block, 0, s, T, coef, cval[0], cval[1], cval[2], cval[3]); break;
        case 4: hipLaunchKernelGGL((table_remap_u16_kernel<4>), grid, block, 0, s, T, coef, cval[0], cval[1], cval[2], cval[3]); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gs360
