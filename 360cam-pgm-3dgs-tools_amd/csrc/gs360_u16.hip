// gs360_u16.hip -- cv2.remap on CV_16U sources and the arithmetic self-test.
// (The 16-bit EQUIRECT sampler lives in gs360_kernels.hip: eq_views_kernel<..., ES = 2>, the 8-bit kernel's skeleton on 2-byte samples.)
//
//   table_remap_u16_kernel   cv2.remap on CV_16U sources (DF:735 keeps 16-bit inputs at native depth; DF:2001-2014):
//                            OpenCV routes ushort to its FLOAT-weight samplers -- 2-D weight = cy[k1] * cx[k2] in float32
//                            from the 1-D phase tables, float32 accumulation in OpenCV's expression order, cvRound +
//                            saturate to [0, 65535].  Order (compiled with -ffp-contract=off): window inside the image --
//                            bilinear ((S00 w0 + S01 w1) + S10 w2) + S11 w3; bicubic / lanczos4 sum each window row left to
//                            right and add the row sums row by row; window on the border (BORDER_CONSTANT) -- bilinear replaces
//                            outside taps by the border value, bicubic / lanczos4 start from the border value cv and add
//                            (S - cv) w for every in-image tap in row-major order.
//
// All jobs of a call (the views of a pair) share one launch; the maps / valid flags of a wavefront's four rows go out together;
// bilinear RGB windows inside the image take a pipelined path (all eight row reads in flight, then the float32 blend); everything
// else -- bicubic, Lanczos, borders, other channel counts -- runs the straight-line sampler one row slot at a time; rows leave as
// whole dwords (gs360_rowstore.h).
#include "gs360_eqspec.h"
#include "gs360_rowstore.h"

namespace gs360 {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int cv_round_u16(float v) {
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return (int)0x80000000;
    return (int)__builtin_rintf(v);
}
__device__ __forceinline__ int sat_s16_u16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ uint16_t sat_u16(float v) { return (uint16_t)min(max(cv_round_u16(v), 0), 65535); }

// cv2.remap(CV_16U) -- see the file header.  coef: 32 phases x (2 + 4 + 8) float32 1-D coefficients (linear, cubic, lanczos4).
// All jobs of a call (the views of a dual-fisheye pair) in ONE launch, tiles dealt to the XCDs in contiguous chunks like the
// 8-bit kernel: six 1750^2 launches left a fifth of the machine idle in their tails.
template <int C, int INTERP>
__global__ __launch_bounds__(64 * kWaves) void table_remap_u16_kernel(const TableBatch B, const float* __restrict__ coef, const uint16_t c0,
                                                                      const uint16_t c1, const uint16_t c2, const uint16_t c3) {
    int t = (blockIdx.x & 7) * B.chunk + (blockIdx.x >> 3);
    if (t >= B.total_tiles) return;
    int j = 0;
    while (j + 1 < B.n_jobs && t >= B.job[j + 1].tile_base) ++j;
    const TableLaunch& T = B.job[j];          // wave-uniform: fields are read from the kernel argument on demand
    t -= T.tile_base;
    const int tiles_x = T.tiles_x;
    const int tile_y = t / tiles_x, tile_x = t - tile_y * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0t = tile_x * kTileW;
    const int n_px = min(kTileW, T.w - x0t);
    const int x = min(x0t + lane, T.w - 1);   // lanes past the edge redo the last column: every lane stays active for the packed stores
    const uint32_t cval[4] = {c0, c1, c2, c3};   // (as dwords: a uint16_t[4] local ended up in scratch memory in the nearest instantiation)
    const uint16_t* __restrict__ src = reinterpret_cast<const uint16_t*>(T.src);
    const size_t ss = (size_t)T.src_stride >> 1;
    const int W = T.W, H = T.H;
    constexpr int interp = INTERP;            // one interpolation per call (gs360_remap_tables_u16): a compile-time constant
    const int ks = interp == GS360_INTERP_LINEAR ? 2 : (interp == GS360_INTERP_CUBIC ? 4 : 8);
    const float* tab = coef + (interp == GS360_INTERP_LINEAR ? 0 : (interp == GS360_INTERP_CUBIC ? 64 : 192));
    // the job's fields as values (read once), the maps and valid flags of the wavefront's four rows in flight together
    const float* __restrict__ map_x = T.map_x;
    const float* __restrict__ map_y = T.map_y;
    const uint8_t* __restrict__ vmask = T.valid;
    uint8_t* const dst = T.dst;
    const int64_t dst_stride = T.dst_stride;
    const int tw = T.w, th = T.h, fill = T.fill;
    const int ybase = tile_y * kTileH + wave * kRowsPerWave;
    float mxs[kRowsPerWave], mys[kRowsPerWave];
    bool inval[kRowsPerWave], done[kRowsPerWave];
    uint32_t px[kRowsPerWave][4];
    // (branch-free: without a valid map the byte is read from the map itself and ignored -- see table_remap_kernel)
    const uint8_t* __restrict__ vptr = vmask ? vmask : reinterpret_cast<const uint8_t*>(map_x);
    const bool has_valid = vmask != nullptr;
    uint8_t vbyte[kRowsPerWave];
    if (T.packed) {                           // a map plan (wave-uniform): one dword and one byte per pixel, see gs360_kernels.hip
        uint32_t pw[kRowsPerWave];
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const size_t p = (size_t)min(ybase + s, th - 1) * tw + x;
            pw[s] = T.packed[p];
            vbyte[s] = T.packed_hi[p];
            done[s] = false;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            planned_coords(pw[s], vbyte[s], INTERP == GS360_INTERP_NEAREST, mxs[s], mys[s]);
            inval[s] = (T.use_valid != 0) & ((vbyte[s] & 4) == 0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const size_t p = (size_t)min(ybase + s, th - 1) * tw + x;
            mxs[s] = map_x[p];
            mys[s] = map_y[p];
            vbyte[s] = vptr[p];
            done[s] = false;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) inval[s] = has_valid & (vbyte[s] == 0);
    }
    if constexpr (C == 3) {
        // bilinear, RGB, windows inside the image: the two row reads of ALL four slots (dword-aligned 16-byte reads of the 12 tap
        // bytes) and their weight reads are issued before any is consumed; float32 blend in OpenCV's expression order
        if (interp == GS360_INTERP_LINEAR && W >= 8 && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)T.src_stride) & 1) == 0) {
            uint32_t ra[kRowsPerWave][4], rb[kRowsPerWave][4], sh[kRowsPerWave];
            float2 cys[kRowsPerWave], cxs[kRowsPerWave];
            bool fast[kRowsPerWave];
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                const int sx = cv_round_u16(mxs[s] * 32.0f), sy = cv_round_u16(mys[s] * 32.0f);
                const int ix = sat_s16_u16(sx >> 5), iy = sat_s16_u16(sy >> 5);
                fast[s] = !inval[s] && ix >= 0 && iy >= 0 && ix + 3 <= W && iy + 2 <= H && ybase + s < th;
                const int xa = min(max(ix, 0), W - 3), ya = min(max(iy, 0), H - 2);
                const uint8_t* p0 = reinterpret_cast<const uint8_t*>(src) + (size_t)ya * (size_t)T.src_stride + (size_t)xa * 6;
                const uint8_t* p1 = p0 + (size_t)T.src_stride;
                const uint32_t o0 = (uint32_t)reinterpret_cast<uintptr_t>(p0) & 3u, o1 = (uint32_t)reinterpret_cast<uintptr_t>(p1) & 3u;
                sh[s] = o0 | (o1 << 2);
                const uint32_t* q0 = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(p0 - o0, 4));
                const uint32_t* q1 = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(p1 - o1, 4));
#pragma unroll
                for (int k = 0; k < 4; ++k) { ra[s][k] = q0[k]; rb[s][k] = q1[k]; }
                cys[s] = *reinterpret_cast<const float2*>(tab + (sy & 31) * 2);
                cxs[s] = *reinterpret_cast<const float2*>(tab + (sx & 31) * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                if (!fast[s]) continue;
                const uint32_t o0 = sh[s] & 3u, o1 = sh[s] >> 2;
                const uint32_t d[2][3] = {{__builtin_amdgcn_alignbyte(ra[s][1], ra[s][0], o0), __builtin_amdgcn_alignbyte(ra[s][2], ra[s][1], o0),
                                           __builtin_amdgcn_alignbyte(ra[s][3], ra[s][2], o0)},
                                          {__builtin_amdgcn_alignbyte(rb[s][1], rb[s][0], o1), __builtin_amdgcn_alignbyte(rb[s][2], rb[s][1], o1),
                                           __builtin_amdgcn_alignbyte(rb[s][3], rb[s][2], o1)}};
                float v[2][6];
#pragma unroll
                for (int ky = 0; ky < 2; ++ky) {
                    v[ky][0] = (float)(d[ky][0] & 0xffffu); v[ky][1] = (float)(d[ky][0] >> 16); v[ky][2] = (float)(d[ky][1] & 0xffffu);
                    v[ky][3] = (float)(d[ky][1] >> 16); v[ky][4] = (float)(d[ky][2] & 0xffffu); v[ky][5] = (float)(d[ky][2] >> 16);
                }
                const float w00 = cys[s].x * cxs[s].x, w01 = cys[s].x * cxs[s].y, w10 = cys[s].y * cxs[s].x, w11 = cys[s].y * cxs[s].y;
#pragma unroll
                for (int c = 0; c < 3; ++c) px[s][c] = sat_u16(v[0][c] * w00 + v[0][3 + c] * w01 + v[1][c] * w10 + v[1][3 + c] * w11);
                done[s] = true;
            }
        }
    }
    if constexpr (C == 3 && interp == GS360_INTERP_CUBIC) {
        // bicubic, RGB, windows inside the image: software pipeline over the four row slots -- the four row reads (7 aligned dwords
        // each) of slot s + 1 are issued before slot s is blended, so a wavefront always has a slot's reads in flight while it
        // computes (one slot at a time the vector ALU was 53 % busy).  Reads go to a clamped, always-valid window; `fast` says
        // whether the result is the pixel's (otherwise the straight-line sampler below redoes it).  Float32 sums in OpenCV's order.
        if (W >= 8 && H >= 4 && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)T.src_stride) & 3) == 0) {
            uint32_t raw[2][4][7];
            uint32_t sh[kRowsPerWave];
            int fxs[kRowsPerWave], fys[kRowsPerWave];
            bool fast[kRowsPerWave];
            auto issue = [&](int sl, uint32_t (&buf)[4][7]) {
                const int sx = cv_round_u16(mxs[sl] * 32.0f), sy = cv_round_u16(mys[sl] * 32.0f);
                const int x0 = sat_s16_u16(sx >> 5) - 1, y0 = sat_s16_u16(sy >> 5) - 1;
                fxs[sl] = sx & 31;
                fys[sl] = sy & 31;
                fast[sl] = !inval[sl] && x0 >= 0 && y0 >= 0 && x0 + 6 <= W && y0 + 4 <= H && ybase + sl < th;
                const int xa = min(max(x0, 0), W - 6), ya = min(max(y0, 0), H - 4);
                const uint8_t* p0 = reinterpret_cast<const uint8_t*>(src) + (size_t)ya * (size_t)T.src_stride + (size_t)xa * 6;
                const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(p0) & 3u;      // the stride is a multiple of 4: one shift
                sh[sl] = o;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(p0 - o + (size_t)ky * (size_t)T.src_stride, 4));
#pragma unroll
                    for (int t = 0; t < 7; ++t) buf[ky][t] = q[t];
                }
            };
            // (channels 0 and 1 ride in one packed-float32 register pair: v_pk_mul_f32 / v_pk_add_f32 are two IEEE operations per
            // instruction -- the same roundings, in the same order, as the scalar form)
            auto finish = [&](int sl, const uint32_t (&buf)[4][7]) {
                const float* cy = tab + fys[sl] * 4;
                const float* cx = tab + fxs[sl] * 4;
                f32x2 sum01 = {0.f, 0.f};
                float sum2 = 0.f;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    uint32_t d[6];
#pragma unroll
                    for (int t = 0; t < 6; ++t) d[t] = __builtin_amdgcn_alignbyte(buf[ky][t + 1], buf[ky][t], sh[sl]);
                    const float cyk = cy[ky];
                    f32x2 rs01 = {0.f, 0.f};
                    float rs2 = 0.f;
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const int e = kx * 3;                               // halfwords e, e + 1, e + 2 = the tap's three channels
                        auto hw = [&](int h) { return (float)((h & 1) ? (d[h >> 1] >> 16) : (d[h >> 1] & 0xffffu)); };
                        const float w = cyk * cx[kx];
                        const f32x2 v01 = {hw(e), hw(e + 1)};
                        const f32x2 w01 = {w, w};
                        const f32x2 t01 = v01 * w01;
                        const float t2 = hw(e + 2) * w;
                        rs01 = kx == 0 ? t01 : rs01 + t01;
                        rs2 = kx == 0 ? t2 : rs2 + t2;
                    }
                    sum01 = ky == 0 ? rs01 : sum01 + rs01;
                    sum2 = ky == 0 ? rs2 : sum2 + rs2;
                }
                if (fast[sl]) {
                    px[sl][0] = sat_u16(sum01.x);
                    px[sl][1] = sat_u16(sum01.y);
                    px[sl][2] = sat_u16(sum2);
                    done[sl] = true;
                }
            };
            issue(0, raw[0]);
#pragma unroll
            for (int sl = 0; sl < kRowsPerWave; ++sl) {
                if (sl + 1 < kRowsPerWave) issue(sl + 1, raw[(sl + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                finish(sl, raw[sl & 1]);
            }
        }
    }
    // one pixel, straight-line (every interpolation, borders, any channel count): the rolled loop below runs it for the slots the
    // batched path above did not finish
    auto sample_one = [&](float mx, float my, bool iv, uint32_t (&out)[4]) {
        if (iv) {
#pragma unroll
            for (int c = 0; c < C; ++c) out[c] = (uint32_t)fill;
            return;
        }
        if (interp == GS360_INTERP_NEAREST) {
            const int ix = sat_s16_u16(cv_round_u16(mx)), iy = sat_s16_u16(cv_round_u16(my));
            const bool in = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
#pragma unroll
            for (int c = 0; c < C; ++c) out[c] = in ? src[(size_t)iy * ss + (size_t)ix * C + c] : cval[c];
            return;
        }
        const int sx = cv_round_u16(mx * 32.0f), sy = cv_round_u16(my * 32.0f);
        const int fx = sx & 31, fy = sy & 31;
        const int ix = sat_s16_u16(sx >> 5), iy = sat_s16_u16(sy >> 5);
        const int x0 = ix - (ks / 2 - 1), y0 = iy - (ks / 2 - 1);
        if (x0 >= W || x0 + ks <= 0 || y0 >= H || y0 + ks <= 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) out[c] = cval[c];
            return;
        }
        const float* cy = tab + fy * ks;
        const float* cx = tab + fx * ks;
        const bool inside = x0 >= 0 && x0 + ks <= W && y0 >= 0 && y0 + ks <= H;
        if constexpr (C == 3) {
            // windows inside the image, RGB: the ks taps of a window row are 6 ks contiguous bytes -> dword-aligned wide reads
            // shifted into place (2-byte loads otherwise); the float32 accumulation order is the one spelled out in the header
            if (inside && x0 + ks + 2 <= W && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)T.src_stride) & 3) == 0) {
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
                    // bicubic (4 x 4) / Lanczos-4 (8 x 8): a window row is 6 ks contiguous bytes = 3 ks / 2 dwords + one for the
                    // misalignment; row sums left to right, rows added top to bottom (Lanczos starts from 0.f, as OpenCV does)
                    constexpr int KS = interp == GS360_INTERP_LANCZOS4 ? 8 : 4;
                    constexpr int ND = 3 * KS / 2;
                    float cxs[KS];
#pragma unroll
                    for (int kx = 0; kx < KS; ++kx) cxs[kx] = cx[kx];
                    // (bicubic: all four rows' reads in flight together -- unrolled by two it runs 17 % slower, at 6 instead of 4
                    // wavefronts per SIMD; Lanczos: two rows at a time keep the instantiation at 126 registers)
                    constexpr int kRowUnroll = KS == 4 ? 4 : 2;
#pragma unroll kRowUnroll
                    for (int ky = 0; ky < KS; ++ky) {
                        const uint16_t* pp = src + (size_t)(y0 + ky) * ss + (size_t)x0 * 3;
                        const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(pp) & 3u;
                        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(reinterpret_cast<const uint8_t*>(pp) - o, 4));
                        uint32_t r[ND + 1], d[ND];
#pragma unroll
                        for (int t = 0; t < ND + 1; ++t) r[t] = q[t];
#pragma unroll
                        for (int t = 0; t < ND; ++t) d[t] = __builtin_amdgcn_alignbyte(r[t + 1], r[t], o);
                        const float cyk = cy[ky];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float rs = 0.f;
#pragma unroll
                            for (int kx = 0; kx < KS; ++kx) {
                                const int e = kx * 3 + c;
                                const float v = (float)((e & 1) ? (d[e >> 1] >> 16) : (d[e >> 1] & 0xffffu));
                                const float term = v * (cyk * cxs[kx]);
                                rs = kx == 0 ? term : rs + term;
                            }
                            sum[c] = (ky == 0 && interp == GS360_INTERP_CUBIC) ? rs : sum[c] + rs;
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) out[c] = sat_u16(sum[c]);
                return;
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
    };
    // The rolled loop always works on slot 0 and rotates the four slots through the registers (back in place after four turns):
    // picking the slot with `s == k ? a[k] : ...` made the compiler keep the arrays in scratch.
    static_assert(kRowsPerWave == 4, "four row slots");
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        if (!done[0] && ybase + s < th) sample_one(mxs[0], mys[0], inval[0], px[0]);
        const float tx = mxs[0], ty = mys[0];
        const bool ti = inval[0], td = done[0];
        uint32_t tp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) tp[c] = px[0][c];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mxs[k] = mxs[k + 1]; mys[k] = mys[k + 1]; inval[k] = inval[k + 1]; done[k] = done[k + 1];
#pragma unroll
            for (int c = 0; c < 4; ++c) px[k][c] = px[k + 1][c];
        }
        mxs[3] = tx; mys[3] = ty; inval[3] = ti; done[3] = td;
#pragma unroll
        for (int c = 0; c < 4; ++c) px[3][c] = tp[c];
    }
    // rows leave as whole dwords (96 per 64 RGB pixels) instead of three 2-byte stores per lane
    const RowPack rp = make_row_pack();
    const bool aligned4 = ((dst_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (ybase + s < th)
            store_row16<C>(dst + (int64_t)(ybase + s) * dst_stride + (int64_t)x0t * (2 * C), px[s], n_px, aligned4, rp, false, false);
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

namespace {
// 16-bit PPM frames arrive big-endian (video decode pipe, gs360/video.py): swapped in place on the device after the upload, on the
// upload stream -- a 177 MB rgb48 8K frame is 60 us here against a 35 ms pass over pinned memory on the reader thread.
// Grid-stride over dwords (two samples each: v_perm), the odd head / tail sample by one lane each.
__global__ __launch_bounds__(256) void bswap16_kernel(uint16_t* buf, size_t n) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(buf);
    const size_t head = (a & 2) ? 1 : 0;                         // samples before the first dword boundary
    const size_t nd = (n - (n < head ? n : head)) / 2;             // whole dwords after the head
    uint32_t* d = reinterpret_cast<uint32_t*>(buf + head);
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < nd; i += step) d[i] = __builtin_amdgcn_perm(0u, d[i], 0x02030001u);
    if (tid == 0) {
        if (head && n) buf[0] = (uint16_t)((buf[0] >> 8) | (buf[0] << 8));
        const size_t tail = head + 2 * nd;
        if (tail < n) buf[tail] = (uint16_t)((buf[tail] >> 8) | (buf[tail] << 8));
    }
}
}  // namespace

hipError_t launch_bswap16(uint16_t* buf, size_t n, hipStream_t s) {
    const size_t nd = n / 2 + 1;
    unsigned blocks = (unsigned)((nd + 255) / 256 < 256 * 16 ? (nd + 255) / 256 : 256 * 16);
    hipLaunchKernelGGL(bswap16_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, buf, n);
    return hipGetLastError();
}

hipError_t launch_arith_selftest(uint32_t seed, int blocks, int iters, unsigned long long* d_bad, hipStream_t s) {
    hipLaunchKernelGGL(arith_selftest_kernel, dim3((unsigned)blocks), dim3(256), 0, s, seed, iters, d_bad);
    return hipGetLastError();
}

hipError_t launch_table_u16_batch(TableBatch& B, int C, const float* coef, const uint16_t cval[4], hipStream_t s) {
    int base = 0;
    for (int j = 0; j < B.n_jobs; ++j) {
        TableLaunch& L = B.job[j];
        L.tiles_x = (L.w + kTileW - 1) / kTileW;
        L.tile_base = base;
        base += L.tiles_x * ((L.h + kTileH - 1) / kTileH);
    }
    B.total_tiles = base;
    B.chunk = (base + 7) / 8;
    if (base == 0) return hipSuccess;
    dim3 grid((unsigned)(B.chunk * 8)), block(64 * kWaves);
    const int interp = B.job[0].interp;       // the same for every job of a call
#define GS360_T16(CC, II) hipLaunchKernelGGL((table_remap_u16_kernel<CC, II>), grid, block, 0, s, B, coef, cval[0], cval[1], cval[2], cval[3])
#define GS360_T16_C(CC)                                                             \
    switch (interp) {                                                               \
        case GS360_INTERP_NEAREST: GS360_T16(CC, GS360_INTERP_NEAREST); break;      \
        case GS360_INTERP_LINEAR: GS360_T16(CC, GS360_INTERP_LINEAR); break;        \
        case GS360_INTERP_CUBIC: GS360_T16(CC, GS360_INTERP_CUBIC); break;          \
        case GS360_INTERP_LANCZOS4: GS360_T16(CC, GS360_INTERP_LANCZOS4); break;    \
        default: return hipErrorInvalidValue;                                       \
    }
    switch (C) {
        case 1: GS360_T16_C(1) break;
        case 3: GS360_T16_C(3) break;
        case 4: GS360_T16_C(4) break;
        default: return hipErrorInvalidValue;
    }
#undef GS360_T16_C
#undef GS360_T16
    return hipGetLastError();
}

}  // namespace gs360
