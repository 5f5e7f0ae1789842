// gs360_table.hip -- cv2.remap semantics on gfx950: table_remap_kernel (cv2.remap(INTER_NEAREST|LINEAR|CUBIC|LANCZOS4, BORDER_CONSTANT) +
// valid fill, cli_tools/gs360_DualFisheyeDistortionCalibration.py:2001-2014 / :2031-2043 / :1198-1212), map plans (map_pack_kernel) and the
// fused dual-fisheye kernel fe_views_kernel (FE-SPEC v1, DF:1759-1823 in-kernel).  Split out of gs360_kernels.hip in round 5; the tiling,
// lane maps and store paths are described there and in DESIGN.md section 5.
//
// Compile with -ffp-contract=off (see gs360_kernels.hip).
#include "gs360_sampler.h"
#include "gs360_cvremap.h"

namespace gs360 {

// ------------------------------------------------------------------------------------------------
// cv2.remap semantics (shared by the table kernel and the fused fisheye kernel)
// ------------------------------------------------------------------------------------------------
// ---- map plans -------------------------------------------------------------------------------------------------------------------
// cv2.remap turns its float maps into 1/32-pixel fixed point on every call (cvRound(map * 32), integer part saturated to int16) before
// any sampling; a plan does that once and keeps the result in 5 bytes per pixel instead of the 9 of two floats and a valid byte.  The
// integer part is clamped to [-8, 4087]: every position whose widest window (Lanczos-4: x - 3 .. x + 4) still touches a source of up
// to 4079 x 4079 pixels is kept as it is, and one that is moved had no tap inside the image before and has none after -- the border
// constant either way.  The samplers take the position back as floats k / 32, exact, whose cvRound(. * 32) is k again.
__global__ __launch_bounds__(256) void map_pack_kernel(const float* __restrict__ map_x, const float* __restrict__ map_y,
                                                       const uint8_t* __restrict__ valid, int64_t n, int nearest,
                                                       uint32_t* __restrict__ packed, uint8_t* __restrict__ packed_hi) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float mx = map_x[i], my = map_y[i];
    int ix, iy, fx = 0, fy = 0;
    if (nearest) {
        ix = sat_s16(cv_round(mx));
        iy = sat_s16(cv_round(my));
    } else {
        const int sx = cv_round(mx * 32.0f), sy = cv_round(my * 32.0f);
        fx = sx & 31; fy = sy & 31;
        ix = sat_s16(sx >> 5);
        iy = sat_s16(sy >> 5);
    }
    ix = min(max(ix, -8), kMapPlanMaxDim + 8) + 8;
    iy = min(max(iy, -8), kMapPlanMaxDim + 8) + 8;
    packed[i] = (uint32_t)ix | ((uint32_t)iy << 12) | ((uint32_t)fx << 24) | ((uint32_t)(fy & 7) << 29);
    packed_hi[i] = (uint8_t)((fy >> 3) | ((!valid || valid[i]) ? 4 : 0));
}

template <int C>
__device__ __forceinline__ void cv_sample_nearest(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                  float mx, float my, const uint8_t (&cval)[4], uint32_t (&out)[4]) {
    int ix = sat_s16(cv_round(mx)), iy = sat_s16(cv_round(my));
    bool inside = (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
    int xa = min(max(ix, 0), W - 1), ya = min(max(iy, 0), H - 1);
    const uint8_t* s = src + (int64_t)ya * stride + (int64_t)xa * C;
    if constexpr (C == 3) {
        // one gather instead of three: the pixel's 3 bytes lie in the 8 bytes that start at its dword -- inside the row for every
        // column but the last two, which keep the byte reads.  (Single exit: an early return here put the callers' pixel arrays
        // into scratch memory, 81 us -> 2 ms per cfg4 pair; tests/test_capi_load.py now watches the compiler's report.)
        uint32_t v;
        if (xa <= W - 3) {
            const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(s) & 3u;
            const uint2 q = *reinterpret_cast<const uint2*>(__builtin_assume_aligned(s - o, 4));
            v = __builtin_amdgcn_alignbyte(q.y, q.x, o);
        } else {
            v = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        }
        out[0] = inside ? (v & 0xffu) : (uint32_t)cval[0];
        out[1] = inside ? ((v >> 8) & 0xffu) : (uint32_t)cval[1];
        out[2] = inside ? ((v >> 16) & 0xffu) : (uint32_t)cval[2];
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            uint32_t v = s[c];
            out[c] = inside ? v : (uint32_t)cval[c];
        }
    }
}

// remapBicubic, BORDER_CONSTANT: 4x4 window at (ix-1, iy-1); int16 weights (sum 32768) from the 32x32-phase table;
// taps outside the image are the border constant; (sum + 2^14) >> 15 saturated to u8.
template <int C>
__device__ __forceinline__ void cv_sample_cubic(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                float mx, float my, const uint8_t (&cval)[4],
                                                const int16_t* __restrict__ tab, uint32_t (&out)[4]) {
    int sx = cv_round(mx * 32.0f), sy = cv_round(my * 32.0f);
    int fx = sx & 31, fy = sy & 31;
    int x0 = sat_s16(sx >> 5) - 1, y0 = sat_s16(sy >> 5) - 1;
    bool outside = x0 >= W || x0 + 4 <= 0 || y0 >= H || y0 + 4 <= 0;
    const uint4* wq = reinterpret_cast<const uint4*>(tab + (fy * 32 + fx) * 16);   // 32 B = two 16-B reads
    uint4 wa = wq[0], wb = wq[1];
    const uint32_t wpk[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        int yy = y0 + ky;
        bool yin = (unsigned)yy < (unsigned)H;
        const uint8_t* row = src + (int64_t)min(max(yy, 0), H - 1) * stride;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            int xx = x0 + kx;
            bool in = yin && ((unsigned)xx < (unsigned)W);
            const uint8_t* px = row + (int64_t)min(max(xx, 0), W - 1) * C;
            uint32_t pk = wpk[(ky * 4 + kx) >> 1];
            int w = (int)(int16_t)((kx & 1) ? (pk >> 16) : (pk & 0xffffu));
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)(in ? (uint32_t)px[c] : (uint32_t)cval[c]) * w;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        int r = (acc[c] + (1 << 14)) >> 15;
        out[c] = outside ? (uint32_t)cval[c] : (uint32_t)min(max(r, 0), 255);
    }
}

// remapLanczos4, BORDER_CONSTANT: 8x8 window at (ix-3, iy-3), same fixed-point scheme as the bicubic sampler with the
// 32x32-phase x 64-entry int16 table (128 B per phase, one 16-byte read per window row).  Rarely selected
// (`--interpolation lanczos4`, DF:229-234), so it is the straight-line form only.
template <int C>
__device__ __forceinline__ void cv_sample_lanczos4(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                   float mx, float my, const uint8_t (&cval)[4],
                                                   const int16_t* __restrict__ tab, uint32_t (&out)[4]) {
    int sx = cv_round(mx * 32.0f), sy = cv_round(my * 32.0f);
    int fx = sx & 31, fy = sy & 31;
    int x0 = sat_s16(sx >> 5) - 3, y0 = sat_s16(sy >> 5) - 3;
    bool outside = x0 >= W || x0 + 8 <= 0 || y0 >= H || y0 + 8 <= 0;
    const uint4* wq = reinterpret_cast<const uint4*>(tab + (fy * 32 + fx) * 64);
    if constexpr (C == 3) {
        // window inside the image (and its aligned 28-byte row reads inside the row): the 8 RGB taps of a window row are 24
        // contiguous bytes -> seven dwords from the dword boundary below them, shifted into place; two taps x two packed
        // int16 weights per v_dot2 (12 per row) instead of 24 byte loads and 24 multiply-adds per row.  Exact integers.
        if (x0 >= 0 && y0 >= 0 && x0 + 10 <= W && y0 + 8 <= H && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 3) == 0) {
            int a3[3] = {0, 0, 0};
#pragma unroll
            for (int ky = 0; ky < 8; ++ky) {
                const uint4 wr = wq[ky];
                const uint32_t wpk[4] = {wr.x, wr.y, wr.z, wr.w};
                const uint8_t* p = src + (int64_t)(y0 + ky) * stride + (int64_t)x0 * 3;
                const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;
                const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(p - o, 4));
                uint32_t r[7], d[6];
#pragma unroll
                for (int t = 0; t < 7; ++t) r[t] = q[t];
#pragma unroll
                for (int t = 0; t < 6; ++t) d[t] = __builtin_amdgcn_alignbyte(r[t + 1], r[t], o);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int p0 = 6 * m + c, p1 = p0 + 3;            // bytes of taps 2m and 2m+1, channel c
                        a3[c] = dot2_i16(__builtin_amdgcn_perm(d[p1 >> 2], d[p0 >> 2], GS360_PAIR(p0 & 3, 4 + (p1 & 3))), wpk[m], a3[c]);
                    }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = (uint32_t)min(max((a3[c] + (1 << 14)) >> 15, 0), 255);
            return;
        }
    }
    if (outside) {                                        // the whole window outside the image: the border value, no taps
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = (uint32_t)cval[c];
        return;
    }
    int acc[4] = {0, 0, 0, 0};
#pragma unroll 2
    for (int ky = 0; ky < 8; ++ky) {
        const uint4 wr = wq[ky];
        const uint32_t wpk[4] = {wr.x, wr.y, wr.z, wr.w};
        int yy = y0 + ky;
        bool yin = (unsigned)yy < (unsigned)H;
        const uint8_t* row = src + (int64_t)min(max(yy, 0), H - 1) * stride;
#pragma unroll
        for (int kx = 0; kx < 8; ++kx) {
            int xx = x0 + kx;
            bool in = yin && ((unsigned)xx < (unsigned)W);
            const uint8_t* px = row + (int64_t)min(max(xx, 0), W - 1) * C;
            uint32_t pk = wpk[kx >> 1];
            int w = (int)(int16_t)((kx & 1) ? (pk >> 16) : (pk & 0xffffu));
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)(in ? (uint32_t)px[c] : (uint32_t)cval[c]) * w;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        int r = (acc[c] + (1 << 14)) >> 15;
        out[c] = outside ? (uint32_t)cval[c] : (uint32_t)min(max(r, 0), 255);
    }
}

// cv2 Lanczos-4 for an RGB window inside the image with the 2-D weights REBUILT per pixel (TableLaunch::lz_c1 / lz_cen; `lds` = the
// workgroup's copy: 256 floats of 1-D coefficients, then 2048 dwords of patched pairs).  OpenCV's table entry is
// saturate_cast<short>(cvRound((cy * cx) * 2^15)): the float32 product cy * (cx * 2^15) is the same float (a power of two scales
// exactly), and adding 1.5 * 2^23 rounds it to nearest-even into the low mantissa bits, whose low 16 are the int16 weight.  Only
// the block the table's sum fix-up patches (rows 4-5, taps 4-5: shipped per phase) and phase 0's one saturated entry differ.
// 128 B of a 128 KiB table per pixel through a 32 KiB L1 was what bounded this sampler, not its 64 taps.
// Returns false (nothing written) when the window is not inside: the caller falls back to cv_sample_lanczos4.
__device__ __forceinline__ bool cv_lanczos4_rgb_rebuilt(const uint8_t* __restrict__ src, int64_t stride, int W, int H, float mx, float my,
                                                        const float* lds, uint32_t (&out)[4]) {
    const int sx = cv_round(mx * 32.0f), sy = cv_round(my * 32.0f);
    const int fx = sx & 31, fy = sy & 31;
    const int x0 = sat_s16(sx >> 5) - 3, y0 = sat_s16(sy >> 5) - 3;
    if (!(x0 >= 0 && y0 >= 0 && x0 + 10 <= W && y0 + 8 <= H && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 3) == 0)) return false;
    const float4* cyq = reinterpret_cast<const float4*>(lds + fy * 8);
    const float4* cxq = reinterpret_cast<const float4*>(lds + fx * 8);
    const float4 cya = cyq[0], cyb = cyq[1], cxa = cxq[0], cxb = cxq[1];
    const float cy[8] = {cya.x, cya.y, cya.z, cya.w, cyb.x, cyb.y, cyb.z, cyb.w};
    const float cx32[8] = {cxa.x * 32768.0f, cxa.y * 32768.0f, cxa.z * 32768.0f, cxa.w * 32768.0f,
                           cxb.x * 32768.0f, cxb.y * 32768.0f, cxb.z * 32768.0f, cxb.w * 32768.0f};
    const int phase = fy * 32 + fx;
    const uint2 cen = reinterpret_cast<const uint2*>(lds + 256)[phase];
    int a3[3] = {0, 0, 0};
#pragma unroll 2
    for (int ky = 0; ky < 8; ++ky) {
        uint32_t wpk[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float t0 = cy[ky] * cx32[2 * m] + 12582912.0f;          // (contraction is off: product and sum round separately)
            const float t1 = cy[ky] * cx32[2 * m + 1] + 12582912.0f;
            wpk[m] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, t1), __builtin_bit_cast(uint32_t, t0), 0x05040100u);
        }
        if (ky == 3) wpk[1] = phase == 0 ? 0x7fff0000u : wpk[1];          // cy = cx = 1: 2^15 saturates to 32767 in the table
        if (ky == 4) wpk[2] = cen.x;
        if (ky == 5) wpk[2] = cen.y;
        const uint8_t* p = src + (int64_t)(y0 + ky) * stride + (int64_t)x0 * 3;
        const uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(p - o, 4));
        uint32_t r[7], d[6];
#pragma unroll
        for (int t = 0; t < 7; ++t) r[t] = q[t];
#pragma unroll
        for (int t = 0; t < 6; ++t) d[t] = __builtin_amdgcn_alignbyte(r[t + 1], r[t], o);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int p0 = 6 * m + c, p1 = p0 + 3;                    // bytes of taps 2m and 2m+1, channel c
                a3[c] = dot2_i16(__builtin_amdgcn_perm(d[p1 >> 2], d[p0 >> 2], GS360_PAIR(p0 & 3, 4 + (p1 & 3))), wpk[m], a3[c]);
            }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (uint32_t)min(max((a3[c] + (1 << 14)) >> 15, 0), 255);
    return true;
}

// Split bilinear fetch for cv2 semantics (same idea as eq_fetch): the two row reads are issued unconditionally from
// a clamped, always-valid position so that a wavefront keeps all its gathers in flight; `fast` says the 2x2
// footprint was fully inside the image and the wide read stayed in-row, otherwise the pixel is redone afterwards by
// the straight-line border path (cv_sample_linear).  Needs W >= 8 and 32-bit tap offsets (checked on the host).
template <int C>
struct CvTaps {
    uint2 t0, t1;
    RowsRaw raw;    // C == 3: the loads in flight (cv_taps_finish)
    int fx, fy;
    bool fast;
};

// (sx, sy: the 1/32-pixel fixed point cv2.remap derives from the maps, cvRound(map * 32); a map plan holds them ready-made)
template <int C>
__device__ __forceinline__ CvTaps<C> cv_fetch_linear_fx(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                        const int sx, const int sy) {
    const int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
    constexpr int kBack = (C == 3) ? 5 : 2;
    CvTaps<C> t;
    t.fx = sx & 31;
    t.fy = sy & 31;
    t.fast = (uint32_t)ix <= (uint32_t)(W - kBack) && (uint32_t)iy < (uint32_t)(H - 1);       // both >= 0 and inside (W >= 8)
    // a window that is not `fast` is redone by the border sampler: its reads only have to be readable -- the image's first bytes
    const uint32_t o0 = t.fast ? __umul24((uint32_t)iy, (uint32_t)stride) + (uint32_t)ix * C : 0u;
    const uint32_t o1 = o0 + (t.fast ? (uint32_t)stride : 0u);
    const uint8_t* r0 = src + o0;
    const uint8_t* r1 = src + o1;
    if constexpr (C == 1) {
        uint16_t a, b;
        __builtin_memcpy(&a, r0, 2);
        __builtin_memcpy(&b, r1, 2);
        t.t0 = make_uint2(a, 0);
        t.t1 = make_uint2(b, 0);
    } else if constexpr (C == 3) {
        t.raw = ld_rows_rgb_issue(src, o0, o1);
    } else {
        t.t0 = ld_u64(r0);
        t.t1 = ld_u64(r1);
    }
    return t;
}
template <int C>
__device__ __forceinline__ CvTaps<C> cv_fetch_linear(const uint8_t* __restrict__ src, int64_t stride, int W, int H, float mx, float my) {
    return cv_fetch_linear_fx<C>(src, stride, W, H, cv_round(mx * 32.0f), cv_round(my * 32.0f));
}

template <int C>
__device__ __forceinline__ void cv_blend_fast(CvTaps<C>& t, uint32_t (&out)[4]) {
    if constexpr (C == 3) ld_rows_rgb_finish(t.raw, t.t0, t.t1);
    EqTaps<C> e;
    e.t0 = t.t0;
    e.t1 = t.t1;
    e.fix = false;
    eq_blend<C>(e, t.fx, t.fy, out);     // same 1/32-px weights: only the fractional bits of sx, sy are used
}

// cv2 bicubic for the wavefront's four row slots of an RGB image.  Windows that lie inside the image take the
// equirect kernel's path (four dword-aligned 16-byte row reads + the 32-byte weight entry issued together, dot-product
// blend); the others are redone by the straight-line border sampler.  Needs W >= 8 and 32-bit tap offsets (`pipelined`).
__device__ __forceinline__ void cv_cubic_slots_rgb(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                                   const float (&mxs)[kRowsPerWave], const float (&mys)[kRowsPerWave],
                                                   const uint8_t (&cval)[4], const int16_t* __restrict__ tab,
                                                   const int16_t* tab_lds, uint32_t (&px)[kRowsPerWave][4]) {
    // tab_lds: the workgroup's LDS copy of the table for the fast path (every lane reads another 32-byte entry: left in global
    // memory the 32 KiB table competes with the source lines for the 32 KiB vector L1 and costs two more gathers per pixel)
    static_assert(kRowsPerWave == 4, "four row slots");
    const bool stride4 = uniform_here((int)(stride & 3)) == 0;
    bool fast[4];
#pragma unroll
    for (int s0 = 0; s0 < 4; s0 += 2) {                   // two slots at a time: 8 row reads in flight, 24 tap dwords live
        EqCubicTaps t[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int sx = cv_round(mxs[s0 + u] * 32.0f), sy = cv_round(mys[s0 + u] * 32.0f);
            const int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
            fast[s0 + u] = ix >= 1 && iy >= 1 && ix <= W - 5 && iy <= H - 3;
            t[u] = cubic_issue_rgb(src, (uint32_t)stride, stride4, W, H, ix, iy, sx & 31, sy & 31);
        }
        __builtin_amdgcn_sched_barrier(0);
        eq_cubic_blend(t[0], tab_lds, stride4, px[s0]);
        eq_cubic_blend(t[1], tab_lds, stride4, px[s0 + 1]);
    }
    if (any_lane(!(fast[0] && fast[1] && fast[2] && fast[3]))) {
        // border windows: ONE copy of the straight-line sampler in a rolled loop, so that its 48 byte loads do not set the
        // register budget of the path above.  The loop always works on slot 0 and ROTATES the four slots after every turn (plain
        // register moves, back in place after four turns): picking the slot with `rr == k ? a[k] : ...` made the compiler keep the
        // coordinate arrays in scratch memory and store them there on the hot path of every tile (+4 B/px of writes, measured as
        // WRITE_SIZE 54 -> 108 MB per cfg4 launch).
        float x0 = mxs[0], x1 = mxs[1], x2 = mxs[2], x3 = mxs[3], y0 = mys[0], y1 = mys[1], y2 = mys[2], y3 = mys[3];
        bool f0 = fast[0], f1 = fast[1], f2 = fast[2], f3 = fast[3];
        uint32_t p0[3] = {px[0][0], px[0][1], px[0][2]}, p1[3] = {px[1][0], px[1][1], px[1][2]},
                 p2[3] = {px[2][0], px[2][1], px[2][2]}, p3[3] = {px[3][0], px[3][1], px[3][2]};
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            if (!f0) {
                uint32_t o[4];
                cv_sample_cubic<3>(src, stride, W, H, x0, y0, cval, tab, o);
                p0[0] = o[0]; p0[1] = o[1]; p0[2] = o[2];
            }
            const float tx = x0, ty = y0;
            const bool tf = f0;
            x0 = x1; x1 = x2; x2 = x3; x3 = tx;
            y0 = y1; y1 = y2; y2 = y3; y3 = ty;
            f0 = f1; f1 = f2; f2 = f3; f3 = tf;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t t = p0[c];
                p0[c] = p1[c]; p1[c] = p2[c]; p2[c] = p3[c]; p3[c] = t;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { px[0][c] = p0[c]; px[1][c] = p1[c]; px[2][c] = p2[c]; px[3][c] = p3[c]; }
    }
}

// One instantiation per interpolation: the 8x8 Lanczos window would otherwise set the register budget (and with it the
// occupancy) of the bilinear path.
template <int C, int INTERP>
__device__ __forceinline__ void table_remap_tile(const TableBatch& B, const int b, const int16_t* s_wtab) {
    int t = (b & 7) * B.chunk + (b >> 3);
    if (t >= B.total_tiles) return;
    int j = 0;
    while (j + 1 < B.n_jobs && t >= B.job[j + 1].tile_base) ++j;
    const TableLaunch& L = B.job[j];          // wave-uniform: fields are read from the kernel argument on demand
    t -= L.tile_base;
    const int tiles_x = L.tiles_x;
    int tile_y = t / tiles_x, tile_x = t - tile_y * tiles_x;
    // (behind an optimisation barrier: in the persistent variant the lane-derived constants would otherwise be hoisted out of the
    // tile loop and cost the kernel its fourth wavefront per SIMD)
    const int lane = lane_here(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const RowPack rp = make_row_pack(lane);
    const int x0 = tile_x * kTileW;
    const int n_px = min(kTileW, L.w - x0);
    const int xc = min(x0 + lane, L.w - 1);
    // (rows off a dword boundary take the byte stores here -- store_row<C, false>: the re-sliced dword path that pays in the equirect
    // kernels costs these kernels 12-15 %, cfg4 through plans 51 -> 59 us per pair)
    const bool aligned4 = ((L.dst_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(L.dst) & 3) == 0);
    // FLAT form (L.flat, set on the host for a tight, dword-aligned output whose rows are not whole dwords -- the tool's default 1750-pixel
    // views have 5250-byte rows, and a row that starts off a dword boundary leaves as three byte stores per pixel).  The output is
    // then addressed by the flat pixel number p = y w + x, as the maps, valid flags and plans are anyway, and "row y" becomes the span
    // [r(y), r(y + 1)) with r(y) = y w rounded up to a multiple of four (r(h) = h w): up to three pixels at the start of a row belong
    // to the span above.  Every span, and every 64-pixel tile cut from it, starts on a 12-byte = dword boundary of the tight output
    // whatever the width; a span is at most w + 3 pixels long (the host adds that to the tile count).  h w < 2^30: 32-bit indices.
    const bool flat = uniform_here(L.flat) != 0;
    const int n_flat = uniform_here(L.h * L.w);
    auto slot_span = [&](const int y, int& first, int& n) {           // first flat pixel of lane 0 and the pixels to store, for row slot y
        const int yc = min(y, L.h - 1);
        const int r0 = (yc * L.w + 3) & ~3;
        const int r1 = yc + 1 < L.h ? ((yc + 1) * L.w + 3) & ~3 : n_flat;
        first = r0 + x0;
        n = y < L.h ? max(0, min(kTileW, r1 - first)) : 0;
    };
    constexpr bool kFastCubic = (INTERP == GS360_INTERP_CUBIC) && (C == 3);
    if ((INTERP == GS360_INTERP_LINEAR || INTERP == GS360_INTERP_NEAREST || kFastCubic) && L.pipelined) {
        // maps of the wavefront's 4 rows -> all gathers in flight -> blend -> border/valid fix-ups -> packed stores
        const int ybase = tile_y * kTileH + wave * kRowsPerWave;
        float mxs[kRowsPerWave], mys[kRowsPerWave];
        bool inval[kRowsPerWave];
        // the twelve map / valid reads of the four row slots go out together: behind a run-time `if (L.valid)` the compiler waits
        // for each valid byte (and with it for the slot's map reads) before it issues the next slot's -- four serial round trips
        // per tile.  Without a valid map the byte is read from the map itself (h * w readable bytes) and ignored.
        const uint8_t* __restrict__ vptr = L.valid ? L.valid : reinterpret_cast<const uint8_t*>(L.map_x);
        const bool has_valid = L.valid != nullptr;
        uint8_t vbyte[kRowsPerWave];
        int sxi[kRowsPerWave] = {0, 0, 0, 0}, syi[kRowsPerWave] = {0, 0, 0, 0};   // bilinear: the positions in 1/32-pixel fixed point
        int first[kRowsPerWave], n_st[kRowsPerWave];      // flat form: the slot's first pixel and its pixel count (wave-uniform)
        uint32_t idx[kRowsPerWave];                       // this lane's map entry (clamped to a readable one)
#pragma unroll
        for (int rr = 0; rr < kRowsPerWave; ++rr) {
            if (flat) {
                slot_span(ybase + rr, first[rr], n_st[rr]);
                idx[rr] = (uint32_t)min(first[rr] + lane, n_flat - 1);
            } else {
                first[rr] = 0;
                n_st[rr] = 0;
                idx[rr] = (uint32_t)(min(ybase + rr, L.h - 1) * L.w + xc);
            }
        }
        if (L.packed) {                       // a map plan (wave-uniform): one dword and one byte per pixel
            uint32_t pw[kRowsPerWave];
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) {
                const uint32_t o = idx[rr];
                pw[rr] = L.packed[o];
                vbyte[rr] = L.packed_hi[o];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) {
                planned_coords(pw[rr], vbyte[rr], INTERP == GS360_INTERP_NEAREST, mxs[rr], mys[rr]);
                inval[rr] = (L.use_valid != 0) & ((vbyte[rr] & 4) == 0);
                // the bilinear fetch takes the fixed point as it is packed (the floats are for the border / bicubic paths)
                sxi[rr] = ((int)(pw[rr] & 0xfffu) - 8) * 32 + (int)((pw[rr] >> 24) & 31u);
                syi[rr] = ((int)((pw[rr] >> 12) & 0xfffu) - 8) * 32 + (int)((pw[rr] >> 29) | ((vbyte[rr] & 3u) << 3));
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) {
                const uint32_t o = idx[rr];
                mxs[rr] = L.map_x[o];         // (non-temporal map loads were measured: no difference)
                mys[rr] = L.map_y[o];
                vbyte[rr] = vptr[o];
            }
            __builtin_amdgcn_sched_barrier(0);    // all twelve in flight before anything else is scheduled
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) {
                inval[rr] = has_valid & (vbyte[rr] == 0);
                if constexpr (INTERP == GS360_INTERP_LINEAR) {
                    sxi[rr] = cv_round(mxs[rr] * 32.0f);
                    syi[rr] = cv_round(mys[rr] * 32.0f);
                }
            }
        }
        uint32_t px[kRowsPerWave][4];
        if constexpr (kFastCubic) {
            cv_cubic_slots_rgb(L.src, L.src_stride, L.W, L.H, mxs, mys, L.cval, L.cubic_tab, s_wtab, px);
        } else if constexpr (INTERP == GS360_INTERP_NEAREST) {   // mask cutting (DF:2031-2043): the 4 slots' reads in flight together
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) cv_sample_nearest<C>(L.src, L.src_stride, L.W, L.H, mxs[rr], mys[rr], L.cval, px[rr]);
        } else {
            CvTaps<C> taps[kRowsPerWave];
            bool any_slow = false;
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) {
                taps[rr] = cv_fetch_linear_fx<C>(L.src, L.src_stride, L.W, L.H, sxi[rr], syi[rr]);
                any_slow |= !taps[rr].fast;
            }
            __builtin_amdgcn_sched_barrier(0);    // every gather of the wavefront's four rows in flight before the first is consumed
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr) cv_blend_fast<C>(taps[rr], px[rr]);
            if (any_lane(any_slow)) {
#pragma unroll
                for (int rr = 0; rr < kRowsPerWave; ++rr)
                    if (!taps[rr].fast) cv_sample_linear<C>(L.src, L.src_stride, L.W, L.H, mxs[rr], mys[rr], L.cval, px[rr]);
            }
        }
#pragma unroll
        for (int rr = 0; rr < kRowsPerWave; ++rr) {
            if (inval[rr]) {
#pragma unroll
                for (int c = 0; c < C; ++c) px[rr][c] = (uint32_t)L.fill;
            }
            const int y = ybase + rr;
            if (flat) {
                if (n_st[rr] > 0) store_row<C>(L.dst + (size_t)(uint32_t)(first[rr] * C), px[rr], n_st[rr], true, rp);
            } else if (y < L.h) {
                store_row<C, false>(L.dst + (int64_t)y * L.dst_stride + (int64_t)x0 * C, px[rr], n_px, aligned4, rp);
            }
        }
        return;
    }
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int y = tile_y * kTileH + wave * kRowsPerWave + rr;
        if (y >= L.h) break;
        int first = 0, n_st = 0;
        if (flat) {
            slot_span(y, first, n_st);
            if (n_st == 0) continue;
        }
        int64_t o = flat ? (int64_t)min(first + lane, n_flat - 1) : (int64_t)y * L.w + xc;
        float mx, my;
        bool inval;
        if (L.packed) {
            const uint32_t hb = L.packed_hi[o];
            planned_coords(L.packed[o], hb, INTERP == GS360_INTERP_NEAREST, mx, my);
            inval = L.use_valid && !(hb & 4);
        } else {
            mx = L.map_x[o]; my = L.map_y[o];        // 256 B per wavefront row, coalesced
            inval = L.valid && !L.valid[o];
        }
        uint32_t px[4];
        if constexpr (INTERP == GS360_INTERP_LINEAR) cv_sample_linear<C>(L.src, L.src_stride, L.W, L.H, mx, my, L.cval, px);
        else if constexpr (INTERP == GS360_INTERP_CUBIC) cv_sample_cubic<C>(L.src, L.src_stride, L.W, L.H, mx, my, L.cval, L.cubic_tab, px);
        else if constexpr (INTERP == GS360_INTERP_LANCZOS4) {
            // 64 taps: pixels the valid map rules out are not sampled at all.  (The same test in front of the cheaper samplers
            // made the compiler index the RGBA bicubic accumulators through scratch memory.)
            if (!inval) {
                bool done = false;
                if constexpr (C == 3) {
                    if (L.lz_c1) done = cv_lanczos4_rgb_rebuilt(L.src, L.src_stride, L.W, L.H, mx, my, reinterpret_cast<const float*>(s_wtab), px);
                }
                if (!done) cv_sample_lanczos4<C>(L.src, L.src_stride, L.W, L.H, mx, my, L.cval, L.cubic_tab, px);
            }
        } else cv_sample_nearest<C>(L.src, L.src_stride, L.W, L.H, mx, my, L.cval, px);
        if (inval) {
#pragma unroll
            for (int c = 0; c < C; ++c) px[c] = (uint32_t)L.fill;
        }
        if (flat) store_row<C>(L.dst + (size_t)(uint32_t)(first * C), px, n_st, true, rp);
        else store_row<C, false>(L.dst + (int64_t)y * L.dst_stride + (int64_t)x0 * C, px, n_px, aligned4, rp);
    }
}

// Bicubic RGB keeps a 32 KiB LDS copy of the weight table: filled once per workgroup, so its workgroups are PERSISTENT (the
// launcher caps the grid at a few workgroups per CU and each walks tiles b, b + gridDim.x, ... -- the stride is a multiple of 8,
// so a workgroup stays inside its XCD's chunk of tiles).  One 64 x 16 tile per workgroup meant 32 bytes of table fill per
// output pixel: as many bytes as the pixel's own weight entry, eight more 1 KiB loads per wavefront next to its sixteen row
// gathers, and a load -> LDS -> barrier bubble in front of every tile.
template <int C, int INTERP>
__global__ __launch_bounds__(64 * kWaves) void table_remap_kernel(const TableBatch B) {
    constexpr bool kFastCubic = (INTERP == GS360_INTERP_CUBIC) && (C == 3);
    constexpr bool kLanczosRgb = (INTERP == GS360_INTERP_LANCZOS4) && (C == 3);   // 256 floats + 2048 dwords (cv_lanczos4_rgb_rebuilt)
    __shared__ __attribute__((aligned(16))) int16_t s_wtab[kFastCubic ? 32 * 32 * 16 : (kLanczosRgb ? (256 + 2048) * 2 : 8)];
    if constexpr (kFastCubic) {
        if (B.job[0].cubic_tab) {             // (the context's table: the same pointer in every job)
            cubic_lds_fill(s_wtab, B.job[0].cubic_tab, 64 * kWaves);
            __syncthreads();
        }
#pragma unroll 1
        for (int b = blockIdx.x; b < B.chunk * 8; b += gridDim.x) table_remap_tile<C, INTERP>(B, b, s_wtab);
    } else if constexpr (kLanczosRgb) {
        if (B.job[0].lz_c1) {                 // (wave-uniform; the context's tables, the same in every job)
            uint32_t* l = reinterpret_cast<uint32_t*>(s_wtab);
            l[threadIdx.x] = reinterpret_cast<const uint32_t*>(B.job[0].lz_c1)[threadIdx.x];
#pragma unroll
            for (int i = 0; i < 2048 / (64 * kWaves); ++i) l[256 + i * 64 * kWaves + threadIdx.x] = B.job[0].lz_cen[i * 64 * kWaves + threadIdx.x];
            __syncthreads();
        }
#pragma unroll 1
        for (int b = blockIdx.x; b < B.chunk * 8; b += gridDim.x) table_remap_tile<C, INTERP>(B, b, s_wtab);
    } else {
        table_remap_tile<C, INTERP>(B, blockIdx.x, s_wtab);
    }
}

// ------------------------------------------------------------------------------------------------
// FE-SPEC v1: fused fisheye -> perspective
// ------------------------------------------------------------------------------------------------
// All views of a call in one launch.  The view block is used through a REFERENCE into the by-value kernel argument
// (scalar loads on demand): copying a dynamically indexed 148-byte block into a local made the compiler spill the whole
// argument array to scratch (2368 B/lane, 13x slower).
template <int C, int INTERP>
__device__ __forceinline__ void fe_views_tile(const FeBatch& B, const int b, const int16_t* s_wtab) {
    const FeCommon& L = B.common;
    int t = (b & 7) * L.chunk + (b >> 3);
    if (t >= L.total_tiles) return;
    int j = 0;
    while (j + 1 < L.n_views && t >= B.view[j + 1].tile_base) ++j;
    const FeView& V = B.view[j];              // a reference: fields are fetched from the kernel argument on demand
    t -= V.tile_base;
    int tile_y = t / V.tiles_x, tile_x = t - tile_y * V.tiles_x;
    constexpr bool kFastCubic = (INTERP == GS360_INTERP_CUBIC) && (C == 3);
    // (persistent variant: lane-derived constants stay inside the tile, see table_remap_tile)
    const int lane = kFastCubic ? lane_here() : (int)(threadIdx.x & 63), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x0 = tile_x * kTileW;
    const int n_px = min(kTileW, V.out_w - x0);
    const int xc = min(x0 + lane, V.out_w - 1);
    const int64_t dstride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * C;
    const bool aligned4 = ((dstride & 3) == 0) && ((reinterpret_cast<uintptr_t>(V.dst) & 3) == 0);
    const RowPack rp = make_row_pack(lane);
    const float x = (float)(2 * xc + 1 - V.out_w) * V.sxu;

    const int ybase = tile_y * kTileH + wave * kRowsPerWave;
    const bool pipelined = (INTERP == GS360_INTERP_LINEAR) && L.pipelined;
    float mxs[kRowsPerWave], mys[kRowsPerWave];
    bool oks[kRowsPerWave];
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int y = min(ybase + rr, V.out_h - 1);
        float yv = (float)(2 * y + 1 - V.out_h) * V.syv;       // ray y = -yv
        float Y = __builtin_fmaf(-V.cp, yv, V.sp);
        float z1 = __builtin_fmaf(V.sp, yv, V.cp);
        float X = __builtin_fmaf(V.cy, x, V.sy * z1);
        float Z = __builtin_fmaf(-V.sy, x, V.cy * z1);
        // N >= 1; d = N (N + Z) is 0 or >= ~1e-7 and <= ~1e7, so 2 / d and both square roots stay far from the denormal /
        // overflow ranges in which the generic IEEE expansions differ from the reduced ones (gs360_eqspec.h)
        float N = eq_sqrt_normal(__builtin_fmaf(x, x, __builtin_fmaf(yv, yv, 1.0f)));
        float d = N * (N + Z);
        float s = d > 0.0f ? eq_sqrt_normal(eq_div(2.0f, d)) : 0.0f;
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
        mxs[rr] = mx;
        mys[rr] = my;
        // bitwise: short-circuit && turned into nested exec-mask branches with scalar loads inside each of the four slots
        oks[rr] = (Z >= V.cos_tmax * N) & (mx >= 0.0f) & (mx <= V.wmax) & (my >= 0.0f) & (my <= V.hmax);
    }
    uint32_t px[kRowsPerWave][4];
    if ((INTERP == GS360_INTERP_CUBIC) && (C == 3) && L.pipelined) {
        if constexpr (C == 3) cv_cubic_slots_rgb(V.src, L.src_stride, V.W, V.H, mxs, mys, L.cval, L.cubic_tab, s_wtab, px);
    } else if (pipelined) {
        CvTaps<C> taps[kRowsPerWave];
        bool any_slow = false;
#pragma unroll
        for (int rr = 0; rr < kRowsPerWave; ++rr) {
            taps[rr] = cv_fetch_linear<C>(V.src, L.src_stride, V.W, V.H, mxs[rr], mys[rr]);
            any_slow |= !taps[rr].fast;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < kRowsPerWave; ++rr) cv_blend_fast<C>(taps[rr], px[rr]);
        if (any_lane(any_slow)) {
#pragma unroll
            for (int rr = 0; rr < kRowsPerWave; ++rr)
                if (!taps[rr].fast) cv_sample_linear<C>(V.src, L.src_stride, V.W, V.H, mxs[rr], mys[rr], L.cval, px[rr]);
        }
    } else {
        // one copy of the straight-line sampler in a rolled loop; the slot is picked with wave-uniform selects so that the
        // coordinate / pixel arrays stay in registers
        static_assert(kRowsPerWave == 4, "four row slots");
#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const float mx = rr == 0 ? mxs[0] : rr == 1 ? mxs[1] : rr == 2 ? mxs[2] : mxs[3];
            const float my = rr == 0 ? mys[0] : rr == 1 ? mys[1] : rr == 2 ? mys[2] : mys[3];
            uint32_t o[4];
            if constexpr (INTERP == GS360_INTERP_LINEAR) cv_sample_linear<C>(V.src, L.src_stride, V.W, V.H, mx, my, L.cval, o);
            else if constexpr (INTERP == GS360_INTERP_CUBIC) cv_sample_cubic<C>(V.src, L.src_stride, V.W, V.H, mx, my, L.cval, L.cubic_tab, o);
            else if constexpr (INTERP == GS360_INTERP_LANCZOS4) cv_sample_lanczos4<C>(V.src, L.src_stride, V.W, V.H, mx, my, L.cval, L.cubic_tab, o);
            else cv_sample_nearest<C>(V.src, L.src_stride, V.W, V.H, mx, my, L.cval, o);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (rr == k) px[k][c] = o[c];
        }
    }
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int y = ybase + rr;
        if (y >= V.out_h) break;
        if (!oks[rr] && L.mask_outside) {
#pragma unroll
            for (int c = 0; c < C; ++c) px[rr][c] = (uint32_t)L.mask_value;
        }
        store_row<C, false>(V.dst + (int64_t)y * dstride + (int64_t)x0 * C, px[rr], n_px, aligned4, rp);
        if (V.valid_out && lane < n_px) V.valid_out[(int64_t)y * V.out_w + x0 + lane] = oks[rr] ? 1 : 0;
    }
}

// (bicubic RGB: persistent workgroups around one LDS weight-table fill, as in table_remap_kernel)
template <int C, int INTERP>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu((INTERP == GS360_INTERP_CUBIC && C == 3) ? 4 : 1, 8)))
void fe_views_kernel(const FeBatch B) {
    constexpr bool kFastCubic = (INTERP == GS360_INTERP_CUBIC) && (C == 3);
    __shared__ __attribute__((aligned(16))) int16_t s_wtab[kFastCubic ? 32 * 32 * 16 : 8];
    if constexpr (kFastCubic) {
        if (B.common.pipelined) {
            cubic_lds_fill(s_wtab, B.common.cubic_tab, 64 * kWaves);
            __syncthreads();
        }
#pragma unroll 1
        for (int b = blockIdx.x; b < B.common.grid_total; b += gridDim.x) fe_views_tile<C, INTERP>(B, b, s_wtab);
    } else {
        fe_views_tile<C, INTERP>(B, blockIdx.x, s_wtab);
    }
}

namespace {

template <int C>
void launch_table_c(const TableBatch& B, dim3 grid, dim3 block, hipStream_t s) {
    switch (B.job[0].interp) {
        case GS360_INTERP_LINEAR: hipLaunchKernelGGL((table_remap_kernel<C, GS360_INTERP_LINEAR>), grid, block, 0, s, B); break;
        case GS360_INTERP_CUBIC: hipLaunchKernelGGL((table_remap_kernel<C, GS360_INTERP_CUBIC>), grid, block, 0, s, B); break;
        case GS360_INTERP_LANCZOS4: hipLaunchKernelGGL((table_remap_kernel<C, GS360_INTERP_LANCZOS4>), grid, block, 0, s, B); break;
        default: hipLaunchKernelGGL((table_remap_kernel<C, GS360_INTERP_NEAREST>), grid, block, 0, s, B); break;
    }
}

template <int C>
void launch_fisheye_c(const FeBatch& B, dim3 grid, dim3 block, hipStream_t s) {
    switch (B.common.interp) {
        case GS360_INTERP_LINEAR: hipLaunchKernelGGL((fe_views_kernel<C, GS360_INTERP_LINEAR>), grid, block, 0, s, B); break;
        case GS360_INTERP_CUBIC: hipLaunchKernelGGL((fe_views_kernel<C, GS360_INTERP_CUBIC>), grid, block, 0, s, B); break;
        case GS360_INTERP_LANCZOS4: hipLaunchKernelGGL((fe_views_kernel<C, GS360_INTERP_LANCZOS4>), grid, block, 0, s, B); break;
        default: hipLaunchKernelGGL((fe_views_kernel<C, GS360_INTERP_NEAREST>), grid, block, 0, s, B); break;
    }
}

}  // namespace

hipError_t launch_table_batch(TableBatch& B, int C, hipStream_t s) {
    int base = 0;
    for (int j = 0; j < B.n_jobs; ++j) {
        TableLaunch& L = B.job[j];
        L.tiles_x = (L.w + (L.flat ? 3 : 0) + kTileW - 1) / kTileW;      // (flat form: a row's span may be three pixels longer)
        L.tile_base = base;
        base += L.tiles_x * ((L.h + kTileH - 1) / kTileH);
    }
    B.total_tiles = base;
    B.chunk = (base + 7) / 8;
    if (base == 0) return hipSuccess;
    dim3 grid((unsigned)(B.chunk * 8)), block(64 * kWaves);
    // persistent workgroups for the kernel with a per-workgroup LDS table (see table_remap_kernel)
    if (C == 3 && (B.job[0].interp == GS360_INTERP_CUBIC || B.job[0].interp == GS360_INTERP_LANCZOS4) && B.persist_blocks > 0 &&
        (unsigned)B.persist_blocks < grid.x)
        grid.x = (unsigned)(B.persist_blocks + 7) & ~7u;
    switch (C) {
        case 1: launch_table_c<1>(B, grid, block, s); break;
        case 3: launch_table_c<3>(B, grid, block, s); break;
        case 4: launch_table_c<4>(B, grid, block, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_map_pack(const float* map_x, const float* map_y, const uint8_t* valid, int64_t n, int nearest,
                           uint32_t* packed, uint8_t* packed_hi, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(map_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, map_x, map_y, valid, n, nearest, packed, packed_hi);
    return hipGetLastError();
}

hipError_t launch_table(const TableLaunch& L, int C, hipStream_t s) {
    TableBatch B;
    B.job[0] = L;
    B.n_jobs = 1;
    B.persist_blocks = 0;
    return launch_table_batch(B, C, s);
}

hipError_t launch_fisheye(const FeLaunch& L, int C, hipStream_t s) {
    FeBatch B;
    int base = 0;
    for (int k = 0; k < L.n_views; ++k) {
        B.view[k] = L.view[k];
        B.view[k].tile_base = base;
        base += L.view[k].tiles_x * L.view[k].tiles_y;
    }
    FeCommon& K = B.common;
    K.n_views = L.n_views;
    K.total_tiles = base;
    K.chunk = (base + 7) / 8;
    K.interp = L.interp; K.mask_outside = L.mask_outside; K.mask_value = L.mask_value;
    K.src_stride = L.src_stride; K.dst_stride = L.dst_stride;
    for (int i = 0; i < 4; ++i) K.cval[i] = L.cval[i];
    K.cubic_tab = L.cubic_tab;
    K.pipelined = L.pipelined;
    if (base == 0) return hipSuccess;
    dim3 grid((unsigned)(K.chunk * 8)), block(64 * kWaves);
    K.grid_total = (int32_t)grid.x;
    if (C == 3 && L.interp == GS360_INTERP_CUBIC && L.persist_blocks > 0 && (unsigned)L.persist_blocks < grid.x)
        grid.x = (unsigned)(L.persist_blocks + 7) & ~7u;
    switch (C) {
        case 1: launch_fisheye_c<1>(B, grid, block, s); break;
        case 3: launch_fisheye_c<3>(B, grid, block, s); break;
        case 4: launch_fisheye_c<4>(B, grid, block, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace gs360
