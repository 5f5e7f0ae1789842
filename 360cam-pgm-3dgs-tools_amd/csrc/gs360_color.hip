// gs360_color.hip -- the dual-fisheye tool's input colour stage on the GPU (include/gs360.h, "input colour stage").
//
// Reference: apply_input_color_pipeline, cli_tools/gs360_DualFisheyeDistortionCalibration.py:684-725 --
// u8 -> float01 (DF:603-613) -> .cube trilinear (DF:620-681) -> rec709_to_srgb (DF:565-600, optional) -> u8 (DF:616-618).
// The two scalar ends (level -> grid position, LUT output -> 8-bit level) arrive as host-built tables (256 x 3 floats
// and 255 thresholds, see the header); the interpolation in between is done here in the reference's float32
// operation order (sub, mul, add -- never fused: the library is built with -ffp-contract=off), so the result is the
// byte the NumPy pipeline produces.
//
// Layout.  The red channel has only 256 possible grid positions, so the first of the three interpolation stages
// (along red, DF:672-675) is evaluated once per plan for every (blue node, green node, red LEVEL) with exactly the
// reference's operations.  The results are stored CELL-major: for every (blue cell, green cell, red level) the four
// corners the green and blue stages need -- c00, c10, c01, c11, float3 each -- sit next to each other in one 64-byte
// entry (48 bytes used; n*n*256 entries: 17.8 MB for a 33^3 cube, Infinity-Cache resident).  A pixel then reads ONE
// aligned 48-byte piece of one cache line instead of four 12-byte pieces from four table rows (round 3: the stage ran at
// 0.18 of the roofline on a smooth image and 0.04 on noise, where every pixel pulled four lines).  The output quantiser is a 1024-bin lower-bound table plus
// one or two threshold compares (exact: the bin table is derived from the thresholds), falling back to a binary
// search when the thresholds are too dense for that.
//
// The cube.  An 8-bit pixel has 2^24 possible values and the whole stage is a pure function of the value, so a plan evaluates that
// function ONCE for every value (color_cube_kernel: the pipeline above, 16.7 M evaluations -- one 4096 x 4096 image's worth of work) into
// a 64 MiB table of output pixels indexed by the input pixel, (blue << 16 | green << 8 | red) -> red | green << 8 | blue << 16.  Applying a
// plan is then ONE dword read per pixel and no arithmetic (color_cube_quad_kernel / color_cube_bytes_kernel); the table sits in HBM (a
// 4400th of it), the part of it an image touches -- a natural image occupies a small fraction of the colour cube -- in the L2 / Infinity
// Cache.  Results are those of the evaluating kernels by construction.  GS360_COLOR_CUBE=0 at plan creation keeps the per-pixel
// evaluation (A/B, or to save the 64 MiB).
#include <cstring>
#include <vector>

#include "gs360_kernels.h"

namespace gs360 {

namespace {

constexpr int kColorThreads = 256;
constexpr int kBins = 1024;
constexpr int kColorBlocks = 2048;   // persistent blocks of the dword-aligned path: 256 CUs x 8

struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

struct __attribute__((aligned(16))) CellEntry { float v[16]; };   // c00.xyz c10.xyz c01.xyz c11.xyz + 4 floats of padding

struct ColorArgs {
    const uint8_t* src;
    uint8_t* dst;
    const CellEntry* rtab;  // [blue cell][green cell][red level]
    const float* tables;    // 768 level positions (R,G,B x 256), 256 thresholds, then kBins/4 dwords of packed bin levels
    int32_t H, W;
    int32_t n;              // LUT edge length
    int32_t red;            // memory index of the red channel (0 or 2)
    int64_t src_stride, dst_stride;
};

struct Cell {               // one channel's LUT cell: lower index, upper index, weight of the upper node
    int i0, i1;
    float f;
};

__device__ __forceinline__ Cell cell_of(float pos, int nmax) {
    // DF:651-653: idx0 = floor(pos); idx1 = min(idx0 + 1, max_index); frac = pos - idx0
    const float fl = floorf(pos);
    Cell c;
    c.i0 = (int)fl;
    c.i1 = min(c.i0 + 1, nmax);
    c.f = pos - fl;
    return c;
}

__device__ __forceinline__ float lerp_ref(float a, float b, float t) { return a + (b - a) * t; }   // DF:672-679

struct Lds {
    float pos[768];
    float thr[260];          // [0] unused, [1..255] thresholds, [256..] = +inf sentinels
    uint8_t bin[kBins];
};

// number of thresholds <= clip(x, 0, 1) among thr[1..255] (non-decreasing); NaN counts as 0
template <int FIX>
__device__ __forceinline__ int level_of(const Lds& S, float x) {
    const float xc = fminf(fmaxf(x, 0.0f), 1.0f);                       // fmaxf(NaN, 0) = 0
    if (FIX == 0) {          // general: binary search
        int lv = 0;
#pragma unroll
        for (int bit = 128; bit > 0; bit >>= 1) {
            const int cand = lv | bit;
            lv = (xc >= S.thr[cand]) ? cand : lv;
        }
        return lv;
    }
    int lv = S.bin[min((int)(xc * (float)kBins), kBins - 1)];           // thresholds <= the bin's lower edge
#pragma unroll
    for (int k = 0; k < FIX; ++k) lv += (xc >= S.thr[lv + 1]) ? 1 : 0;  // at most FIX thresholds lie inside a bin
    return lv;
}

struct Rgb8 { int r, g, b; };

struct Taps {               // the four red-interpolated table entries around a pixel and its green/blue weights
    F3 c00, c10, c01, c11;
    float gf, bf;
};

// first half of the pipeline for one pixel: locate the cell and issue the table read (three aligned 16-byte loads of one entry)
__device__ __forceinline__ Taps color_fetch(const ColorArgs& A, const Lds& S, int vr, int vg, int vb) {
    const int n = A.n, nmax = n - 1;
    const Cell g = cell_of(S.pos[256 + vg], nmax);
    const Cell b = cell_of(S.pos[512 + vb], nmax);
    const char* base = (const char*)A.rtab;        // the table is < 4 GiB: 32-bit byte offsets from a scalar base
    const float4* e = (const float4*)(base + (size_t)(((((uint32_t)(b.i0 * n + g.i0)) << 8) + (uint32_t)vr) << 6));
    const float4 q0 = e[0], q1 = e[1], q2 = e[2];
    Taps t;
    t.c00 = {q0.x, q0.y, q0.z};
    t.c10 = {q0.w, q1.x, q1.y};
    t.c01 = {q1.z, q1.w, q2.x};
    t.c11 = {q2.y, q2.z, q2.w};
    t.gf = g.f;
    t.bf = b.f;
    return t;
}

// second half: green stage, blue stage (DF:676-679), quantise
template <int FIX>
__device__ __forceinline__ Rgb8 color_finish(const Lds& S, const Taps& t) {
    Rgb8 q;
    q.r = level_of<FIX>(S, lerp_ref(lerp_ref(t.c00.x, t.c10.x, t.gf), lerp_ref(t.c01.x, t.c11.x, t.gf), t.bf));
    q.g = level_of<FIX>(S, lerp_ref(lerp_ref(t.c00.y, t.c10.y, t.gf), lerp_ref(t.c01.y, t.c11.y, t.gf), t.bf));
    q.b = level_of<FIX>(S, lerp_ref(lerp_ref(t.c00.z, t.c10.z, t.gf), lerp_ref(t.c01.z, t.c11.z, t.gf), t.bf));
    return q;
}

template <int FIX>
__device__ __forceinline__ Rgb8 color_px(const ColorArgs& A, const Lds& S, int vr, int vg, int vb) {
    return color_finish<FIX>(S, color_fetch(A, S, vr, vg, vb));
}

__device__ __forceinline__ void load_tables(const ColorArgs& A, Lds& S) {
    constexpr int kPer = (1024 + kBins / 4) / kColorThreads;               // 5 dwords per thread, all in flight at once
    static_assert((1024 + kBins / 4) % kColorThreads == 0, "table size");
    float v[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) v[j] = A.tables[j * kColorThreads + threadIdx.x];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int i = j * kColorThreads + threadIdx.x;
        if (i < 768) S.pos[i] = v[j];
        else if (i < 1024) S.thr[i - 768] = v[j];
        else ((float*)S.bin)[i - 1024] = v[j];
    }
    if (threadIdx.x < 4) S.thr[256 + threadIdx.x] = __builtin_inff();
    __syncthreads();
}

// Any alignment: one thread per pixel, byte loads and stores.
template <int C, int FIX>
__global__ __launch_bounds__(kColorThreads) void color_lut_bytes_kernel(ColorArgs A) {
    __shared__ Lds S;
    load_tables(A, S);
    const int x = blockIdx.x * kColorThreads + threadIdx.x;
    if (x >= A.W) return;
    const uint8_t* sp = A.src + (int64_t)blockIdx.y * A.src_stride + (int64_t)x * C;
    uint8_t* dp = A.dst + (int64_t)blockIdx.y * A.dst_stride + (int64_t)x * C;
    const int iR = A.red, iB = 2 - A.red;
    const int alpha = (C == 4) ? sp[3] : 0;
    const Rgb8 q = color_px<FIX>(A, S, sp[iR], sp[1], sp[iB]);
    dp[iR] = (uint8_t)q.r; dp[1] = (uint8_t)q.g; dp[iB] = (uint8_t)q.b;
    if (C == 4) dp[3] = (uint8_t)alpha;
}

// Rows that start on a dword boundary: one thread per 4 pixels = C dwords in, C dwords out, so a wavefront moves
// 768 (C=3) or 1024 (C=4) contiguous bytes per row segment.  Blocks are persistent (the LDS tables are loaded once
// per block) and walk 1024-pixel row segments in row-major order.
constexpr int kColorWaves = 5;
template <int C, int FIX>
__global__ __launch_bounds__(kColorThreads) __attribute__((amdgpu_waves_per_eu(kColorWaves, kColorWaves))) void color_lut_quad_kernel(ColorArgs A) {
    __shared__ Lds S;
    load_tables(A, S);
    const int iR = A.red, iB = 2 - A.red;
    const bool bgr = A.red != 0;
    const int segs = (A.W + 4 * kColorThreads - 1) / (4 * kColorThreads);
    const int total = segs * A.H;
    // the image dwords of a thread's NEXT row segment are requested before the current one is worked on: of the chain image read ->
    // level tables (LDS) -> entry read -> blend -> quantiser tables (LDS) -> store, the first link then costs nothing
    auto seg_src = [&](int t, int& y, int& x) {
        y = t / segs;
        x = ((t - y * segs) * kColorThreads + threadIdx.x) * 4;
        return A.src + (int64_t)y * A.src_stride + (int64_t)x * C;
    };
    uint32_t wn[C];
#pragma unroll
    for (int i = 0; i < C; ++i) wn[i] = 0;
    {
        int y0, x0;
        const uint8_t* p0 = seg_src(blockIdx.x, y0, x0);
        if (blockIdx.x < (unsigned)total && x0 + 4 <= A.W) {
#pragma unroll
            for (int i = 0; i < C; ++i) wn[i] = ((const uint32_t*)p0)[i];
        }
    }
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int y, x;
        const uint8_t* sp = seg_src(t, y, x);
        uint32_t w[C];
#pragma unroll
        for (int i = 0; i < C; ++i) w[i] = wn[i];
        {
            int yn, xn;
            const int tn = t + (int)gridDim.x;
            const uint8_t* pn = seg_src(tn, yn, xn);
            if (tn < total && xn + 4 <= A.W) {
#pragma unroll
                for (int i = 0; i < C; ++i) wn[i] = ((const uint32_t*)pn)[i];
            }
        }
        if (x >= A.W) continue;
        uint8_t* dp = A.dst + (int64_t)y * A.dst_stride + (int64_t)x * C;
        if (x + 4 <= A.W) {
            Taps taps[4];                          // all 16 table reads of the four pixels are in flight together
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                int v[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = p * C + c;
                    v[c] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xff;
                }
                taps[p] = color_fetch(A, S, bgr ? v[2] : v[0], v[1], bgr ? v[0] : v[2]);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from sinking reads below the first blend
            uint32_t o[C];
#pragma unroll
            for (int i = 0; i < C; ++i) o[i] = (C == 4) ? (w[i] & 0xff000000u) : 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const Rgb8 q = color_finish<FIX>(S, taps[p]);
                const int out3[3] = {bgr ? q.b : q.r, q.g, bgr ? q.r : q.b};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = p * C + c;
                    o[byte >> 2] |= (uint32_t)out3[c] << (8 * (byte & 3));
                }
            }
            uint32_t* d32 = (uint32_t*)dp;
#pragma unroll
            for (int i = 0; i < C; ++i) d32[i] = o[i];
        } else {
            for (int p = 0; x + p < A.W; ++p) {
                const uint8_t* s1 = sp + p * C;
                uint8_t* d1 = dp + p * C;
                const int alpha = (C == 4) ? s1[3] : 0;
                const Rgb8 q = color_px<FIX>(A, S, s1[iR], s1[1], s1[iB]);
                d1[iR] = (uint8_t)q.r; d1[1] = (uint8_t)q.g; d1[iB] = (uint8_t)q.b;
                if (C == 4) d1[3] = (uint8_t)alpha;
            }
        }
    }
}

// ---- the cube: every possible 8-bit pixel through the pipeline once, then one read per pixel ---------------------------------------
struct CubeArgs {
    const uint8_t* src;
    uint8_t* dst;
    const uint32_t* cube;   // [blue][green][red] -> red | green << 8 | blue << 16
    int32_t H, W;
    int32_t red;            // memory index of the red channel (0 or 2)
    int64_t src_stride, dst_stride;
};

template <int FIX>
__global__ __launch_bounds__(kColorThreads) void color_cube_kernel(ColorArgs A, uint32_t* cube) {
    __shared__ Lds S;
    load_tables(A, S);
    for (uint32_t i = blockIdx.x * kColorThreads + threadIdx.x; i < (1u << 24); i += gridDim.x * kColorThreads) {
        const Rgb8 q = color_px<FIX>(A, S, (int)(i & 255u), (int)((i >> 8) & 255u), (int)(i >> 16));
        cube[i] = (uint32_t)q.r | ((uint32_t)q.g << 8) | ((uint32_t)q.b << 16);
    }
}

// v_perm_b32 selectors (bytes 0-3 = the second operand, 0x0c = a zero byte): keep the three colour bytes of a pixel in memory order
// (red first) or swap the outer two (blue first); the swap is its own inverse, so the same selector turns a table entry back into
// memory order
__device__ __forceinline__ uint32_t cube_selector(int red) { return red != 0 ? 0x0c000102u : 0x0c020100u; }

template <int C>
__global__ __launch_bounds__(kColorThreads) void color_cube_bytes_kernel(CubeArgs A) {
    const int x = blockIdx.x * kColorThreads + threadIdx.x;
    if (x >= A.W) return;
    const uint8_t* sp = A.src + (int64_t)blockIdx.y * A.src_stride + (int64_t)x * C;
    uint8_t* dp = A.dst + (int64_t)blockIdx.y * A.dst_stride + (int64_t)x * C;
    const int iR = A.red, iB = 2 - A.red;
    const int alpha = (C == 4) ? sp[3] : 0;
    const uint32_t q = A.cube[(uint32_t)sp[iR] | ((uint32_t)sp[1] << 8) | ((uint32_t)sp[iB] << 16)];
    dp[iR] = (uint8_t)q; dp[1] = (uint8_t)(q >> 8); dp[iB] = (uint8_t)(q >> 16);
    if (C == 4) dp[3] = (uint8_t)alpha;
}

// Rows that start on a dword boundary: a thread takes kCubeQuads groups of four pixels (C dwords in, four table reads, C dwords out each),
// the groups of a thread one wavefront-row (256 pixels x 4) apart so that every load and store instruction of a wavefront covers
// contiguous bytes; all table reads of a thread are in flight together.
constexpr int kCubeQuads = 2;
template <int C>
__global__ __launch_bounds__(kColorThreads) void color_cube_quad_kernel(CubeArgs A) {
    const uint32_t sel = cube_selector(A.red);
    const int x0 = (blockIdx.x * kCubeQuads * kColorThreads + threadIdx.x) * 4;
    const uint8_t* srow = A.src + (int64_t)blockIdx.y * A.src_stride;
    uint8_t* drow = A.dst + (int64_t)blockIdx.y * A.dst_stride;
    uint32_t w[kCubeQuads][C], q[kCubeQuads][4];
#pragma unroll
    for (int k = 0; k < kCubeQuads; ++k) {
        const int x = x0 + k * 4 * kColorThreads;
        if (x + 4 <= A.W) {
#pragma unroll
            for (int i = 0; i < C; ++i) w[k][i] = ((const uint32_t*)(srow + (int64_t)x * C))[i];
        }
    }
#pragma unroll
    for (int k = 0; k < kCubeQuads; ++k) {
        const int x = x0 + k * 4 * kColorThreads;
        if (x + 4 <= A.W) {
            uint32_t px[4];
            if (C == 4) {
#pragma unroll
                for (int p = 0; p < 4; ++p) px[p] = w[k][p];
            } else {
                px[0] = w[k][0];
                px[1] = __builtin_amdgcn_alignbit(w[k][1], w[k][0], 24);
                px[2] = __builtin_amdgcn_alignbit(w[k][2], w[k][1], 16);
                px[3] = w[k][2] >> 8;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) q[k][p] = A.cube[__builtin_amdgcn_perm(0u, px[p], sel)];
        }
    }
#pragma unroll
    for (int k = 0; k < kCubeQuads; ++k) {
        const int x = x0 + k * 4 * kColorThreads;
        if (x >= A.W) continue;
        if (x + 4 <= A.W) {
            uint32_t o[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) o[p] = __builtin_amdgcn_perm(0u, q[k][p], sel);
            uint32_t* d32 = (uint32_t*)(drow + (int64_t)x * C);
            if (C == 4) {
#pragma unroll
                for (int p = 0; p < 4; ++p) d32[p] = o[p] | (w[k][p] & 0xff000000u);
            } else {
                d32[0] = o[0] | (o[1] << 24);
                d32[1] = (o[1] >> 8) | (o[2] << 16);
                d32[2] = (o[2] >> 16) | (o[3] << 8);
            }
        } else {                                   // the last, partial group of a row
            const int iR = A.red, iB = 2 - A.red;
            for (int p = 0; x + p < A.W; ++p) {
                const uint8_t* s1 = srow + (int64_t)(x + p) * C;
                uint8_t* d1 = drow + (int64_t)(x + p) * C;
                const int alpha = (C == 4) ? s1[3] : 0;
                const uint32_t v = A.cube[(uint32_t)s1[iR] | ((uint32_t)s1[1] << 8) | ((uint32_t)s1[iB] << 16)];
                d1[iR] = (uint8_t)v; d1[1] = (uint8_t)(v >> 8); d1[iB] = (uint8_t)(v >> 16);
                if (C == 4) d1[3] = (uint8_t)alpha;
            }
        }
    }
}

// Plan creation: the red stage of DF:672-675 for the four corners of every (blue cell, green cell) at every red level.  The upper
// neighbours are min(i + 1, n - 1) as in cell_of, so the last cell of an axis repeats its own node (its weight is then 0).
__global__ void color_rtab_kernel(const float* lut /* [b][g][r][3] */, const float* pos_r /* 256 */, CellEntry* rtab, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n * 256) return;
    const int level = idx & 255, cell = idx >> 8;         // cell = b0*n + g0
    const int b0 = cell / n, g0 = cell - b0 * n;
    const int b1 = min(b0 + 1, n - 1), g1 = min(g0 + 1, n - 1);
    const Cell r = cell_of(pos_r[level], n - 1);
    const int rows[4] = {b0 * n + g0, b0 * n + g1, b1 * n + g0, b1 * n + g1};    // c00, c10, c01, c11
    CellEntry o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* lo = lut + ((size_t)rows[k] * n + r.i0) * 3;
        const float* hi = lut + ((size_t)rows[k] * n + r.i1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) o.v[3 * k + c] = lerp_ref(lo[c], hi[c], r.f);
    }
    o.v[12] = o.v[13] = o.v[14] = o.v[15] = 0.0f;
    rtab[idx] = o;
}

// ---- 16-bit images (DF:603-618 handle uint16 like uint8, with 65535 levels) ------------------------------------------------
// With 65536 input levels per channel nothing is tabulated on the input side: float01 conversion, domain mapping, cell
// lookup and all three interpolation stages are evaluated per pixel with the reference's own float32 operations (IEEE
// division: the library is built with correctly rounded division).  Output side: `passthrough` is rint(clip(x) * 65535)
// in-kernel; the sRGB re-encode goes through NumPy's implementation-defined float32 power, so it arrives as sorted
// thresholds like the 8-bit path -- but in PIECES (the composite is only piecewise monotone at 16-bit resolution: Rec.709
// knee, sRGB toe): piece p covers clip(x) in [start[p], start[p+1]), level = base[p] + #(thresholds of p <= clip(x)).
struct Color16Args {
    const uint16_t* src;
    uint16_t* dst;
    const float* lut;        // [b][g][r][3]
    const float* thr;        // concatenated per-piece thresholds (sorted within a piece)
    const void* bins;        // kBins16 entries of Bin16 (color16_build_bins)
    float dmin[3], span[3];
    float start[4];          // piece lower bounds (start[0] unused)
    int32_t base[4], off[5];
    int32_t n_pieces;        // 0: passthrough
    int32_t H, W, n, red;
    int64_t src_stride, dst_stride;   // bytes
};

constexpr int kBins16 = 65536;
struct __attribute__((aligned(16))) Bin16 {   // answers for x = b / 65536
    uint32_t kp;                               // piece that holds x << 28 | thresholds of that piece <= x (absolute index)
    float t[3];                                // the next three thresholds of the piece (+inf behind its end)
};
// The three output levels of a pixel.  A value's bin entry (one 16-byte read) counts the thresholds up to the bin's lower edge (<= the
// value: the scaling by 2^16 and the truncation are exact) and brings the next three along -- enough wherever the encode curve rises by
// less than three levels per 2^-16; past them the thresholds are counted one by one.  A bin that straddles a piece boundary (at most
// three do) takes the binary search instead.
__device__ __forceinline__ void levels16_of(const Color16Args& A, const float (&x)[3], int (&q)[3]) {
    float xc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xc[c] = fminf(fmaxf(x[c], 0.0f), 1.0f);
    if (A.n_pieces == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = (int)__builtin_rintf(xc[c] * 65535.0f);
        return;
    }
    const Bin16* bins = reinterpret_cast<const Bin16*>(A.bins);
    Bin16 e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) e[c] = bins[min((int)(xc[c] * (float)kBins16), kBins16 - 1)];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int p = 0;
        if (A.n_pieces > 1 && xc[c] >= A.start[1]) p = 1;
        if (A.n_pieces > 2 && xc[c] >= A.start[2]) p = 2;
        if (A.n_pieces > 3 && xc[c] >= A.start[3]) p = 3;
        const int first = A.off[p], end = A.off[p + 1];
        int k;
        if ((int)(e[c].kp >> 28) == p) {
            k = (int)(e[c].kp & 0x0fffffffu);
            const int n3 = (e[c].t[0] <= xc[c] ? 1 : 0) + (e[c].t[1] <= xc[c] ? 1 : 0) + (e[c].t[2] <= xc[c] ? 1 : 0);   // sorted
            k += n3;
            if (n3 == 3)
                while (k < end && A.thr[k] <= xc[c]) ++k;
        } else {
            const float* t = A.thr + first;
            int lo = 0, hi = end - first;                  // count of thresholds <= xc (upper bound)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (t[mid] <= xc[c]) lo = mid + 1; else hi = mid;
            }
            k = first + lo;
        }
        q[c] = A.base[p] + (k - first);
    }
}

// one pixel: float01 conversion, domain mapping, cell lookup, the three interpolation stages in the reference's order, output levels
__device__ __forceinline__ void color16_px(const Color16Args& A, const int (&v)[3], int (&q)[3]) {
    const int n = A.n, nmax = n - 1;
    Cell c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float f01 = (float)v[k] / 65535.0f;                                        // DF:609-610
        const float coord = fminf(fmaxf((f01 - A.dmin[k]) / A.span[k], 0.0f), 1.0f);     // DF:647
        c[k] = cell_of(coord * (float)nmax, nmax);                                       // DF:648-653
    }
    // the eight corners as 12-byte nodes, all requested before the first blend
    const F3* T = reinterpret_cast<const F3*>(A.lut);
    auto at = [&](int bi, int gi, int ri) { return T[((uint32_t)bi * (uint32_t)n + (uint32_t)gi) * (uint32_t)n + (uint32_t)ri]; };
    const F3 n000 = at(c[2].i0, c[1].i0, c[0].i0), n001 = at(c[2].i0, c[1].i0, c[0].i1);
    const F3 n010 = at(c[2].i0, c[1].i1, c[0].i0), n011 = at(c[2].i0, c[1].i1, c[0].i1);
    const F3 n100 = at(c[2].i1, c[1].i0, c[0].i0), n101 = at(c[2].i1, c[1].i0, c[0].i1);
    const F3 n110 = at(c[2].i1, c[1].i1, c[0].i0), n111 = at(c[2].i1, c[1].i1, c[0].i1);
    auto stage3 = [&](float v000, float v001, float v010, float v011, float v100, float v101, float v110, float v111) {
        const float c00 = lerp_ref(v000, v001, c[0].f), c10 = lerp_ref(v010, v011, c[0].f);                         // DF:672-675
        const float c01 = lerp_ref(v100, v101, c[0].f), c11 = lerp_ref(v110, v111, c[0].f);
        return lerp_ref(lerp_ref(c00, c10, c[1].f), lerp_ref(c01, c11, c[1].f), c[2].f);                            // DF:676-679
    };
    const float xs[3] = {stage3(n000.x, n001.x, n010.x, n011.x, n100.x, n101.x, n110.x, n111.x),
                         stage3(n000.y, n001.y, n010.y, n011.y, n100.y, n101.y, n110.y, n111.y),
                         stage3(n000.z, n001.z, n010.z, n011.z, n100.z, n101.z, n110.z, n111.z)};
    levels16_of(A, xs, q);
}

// Any alignment: one thread per pixel, 16-bit loads and stores.
template <int C>
__global__ __launch_bounds__(kColorThreads) void color_lut_u16_kernel(Color16Args A) {
    const int x = blockIdx.x * kColorThreads + threadIdx.x;
    if (x >= A.W) return;
    const uint16_t* sp = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(A.src) + (int64_t)blockIdx.y * A.src_stride) + (int64_t)x * C;
    uint16_t* dp = reinterpret_cast<uint16_t*>(reinterpret_cast<uint8_t*>(A.dst) + (int64_t)blockIdx.y * A.dst_stride) + (int64_t)x * C;
    const int iR = A.red, iB = 2 - A.red;
    const int v[3] = {sp[iR], sp[1], sp[iB]};
    const int alpha = (C == 4) ? sp[3] : 0;
    int q[3];
    color16_px(A, v, q);
    dp[iR] = (uint16_t)q[0]; dp[1] = (uint16_t)q[1]; dp[iB] = (uint16_t)q[2];
    if (C == 4) dp[3] = (uint16_t)alpha;
}

// Rows that start on a dword boundary: one thread per two pixels = C dwords in, C dwords out (the texture path is what bounds this
// kernel -- 16-bit accesses cost an instruction each, like dwords)
template <int C>
__global__ __launch_bounds__(kColorThreads) void color_lut_u16_pair_kernel(Color16Args A) {
    const int x = (blockIdx.x * kColorThreads + threadIdx.x) * 2;
    if (x >= A.W) return;
    const uint8_t* srow = reinterpret_cast<const uint8_t*>(A.src) + (int64_t)blockIdx.y * A.src_stride;
    uint8_t* drow = reinterpret_cast<uint8_t*>(A.dst) + (int64_t)blockIdx.y * A.dst_stride;
    const int iR = A.red, iB = 2 - A.red;
    if (x + 2 > A.W) {                                  // the last pixel of an odd-width row
        const uint16_t* sp = reinterpret_cast<const uint16_t*>(srow) + (int64_t)x * C;
        uint16_t* dp = reinterpret_cast<uint16_t*>(drow) + (int64_t)x * C;
        const int v[3] = {sp[iR], sp[1], sp[iB]};
        const int alpha = (C == 4) ? sp[3] : 0;
        int q[3];
        color16_px(A, v, q);
        dp[iR] = (uint16_t)q[0]; dp[1] = (uint16_t)q[1]; dp[iB] = (uint16_t)q[2];
        if (C == 4) dp[3] = (uint16_t)alpha;
        return;
    }
    uint32_t w[C];
#pragma unroll
    for (int i = 0; i < C; ++i) w[i] = reinterpret_cast<const uint32_t*>(srow + (int64_t)x * C * 2)[i];
    auto sample = [&](int i) { return (int)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu); };   // 16-bit sample i of the pair
    uint32_t o[C];
#pragma unroll
    for (int i = 0; i < C; ++i) o[i] = 0u;
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        const int m[3] = {sample(px * C), sample(px * C + 1), sample(px * C + 2)};          // memory order
        const int v[3] = {iR == 0 ? m[0] : m[2], m[1], iR == 0 ? m[2] : m[0]};
        int q[3];
        color16_px(A, v, q);
        const int out3[3] = {iR == 0 ? q[0] : q[2], q[1], iR == 0 ? q[2] : q[0]};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int i = px * C + ch;
            o[i >> 1] |= (uint32_t)out3[ch] << (16 * (i & 1));
        }
        if (C == 4) o[(px * C + 3) >> 1] |= w[(px * C + 3) >> 1] & 0xffff0000u;             // alpha: the high half of dwords 1 and 3
    }
#pragma unroll
    for (int i = 0; i < C; ++i) reinterpret_cast<uint32_t*>(drow + (int64_t)x * C * 2)[i] = o[i];
}

template <int C, int FIX>
void launch_variant(const ColorArgs& A, bool aligned, hipStream_t s) {
    if (aligned) {
        const long total = (long)((A.W + 4 * kColorThreads - 1) / (4 * kColorThreads)) * A.H;
        dim3 grid((unsigned)(total < kColorBlocks ? total : kColorBlocks));
        hipLaunchKernelGGL((color_lut_quad_kernel<C, FIX>), grid, dim3(kColorThreads), 0, s, A);
    } else {
        dim3 grid((unsigned)((A.W + kColorThreads - 1) / kColorThreads), (unsigned)A.H);
        hipLaunchKernelGGL((color_lut_bytes_kernel<C, FIX>), grid, dim3(kColorThreads), 0, s, A);
    }
}

}  // namespace

size_t color_rtab_bytes(int lut_size) { return (size_t)lut_size * lut_size * 256 * sizeof(CellEntry); }
size_t color_tables_floats() { return 1024 + kBins / 4; }

// Host: derive the bin table from the thresholds; returns the number of in-bin fix-up compares needed (1 or 2),
// or 0 when some bin holds more than two thresholds (binary search variant).
int color_build_bins(const float* thr /* 256, [0] unused */, uint8_t* bins /* kBins */) {
    int worst = 0;
    for (int i = 0; i < kBins; ++i) {
        const float lo = (float)i / (float)kBins;                        // exact
        const float hi = (float)(i + 1) / (float)kBins;
        int below = 0, inside = 0;
        for (int k = 1; k < 256; ++k) {
            if (thr[k] <= lo) ++below;
            else if (i == kBins - 1 ? thr[k] <= 1.0f : thr[k] < hi) ++inside;
        }
        bins[i] = (uint8_t)below;
        if (inside > worst) worst = inside;
    }
    return worst <= 1 ? 1 : (worst == 2 ? 2 : 0);
}

size_t color16_bins_bytes() { return (size_t)kBins16 * sizeof(Bin16); }

// Host: bin b answers for x = b / 65536 -- the piece that holds x, how many of that piece's thresholds are <= x (as an absolute index
// into the concatenated threshold array; plan16 creation bounds it by 2^20) and the next three thresholds of the piece.
void color16_build_bins(int n_pieces, const float* start, const int32_t* off, const float* thr, void* bins_out) {
    Bin16* bins = static_cast<Bin16*>(bins_out);
    for (int b = 0; b < kBins16; ++b) {
        const float edge = (float)b / (float)kBins16;                    // exact
        int p = 0;
        for (int q = 1; q < n_pieces; ++q)
            if (edge >= start[q]) p = q;
        const float* lo = thr + off[p];
        const float* const end = thr + off[p + 1];
        const float* hi = end;
        while (lo < hi) {                                                // upper bound: first threshold > edge
            const float* mid = lo + (hi - lo) / 2;
            if (*mid <= edge) lo = mid + 1; else hi = mid;
        }
        bins[b].kp = ((uint32_t)p << 28) | (uint32_t)(lo - thr);
        for (int j = 0; j < 3; ++j) bins[b].t[j] = lo + j < end ? lo[j] : __builtin_inff();
    }
}

hipError_t build_color_rtab(const float* d_lut, const float* d_pos_r, void* d_rtab, int lut_size, hipStream_t s) {
    const int total = lut_size * lut_size * 256;
    hipLaunchKernelGGL(color_rtab_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_lut, d_pos_r, (CellEntry*)d_rtab, lut_size);
    return hipGetLastError();
}

size_t color_cube_bytes() { return ((size_t)1 << 24) * sizeof(uint32_t); }

hipError_t build_color_cube(const ColorLaunch& L, void* d_cube, hipStream_t s) {
    ColorArgs A;
    A.src = nullptr; A.dst = nullptr; A.rtab = (const CellEntry*)L.rtab; A.tables = L.tables;
    A.H = 0; A.W = 0; A.n = L.lut_size; A.red = 0; A.src_stride = 0; A.dst_stride = 0;
    const dim3 grid(kColorBlocks), block(kColorThreads);
    if (L.fixups == 1) hipLaunchKernelGGL((color_cube_kernel<1>), grid, block, 0, s, A, (uint32_t*)d_cube);
    else if (L.fixups == 2) hipLaunchKernelGGL((color_cube_kernel<2>), grid, block, 0, s, A, (uint32_t*)d_cube);
    else hipLaunchKernelGGL((color_cube_kernel<0>), grid, block, 0, s, A, (uint32_t*)d_cube);
    return hipGetLastError();
}

static hipError_t launch_color_cube(const ColorLaunch& L, int C, hipStream_t s) {
    CubeArgs A;
    A.src = L.src; A.dst = L.dst; A.cube = (const uint32_t*)L.cube; A.H = L.H; A.W = L.W; A.red = L.red_index;
    A.src_stride = L.src_stride; A.dst_stride = L.dst_stride;
    const bool aligned = (((uintptr_t)L.src | (uintptr_t)L.dst | (uint64_t)L.src_stride | (uint64_t)L.dst_stride) & 3u) == 0;
    if (aligned) {
        const int per_block = 4 * kCubeQuads * kColorThreads;
        const dim3 grid((unsigned)((L.W + per_block - 1) / per_block), (unsigned)L.H);
        if (C == 3) hipLaunchKernelGGL((color_cube_quad_kernel<3>), grid, dim3(kColorThreads), 0, s, A);
        else hipLaunchKernelGGL((color_cube_quad_kernel<4>), grid, dim3(kColorThreads), 0, s, A);
    } else {
        const dim3 grid((unsigned)((L.W + kColorThreads - 1) / kColorThreads), (unsigned)L.H);
        if (C == 3) hipLaunchKernelGGL((color_cube_bytes_kernel<3>), grid, dim3(kColorThreads), 0, s, A);
        else hipLaunchKernelGGL((color_cube_bytes_kernel<4>), grid, dim3(kColorThreads), 0, s, A);
    }
    return hipGetLastError();
}

hipError_t launch_color16(const Color16Launch& L, int C, hipStream_t s) {
    Color16Args A;
    A.src = (const uint16_t*)L.src; A.dst = (uint16_t*)L.dst; A.lut = L.lut; A.thr = L.thr; A.bins = L.bins;
    for (int k = 0; k < 3; ++k) { A.dmin[k] = L.dmin[k]; A.span[k] = L.span[k]; }
    for (int k = 0; k < 4; ++k) { A.start[k] = L.start[k]; A.base[k] = L.base[k]; }
    for (int k = 0; k < 5; ++k) A.off[k] = L.off[k];
    A.n_pieces = L.n_pieces; A.H = L.H; A.W = L.W; A.n = L.lut_size; A.red = L.red_index;
    A.src_stride = L.src_stride; A.dst_stride = L.dst_stride;
    const bool aligned = (((uintptr_t)L.src | (uintptr_t)L.dst | (uint64_t)L.src_stride | (uint64_t)L.dst_stride) & 3u) == 0;
    if (aligned) {
        dim3 grid((unsigned)(((L.W + 1) / 2 + kColorThreads - 1) / kColorThreads), (unsigned)L.H);
        if (C == 3) hipLaunchKernelGGL((color_lut_u16_pair_kernel<3>), grid, dim3(kColorThreads), 0, s, A);
        else hipLaunchKernelGGL((color_lut_u16_pair_kernel<4>), grid, dim3(kColorThreads), 0, s, A);
    } else {
        dim3 grid((unsigned)((L.W + kColorThreads - 1) / kColorThreads), (unsigned)L.H);
        if (C == 3) hipLaunchKernelGGL((color_lut_u16_kernel<3>), grid, dim3(kColorThreads), 0, s, A);
        else hipLaunchKernelGGL((color_lut_u16_kernel<4>), grid, dim3(kColorThreads), 0, s, A);
    }
    return hipGetLastError();
}

hipError_t launch_color(const ColorLaunch& L, int C, hipStream_t s) {
    if (L.cube) return launch_color_cube(L, C, s);
    ColorArgs A;
    A.src = L.src; A.dst = L.dst; A.rtab = (const CellEntry*)L.rtab; A.tables = L.tables;
    A.H = L.H; A.W = L.W; A.n = L.lut_size; A.red = L.red_index;
    A.src_stride = L.src_stride; A.dst_stride = L.dst_stride;
    const bool aligned = (((uintptr_t)L.src | (uintptr_t)L.dst | (uint64_t)L.src_stride | (uint64_t)L.dst_stride) & 3u) == 0;
    if (C == 3) {
        if (L.fixups == 1) launch_variant<3, 1>(A, aligned, s);
        else if (L.fixups == 2) launch_variant<3, 2>(A, aligned, s);
        else launch_variant<3, 0>(A, aligned, s);
    } else {
        if (L.fixups == 1) launch_variant<4, 1>(A, aligned, s);
        else if (L.fixups == 2) launch_variant<4, 2>(A, aligned, s);
        else launch_variant<4, 0>(A, aligned, s);
    }
    return hipGetLastError();
}

}  // namespace gs360
