// gs360_color.hip -- the dual-fisheye tool's input colour stage on the GPU (include/gs360.h, "input colour stage").
//
// Reference: apply_input_color_pipeline, cli_tools/gs360_DualFisheyeDistortionCalibration.py:684-725 --
// u8 -> float01 (DF:603-613) -> .cube trilinear (DF:620-681) -> rec709_to_srgb (DF:565-600, optional) -> u8 (DF:616-...).
// The two scalar ends (level -> grid position, LUT output -> 8-bit level) arrive as host-built tables (256 x 3 floats
// and 255 thresholds, see the header); the kernel does the 8-texel interpolation in the reference's float32 operation
// order (sub, mul, add -- never fused: the library is built with -ffp-contract=off), so the result is the byte the
// NumPy pipeline produces.  A streaming kernel: HBM-bound for images with colour locality; the 8 LUT texel reads per
// pixel come from L2 (a 33^3 table is 575 KB as float4).
#include <cstring>

#include "gs360_kernels.h"

namespace gs360 {

namespace {

constexpr int kColorThreads = 256;

struct ColorArgs {
    const uint8_t* src;
    uint8_t* dst;
    const float4* lut;      // [b][g][r] -> (R,G,B,0)
    const float* tables;    // 768 level positions (R,G,B x 256) followed by 256 output thresholds
    int32_t H, W;
    int32_t n;              // LUT edge length
    int32_t red;            // memory index of the red channel (0 or 2)
    int64_t src_stride, dst_stride;
};

struct Cell {               // one channel's LUT cell: lower index, upper index, weight of the upper texel
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

__device__ __forceinline__ int level_of(const float* thr, float x) {
    // number of thresholds <= x among thr[1..255] (non-decreasing); NaN compares false everywhere -> 0
    int lv = 0;
#pragma unroll
    for (int bit = 128; bit > 0; bit >>= 1) {
        const int cand = lv | bit;
        lv = (x >= thr[cand]) ? cand : lv;
    }
    return lv;
}

struct Rgb8 { int r, g, b; };

// the whole pipeline for one pixel: three 8-bit levels in, three out
__device__ __forceinline__ Rgb8 color_px(const ColorArgs& A, const float* s_pos, const float* s_thr, int vr, int vg, int vb) {
    const int n = A.n, nmax = n - 1;
    const Cell r = cell_of(s_pos[vr], nmax);
    const Cell g = cell_of(s_pos[256 + vg], nmax);
    const Cell b = cell_of(s_pos[512 + vb], nmax);
    const int row00 = (b.i0 * n + g.i0) * n, row10 = (b.i0 * n + g.i1) * n;
    const int row01 = (b.i1 * n + g.i0) * n, row11 = (b.i1 * n + g.i1) * n;
    // all eight texel reads are issued before the arithmetic starts
    const float4 c000 = A.lut[row00 + r.i0], c100 = A.lut[row00 + r.i1];
    const float4 c010 = A.lut[row10 + r.i0], c110 = A.lut[row10 + r.i1];
    const float4 c001 = A.lut[row01 + r.i0], c101 = A.lut[row01 + r.i1];
    const float4 c011 = A.lut[row11 + r.i0], c111 = A.lut[row11 + r.i1];
    float o[3];
    int k = 0;
#define GS360_TRI(ch)                                                   \
    {                                                                   \
        const float c00 = lerp_ref(c000.ch, c100.ch, r.f);              \
        const float c10 = lerp_ref(c010.ch, c110.ch, r.f);              \
        const float c01 = lerp_ref(c001.ch, c101.ch, r.f);              \
        const float c11 = lerp_ref(c011.ch, c111.ch, r.f);              \
        const float c0 = lerp_ref(c00, c10, g.f);                       \
        const float c1 = lerp_ref(c01, c11, g.f);                       \
        o[k++] = lerp_ref(c0, c1, b.f);                                 \
    }
    GS360_TRI(x) GS360_TRI(y) GS360_TRI(z)
#undef GS360_TRI
    Rgb8 q;
    q.r = level_of(s_thr, o[0]);
    q.g = level_of(s_thr, o[1]);
    q.b = level_of(s_thr, o[2]);
    return q;
}

__device__ __forceinline__ void load_tables(const ColorArgs& A, float* s_pos, float* s_thr) {
    for (int i = threadIdx.x; i < 1024; i += kColorThreads) {
        const float v = A.tables[i];
        if (i < 768) s_pos[i] = v; else s_thr[i - 768] = v;
    }
    __syncthreads();
}

// Any alignment: one thread per pixel, byte loads and stores.
template <int C>
__global__ __launch_bounds__(kColorThreads) void color_lut_bytes_kernel(ColorArgs A) {
    __shared__ float s_pos[768];
    __shared__ float s_thr[256];
    load_tables(A, s_pos, s_thr);
    const int x = blockIdx.x * kColorThreads + threadIdx.x;
    if (x >= A.W) return;
    const uint8_t* sp = A.src + (int64_t)blockIdx.y * A.src_stride + (int64_t)x * C;
    uint8_t* dp = A.dst + (int64_t)blockIdx.y * A.dst_stride + (int64_t)x * C;
    const int iR = A.red, iB = 2 - A.red;
    const int alpha = (C == 4) ? sp[3] : 0;
    const Rgb8 q = color_px(A, s_pos, s_thr, sp[iR], sp[1], sp[iB]);
    dp[iR] = (uint8_t)q.r; dp[1] = (uint8_t)q.g; dp[iB] = (uint8_t)q.b;
    if (C == 4) dp[3] = (uint8_t)alpha;
}

// Rows that start on a dword boundary: one thread per 4 pixels = C dwords in, C dwords out, so a wavefront moves
// 768 (C=3) or 1024 (C=4) contiguous bytes per row segment.
template <int C>
__global__ __launch_bounds__(kColorThreads) void color_lut_quad_kernel(ColorArgs A) {
    __shared__ float s_pos[768];
    __shared__ float s_thr[256];
    load_tables(A, s_pos, s_thr);
    const int x = (blockIdx.x * kColorThreads + threadIdx.x) * 4;
    if (x >= A.W) return;
    const uint8_t* sp = A.src + (int64_t)blockIdx.y * A.src_stride + (int64_t)x * C;
    uint8_t* dp = A.dst + (int64_t)blockIdx.y * A.dst_stride + (int64_t)x * C;
    const int iR = A.red, iB = 2 - A.red;
    const bool bgr = A.red != 0;
    if (x + 4 <= A.W) {
        uint32_t w[C];
        const uint32_t* s32 = (const uint32_t*)sp;
#pragma unroll
        for (int i = 0; i < C; ++i) w[i] = s32[i];
        uint32_t o[C];
#pragma unroll
        for (int i = 0; i < C; ++i) o[i] = (C == 4) ? (w[i] & 0xff000000u) : 0u;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int v[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int byte = p * C + c;
                v[c] = (w[byte >> 2] >> (8 * (byte & 3))) & 0xff;
            }
            const Rgb8 q = color_px(A, s_pos, s_thr, bgr ? v[2] : v[0], v[1], bgr ? v[0] : v[2]);
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
            const Rgb8 q = color_px(A, s_pos, s_thr, s1[iR], s1[1], s1[iB]);
            d1[iR] = (uint8_t)q.r; d1[1] = (uint8_t)q.g; d1[iB] = (uint8_t)q.b;
            if (C == 4) d1[3] = (uint8_t)alpha;
        }
    }
}

}  // namespace

hipError_t launch_color(const ColorLaunch& L, int C, hipStream_t s) {
    ColorArgs A;
    A.src = L.src; A.dst = L.dst; A.lut = (const float4*)L.lut; A.tables = L.tables;
    A.H = L.H; A.W = L.W; A.n = L.lut_size; A.red = L.red_index;
    A.src_stride = L.src_stride; A.dst_stride = L.dst_stride;
    const bool aligned = (((uintptr_t)L.src | (uintptr_t)L.dst | (uint64_t)L.src_stride | (uint64_t)L.dst_stride) & 3u) == 0;
    if (aligned) {
        const int quads = (L.W + 3) / 4;
        dim3 grid((unsigned)((quads + kColorThreads - 1) / kColorThreads), (unsigned)L.H);
        if (C == 3) hipLaunchKernelGGL(color_lut_quad_kernel<3>, grid, dim3(kColorThreads), 0, s, A);
        else hipLaunchKernelGGL(color_lut_quad_kernel<4>, grid, dim3(kColorThreads), 0, s, A);
    } else {
        dim3 grid((unsigned)((L.W + kColorThreads - 1) / kColorThreads), (unsigned)L.H);
        if (C == 3) hipLaunchKernelGGL(color_lut_bytes_kernel<3>, grid, dim3(kColorThreads), 0, s, A);
        else hipLaunchKernelGGL(color_lut_bytes_kernel<4>, grid, dim3(kColorThreads), 0, s, A);
    }
    return hipGetLastError();
}

}  // namespace gs360
