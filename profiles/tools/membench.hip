// membench: what the MI355X memory system delivers for the headline kernel's traffic SHAPE, without any of its arithmetic.
// The source-major equirect kernel moves 58.8 MB of reads + 11.5 MB of stores per 8K frame (5.1 : 1), the reads as row pieces of ~784 bytes
// at a 23,040-byte stride, HBM-cold (16 distinct frames = 1.4 GB per launch).  Modes (all over the same 1.4 GB, each launch touches it once):
//   read     contiguous 1 KiB per wavefront and load instruction, registers            (the read-only streaming rate)
//   copy     the same + one 16-byte store per load                                      (the 6.29 TB/s "streaming copy" of DESIGN.md)
//   mix      five loads, one store                                                      (the kernel's read : write ratio, contiguous)
//   rows     784-byte row pieces at the frame's stride, boxes of 32 rows, registers     (the kernel's read shape)
//   rowsmix  rows + one store per five loads                                            (shape + ratio)
//   dma      `rows` through global_load_lds (LDS copies, nothing reads the LDS)         (the kernel's read instruction)
//   dmamix   dma + stores
//   rows5 / dma5  the kernel's own store shape: 768 bytes per five rows, stores 5 dwordx3 | 6 dwordx3 non-temporal | 7 dwordx4 | 8 dwordx4 non-temporal
//   write    contiguous stores only
// usage: membench <mode> [waves_per_wg=4] [wgs_per_cu=2..8] [unroll=8] [reps=20] [stores: 1 plain, 2 non-temporal, 3 into a 16 MB window, 4 a box's stores at its end]
// build: hipcc --offload-arch=gfx950 -O3 -o membench membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void global_void_t;
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int kFrames = 16, kW = 7680, kH = 3840, kRow = kW * 3;            // 23,040 bytes per row
constexpr size_t kFrame = (size_t)kRow * kH;                                // 88,473,600
constexpr int kPiece = 784, kLanes = kPiece / 16, kBoxRows = 32;            // 49 lanes x 16 bytes
constexpr int kStoreLanes = 39;                                             // one 16-byte store per lane < 39 and four rows: 624 : 3,136 bytes = 1 : 5.03
constexpr int kBoxesX = kRow / kPiece, kBoxesY = kH / kBoxRows;             // 29 x 120 boxes per frame (22,736 of 23,040 bytes per row)

// contiguous: a wavefront's k-th piece is 1 KiB at (k * total_waves + wave) KiB
template <int UNROLL, int STORE_EVERY>   // STORE_EVERY: 0 = none, 1 = every load, 5 = one per five loads
__global__ void __launch_bounds__(256) stream_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, uint32_t* sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t o = i;
    uint32_t acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (STORE_EVERY == 1) dst[i + u * stride] = v[u];
            else if (STORE_EVERY > 1 && u % (STORE_EVERY > 1 ? STORE_EVERY : 1) == 0) { uint4 w = v[u]; for (int t = 1; t < STORE_EVERY && u + t < UNROLL; ++t) { w.x ^= v[u + t].x; w.y ^= v[u + t].y; w.z ^= v[u + t].z; w.w ^= v[u + t].w; } dst[o] = w; o += stride; }
            else if (STORE_EVERY == 0) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (STORE_EVERY == 0 && acc == 0x9e3779b9u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) write_kernel(uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const uint4 w = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = w;
}

// boxes: a wavefront takes whole boxes (32 rows x 784 bytes), UNROLL rows in flight
template <int UNROLL, int SK, bool DMA>   // SK: 0 no stores, 1 plain, 2 non-temporal, 3 plain into a 16 MB window (never leaves the caches), 4 a box's stores at its end
__global__ void __launch_bounds__(256) rows_kernel(const uint8_t* __restrict__ src, uint4* __restrict__ dst, int n_boxes, uint32_t* sink) {
    extern __shared__ uint8_t s_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int waves = gridDim.x * nwv;
    // XCD-contiguous: block b runs on XCD b % 8; give an XCD a contiguous run of boxes (neighbouring boxes share DRAM pages)
    const int per_xcd = (gridDim.x + 7) / 8;
    const int wave0 = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * nwv + wv;
    uint8_t* const my = s_lds + wv * (UNROLL * 1024);
    constexpr bool STORES = SK != 0;
    uint32_t acc = 0;
    size_t o = (size_t)wave0 * 64 + lane;
    auto put = [&](const uint4 w) {
        if constexpr (SK == 2) { __builtin_nontemporal_store(w.x, &dst[o].x); __builtin_nontemporal_store(w.y, &dst[o].y); __builtin_nontemporal_store(w.z, &dst[o].z); __builtin_nontemporal_store(w.w, &dst[o].w); }
        else if constexpr (SK == 3) dst[o & ((1u << 20) - 1)] = w;
        else dst[o] = w;
        o += (size_t)waves * 64;
    };
    for (int b = wave0; b < n_boxes; b += waves) {
        uint4 held[kBoxRows / 4];
        const int f = b / (kBoxesX * kBoxesY), r = b - f * (kBoxesX * kBoxesY), by = r / kBoxesX, bx = r - by * kBoxesX;
        const uint8_t* p = src + (size_t)f * kFrame + (size_t)(by * kBoxRows) * kRow + bx * kPiece + lane * 16;
        if (lane < kLanes) {
            for (int row = 0; row < kBoxRows; row += UNROLL) {
                if constexpr (DMA) {
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u)
                        __builtin_amdgcn_global_load_lds((global_void_t*)(p + (size_t)(row + u) * kRow), (lds_void_t*)(my + u * 1024), 16, 0, 0);
                    if (STORES && lane < kStoreLanes) {
                        uint4 w = {(uint32_t)row, (uint32_t)b, 0u, 0u};
#pragma unroll
                        for (int u = 0; u < UNROLL; u += 4) { if constexpr (SK == 4) held[0] = w; else put(w); }
                    }
                } else {
                    uint4 v[UNROLL];
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) v[u] = *(const uint4*)(p + (size_t)(row + u) * kRow);
#pragma unroll
                    for (int u = 0; u < UNROLL; u += 4) {
                        uint4 w = v[u];
#pragma unroll
                        for (int t = 1; t < 4; ++t) { w.x ^= v[u + t].x; w.y ^= v[u + t].y; w.z ^= v[u + t].z; w.w ^= v[u + t].w; }
                        if constexpr (SK == 4) held[(row + u) / 4 % (kBoxRows / 4)] = w;
                        else if (STORES) { if (lane < kStoreLanes) put(w); }
                        else acc ^= w.x ^ w.y ^ w.z ^ w.w;
                    }
                }
            }
            if constexpr (SK == 4) {
                if (lane < kStoreLanes) {
#pragma unroll
                    for (int k = 0; k < kBoxRows / 4; ++k) put(DMA ? held[0] : held[k]);
                }
            }
        }
    }
    if constexpr (DMA) { __builtin_amdgcn_s_waitcnt(0); }
    if (!STORES && acc == 0x9e3779b9u) sink[0] = acc;
}

// the kernel's own store shape: per five box rows (3,920 bytes read) ONE store instruction of 768 contiguous bytes -- as 64 lanes x 12 bytes
// (dwordx3 at a 12-byte stride: what the consumers issue) or as 48 lanes x 16 aligned bytes -- plain or non-temporal.  30 rows per box.
template <int SK, bool DMA>   // SK: 5 dwordx3 plain, 6 dwordx3 non-temporal, 7 dwordx4 plain, 8 dwordx4 non-temporal
__global__ void __launch_bounds__(256) rows5_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n_boxes, uint32_t* sink) {
    extern __shared__ uint8_t s_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int waves = gridDim.x * nwv;
    const int per_xcd = (gridDim.x + 7) / 8;
    const int wave0 = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * nwv + wv;
    uint8_t* const my = s_lds + wv * (5 * 1024);
    size_t o = (size_t)wave0 * 768;
    for (int b = wave0; b < n_boxes; b += waves) {
        const int f = b / (kBoxesX * kBoxesY), r = b - f * (kBoxesX * kBoxesY), by = r / kBoxesX, bx = r - by * kBoxesX;
        const uint8_t* p = src + (size_t)f * kFrame + (size_t)(by * kBoxRows) * kRow + bx * kPiece + lane * 16;
        for (int row = 0; row < 30; row += 5) {
            uint4 w = {(uint32_t)row, (uint32_t)b, 0u, 0u};
            if (lane < kLanes) {
                if constexpr (DMA) {
#pragma unroll
                    for (int u = 0; u < 5; ++u)
                        __builtin_amdgcn_global_load_lds((global_void_t*)(p + (size_t)(row + u) * kRow), (lds_void_t*)(my + u * 1024), 16, 0, 0);
                } else {
                    uint4 v[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) v[u] = *(const uint4*)(p + (size_t)(row + u) * kRow);
#pragma unroll
                    for (int u = 0; u < 5; ++u) { w.x ^= v[u].x; w.y ^= v[u].y; w.z ^= v[u].z; w.w ^= v[u].w; }
                }
            }
            if constexpr (SK == 5 || SK == 6) {
                uint32_t* q = (uint32_t*)(dst + o + lane * 12);
                if constexpr (SK == 6) { __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q + 1); __builtin_nontemporal_store(w.z, q + 2); }
                else { q[0] = w.x; q[1] = w.y; q[2] = w.z; }
            } else if (lane < 48) {
                uint32_t* q = (uint32_t*)(dst + o + lane * 16);
                if constexpr (SK == 8) { __builtin_nontemporal_store(w.x, q); __builtin_nontemporal_store(w.y, q + 1); __builtin_nontemporal_store(w.z, q + 2); __builtin_nontemporal_store(w.w, q + 3); }
                else *(uint4*)q = w;
            }
            o += (size_t)waves * 768;
        }
    }
    if constexpr (DMA) { __builtin_amdgcn_s_waitcnt(0); }
    if (n_boxes < 0) sink[0] = 1;
}

// One probe: `mode` as above on device `dev`; out[0..3] = median TB/s (reads + writes), median reads, median writes, best TB/s.
// Returns 0, or a negative number with nothing left allocated.  Also the entry point of libgs360probe.so (bench.py calls it in-process,
// after its timed region, so that `roofline.memsys` carries this box's own reading; hipcc ... -shared -fPIC builds it from this file).
extern "C" int gs360_membench(const char* mode, int dev, int wpw, int per_cu, int unroll, int reps, int sk, double* out) {
    uint8_t* src = nullptr; uint4* dst = nullptr; uint32_t* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
#define TRY(x) do { if (rc == 0 && (x) != hipSuccess) rc = -__LINE__; } while (0)
    if (wpw < 1 || wpw > 4 || per_cu < 1 || per_cu > 16 || reps < 1 || reps > 1000) return -1;
    TRY(hipSetDevice(dev));
    hipDeviceProp_t prop;
    TRY(hipGetDeviceProperties(&prop, dev));
    if (rc) return rc;
    const int cus = prop.multiProcessorCount;
    const size_t bytes = kFrame * kFrames, n16 = bytes / 16;
    TRY(hipMalloc(&src, bytes + 4096)); TRY(hipMalloc(&dst, bytes + 4096)); TRY(hipMalloc(&sink, 64));
    TRY(hipMemset(src, 1, bytes)); TRY(hipMemset(dst, 0, bytes));
    const int blocks = cus * per_cu, threads = wpw * 64;
    const int n_boxes = kFrames * kBoxesX * kBoxesY;
    double rd = 0, wr = 0;
    bool known = true;
    auto launch = [&]() {
#define STREAM(U, S) hipLaunchKernelGGL((stream_kernel<U, S>), dim3(blocks), dim3(threads), 0, 0, (const uint4*)src, dst, n16, sink)
#define ROWS(U, S, D) hipLaunchKernelGGL((rows_kernel<U, S, D>), dim3(blocks), dim3(threads), (D) ? wpw * U * 1024 : 0, 0, src, dst, n_boxes, sink)
#define BYU(M) do { if (unroll == 4) { M(4); } else if (unroll == 16) { M(16); } else { M(8); } } while (0)
        if (!strcmp(mode, "read")) {
#define M(U) STREAM(U, 0)
            BYU(M); rd = (double)bytes; wr = 0;
#undef M
        } else if (!strcmp(mode, "copy")) {
#define M(U) STREAM(U, 1)
            BYU(M); rd = wr = (double)bytes;
#undef M
        } else if (!strcmp(mode, "mix")) {
            if (unroll == 5) STREAM(5, 5); else if (unroll == 20) STREAM(20, 5); else STREAM(10, 5);
            rd = (double)bytes; wr = rd / 5;
        } else if (!strcmp(mode, "rows")) {
#define M(U) ROWS(U, 0, false)
            BYU(M); rd = (double)n_boxes * kBoxRows * kPiece; wr = 0;
#undef M
        } else if (!strcmp(mode, "rowsmix")) {
#define M(U) do { if (sk == 2) ROWS(U, 2, false); else if (sk == 3) ROWS(U, 3, false); else if (sk == 4) ROWS(U, 4, false); else ROWS(U, 1, false); } while (0)
            BYU(M); rd = (double)n_boxes * kBoxRows * kPiece;
#undef M
            wr = (double)n_boxes * (kBoxRows / 4) * kStoreLanes * 16;
        } else if (!strcmp(mode, "dma")) {
#define M(U) ROWS(U, 0, true)
            BYU(M); rd = (double)n_boxes * kBoxRows * kPiece; wr = 0;
#undef M
        } else if (!strcmp(mode, "dmamix")) {
#define M(U) do { if (sk == 2) ROWS(U, 2, true); else if (sk == 3) ROWS(U, 3, true); else if (sk == 4) ROWS(U, 4, true); else ROWS(U, 1, true); } while (0)
            BYU(M); rd = (double)n_boxes * kBoxRows * kPiece;
#undef M
            wr = (double)n_boxes * (kBoxRows / 4) * kStoreLanes * 16;
        } else if (!strcmp(mode, "rows5") || !strcmp(mode, "dma5")) {
            const bool dma = mode[0] == 'd';
#define R5(S) do { if (dma) hipLaunchKernelGGL((rows5_kernel<S, true>), dim3(blocks), dim3(threads), wpw * 5 * 1024, 0, src, (uint8_t*)dst, n_boxes, sink); \
                   else hipLaunchKernelGGL((rows5_kernel<S, false>), dim3(blocks), dim3(threads), 0, 0, src, (uint8_t*)dst, n_boxes, sink); } while (0)
            if (sk == 6) R5(6); else if (sk == 7) R5(7); else if (sk == 8) R5(8); else R5(5);
            rd = (double)n_boxes * 30 * kPiece; wr = (double)n_boxes * 6 * 768;
        } else if (!strcmp(mode, "write")) {
            hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(threads), 0, 0, dst, n16); rd = 0; wr = (double)bytes;
        } else known = false;
    };
    TRY(hipEventCreate(&e0)); TRY(hipEventCreate(&e1));
    std::vector<float> t((size_t)reps, 0.f);
    if (rc == 0) {
        for (int i = 0; i < 3 && known; ++i) launch();      // settle: ~1 ms per launch; the clocks ramp over ~100 ms of load
        if (!known) rc = -2;
        TRY(hipDeviceSynchronize());
        if (rc == 0) { TRY(hipEventRecord(e0)); float ms = 0; int n = 0; do { launch(); TRY(hipEventRecord(e1)); TRY(hipEventSynchronize(e1)); TRY(hipEventElapsedTime(&ms, e0, e1)); } while (rc == 0 && ms < 150.f && ++n < 100000); }
        for (int i = 0; i < reps && rc == 0; ++i) { TRY(hipEventRecord(e0)); launch(); TRY(hipEventRecord(e1)); TRY(hipEventSynchronize(e1)); TRY(hipEventElapsedTime(&t[(size_t)i], e0, e1)); }
        TRY(hipGetLastError());
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (sink) (void)hipFree(sink);
#undef TRY
    if (rc) return rc;
    std::sort(t.begin(), t.end());
    const double med = t[(size_t)reps / 2] * 1e-3, best = t[0] * 1e-3;
    out[0] = (rd + wr) / med / 1e12; out[1] = rd / med / 1e12; out[2] = wr / med / 1e12; out[3] = (rd + wr) / best / 1e12;
    out[4] = rd / 1e6; out[5] = wr / 1e6; out[6] = med * 1e3;
    return 0;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "read";
    const int wpw = argc > 2 ? atoi(argv[2]) : 4, per_cu = argc > 3 ? atoi(argv[3]) : 4, unroll = argc > 4 ? atoi(argv[4]) : 8, reps = argc > 5 ? atoi(argv[5]) : 20, sk = argc > 6 ? atoi(argv[6]) : 1;
    double o[7];
    const int rc = gs360_membench(mode, 0, wpw, per_cu, unroll, reps, sk, o);
    if (rc) { printf("%s: failed (%d)\n", mode, rc); return 1; }
    printf("%-8s waves/wg %d wgs/cu %d unroll %2d stores %d : read %7.1f MB write %7.1f MB  median %.3f ms = %.2f TB/s (reads %.2f, writes %.2f)  best %.2f TB/s\n", mode, wpw, per_cu, unroll, sk,
           o[4], o[5], o[6], o[0], o[1], o[2], o[3]);
    return 0;
}
