// Feasibility probe for LDS-staged footprints: how fast can a CU pull short runs of 128-B lines (8 contiguous lines per
// source row, rows 23040 B apart) into LDS with global_load_lds_dwordx4, HBM-cold, vs. register-staged loads?
// build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_probe ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kStride = 23040;          // 7680 * 3
constexpr int kRowsFrame = 3840;
constexpr size_t kFrameBytes = (size_t)kStride * kRowsFrame;

struct P { const uint8_t* frames[8]; uint32_t* out; int tiles_per_frame; int tiles_x; int rows_per_pass; int lines_per_row; int passes; int mode; int row_step_x10; };

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void s_prefetch64(const uint8_t* line) {
    // one 64-byte scalar load whose data is never used: pulls the line into L2 through the scalar-cache miss path
    const uint64_t a = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)line) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)line >> 32)) << 32);
    asm volatile("s_load_dwordx16 s[80:95], %0, 0x0" :: "s"(a) : "s80","s81","s82","s83","s84","s85","s86","s87","s88","s89","s90","s91","s92","s93","s94","s95","memory");
}

template <int MODE>   // 3: scalar loads only; 4: scalar prefetch of odd rows, then vector loads of everything; 0: LDS-DMA, 1: reg-staged (global_load_dwordx4 + ds_write_b128), 2: registers only
__global__ __launch_bounds__(256) void probe(P p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int b = blockIdx.x;
    int chunk = (gridDim.x + 7) / 8;
    int t = (b & 7) * chunk + (b >> 3);
    if (t >= (int)gridDim.x) return;
    int f = t / p.tiles_per_frame, r = t - f * p.tiles_per_frame;
    int ty = r / p.tiles_x, tx = r - ty * p.tiles_x;
    const uint8_t* src = p.frames[f & 7];
    const int tid = threadIdx.x, grp = tid >> 3, sub = tid & 7;
    uint32_t acc = 0;
    const int L = p.rows_per_pass * p.lines_per_row;
    if (MODE == 3 || MODE == 4) {
        // each wave prefetches its share of the tile's lines (all passes) with scalar loads
        const int wave = tid >> 6;
        const int Lt = L * p.passes;
        for (int sl = wave; sl < Lt; sl += 4) {
            int pass = sl / L, s2 = sl - pass * L;
            int k = s2 / p.lines_per_row, j = s2 - k * p.lines_per_row;
            if (MODE == 4 && (k & 1) == 0) continue;
            int row = (ty * p.passes + pass) * p.rows_per_pass + k;
            int line = tx * p.lines_per_row + j;
            s_prefetch64(src + (size_t)row * kStride + (size_t)line * 128);
        }
        if (MODE == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
    }
    for (int pass = 0; pass < p.passes; ++pass) {
        // rows of this pass: pairs of adjacent rows every row_step
        const int row_base = (ty * p.passes + pass) * p.rows_per_pass;
        for (int it = 0; it * 32 < L; ++it) {
            int slot = it * 32 + grp;
            if (slot < L) {
                int k = slot / p.lines_per_row, j = slot - k * p.lines_per_row;
                int row = row_base + k;
                int line = tx * p.lines_per_row + j;
                const uint8_t* g = src + (size_t)row * kStride + (size_t)line * 128 + sub * 16;
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                     (lds_void*)(lds + (size_t)it * 4096 + (tid >> 6) * 1024), 16, 0, 0);
                } else if (MODE == 1) {
                    uint4 v = *reinterpret_cast<const uint4*>(g);
                    *reinterpret_cast<uint4*>(lds + (size_t)slot * 128 + sub * 16) = v;
                } else {
                    uint4 v = *reinterpret_cast<const uint4*>(g);
                    acc += v.x ^ v.y ^ v.z ^ v.w;
                }
            }
        }
        if (MODE != 2 && MODE != 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // gather phase stand-in: 24 dword reads per lane at pseudo-random addresses inside the staged lines
            uint32_t h = tid * 2654435761u + pass * 97u;
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                h = h * 1664525u + 1013904223u;
                uint32_t a = ((h >> 8) % (uint32_t)(L * 128 - 16)) & ~3u;
                acc += *reinterpret_cast<const uint32_t*>(lds + a);
            }
            __syncthreads();
        }
    }
    if (acc == 0x12345678u) p.out[t] = acc;
}

int main(int argc, char** argv) {
    int rows = argc > 1 ? atoi(argv[1]) : 32, lpr = argc > 2 ? atoi(argv[2]) : 8, passes = argc > 3 ? atoi(argv[3]) : 2;
    int lds_kb = argc > 4 ? atoi(argv[4]) : 0;
    P p;
    std::vector<void*> bufs;
    for (int i = 0; i < 8; ++i) { void* d; CK(hipMalloc(&d, kFrameBytes + 4096)); CK(hipMemset(d, i + 1, kFrameBytes)); p.frames[i] = (const uint8_t*)d; }
    CK(hipMalloc((void**)&p.out, 1 << 22));
    p.tiles_x = 180 / lpr; p.tiles_per_frame = p.tiles_x * (kRowsFrame / (rows * passes)); p.rows_per_pass = rows; p.lines_per_row = lpr; p.passes = passes; p.row_step_x10 = 46;
    const int L = rows * lpr;
    size_t lds = (size_t)((L + 31) / 32) * 4096 + 64;
    if (lds_kb * 1024 > (int)lds) lds = (size_t)lds_kb * 1024;
    int grid = p.tiles_per_frame * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 5; ++mode) {
        auto launch = [&]() {
            if (mode == 0) { CK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), lds, 0, p); }
            if (mode == 1) { CK(hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), lds, 0, p); }
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, p);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, 0, p);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, p);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int N = 20;
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
        double lines = (double)grid * passes * L;
        printf("mode %d rows %d lpr %d passes %d lds %zu B: %.1f us/launch, %.2f M lines, %.2f TB/s of line bytes\n", mode, rows, lpr, passes, lds,
               ms * 1e3, lines / 1e6, lines * 128 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
