// micro-benchmark: TCP tag lookups per wave-level load instruction for several access shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
template<int MODE> __global__ void k(const uint8_t* __restrict__ src, uint32_t* out, int stride_bytes, int iters) {
    int lane = threadIdx.x & 63;
    size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        size_t base = (wave * iters + it) * 4096;   // each wave-iteration gets its own 4 KB window
        if (MODE == 0) { uint4 v = *(const uint4*)(src + base + lane * 16); acc += v.x ^ v.y ^ v.z ^ v.w; }           // coalesced 16B
        if (MODE == 1) { uint32_t v = *(const uint32_t*)(src + base + lane * 4); acc += v; }                         // coalesced 4B
        if (MODE == 2) { uint2 v; __builtin_memcpy(&v, src + base + lane * stride_bytes, 8); acc += v.x ^ v.y; }      // unaligned 8B at stride
        if (MODE == 3) { const uint8_t* p = src + base + lane * stride_bytes; uint32_t o = (uintptr_t)p & 3; const uint32_t* q = (const uint32_t*)__builtin_assume_aligned(p - o, 4); acc += q[0] ^ q[1] ^ q[2]; } // aligned 12B
        if (MODE == 4) { uint2 v = *(const uint2*)(src + base + lane * 8); acc += v.x ^ v.y; }                        // coalesced 8B
        if (MODE == 5) { const uint8_t* p = src + base + lane * stride_bytes; const uint4* q = (const uint4*)((uintptr_t)p & ~(uintptr_t)15); uint4 v = *q; acc += v.x ^ v.y ^ v.z ^ v.w; } // aligned 16B containing p
        if (MODE == 6) { const uint8_t* p = src + base + lane * stride_bytes; const uint2* q = (const uint2*)((uintptr_t)p & ~(uintptr_t)7); uint2 v = *q; acc += v.x ^ v.y; } // aligned 8B containing p
    }
    out[wave * 64 + lane] = acc;
}
int main(int argc, char** argv) {
    int mode = atoi(argv[1]); int stride = argc > 2 ? atoi(argv[2]) : 14;
    int blocks = 2048, iters = 16; size_t waves = (size_t)blocks * 4; size_t bytes = waves * iters * 4096 + 4096;
    uint8_t* src; uint32_t* out; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&out, waves * 64 * 4)); CK(hipMemset(src, 1, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, src, out, stride, iters); break;
        }
        CK(hipDeviceSynchronize());
    }
    printf("mode %d stride %d wave_loads %zu\n", mode, stride, waves * iters);
    return 0;
}
