// TA / TCP cost of one 64-lane gather instruction as a function of its SHAPE, from an L1-resident 16 KB buffer:
// lane l reads NDW dwords at byte offset ((l / LPR) * RS + (l % LPR) * S + shift) & mask -- LPR lanes per source row,
// S bytes between neighbouring lanes, RS bytes between rows.  Prints cycles per instruction per CU (16 waves per CU);
// run under `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum` for the tag lookups per instruction of each dispatch.
//   hipcc --offload-arch=gfx950 -O2 -o tashape tashape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
template <int N> struct Reg;
template <> struct Reg<1> { typedef unsigned type; };
template <> struct Reg<2> { typedef unsigned type __attribute__((ext_vector_type(2))); };
template <> struct Reg<3> { typedef unsigned type __attribute__((ext_vector_type(3))); };
template <> struct Reg<4> { typedef unsigned type __attribute__((ext_vector_type(4))); };
__device__ inline unsigned first(unsigned v) { return v; }
template <typename T> __device__ inline unsigned first(T v) { return v.x; }
template <int NDW>
__global__ void k(const unsigned* __restrict__ buf, unsigned* out, int iters, int lpr, int S, int RS, unsigned mask) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned off = (lane / lpr) * RS + (lane % lpr) * S + (threadIdx.x >> 6) * 36u;
    unsigned acc = 0;
    const char* base = (const char*)buf;
    for (int i = 0; i < iters; ++i) {
        const unsigned* p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (const unsigned*)(base + ((off + j * 340u + i * 52u) & mask));
        typedef typename Reg<NDW>::type R;
        R r0, r1, r2, r3, r4, r5, r6, r7;
#define LD8(op) asm volatile(op " %0, %8, off\n" op " %1, %9, off\n" op " %2, %10, off\n" op " %3, %11, off\n" op " %4, %12, off\n" op " %5, %13, off\n" op " %6, %14, off\n" op " %7, %15, off\n s_waitcnt vmcnt(0)" \
            : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) \
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]) : "memory")
        if constexpr (NDW == 1) LD8("global_load_dword");
        if constexpr (NDW == 2) LD8("global_load_dwordx2");
        if constexpr (NDW == 3) LD8("global_load_dwordx3");
        if constexpr (NDW == 4) LD8("global_load_dwordx4");
        acc += first(r0) + first(r1) + first(r2) + first(r3) + first(r4) + first(r5) + first(r6) + first(r7);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int NDW>
void run(const unsigned* buf, unsigned* out, int lpr, int S, int RS, unsigned mask) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200, blocks = 256 * 4;   // 4 blocks x 4 waves per CU
    hipLaunchKernelGGL((k<NDW>), dim3(blocks), dim3(256), 0, 0, buf, out, 10, lpr, S, RS, mask); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<NDW>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, lpr, S, RS, mask); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_cu = (double)iters * 8 * 16;
    printf("ndw %d lanes/row %2d stride %2d rowstride %4d align %2d : %.3f ms  cycles/instr/CU@2.4GHz = %5.1f   (wave-instr in the timed dispatch: %d)\n",
           NDW, lpr, S, RS, (mask & 1) ? 1 : ((mask & 2) ? 2 : ((mask & 12) ? 4 : 16)), ms, ms * 1e-3 * 2.4e9 / instr_per_cu, iters * 8 * blocks * 4);
}
int main() {
    unsigned *buf, *out; CK(hipMalloc(&buf, 1 << 16)); CK(hipMemset(buf, 1, 1 << 16)); CK(hipMalloc(&out, 256 * 4 * 256 * 4));
    const int strides[] = {4, 8, 12, 16, 20, 28};
    const int lprs[] = {64, 32, 16, 8, 4};
    for (int lpr : lprs)
        for (int S : strides) {
            if (lpr * S > 1100) continue;
            run<1>(buf, out, lpr, S, 1092, 0x3ffcu);
            run<2>(buf, out, lpr, S, 1092, 0x3ffcu);
            run<3>(buf, out, lpr, S, 1092, 0x3ffcu);
            run<4>(buf, out, lpr, S, 1092, 0x3ffcu);
            run<4>(buf, out, lpr, S, 1092, 0x3ff0u);
        }
    // byte-misaligned lanes (mask keeps every byte offset): what an unaligned tap read would cost
    for (int S : {6, 14}) {
        run<2>(buf, out, 64, S, 1092, 0x3fffu);
        run<3>(buf, out, 64, S, 1092, 0x3fffu);
        run<2>(buf, out, 16, S, 1092, 0x3fffu);
        run<2>(buf, out, 64, S, 1092, 0x3ffeu);
    }
    // perfectly coalesced references
    run<1>(buf, out, 64, 4, 0, 0x3ffcu);
    run<4>(buf, out, 64, 16, 0, 0x3ff0u);
    return 0;
}
