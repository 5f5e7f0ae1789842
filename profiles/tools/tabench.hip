// TA throughput of 64-lane gathers from an L1-resident 16 KB buffer: cycles per instruction for dword/x2/x3/x4 and for
// partially active wavefronts (scattered vs contiguous active lanes).
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
template <int NDW, int MODE>
__global__ void k(const unsigned* __restrict__ buf, unsigned* out, int iters) {
    const unsigned lane = threadIdx.x & 63;
    // addresses: 13.8-byte stride like the kernel (aligned down to 4), all inside 16 KB
    unsigned off = ((lane * 14u + (threadIdx.x >> 6) * 1024u) & 0x3ffcu);
    bool active = MODE == 0 ? true : (MODE == 1 ? (lane & 3) == 3 : (MODE == 2 ? lane < 16 : (lane & 1) == 0));
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        if (active) {
            const char* base = (const char*)buf;
            const unsigned* p0 = (const unsigned*)(base + ((off + 0 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p1 = (const unsigned*)(base + ((off + 1 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p2 = (const unsigned*)(base + ((off + 2 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p3 = (const unsigned*)(base + ((off + 3 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p4 = (const unsigned*)(base + ((off + 4 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p5 = (const unsigned*)(base + ((off + 5 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p6 = (const unsigned*)(base + ((off + 6 * 448u + i * 64u) & 0x3ffcu));
            const unsigned* p7 = (const unsigned*)(base + ((off + 7 * 448u + i * 64u) & 0x3ffcu));
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            typedef unsigned u3 __attribute__((ext_vector_type(3)));
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            typedef typename Reg<NDW>::type R;
            R r0, r1, r2, r3, r4, r5, r6, r7;
#define LD8(op) asm volatile(op " %0, %8, off\n" op " %1, %9, off\n" op " %2, %10, off\n" op " %3, %11, off\n" op " %4, %12, off\n" op " %5, %13, off\n" op " %6, %14, off\n" op " %7, %15, off\n s_waitcnt vmcnt(0)" \
                : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) \
                : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7) : "memory")
            if constexpr (NDW == 1) LD8("global_load_dword");
            if constexpr (NDW == 2) LD8("global_load_dwordx2");
            if constexpr (NDW == 3) LD8("global_load_dwordx3");
            if constexpr (NDW == 4) LD8("global_load_dwordx4");
            acc += first(r0) + first(r1) + first(r2) + first(r3) + first(r4) + first(r5) + first(r6) + first(r7);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int NDW, int MODE>
void run(const char* name, const unsigned* buf, unsigned* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 400, blocks = 256 * 4;   // 4 blocks x 4 waves per CU
    hipLaunchKernelGGL((k<NDW, MODE>), dim3(blocks), dim3(256), 0, 0, buf, out, 10); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<NDW, MODE>), dim3(blocks), dim3(256), 0, 0, buf, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_cu = (double)iters * 8 * 16;   // 16 waves per CU
    printf("%-34s %.3f ms  cycles/instr/CU@2.4GHz = %.1f\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
}
int main() {
    unsigned *buf, *out; CK(hipMalloc(&buf, 1 << 16)); CK(hipMemset(buf, 1, 1 << 16)); CK(hipMalloc(&out, 256 * 4 * 256 * 4));
    run<1, 0>("dword   all lanes", buf, out);
    run<2, 0>("dwordx2 all lanes", buf, out);
    run<3, 0>("dwordx3 all lanes", buf, out);
    run<4, 0>("dwordx4 all lanes", buf, out);
    run<3, 1>("dwordx3 lanes with l%4==3", buf, out);
    run<3, 2>("dwordx3 lanes 0..15", buf, out);
    run<3, 3>("dwordx3 even lanes", buf, out);
    run<2, 1>("dwordx2 lanes with l%4==3", buf, out);
    run<1, 1>("dword   lanes with l%4==3", buf, out);
    return 0;
}
