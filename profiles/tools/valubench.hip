// micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the gather kernel uses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define BODY(name, asmstr) \
__global__ void name(unsigned* out, int iters) { \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13, f = 17, g = 19, h = 23; \
    for (int i = 0; i < iters; ++i) { REP64(asm volatile(asmstr : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h; }
// each asm block = 8 independent instructions
BODY(k_fma,  "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1")
BODY(k_mulf, "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %4\n v_mul_f32 %4, %4, %5\n v_mul_f32 %5, %5, %6\n v_mul_f32 %6, %6, %7\n v_mul_f32 %7, %7, %0")
BODY(k_addu, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0")
BODY(k_and,  "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0")
BODY(k_mul24,"v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %4\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %0")
BODY(k_mad24,"v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %3, %3, %4, %5\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %0\n v_mad_u32_u24 %7, %7, %0, %1")
BODY(k_sdwa, "v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_mul_u32_u24_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
BODY(k_add3, "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %1, %1, %2, %3\n v_add3_u32 %2, %2, %3, %4\n v_add3_u32 %3, %3, %4, %5\n v_add3_u32 %4, %4, %5, %6\n v_add3_u32 %5, %5, %6, %7\n v_add3_u32 %6, %6, %7, %0\n v_add3_u32 %7, %7, %0, %1")
BODY(k_mullo,"v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %0")
BODY(k_rcp,  "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7")
BODY(k_cnd,  "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc")
BODY(k_perm, "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %4\n v_perm_b32 %3, %3, %4, %5\n v_perm_b32 %4, %4, %5, %6\n v_perm_b32 %5, %5, %6, %7\n v_perm_b32 %6, %6, %7, %0\n v_perm_b32 %7, %7, %0, %1")
BODY(k_dot4, "v_dot4_u32_u8 %0, %0, %1, %2\n v_dot4_u32_u8 %1, %1, %2, %3\n v_dot4_u32_u8 %2, %2, %3, %4\n v_dot4_u32_u8 %3, %3, %4, %5\n v_dot4_u32_u8 %4, %4, %5, %6\n v_dot4_u32_u8 %5, %5, %6, %7\n v_dot4_u32_u8 %6, %6, %7, %0\n v_dot4_u32_u8 %7, %7, %0, %1")
BODY(k_cvtub,"v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3\n v_cvt_f32_ubyte1 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte1 %6, %6\n v_cvt_f32_ubyte1 %7, %7")
BODY(k_alignb,"v_alignbyte_b32 %0, %0, %1, %2\n v_alignbyte_b32 %1, %1, %2, %3\n v_alignbyte_b32 %2, %2, %3, %4\n v_alignbyte_b32 %3, %3, %4, %5\n v_alignbyte_b32 %4, %4, %5, %6\n v_alignbyte_b32 %5, %5, %6, %7\n v_alignbyte_b32 %6, %6, %7, %0\n v_alignbyte_b32 %7, %7, %0, %1")
BODY(k_lshlor,"v_lshl_or_b32 %0, %0, %1, %2\n v_lshl_or_b32 %1, %1, %2, %3\n v_lshl_or_b32 %2, %2, %3, %4\n v_lshl_or_b32 %3, %3, %4, %5\n v_lshl_or_b32 %4, %4, %5, %6\n v_lshl_or_b32 %5, %5, %6, %7\n v_lshl_or_b32 %6, %6, %7, %0\n v_lshl_or_b32 %7, %7, %0, %1")

#define BODY64(name, asmstr) \
__global__ void name(unsigned* out, int iters) { \
    unsigned long long a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13, f = 17, g = 19, h = 23; \
    for (int i = 0; i < iters; ++i) { REP64(asm volatile(asmstr : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) } \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ h); }
BODY64(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %4\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %4, %4, %5, %6\n v_pk_fma_f32 %5, %5, %6, %7\n v_pk_fma_f32 %6, %6, %7, %0\n v_pk_fma_f32 %7, %7, %0, %1")
BODY64(k_pkmul, "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %6\n v_pk_mul_f32 %6, %6, %7\n v_pk_mul_f32 %7, %7, %0")
BODY64(k_pkadd, "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %6\n v_pk_add_f32 %6, %6, %7\n v_pk_add_f32 %7, %7, %0")
BODY(k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7")
BODY(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7")
BODY(k_cvti, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7")
BODY(k_divfix, "v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %4\n v_div_fixup_f32 %3, %3, %4, %5\n v_div_fixup_f32 %4, %4, %5, %6\n v_div_fixup_f32 %5, %5, %6, %7\n v_div_fixup_f32 %6, %6, %7, %0\n v_div_fixup_f32 %7, %7, %0, %1")
BODY(k_divfmas, "v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %4\n v_div_fmas_f32 %3, %3, %4, %5\n v_div_fmas_f32 %4, %4, %5, %6\n v_div_fmas_f32 %5, %5, %6, %7\n v_div_fmas_f32 %6, %6, %7, %0\n v_div_fmas_f32 %7, %7, %0, %1")
BODY(k_minf, "v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0")
BODY(k_cmp, "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %4\n v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %5, %6\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %7, %0")
BODY(k_swap, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0")
typedef void (*kern_t)(unsigned*, int);
int main() {
    struct { const char* name; kern_t k; bool pk; } ks[] = { {"v_fma_f32", k_fma, 0}, {"v_mul_f32", k_mulf, 0}, {"v_add_u32", k_addu, 0}, {"v_and_b32", k_and, 0},
        {"v_mul_u32_u24", k_mul24, 0}, {"v_mad_u32_u24", k_mad24, 0}, {"v_mul_u32_u24_sdwa", k_sdwa, 0}, {"v_add3_u32", k_add3, 0}, {"v_mul_lo_u32", k_mullo, 0},
        {"v_rcp_f32", k_rcp, 0}, {"v_cndmask_b32", k_cnd, 0}, {"v_perm_b32", k_perm, 0}, {"v_dot4_u32_u8", k_dot4, 0}, {"v_cvt_f32_ubyte1", k_cvtub, 0},
        {"v_alignbyte_b32", k_alignb, 0}, {"v_lshl_or_b32", k_lshlor, 0},
        {"v_pk_fma_f32", k_pkfma, 1}, {"v_pk_mul_f32", k_pkmul, 1}, {"v_pk_add_f32", k_pkadd, 1}, {"v_sqrt_f32", k_sqrt, 0}, {"v_rndne_f32", k_rndne, 0},
        {"v_cvt_i32_f32", k_cvti, 0}, {"v_div_fixup_f32", k_divfix, 0}, {"v_div_fmas_f32", k_divfmas, 0}, {"v_min_f32", k_minf, 0}, {"v_cmp_gt_f32", k_cmp, 0},
        {"v_permlane32_swap", k_swap, 0} };
    unsigned* out; CK(hipMalloc(&out, 256 * 2048 * 8 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    double clk = p.clockRate * 1e3;   // Hz (max)
    int iters = 200;
    for (auto& kk : ks) {
        for (int waves_per_simd : {1, 8}) {
            int blocks = 256 * waves_per_simd;     // 256 threads = 4 waves = one per SIMD; blocks per CU = waves_per_simd
            hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, out, 10); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(kk.k, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double inst_per_simd = (double)iters * 64 * 8 * waves_per_simd;  // wave-instructions issued on one SIMD
            printf("%-22s waves/SIMD=%d  %.3f ms  ns/inst/SIMD=%.3f  cycles@2.4GHz=%.2f\n", kk.name, waves_per_simd, ms, ms * 1e6 / inst_per_simd, ms * 1e-3 * 2.4e9 / inst_per_simd);
        }
    }
    (void)clk;
    return 0;
}
