// Replays the exact line lists a staged eq_views kernel would load for cfg2 (scratch/cfg2_plan.bin, 2 passes per tile),
// 8 frames per launch, XCD-chunked; measures the staging-only time (upper bound for any LDS-staged implementation).
// usage: replay_probe <plan.bin> <lds_lines_capacity> <order: 0 view-major, 1 sorted by first line (source-major)> <mode 0 ldsdma / 2 regs>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr size_t kFrameBytes = (size_t)23040 * 3840;
struct P { const uint8_t* frames[8]; const uint32_t* off; const uint32_t* lines; const uint32_t* perm; uint32_t* out; uint8_t* dst; int tiles_per_frame; int cap; };
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE>
__global__ __launch_bounds__(256) void replay(P p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    int b = blockIdx.x;
    int chunk = (gridDim.x + 7) / 8;
    int t = (b & 7) * chunk + (b >> 3);
    if (t >= (int)gridDim.x) return;
    int f = t / p.tiles_per_frame, r = p.perm[t - f * p.tiles_per_frame];
    const uint8_t* src = p.frames[f];
    const int tid = threadIdx.x, grp = tid >> 3, sub = tid & 7;
    uint32_t acc = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t o0 = p.off[r * 2 + pass], n = p.off[r * 2 + pass + 1] - o0;
        for (uint32_t s0 = 0; s0 < n; s0 += p.cap) {
            const uint32_t m = min((uint32_t)p.cap, n - s0);
            uint32_t idx[16];
#pragma unroll
            for (uint32_t it = 0; it < 16; ++it) {          // all plan entries of the stage first (one batch of loads)
                uint32_t slot = it * 32 + grp;
                idx[it] = slot < m ? p.lines[o0 + s0 + slot] : 0xffffffffu;
            }
#pragma unroll
            for (uint32_t it = 0; it < 16; ++it) {
                if (idx[it] != 0xffffffffu) {
                    const uint8_t* g = src + (size_t)idx[it] * 128 + sub * 16;
                    if (MODE == 0)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (lds_void*)(lds + (size_t)it * 4096 + (tid >> 6) * 1024), 16, 0, 0);
                    else { uint4 v = *reinterpret_cast<const uint4*>(g); acc += v.x ^ v.y ^ v.z ^ v.w; }
                }
            }
            if (MODE == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                uint32_t h = tid * 2654435761u + pass * 97u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    h = h * 1664525u + 1013904223u;
                    uint32_t a = ((h >> 8) % (uint32_t)(m * 128 - 16)) & ~3u;
                    acc += *reinterpret_cast<const uint32_t*>(lds + a);
                }
                __syncthreads();
            }
        }
        // the pass's output: 1024 px x 3 B, contiguous, streamed
        uint32_t* d = reinterpret_cast<uint32_t*>(p.dst + ((size_t)t * 2 + pass) * 3072);
        for (int k = tid; k < 768; k += 256) __builtin_nontemporal_store(acc + k, d + k);
    }
    if (acc == 0x12345678u) p.out[t] = acc;
}

int main(int argc, char** argv) {
    FILE* fp = fopen(argv[1], "rb"); if (!fp) { printf("no plan\n"); return 1; }
    uint32_t np; fread(&np, 4, 1, fp);
    std::vector<uint32_t> off(np + 1); fread(off.data(), 4, np + 1, fp);
    std::vector<uint32_t> lines(off[np]); fread(lines.data(), 4, off[np], fp); fclose(fp);
    int cap = argc > 2 ? atoi(argv[2]) : 512, order = argc > 3 ? atoi(argv[3]) : 0, mode = argc > 4 ? atoi(argv[4]) : 0;
    int tiles = np / 2;
    std::vector<uint32_t> perm(tiles);
    for (int i = 0; i < tiles; ++i) perm[i] = i;
    if (order == 1) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        // source-major: by latitude band (96 rows) then by longitude
        uint32_t la = lines[off[a * 2]], lb = lines[off[b * 2]];
        uint32_t ra = la / 180 / 96, rb = lb / 180 / 96;
        if (ra != rb) return ra < rb;
        return la % 180 < lb % 180; });
    P p;
    for (int i = 0; i < 8; ++i) { void* d; CK(hipMalloc(&d, kFrameBytes + 4096)); CK(hipMemset(d, i + 1, kFrameBytes)); p.frames[i] = (const uint8_t*)d; }
    uint32_t *d_off, *d_lines, *d_perm; 
    CK(hipMalloc((void**)&d_off, off.size() * 4)); CK(hipMemcpy(d_off, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_lines, lines.size() * 4)); CK(hipMemcpy(d_lines, lines.data(), lines.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&d_perm, perm.size() * 4)); CK(hipMemcpy(d_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&p.out, 1 << 22)); CK(hipMalloc((void**)&p.dst, (size_t)tiles * 8 * 2 * 3072 + 4096));
    p.off = d_off; p.lines = d_lines; p.perm = d_perm; p.tiles_per_frame = tiles; p.cap = cap;
    size_t ldsb = (size_t)((cap + 31) / 32) * 4096 + 64;
    int grid = tiles * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (mode == 0) { CK(hipFuncSetAttribute((const void*)replay<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); hipLaunchKernelGGL(replay<0>, dim3(grid), dim3(256), ldsb, 0, p); }
        else hipLaunchKernelGGL(replay<2>, dim3(grid), dim3(256), 0, 0, p);
    };
    const int N = argc > 5 ? atoi(argv[5]) : 20;
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
    printf("cap %d order %d mode %d lds %zu: %.1f us/launch (%.2f us/frame), %.2f M line loads/launch\n", cap, order, mode, ldsb, ms * 1e3, ms * 1e3 / 8, off[np] * 8 / 1e6);
    return 0;
}
