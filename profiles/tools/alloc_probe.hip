#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    CK(hipFree(0));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        void *d = nullptr, *h = nullptr;
        double t0 = now(); CK(hipMalloc(&d, 300u << 20)); double t1 = now(); CK(hipHostMalloc(&h, 108u << 20, hipHostMallocDefault)); double t2 = now();
        printf("rep %d: hipMalloc 300 MiB %.2f ms, hipHostMalloc 108 MiB %.2f ms\n", rep, t1 - t0, t2 - t1);
        const int n = 1300000; size_t cub = 0;
        uint32_t* k = (uint32_t*)d; uint32_t* ko = k + n; uint32_t* v = ko + n; uint32_t* vo = v + n; void* tmp = vo + n;
        double t3 = now();
        CK(hipcub::DeviceRadixSort::SortPairs(nullptr, cub, k, ko, v, vo, n, 0, 32, s));
        CK(hipcub::DeviceRadixSort::SortPairs(tmp, cub, k, ko, v, vo, n, 0, 32, s)); CK(hipStreamSynchronize(s));
        double t4 = now();
        CK(hipcub::DeviceRadixSort::SortPairs(tmp, cub, k, ko, v, vo, n, 0, 32, s)); CK(hipStreamSynchronize(s));
        double t5 = now();
        CK(hipMemcpyAsync(h, d, 83u << 20, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        double t6 = now();
        void* pg = malloc(83u << 20);
        CK(hipMemcpy(pg, d, 83u << 20, hipMemcpyDeviceToHost));
        double t7 = now();
        CK(hipMemcpy(pg, d, 83u << 20, hipMemcpyDeviceToHost));
        double t8 = now();
        printf("rep %d: first sort %.2f ms, second sort %.2f ms, D2H 83 MiB pinned %.2f ms, pageable first %.2f ms, pageable again %.2f ms\n", rep, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7);
        free(pg);
        double t9 = now(); CK(hipHostFree(h)); CK(hipFree(d)); printf("rep %d: frees %.2f ms\n", rep, now() - t9);
    }
    return 0;
}
