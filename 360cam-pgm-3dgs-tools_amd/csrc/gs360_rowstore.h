// gs360_rowstore.h -- row stores shared by the kernels of gs360_kernels.hip and gs360_u16.hip: the per-lane constants of the
// dword re-slicing (RowPack) and the packed store of 16-bit pixels.
#pragma once
#include "gs360_kernels.h"

namespace gs360 {

// The two per-lane constants of the dword re-slicing are computed once per tile by the caller (RowPack): inside the
// ring-member loop of eq_views_kernel they must neither be recomputed by every one of the 8 row stores of an iteration nor
// be left to the compiler's hoisting, which drags every other invariant of the store paths along and spills.
struct RowPack {
    int a4;   // 4 * ((4 * lane) / 3): ds_bpermute address of the first pixel contributing to dword `lane` of the row
    int sh;   // 8 * ((4 * lane) % 3): its bit offset
    uint32_t sel;   // v_perm_b32 selector that cuts dword `lane` out of the two 24-bit pixels (pa in bytes 0..3, pb in 4..7)
    int lane;
};
__device__ __forceinline__ RowPack make_row_pack(const int lane) {
    const int t = lane / 3;
    RowPack rp;
    rp.a4 = 4 * (lane + t);
    rp.sh = 8 * (lane - 3 * t);
    rp.sel = rp.sh == 0 ? 0x04020100u : (rp.sh == 8 ? 0x05040201u : 0x06050402u);
    rp.lane = lane;
    return rp;
}
__device__ __forceinline__ RowPack make_row_pack() { return make_row_pack((int)(threadIdx.x & 63)); }
// Row store of 16-bit pixels.  RGB: lane l holds pixel l (or n_px-1-l) as A = c0 | c1 << 16, B = c2; the row's dword stream
// takes dword 3m from A[2m], 3m+1 from B[2m] | A[2m+1] << 16, 3m+2 from A[2m+1] >> 16 | B[2m+1] << 16: two cross-lane reads per
// dword, 96 dwords per 64 pixels = one and a half store instructions of whole dwords.
template <int C>
__device__ __forceinline__ void store_row16(uint8_t* row, const uint32_t (&px)[4], int n_px, bool aligned4, const RowPack& rp,
                                            bool reversed, bool skip_first) {
    const int lane = rp.lane;
    const int pos = reversed ? n_px - 1 - lane : lane;
    if constexpr (C == 3) {
        if (aligned4 && (n_px & 1) == 0) {
            const uint32_t A = px[0] | (px[1] << 16), B = px[2];
            const int n_dw = (3 * n_px) >> 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int d = lane + 64 * j;
                const int m = (d * 21846) >> 16, r = d - 3 * m;            // d / 3, d % 3 for d < 2^15
                const int pa = r == 0 ? 2 * m : 2 * m + 1, pb = r == 1 ? 2 * m : 2 * m + 1;    // pixels that supply A / B
                const int la = reversed ? n_px - 1 - pa : pa, lb = reversed ? n_px - 1 - pb : pb;
                const uint32_t va = (uint32_t)__builtin_amdgcn_ds_bpermute((la & 63) << 2, (int)A);
                const uint32_t vb = (uint32_t)__builtin_amdgcn_ds_bpermute((lb & 63) << 2, (int)B);
                const uint32_t dw = r == 0 ? va : (r == 1 ? ((vb & 0xffffu) | (va << 16)) : ((va >> 16) | (vb << 16)));
                if (d < n_dw) __builtin_nontemporal_store(dw, reinterpret_cast<uint32_t*>(row) + d);
            }
            return;
        }
    }
    if (lane < n_px && !(skip_first && pos == 0)) {
        uint8_t* q = row + (int64_t)pos * (2 * C);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint16_t v = (uint16_t)px[c];
            __builtin_memcpy(q + 2 * c, &v, 2);
        }
    }
}

}  // namespace gs360
