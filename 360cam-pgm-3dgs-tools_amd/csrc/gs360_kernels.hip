// gs360_kernels.hip -- hand-written gfx950 kernels for the 360PerspCut reprojection hot path.
//
//   eq_views_kernel      equirect -> rectilinear views, analytic in-kernel map (EQ-SPEC v1), bilinear or cubic, optional
//                        fused keep-mask; replaces the per-view ffmpeg v360 processes of
//                        cli_tools/gs360_360PerspCut.py:310-314.
//   table_remap_kernel   cv2.remap(INTER_NEAREST|LINEAR|CUBIC, BORDER_CONSTANT) + valid fill, replaces
//                        cli_tools/gs360_DualFisheyeDistortionCalibration.py:2001-2014 / :2031-2043 / :1198-1212.
//   fe_views_kernel      fused dual-fisheye -> perspective (FE-SPEC v1), DF:1759-1823 evaluated in-kernel.
//
// All three are memory-system-bound byte gathers (no contraction -> no MFMA).  Work decomposition: one 64-column x
// 16-row output tile per 256-thread workgroup; a wavefront owns 4 row slots of the tile and its 64 lanes are 64
// CONSECUTIVE pixels of one output row.  A wavefront first computes the source coordinates of all its pixels, then
// issues ALL tap reads with no control flow in between (dword-aligned 12-byte reads + v_alignbyte: the texture-address
// path merges aligned lane accesses into line requests, misaligned ones are looked up lane by lane), then blends.
// Each lane packs its RGB result into a dword and the row is written as whole dwords after a two-shuffle
// (ds_bpermute) repack, i.e. 192 contiguous bytes per wavefront row.  For strongly minified RGB views the equirect
// kernel switches (per view, host decision) to a BLOCKED lane map: every gather covers a 4-row x 16-column patch,
// which touches about half the cache lines where a view row bends across many source rows; results are then
// transposed through LDS into row-segment stores (store_patch_rgb).  Tiles are numbered row-major per
// view and handed to XCDs in contiguous chunks (block b runs on XCD b % 8) so that neighbouring tiles --
// which share source cache lines -- hit the same per-XCD L2.  DESIGN.md section 5 and profiles/HISTORY.md have the measurements behind
// each of these choices.
//
// Compile with -ffp-contract=off: the float32 specs are defined operation by operation and must match the
// CPU oracle bit for bit; only explicit __builtin_fmaf may fuse.
#include <type_traits>

#include "gs360_sampler.h"

namespace gs360 {

constexpr int kEqWaves = 5;     // wavefronts per SIMD of the bilinear equirect kernel (measured optimum, see the kernel comment)
constexpr int kEqRowsKernel = 1;
constexpr int kEqRowsWaves = 5;    // bilinear RGB kernel of launches without blocked views (pipelined member loop only)
constexpr int kEqStagedWaves = 4;    // LDS-staged bilinear RGB kernel: 40 KiB of LDS per workgroup = four workgroups per CU
constexpr int kStageEnable = 1; // 0: every pass of eq_staged_kernel takes the gather form (A/B of the lane map alone)
constexpr int kEqCubicWaves = 4;    // wavefronts per SIMD of the cubic equirect kernel (124 registers; 40 KiB of LDS = four workgroups per CU)
constexpr int kRingPark = 0;    // bilinear kernel: ring-shared coordinates that wait in LDS between members (0 none, 1 latitude, 3 all)
constexpr int kRingParkCubic = 1;
constexpr int kEqLean = 1;      // bilinear RGB row-per-slot views: the lean, software-pipelined member loop (0: the round-3 loop, A/B reference)


// Store for the blocked lane map (RGB).  A wavefront holds a patch of 4 rows x (16 NS) columns in NS slots -- lane l of
// slot s0+s is pixel (row l>>4, column 16 s + (l&15)).  Pixels are written as packed dwords into the wavefront's LDS
// slice in memory order (mirrored passes write mirrored positions) and read back as the 3-byte-packed dword stream
// of each row, so a store instruction writes 256 contiguous bytes of the patch.  Row r of the patch is image row
// y0 + ystep r (ystep = -1 for the horizon-mirrored patch of a level view); rows >= nrows and columns >= n_w are
// not written.  Falls back to per-lane byte stores when the row segment is not dword-aligned.
template <int NS>
__device__ __forceinline__ void store_patch_rgb(uint8_t* dst, int64_t dstride, uint32_t* lds, const uint32_t (&px)[kRowsPerWave][4],
                                                int s0, int y0, int ystep, int nrows, int col0, int n_w, bool aligned4,
                                                bool reversed, bool skip_first) {
    const int lane = lane_here(), r = lane >> 4, c16 = lane & 15;
    if (aligned4 && (n_w & 3) == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int col = 16 * s + c16;
            const int pos = reversed ? n_w - 1 - col : col;
            if (col < n_w) lds[r * 64 + pos] = px[s0 + s][0] | (px[s0 + s][1] << 8) | (px[s0 + s][2] << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int dpr = (3 * n_w) >> 2;                    // dwords per row segment
#pragma unroll
        for (int j = 0; j < (NS * 48 + 63) / 64; ++j) {
            const int d = lane + 64 * j;
            const int rr = (d >= dpr ? 1 : 0) + (d >= 2 * dpr ? 1 : 0) + (d >= 3 * dpr ? 1 : 0);
            const int k = d - rr * dpr;
            const int k3 = (k * 21846) >> 16;              // k / 3 for k < 2^15
            const int a = k + k3, sh = 8 * (k - 3 * k3);   // (4k)/3, 8*((4k)%3)
            const bool live = d < 4 * dpr && rr < nrows;
            const int idx = live ? rr * 64 + a : 0;
            const uint32_t pa = lds[idx], pb = lds[idx + 1];
            const uint32_t dw = (pa >> sh) | (pb << (24 - sh));
            if (live) {
                uint32_t* q = reinterpret_cast<uint32_t*>(dst + (int64_t)(y0 + ystep * rr) * dstride + (int64_t)col0 * 3) + k;
                // 4-slot patches leave as whole 192-byte row segments: streamed past the caches.  The 96-byte halves of a
                // level view's row segment come from two wavefronts: regular stores let L2 merge them into full lines
                // (non-temporal: WRITE_SIZE +17 %, launch +3.5 %).
                if (NS == 4) __builtin_nontemporal_store(dw, q); else *q = dw;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int col = 16 * s + c16;
        const int pos = reversed ? n_w - 1 - col : col;
        if (col < n_w && r < nrows && !(skip_first && pos == 0)) {
            uint8_t* q = dst + (int64_t)(y0 + ystep * r) * dstride + (int64_t)(col0 + pos) * 3;
            q[0] = (uint8_t)px[s0 + s][0]; q[1] = (uint8_t)px[s0 + s][1]; q[2] = (uint8_t)px[s0 + s][2];
        }
    }
}

// How eq_pass hands its pixels to the blocked store (wave-uniform).  mode 0: row-per-slot map (store_row).
struct BlkStore {
    uint32_t* lds;       // this wavefront's 256-dword slice
    int mode;            // 1: general view, 4 slots = one 4x64 patch;  2: level view, slots {0,1} top patch, {2,3} its horizon mirror
    int y0, nrows;       // first image row of the (top) patch, valid rows
    int ystep;           // mode 1: +1, or -1 for a flipped ring member (rows run upwards from y0)
    int yb, nrows_b;     // mode 2: first row of the mirrored patch (rows run upwards), valid rows
    int sub;             // mode 2: first tile column of this wavefront's 32-column half
};

// Tap fetch for the equirect sampler, split in two so that all gathers of a wavefront can be in flight at once:
//   eq_fetch   issues the two row reads of one pixel with NO control flow (the column is clamped so that the
//              2*C-byte read never leaves the row); lanes whose right tap wraps around the 360-degree seam, or
//              sits in the last columns, are flagged and repaired later by eq_sample_slow.
//   eq_blend   unpacks the taps and applies the 1/32-px fixed-point bilinear weights.
template <int C>
__device__ __forceinline__ EqTaps<C> eq_fetch(const uint8_t* __restrict__ src, int64_t stride, int W, int H, int sx, int sy) {
    const int ix = sx >> 5, iy = sy >> 5;
    // |lat| <= pi/2 by construction of eq_atan2_red, so sy lies in [-16, 32 H - 16] and iy in [-1, H - 1]: one clamp per row
    const int y0 = max(iy, 0), y1 = min(iy + 1, H - 1);
    constexpr int kBack = (C == 3) ? 5 : 2;      // the 12-byte aligned read of RGB taps may run 6 bytes past them
    const int ixl = min(ix, W - kBack);
    // 32-bit byte offsets from the wave-uniform frame base (host guarantees H * stride < 2^32, stride < 2^24):
    // one full-rate v_mad_u32_u24 per row instead of 64-bit multiply/add chains, and the load can use the
    // SGPR-base + VGPR-offset addressing form.
    const uint32_t col = (uint32_t)ixl * C;
    const uint32_t o0 = __umul24((uint32_t)y0, (uint32_t)stride) + col;
    const uint32_t o1 = __umul24((uint32_t)y1, (uint32_t)stride) + col;
    const uint8_t* r0 = src + o0;
    const uint8_t* r1 = src + o1;
    EqTaps<C> t;
    t.fix = ix != ixl;
    if constexpr (C == 1) {
        uint16_t a, b;
        __builtin_memcpy(&a, r0, 2);
        __builtin_memcpy(&b, r1, 2);
        t.t0 = make_uint2(a, 0);
        t.t1 = make_uint2(b, 0);
    } else if constexpr (C == 3) {
        t.raw = ld_rows_rgb_issue(src, o0, o1);
    } else {
        t.t0 = ld_u64(r0);
        t.t1 = ld_u64(r1);
    }
    return t;
}
template <int C>
__device__ __forceinline__ void eq_taps_finish(EqTaps<C>& t) {
    if constexpr (C == 3) ld_rows_rgb_finish(t.raw, t.t0, t.t1);
}

// (dot2_i16, dot2_i16_from, GS360_PAIR and the RGB row blend: gs360_blend.h, shared with gs360_srcmajor.hip)

// byte-wise path with the horizontal wrap (ix + 1 == W -> column 0); used only for flagged lanes
template <int C>
__device__ __forceinline__ void eq_sample_slow(const uint8_t* __restrict__ src, int64_t stride, int W, int H,
                                               int sx, int sy, uint32_t (&out)[4]) {
    const int fx = sx & 31, ix = sx >> 5, fy = sy & 31, iy = sy >> 5;
    const int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
    const uint8_t* r0 = src + (int64_t)y0 * stride;
    const uint8_t* r1 = src + (int64_t)y1 * stride;
    const uint32_t a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
    const uint32_t w00 = a0 * b0, w01 = a1 * b0, w10 = a0 * b1, w11 = a1 * b1;
    const int ix1 = (ix + 1 == W) ? 0 : ix + 1;
#pragma unroll
    for (int c = 0; c < C; ++c)
        out[c] = blend(r0[ix * C + c], r0[ix1 * C + c], r1[ix * C + c], r1[ix1 * C + c], w00, w01, w10, w11);
}

template <int C>
__device__ __forceinline__ void eq_cubic_slow(const EqSrc& L, const int16_t* wtab, const uint8_t* __restrict__ src, int sx, int sy, uint32_t (&out)[4]) {
    const int fx = sx & 31, fy = sy & 31, ix = sx >> 5, iy = sy >> 5;
    uint32_t wpk[8];
    cubic_lds_weights(wtab, fy * 32 + fx, wpk);
    int cols[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
        int xx = ix - 1 + kx;
        cols[kx] = xx < 0 ? xx + L.W : (xx >= L.W ? xx - L.W : xx);
    }
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint8_t* row = src + (int64_t)min(max(iy - 1 + ky, 0), L.H - 1) * L.src_stride;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const uint32_t pk = wpk[(ky * 4 + kx) >> 1];
            const int w = (int)(int16_t)((kx & 1) ? (pk >> 16) : (pk & 0xffffu));
            const uint8_t* px = row + (int64_t)cols[kx] * C;
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int)px[c] * w;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = (uint32_t)min(max((acc[c] + (1 << 14)) >> 15, 0), 255);
}

// One pass over the wavefront's 4 row slots for one column per lane: all gathers first, then blend, repair, store.
//   reversed = false: lane l is pixel l of the row segment starting at `col0`
//   reversed = true : lane l is pixel n_px-1-l (the mirrored half of the view)
// keep-mask sample of one pixel: nearest texel of the EQ-SPEC coordinate, wrap in x, clamp in y.  The kernels only ever test
// `mask < 128`, so the C ABI thresholds each mask once per call into a BIT image (mask_pack_kernel below): row pitch
// `mask_stride` bytes (whole dwords), one extra column (bit W = bit 0: the wrap) and one extra row (row H = row H - 1: the
// clamp; sy + 16 lies in [0, 32 H] because |lat| <= pi/2), so the sample is an aligned dword read + a bit test, no wrap, no
// clamp, from an image that stays in L2 (3.7 MB for an 8K frame instead of 29.5 MB of bytes).  Returns the dword shifted so that
// bit 0 is the keep bit.
__device__ __forceinline__ uint32_t eq_mask_at(const EqSrc& L, const uint8_t* __restrict__ mask, int sx, int sy) {
    const uint32_t t = (uint32_t)(sx + 16);                      // xn = t >> 5 in [0, W]
    const uint32_t row = __umul24((uint32_t)(sy + 16) >> 5, (uint32_t)L.mask_stride);
    const uint32_t w = *reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(mask + (size_t)(row + ((t >> 8) & ~3u)), 4));
    return w >> ((t >> 5) & 31u);
}

// Hand the wavefront's pixels to memory: blocked patches (store_patch_rgb) or one row per slot (store_row).
// MODE (compile time; the kernel switches once per tile, outside the ring-member loop, so that each loop body carries only
// its own lane map's invariants): 0 = one row per slot, 1 = blocked patches of a general view, 2 = blocked level view.
template <int C, int MODE, bool SHIFTED = true>
__device__ __forceinline__ void eq_store(uint8_t* dst, int64_t dstride, const uint32_t (&px)[kRowsPerWave][4],
                                         const int (&ys)[kRowsPerWave], const bool (&row_ok)[kRowsPerWave],
                                         int col0, int n_px, bool reversed, bool aligned4, bool skip_first,
                                         const BlkStore blk, const RowPack& rp) {
    if constexpr (C == 3 && kRowsPerWave == 4 && kWaves == 4) {   // == kBlocked of the kernel
      if constexpr (MODE != 0) {
        if constexpr (MODE == 1) {
            store_patch_rgb<4>(dst, dstride, blk.lds, px, 0, blk.y0, blk.ystep, blk.nrows, col0, n_px, aligned4, reversed, skip_first);
            return;
        } else {
            // this wavefront owns tile columns [sub, sub + n_w); in the mirrored pass they sit at the other end of the segment
            const int n_w = min(max(n_px - blk.sub, 0), 32);
            const int c0 = reversed ? col0 + n_px - blk.sub - n_w : col0 + blk.sub;
            const bool al = aligned4 && (((c0 * 3) & 3) == 0);
            const bool skip = skip_first && n_w > 0 && blk.sub + n_w == n_px;   // the centre column is this half's last one
            store_patch_rgb<2>(dst, dstride, blk.lds, px, 0, blk.y0, 1, blk.nrows, c0, n_w, al, reversed, skip);
            store_patch_rgb<2>(dst, dstride, blk.lds, px, 2, blk.yb, -1, blk.nrows_b, c0, n_w, al, reversed, skip);
            return;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s)
        if (row_ok[s]) store_row<C, SHIFTED>(dst + (int64_t)ys[s] * dstride + (int64_t)col0 * C, px[s], n_px, aligned4, rp, reversed, skip_first);
}

// The bilinear pass in two steps, so that the ring-member loop can put the NEXT pass's gathers in flight before the current
// pass's pixels are shuffled and stored (eq_views_tile):
//   eq_pass_issue    the tap reads (and keep-mask bytes) of the wavefront's four row slots, no control flow, nothing consumed
//   eq_pass_resolve  shift / blend / repair of flagged lanes / keep-mask -> px
template <int C, bool MASKED>
struct EqPassTaps {
    EqTaps<C> taps[kRowsPerWave];
    uint32_t keep[MASKED ? kRowsPerWave : 1];
};
template <int C, bool MASKED>
__device__ __forceinline__ void eq_pass_issue(const EqSrc& L, const uint8_t* __restrict__ src, const uint8_t* __restrict__ mask,
                                              const int (&sxs)[kRowsPerWave], const int (&sys)[kRowsPerWave], EqPassTaps<C, MASKED>& T) {
    if constexpr (MASKED) {                               // compile time: the mask reads join the tap reads in flight (behind a
#pragma unroll                                            // run-time test the compiler waits for each of them inside the branch)
        for (int s = 0; s < kRowsPerWave; ++s) T.keep[s] = eq_mask_at(L, mask, sxs[s], sys[s]);
    }
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s) T.taps[s] = eq_fetch<C>(src, L.src_stride, L.W, L.H, sxs[s], sys[s]);
}
template <int C, bool MASKED>
__device__ __forceinline__ void eq_pass_resolve(const EqSrc& L, const uint8_t* __restrict__ src,
                                                const int (&sxs)[kRowsPerWave], const int (&sys)[kRowsPerWave],
                                                EqPassTaps<C, MASKED>& T, uint32_t (&px)[kRowsPerWave][4]) {
    bool any_fix = false;
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s) {
        any_fix |= T.taps[s].fix;
        eq_taps_finish<C>(T.taps[s]);
        eq_blend<C>(T.taps[s], sxs[s], sys[s], px[s]);
    }
    if (any_lane(any_fix)) {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s)
            if (T.taps[s].fix) eq_sample_slow<C>(src, L.src_stride, L.W, L.H, sxs[s], sys[s], px[s]);
    }
    if constexpr (MASKED) {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s)
            if (!(T.keep[s] & 1u)) px[s][0] = px[s][1] = px[s][2] = px[s][3] = 0;
    }
}

template <int C, bool CUBIC, int MODE, bool MASKED>
__device__ __forceinline__ void eq_pass(const EqSrc& L, const uint8_t* __restrict__ src, const uint8_t* __restrict__ mask,
                                        uint8_t* dst, int64_t dstride,
                                        const int (&sxs)[kRowsPerWave], const int (&sys)[kRowsPerWave],
                                        const int (&ys)[kRowsPerWave], const bool (&row_ok)[kRowsPerWave],
                                        int col0, int n_px, bool reversed, bool aligned4, bool skip_first,
                                        const int16_t* wtab, const BlkStore blk, const RowPack& rp) {
    if constexpr (CUBIC) {
        uint32_t px[kRowsPerWave][4];
        if constexpr (C == 3) {
            // two row slots at a time: 8 tap reads + 4 weight reads in flight, 24 tap dwords live
#pragma unroll
            for (int s0 = 0; s0 < kRowsPerWave; s0 += 2) {
                EqCubicTaps ta = eq_cubic_fetch(L, src, sxs[s0], sys[s0]);
                EqCubicTaps tb = eq_cubic_fetch(L, src, sxs[s0 + 1], sys[s0 + 1]);
                __builtin_amdgcn_sched_barrier(0);        // all 8 row reads in flight before the first result is touched
                eq_cubic_blend(ta, wtab, L.stride4, px[s0]);
                eq_cubic_blend(tb, wtab, L.stride4, px[s0 + 1]);
                if (any_lane(ta.fix | tb.fix)) {
                    if (ta.fix) eq_cubic_slow<C>(L, wtab, src, sxs[s0], sys[s0], px[s0]);
                    if (tb.fix) eq_cubic_slow<C>(L, wtab, src, sxs[s0 + 1], sys[s0 + 1], px[s0 + 1]);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) eq_cubic_slow<C>(L, wtab, src, sxs[s], sys[s], px[s]);
        }
        if constexpr (MASKED) {
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s)
                if (!(eq_mask_at(L, mask, sxs[s], sys[s]) & 1u)) px[s][0] = px[s][1] = px[s][2] = px[s][3] = 0;
        }
        eq_store<C, MODE, false>(dst, dstride, px, ys, row_ok, col0, n_px, reversed, aligned4, skip_first, blk, rp);
        return;
    }
    EqPassTaps<C, MASKED> T;
    eq_pass_issue<C, MASKED>(L, src, mask, sxs, sys, T);
    __builtin_amdgcn_sched_barrier(0);                    // every gather of the pass is in flight before the first is consumed
    uint32_t px[kRowsPerWave][4];
    eq_pass_resolve<C, MASKED>(L, src, sxs, sys, T, px);
    eq_store<C, MODE>(dst, dstride, px, ys, row_ok, col0, n_px, reversed, aligned4, skip_first, blk, rp);
}

// ------------------------------------------------------------------------------------------------
// 16-bit samples (16-bit stills keep their depth through the reference's ffmpeg path, PC:327-347; > 8-bit videos leave as
// rgb48le, PC:343-347): the equirect kernel's skeleton -- mirror / horizon symmetry, yaw rings, paired row gathers, all gathers
// of a pass in flight -- with 2-byte elements.  Same quantised coordinates and the same integer arithmetic as the 8-bit
// sampler: bilinear (sum S a b + 512) >> 10, bicubic with the fixed-point Keys table, columns wrap, rows clamp.
// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2_u16(uint32_t a, uint32_t b, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), acc, false);
}
__device__ __forceinline__ uint32_t udot2_u16_from(uint32_t a, uint32_t b, uint32_t start_uniform) {   // see dot2_i16_from
    uint32_t r;
    asm("v_dot2_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(start_uniform));
    return r;
}
__device__ __forceinline__ uint16_t ld_u16(const uint8_t* p) {
    uint16_t v;
    __builtin_memcpy(&v, p, 2);
    return v;
}

// RGB bilinear: the two taps of a row are 12 contiguous bytes -> one dword-aligned 16-byte read per row (rows paired across
// the wavefront halves like the 8-bit fetch), shifted into place afterwards.
struct Eq16Taps {
    uint32_t a[4], b[4];
    uint32_t sh;          // (o0 & 3) | (o1 & 3) << 2
    bool fix;
};
__device__ __forceinline__ Eq16Taps eq16_issue_rgb(const uint8_t* __restrict__ src, uint32_t stride, int W, int H, int sx, int sy) {
    const int ix = sx >> 5, iy = sy >> 5;
    const int y0 = max(iy, 0), y1 = min(iy + 1, H - 1);
    const int ixl = min(ix, W - 3);                        // the 16-byte read of the 12 tap bytes stays inside the row
    const uint32_t col = (uint32_t)ixl * 6u;
    const uint32_t o0 = __umul24((uint32_t)y0, stride) + col, o1 = __umul24((uint32_t)y1, stride) + col;
    Eq16Taps t;
    t.fix = ix != ixl;
    t.sh = (o0 & 3u) | ((o1 & 3u) << 2);
    const u32x2 adr = __builtin_amdgcn_permlane32_swap(o0 & ~3u, o1 & ~3u, false, false);
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + adr.x, 4));
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + adr.y, 4));
#pragma unroll
    for (int k = 0; k < 4; ++k) { t.a[k] = qa[k]; t.b[k] = qb[k]; }
    return t;
}
__device__ __forceinline__ void eq16_blend_rgb(const Eq16Taps& t, int sx, int sy, uint32_t (&out)[4]) {
    uint32_t r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32x2 d = __builtin_amdgcn_permlane32_swap(t.a[k], t.b[k], false, false);   // .x = row y0, .y = row y1, own pixel
        r0[k] = d.x; r1[k] = d.y;
    }
    const uint32_t s0 = t.sh & 3u, s1 = t.sh >> 2;
    // halfwords of a row: R0 G0 | B0 R1 | G1 B1
    const uint32_t d0 = __builtin_amdgcn_alignbyte(r0[1], r0[0], s0), d1 = __builtin_amdgcn_alignbyte(r0[2], r0[1], s0),
                   d2 = __builtin_amdgcn_alignbyte(r0[3], r0[2], s0);
    const uint32_t e0 = __builtin_amdgcn_alignbyte(r1[1], r1[0], s1), e1 = __builtin_amdgcn_alignbyte(r1[2], r1[1], s1),
                   e2 = __builtin_amdgcn_alignbyte(r1[3], r1[2], s1);
    const int fx = sx & 31, fy = sy & 31;
    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
    const uint32_t wr0 = ah * (uint32_t)(32 - fy), wr1 = ah * (uint32_t)fy;       // (a0 b | a1 b << 16), a1 b <= 1024
    // v_perm_b32 pairs the two taps of a channel (bytes 0..3 come from the second operand, 4..7 from the first), v_dot2_u32_u16
    // multiplies both by the packed weights: sum <= 65535 * 1024 + 512 < 2^32
    out[0] = udot2_u16(__builtin_amdgcn_perm(e1, e0, 0x07060100u), wr1, udot2_u16_from(__builtin_amdgcn_perm(d1, d0, 0x07060100u), wr0, 512u)) >> 10;
    out[1] = udot2_u16(__builtin_amdgcn_perm(e2, e0, 0x05040302u), wr1, udot2_u16_from(__builtin_amdgcn_perm(d2, d0, 0x05040302u), wr0, 512u)) >> 10;
    out[2] = udot2_u16(__builtin_amdgcn_perm(e2, e1, 0x07060100u), wr1, udot2_u16_from(__builtin_amdgcn_perm(d2, d1, 0x07060100u), wr0, 512u)) >> 10;
}
// any channel count, horizontal wrap: element by element (also the repair path of the RGB fast path)
template <int C>
__device__ __forceinline__ void eq16_sample_slow(const uint8_t* __restrict__ src, int64_t stride, int W, int H, int sx, int sy, uint32_t (&out)[4]) {
    const int fx = sx & 31, ix = sx >> 5, fy = sy & 31, iy = sy >> 5;
    const int y0 = min(max(iy, 0), H - 1), y1 = min(max(iy + 1, 0), H - 1);
    const uint8_t* r0 = src + (int64_t)y0 * stride;
    const uint8_t* r1 = src + (int64_t)y1 * stride;
    const uint32_t a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
    const int ix1 = (ix + 1 == W) ? 0 : ix + 1;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const uint32_t acc = ((uint32_t)ld_u16(r0 + 2 * (ix * C + c)) * a0 + (uint32_t)ld_u16(r0 + 2 * (ix1 * C + c)) * a1) * b0 +
                             ((uint32_t)ld_u16(r1 + 2 * (ix * C + c)) * a0 + (uint32_t)ld_u16(r1 + 2 * (ix1 * C + c)) * a1) * b1;
        out[c] = (acc + 512u) >> 10;
    }
}

// RGB bicubic: the four taps of a window row are 24 contiguous bytes -> seven dwords from the dword boundary below them.
// sum w S over 16 taps needs 34 bits; with S = S' + 32768 (S' signed: one XOR per dword) it is sum w S' + 32768 * sum w, and
// sum w = 32768 for every phase of the table (the fix-up of OpenCV's initInterTab2D; asserted in tests/test_u16.py), while
// |sum w S'| <= 32768 * sum |w| < 2^31 (sum |w| <= 1.9 * 32768): the 16 products accumulate exactly in v_dot2_i32_i16.
struct Eq16CubicTaps {
    uint32_t r[4][7];
    uint32_t sh;
    int phase;
    bool fix;
};
__device__ __forceinline__ Eq16CubicTaps eq16_cubic_issue_rgb(const uint8_t* __restrict__ src, uint32_t stride, bool stride4, int W, int H, int sx, int sy) {
    const int ix = sx >> 5, iy = sy >> 5;
    Eq16CubicTaps t;
    const int x0 = clamp0_uniform(ix - 1, W - 5);           // 28-byte aligned read of 24 tap bytes stays in-row
    t.fix = (x0 != ix - 1);
    t.phase = (sy & 31) * 32 + (sx & 31);
    const uint32_t col = (uint32_t)x0 * 6u;
    uint32_t offs[4];                                       // as in cubic_issue_rgb: 32-bit offsets from the scalar base
    if (stride4) {                                          // one misalignment (0 or 2) for all four rows
        const uint32_t o = ((uint32_t)reinterpret_cast<uintptr_t>(src) + col) & 3u;
        t.sh = o;
        const uint32_t cb = col - o;
        if (!any_lane(iy < 1 || iy > H - 3)) {              // common case: rows off0 + k * stride
            offs[0] = __umul24((uint32_t)(iy - 1), stride) + cb;
#pragma unroll
            for (int ky = 1; ky < 4; ++ky) offs[ky] = offs[ky - 1] + stride;
        } else {
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) offs[ky] = __umul24((uint32_t)min(max(iy - 1 + ky, 0), H - 1), stride) + cb;
        }
    } else {
        t.sh = 0;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const uint32_t off = __umul24((uint32_t)min(max(iy - 1 + ky, 0), H - 1), stride) + col;
            const uint32_t o = ((uint32_t)reinterpret_cast<uintptr_t>(src) + off) & 3u;
            offs[ky] = off - o;
            t.sh |= o << (2 * ky);
        }
    }
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + (size_t)offs[ky], 4));
#pragma unroll
        for (int k = 0; k < 7; ++k) t.r[ky][k] = q[k];
    }
    return t;
}
// selector of v_perm_b32(a = dword of halfword e1, b = dword of halfword e0): (h_e0 | h_e1 << 16)
#define GS360_PAIR16(e0, e1) ((uint32_t)(2 * ((e0) & 1)) | ((uint32_t)(2 * ((e0) & 1) + 1) << 8) | ((uint32_t)(4 + 2 * ((e1) & 1)) << 16) | \
                              ((uint32_t)(5 + 2 * ((e1) & 1)) << 24))
template <bool ONE_SHIFT>
__device__ __forceinline__ void eq16_cubic_rows(const Eq16CubicTaps& t, const uint32_t (&wpk)[8], int (&acc)[3]) {
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint32_t o = ONE_SHIFT ? t.sh : ((t.sh >> (2 * ky)) & 3u);
        uint32_t d[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = __builtin_amdgcn_alignbyte(t.r[ky][k + 1], t.r[ky][k], o) ^ 0x80008000u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {                     // halfword e = 3 kx + c of the row's twelve
            const uint32_t p = __builtin_amdgcn_perm(d[(3 + c) >> 1], d[c >> 1], GS360_PAIR16(c, 3 + c));
            acc[c] = ky == 0 ? dot2_i16_from(p, wpk[0], 1 << 14) : dot2_i16(p, wpk[2 * ky], acc[c]);
            acc[c] = dot2_i16(__builtin_amdgcn_perm(d[(9 + c) >> 1], d[(6 + c) >> 1], GS360_PAIR16(6 + c, 9 + c)), wpk[2 * ky + 1], acc[c]);
        }
    }
}
__device__ __forceinline__ void eq16_cubic_blend_rgb(const Eq16CubicTaps& t, const int16_t* wtab, bool stride4, uint32_t (&out)[4]) {
    uint32_t wpk[8];
    cubic_lds_weights(wtab, t.phase, wpk);
    // (sum w S' + 2^30 + 2^14) >> 15 in 32 bits: the chains start at 2^14 -- |sum w S'| + 2^14 <= 32768 * 1.9 * 32768 + 2^14 < 2^31
    // (the bound asserted on the table in tests/test_u16.py) -- and 2^30 >> 15 = 32768 is added after the shift, exactly
    int acc[3];
    if (stride4) eq16_cubic_rows<true>(t, wpk, acc);
    else eq16_cubic_rows<false>(t, wpk, acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (uint32_t)min(max((acc[c] >> 15) + 32768, 0), 65535);
}
template <int C>
__device__ __forceinline__ void eq16_cubic_slow(const EqSrc& L, const int16_t* wtab, const uint8_t* __restrict__ src, int sx, int sy, uint32_t (&out)[4]) {
    const int fx = sx & 31, fy = sy & 31, ix = sx >> 5, iy = sy >> 5;
    uint32_t wpk[8];
    cubic_lds_weights(wtab, fy * 32 + fx, wpk);
    int cols[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
        const int xx = ix - 1 + kx;
        cols[kx] = xx < 0 ? xx + L.W : (xx >= L.W ? xx - L.W : xx);
    }
    int64_t acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint8_t* row = src + (int64_t)min(max(iy - 1 + ky, 0), L.H - 1) * L.src_stride;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const uint32_t pk = wpk[(ky * 4 + kx) >> 1];
            const int w = (int)(int16_t)((kx & 1) ? (pk >> 16) : (pk & 0xffffu));
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] += (int64_t)((int)ld_u16(row + 2 * (cols[kx] * C + c)) * w);
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int64_t v = (acc[c] + (1 << 14)) >> 15;
        out[c] = (uint32_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
}

// one pass (see eq_pass) over the wavefront's four row slots, 16-bit samples, row-per-slot lane map
template <int C, bool CUBIC>
__device__ __forceinline__ void eq_pass16(const EqSrc& L, const uint8_t* __restrict__ src, uint8_t* dst, int64_t dstride,
                                          const int (&sxs)[kRowsPerWave], const int (&sys)[kRowsPerWave],
                                          const int (&ys)[kRowsPerWave], const bool (&row_ok)[kRowsPerWave],
                                          int col0, int n_px, bool reversed, bool aligned4, bool skip_first,
                                          const int16_t* wtab, const RowPack& rp) {
    uint32_t px[kRowsPerWave][4];
    const bool wide = (C == 3) && ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)L.src_stride) & 1) == 0 && L.W >= 8;
    if constexpr (C == 3) {
      if (wide) {
        if constexpr (CUBIC) {
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {      // one slot at a time: 8 row reads (28 dwords) in flight
                const Eq16CubicTaps t = eq16_cubic_issue_rgb(src, (uint32_t)L.src_stride, L.stride4, L.W, L.H, sxs[s], sys[s]);
                __builtin_amdgcn_sched_barrier(0);
                eq16_cubic_blend_rgb(t, wtab, L.stride4, px[s]);
                if (any_lane(t.fix)) {
                    if (t.fix) eq16_cubic_slow<C>(L, wtab, src, sxs[s], sys[s], px[s]);
                }
            }
        } else {
            Eq16Taps taps[kRowsPerWave];
            bool any_fix = false;
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                taps[s] = eq16_issue_rgb(src, (uint32_t)L.src_stride, L.W, L.H, sxs[s], sys[s]);
                any_fix |= taps[s].fix;
            }
            __builtin_amdgcn_sched_barrier(0);            // every gather of the pass is in flight before the first is consumed
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) eq16_blend_rgb(taps[s], sxs[s], sys[s], px[s]);
            if (any_lane(any_fix)) {
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s)
                    if (taps[s].fix) eq16_sample_slow<C>(src, L.src_stride, L.W, L.H, sxs[s], sys[s], px[s]);
            }
        }
      }
    }
    if (!wide) {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            if constexpr (CUBIC) eq16_cubic_slow<C>(L, wtab, src, sxs[s], sys[s], px[s]);
            else eq16_sample_slow<C>(src, L.src_stride, L.W, L.H, sxs[s], sys[s], px[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s)
        if (row_ok[s]) store_row16<C>(dst + (int64_t)ys[s] * dstride + (int64_t)col0 * (2 * C), px[s], n_px, aligned4, rp, reversed, skip_first);
}

// equirect -> rectilinear views.  The pinhole grid is mirror-symmetric about the view's vertical axis:
// x(w-1-i) = -x(i) exactly, so latitude (even in x) is shared by the pixel pair (i, w-1-i) and longitude only
// changes sign before the final fma/rint.  Level views (pitch 0) are also symmetric about the horizon:
// yv(h-1-j) = -yv(j) exactly, latitude flips sign (rint is odd) -> one atan2 serves four pixels, and the
// longitude term depends on the column only.  All of this is bit-identical to evaluating EQ-SPEC v1 per pixel.
// Occupancy is pinned (kEqWaves wavefronts per SIMD): with more resident wavefronts their gathers evict each
// other's lines from the 32 KiB vector L1, with fewer the miss queue runs dry.  Measured on cfg2, us per frame --
// row-per-slot lane map: 3 / 4 / 6 wavefronts: 27.4 / 23.1 / 24.3; blocked lane map: 3 / 4 / 5 / 6: 22.1 / 20.8 / 20.3 /
// 23.3 (6 spills).  5 is also the better choice for the arithmetic-bound large-view configs (cfg1/3/5; re-measured on the
// round-3 kernel, 88 registers: 6 wavefronts = 80 registers + 6 spills run cfg1/2/3/5 6-10 % slower).  The cubic
// variant needs 128 VGPRs and stays at 4.
// LDS of a workgroup besides the cubic weight table: the blocked store's transpose slices (256 dwords per wavefront; its read-back
// may touch the dword after the slice, which is the next slice or the first parked dword -- never used), then the parked ring
// coordinates (see the member loop).  Every thread / wavefront only ever touches its own part: no barrier.
template <int C, bool CUBIC, int ES, bool ROWS = false, bool MASKED = false>     // ROWS: instantiation for launches without blocked views
struct EqLds {
    static constexpr bool kBlocked = !ROWS && (C == 3) && (ES == 1) && (kRowsPerWave == 4) && (kWaves == 4);   // blocked lane map available
    static constexpr int kParkN = (CUBIC && ES == 2) ? 3 : (CUBIC ? kRingParkCubic : kRingPark);     // 0 none, 1 latitude only, 3 all three
    static constexpr int kBlkDw = kBlocked ? kWaves * 256 : 0;
    static constexpr int kParkDw = kParkN * kRowsPerWave * 64 * kWaves;
    // the pipelined bilinear member loop keeps ALL ring-shared coordinates in LDS (three 16-byte entries per thread)
    // lean member loop (bilinear RGB, row-per-slot map): six int4 entries per thread -- latitude, left / mirrored longitude, the two
    // tap rows' byte offsets and the vertical phase of the current pitch sign
    static constexpr bool kLean = kEqLean && C == 3 && !CUBIC && ES == 1 && kRowsPerWave == 4;
    static constexpr int kPipeDw = kLean ? (MASKED ? 7 : 6) * kRowsPerWave * 64 * kWaves : 0;   // masked: + the keep-bit image's row offsets
    static constexpr int kDwords = kBlkDw + (kParkDw + kPipeDw ? kParkDw + kPipeDw : 4);
};

// One tile of a ring (all its members, both halves) -- the body of eq_views_kernel.  `b` is the workgroup's position in the
// tile order (blockIdx.x, or the persistent walk of the cubic variants).
template <int C, bool CUBIC, bool MASKED, int ES, bool ROWS = false>
__device__ __forceinline__ void eq_views_tile(const EqLaunch& L, const int b, const int16_t* s_wtab, uint32_t* const s_lds) {
    // XCD-aware tile order: XCD x (= b % 8) walks tiles [x*chunk, (x+1)*chunk), or -- when the launch mixes rings of
    // different sizes, whose tiles differ in cost -- runs of 2^g consecutive tiles dealt round-robin to the XCDs
    int t = (b & 7) * L.chunk + (b >> 3);
    if (L.xcd_group_log2 >= 0) {
        const int q = b >> 3, g = L.xcd_group_log2;
        t = ((((q >> g) << 3) + (b & 7)) << g) + (q & ((1 << g) - 1));
    }
    if (t >= L.total_tiles) return;
    constexpr bool kBlocked = EqLds<C, CUBIC, ES, ROWS, MASKED>::kBlocked;
    constexpr int kParkN = EqLds<C, CUBIC, ES, ROWS, MASKED>::kParkN;
    constexpr int kBlkDw = EqLds<C, CUBIC, ES, ROWS, MASKED>::kBlkDw;
    uint32_t* const s_blk = s_lds;
    int* const s_park = reinterpret_cast<int*>(s_lds + kBlkDw);
    int f = t / L.tiles_per_frame;
    int r = t - f * L.tiles_per_frame;
    int g = 0;
    while (g + 1 < L.n_rings && r >= L.view[L.ring_first[g + 1]].tile_base) ++g;
    const int k0 = L.ring_first[g], n_members = L.ring_count[g];
    const EqView& V = L.view[k0];          // the ring's geometry (every member has the same, up to x0i32 and the pitch sign)
    r -= V.tile_base;
    const int tile_y = r / V.tiles_x, tile_x = r - tile_y * V.tiles_x;

    // the wavefront index as a scalar: everything derived from it (row slots, patch origins) then lives in SGPRs
    // (the lane index behind an optimisation barrier in the persistent variants: nothing derived from it is hoisted out of the walk)
    const int lane = CUBIC ? lane_here() : (int)(threadIdx.x & 63), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half_w = (V.out_w + 1) >> 1;               // columns [0, half_w) are computed, the rest mirrored
    const int x0 = tile_x * kTileW;
    const int n_px = min(kTileW, half_w - x0);
    const int xl = min(x0 + lane, half_w - 1);            // lanes past the edge recompute the last column (they stay
                                                          // active for the store shuffles)
    const uint8_t* __restrict__ src = L.src[f];
    const uint8_t* __restrict__ mask = L.mask[f];
    const int64_t dstride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * C * ES;
    const float x = (float)(2 * xl + 1 - V.out_w) * V.sxu;

    // ---- row slots of this wavefront -------------------------------------------------------------
    int ys[kRowsPerWave];
    bool row_ok[kRowsPerWave];
    const bool level = V.level != 0;                      // wave-uniform
    if (level) {
        const int top_h = (V.out_h + 1) >> 1;
#pragma unroll
        for (int s = 0; s < kHalfRows; ++s) {
            const int y = tile_y * (kTileH / 2) + wave * kHalfRows + s;
            ys[s] = min(y, top_h - 1);
            ys[s + kHalfRows] = V.out_h - 1 - ys[s];
            row_ok[s] = y < top_h;
            row_ok[s + kHalfRows] = row_ok[s] && (ys[s + kHalfRows] != ys[s]);
        }
    } else {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const int y = tile_y * kTileH + wave * kRowsPerWave + s;
            ys[s] = min(y, V.out_h - 1);
            row_ok[s] = y < V.out_h;
        }
    }

    // ---- coordinates (EQ-SPEC v1) ------------------------------------------------------------------
    int sxl[kRowsPerWave], sxm[kRowsPerWave], sys[kRowsPerWave];
    const bool blocked = kBlocked && V.blocked != 0;      // wave-uniform, chosen per view on the host
    // A wavefront of a level view's blocked tile owns 32 columns; in the last tile of a row of tiles (400 = 6 x 64 + 16
    // for an 800-pixel view) the upper half has nothing to do.  No workgroup-level barrier follows, so it can leave.
    if (level && blocked && (wave & 1) * 32 >= n_px) return;
    if (level && blocked) {
        // level view, blocked lane map: the wavefront owns 4 top rows x 32 columns (slots 0,1 = its two 16-column
        // groups) and their horizon mirrors (slots 2,3).  Longitude is per column, latitude per (column, row) and
        // shared with the mirrored row with its sign flipped, as in the row-per-slot form below.
        const int top_h = (V.out_h + 1) >> 1;
        const int yt = min(tile_y * (kTileH / 2) + (wave >> 1) * 4 + (lane >> 4), top_h - 1);
        const float yv = (float)(2 * yt + 1 - V.out_h) * V.syv;
        const float cy = __builtin_fmaf(-V.cp, yv, V.sp);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int xs = min(x0 + (wave & 1) * 32 + 16 * s + (lane & 15), half_w - 1);
            const float xx = (float)(2 * xs + 1 - V.out_w) * V.sxu;
            int Kl, Kt;
            const float rl = eq_atan2_red(xx, 1.0f, Kl);
            const float h = eq_sqrt(__builtin_fmaf(xx, xx, 1.0f));
            const float rt = eq_atan2_red<true>(cy, h, Kt);
            const int q = Kt * 8 * L.H + (int)__builtin_rintf(rt * L.ky32);
            sys[s] = L.y0i32 - q;
            sys[s + 2] = L.y0i32 + q;
            sxl[s] = sxl[s + 2] = eq_lon_base(rl, Kl, L, V.x0f32);
            sxm[s] = sxm[s + 2] = eq_lon_base(-rl, -Kl, L, V.x0f32);
        }
    } else if (level) {
        int Kl;
        const float rl = eq_atan2_red(x, 1.0f, Kl);       // b = fma(0, yv, 1) = 1 for every row
        const int sx_left = eq_lon_base(rl, Kl, L, V.x0f32), sx_mirror = eq_lon_base(-rl, -Kl, L, V.x0f32);
        const float h = eq_sqrt(__builtin_fmaf(x, x, 1.0f));
#pragma unroll
        for (int s = 0; s < kHalfRows; ++s) {
            const float yv = (float)(2 * ys[s] + 1 - V.out_h) * V.syv;
            const float cy = __builtin_fmaf(-V.cp, yv, V.sp);
            int Kt;
            const float rt = eq_atan2_red<true>(cy, h, Kt);
            const int q = Kt * 8 * L.H + (int)__builtin_rintf(rt * L.ky32);
            sys[s] = L.y0i32 - q;
            sys[s + kHalfRows] = L.y0i32 + q;
            sxl[s] = sxl[s + kHalfRows] = sx_left;
            sxm[s] = sxm[s + kHalfRows] = sx_mirror;
        }
    } else {
        if (blocked) {
        // Blocked lane map: slot s is a 16-column group and lane>>4 the row, so every gather instruction covers a
        // compact 4-row x 16-column patch of the view instead of 64 pixels of one row.  Where the view bends across
        // source rows this halves the cache lines an instruction touches (CPU model of a cfg2 view: 0.78 instead of
        // 1.41 lines per pixel) -- the vector L1's tag pipeline is the limiter of this kernel.
        const int yl = min(tile_y * kTileH + wave * kRowsPerWave + (lane >> 4), V.out_h - 1);
        const float yvl = (float)(2 * yl + 1 - V.out_h) * V.syv;
        const float bzl = __builtin_fmaf(V.sp, yvl, V.cp);
        const float cyl = __builtin_fmaf(-V.cp, yvl, V.sp);
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const int xs = min(x0 + 16 * s + (lane & 15), half_w - 1);
            const float xx = (float)(2 * xs + 1 - V.out_w) * V.sxu;
            const float h = eq_sqrt(__builtin_fmaf(xx, xx, bzl * bzl));
            int Kl, Kt;
            const float rl = eq_atan2_red(xx, bzl, Kl);
            const float rt = eq_atan2_red<true>(cyl, h, Kt);
            sxl[s] = eq_lon_base(rl, Kl, L, V.x0f32);
            sxm[s] = eq_lon_base(-rl, -Kl, L, V.x0f32);
            sys[s] = L.y0i32 - Kt * 8 * L.H - (int)__builtin_rintf(rt * L.ky32);
        }
        } else {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            float yv = (float)(2 * ys[s] + 1 - V.out_h) * V.syv;
            float xr = x, bz, cy;
            if (V.fish) {      // wave-uniform: equidistant-fisheye output (EQ-SPEC v1 projection F), ray = (u S, v S, C)
                const float q = __builtin_fmaf(x, x, yv * yv);
                const float S = eq_poly8(kEqFishS, q), Cz = eq_poly8(kEqFishC, q);
                xr = x * S;
                yv = yv * S;
                bz = __builtin_fmaf(V.sp, yv, V.cp * Cz);
                cy = __builtin_fmaf(-V.cp, yv, V.sp * Cz);
            } else {
                bz = __builtin_fmaf(V.sp, yv, V.cp);    // forward component after pitch
                cy = __builtin_fmaf(-V.cp, yv, V.sp);   // up component after pitch
            }
            const float h = eq_sqrt(__builtin_fmaf(xr, xr, bz * bz));
            int Kl, Kt;
            const float rl = eq_atan2_red(xr, bz, Kl);
            const float rt = eq_atan2_red<true>(cy, h, Kt);
            sxl[s] = eq_lon_base(rl, Kl, L, V.x0f32);
            sxm[s] = eq_lon_base(-rl, -Kl, L, V.x0f32);
            sys[s] = L.y0i32 - Kt * 8 * L.H - (int)__builtin_rintf(rt * L.ky32);
        }
        }
    }

    // ---- every member of the ring: left half, then the mirrored half ------------------------------------
    // sxl / sxm hold the ring-shared longitude part (eq_lon_base); a member adds its integer offset and wraps.  A member
    // with the opposite pitch sign is the ring's geometry upside down: with sp' = -sp and yv' = -yv (row h-1-j) the forward
    // component fma(sp, yv, cp) is unchanged and the up component fma(-cp, yv, sp) changes sign, both exactly, so longitude is
    // the same and latitude is negated (q -> -q, rint is odd) -- the same argument as the horizon mirror of level views.
    BlkStore bs;
    bs.lds = nullptr; bs.mode = 0; bs.y0 = bs.nrows = bs.yb = bs.nrows_b = bs.sub = 0; bs.ystep = 1;
    if (blocked) {
        bs.lds = s_blk + wave * 256;
        if (level) {
            const int top_h = (V.out_h + 1) >> 1;
            bs.mode = 2;
            bs.y0 = tile_y * (kTileH / 2) + (wave >> 1) * 4;
            bs.nrows = min(max(top_h - bs.y0, 0), 4);
            bs.yb = V.out_h - 1 - bs.y0;
            // with an odd height the last top row is the horizon row itself: it has no mirror
            bs.nrows_b = bs.nrows - (((V.out_h & 1) && bs.nrows > 0 && bs.y0 + bs.nrows == top_h) ? 1 : 0);
            bs.sub = (wave & 1) * 32;
        } else {
            bs.mode = 1;
            bs.y0 = tile_y * kTileH + wave * kRowsPerWave;
            bs.nrows = min(max(V.out_h - bs.y0, 0), 4);
        }
    }
    const int W32 = uniform_here(32 * L.W);
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s) {              // shared longitude parts into [0, 32W): members then wrap with one min
        sxl[s] = eq_lon_norm(sxl[s], W32);
        sxm[s] = eq_lon_norm(sxm[s], W32);
    }
    const bool centre_dup = (V.out_w & 1) && (x0 + n_px == half_w);
    const RowPack rp = make_row_pack(lane);
    // The 12 shared coordinates of a lane would stay live across the whole member loop on top of the sampler's own peak
    // (bilinear: 123 registers against the 96 of 5 wavefronts per SIMD).  Each thread parks its own values in LDS and takes
    // them back at the top of every iteration -- no barrier, a thread only ever reads what it wrote.
    if constexpr (kParkN > 0) {
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            s_park[(kParkN * s + 0) * 64 * kWaves + threadIdx.x] = sys[s];
            if constexpr (kParkN == 3) {
                s_park[(3 * s + 1) * 64 * kWaves + threadIdx.x] = sxl[s];
                s_park[(3 * s + 2) * 64 * kWaves + threadIdx.x] = sxm[s];
            }
        }
    }
    // Everything the loop body reads from the kernel argument is taken into registers HERE, behind an optimisation barrier: left
    // alone, the compiler re-reads kernel-argument fields inside the loop whenever scalar registers run short (each read is a
    // scalar-cache round trip the wavefront waits for: five per iteration measured, +20 % time at equal instruction counts).
    EqSrc S;
    S.W = uniform_here(L.W);
    S.H = uniform_here(L.H);
    S.src_stride = (int64_t)(uint32_t)uniform_here((int)L.src_stride);          // < 2^24 (checked by the host)
    S.mask_stride = (int64_t)(uint32_t)uniform_here((int)L.mask_stride);        // H * mask_stride < 2^32
    S.stride4 = uniform_here((int)(L.src_stride & 3)) == 0;
    const int out_h = uniform_here(V.out_h), out_w = uniform_here(V.out_w);
    const int y0x2 = uniform_here(2 * L.y0i32);
    const int dst_base = uniform_here(f * L.n_views + k0);
    auto members = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    if constexpr (EqLds<C, CUBIC, ES, ROWS, MASKED>::kLean && MODE == 0) {
        // The preset-shaped views are bound by the vector ALU, not by memory (probes: tap reads folded into two cache lines and stores
        // dropped leave 79 / 64 / 81 % of the cfg1 / cfg3 / cfg5 time), so this loop does per pass only what changes per pass:
        //  * everything a ring shares waits in LDS (each thread's own int4 entries, no barrier): latitude, the left / mirrored
        //    longitude bases, and -- derived once per pitch sign -- the byte offsets of the two tap rows and the vertical phase;
        //    a pass adds the member's integer longitude offset, wraps, and adds column offsets to the parked row offsets;
        //  * the horizontal phase is the base's (x0i32 and 32 W are multiples of 32);
        //  * v_alignbyte reads only the low two bits of its shift operand: the tap offsets are passed as they are;
        //  * one turn per pass, software-pipelined: the NEXT pass's gathers are issued before the current pass's pixels are
        //    shuffled and stored, the last pass is resolved behind the loop (a conditional issue would keep consumed tap
        //    registers alive: copies in front of every use).
        // Members with the opposite pitch sign re-derive the latitude entries once (the host sorts them behind the others).
        // The mirrored pass is fetched and blended even for the one-column centre tile of an odd-width view; only its store is dropped.
        int4* const park = reinterpret_cast<int4*>(s_park) + threadIdx.x;    // [k][256]
        constexpr int kP = 64 * kWaves;
        const uint32_t stride = (uint32_t)S.src_stride;
        const int fix_from = uniform_here(32 * (S.W - 4));                   // ix > W - 5: the 12-byte read would leave the row
        bool flipstate = false;
        auto derive_lat = [&](const int4 lat, const bool flip) {             // row offsets and vertical phase for one pitch sign
            const int la[4] = {lat.x, lat.y, lat.z, lat.w};
            int r0[4], r1[4], fy[4];
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                const int sy = flip ? y0x2 - la[s] : la[s];
                const int iy = sy >> 5;                                      // in [-1, H - 1] (|lat| <= pi/2 by construction)
                r0[s] = (int)__umul24((uint32_t)max(iy, 0), stride);
                r1[s] = (int)__umul24((uint32_t)min(iy + 1, S.H - 1), stride);
                fy[s] = sy & 31;
            }
            park[3 * kP] = make_int4(r0[0], r0[1], r0[2], r0[3]);
            park[4 * kP] = make_int4(r1[0], r1[1], r1[2], r1[3]);
            park[5 * kP] = make_int4(fy[0], fy[1], fy[2], fy[3]);
            if constexpr (MASKED) {                       // byte offset of the pixel's row in the keep-bit image
                int mr[4];
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s)
                    mr[s] = (int)__umul24((uint32_t)((flip ? y0x2 - la[s] : la[s]) + 16) >> 5, (uint32_t)S.mask_stride);
                park[6 * kP] = make_int4(mr[0], mr[1], mr[2], mr[3]);
            }
        };
        park[0 * kP] = make_int4(sys[0], sys[1], sys[2], sys[3]);
        park[1 * kP] = make_int4(sxl[0], sxl[1], sxl[2], sxl[3]);
        park[2 * kP] = make_int4(sxm[0], sxm[1], sxm[2], sxm[3]);
        int2 mem = *reinterpret_cast<const int2*>(&L.view[k0].x0i32);
        uint8_t* dst = L.dst[dst_base];
        flipstate = mem.y != 0;
        derive_lat(make_int4(sys[0], sys[1], sys[2], sys[3]), flipstate);
        const bool has_mirror = n_px > (centre_dup ? 1 : 0);
        const int col0_m = out_w - x0 - n_px;
        // the pass in flight: raw tap dwords (fetch order), tap byte offsets (low two bits = misalignment), longitude coordinate
        uint32_t ra[kRowsPerWave][3], rb[kRowsPerWave][3], o0[kRowsPerWave], o1[kRowsPerWave], keep[kRowsPerWave], kbit[kRowsPerWave];
        int cx[kRowsPerWave];
        // (the parked entries a pass needs are read one step ahead by the caller: an LDS round trip in front of the gathers would
        // delay every one of them)
        auto issue = [&](const int4 lon, const int4 r0q, const int4 r1q, const int4 mrq, const int x0i) {
            const int lo[4] = {lon.x, lon.y, lon.z, lon.w};
            const int r0[4] = {r0q.x, r0q.y, r0q.z, r0q.w}, r1[4] = {r1q.x, r1q.y, r1q.z, r1q.w};
            if constexpr (MASKED) {                       // keep bits: one aligned dword of the bit image per pixel (see eq_mask_at)
                const int mr[4] = {mrq.x, mrq.y, mrq.z, mrq.w};
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s) {
                    const uint32_t t = (uint32_t)eq_lon_member(lo[s], x0i, W32) + 16u;
                    kbit[s] = t >> 5;                     // nearest column; its low five bits pick the bit (v_bfe reads only those)
                    keep[s] = *reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(mask + (size_t)((uint32_t)mr[s] + ((t >> 8) & ~3u)), 4));
                }
            }
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                cx[s] = eq_lon_member(lo[s], x0i, W32);
                const uint32_t col = (uint32_t)min(cx[s] >> 5, S.W - 5) * 3u;
                o0[s] = (uint32_t)r0[s] + col;
                o1[s] = (uint32_t)r1[s] + col;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + (o0[s] & ~3u), 4));
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + (o1[s] & ~3u), 4));
                ra[s][0] = qa[0]; ra[s][1] = qa[1]; ra[s][2] = qa[2];
                rb[s][0] = qb[0]; rb[s][1] = qb[1]; rb[s][2] = qb[2];
            }
        };
        auto resolve = [&](uint32_t (&pk)[kRowsPerWave], const bool flip, const int4 fyq) {
            const int fys[4] = {fyq.x, fyq.y, fyq.z, fyq.w};
            uint32_t px[kRowsPerWave][4];
            bool any_fix = false;
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                EqTaps<3> t;
                t.t0.x = __builtin_amdgcn_alignbyte(ra[s][1], ra[s][0], GS360_AB(o0[s]));
                t.t0.y = __builtin_amdgcn_alignbyte(ra[s][2], ra[s][1], GS360_AB(o0[s]));
                t.t1.x = __builtin_amdgcn_alignbyte(rb[s][1], rb[s][0], GS360_AB(o1[s]));
                t.t1.y = __builtin_amdgcn_alignbyte(rb[s][2], rb[s][1], GS360_AB(o1[s]));
                eq_blend_f<3>(t, cx[s] & 31, fys[s], px[s]);
                any_fix |= cx[s] >= fix_from;
            }
            if (any_lane(any_fix)) {
                const int4 lat = park[0 * kP];
                const int la[4] = {lat.x, lat.y, lat.z, lat.w};
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s)
                    if (cx[s] >= fix_from) eq_sample_slow<3>(src, S.src_stride, S.W, S.H, cx[s], flip ? y0x2 - la[s] : la[s], px[s]);
            }
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                pk[s] = px[s][0] | (px[s][1] << 8) | (px[s][2] << 16);
                if constexpr (MASKED) pk[s] &= (uint32_t)__builtin_amdgcn_sbfe((int)keep[s], kbit[s] & 31u, 1u);   // 0 or all ones
            }
        };
        const int n_bytes = 3 * n_px, full = n_bytes >> 2, rem = n_bytes & 3;
        const bool ofs32_ok = (uint64_t)out_h * (uint64_t)dstride < 0xffffffffull;      // row offsets of the shifted store path are 32-bit
        auto store_pass = [&](const uint32_t (&pk)[kRowsPerWave], uint8_t* const d, const bool flip, const bool mirror) {
            if (mirror && !has_mirror) return;
            const bool aligned = ((dstride & 3) == 0) && ((reinterpret_cast<uintptr_t>(d) & 3) == 0) &&
                                 (!mirror || ((((col0_m * 3) & 3) == 0) && !centre_dup));
            uint8_t* const d0 = d + (int64_t)(mirror ? col0_m : x0) * 3;
            // the lane-derived constants are made here, behind an optimisation barrier, once per pass (hoisted out of the loop they
            // would be spilled, and a reload from scratch memory waits for every gather in flight)
            int lane = rp.lane, a4 = rp.a4;
            asm volatile("" : "+v"(lane), "+v"(a4));
            if (aligned) {                                // whole-dword rows: eight shuffles in flight together, then the stores
                const int la4 = mirror ? 4 * (n_px - 1) - a4 : a4, lb4 = mirror ? la4 - 4 : la4 + 4;
                uint32_t dw[kRowsPerWave];
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s) {
                    const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(la4 & 252, (int)pk[s]);
                    const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(lb4 & 252, (int)pk[s]);
                    dw[s] = __builtin_amdgcn_perm(pb, pa, rp.sel);
                }
                const uint32_t off = (uint32_t)lane << 2;
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s) {
                    if (!row_ok[s]) continue;
                    uint8_t* const row = d0 + (int64_t)(flip ? out_h - 1 - ys[s] : ys[s]) * dstride;
                    if (lane < full) __builtin_nontemporal_store(dw[s], reinterpret_cast<uint32_t*>(row + (size_t)off));
                    if (rem && lane == full)
                        for (int k = 0; k < rem; ++k) row[4 * full + k] = (uint8_t)(dw[s] >> (8 * k));
                }
                return;
            }
            if (kShiftedStore && !(mirror && centre_dup) && ofs32_ok) {
                // Row segments that start off a dword boundary (widths that are not multiples of four: 5250-byte rows of a 1750-pixel
                // view): the same two shuffles per slot, with the byte stream of the segment re-sliced at the row's own misalignment.
                // Lanes 0..47 write the aligned dwords inside the segment, lanes 48..50 its 0-3 head bytes, lanes 52..54 its 0-3 tail
                // bytes -- one dword store and one byte store per slot instead of three byte stores per pixel (+61 % per frame).
                uint32_t dwv[kRowsPerWave], adr[kRowsPerWave];
                bool as_dword[kRowsPerWave], as_byte[kRowsPerWave];
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s) {
                    const uint32_t rofs = (uint32_t)(flip ? out_h - 1 - ys[s] : ys[s]) * (uint32_t)dstride;          // (ofs32_ok)
                    const int dh = (int)((0u - ((uint32_t)reinterpret_cast<uintptr_t>(d0) + rofs)) & 3u);             // head bytes of this row's segment
                    const int nf = (n_bytes - dh) >> 2, tl = (n_bytes - dh) & 3;
                    const int k = lane & 3;
                    const int sj = lane < 48 ? 4 * lane + dh : (lane < 52 ? k : dh + 4 * nf + k);                   // first stream byte of this lane's piece
                    as_dword[s] = lane < nf;
                    as_byte[s] = (lane >= 48 && lane < 52 && k < dh) || (lane >= 52 && lane < 56 && k < tl);
                    const int a = (sj * 21846) >> 16, b = sj - 3 * a;                                               // pixel, byte in it
                    const int qa = min(a, n_px - 1), qb = min(a + 1, n_px - 1);
                    const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (mirror ? n_px - 1 - qa : qa), (int)pk[s]);
                    const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (mirror ? n_px - 1 - qb : qb), (int)pk[s]);
                    dwv[s] = __builtin_amdgcn_perm(pb, pa, b == 0 ? 0x04020100u : (b == 1 ? 0x05040201u : 0x06050402u));
                    adr[s] = rofs + (uint32_t)sj;
                }
#pragma unroll
                for (int s = 0; s < kRowsPerWave; ++s) {
                    if (!row_ok[s]) continue;
                    if (as_dword[s]) __builtin_nontemporal_store(dwv[s], reinterpret_cast<uint32_t*>(__builtin_assume_aligned(d0 + (size_t)adr[s], 4)));
                    else if (as_byte[s]) d0[(size_t)adr[s]] = (uint8_t)dwv[s];
                }
                return;
            }
            const int pos = mirror ? n_px - 1 - lane : lane;
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                if (!row_ok[s]) continue;
                uint8_t* const row = d0 + (int64_t)(flip ? out_h - 1 - ys[s] : ys[s]) * dstride;
                if (lane < n_px && !(mirror && centre_dup && pos == 0))
                    for (int c = 0; c < 3; ++c) row[(size_t)(uint32_t)(pos * 3 + c)] = (uint8_t)(pk[s] >> (8 * c));
            }
        };
        const int4 mr0 = MASKED ? park[6 * kP] : make_int4(0, 0, 0, 0);
        issue(make_int4(sxl[0], sxl[1], sxl[2], sxl[3]), park[3 * kP], park[4 * kP], mr0, mem.x);
        const int p_last = 2 * n_members - 1;
        int2 mem_nx = mem;
        uint8_t* dst_nx = dst;
        uint32_t pk[kRowsPerWave];
#pragma unroll 1
        for (int p = 0; p < p_last; ++p) {
            const bool mirror = (p & 1) != 0;             // wave-uniform: the pass in flight is a mirrored half
            const bool flip = flipstate;                  // ... and its pitch sign
            uint8_t* const dst_cur = dst;
            if (!mirror) {                                // the next member's scalars, one pass ahead of their use
                const int mn = min((p >> 1) + 1, n_members - 1);
                mem_nx = *reinterpret_cast<const int2*>(&L.view[k0 + mn].x0i32);
                dst_nx = L.dst[dst_base + mn];
            }
            const int4 fyq = park[5 * kP];                // this pass's vertical phases, before a change of sign rewrites them
            if (mirror) {                                 // next: left half of the next member
                mem = mem_nx;
                dst = dst_nx;
                if ((mem.y != 0) != flipstate) {
                    flipstate = !flipstate;
                    derive_lat(park[0 * kP], flipstate);
                }
            }
            // the next pass's entries come back from LDS while this pass is blended
            const int4 lonq = park[(mirror ? 1 : 2) * kP], r0q = park[3 * kP], r1q = park[4 * kP];
            const int4 mrq = MASKED ? park[6 * kP] : make_int4(0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            resolve(pk, flip, fyq);
            __builtin_amdgcn_sched_barrier(0);
            issue(lonq, r0q, r1q, mrq, mem.x);
            __builtin_amdgcn_sched_barrier(0);
            store_pass(pk, dst_cur, flip, mirror);
        }
        __builtin_amdgcn_sched_barrier(0);
        resolve(pk, flipstate, park[5 * kP]);             // the last member's mirrored half
        store_pass(pk, dst, flipstate, true);
        return;
    }
    // the member's own scalars (x0i32, flip: one 8-byte read; dst) are fetched one iteration ahead
    int2 mem_next = *reinterpret_cast<const int2*>(&L.view[k0].x0i32);
    uint8_t* dst_next = L.dst[dst_base];
#pragma unroll 1
    for (int m = 0; m < n_members; ++m) {
        if constexpr (kParkN > 0) {
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                sys[s] = s_park[(kParkN * s + 0) * 64 * kWaves + tid];
                if constexpr (kParkN == 3) {
                    sxl[s] = s_park[(3 * s + 1) * 64 * kWaves + tid];
                    sxm[s] = s_park[(3 * s + 2) * 64 * kWaves + tid];
                }
            }
        }
        const int x0i = mem_next.x;
        const bool flip = mem_next.y != 0;                // wave-uniform
        uint8_t* dst = dst_next;
        {
            const int mn = min(m + 1, n_members - 1);
            mem_next = *reinterpret_cast<const int2*>(&L.view[k0 + mn].x0i32);
            dst_next = L.dst[dst_base + mn];
        }
        const bool base_aligned = ((dstride & 3) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0);
        int sx_l[kRowsPerWave], sx_m[kRowsPerWave], sy_m[kRowsPerWave], ys_m[kRowsPerWave];
        // latitude of a flipped member, y0x2 - sys, and of the others, sys, as ONE multiply-add: sys * (+-1) + (y0x2 | 0)
        const int y_sign = flip ? -1 : 1;
        int y_off = flip ? y0x2 : 0;
        asm volatile("" : "+v"(y_off));                   // (a vector register: the instruction takes one scalar operand)
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            sx_l[s] = eq_lon_member(sxl[s], x0i, W32);
            sx_m[s] = eq_lon_member(sxm[s], x0i, W32);
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(sy_m[s]) : "v"(sys[s]), "s"(y_sign), "v"(y_off));    // |sys| < 2^23 (32 H < 2^21)
            ys_m[s] = flip ? out_h - 1 - ys[s] : ys[s];
        }
        BlkStore bm = bs;
        if (flip) { bm.y0 = out_h - 1 - bs.y0; bm.ystep = -1; }
        if constexpr (ES == 2) eq_pass16<C, CUBIC>(S, src, dst, dstride, sx_l, sy_m, ys_m, row_ok, x0, n_px, false, base_aligned && ((x0 * C * 2) & 3) == 0, false, s_wtab, rp);
        else eq_pass<C, CUBIC, MODE, MASKED>(S, src, mask, dst, dstride, sx_l, sy_m, ys_m, row_ok, x0, n_px, false, base_aligned, false, s_wtab, bm, rp);
        // mirrored segment: columns [w - x0 - n_px, w - x0), lane l holds column w-1-x0-l.  With an odd width the
        // centre column is its own mirror and was already written: drop it from the segment.
        if (n_px > (centre_dup ? 1 : 0)) {
            const int col0 = out_w - x0 - n_px;
            const bool m_aligned = base_aligned && (((col0 * C * ES) & 3) == 0) && !centre_dup;
            if constexpr (ES == 2) eq_pass16<C, CUBIC>(S, src, dst, dstride, sx_m, sy_m, ys_m, row_ok, col0, n_px, true, m_aligned, centre_dup, s_wtab, rp);
            else eq_pass<C, CUBIC, MODE, MASKED>(S, src, mask, dst, dstride, sx_m, sy_m, ys_m, row_ok, col0, n_px, true, m_aligned, centre_dup, s_wtab, bm, rp);
        }
    }
    };
    if constexpr (kBlocked) {
        if (blocked) {
            if (level) members(std::integral_constant<int, 2>{});
            else members(std::integral_constant<int, 1>{});
            return;
        }
    }
    members(std::integral_constant<int, 0>{});
}

// The cubic variants keep a 32 KiB LDS copy of the weight table (left in global memory it would occupy the whole vector L1,
// every lane reading another 32-byte entry), filled once per workgroup.  Their workgroups CAN be persistent -- a capped grid whose
// workgroups walk positions b, b + gridDim.x, ... of the tile order (the stride is a multiple of 8: a workgroup stays on its
// XCD's tiles) -- but the C ABI leaves that off (`persist_blocks` = 0 -> grid_total == gridDim.x, one turn of the loop): measured
// slower than one tile per workgroup, see equirect_views_impl.
// ROWS: the bilinear RGB instantiation for launches in which no view uses the blocked lane map (every preset-shaped view): without
// the two blocked loop bodies the pipelined member loop alone sets the register budget (65-73 registers), so its occupancy can be
// chosen on its own (kEqRowsWaves).
template <int C, bool CUBIC, bool ES2, bool ROWS>
constexpr int eq_kernel_waves() { return (CUBIC || ES2) ? kEqCubicWaves : (ROWS ? kEqRowsWaves : kEqWaves); }
template <int C, bool CUBIC, bool MASKED, int ES = 1, bool ROWS = false>      // ES: bytes per sample (1: uint8, 2: uint16 -- row-per-slot lane map, no mask)
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(eq_kernel_waves<C, CUBIC, ES == 2, ROWS>(), eq_kernel_waves<C, CUBIC, ES == 2, ROWS>()))) void eq_views_kernel(const EqLaunch L) {
    __shared__ __attribute__((aligned(16))) int16_t s_wtab[CUBIC ? 32 * 32 * 16 : 8];
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[EqLds<C, CUBIC, ES, ROWS, MASKED>::kDwords];
    if constexpr (CUBIC) {
        cubic_lds_fill(s_wtab, L.cubic_tab, 64 * kWaves);
        __syncthreads();
#pragma unroll 1
        for (int b = blockIdx.x; b < L.grid_total; b += gridDim.x) eq_views_tile<C, CUBIC, MASKED, ES, ROWS>(L, b, s_wtab, s_lds);
    } else {
        eq_views_tile<C, CUBIC, MASKED, ES, ROWS>(L, blockIdx.x, s_wtab, s_lds);
    }
}

// ------------------------------------------------------------------------------------------------
// eq_staged_kernel -- bilinear RGB, LDS-staged source texels (launches whose views are all mildly minified: the presets)
// ------------------------------------------------------------------------------------------------
// Why: on the preset shapes the gather form is bound by the texture-address path, not by HBM (profiles/r04: TA busy 87 % of the cfg3
// launch; half the gathers = -14 / -18 / -9 % time on cfg1 / cfg3 / cfg5).  A 64-lane dwordx3 gather costs ~17 cycles of that path
// for 8 useful bytes per lane.  Here a wavefront owns a 16 x 16-pixel tile (lane = column + 16 * (row & 3), slot = row >> 2; the
// workgroup's four wavefronts sit side by side, so the workgroup tile is the other kernels' 64 x 16) whose source footprint is a
// compact box whatever the view's orientation.  Per pass (ring member x half) the box -- rows [iy_lo, iy_hi], bytes
// [xa, xa + pitch) of each, xa 16-byte aligned -- is copied into the wavefront's LDS slice by LDS-DMA loads (16 contiguous bytes
// per lane, every byte useful), and every pixel then reads its two 12-byte tap windows from LDS.  The pipeline per turn: blend
// pass p (taps in registers) -> wait for pass p+1's box, read its taps from LDS -> start the DMA of pass p+2's box -> repack and
// store pass p; the DMA flies under a whole blend + store.  No workgroup barrier: a wavefront only touches its own slice.
// A pass whose box does not qualify -- it crosses the 360-degree seam, touches the first / last two rows (pole clamps), or the
// tile's box exceeds the slice -- takes its taps with the gather form of the lean member loop instead (same registers, same
// blend): the two paths are bit-identical by construction, both read the same source bytes.
// Ring-shared coordinates wait in LDS as in the lean loop (latitude, left / mirrored longitude); per pitch sign one more
// entry holds each pixel's row offset inside the box with the vertical phase in its low five bits (the pitch is a multiple of 32).
constexpr int kStageBytesPlain = 6144;
constexpr int kStageBytes = kStageBytesPlain;                         // per wavefront: 24 KiB + 16 KiB of parked coordinates = four workgroups per CU
constexpr int kStageBytesMasked = 5120;   // 20 KiB of slices + 20 KiB of parked entries = four workgroups per CU (6144: three; cfg3 + mask 108 -> 100 us)
// s_waitcnt immediates (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]); issued through the builtin so that the
// compiler's own wait-count bookkeeping sees them (it does not look into inline assembly)
#define GS360_WAIT_VM0() __builtin_amdgcn_s_waitcnt(0x0F70)      /* vmcnt(0) */
#define GS360_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)    /* lgkmcnt(0) */
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

template <bool MASKED>
__global__ __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(kEqStagedWaves, kEqStagedWaves))) void eq_staged_kernel(const EqLaunch L) {
    constexpr int kSliceBytes = MASKED ? kStageBytesMasked : kStageBytes;   // masked: one more parked entry per pixel
    constexpr int kSliceRounds = kSliceBytes / 16 / 64;
    static_assert(kSliceBytes % 1024 == 0, "a slice is whole DMA rounds");
    __shared__ __attribute__((aligned(16))) uint32_t s_stage[kWaves * kSliceBytes / 4];
    __shared__ __attribute__((aligned(16))) int4 s_park[(MASKED ? 5 : 4) * 64 * kWaves];
    const int b = blockIdx.x;
    int t = (b & 7) * L.chunk + (b >> 3);
    if (L.xcd_group_log2 >= 0) {
        const int q = b >> 3, g = L.xcd_group_log2;
        t = ((((q >> g) << 3) + (b & 7)) << g) + (q & ((1 << g) - 1));
    }
    if (t >= L.total_tiles) return;
    const int f = t / L.tiles_per_frame;
    int r = t - f * L.tiles_per_frame;
    int g = 0;
    while (g + 1 < L.n_rings && r >= L.view[L.ring_first[g + 1]].tile_base) ++g;
    const int k0 = L.ring_first[g], n_members = L.ring_count[g];
    const EqView& V = L.view[k0];
    r -= V.tile_base;
    const int tile_y = r / V.tiles_x, tile_x = r - tile_y * V.tiles_x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half_w = (V.out_w + 1) >> 1;
    const int x0 = tile_x * kTileW + 16 * wave;           // this wavefront's first column (left half of the view)
    const int n_w = min(16, half_w - x0);                 // its columns
    if (n_w <= 0) return;                                 // (no workgroup barrier anywhere below)
    const int col = lane & 15, rowq = lane >> 4;
    const int xl = min(x0 + col, half_w - 1);
    const uint8_t* __restrict__ src = L.src[f];
    const uint8_t* __restrict__ mask = L.mask[f];
    const int out_w = uniform_here(V.out_w), out_h = uniform_here(V.out_h);
    const uint32_t dstride = (uint32_t)(L.dst_stride ? L.dst_stride : (int64_t)V.out_w * 3);
    const int W = uniform_here(L.W), H = uniform_here(L.H), W32 = uniform_here(32 * L.W);
    const uint32_t stride = (uint32_t)uniform_here((int)L.src_stride);
    const uint32_t mstride = (uint32_t)uniform_here((int)L.mask_stride);
    const int y0x2 = uniform_here(2 * L.y0i32);

    // ---- coordinates (EQ-SPEC v1, the general per-pixel form; level views give the same bits through it) --------------------------
    const float x = (float)(2 * xl + 1 - V.out_w) * V.sxu;
    int ys[kRowsPerWave], sxl[kRowsPerWave], sxm[kRowsPerWave], sys[kRowsPerWave];
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s) {
        ys[s] = tile_y * kTileH + 4 * s + rowq;           // rows past the image repeat the last one (computed, never stored)
        float yv = (float)(2 * min(ys[s], V.out_h - 1) + 1 - V.out_h) * V.syv;
        float xr = x, bz, cy;
        if (V.fish) {
            const float q = __builtin_fmaf(x, x, yv * yv);
            const float Sx = eq_poly8(kEqFishS, q), Cz = eq_poly8(kEqFishC, q);
            xr = x * Sx;
            yv = yv * Sx;
            bz = __builtin_fmaf(V.sp, yv, V.cp * Cz);
            cy = __builtin_fmaf(-V.cp, yv, V.sp * Cz);
        } else {
            bz = __builtin_fmaf(V.sp, yv, V.cp);
            cy = __builtin_fmaf(-V.cp, yv, V.sp);
        }
        const float h = eq_sqrt(__builtin_fmaf(xr, xr, bz * bz));
        int Kl, Kt;
        const float rl = eq_atan2_red(xr, bz, Kl);
        const float rt = eq_atan2_red<true>(cy, h, Kt);
        sxl[s] = eq_lon_base(rl, Kl, L, V.x0f32);         // in [-18 W, 18 W + 32]: no wrap inside a tile
        sxm[s] = eq_lon_base(-rl, -Kl, L, V.x0f32);
        sys[s] = L.y0i32 - Kt * 8 * L.H - (int)__builtin_rintf(rt * L.ky32);
    }
    // ---- the tile's box in ring-shared coordinates (1/32-texel units): raw longitude ranges of the two halves, latitude range -----
    const int minL = wave_min_i32(min(min(sxl[0], sxl[1]), min(sxl[2], sxl[3]))), maxL = wave_max_i32(max(max(sxl[0], sxl[1]), max(sxl[2], sxl[3])));
    const int minM = wave_min_i32(min(min(sxm[0], sxm[1]), min(sxm[2], sxm[3]))), maxM = wave_max_i32(max(max(sxm[0], sxm[1]), max(sxm[2], sxm[3])));
    const int symin = wave_min_i32(min(min(sys[0], sys[1]), min(sys[2], sys[3]))), symax = wave_max_i32(max(max(sys[0], sys[1]), max(sys[2], sys[3])));
    // row pitch of the box: the taps' texel span + the 12-byte read of the last one + the 16-byte alignment of xa, in 32-byte steps
    const int tspan = (max(maxL - minL, maxM - minM) >> 5) + 1;
    const int pitch = (3 * tspan + 27 + 31) & ~31;
    const int ny_max = ((symax - symin) >> 5) + 3;
    // (pitch <= stride: a box row that starts inside row y ends inside row y + 1 at the latest, and the last row a box may hold is
    // H - 2 -- the copy never leaves the frame)
    const bool tile_fits = ny_max * pitch <= kSliceBytes && (stride & 3u) == 0 && (uint32_t)pitch <= stride && kStageEnable;
    const int nchp = pitch >> 4;                          // 16-byte chunks per box row
    // DMA lane map, fixed for the tile: round k moves chunks 64 k + lane; chunk c = (row c / nchp, piece c % nchp)
    uint32_t voff[kSliceRounds];
    {
        const float inv = 1.0f / (float)nchp;
#pragma unroll
        for (int k = 0; k < kSliceRounds; ++k) {
            const int c = lane + 64 * k;
            const int row = (int)(((float)c + 0.5f) * inv);          // exact for c < 2^10
            voff[k] = (uint32_t)row * stride + (uint32_t)(c - row * nchp) * 16u;
        }
    }
    uint32_t* const stage = s_stage + wave * (kSliceBytes / 4);
    int4* const park = s_park + threadIdx.x;
    constexpr int kP = 64 * kWaves;
#pragma unroll
    for (int s = 0; s < kRowsPerWave; ++s) {              // members wrap with one unsigned minimum: bases into [0, 32 W)
        sxl[s] = eq_lon_norm(sxl[s], W32);
        sxm[s] = eq_lon_norm(sxm[s], W32);
    }
    park[0 * kP] = make_int4(sys[0], sys[1], sys[2], sys[3]);
    park[1 * kP] = make_int4(sxl[0], sxl[1], sxl[2], sxl[3]);
    park[2 * kP] = make_int4(sxm[0], sxm[1], sxm[2], sxm[3]);

    // ---- per pitch sign: the box's rows, each pixel's row offset in it (| vertical phase) ----------------------------------------
    bool flipstate = false;
    int iy_lo = 0, ny = 0;                                // box rows [iy_lo, iy_lo + ny) for the current pitch sign
    bool rows_ok = false;                                 // ... lie inside [0, H - 2]: no pole clamp, the last image row is never staged
    auto derive_lat = [&](const int4 lat, const bool flip) {
        const int lo = flip ? y0x2 - symax : symin, hi = flip ? y0x2 - symin : symax;
        iy_lo = lo >> 5;
        ny = (hi >> 5) + 2 - iy_lo;
        rows_ok = tile_fits && iy_lo >= 0 && iy_lo + ny - 1 <= H - 2;
        const int la[4] = {lat.x, lat.y, lat.z, lat.w};
        int a[4];
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const int sy = flip ? y0x2 - la[s] : la[s];
            a[s] = (int)__umul24((uint32_t)((sy >> 5) - iy_lo), (uint32_t)pitch) | (sy & 31);
        }
        park[3 * kP] = make_int4(a[0], a[1], a[2], a[3]);
        if constexpr (MASKED) {
            int mr[4];
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s)
                mr[s] = (int)__umul24((uint32_t)((flip ? y0x2 - la[s] : la[s]) + 16) >> 5, mstride);
            park[4 * kP] = make_int4(mr[0], mr[1], mr[2], mr[3]);
        }
    };
    // the box of one pass along x: member offset applied to the raw range, reduced to [0, 32 W); qualifies if it does not cross the seam
    struct Box { bool ok; int xa; uint32_t origin; };
    auto box_of = [&](const int x0i, const bool mirror) {
        int a = (mirror ? minM : minL) + x0i;
        const int span = mirror ? maxM - minM : maxL - minL;
        if (a < 0) a += W32;
        if (a >= W32) a -= W32;
        Box bx;
        const int ix_lo = a >> 5, ix_hi = (a + span) >> 5;            // left taps' texels
        bx.ok = rows_ok && ix_hi + 1 <= W - 1;
        bx.xa = (3 * ix_lo) & ~15;
        bx.origin = (uint32_t)iy_lo * stride + (uint32_t)bx.xa;
        return bx;
    };
    auto dma = [&](const Box& bx) {                       // start the copy of a box into the wavefront's slice
        const int total = ny * nchp;
        const uint8_t* const base = src + (size_t)bx.origin;
#pragma unroll
        for (int k = 0; k < kSliceRounds; ++k)
            if (64 * k < total) {                         // wave-uniform
                if (lane + 64 * k < total)
                    __builtin_amdgcn_global_load_lds((global_void_t*)(base + (size_t)voff[k]), (lds_void_t*)(stage + 256 * k), 16, 0, 0);
            }
    };

    int2 mem = *reinterpret_cast<const int2*>(&L.view[k0].x0i32);
    const int dst_base = uniform_here(f * L.n_views + k0);
    uint8_t* dst = L.dst[dst_base];
    flipstate = mem.y != 0;
    derive_lat(make_int4(sys[0], sys[1], sys[2], sys[3]), flipstate);
    const bool centre_dup = (V.out_w & 1) && (x0 + n_w == half_w);
    const bool has_mirror = n_w > (centre_dup ? 1 : 0);
    const int c0_m = out_w - x0 - n_w;                    // first column of the mirrored segment
    const int fix_from = uniform_here(32 * (W - 4));

    // the pass in flight: raw tap dwords, tap byte offsets (low two bits = misalignment), longitude coordinate, vertical phase
    uint32_t ra[kRowsPerWave][3], rb[kRowsPerWave][3], o0[kRowsPerWave], o1[kRowsPerWave], keep[kRowsPerWave], kbit[kRowsPerWave];
    int cx[kRowsPerWave], fyv[kRowsPerWave];
    bool taps_staged = false;                             // how the pass in flight got its taps (wave-uniform)
    // taps of a pass from the staged box (LDS) ...
    auto taps_from_lds = [&](const int4 lon, const int x0i, const Box& bx) {
        const int4 aq = park[3 * kP];
        const int lo[4] = {lon.x, lon.y, lon.z, lon.w}, a[4] = {aq.x, aq.y, aq.z, aq.w};
        const uint8_t* const sb = reinterpret_cast<const uint8_t*>(stage);
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            cx[s] = eq_lon_member(lo[s], x0i, W32);
            fyv[s] = a[s] & 31;
            o0[s] = (uint32_t)(a[s] & ~31) + (uint32_t)(3 * (cx[s] >> 5) - bx.xa);
            o1[s] = o0[s] + (uint32_t)pitch;
            const uint32_t* qa = reinterpret_cast<const uint32_t*>(sb + (o0[s] & ~3u));
            const uint32_t* qb = reinterpret_cast<const uint32_t*>(sb + (o1[s] & ~3u));
            ra[s][0] = qa[0]; ra[s][1] = qa[1]; ra[s][2] = qa[2];
            rb[s][0] = qb[0]; rb[s][1] = qb[1]; rb[s][2] = qb[2];
        }
    };
    // ... or gathered from memory (the lean member loop's fetch, row offsets from scratch: rare)
    auto taps_from_memory = [&](const int4 lon, const int x0i, const bool flip) {
        const int4 lat = park[0 * kP];
        const int lo[4] = {lon.x, lon.y, lon.z, lon.w}, la[4] = {lat.x, lat.y, lat.z, lat.w};
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const int sy = flip ? y0x2 - la[s] : la[s];
            const int iy = sy >> 5;
            cx[s] = eq_lon_member(lo[s], x0i, W32);
            fyv[s] = sy & 31;
            const uint32_t cb = (uint32_t)min(cx[s] >> 5, W - 5) * 3u;
            o0[s] = __umul24((uint32_t)max(iy, 0), stride) + cb;
            o1[s] = __umul24((uint32_t)min(iy + 1, H - 1), stride) + cb;
            const uint32_t* qa = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + (o0[s] & ~3u), 4));
            const uint32_t* qb = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + (o1[s] & ~3u), 4));
            ra[s][0] = qa[0]; ra[s][1] = qa[1]; ra[s][2] = qa[2];
            rb[s][0] = qb[0]; rb[s][1] = qb[1]; rb[s][2] = qb[2];
        }
    };
    auto mask_fetch = [&](const int4 lon, const int x0i) {
        if constexpr (MASKED) {
            const int4 mrq = park[4 * kP];
            const int lo[4] = {lon.x, lon.y, lon.z, lon.w}, mr[4] = {mrq.x, mrq.y, mrq.z, mrq.w};
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                const uint32_t tt = (uint32_t)eq_lon_member(lo[s], x0i, W32) + 16u;
                kbit[s] = tt >> 5;
                keep[s] = *reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(mask + (size_t)((uint32_t)mr[s] + ((tt >> 8) & ~3u)), 4));
            }
        }
    };
    auto resolve = [&](uint32_t (&pk)[kRowsPerWave], const bool flip) {
        uint32_t px[kRowsPerWave][4];
        bool any_fix = false;
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            EqTaps<3> tp;
            tp.t0.x = __builtin_amdgcn_alignbyte(ra[s][1], ra[s][0], GS360_AB(o0[s]));
            tp.t0.y = __builtin_amdgcn_alignbyte(ra[s][2], ra[s][1], GS360_AB(o0[s]));
            tp.t1.x = __builtin_amdgcn_alignbyte(rb[s][1], rb[s][0], GS360_AB(o1[s]));
            tp.t1.y = __builtin_amdgcn_alignbyte(rb[s][2], rb[s][1], GS360_AB(o1[s]));
            eq_blend_f<3>(tp, cx[s] & 31, fyv[s], px[s]);
            any_fix |= cx[s] >= fix_from;
        }
        if (!taps_staged && any_lane(any_fix)) {          // a staged box never reaches the seam columns
            const int4 lat = park[0 * kP];
            const int la[4] = {lat.x, lat.y, lat.z, lat.w};
            EqSrc S;
            S.W = W; S.H = H; S.src_stride = (int64_t)stride; S.mask_stride = 0; S.stride4 = false;
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s)
                if (cx[s] >= fix_from) eq_sample_slow<3>(src, S.src_stride, S.W, S.H, cx[s], flip ? y0x2 - la[s] : la[s], px[s]);
        }
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            pk[s] = px[s][0] | (px[s][1] << 8) | (px[s][2] << 16);
            if constexpr (MASKED) pk[s] &= (uint32_t)__builtin_amdgcn_sbfe((int)keep[s], kbit[s] & 31u, 1u);
        }
    };
    // store of one pass: each 16-lane group holds 16 pixels of one row = 12 dwords; lanes 0..11 of the group re-slice and write them
    const int dsel = col < 12 ? col : 0;                                   // dword of the row fragment this lane writes
    const int a16 = (4 * dsel) / 3, sh16 = (4 * dsel) - 3 * a16;          // first contributing pixel, byte offset in it
    const uint32_t sel16 = sh16 == 0 ? 0x04020100u : (sh16 == 1 ? 0x05040201u : 0x06050402u);
    auto store_pass = [&](const uint32_t (&pk)[kRowsPerWave], uint8_t* const d, const bool flip, const bool mirror) {
        if (mirror && !has_mirror) return;
        const int c0 = mirror ? c0_m : x0;
        const bool aligned = ((dstride & 3u) == 0) && ((reinterpret_cast<uintptr_t>(d) & 3) == 0) && (((c0 * 3) & 3) == 0) &&
                             !(mirror && centre_dup) && ((3 * n_w) & 3) == 0;
        int ln = lane;
        asm volatile("" : "+v"(ln));                      // lane-derived constants are made here, once per pass (see the lean loop)
        const int grp = ln & ~15, cc = ln & 15;
        if (aligned) {
            // pixel at position q of the row fragment sits in lane grp + q (left half) or grp + n_w - 1 - q (mirrored half)
            const int qa = a16, qb = a16 + 1;
            const int la4 = 4 * (grp + (mirror ? n_w - 1 - qa : qa)), lb4 = 4 * (grp + (mirror ? n_w - 1 - qb : qb));
            uint32_t dw[kRowsPerWave];
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(la4 & 252, (int)pk[s]);
                const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(lb4 & 252, (int)pk[s]);
                dw[s] = __builtin_amdgcn_perm(pb, pa, sel16);
            }
            const int full = (3 * n_w) >> 2;
#pragma unroll
            for (int s = 0; s < kRowsPerWave; ++s) {
                const int y = flip ? out_h - 1 - ys[s] : ys[s];
                if (cc < full && ys[s] < out_h)
                    *reinterpret_cast<uint32_t*>(d + (size_t)(__umul24((uint32_t)y, dstride) + (uint32_t)(c0 * 3 + 4 * cc))) = dw[s];
            }
            return;
        }
        const int pos = mirror ? n_w - 1 - cc : cc;
#pragma unroll
        for (int s = 0; s < kRowsPerWave; ++s) {
            const int y = flip ? out_h - 1 - ys[s] : ys[s];
            if (cc < n_w && ys[s] < out_h && !(mirror && centre_dup && pos == 0)) {
                uint8_t* q = d + (size_t)(__umul24((uint32_t)y, dstride) + (uint32_t)((c0 + pos) * 3));
                q[0] = (uint8_t)pk[s]; q[1] = (uint8_t)(pk[s] >> 8); q[2] = (uint8_t)(pk[s] >> 16);
            }
        }
    };

    // ---- the passes: member 0 left, member 0 mirrored, member 1 left, ... ---------------------------------------------------------
    const int p_last = 2 * n_members - 1;
    // prologue: pass 0's taps into registers, pass 1's box on its way
    Box bx = box_of(mem.x, false);
    if (bx.ok) {
        dma(bx);
        GS360_WAIT_VM0();
        taps_from_lds(make_int4(sxl[0], sxl[1], sxl[2], sxl[3]), mem.x, bx);
        taps_staged = true;
    } else {
        taps_from_memory(make_int4(sxl[0], sxl[1], sxl[2], sxl[3]), mem.x, flipstate);
        GS360_WAIT_VM0();                                 // (see the loop)
        taps_staged = false;
    }
    mask_fetch(make_int4(sxl[0], sxl[1], sxl[2], sxl[3]), mem.x);
    Box bx_next = box_of(mem.x, true);                    // pass 1: the mirrored half of member 0, same rows
    GS360_WAIT_LGKM0();    // the slice has been read: it may be overwritten
    __builtin_amdgcn_sched_barrier(0);
    if (p_last >= 1 && bx_next.ok) dma(bx_next);
    int2 mem_nx = mem;
    uint8_t* dst_nx = dst;
    uint32_t pk[kRowsPerWave];
    bool pflip = flipstate;                               // pitch sign of the pass whose taps are in registers
#pragma unroll 1
    for (int p = 0; p < p_last; ++p) {
        const bool mirror = (p & 1) != 0;                 // the pass whose taps are in registers is a mirrored half
        const bool flip = pflip;
        uint8_t* const dst_cur = dst;
        if (!mirror) {                                    // the next member's scalars, ahead of their use
            const int mn = min((p >> 1) + 1, n_members - 1);
            mem_nx = *reinterpret_cast<const int2*>(&L.view[k0 + mn].x0i32);
            dst_nx = L.dst[dst_base + mn];
        }
        __builtin_amdgcn_sched_barrier(0);
        resolve(pk, flip);
        __builtin_amdgcn_sched_barrier(0);
        // pass p + 1: its box was started a turn ago (bx_next); take its taps, then start pass p + 2's box
        const Box bcur = bx_next;
        if (mirror) {                                     // p + 1 = left half of the next member
            mem = mem_nx;
            dst = dst_nx;
        }
        const int4 lon = park[(mirror ? 1 : 2) * kP];
        if (bcur.ok) {
            GS360_WAIT_VM0();
            taps_from_lds(lon, mem.x, bcur);
            taps_staged = true;
        } else {
            // the rare gather form waits for its reads here: left in flight they would sit in front of the next box's DMA in the
            // memory queue, and the blend at the top of the next turn -- which cannot know which form filled its registers -- would
            // have to wait for everything, the DMA included
            taps_from_memory(lon, mem.x, flipstate);
            GS360_WAIT_VM0();
            taps_staged = false;
        }
        mask_fetch(lon, mem.x);
        pflip = flipstate;                                // (the entries derived for pass p + 1's pitch sign are still the current ones)
        // pass p + 2 (if any): mirrored half of the same member, or the left half of the member after it (whose pitch sign may differ:
        // the latitude entry is re-derived, which needs pass p + 1's taps out of LDS first -- they are, and A was read above)
        if (p + 2 <= p_last) {
            if (mirror) {
                bx_next = box_of(mem.x, true);
            } else {
                if ((mem_nx.y != 0) != flipstate) {
                    // pass p + 1 (mirrored half of the current member) already has its taps and phases in registers
                    flipstate = !flipstate;
                    derive_lat(park[0 * kP], flipstate);
                }
                bx_next = box_of(mem_nx.x, false);
            }
            GS360_WAIT_LGKM0();
            __builtin_amdgcn_sched_barrier(0);
            if (bx_next.ok) dma(bx_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        store_pass(pk, dst_cur, flip, mirror);
    }
    __builtin_amdgcn_sched_barrier(0);
    resolve(pk, pflip);
    store_pass(pk, dst, pflip, true);
}


// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
// Keep-mask threshold + pack: bit x of row y = (mask[y][x] >= 128), i.e. the byte's top bit; bit W repeats bit 0 and row H repeats
// row H - 1 (eq_mask_at).  One thread per output dword (32 mask bytes); a streaming pass over the byte masks, run by the C ABI on the
// launch stream in front of the view kernel.
__global__ __launch_bounds__(256) void mask_pack_kernel(const MaskPack P) {
    const int f = blockIdx.z, y = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= P.pitch_dw) return;
    const int ys = min(y, P.H - 1);
    const uint8_t* __restrict__ row = P.src[f] + (int64_t)ys * P.stride;
    const int x0 = d * 32;
    uint32_t bits = 0;
    if (x0 + 32 <= P.W && ((reinterpret_cast<uintptr_t>(row) + (uintptr_t)x0) & 15) == 0) {
        const uint4* q = reinterpret_cast<const uint4*>(row + x0);
        const uint4 a = q[0], b = q[1];
        const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) bits |= (((((v[k] >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xfu) << (4 * k);   // byte top bits -> nibble
    } else {
        for (int k = 0; k < 32; ++k) {
            const int x = x0 + k;
            if (x <= P.W) bits |= (uint32_t)(row[x == P.W ? 0 : x] >> 7) << k;
        }
    }
    if (x0 <= P.W && P.W < x0 + 32) bits |= (uint32_t)(row[0] >> 7) << (P.W - x0);      // the wrap column (fast path of the last full dword)
    P.dst[f][(int64_t)y * P.pitch_dw + d] = bits;
}
hipError_t launch_mask_pack(const MaskPack& P, hipStream_t s) {
    dim3 grid((unsigned)((P.pitch_dw + 255) / 256), (unsigned)(P.H + 1), (unsigned)P.n), block(256);
    hipLaunchKernelGGL(mask_pack_kernel, grid, block, 0, s, P);
    return hipGetLastError();
}

static unsigned eq_grid_blocks(const EqLaunch& L) {
    if (L.xcd_group_log2 < 0) return (unsigned)(L.chunk * 8);
    const int per = 8 << L.xcd_group_log2;                 // tiles per round over the XCDs
    return (unsigned)((L.total_tiles + per - 1) / per * per);
}

hipError_t launch_equirect_staged(const EqLaunch& L, hipStream_t s) {
    dim3 grid(eq_grid_blocks(L)), block(64 * kWaves);
    if (L.mask[0] != nullptr) hipLaunchKernelGGL((eq_staged_kernel<true>), grid, block, 0, s, L);
    else hipLaunchKernelGGL((eq_staged_kernel<false>), grid, block, 0, s, L);
    return hipGetLastError();
}

hipError_t launch_equirect(const EqLaunch& L, int C, hipStream_t s) {
    dim3 grid(eq_grid_blocks(L)), block(64 * kWaves);
    const bool masked = L.mask[0] != nullptr;             // all frames or none (checked by the C ABI)
    bool rows_only = kEqRowsKernel != 0;
    for (int k = 0; k < L.n_views; ++k) rows_only = rows_only && !L.view[k].blocked;
    switch (C) {
        case 1: if (masked) hipLaunchKernelGGL((eq_views_kernel<1, false, true>), grid, block, 0, s, L);
                else hipLaunchKernelGGL((eq_views_kernel<1, false, false>), grid, block, 0, s, L); break;
        case 3: if (rows_only) {
                    if (masked) hipLaunchKernelGGL((eq_views_kernel<3, false, true, 1, true>), grid, block, 0, s, L);
                    else hipLaunchKernelGGL((eq_views_kernel<3, false, false, 1, true>), grid, block, 0, s, L);
                } else if (masked) hipLaunchKernelGGL((eq_views_kernel<3, false, true>), grid, block, 0, s, L);
                else hipLaunchKernelGGL((eq_views_kernel<3, false, false>), grid, block, 0, s, L); break;
        case 4: if (masked) hipLaunchKernelGGL((eq_views_kernel<4, false, true>), grid, block, 0, s, L);
                else hipLaunchKernelGGL((eq_views_kernel<4, false, false>), grid, block, 0, s, L); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// persistent grid of the cubic variants: at most `persist_blocks` workgroups (rounded to the XCD count) walk the tile order
static EqLaunch eq_persistent(const EqLaunch& L0, dim3& grid) {
    EqLaunch L = L0;
    L.grid_total = (int32_t)grid.x;
    if (L.persist_blocks > 0 && (unsigned)L.persist_blocks < grid.x) grid.x = (unsigned)(L.persist_blocks + 7) & ~7u;
    return L;
}

hipError_t launch_equirect_cubic(const EqLaunch& L0, int C, hipStream_t s) {
    dim3 grid(eq_grid_blocks(L0)), block(64 * kWaves);
    const EqLaunch L = eq_persistent(L0, grid);
    const bool masked = L.mask[0] != nullptr;
    // (a rows-only instantiation like the bilinear kernel's halves the SGPR spills -- 61 -> 33 -- and changes nothing measurable:
    // profiles/r04/cubic_split_ab.txt)
    switch (C) {
        case 1: if (masked) hipLaunchKernelGGL((eq_views_kernel<1, true, true>), grid, block, 0, s, L);
                else hipLaunchKernelGGL((eq_views_kernel<1, true, false>), grid, block, 0, s, L); break;
        case 3: if (masked) hipLaunchKernelGGL((eq_views_kernel<3, true, true>), grid, block, 0, s, L);
                else hipLaunchKernelGGL((eq_views_kernel<3, true, false>), grid, block, 0, s, L); break;
        case 4: if (masked) hipLaunchKernelGGL((eq_views_kernel<4, true, true>), grid, block, 0, s, L);
                else hipLaunchKernelGGL((eq_views_kernel<4, true, false>), grid, block, 0, s, L); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_equirect_u16(const EqLaunch& L0, int C, bool cubic, hipStream_t s) {
    dim3 grid(eq_grid_blocks(L0)), block(64 * kWaves);
    const EqLaunch L = cubic ? eq_persistent(L0, grid) : L0;
    if (cubic) {
        switch (C) {
            case 1: hipLaunchKernelGGL((eq_views_kernel<1, true, false, 2>), grid, block, 0, s, L); break;
            case 3: hipLaunchKernelGGL((eq_views_kernel<3, true, false, 2>), grid, block, 0, s, L); break;
            case 4: hipLaunchKernelGGL((eq_views_kernel<4, true, false, 2>), grid, block, 0, s, L); break;
            default: return hipErrorInvalidValue;
        }
    } else {
        switch (C) {
            case 1: hipLaunchKernelGGL((eq_views_kernel<1, false, false, 2>), grid, block, 0, s, L); break;
            case 3: hipLaunchKernelGGL((eq_views_kernel<3, false, false, 2>), grid, block, 0, s, L); break;
            case 4: hipLaunchKernelGGL((eq_views_kernel<4, false, false, 2>), grid, block, 0, s, L); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}

}  // namespace gs360
