// gs360_sampler.h -- device helpers shared by the equirect kernels (gs360_kernels.hip) and the cv2.remap / fused-fisheye kernels
// (gs360_table.hip): lane / uniform helpers, the row-paired RGB tap fetch, the dword re-slicing row store, the bicubic fetch + LDS weight
// table + blend.  (Split out of gs360_kernels.hip in round 5: one hot file per kernel family.)
#pragma once
#include <type_traits>

#include "gs360_kernels.h"
#include "gs360_eqspec.h"
#include "gs360_blend.h"
#include "gs360_rowstore.h"

namespace gs360 {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint2 ld_u64(const uint8_t* p) {
    uint2 v;
    __builtin_memcpy(&v, p, 8);  // unaligned 8-byte load (gfx950 runs in unaligned access mode)
    return v;
}
__device__ __forceinline__ uint32_t byte_of(uint32_t v, int k) { return (v >> (8 * k)) & 0xffu; }

// Lane index / uniform value behind an optimisation barrier.  The store helpers derive a dozen per-lane constants from the
// lane index (dword slicing of 3-byte pixels); inside the ring-member loop of eq_views_kernel the compiler would hoist all
// of them -- for every store variant -- out of the loop and spill (measured: 120 VGPRs spilled at a 96-register budget).
// A value that comes out of a volatile asm cannot be hoisted or merged, so each store recomputes its few constants in place.
__device__ __forceinline__ int lane_here() {
    int l = threadIdx.x & 63;
    asm volatile("" : "+v"(l));
    return l;
}
__device__ __forceinline__ int uniform_here(int v) {
    v = __builtin_amdgcn_readfirstlane(v);     // wave-uniform by construction; a no-op when the value already sits in an SGPR
    asm volatile("" : "+s"(v));
    return v;
}

// "some lane": the comparison's lane mask tested directly.  (__any() goes through an int -- v_cndmask 0/1 + v_cmp_ne per call --
// and these tests sit in the arithmetic-bound inner loops.)
__device__ __forceinline__ bool any_lane(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// min(max(x, 0), hi) for a wave-uniform hi >= 0 as ONE v_med3_i32 (the compiler only fuses the pair when both bounds are constants)
__device__ __forceinline__ int clamp0_uniform(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

// 8 bytes starting at the (unaligned) address p, fetched as ONE dword-aligned 12-byte access and shifted into
// place with v_alignbyte.  The texture-address path merges dword-aligned lane accesses of a quad into cache-line
// requests; byte-misaligned ones are looked up lane by lane (measured: 96 tag lookups per 64-lane instruction).
// Reads bytes [p & ~3, (p & ~3) + 12).
__device__ __forceinline__ uint2 ld_u64_via_aligned96(const uint8_t* p) {
    uint32_t o = (uint32_t)reinterpret_cast<uintptr_t>(p) & 3u;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(p - o, 4));  // stays a global pointer
    uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    uint2 v;
    v.x = __builtin_amdgcn_alignbyte(d1, d0, o);
    v.y = __builtin_amdgcn_alignbyte(d2, d1, o);
    return v;
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define GS360_AB(o) (o)   // v_alignbyte_b32 shifts by S2[1:0] bytes: a byte offset's upper bits need not be masked off
constexpr bool kShiftedStore = true;   // dword stores for row segments that start off a dword boundary (false: byte stores, the A/B reference)

// The 8 tap bytes (two RGB pixels + 2) of rows y0 and y1 of one pixel, at byte offsets o0 / o1 from `src`.
// Row-paired gathers: issued naively, one instruction reads row y0 of all 64 pixels and the next one row y1; where the
// view bends across source rows, row Y is "y0" for one run of lanes and "y1" for the neighbouring run, so the second
// instruction asks for lines the first one has just missed on and the L1 stalls on the pending fill.  Here lanes 0-31
// of the first instruction read row y0 and lanes 32-63 row y1 of the SAME 32 pixels (second instruction: the other 32
// pixels), so both uses of a line meet in one instruction and are merged by the address coalescer.
// v_permlane32_swap (gfx950) builds the two address vectors from (o0, o1) in one operation and puts the returned
// dwords back in pixel order.  Each read is a dword-aligned 12-byte access shifted into place with v_alignbyte.
// Two steps so that a wavefront can put ALL its gathers in flight before the first result is touched (the caller separates
// the steps with a scheduling barrier: left to itself the scheduler interleaves load pairs with their consumers as soon as
// the surrounding code tightens the register budget, which serialises the misses -- measured 20.7 -> 30.1 us per cfg2 frame):
//   ld_rows_rgb_issue   address swap + the two 12-byte loads (raw dwords, still in fetch order)
//   ld_rows_rgb_finish  swap the returned dwords back into pixel order and shift them into place
struct RowsRaw { uint32_t a0, a1, a2, b0, b1, b2, sh; };   // sh = (o0 & 3) | (o1 & 3) << 2
__device__ __forceinline__ RowsRaw ld_rows_rgb_issue(const uint8_t* __restrict__ src, uint32_t o0, uint32_t o1) {
    RowsRaw r;
    r.sh = (o0 & 3u) | ((o1 & 3u) << 2);
    const u32x2 adr = __builtin_amdgcn_permlane32_swap(o0 & ~3u, o1 & ~3u, false, false);
    const uint32_t* qa = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + adr.x, 4));
    const uint32_t* qb = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(src + adr.y, 4));
    r.a0 = qa[0]; r.a1 = qa[1]; r.a2 = qa[2];
    r.b0 = qb[0]; r.b1 = qb[1]; r.b2 = qb[2];
    return r;
}
__device__ __forceinline__ void ld_rows_rgb_finish(const RowsRaw& r, uint2& t0, uint2& t1) {
    const uint32_t s0 = r.sh & 3u, s1 = r.sh >> 2;
    const u32x2 d0 = __builtin_amdgcn_permlane32_swap(r.a0, r.b0, false, false);   // .x = row y0, .y = row y1, own pixel
    const u32x2 d1 = __builtin_amdgcn_permlane32_swap(r.a1, r.b1, false, false);
    const u32x2 d2 = __builtin_amdgcn_permlane32_swap(r.a2, r.b2, false, false);
    t0.x = __builtin_amdgcn_alignbyte(d1.x, d0.x, s0);
    t0.y = __builtin_amdgcn_alignbyte(d2.x, d1.x, s0);
    t1.x = __builtin_amdgcn_alignbyte(d1.y, d0.y, s1);
    t1.y = __builtin_amdgcn_alignbyte(d2.y, d1.y, s1);
}
__device__ __forceinline__ void ld_rows_rgb(const uint8_t* __restrict__ src, uint32_t o0, uint32_t o1, uint2& t0, uint2& t1) {
    ld_rows_rgb_finish(ld_rows_rgb_issue(src, o0, o1), t0, t1);
}

// bilinear blend of one channel, weights a0+a1 = 32, b0+b1 = 32  ->  (sum + 512) >> 10
__device__ __forceinline__ uint32_t blend(uint32_t s00, uint32_t s01, uint32_t s10, uint32_t s11,
                                          uint32_t w00, uint32_t w01, uint32_t w10, uint32_t w11) {
    return (s00 * w00 + s01 * w01 + s10 * w10 + s11 * w11 + 512u) >> 10;
}

// Store one wavefront row segment of n_px pixels.  Lane l holds the pixel at position l of the segment
// (reversed = false) or at position n_px-1-l (reversed = true, the mirrored half of a view); channels in px[0..C-1].
// C == 3: pixels are packed to 24 bits and re-sliced into dwords with two cross-lane shuffles so that the
// row leaves as 4-byte stores (lane j writes bytes 4j..4j+3 = tail of pixel 4j/3 + head of the next one).
// skip_first drops position 0 (the centre column of an odd-width view, which is its own mirror); the caller then
// passes aligned4 = false and the per-lane byte path below handles it.
// SHIFTED = false leaves the off-boundary dword path out (the bicubic equirect kernels sit at their register limit).
template <int C, bool SHIFTED = true>
__device__ __forceinline__ void store_row(uint8_t* row, const uint32_t (&px)[4], int n_px, bool aligned4, const RowPack& rp,
                                          bool reversed = false, bool skip_first = false) {
    const int lane = rp.lane;
    if constexpr (C == 3) {
        if (aligned4) {
            uint32_t packed = px[0] | (px[1] << 8) | (px[2] << 16);
            // source lanes (as byte addresses, wrapped to the wavefront): pixel a and a + 1, or their mirror images
            const int la4 = reversed ? 4 * (n_px - 1) - rp.a4 : rp.a4, lb4 = reversed ? la4 - 4 : la4 + 4;
            const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(la4 & 252, (int)packed);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(lb4 & 252, (int)packed);
            const uint32_t dw = __builtin_amdgcn_perm(pb, pa, rp.sel);   // = (pa >> sh) | (pb << (24 - sh)) on 24-bit pixels, one instruction
            int n_bytes = 3 * n_px, full = n_bytes >> 2, rem = n_bytes & 3;
            if (lane < full) __builtin_nontemporal_store(dw, reinterpret_cast<uint32_t*>(row) + lane);   // written once, never re-read
            if (lane == full && rem)
                for (int k = 0; k < rem; ++k) row[4 * full + k] = (uint8_t)(dw >> (8 * k));
            return;
        }
        if (SHIFTED && kShiftedStore && !skip_first) {
            // a segment that starts off a dword boundary (widths that are not multiples of four): the same two shuffles, the segment's
            // byte stream re-sliced at its own misalignment -- lanes 0..47 write the aligned dwords inside it, lanes 48..50 its 0-3 head
            // bytes, lanes 52..54 its 0-3 tail bytes (one dword store + one byte store instead of three byte stores per pixel)
            const uint32_t packed = px[0] | (px[1] << 8) | (px[2] << 16);
            const int n_bytes = 3 * n_px;
            const int dh = (int)((0u - (uint32_t)reinterpret_cast<uintptr_t>(row)) & 3u);      // head bytes
            const int nf = (n_bytes - dh) >> 2, tl = (n_bytes - dh) & 3, k = lane & 3;
            const int sj = lane < 48 ? 4 * lane + dh : (lane < 52 ? k : dh + 4 * nf + k);     // first stream byte of this lane's piece
            const int a = (sj * 21846) >> 16, b = sj - 3 * a;                                 // pixel, byte in it
            const int qa = min(a, n_px - 1), qb = min(a + 1, n_px - 1);
            const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (reversed ? n_px - 1 - qa : qa), (int)packed);
            const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * (reversed ? n_px - 1 - qb : qb), (int)packed);
            const uint32_t dw = __builtin_amdgcn_perm(pb, pa, b == 0 ? 0x04020100u : (b == 1 ? 0x05040201u : 0x06050402u));
            if (lane < nf) __builtin_nontemporal_store(dw, reinterpret_cast<uint32_t*>(__builtin_assume_aligned(row + sj, 4)));
            else if ((lane >= 48 && lane < 52 && k < dh) || (lane >= 52 && lane < 56 && k < tl)) row[sj] = (uint8_t)dw;
            return;
        }
    }
    const int pos = reversed ? n_px - 1 - lane : lane;
    if constexpr (C == 4) {
        if (aligned4) {
            if (lane < n_px) reinterpret_cast<uint32_t*>(row)[pos] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
            return;
        }
    }
    if (lane < n_px && !(skip_first && pos == 0))
        for (int c = 0; c < C; ++c) row[pos * C + c] = (uint8_t)px[c];
}

// ---- cubic variant of the equirect sampler (4x4 Keys taps, OpenCV fixed-point table) -------------------------------
// Two steps, like the bilinear fetch: cubic_issue_rgb puts the 4 row reads of one RGB pixel in flight (12 contiguous bytes
// each, fetched as a dword-aligned 16-byte read) without control flow and without touching a result; eq_cubic_blend shifts
// the rows into place, reads the 32-byte weight entry (LDS: short latency, so it need not occupy 8 registers while the
// gathers fly) and blends.  Lanes whose window touches the seam or the last columns are flagged and redone by eq_cubic_slow.
struct EqCubicTaps {
    uint32_t raw[4][4];   // the four aligned dwords of each window row, as loaded
    uint32_t sh;          // byte offset of the window inside the aligned read (the same for all four rows)
    int phase;            // fy * 32 + fx
    bool fix;
};

// window anchored at texel (ix, iy) with phase (fx, fy); `fix` = the window's columns could not be read in place.
// `stride4` (wave-uniform): the row stride is a multiple of 4, so all four rows start at the same misalignment (`sh` = that one
// value; otherwise four 2-bit fields).  Every read is src + a 32-bit lane offset: scalar base + vector offset addressing, no
// 64-bit vector adds.
__device__ __forceinline__ EqCubicTaps cubic_issue_rgb(const uint8_t* __restrict__ src, uint32_t stride, bool stride4, int W, int H,
                                                       int ix, int iy, int fx, int fy) {
    EqCubicTaps t;
    const int x0 = clamp0_uniform(ix - 1, W - 6);           // 16-byte aligned read of 12 tap bytes stays in-row
    t.fix = (x0 != ix - 1);
    t.phase = fy * 32 + fx;
    const uint32_t col = (uint32_t)x0 * 3u;
    uint32_t offs[4];                                       // the branches meet on 32-bit offsets, not on pointers
    if (stride4) {
        const uint32_t o = ((uint32_t)reinterpret_cast<uintptr_t>(src) + col) & 3u;
        t.sh = o;
        const uint32_t cb = col - o;
        // Common case (wave-uniform test): no window of the wavefront touches the first or the last image row -- the four rows
        // are off0 + k * stride (the kernel is arithmetic-bound).
        if (!any_lane(iy < 1 || iy > H - 3)) {
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
        t.raw[ky][0] = q[0]; t.raw[ky][1] = q[1]; t.raw[ky][2] = q[2]; t.raw[ky][3] = q[3];
    }
    return t;
}

// The launch constants the samplers need, as plain values (see the ring-member loop of eq_views_kernel: they are handed over
// behind an optimisation barrier so that they stay in registers instead of being re-read from the kernel argument).
struct EqSrc {
    int W, H;
    int64_t src_stride, mask_stride;
    bool stride4;        // src_stride % 4 == 0
};

__device__ __forceinline__ EqCubicTaps eq_cubic_fetch(const EqSrc& L, const uint8_t* __restrict__ src, int sx, int sy) {
    return cubic_issue_rgb(src, (uint32_t)L.src_stride, L.stride4, L.W, L.H, sx >> 5, sy >> 5, sx & 31, sy & 31);
}

// 48 multiply-adds per pixel as 24 v_dot2_i32_i16 (see eq_blend): constant selectors -- the 12 tap bytes of a row are
// b0..b11, channel c owns b[c], b[3+c], b[6+c], b[9+c] -- and the table already stores the weights as int16 pairs.
// `wtab` is the workgroup's LDS copy of the table (cubic_lds_fill / cubic_lds_weights).

// The LDS copy of the 32 x 32-phase weight table is kept as TWO half tables -- window rows 0-1 of every phase (16 bytes each), then
// rows 2-3 -- instead of 1024 entries of 32 bytes: a lane's two 16-byte reads then collide with another lane's only when their
// phases differ by a multiple of 16 instead of 8 (round 3: 59 % of the cubic kernels' LDS-active cycles were bank conflicts).
__device__ __forceinline__ void cubic_lds_weights(const int16_t* wtab, int phase, uint32_t (&wpk)[8]) {
    const uint4* wq = reinterpret_cast<const uint4*>(wtab);
    const uint4 wa = wq[phase], wb = wq[1024 + phase];
    wpk[0] = wa.x; wpk[1] = wa.y; wpk[2] = wa.z; wpk[3] = wa.w; wpk[4] = wb.x; wpk[5] = wb.y; wpk[6] = wb.z; wpk[7] = wb.w;
}
// fill: thread t copies 16-byte piece i of the global table ([phase][2] pieces) to its place in the LDS layout
__device__ __forceinline__ void cubic_lds_fill(int16_t* s_wtab, const int16_t* g_tab, int n_threads) {
    const uint4* g = reinterpret_cast<const uint4*>(g_tab);
    uint4* l = reinterpret_cast<uint4*>(s_wtab);
    for (int i = threadIdx.x; i < 2048; i += n_threads) {
        l[(i & 1) * 1024 + (i >> 1)] = g[i];
    }
}

template <bool ONE_SHIFT>
__device__ __forceinline__ void eq_cubic_rows(const EqCubicTaps& t, const uint32_t (&wpk)[8], int (&acc)[3]) {
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint32_t o = ONE_SHIFT ? t.sh : ((t.sh >> (2 * ky)) & 3u);
        const uint32_t d0 = __builtin_amdgcn_alignbyte(t.raw[ky][1], t.raw[ky][0], o);
        const uint32_t d1 = __builtin_amdgcn_alignbyte(t.raw[ky][2], t.raw[ky][1], o);
        const uint32_t d2 = __builtin_amdgcn_alignbyte(t.raw[ky][3], t.raw[ky][2], o);
        const uint32_t w01 = wpk[2 * ky], w23 = wpk[2 * ky + 1];
        // perm(a, b, sel): bytes 0..3 come from b, 4..7 from a.  Row 0 starts the three chains from the rounding constant.
        const uint32_t p0 = __builtin_amdgcn_perm(d0, d0, GS360_PAIR(0, 3));             // b0, b3
        const uint32_t p1 = __builtin_amdgcn_perm(d1, d0, GS360_PAIR(1, 4));             // b1 = d0.1, b4 = d1.0
        const uint32_t p2 = __builtin_amdgcn_perm(d1, d0, GS360_PAIR(2, 5));             // b2 = d0.2, b5 = d1.1
        if (ky == 0) {
            acc[0] = dot2_i16_from(p0, w01, 1 << 14);
            acc[1] = dot2_i16_from(p1, w01, 1 << 14);
            acc[2] = dot2_i16_from(p2, w01, 1 << 14);
        } else {
            acc[0] = dot2_i16(p0, w01, acc[0]);
            acc[1] = dot2_i16(p1, w01, acc[1]);
            acc[2] = dot2_i16(p2, w01, acc[2]);
        }
        acc[0] = dot2_i16(__builtin_amdgcn_perm(d2, d1, GS360_PAIR(2, 5)), w23, acc[0]);            // b6 = d1.2, b9 = d2.1
        acc[1] = dot2_i16(__builtin_amdgcn_perm(d2, d1, GS360_PAIR(3, 6)), w23, acc[1]);            // b7 = d1.3, b10 = d2.2
        acc[2] = dot2_i16(__builtin_amdgcn_perm(d2, d2, GS360_PAIR(0, 3)), w23, acc[2]);            // b8 = d2.0, b11 = d2.3
    }
}

__device__ __forceinline__ void eq_cubic_blend(const EqCubicTaps& t, const int16_t* wtab, bool stride4, uint32_t (&out)[4]) {
    uint32_t wpk[8];
    cubic_lds_weights(wtab, t.phase, wpk);
    int acc[3];                                           // (sum + 2^14) >> 15: the chains start at 2^14
    if (stride4) eq_cubic_rows<true>(t, wpk, acc);        // wave-uniform
    else eq_cubic_rows<false>(t, wpk, acc);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (uint32_t)min(max(acc[c] >> 15, 0), 255);
}


// the taps of one bilinear pixel (two rows, raw or shifted into place) and their exact-integer blend; shared with cv_blend_fast
template <int C>
struct EqTaps {
    uint2 t0, t1;   // raw bytes of rows y0 / y1 starting at column ix
    RowsRaw raw;    // C == 3: the loads in flight (eq_taps_finish turns them into t0 / t1)
    bool fix;       // needs the slow (wrapping) path
};

template <int C>
__device__ __forceinline__ void eq_blend_f(const EqTaps<C>& t, const int fx, const int fy, uint32_t (&out)[4]);
template <int C>
__device__ __forceinline__ void eq_blend(const EqTaps<C>& t, int sx, int sy, uint32_t (&out)[4]) {
    eq_blend_f<C>(t, sx & 31, sy & 31, out);
}
template <int C>
__device__ __forceinline__ void eq_blend_f(const EqTaps<C>& t, const int fx, const int fy, uint32_t (&out)[4]) {   // fx, fy in [0, 31]
    // (sum S a b + 512) >> 10 with a in {32-fx, fx}, b in {32-fy, fy}: the weights of one row, a0 b | (a1 b) << 16, are
    // one multiply of the packed horizontal pair (a1 b <= 1024 cannot carry into the upper half)
    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);             // < 2^22
    const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
    if constexpr (C == 3) {          // row bytes: r0 g0 b0 r1 | g1 b1 . .
        uint32_t px[3];
        blend_rgb_rows(t.t0, t.t1, fx, fy, px);
        out[0] = px[0]; out[1] = px[1]; out[2] = px[2];
    } else if constexpr (C == 4) {   // row bytes: r0 g0 b0 a0 | r1 g1 b1 a1
#pragma unroll
        for (int c = 0; c < 4; ++c)
            out[c] = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t.t1.y, t.t1.x, GS360_PAIR(c, 4 + c)), wr1,
                                        dot2_i16_from(__builtin_amdgcn_perm(t.t0.y, t.t0.x, GS360_PAIR(c, 4 + c)), wr0, 512)) >> 10;
    } else {                         // row bytes: v0 v1
        out[0] = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t.t1.x, t.t1.x, GS360_PAIR(0, 1)), wr1,
                                    dot2_i16_from(__builtin_amdgcn_perm(t.t0.x, t.t0.x, GS360_PAIR(0, 1)), wr0, 512)) >> 10;
    }
}


}  // namespace gs360
