// gs360_srcmajor.hip -- source-major equirect kernel for calls whose views are yaw rings: BASELINE cfg2 (8K -> 6 x 800^2), the `default`
// preset's ring of eight, `full360coverage`'s level ring of four with its +30 / -30 rings (PC:616-680, PC:794-822).
//
// The gather kernels (gs360_kernels.hip) fetch, per view, every 128-byte source line a view's taps touch: at cfg2's 4.6 source texels per
// output pixel that is 718 k lines per 8K frame for six views whose UNION is 413 k lines (a view uses 6 of every 14 bytes of a line and
// neighbouring views of a ring overlap by half their field).  Here the work is cut along the SOURCE instead: a workgroup owns a tile of
// the panorama (a box of ~768 bytes x 16-32 rows), streams it ONCE into LDS with global_load_lds_dwordx4 and renders, for every view that
// looks at it, the output pixels whose tap pair starts inside it -- from a static plan.  Replaces the same call sites as the gather
// kernel (one `ffmpeg -vf v360` process per (frame, view), gs360_360PerspCut.py:310-314), results bit-identical: coordinates come from
// the EQ-SPEC functions themselves (eq_plan_coords_kernel), the blend is the gather kernels' blend.
//
// What makes one plan serve a whole call:
//   * a ring of N equally spaced views of one pitch is periodic in the source: the member at position q sees what the member at
//     position 0 sees, q d texels further, d = W / N (x0i32 differs by whole multiples of 32 d, everything else is equal), so the plan
//     lists only period 0 -- source bytes [0, 3 d) of every row -- with the member index relative to the period; period k renders member
//     (rel + k) mod N;
//   * latitude mirror: pixel (i, j) of a view at pitch p and pixel (i, h - 1 - j) of the view at pitch -p on the same yaw look at
//     mirrored points, exactly (sy' = 32 H - 32 - sy: rint is odd, no additive constant inside it; the pitch enters as +-sp).  The plan
//     lists the quads that look at or above the equator; the same entries with the tile's rows copied in reverse order and the output
//     row mirrored render the others -- of the same ring when it is level, of the ring at minus its pitch otherwise.
// A tile therefore has 2 N images; a workgroup walks G of them (two alternating LDS buffers: one loader wavefront copies image g + 1
// while eight consumer wavefronts render image g), with the tile's plan entries LDS-resident for all of them.  A call may hold several
// rings of ONE size (sm_eligible): every pitched ring must come with its mirror ring.
//
// Ownership is per output QUAD (four pixels = 12 bytes = three dwords, all rendered by the tile that holds the first pixel's taps), so
// every store is a dword store and no output byte is written twice.  Plan entry: per quad a header (3 x column | row << 14 | view << 26), per
// pixel (LDS byte offset of the top-left tap | fx << 17 | fy << 22).
//
// Masked calls (gs360_equirect_views_masked_u8: BASELINE config 5's fused keep-mask) stage the keep BITS of a tile box next to its texels
// (the nearest texel of a pixel is one of its four taps); their plans hold box row and byte of a tap apart so that the bit position
// follows from the pixel word.  cfg3 + mask 98-104 -> 81-92 us per frame (profiles/r05/srcmajor_masked.txt).
//
// Measured on MI355X (profiles/r05/): cfg2 14.1-15.5 us per frame against 18.7-19.6 for the gather kernel -- bound by the copies (the L1's
// ~64 read requests per CU in flight at ~1,000 cycles each; 70 MB per frame move where the union of lines + the stores is 64 MB); cfg3
// 58-63 us against 75-79 (LDS-staged kernel), cfg1 32-36 against 41-48: copies, stores, LDS and vector ALU each 55-70 % busy.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "gs360_blend.h"
#include "gs360_eqspec.h"
#include "gs360_kernels.h"

namespace gs360 {

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;

constexpr int kSmConsumers = 8;                         // consumer wavefronts per workgroup (+ one loader).  12: +14 .. +25 %, 15: no change
                                                        // (cfg1 / cfg2 / cfg3, profiles/r05/srcmajor/run32_consumer_waves.log)
constexpr int kSmMaxImages = 12;                        // images of a tile one workgroup walks (G)

// ---- plan coordinates: EQ-SPEC v1 for the first `rows` rows of view `vi` (a ring's reference member), every column by the general formula
__global__ __launch_bounds__(256) void eq_plan_coords_kernel(const EqLaunch L, int vi, int2* __restrict__ out, int rows) {
    const EqView& V = L.view[vi];
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    const int ic = min(i, V.out_w - 1);                  // (every lane evaluates: eq_sqrt's fallback branch is wave-uniform)
    const float x = (float)(2 * ic + 1 - V.out_w) * V.sxu;
    const float yv = (float)(2 * j + 1 - V.out_h) * V.syv;
    const float bz = __builtin_fmaf(V.sp, yv, V.cp);
    const float cy = __builtin_fmaf(-V.cp, yv, V.sp);
    const float h = eq_sqrt(__builtin_fmaf(x, x, bz * bz));
    int Kl, Kt;
    const float rl = eq_atan2_red(x, bz, Kl);
    const float rt = eq_atan2_red<true>(cy, h, Kt);
    const int sx = eq_quant_lon(rl, Kl, L, V);
    const int sy = L.y0i32 - Kt * 8 * L.H - (int)__builtin_rintf(rt * L.ky32);
    if (i < V.out_w && j < rows) out[(size_t)j * V.out_w + i] = make_int2(sx, sy);
}

struct SmArgs {
    const uint8_t* src[GS360_MAX_FRAMES];
    uint8_t* dst[GS360_MAX_FRAMES * GS360_MAX_VIEWS];
    const SmTile* tiles;
    const uint32_t* entries;
    int32_t W, H, N, NV, w, h, PB;        // N = members per ring (periods of the source), NV = views of the call (rings x N)
    int32_t G, groups_per_tile, groups_per_frame, total_groups, gchunk;
    int32_t buf_bytes, ent_bytes;
    int32_t qmap[GS360_MAX_VIEWS];        // ring * N + ring position -> view index of the call
    int32_t partner[GS360_MAX_VIEWS];     // ring -> the ring its upside-down images render (itself: level; the ring at minus its pitch otherwise)
    int64_t src_stride, dst_stride;
    // masked calls: the frames' keep-BIT images (mask_pack_kernel: bit x of row y, mask_stride bytes per row, W / 32 dwords carry the W bits)
    const uint8_t* mask[GS360_MAX_FRAMES];
    int32_t mask_stride, mask_dw, mbuf_bytes, pad_;
};
static_assert(sizeof(SmArgs) <= 4096, "SmArgs travels as a kernel argument");

constexpr int kSmMaskedPitch = 1024;                    // masked plans: at most this many bytes per box row (the pixel word holds row and byte apart)
constexpr int kSmMaskRowBytes = 64;                     // ... and per row of the keep-bit box (16 dwords = 512 texels)

// RS ("register staging", unmasked calls whose boxes are at most kSmRsRows x CW rows of at most 1 KiB): NO loader wavefront and no LDS
// copies -- every consumer wavefront loads its rows of image g + 1 (16 bytes per lane, one box row per instruction) into registers before it
// renders image g and writes them to the other buffer afterwards.  A CU takes LDS copies (global_load_lds) at ~25 GB/s whatever issues them
// (profiles/r06/table_stage/README.md) -- cfg2's copies alone are at 22.6 GB/s per CU -- ordinary loads go through the L1 at 64 B per cycle.
constexpr int kSmRsRows = 5;

template <int CW, bool MASKED, bool RS>
__global__ __launch_bounds__(64 * (CW + (RS ? 0 : 1))) void eq_srcmajor_kernel(const SmArgs P) {
    constexpr int NL = RS ? 0 : 1;                       // loader wavefronts in front of the consumers
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    __shared__ uint8_t* s_dst[kSmMaxImages * GS360_MAX_VIEWS];
    // XCD-aware order: XCD x (= block % 8) walks a contiguous chunk of the (frame, tile, image group) order, so the images of a tile,
    // the neighbouring tiles (whose boxes share halo lines) and the tiles that complete each other's output lines meet in one L2
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = P.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    uint8_t* const s_ent = s_lds;                        // [headers: nq dwords][pixel words: 4 nq dwords]
    uint8_t* const s_tile = s_lds + P.ent_bytes;         // two tile buffers of buf_bytes
    uint8_t* const s_mask = s_tile + 2 * P.buf_bytes;    // (MASKED) two keep-bit boxes of mbuf_bytes
    if (tid < G * P.NV) {                                // destination of (image, plan view = ring * N + relative member): period k renders
        const int g = tid / P.NV, v = tid - g * P.NV;    // member (rel + k) mod N, of the ring itself or -- upside down -- of its mirror ring
        const int c = v / P.N, img = g0 + g;
        int q = v - c * P.N + (img >> 1);
        if (q >= P.N) q -= P.N;
        const int c2 = (img & 1) ? P.partner[c] : c;
        s_dst[g * GS360_MAX_VIEWS + v] = P.dst[f * P.NV + P.qmap[c2 * P.N + q]];
    }
    const uint8_t* __restrict__ src = P.src[f];
    const int rowbytes = 3 * P.W;
    const int pitch = T.wch * 16;
    const int nq = T.nq;
    // image = period * 2 + flip.  One copy instruction per (box row, block of 64 chunks); the exec mask is set ONCE per block of chunks:
    // a mask that changes from row to row makes the wavefront wait for each copy's address phase (measured: 2-3x the issue time).
    auto dma = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * P.PB;                  // < rowbytes
        for (int cb = 0; cb < T.wch; cb += 64) {
            int x = xk + (cb + lane) * 16;               // the box may run across the 360-degree seam
            if (x >= rowbytes) x -= rowbytes;
            if (cb + lane < T.wch) {
                int y = flip ? P.H - 1 - T.y0 : T.y0;    // a flipped image takes the mirrored rows in reverse order
                const int ystep = flip ? -1 : 1;
                for (int row = 0; row < T.nrows; ++row, y += ystep) {
                    const int yc = min(max(y, 0), P.H - 1);                 // EQ-SPEC clamps tap rows to [0, H - 1]
                    const uint8_t* rowp = src + (size_t)yc * P.src_stride;
                    __builtin_amdgcn_global_load_lds((global_void_t*)(rowp + (uint32_t)x), (lds_void_t*)(buf + row * pitch + cb * 16), 16, 0, 0);
                }
            }
        }
        if constexpr (MASKED) {
            // the keep bits of the box: the nearest texel of a pixel is one of its four taps, so rows [y0, y0 + nrows) x the box's texels
            // hold every bit this image can ask for.  Four box rows per copy instruction (a dword per lane, 16 dwords per row); the bit
            // image wraps at W bits = W / 32 dwords (W % 32 == 0), the box's first bit sits at (x0 / 3) % 32 in every period (d % 32 == 0).
            uint8_t* const mbuf = s_mask + (buf != s_tile ? P.mbuf_bytes : 0);
            const uint8_t* __restrict__ mk = P.mask[f];
            const int mdw = T.pad1;                      // dwords per box row that hold its bits
            int d = ((T.x0 / 3 + k * (P.PB / 3)) >> 5) + (lane & 15);
            if (d >= P.mask_dw) d -= P.mask_dw;
            if ((lane & 15) < mdw) {
                const int r4 = lane >> 4;
                for (int row0 = 0; row0 < T.nrows; row0 += 4) {               // (the buffer holds whole groups of four rows)
                    const int y = flip ? P.H - 1 - T.y0 - (row0 + r4) : T.y0 + row0 + r4;
                    const int yc = min(max(y, 0), P.H - 1);
                    __builtin_amdgcn_global_load_lds((global_void_t*)(mk + (size_t)yc * (size_t)P.mask_stride + (size_t)d * 4), (lds_void_t*)(mbuf + row0 * kSmMaskRowBytes), 4, 0, 0);
                }
            }
        }
    };
    // (RS) this wavefront's rows of an image, in registers between their loads and their LDS writes
    struct Stage { uint4 a, b, c, d, e; };
    static_assert(kSmRsRows == 5, "five staged rows per wavefront");
    auto rs_load = [&](const int img) {
        const int k = img >> 1;
        const bool flip = img & 1;
        int x = T.x0 + k * P.PB + lane * 16;             // the box may run across the 360-degree seam
        if (x >= rowbytes) x -= rowbytes;
        const uint8_t* const colp = src + (uint32_t)x;
        auto row_of = [&](const int i) {
            const int row = wave + CW * i;
            const int y = flip ? P.H - 1 - T.y0 - row : T.y0 + row;          // a flipped image takes the mirrored rows in reverse order
            const int yc = min(max(y, 0), P.H - 1);                         // EQ-SPEC clamps tap rows to [0, H - 1]
            return *reinterpret_cast<const uint4*>(__builtin_assume_aligned(colp + (size_t)yc * P.src_stride, 16));
        };
        Stage o;
        o.a = o.b = o.c = o.d = o.e = make_uint4(0u, 0u, 0u, 0u);
        if (lane < T.wch) {
            o.a = row_of(0);                             // (nrows >= CW for every tile the host sends here)
            if (wave + CW < T.nrows) o.b = row_of(1);
            if (wave + 2 * CW < T.nrows) o.c = row_of(2);
            if (wave + 3 * CW < T.nrows) o.d = row_of(3);
            if (wave + 4 * CW < T.nrows) o.e = row_of(4);
        }
        return o;
    };
    auto rs_store = [&](const Stage& o, uint8_t* const buf) {
        if (lane < T.wch) {
            uint8_t* const p = buf + wave * pitch + lane * 16;
            *reinterpret_cast<uint4*>(p) = o.a;
            if (wave + CW < T.nrows) *reinterpret_cast<uint4*>(p + CW * pitch) = o.b;
            if (wave + 2 * CW < T.nrows) *reinterpret_cast<uint4*>(p + 2 * CW * pitch) = o.c;
            if (wave + 3 * CW < T.nrows) *reinterpret_cast<uint4*>(p + 3 * CW * pitch) = o.d;
            if (wave + 4 * CW < T.nrows) *reinterpret_cast<uint4*>(p + 4 * CW * pitch) = o.e;
        }
    };
    if (wave == 0) {
        const uint8_t* ge = reinterpret_cast<const uint8_t*>(P.entries + T.eoff);
        const int eb = 20 * nq;                          // a multiple of 64 bytes (nq % 16 == 0)
        for (int o = 0; o < eb; o += 1024)
            if (o + lane * 16 < eb)
                __builtin_amdgcn_global_load_lds((global_void_t*)(ge + o + lane * 16), (lds_void_t*)(s_ent + o), 16, 0, 0);
        if constexpr (!RS) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): entries and first image have landed
    }
    if constexpr (RS) {
        rs_store(rs_load(g0), s_tile);
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wavefront's rows are in LDS
    }
    __builtin_amdgcn_s_barrier();
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint32_t* const e_hdr = reinterpret_cast<const uint32_t*>(s_ent);
    const uint32_t* const e_px = e_hdr + nq;
    const int npx = 4 * nq;
    const int dstride = (int)P.dst_stride;               // bytes per output row (a multiple of 4)
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        Stage nxt;
        nxt.a = nxt.b = nxt.c = nxt.d = nxt.e = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (RS) {
            if (g + 1 < G) nxt = rs_load(g0 + g + 1);
        }
        if (!RS && wave == 0) {
            if (g + 1 < G) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        } else {
            // Consumers: a pixel per lane (neighbouring lanes' tap windows are 14 bytes apart: a quad per lane would scatter them 55 bytes
            // apart, 4-way bank conflicts), no memory reads at all -- plan entries and taps come from LDS -- so nothing in this loop ever
            // waits for the memory queue the copies sit in; the only memory instruction is the store.
            // The loop is vector-ALU-bound for weakly minified rings (cfg1: 20 M pixels per frame), so it does per pixel only what differs
            // per pixel: the entry addresses are running pointers; a wavefront turn (16 quads) never mixes views (the plan pads view groups),
            // so the destination base is one scalar pair and the store takes the SGPR-base + 32-bit-offset form; row and column offsets are
            // two 24-bit multiply-adds with the image's flip folded into the scalars; the tap rows' LDS addresses differ by the scalar pitch
            // and v_alignbyte reads only the low two bits of its shift operand (the plan word itself serves for both rows).
            const bool flip = (g0 + g) & 1;
            const int row_step = flip ? -dstride : dstride;                                  // |.| < 2^22 (checked by the host)
            const int lane_off = (flip ? (P.h - 1) * dstride : 0) + 4 * min(k4, 2);           // flipped image: rows run upwards from h - 1
            int cur_vrel = -1;
            uint64_t dbase = 0;
            const uint8_t* const cur_mask = s_mask + (g & 1) * P.mbuf_bytes;
            const uint32_t msh = T.pad0 & 255u, x0m3 = T.pad0 >> 8;
            const int fy_up = flip ? 17 : 16;
            auto turn = [&](const uint8_t* const pp, const uint8_t* const hp) {
                const uint32_t pw = *reinterpret_cast<const uint32_t*>(pp);
                const uint32_t hd = *reinterpret_cast<const uint32_t*>(hp);
                const int fx = (pw >> 17) & 31, fy = (pw >> 22) & 31;
                // tap offset: unmasked plans hold it whole; masked ones box row (7 bits) and byte in the row (10 bits) apart, for the keep lookup
                const uint32_t xbyte = pw & (kSmMaskedPitch - 1), rho = (pw >> 10) & 127u;
                const uint32_t toff = MASKED ? (__umul24(rho, (uint32_t)pitch) + xbyte) & ~3u : pw & 0x1fffcu;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + toff);
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + toff + pitch);
                const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
                uint2 t0, t1;                            // rows iy, iy + 1: bytes r0 g0 b0 r1 | g1 b1 . .
                t0.x = __builtin_amdgcn_alignbyte(a1, a0, pw); t0.y = __builtin_amdgcn_alignbyte(a2, a1, pw);
                t1.x = __builtin_amdgcn_alignbyte(b1, b0, pw); t1.y = __builtin_amdgcn_alignbyte(b2, b1, pw);
                uint32_t px[3];
                blend_rgb_rows(t0, t1, fx, fy, px);
                uint32_t pk;                             // r | g << 8 | b << 16 in two instructions (the compiler prefers two shifts and a three-way or)
                asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(pk) : "v"(px[1]), "v"(px[0]));
                asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(pk) : "v"(px[2]), "v"(pk));
                if constexpr (MASKED) {
                    // keep bit of the NEAREST texel ((sx + 16) >> 5, (sy + 16) >> 5): column = the tap's texel + (fx >= 16); row = the tap row
                    // + (fy >= 16) -- in an upside-down image + (fy > 16): there the mirrored coordinate 32 H - 32 - sy is what gets rounded,
                    // and exactly half-way both round up.
                    const uint32_t tx = __umul24(xbyte + x0m3, 0xAAABu) >> 17;          // / 3, exact below 2^16
                    const uint32_t bitpos = msh + tx + (fx >= 16 ? 1u : 0u);
                    const uint32_t rr = rho + (fy >= fy_up ? 1u : 0u);
                    const uint32_t kw = *reinterpret_cast<const uint32_t*>(cur_mask + rr * kSmMaskRowBytes + ((bitpos >> 5) << 2));
                    pk &= (uint32_t)__builtin_amdgcn_sbfe((int)kw, bitpos & 31u, 1u);         // 0 or all ones
                }
                // lanes 4m .. 4m + 3 hold a quad: lane k cuts dword k of its 12 bytes out of pixels k and k + 1
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, true);          // quad_perm [1,2,3,3]
                const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                // the turn's view (wave-uniform by construction of the plan): its destination base is reloaded only when the view changes
                const int vrel = __builtin_amdgcn_readfirstlane((int)(hd >> 26));
                if (vrel != cur_vrel) {                  // scalar compare + branch: a tile has two or three view groups
                    const uint2 dq = *reinterpret_cast<const uint2*>(&s_dst[g * GS360_MAX_VIEWS + (vrel & 15)]);
                    dbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)dq.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)dq.x);
                    cur_vrel = vrel;
                }
                // header: 3 x column | row << 14 | view << 26
                const uint32_t off = (hd & 0x3fffu) + (uint32_t)(__mul24((int)((hd >> 14) & 0xfffu), row_step) + lane_off);
                // lane 3 repeats lane 2's store (same dword, same address): an UNCONDITIONAL store keeps the loop body straight-line
                const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, true);           // quad_perm [0,1,2,2]
                *(__attribute__((address_space(1))) uint32_t*)(dbase + off) = dwq;                 // global store, scalar base + 32-bit offset
            };
            // two turns per trip: the second one's entry addresses are immediate offsets of the first one's
            const uint8_t* pxp = reinterpret_cast<const uint8_t*>(e_px) + ((wave - NL) * 64 + lane) * 4;
            const uint8_t* hdp = reinterpret_cast<const uint8_t*>(e_hdr) + (((wave - NL) * 64 + lane) >> 2) * 4;
            int i0 = (wave - NL) * 64;
            for (; i0 + 64 * CW < npx; i0 += 128 * CW, pxp += 512 * CW, hdp += 128 * CW) {
                turn(pxp, hdp);
                turn(pxp + 256 * CW, hdp + 64 * CW);
            }
            if (i0 < npx) turn(pxp, hdp);
        }
        if constexpr (RS) {
            if (g + 1 < G) rs_store(nxt, s_tile + ((g + 1) & 1) * P.buf_bytes);
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the rows are in LDS before anybody passes the barrier
        }
        __builtin_amdgcn_s_barrier();                    // image g + 1 has landed AND every consumer is done with image g's buffer
    }
}

}  // namespace

// ---- host side: the plan ------------------------------------------------------------------------------------------------------------
struct SmPlan {
    // key
    int W = 0, H = 0, N = 0, n_rings = 0, w = 0, h = 0, Bx = 0, R = 0;
    bool masked = false;                   // plan of masked calls: fixed LDS pitch, keep-bit boxes
    uint32_t sxu = 0, syv = 0;             // float bits
    uint32_t ring_key[GS360_MAX_VIEWS][4]; // per ring: sp, cp, x0f32 (float bits), x0i32 of its reference member
    // contents
    SmTile* d_tiles = nullptr;
    uint32_t* d_entries = nullptr;
    int n_tiles = 0, buf_bytes = 0, ent_bytes = 0, mbuf_bytes = 0, PB = 0;
    int rows = 0;                          // tile rows the builder ended with (R, or a half / quarter of it when R did not fit)
    int max_wch = 0, max_nrows = 0, min_nrows = 1 << 30;   // over the plan's tiles (the register-staging kernel takes rows of <= 64 chunks, kSmConsumers .. 5 kSmConsumers rows)
    int box_pct = 0;                       // bytes of all tile boxes in percent of the tile grid cells they stand for (halos, cut tiles)
    uint64_t stamp = 0;
    int pins = 0;                          // calls that hold this plan between sm_prepare and sm_release (guarded by the cache's lock): never evicted
};

void sm_plan_free(SmPlan* p) {
    if (!p) return;
    if (p->d_tiles) (void)hipFree(p->d_tiles);
    if (p->d_entries) (void)hipFree(p->d_entries);
    delete p;
}

namespace {

uint32_t fbits(float v) { uint32_t b; std::memcpy(&b, &v, 4); return b; }

struct Quad { int32_t tid, vslot, j, i0; int32_t xr[4], iy[4], ph[4]; };

constexpr int kSmQuadCap = 1632;                        // quads of one plan tile (32 KiB of entries); a denser tile is cut into several

// ---- plan building on the GPU: the quads of a geometry, generated and ordered on the device ---------------------------------------------
// Round 5 copied the coordinates back and generated, sorted and tiled ~1 M quads on the calling thread (cfg3: 0.13-0.2 s).  Now the quads are
// generated on the device in the host loop's own order (ring, row, column: a block per row, ranks by a block-wide prefix sum), their sort
// keys (tile << 5 | view slot) go through hipCUB's stable radix sort, the quads are gathered into that order and come back ONCE, ordered;
// only the tile assembly (boxes, cut tiles, padded view groups) remains on the host.
// A quad belongs to the plan when its first pixel looks at or above the equator (sy <= 16 H - 16: the upside-down images then render exactly
// the others -- sy' = 32 H - 32 - sy is the mirror ring's pixel (i, h - 1 - j); quads ON the equator are rendered twice, same bytes to the
// same place; for a level ring that is the upper half of its rows).
__global__ __launch_bounds__(256) void sm_quad_count_kernel(const int2* __restrict__ xy, const int w, const int nqx, const int centre, int* __restrict__ rowcnt) {
    __shared__ int s_n;
    const int j = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int n = 0;
    for (int qx = threadIdx.x; qx < nqx; qx += 256) n += xy[(size_t)j * w + 4 * qx].y <= centre ? 1 : 0;
    atomicAdd(&s_n, n);
    __syncthreads();
    if (threadIdx.x == 0) rowcnt[j] = s_n;
}
// exclusive prefix over the rows of one ring (fewer than 4096 rows: one block, 16 rows per thread), continuing at *total
__global__ __launch_bounds__(256) void sm_row_scan_kernel(const int* __restrict__ rowcnt, const int rows, int* __restrict__ rowoff, int* __restrict__ total) {
    __shared__ int s_sum[256];
    const int t = (int)threadIdx.x, j0 = 16 * t;
    int cnt[16], mine = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { cnt[k] = j0 + k < rows ? rowcnt[j0 + k] : 0; mine += cnt[k]; }
    s_sum[t] = mine;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                  // inclusive scan of the threads' sums
        const int v = t >= d ? s_sum[t - d] : 0;
        __syncthreads();
        s_sum[t] += v;
        __syncthreads();
    }
    int acc = *total + s_sum[t] - mine;
#pragma unroll
    for (int k = 0; k < 16; ++k) { if (j0 + k < rows) rowoff[j0 + k] = acc; acc += cnt[k]; }
    __syncthreads();                                     // (every thread has read *total)
    if (t == 255) *total = acc;
}
// flags[0] = smallest tap row over the plan's quads (atomicMin), flags[1] = a quad that is not monotone in longitude was seen (a view over a pole)
__global__ __launch_bounds__(256) void sm_quad_emit_kernel(const int2* __restrict__ xy, const int w, const int nqx, const int centre, const int c, const int N,
                                                           const int PB, const int rowbytes, const int* __restrict__ rowoff, Quad* __restrict__ out,
                                                           int* __restrict__ flags) {
    __shared__ int s_wave[4], s_base;
    const int j = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = rowoff[j];
    __syncthreads();
    for (int q0 = 0; q0 < nqx; q0 += 256) {
        const int qx = q0 + (int)threadIdx.x;
        const int2* p = &xy[(size_t)j * w + 4 * min(qx, nqx - 1)];
        const int2 p0v = p[0];
        const bool keep = qx < nqx && p0v.y <= centre;
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int base = s_base;
        for (int k = 0; k < wv; ++k) base += s_wave[k];
        if (keep) {
            Quad Q;
            const int xb0 = 3 * (p0v.x >> 5);            // in [0, 3 W): the period p0 = xb0 / PB is in [0, N)
            const int pp = xb0 / PB;
            Q.tid = 0;
            Q.vslot = c * N + (pp ? N - pp : 0);
            Q.j = j; Q.i0 = 4 * qx;
            int ymin = 1 << 30, bad = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int2 pk = p[k];
                int dx = 3 * (pk.x >> 5) - xb0;              // longitude grows with the column; unwrap across the seam
                if (dx < 0) dx += rowbytes;
                if (dx > rowbytes / 2) bad = 1;              // (not a monotone quad: a view over a pole; refuse rather than trust)
                Q.xr[k] = xb0 - pp * PB + dx;                // relative to the QUAD's period (may run past its end: the copy wraps)
                Q.iy[k] = pk.y >> 5;
                Q.ph[k] = (pk.x & 31) | ((pk.y & 31) << 5);
                ymin = min(ymin, Q.iy[k]);
            }
            out[base + before] = Q;
            atomicMin(&flags[0], ymin);
            if (bad) atomicOr(&flags[1], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void sm_quad_keys_kernel(const Quad* __restrict__ quads, const int n, const int ytop, const int R, const int Bx, const int ntx,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tid = ((quads[i].iy[0] - ytop) / R) * ntx + quads[i].xr[0] / Bx;
    keys[i] = ((uint32_t)tid << 5) | (uint32_t)quads[i].vslot;
    idx[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void sm_quad_gather_kernel(const Quad* __restrict__ quads, const uint32_t* __restrict__ idx, const int n, Quad* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = quads[idx[i]];
}

// Device and host scratch of the builder, kept by the cache between builds (a build holds scratch.mu: builds are serialised among
// themselves, never against launches); grows only.
bool sm_scratch_fit(SmScratch& sc, size_t dev_bytes, size_t host_bytes, hipError_t* herr) {
    if (sc.dev_cap < dev_bytes) {
        if (sc.dev) (void)hipFree(sc.dev);
        sc.dev = nullptr; sc.dev_cap = 0;
        if ((*herr = hipMalloc(&sc.dev, dev_bytes)) != hipSuccess) return false;
        sc.dev_cap = dev_bytes;
    }
    if (sc.host_cap < host_bytes) {
        // ordinary (pageable) host memory: on this platform a copy out of the device into a block that has been touched before runs at the
        // pinned rate (83 MiB in 1.56 ms either way), and hipHostMalloc of cfg3's 108 MB costs 15 ms of the first build + 8 ms at context
        // destruction (profiles/r06/alloc_probe.txt)
        std::free(sc.host);
        sc.host = nullptr; sc.host_cap = 0;
        if (!(sc.host = std::aligned_alloc(4096, (host_bytes + 4095) & ~(size_t)4095))) { *herr = hipErrorOutOfMemory; return false; }
        sc.host_cap = host_bytes;
    }
    return true;
}

struct SmQuads {                                         // the geometry's quads on the device (in the scratch block), generation order
    Quad* d_quads = nullptr;                             // [n]
    Quad* d_sorted = nullptr;                            // [n]
    uint32_t *d_keys = nullptr, *d_keys_out = nullptr, *d_idx = nullptr, *d_idx_out = nullptr;
    void* d_cub = nullptr;
    size_t cub_bytes = 0;
    int n = 0, ytop = 1 << 30;
};
// 0: collected; 1: the geometry does not fit (a view over a pole: quads that are not monotone in longitude); < 0: HIP error in *herr
int sm_collect_quads(const EqLaunch& L0, const SmShape& S, SmScratch& sc, hipStream_t s, SmQuads* out, hipError_t* herr) {
    const EqView& V = L0.view[0];
    const int W = L0.W, H = L0.H, N = S.N, w = V.out_w, h = V.out_h;
    const int PB = 3 * (W / N), rowbytes = 3 * W;
    const int nqx = w / 4;
    const int centre = 16 * H - 16;
    size_t cand = 0;                                     // candidate quads: an upper bound of the plan's
    for (int c = 0; c < S.n_rings; ++c) cand += (size_t)(L0.view[S.ref[c]].level ? (h + 1) / 2 : h) * nqx;
    size_t cub_bytes = 0;
    *herr = hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)cand, 0, 32, s);
    if (*herr != hipSuccess) return -1;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t b_xy = up((size_t)h * w * sizeof(int2)), b_rows = up((size_t)h * 4), b_q = up(cand * sizeof(Quad)), b_k = up(cand * 4), b_cub = up(cub_bytes);
    const size_t dev_bytes = b_xy + 2 * b_rows + 256 + 2 * b_q + 4 * b_k + b_cub;
    if (!sm_scratch_fit(sc, dev_bytes, b_q + b_k + 256, herr)) return -1;
    uint8_t* d = (uint8_t*)sc.dev;
    int2* d_xy = (int2*)d; d += b_xy;
    int* d_rowcnt = (int*)d; d += b_rows;
    int* d_rowoff = (int*)d; d += b_rows;
    int* d_flags = (int*)d; d += 256;                    // [0] ytop, [1] bad, [2] total
    out->d_quads = (Quad*)d; d += b_q;
    out->d_sorted = (Quad*)d; d += b_q;
    out->d_keys = (uint32_t*)d; d += b_k;
    out->d_keys_out = (uint32_t*)d; d += b_k;
    out->d_idx = (uint32_t*)d; d += b_k;
    out->d_idx_out = (uint32_t*)d; d += b_k;
    out->d_cub = d; out->cub_bytes = cub_bytes;
    const int init[3] = {1 << 30, 0, 0};
    *herr = hipMemcpyAsync(d_flags, init, sizeof(init), hipMemcpyHostToDevice, s);
    for (int c = 0; c < S.n_rings && *herr == hipSuccess; ++c) {
        const bool level = L0.view[S.ref[c]].level != 0;
        const int rows = level ? (h + 1) / 2 : h;
        hipLaunchKernelGGL(eq_plan_coords_kernel, dim3((w + 255) / 256, rows), dim3(256), 0, s, L0, S.ref[c], d_xy, rows);
        hipLaunchKernelGGL(sm_quad_count_kernel, dim3(rows), dim3(256), 0, s, d_xy, w, nqx, centre, d_rowcnt);
        hipLaunchKernelGGL(sm_row_scan_kernel, dim3(1), dim3(256), 0, s, d_rowcnt, rows, d_rowoff, d_flags + 2);
        hipLaunchKernelGGL(sm_quad_emit_kernel, dim3(rows), dim3(256), 0, s, d_xy, w, nqx, centre, c, N, PB, rowbytes, d_rowoff, out->d_quads, d_flags);
        *herr = hipGetLastError();
    }
    int flags[3] = {0, 0, 0};
    if (*herr == hipSuccess) *herr = hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, s);
    if (*herr == hipSuccess) *herr = hipStreamSynchronize(s);
    if (*herr != hipSuccess) return -1;
    out->ytop = flags[0];
    out->n = flags[2];
    return (flags[1] || flags[0] < 0 || flags[2] == 0) ? 1 : 0;      // (a view that reaches the pole row: the gather kernels' clamp path)
}

// 0: plan built; 1: these tiles do not fit the kernel (the caller tries smaller ones, or falls back to the gather kernels); < 0: HIP error in *herr
int sm_build_plan(const EqLaunch& L0, const SmShape& S, const SmQuads& QS, SmScratch& sc, int Bx, int R, bool masked, bool smallest, size_t lds_limit, hipStream_t s,
                  SmPlan** out, hipError_t* herr) {
    const int W = L0.W, N = S.N;
    const int PB = 3 * (W / N), rowbytes = 3 * W;
    const int ntx = (PB + Bx - 1) / Bx;
    const size_t nq_all = (size_t)QS.n;
    // Order (tile, view slot, row, column): the quads were generated ring by ring in (row, column) order, so a STABLE sort by
    // (tile << 5 | view slot) gives it; sorted quads and keys come back through the host block.
    const int nb = (int)((nq_all + 255) / 256);
    hipLaunchKernelGGL(sm_quad_keys_kernel, dim3(nb), dim3(256), 0, s, QS.d_quads, (int)nq_all, QS.ytop, R, Bx, ntx, QS.d_keys, QS.d_idx);
    size_t cub_bytes = QS.cub_bytes;
    *herr = hipcub::DeviceRadixSort::SortPairs(QS.d_cub, cub_bytes, (const uint32_t*)QS.d_keys, QS.d_keys_out, (const uint32_t*)QS.d_idx, QS.d_idx_out, (int)nq_all, 0, 32, s);
    if (*herr != hipSuccess) return -1;
    hipLaunchKernelGGL(sm_quad_gather_kernel, dim3(nb), dim3(256), 0, s, QS.d_quads, QS.d_idx_out, (int)nq_all, QS.d_sorted);
    Quad* const quads = (Quad*)sc.host;
    uint32_t* const keys = (uint32_t*)((uint8_t*)sc.host + (((nq_all * sizeof(Quad)) + 255) & ~(size_t)255));
    *herr = hipGetLastError();
    if (*herr == hipSuccess) *herr = hipMemcpyAsync(keys, QS.d_keys_out, nq_all * 4, hipMemcpyDeviceToHost, s);
    if (*herr == hipSuccess) *herr = hipStreamSynchronize(s);
    if (*herr != hipSuccess) return -1;
    auto tat = [&](const size_t i) -> int { return (int)(keys[i] >> 5); };
    auto vat = [&](const size_t i) -> int { return (int)(keys[i] & 31u); };
    // Too tall for this geometry?  (The test at the end of this function: most boxes cut in two or more.)  A tile is cut
    // ceil(padded quads / kSmQuadCap) times, which one pass over the ordered keys counts exactly: when that exceeds the limit the quads
    // are not even copied back (cfg3 tries 32 rows first and ends at 16).
    if (!smallest) {
        size_t boxes = 0, pieces = 0;
        for (size_t a = 0; a < nq_all;) {
            size_t b = a, padded = 0, run = 0;           // the tile's quads with every view group padded to whole turns of 16, as the loop below does
            while (b < nq_all && tat(b) == tat(a)) {
                ++run;
                if (b + 1 == nq_all || tat(b + 1) != tat(a) || vat(b + 1) != vat(b)) { padded += (run + 15) & ~(size_t)15; run = 0; }
                ++b;
            }
            ++boxes;
            pieces += (padded + kSmQuadCap - 1) / kSmQuadCap;
            a = b;
        }
        if (pieces > boxes + boxes / 4) return 1;
    }
    *herr = hipMemcpyAsync(quads, QS.d_sorted, nq_all * sizeof(Quad), hipMemcpyDeviceToHost, s);
    if (*herr == hipSuccess) *herr = hipStreamSynchronize(s);
    if (*herr != hipSuccess) return -1;
    auto qat = [&](const size_t i) -> const Quad& { return quads[i]; };
    std::vector<SmTile> tiles;
    std::vector<uint32_t> ent;
    std::vector<const Quad*> list;
    int buf_bytes = 0, ent_bytes = 0, mbuf_bytes = 0, n_boxes = 0;
    long long box_sum = 0;
    for (size_t a = 0; a < nq_all;) {
        size_t b = a;
        while (b < nq_all && tat(b) == tat(a)) ++b;
        // entries in (view, row, column) order, every VIEW GROUP padded to whole wavefront turns (16 quads = 64 pixels) with copies of its
        // last quad -- same values to the same addresses -- so that a turn never mixes views (the consumers keep the destination base in
        // scalar registers)
        list.clear();
        for (size_t q = a; q < b; ++q) {
            list.push_back(&qat(q));
            if (q + 1 == b || vat(q + 1) != vat(q))
                while (list.size() % 16) list.push_back(&qat(q));
        }
        ++n_boxes;
        // a tile many views look at closely (weak minification, or a pitched ring's rows near the pole) is cut into plan tiles of at most
        // kSmQuadCap quads, each with the box of ITS quads
        for (size_t c0 = 0; c0 < list.size(); c0 += kSmQuadCap) {
            const int nqp = (int)std::min<size_t>(kSmQuadCap, list.size() - c0);
            int xmin = 1 << 30, xmax = 0, ymin = 1 << 30, ymax = 0;
            for (int q = 0; q < nqp; ++q)
                for (int k = 0; k < 4; ++k) {
                    xmin = std::min(xmin, list[c0 + q]->xr[k]); xmax = std::max(xmax, list[c0 + q]->xr[k]);
                    ymin = std::min(ymin, list[c0 + q]->iy[k]); ymax = std::max(ymax, list[c0 + q]->iy[k]);
                }
            SmTile T;
            T.x0 = xmin & ~15;
            T.wch = (xmax + 6 - T.x0 + 15) / 16;
            T.y0 = ymin;
            T.nrows = ymax - ymin + 2;
            T.eoff = (int32_t)ent.size();
            T.nq = nqp;
            T.pad0 = T.pad1 = 0;
            const int pitch = T.wch * 16;
            if ((size_t)T.nrows * pitch >= (1u << 17) || T.x0 >= PB || T.wch * 16 > rowbytes) return 1;
            if (masked) {                                // first bit of the box in its first keep dword | first byte's offset in its texel; keep dwords per row
                if (T.wch * 16 > kSmMaskedPitch || T.nrows > 127) return 1;
                const int msh = (T.x0 / 3) & 31;
                T.pad0 = msh | ((T.x0 % 3) << 8);
                T.pad1 = (msh + (T.wch * 16 + 2) / 3 + 1 + 31) / 32;            // (taps reach one texel past a box byte's own: + 1)
                if (T.pad1 > kSmMaskRowBytes / 4) return 1;
                mbuf_bytes = std::max(mbuf_bytes, ((T.nrows + 3) & ~3) * kSmMaskRowBytes);
            }
            ent.resize(ent.size() + 5 * (size_t)nqp);
            uint32_t* hdr = ent.data() + T.eoff;
            uint32_t* px = hdr + nqp;
            for (int q = 0; q < nqp; ++q) {
                const Quad& Q = *list[c0 + q];
                hdr[q] = (uint32_t)(3 * Q.i0) | ((uint32_t)Q.j << 14) | ((uint32_t)Q.vslot << 26);
                for (int k = 0; k < 4; ++k)
                    px[4 * q + k] = (masked ? (uint32_t)(Q.xr[k] - T.x0) | ((uint32_t)(Q.iy[k] - T.y0) << 10)
                                            : (uint32_t)((Q.iy[k] - T.y0) * pitch + (Q.xr[k] - T.x0))) | ((uint32_t)Q.ph[k] << 17);
            }
            buf_bytes = std::max(buf_bytes, T.nrows * pitch);
            box_sum += (long long)T.nrows * T.wch * 16;
            ent_bytes = std::max(ent_bytes, 20 * nqp);
            tiles.push_back(T);
        }
        a = b;
    }
    buf_bytes = (buf_bytes + 63) & ~63;
    ent_bytes = (ent_bytes + 63) & ~63;
    if ((size_t)ent_bytes + 2 * (size_t)buf_bytes + 2 * (size_t)mbuf_bytes > lds_limit) return 1;
    // most boxes cut in two or more: the tile is too tall for this geometry (every cut copies much of the box again), a smaller one serves better
    if (!smallest && tiles.size() > (size_t)n_boxes + (size_t)n_boxes / 4) return 1;
    SmPlan* p = new (std::nothrow) SmPlan();
    if (!p) { *herr = hipErrorOutOfMemory; return -1; }
    p->n_tiles = (int)tiles.size(); p->buf_bytes = buf_bytes; p->ent_bytes = ent_bytes; p->mbuf_bytes = mbuf_bytes; p->PB = PB;
    p->box_pct = (int)(100 * box_sum / ((long long)n_boxes * std::min(Bx, PB) * R));
    for (const SmTile& T : tiles) { p->max_wch = std::max(p->max_wch, T.wch); p->max_nrows = std::max(p->max_nrows, T.nrows); p->min_nrows = std::min(p->min_nrows, T.nrows); }
    *herr = hipMalloc((void**)&p->d_tiles, tiles.size() * sizeof(SmTile));
    if (*herr == hipSuccess) *herr = hipMalloc((void**)&p->d_entries, ent.size() * 4 + 1024);      // (slack: the entry copy reads whole 16-byte chunks)
    if (*herr == hipSuccess) *herr = hipMemcpyAsync(p->d_tiles, tiles.data(), tiles.size() * sizeof(SmTile), hipMemcpyHostToDevice, s);
    if (*herr == hipSuccess) *herr = hipMemcpyAsync(p->d_entries, ent.data(), ent.size() * 4, hipMemcpyHostToDevice, s);
    if (*herr == hipSuccess) *herr = hipStreamSynchronize(s);     // the host vectors go out of scope
    if (*herr != hipSuccess) { sm_plan_free(p); return -1; }
    *out = p;
    return 0;
}

void sm_ring_key(const EqLaunch& L, const SmShape& S, uint32_t (*key)[4]) {
    for (int c = 0; c < S.n_rings; ++c) {
        const EqView& V = L.view[S.ref[c]];
        key[c][0] = fbits(V.sp); key[c][1] = fbits(V.cp); key[c][2] = fbits(V.x0f32); key[c][3] = (uint32_t)V.x0i32;
    }
}

}  // namespace

// Can this launch take the source-major kernel?  Checks only; builds nothing.  Fills *S: how the views fall into yaw rings.
// The views must be rings of ONE size N (N equally spaced members, every position taken) with one view geometry, each ring level or
// accompanied by the ring at minus its pitch on the same yaws (`full360coverage`: a level ring of four and the +30 / -30 pair;
// `fisheyelike`: five rings of two; PC:616-680, :794-822).
bool sm_eligible(const EqLaunch& L, int C, int esize, int interp, bool masked, SmShape* S) {
    if (C != 3 || esize != 1 || interp != GS360_INTERP_LINEAR) return false;
    if (masked && L.W % 32) return false;                // (the keep-bit image wraps at whole dwords; the ring period is checked below)
    const int NV = L.n_views;
    const EqView& V = L.view[0];
    if (NV < 2 || NV > GS360_MAX_VIEWS) return false;
    if (V.out_w % 4 || V.out_w >= 4096 || V.out_h >= 4096 || V.out_w < 8 || V.out_h < 2) return false;
    if ((3 * L.W) % 16 || L.src_stride % 16) return false;
    const int64_t dstride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * 3;
    if (dstride % 4 || dstride >= (1 << 22) || dstride * V.out_h >= ((int64_t)1 << 31)) return false;    // (24-bit multiply-adds form the row offsets)
    for (int f = 0; f < L.n_frames; ++f)
        if ((uintptr_t)L.src[f] & 15) return false;
    for (int i = 0; i < L.n_frames * NV; ++i)
        if ((uintptr_t)L.dst[i] & 3) return false;
    for (int k = 0; k < NV; ++k) {
        const EqView& A = L.view[k];
        if (A.fish || A.flip || fbits(A.sxu) != fbits(V.sxu) || fbits(A.syv) != fbits(V.syv) || A.out_w != V.out_w || A.out_h != V.out_h) return false;
    }
    // The largest ring size N the views fall into: a ring = the views of one pitch whose yaws differ by whole multiples of 360 / N degrees
    // (x0i32 by whole multiples of d = W / N texels, x0f32 equal), every one of its N positions taken once.  A member's position is
    // ABSOLUTE (x0i32 / 32 d), so a ring and its mirror ring number their members alike.
    for (int N = NV; N >= 2; --N) {
        if (NV % N || L.W % N || (3 * (L.W / N)) % 16 || (masked && (L.W / N) % 32)) continue;
        const int d32 = 32 * (L.W / N);
        int seen[GS360_MAX_VIEWS];
        S->N = N; S->n_rings = 0;
        bool ok = true;
        for (int k = 0; k < NV && ok; ++k) {
            const EqView& A = L.view[k];
            int c = 0;
            for (; c < S->n_rings; ++c) {
                const EqView& B = L.view[S->ref[c]];
                if (fbits(B.sp) == fbits(A.sp) && fbits(B.cp) == fbits(A.cp) && fbits(B.x0f32) == fbits(A.x0f32) && B.x0i32 % d32 == A.x0i32 % d32) break;
            }
            if (c == S->n_rings) {
                if (S->n_rings == NV / N) { ok = false; break; }
                S->ref[c] = k; seen[c] = 0; ++S->n_rings;
            }
            const int pos = A.x0i32 / d32;
            if (pos < 0 || pos >= N || (seen[c] >> pos & 1)) { ok = false; break; }
            seen[c] |= 1 << pos;
            S->qmap[c * N + pos] = k;
        }
        if (!ok || S->n_rings != NV / N) continue;           // (rings x N == NV and no position taken twice: every ring is full)
        // the mirror ring: a level ring mirrors itself (sp == +0.0f: make_eq_view's level form), a pitched one needs the ring at minus its pitch
        for (int c = 0; c < S->n_rings && ok; ++c) {
            const EqView& A = L.view[S->ref[c]];
            S->partner[c] = A.level ? c : -1;
            for (int e = 0; e < S->n_rings && !A.level; ++e) {
                const EqView& B = L.view[S->ref[e]];
                if (!B.level && fbits(B.cp) == fbits(A.cp) && fbits(B.sp) == (fbits(A.sp) ^ 0x80000000u) && fbits(B.x0f32) == fbits(A.x0f32) &&
                    B.x0i32 % d32 == A.x0i32 % d32) S->partner[c] = e;
            }
            if (S->partner[c] < 0) ok = false;
        }
        if (!ok) continue;
        for (int c = 0; c < S->n_rings; ++c) S->ref[c] = S->qmap[c * N];      // the member at position 0: the plan's coordinates are its coordinates
        return true;
    }
    return false;
}

namespace {

bool sm_key_matches(const SmPlan* p, const EqLaunch& L, const SmShape& S, int Bx, int R, bool masked, const uint32_t (*key)[4]) {
    const EqView& V = L.view[0];
    return p->W == L.W && p->H == L.H && p->N == S.N && p->n_rings == S.n_rings && p->w == V.out_w && p->h == V.out_h && p->Bx == Bx && p->R == R &&
           p->masked == masked && p->sxu == fbits(V.sxu) && p->syv == fbits(V.syv) && std::memcmp(p->ring_key, key, sizeof(key[0]) * S.n_rings) == 0;
}

// The plan of (launch geometry, tile shape) from the context's cache, built on a miss.  Called with `lk` held; the lock is RELEASED while a
// plan is built (coordinate, quad and sort kernels, one copy back, the host-side tile assembly: 1-3 ms for cfg2, 16-20 ms for cfg3 -- other
// slots' calls must not queue behind that) and two threads that miss on the same geometry at once both build, the second result is
// dropped.  Evicted and dropped plans go to the cache's graveyard: hipFree synchronises the device, so they are released where the caller
// waits for the device anyway (gs360_sync, context destruction).  Returns nullptr with *herr == hipSuccess for a geometry that does not
// fit (remembered as an empty plan, or every call would plan again), nullptr with *herr set on a HIP error.
SmPlan* sm_get_plan(const EqLaunch& L, const SmShape& S, SmCache& cache, std::unique_lock<std::mutex>& lk, int Bx, int R, bool masked, size_t lds_limit,
                    hipStream_t s, hipError_t* herr) {
    const EqView& V = L.view[0];
    const int N = S.N;
    uint32_t key[GS360_MAX_VIEWS][4];
    sm_ring_key(L, S, key);
    auto find = [&]() -> SmPlan* {
        for (SmPlan* p : cache.plans)
            if (sm_key_matches(p, L, S, Bx, R, masked, key)) return p;
        return nullptr;
    };
    SmPlan* plan = find();
    if (!plan) {
        lk.unlock();
        const auto t_build = std::chrono::steady_clock::now();
        int rr = R, rc = 1, built = R;
        SmPlan* fresh = nullptr;
        try {
            std::lock_guard<std::mutex> build_lock(cache.scratch.mu);      // (builds share the scratch blocks: one at a time; launches are not held up)
            SmQuads QS;                                  // coordinates and quads once; only keys, sort and tiling are redone for smaller tiles
            rc = sm_collect_quads(L, S, cache.scratch, s, &QS, herr);
            for (int attempt = 0; rc == 0; ++attempt) {
                built = rr;
                rc = sm_build_plan(L, S, QS, cache.scratch, Bx, rr, masked, attempt == 2 || rr == 8, lds_limit, s, &fresh, herr);
                const int next = std::max(8, rr / 2);
                if (rc != 1 || attempt == 2 || next == rr) break;        // built, failed, or nothing smaller left to try (rc stays 1: remembered as not fitting)
                rr = next;
                rc = 0;
            }
        } catch (const std::bad_alloc&) {
            rc = -1;
            *herr = hipErrorOutOfMemory;
        }
        if (rc == 1) fresh = new (std::nothrow) SmPlan();
        lk.lock();
        ++cache.builds;
        cache.build_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_build).count();
        if (rc < 0) return nullptr;
        if (!fresh) { *herr = hipErrorOutOfMemory; return nullptr; }
        fresh->W = L.W; fresh->H = L.H; fresh->N = N; fresh->n_rings = S.n_rings; fresh->w = V.out_w; fresh->h = V.out_h; fresh->Bx = Bx; fresh->R = R;
        fresh->rows = built; fresh->masked = masked;
        fresh->sxu = fbits(V.sxu); fresh->syv = fbits(V.syv);
        std::memset(fresh->ring_key, 0, sizeof(fresh->ring_key));
        std::memcpy(fresh->ring_key, key, sizeof(key[0]) * S.n_rings);
        plan = find();
        if (plan) {
            cache.graveyard.push_back(fresh);            // another thread built the same geometry meanwhile: theirs is in use, ours goes
        } else {
            plan = fresh;
            if (cache.plans.size() >= cache.cap) {       // evict the least recently used plan no call holds (all held: the cache grows for now)
                size_t lru = cache.plans.size();
                for (size_t i = 0; i < cache.plans.size(); ++i)
                    if (cache.plans[i]->pins == 0 && (lru == cache.plans.size() || cache.plans[i]->stamp < cache.plans[lru]->stamp)) lru = i;
                if (lru < cache.plans.size()) {
                    cache.graveyard.push_back(cache.plans[lru]);
                    cache.plans.erase(cache.plans.begin() + (long)lru);
                }
            }
            cache.plans.push_back(plan);
        }
    }
    uint64_t newest = 0;
    for (SmPlan* p : cache.plans) newest = std::max(newest, p->stamp);
    plan->stamp = newest + 1;
    return plan->n_tiles ? plan : nullptr;
}

// Images per workgroup G (a divisor of the tile's 2 N images).  A workgroup takes about G + 1 image times (its first copy is not hidden)
// and the job runs in ceil(workgroups / resident workgroups) rounds, so G minimises rounds x (G + 1); ties go to the larger G (the plan
// entries are copied once per workgroup).  Measured (cfg2, one frame per call, 32-row tiles): G = 6 -> 19.3 us, 3 -> 21.6, 2 -> 23.4,
// 4 -> 24.9, 12 -> 31.1 (189 workgroups for 256 CUs); sixteen frames: G = 12 (profiles/r05/srcmajor_images_sweep.txt).
struct SmPick { int G; long long wgs, rounds, resident; };
SmPick sm_pick_images(const SmPlan& plan, int N, int n_frames, int n_cu) {
    const size_t lds_wg = (size_t)plan.ent_bytes + 2 * (size_t)plan.buf_bytes + 2 * (size_t)plan.mbuf_bytes + 2048;
    SmPick k;
    k.resident = (long long)n_cu * (long long)std::max<size_t>(1, (160 * 1024) / lds_wg);
    const long long items = (long long)plan.n_tiles * 2 * N * n_frames;
    long long best = -1;
    k.G = 1; k.wgs = items; k.rounds = 1;
    for (int g = 1; g <= kSmMaxImages; ++g) {
        if ((2 * N) % g) continue;
        const long long wgs = items / g, rounds = (wgs + k.resident - 1) / k.resident, c = rounds * (g + 1);
        if (best < 0 || c <= best) { best = c; k.G = g; k.wgs = wgs; k.rounds = rounds; }
    }
    return k;
}

}  // namespace

void sm_cache_drain(SmCache& cache) {
    std::vector<SmPlan*> dead;
    {
        std::lock_guard<std::mutex> lock(cache.mu);
        dead.swap(cache.graveyard);
    }
    for (SmPlan* p : dead) sm_plan_free(p);
}

void sm_cache_destroy(SmCache& cache) {
    sm_cache_drain(cache);
    if (cache.scratch.dev) (void)hipFree(cache.scratch.dev);
    std::free(cache.scratch.host);
    cache.scratch = SmScratch();
    std::lock_guard<std::mutex> lock(cache.mu);
    for (SmPlan* p : cache.plans) sm_plan_free(p);
    cache.plans.clear();
}

// The plan a CALL renders with -- decided ONCE, before its first chunk of frames, and held (never evicted) until sm_release: a call of more
// than GS360_MAX_FRAMES frames must not change plan -- or lose it -- between chunks after earlier chunks were launched.  `L` = the first
// chunk.  Returns 0 with *out set, 1 (the geometry does not fit, or its boxes outgrow their grid cells: the caller takes the gather
// kernels) or -1 with *herr set.
int sm_prepare(const EqLaunch& L, const SmShape& S, SmCache& cache, bool masked, int Bx, int R, int G_opt, bool adapt, int max_box_pct, size_t lds_limit, int n_cu,
               hipStream_t s, hipError_t* herr, SmPlan** out, int* box_pct) {
    *herr = hipSuccess;
    *out = nullptr;
    std::unique_lock<std::mutex> lk(cache.mu);
    if (cache.graveyard.size() > 64) {                   // nobody called gs360_sync for a long time: release here (hipFree waits for the device)
        std::vector<SmPlan*> dead;
        dead.swap(cache.graveyard);
        cache.inline_frees += dead.size();
        lk.unlock();
        for (SmPlan* p : dead) sm_plan_free(p);
        lk.lock();
    }
    SmPlan* plan = sm_get_plan(L, S, cache, lk, Bx, R, masked, lds_limit, s, herr);
    if (!plan) return *herr == hipSuccess ? 1 : -1;
    *box_pct = plan->box_pct;
    if (max_box_pct > 0 && plan->box_pct > max_box_pct) return 1;     // (automatic selection only) boxes far larger than their grid cells: views stretched towards a pole
    // A job that cannot fill the GPU once even with the longest workgroups (cfg2: 188 tiles x 12 images / 12 = 188 workgroups per frame for
    // 512 places) runs faster on tiles of half the height -- twice the workgroups, each half as long, and the smaller LDS footprint lets
    // three of them share a CU: cfg2 19.2 -> 16.0 us for one frame per call, 16.1 -> 13.3 for two; from three frames on the tall tiles
    // win again (16.1 against 16.5, four: 15.0 against 17.2: the boxes' halo bytes), and 16-row tiles are never halved (8K -> 6 x 1200^2,
    // 8 x 1600^2, 12 x 800^2 at one frame: +8 .. +22 %).  profiles/r05/srcmajor_small_jobs.txt
    if (adapt && G_opt == 0 && plan->rows >= 32) {
        const SmPick pick = sm_pick_images(*plan, S.N, L.n_frames, n_cu);
        int gmax = 1;
        for (int g = 1; g <= kSmMaxImages; ++g) if ((2 * S.N) % g == 0) gmax = g;
        const long long coarsest = (long long)plan->n_tiles * 2 * S.N * L.n_frames / gmax;
        if (coarsest * 10 < pick.resident * 9) {
            ++plan->pins;                                // (the lock is released while the half-height plan is built)
            SmPlan* half = sm_get_plan(L, S, cache, lk, Bx, plan->rows / 2, masked, lds_limit, s, herr);
            --plan->pins;
            if (!half && *herr != hipSuccess) return -1;
            // (halo rows weigh more in the half-height boxes: one that outgrows the limit leaves the call on the full-height plan)
            if (half && !(max_box_pct > 0 && half->box_pct > max_box_pct)) plan = half;
        }
    }
    ++plan->pins;
    *out = plan;
    return 0;
}

void sm_release(SmCache& cache, SmPlan* plan) {
    if (!plan) return;
    std::lock_guard<std::mutex> lock(cache.mu);
    --plan->pins;
}

// Renders one chunk of frames through the source-major kernel with the call's plan.  Returns 0 or -1 with *herr set.
int sm_launch(const EqLaunch& L, const SmShape& S, const SmPlan* plan, int G_opt, bool stage_regs, size_t lds_limit, int n_cu, hipStream_t s, hipError_t* herr, int* info) {
    int* const box_pct = info;                           // info[0..3]: the plan's box overhead in percent, its tile rows, images per workgroup, register staging
    const EqView& V = L.view[0];
    const int N = S.N, NV = L.n_views;
    *herr = hipSuccess;
    const bool masked = L.mask[0] != nullptr;
    const SmPick pick = sm_pick_images(*plan, N, L.n_frames, n_cu);
    box_pct[0] = plan->box_pct; box_pct[1] = plan->rows; box_pct[2] = pick.G;
    SmArgs P;
    std::memset(&P, 0, sizeof(P));
    for (int f = 0; f < L.n_frames; ++f) P.src[f] = L.src[f];
    for (int i = 0; i < L.n_frames * NV; ++i) P.dst[i] = L.dst[i];
    P.tiles = plan->d_tiles; P.entries = plan->d_entries;
    P.W = L.W; P.H = L.H; P.N = N; P.NV = NV; P.w = V.out_w; P.h = V.out_h; P.PB = plan->PB;
    int G = pick.G;
    if (G_opt > 0 && G_opt <= kSmMaxImages && (2 * N) % G_opt == 0) G = G_opt;          // option "srcmajor_images" (probes)
    box_pct[2] = G;
    P.G = G; P.groups_per_tile = 2 * N / G; P.groups_per_frame = plan->n_tiles * P.groups_per_tile;
    P.total_groups = P.groups_per_frame * L.n_frames; P.gchunk = (P.total_groups + 7) / 8;
    P.buf_bytes = plan->buf_bytes; P.ent_bytes = plan->ent_bytes;
    for (int i = 0; i < NV; ++i) P.qmap[i] = S.qmap[i];
    for (int c = 0; c < S.n_rings; ++c) P.partner[c] = S.partner[c];
    P.src_stride = L.src_stride;
    P.dst_stride = L.dst_stride ? L.dst_stride : (int64_t)V.out_w * 3;
    const size_t lds = (size_t)plan->ent_bytes + 2 * (size_t)plan->buf_bytes + 2 * (size_t)plan->mbuf_bytes;
    if (masked) {
        for (int f = 0; f < L.n_frames; ++f) P.mask[f] = L.mask[f];
        P.mask_stride = (int32_t)L.mask_stride; P.mask_dw = L.W / 32; P.mbuf_bytes = plan->mbuf_bytes;
    }
    const bool rs = stage_regs && !masked && plan->max_wch <= 64 && plan->min_nrows >= kSmConsumers && plan->max_nrows <= kSmRsRows * kSmConsumers &&
                    L.src_stride % 16 == 0;
    box_pct[3] = rs ? 1 : 0;
    const void* const kernel = masked ? (const void*)eq_srcmajor_kernel<kSmConsumers, true, false>
                                      : (rs ? (const void*)eq_srcmajor_kernel<kSmConsumers, false, true> : (const void*)eq_srcmajor_kernel<kSmConsumers, false, false>);
    *herr = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);   // (per device: cheap, host side)
    if (*herr != hipSuccess) return -1;
    void* args[] = {(void*)&P};
    *herr = hipLaunchKernel(kernel, dim3((unsigned)(P.gchunk * 8)), dim3((unsigned)(64 * (kSmConsumers + (rs ? 0 : 1)))), args, lds, s);
    if (*herr == hipSuccess) *herr = hipGetLastError();
    return *herr == hipSuccess ? 0 : -1;
}

}  // namespace gs360
