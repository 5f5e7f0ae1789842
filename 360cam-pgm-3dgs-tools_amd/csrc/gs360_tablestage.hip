// gs360_tablestage.hip -- LDS-staged, plan-driven cv2.remap for the dual-fisheye tool's hot call: INTER_LINEAR, BORDER_CONSTANT, 8-bit RGB,
// `out[~valid] = mask_value` (cli_tools/gs360_DualFisheyeDistortionCalibration.py:2001-2014; the maps are those of DF:1759-1823, applied to
// every lens pair of a run, DF:2582-2592).  north_star's shape for this half of the path: one output tile per workgroup turn, source
// texels staged in LDS.
//
// table_remap_kernel (gs360_table.hip) gathers per pixel: through a map plan it reads 5 plan bytes and two 12-byte tap windows per pixel
// from memory -- at cfg4's 1.26 source texels per output pixel that is five vector-memory instructions per 64 pixels on the texture path
// (72 % busy) next to ~85 vector-ALU instructions of coordinate unpacking, border tests and address arithmetic.  Here a map plan grows a
// STAGE PLAN per source size (built on the GPU at the first call, table_stage_plan_kernel): the output is cut into tiles of 64 pixels x R
// rows; per tile a RECORD -- a head (the box of source texels its pixels' taps touch: first byte, first row, rows, 16-byte chunks per
// row) and per pixel ONE dword, LDS byte offset of the top-left tap | fx << 17 | fy << 22, tile-major.  A workgroup of eight wavefronts
// walks a run of tiles; everything a tile needs is requested one tile AHEAD by ordinary loads into registers -- its head, each wavefront's
// plan words (four row slots) and each wavefront's share of the box (16 bytes per lane) -- and the box is written to the other LDS buffer
// after the current tile's rendering: two tap windows per pixel from LDS, the gather kernels' exact-integer blend (gs360_blend.h), a
// quad's four 24-bit pixels re-sliced into three dwords by two DPP moves, one dword store per lane.  No coordinate arithmetic, no border
// test, no map read in the loop.
//
// Measured (MI355X, cfg4 = 2 x 4000^2 -> 6 x 1750^2 through plans, four pairs in turn, HBM-cold; profiles/r06/table_stage/): 64-65 us per
// pair against 69-71 for the gather kernel in the same runs (177 + 59 MB moved instead of 196 + 60, 19.8 M instead of 24.2 M vector
// instructions).  What was tried on the way and why it lost: LDS copies (global_load_lds) by a loader wavefront -- a CU takes one 1 KiB copy
// instruction per ~100 cycles whatever issues it (25 GB/s per CU: the boxes and words alone 29 us), one wavefront one per ~390 cycles, and
// beside consumers that read LDS the loaders' issue slows to ~1000 cycles per instruction (91 us per pair with one loader, 78 with three);
// every wavefront copying its share before rendering (the copy issue and the rendering then alternate: 100 us); thinner tiles (a row segment
// of a yawed or pitched view runs diagonally through the fisheye image: the bounding boxes of 64 x 8 tiles hold 2.6x their texels, those
// of 64 x 32 tiles 1.6x).
//
// Pixels the loop cannot serve carry bit 31 and a kind: the valid fill (DF:2009-2014), the border constant (all four taps outside), and
// SLOW ones -- a tap pair that straddles the image border, or any pixel of a tile whose box exceeds the LDS budget (arbitrary maps) --
// which are redone from memory by the straight-line sampler of the gather kernel (cv_sample_linear, same results by construction).
//
// Output addressing is the gather kernel's FLAT form: row y of a tight output is the span [r(y), r(y + 1)) of the flat pixel stream,
// r(y) = y w rounded up to a multiple of four, so that every 64-pixel segment -- and every quad -- starts on a 12-byte boundary whatever
// the width (the tool's default 1750-pixel views have 5250-byte rows).  Segments shorter than 64 pixels are padded in the PLAN with copies
// of their last quad: the padded lanes compute and store the same dwords to the same addresses, the loop stays unconditional.
#include <cstring>
#include <new>

#include "gs360_cvremap.h"

namespace gs360 {

namespace {

constexpr int kTsWaves = 8;                             // wavefronts per workgroup
constexpr int kTsRecHead = 64;                          // bytes of a tile record in front of its plan words (TsTile + padding)
constexpr int kTsRecBytes = kTsRecHead + 8 * 64 * 4 * 4;   // ... and of the record: 8 wavefronts x 64 lanes x 4 row slots x one dword
constexpr uint32_t kTsSpecial = 0x80000000u;            // plan word: bit 31 = not served by the loop; bits 29-30 = kind
constexpr uint32_t kTsFill = 0u << 29, kTsBorder = 1u << 29, kTsSlow = 2u << 29;

// first flat pixel of row y's span (tight outputs): y w rounded up to a multiple of four; r(h) = h w
__host__ __device__ __forceinline__ int ts_span_start(const int y, const int h, const int w) { return y < h ? (y * w + 3) & ~3 : h * w; }

// ---- plan builder: one workgroup per tile ---------------------------------------------------------------------------------------------
// Pass 1 classifies the tile's pixels and reduces the box of the FAST ones (all taps that carry weight inside the image); pass 2 writes the
// words.  A right / bottom tap that falls outside with weight zero (ix == W - 1 with fx == 0; cv2 multiplies the border constant by 0
// there) is served from the box: the loader clamps box rows to H - 1 and the bytes right of a row are readable (next row, or the slack
// include/gs360.h asks for).
struct TsPix { int ix, iy, fx, fy, kind; };             // kind: -1 fast, else kTsFill / kTsBorder / kTsSlow >> 29
__device__ __forceinline__ TsPix ts_classify(const uint32_t* __restrict__ packed, const uint8_t* __restrict__ hi, const int p, const int W, const int H,
                                             const int use_valid) {
    const uint32_t P = packed[p], hb = hi[p];
    TsPix q;
    q.ix = (int)(P & 0xfffu) - 8;
    q.iy = (int)((P >> 12) & 0xfffu) - 8;
    q.fx = (int)((P >> 24) & 31u);
    q.fy = (int)((P >> 29) | ((hb & 3u) << 3));
    if (use_valid && !(hb & 4u)) q.kind = (int)(kTsFill >> 29);
    else if (q.ix >= W || q.ix + 1 < 0 || q.iy >= H || q.iy + 1 < 0) q.kind = (int)(kTsBorder >> 29);      // remapBilinear: nothing sampled
    else if (q.ix >= 0 && q.iy >= 0 && (q.ix <= W - 2 || (q.ix == W - 1 && q.fx == 0)) && (q.iy <= H - 2 || (q.iy == H - 1 && q.fy == 0))) q.kind = -1;
    else q.kind = (int)(kTsSlow >> 29);
    return q;
}

__global__ __launch_bounds__(256) void table_stage_plan_kernel(const uint32_t* __restrict__ packed, const uint8_t* __restrict__ hi, const int h, const int w,
                                                               const int W, const int H, const int R, const int tiles_x, const int use_valid,
                                                               const int box_budget, uint8_t* __restrict__ recs,
                                                               int* __restrict__ stats /* [0] largest box, [1] tiles without a box that wanted one */) {
    __shared__ int s_box[4];                             // min x, max x, min y, max y over the fast pixels
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_box[0] = 1 << 30; s_box[1] = -1; s_box[2] = 1 << 30; s_box[3] = -1; }
    __syncthreads();
    auto pixel_of = [&](const int r, int& p) -> bool {   // flat pixel of (row slot r, this lane); false: the slot stores nothing
        const int y = ty * R + r;
        if (y >= h) return false;
        const int first = ts_span_start(y, h, w) + 64 * tx;
        const int n = min(64, ts_span_start(y + 1, h, w) - first);
        if (n <= 0) return false;
        const int l = lane < n ? lane : ((n - 1) & ~3) + (lane & 3);      // padding: copies of the segment's last quad
        p = first + min(l, n - 1);                       // (n is a multiple of four whenever h w is)
        return true;
    };
    for (int r = wv; r < R; r += 4) {
        int p;
        if (!pixel_of(r, p)) continue;
        const TsPix q = ts_classify(packed, hi, p, W, H, use_valid);
        if (q.kind < 0) {
            atomicMin(&s_box[0], q.ix); atomicMax(&s_box[1], q.ix);
            atomicMin(&s_box[2], q.iy); atomicMax(&s_box[3], q.iy);
        }
    }
    __syncthreads();
    TsTile T;
    T.x0b = T.y0 = T.nrows = T.wch = T.magic = T.chunks = 0;
    T.ty = ty; T.tx = tx;
    bool boxed = s_box[1] >= 0;
    if (boxed) {
        T.x0b = (3 * s_box[0]) & ~15;
        T.wch = ((((3 * s_box[1] - T.x0b) & ~3) + 12) + 15) >> 4;          // the consumers read three dwords from the dword below a tap
        T.y0 = s_box[2];
        T.nrows = s_box[3] - s_box[2] + 2;
        if (T.nrows * T.wch * 16 > box_budget) {         // a map that scatters this tile's taps: its pixels go the slow way
            boxed = false;
            T.x0b = T.y0 = T.nrows = T.wch = 0;
            if (threadIdx.x == 0) atomicAdd(&stats[1], 1);
        } else {
            T.chunks = T.nrows * T.wch;                  // < 2^13 (the budget is < 128 KiB)
            T.magic = ((1 << 20) + T.wch - 1) / T.wch;   // chunk c lies in box row (c * magic) >> 20: exact for c < 2^20 / wch
            if (threadIdx.x == 0) atomicMax(&stats[0], T.chunks * 16);
        }
    }
    uint8_t* const rec = recs + (size_t)tile * (size_t)kTsRecBytes;
    if (threadIdx.x == 0) *reinterpret_cast<TsTile*>(rec) = T;
    const int pitch = T.wch * 16;
    uint32_t* const wt = reinterpret_cast<uint32_t*>(rec + kTsRecHead);
    for (int r = wv; r < R; r += 4) {
        int p;
        uint32_t word = kTsSpecial | kTsFill;            // (slots that store nothing: rendered as a fill, never stored)
        if (pixel_of(r, p)) {
            const TsPix q = ts_classify(packed, hi, p, W, H, use_valid);
            if (q.kind >= 0) word = kTsSpecial | ((uint32_t)q.kind << 29);
            else if (!boxed) word = kTsSpecial | kTsSlow;
            else word = (uint32_t)((q.iy - T.y0) * pitch + 3 * q.ix - T.x0b) | ((uint32_t)q.fx << 17) | ((uint32_t)q.fy << 22);
        }
        // a wavefront's four row slots (rows r, r + 8, r + 16, r + 24) side by side per lane: ONE 16-byte load per lane and tile
        wt[(((r % kTsWaves) * 64 + lane) << 2) + r / kTsWaves] = word;
    }
    for (int i = threadIdx.x; i < kTsWaves * 64 * 4; i += 256)       // slots past the tile's rows (R < 32): nothing to render
        if ((i & 3) * kTsWaves + ((i >> 2) >> 6) >= R) wt[i] = kTsSpecial | kTsFill;
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------------------
struct TsJob {
    const uint8_t* src;
    uint8_t* dst;
    const uint32_t* packed;              // the map plan's own 5 bytes per pixel: read by SLOW pixels only
    const uint8_t* packed_hi;
    const uint8_t* recs;                 // tile records: 64-byte head (TsTile) + R * 64 plan words each
    int32_t W, H, h, w;
    int32_t src_stride, dst_stride;      // bytes (< 2^31, checked by the host)
    int32_t tight;                       // the output is the flat pixel stream (dst_stride == 3 w); else w % 4 == 0 and rows start on dwords
    int32_t tile_base;
    uint32_t fillpk;                     // the fill value on all three channels
    int32_t pad_;
};
struct TsArgs {
    TsJob job[GS360_MAX_VIEWS];
    int32_t n_jobs, total_tiles, chunk, R;
    int32_t buf_bytes;                   // one LDS box buffer: the largest box of the launch + slack
    uint32_t cvalpk;                     // the border constant, r | g << 8 | b << 16
    uint8_t cval[4];
};
static_assert(sizeof(TsArgs) <= 4096, "TsArgs travels as a kernel argument");

// Box chunks (16 bytes each) one wavefront stages per tile: ceil(budget / 16 / 64 / kTsWaves)
constexpr int kTsStageRegs = 4;

template <int NW>
__global__ __launch_bounds__(64 * NW) void table_staged_kernel(const TsArgs P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    // XCD x (= block % 8) owns a contiguous chunk of the tile order (view, tile row, tile column); its workgroups walk the chunk
    // interleaved, so that at any time an XCD's L2 serves a window of neighbouring tiles -- whose boxes share halo rows and lines
    const int b = blockIdx.x, xcd = b & 7, nj = (int)(gridDim.x >> 3);
    const int t_end = min((xcd + 1) * P.chunk, P.total_tiles);
    const int t0 = xcd * P.chunk + (b >> 3);
    if (t0 >= t_end) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int R = P.R;
    constexpr int rec_bytes = kTsRecBytes;
    // The job of a tile: the tile order runs job by job and a workgroup's tiles only move forward, so each of the three places that ask
    // (this tile, the next one, the one after it) keeps its own cursor.
    struct Where { int j, lt; };
    auto locate = [&](const int tg, int& cj) {
        while (cj + 1 < P.n_jobs && tg >= P.job[cj + 1].tile_base) ++cj;
        Where q;
        q.j = cj; q.lt = tg - P.job[cj].tile_base;
        return q;
    };
    int cj0 = 0, cj1 = 0, cj2 = 0;
    // Everything a tile needs from memory is requested ONE TILE AHEAD into registers by ordinary loads, by every wavefront for itself:
    //   * the head (eight dwords, one per lane of the first eight; unpacked with v_readlane) -- two tiles ahead, since the box loads need it;
    //   * the plan words of the wavefront's four row slots (a coalesced dword per lane and slot);
    //   * the wavefront's share of the box: chunk c = 64 (wave + NW k) + lane, 16 bytes per lane, written to LDS after this tile's rendering.
    // LDS copies (global_load_lds) were measured first and are NOT used: a CU takes one 1 KiB copy instruction per ~100 cycles whatever
    // issues it (25 GB/s: cfg4's boxes and words alone 29 us), one wavefront one per ~390, and while the consumers read LDS the loaders'
    // issue slows to ~1000 cycles per instruction (profiles/r06/table_stage/).  Ordinary loads go through the L1 at 64 bytes per cycle.
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const int k4off = 4 * min(k4, 2);
    const uint32_t lane_off = (uint32_t)(12 * (lane >> 2) + k4off);       // a lane's dword inside a 64-pixel segment
    auto head_load = [&](const int tg) -> uint32_t {
        const Where q = locate(tg, cj2);
        return reinterpret_cast<const uint32_t*>(P.job[q.j].recs + (size_t)q.lt * (size_t)rec_bytes)[lane & 7];
    };
    struct Words { uint32_t a, b, c, d; };
    auto words_load = [&](const int tg) {
        const Where q = locate(tg, cj1);
        const uint4 v = reinterpret_cast<const uint4*>(P.job[q.j].recs + (size_t)q.lt * (size_t)rec_bytes + kTsRecHead)[wave * 64 + lane];
        Words o;
        o.a = v.x; o.b = v.y; o.c = v.z; o.d = v.w;      // (slots past the tile's rows hold the fill word)
        return o;
    };
    struct Head { int x0b, y0, wch, magic, chunks, ty, tx; };
    auto unpack = [&](const uint32_t hv) {
        Head T;
        T.x0b = __builtin_amdgcn_readlane((int)hv, 0); T.y0 = __builtin_amdgcn_readlane((int)hv, 1);
        T.wch = __builtin_amdgcn_readlane((int)hv, 3); T.magic = __builtin_amdgcn_readlane((int)hv, 4);
        T.chunks = __builtin_amdgcn_readlane((int)hv, 5); T.ty = __builtin_amdgcn_readlane((int)hv, 6); T.tx = __builtin_amdgcn_readlane((int)hv, 7);
        return T;
    };
    // the box's rows lie back to back in LDS (pitch = wch * 16): chunk c goes to byte 16 c whatever box row it belongs to
    static_assert(kTsStageRegs == 4, "four staged chunks per lane");
    struct Stage { uint4 a, b, c, d; };
    auto box_load = [&](const int j, const Head& T) {
        const TsJob& J = P.job[j];
        const uint8_t* const src = J.src;
        const int H1 = J.H - 1, stride = J.src_stride;
        auto chunk = [&](const int k) {
            const int cc = max(min(64 * (wave + NW * k) + lane, T.chunks - 1), 0);      // (lanes past the box repeat its last chunk)
            const int row = (int)(((uint32_t)cc * (uint32_t)T.magic) >> 20), col = cc - row * T.wch;
            const int yc = min(T.y0 + row, H1);          // (a bottom tap of weight zero: any readable row)
            return *reinterpret_cast<const uint4*>(__builtin_assume_aligned(src + (size_t)yc * (size_t)stride + (uint32_t)(T.x0b + col * 16), 4));
        };
        Stage o;
        o.a = o.b = o.c = o.d = make_uint4(0u, 0u, 0u, 0u);
        if (64 * wave < T.chunks) o.a = chunk(0);        // (wave-uniform: a box of 12 KiB is one or two loads per wavefront)
        if (64 * (wave + NW) < T.chunks) o.b = chunk(1);
        if (64 * (wave + 2 * NW) < T.chunks) o.c = chunk(2);
        if (64 * (wave + 3 * NW) < T.chunks) o.d = chunk(3);
        return o;
    };
    auto box_store = [&](const Head& T, const Stage& o, uint8_t* const box) {
        const int c = 64 * wave + lane;
        if (c < T.chunks) *reinterpret_cast<uint4*>(box + c * 16) = o.a;
        if (c + 64 * NW < T.chunks) *reinterpret_cast<uint4*>(box + (c + 64 * NW) * 16) = o.b;
        if (c + 128 * NW < T.chunks) *reinterpret_cast<uint4*>(box + (c + 128 * NW) * 16) = o.c;
        if (c + 192 * NW < T.chunks) *reinterpret_cast<uint4*>(box + (c + 192 * NW) * 16) = o.d;
    };
    // prologue: heads of the first two tiles, words and box of the first
    uint32_t hv0 = head_load(t0), hv1 = t0 + nj < t_end ? head_load(t0 + nj) : 0u;
    Words pwn = words_load(t0);
    Head T = unpack(hv0);
    box_store(T, box_load(locate(t0, cj0).j, T), s_lds);
    __syncthreads();                                     // (prologue)
    // the current job's constants in scalar registers, re-read only when a tile belongs to the next job
    int jj = -1, h = 0, w = 0, tight = 0, dstride = 0;
    uint32_t fillpk = 0;
    uint8_t* dstp = nullptr;
    const uint32_t cvalpk = P.cvalpk;
    int g = 0;
    for (int t = t0; t < t_end; t += nj, ++g) {
        uint32_t pw[4] = {pwn.a, pwn.b, pwn.c, pwn.d};
        const int j_cur = locate(t, cj0).j;
        if (j_cur != jj) {
            jj = j_cur;
            const TsJob& J = P.job[jj];
            h = J.h; w = J.w; tight = J.tight; dstride = J.dst_stride; fillpk = J.fillpk; dstp = J.dst;
        }
        const TsJob& J = P.job[jj];
        // requests for the next tile (and the head of the one after it)
        const bool more = t + nj < t_end;
        const Head Tn = unpack(hv1);
        Stage st;
        st.a = st.b = st.c = st.d = make_uint4(0u, 0u, 0u, 0u);
        uint32_t hv2 = 0u;
        if (more) {
            pwn = words_load(t + nj);
            st = box_load(cj1, Tn);
            if (t + 2 * nj < t_end) hv2 = head_load(t + 2 * nj);
        }
        const uint8_t* const box = s_lds + (g & 1) * P.buf_bytes;
        const int pitch = 16 * T.wch, ty = T.ty, tx = T.tx;
        // A row slot's place in the output (wave-uniform): row y of a tight output is the span [r(y), r(y + 1)) of the flat pixel stream,
        // r(y) = y w rounded up to a multiple of four (h w is one, so r(h) = h w; with rows of whole dwords r(y) = y w).
        int first[4], nq1[4];
        uint32_t seg[4], pk[4];
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = wave + k * NW, y = ty * R + r;
            first[k] = ((y * w + 3) & ~3) + 64 * tx;
            const int n = min(64, (((y + 1) * w + 3) & ~3) - first[k]);
            seg[k] = tight ? 3u * (uint32_t)first[k] : (uint32_t)y * (uint32_t)dstride + 192u * (uint32_t)tx;
            live[k] = r < R && y < h && n > 0;
            nq1[k] = (n >> 2) - 1;
        }
        // all eight tap windows of the four slots in flight, then the blends, then the stores: the wavefronts of a workgroup run in lockstep
        // behind the tile barrier -- what one waits for (an LDS round trip) all wait for at the same time
        uint32_t ta[4][3], tb[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t toff = pw[k] & 0x1fffcu;
            const uint32_t* qa = reinterpret_cast<const uint32_t*>(box + toff);
            const uint32_t* qb = reinterpret_cast<const uint32_t*>(box + toff + pitch);
            ta[k][0] = qa[0]; ta[k][1] = qa[1]; ta[k][2] = qa[2];
            tb[k][0] = qb[0]; tb[k][1] = qb[1]; tb[k][2] = qb[2];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint2 t0, t1;                                // rows iy, iy + 1: bytes r0 g0 b0 r1 | g1 b1 . .
            t0.x = __builtin_amdgcn_alignbyte(ta[k][1], ta[k][0], pw[k]); t0.y = __builtin_amdgcn_alignbyte(ta[k][2], ta[k][1], pw[k]);
            t1.x = __builtin_amdgcn_alignbyte(tb[k][1], tb[k][0], pw[k]); t1.y = __builtin_amdgcn_alignbyte(tb[k][2], tb[k][1], pw[k]);
            uint32_t px[3];
            blend_rgb_rows(t0, t1, (int)((pw[k] >> 17) & 31u), (int)((pw[k] >> 22) & 31u), px);
            asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(pk[k]) : "v"(px[1]), "v"(px[0]));
            asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(pk[k]) : "v"(px[2]), "v"(pk[k]));
        }
        if (any_lane((int)(pw[0] | pw[1] | pw[2] | pw[3]) < 0)) {
            // pixels the loop does not serve (fill, border constant, SLOW: redone from memory through the plan's own position).
            // ONE copy of the fix-up in a rolled loop that always works on slot 0 and rotates the four slots after every turn
            // (plain register moves, back in place after four turns): indexing the slot arrays would put them in scratch memory.
            uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2], w3 = pw[3], p0 = pk[0], p1 = pk[1], p2 = pk[2], p3 = pk[3];
            int f0 = first[0], f1 = first[1], f2 = first[2], f3 = first[3], n0 = nq1[0], n1 = nq1[1], n2 = nq1[2], n3 = nq1[3];
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                if (any_lane((int)w0 < 0)) {
                    const uint32_t kind = w0 & (3u << 29);
                    uint32_t spk = kind == kTsFill ? fillpk : cvalpk;
                    if ((int)w0 < 0 && kind == kTsSlow) {
                        const int p = min(max(f0 + 4 * min(lane >> 2, max(n0, 0)) + k4, 0), h * w - 1);
                        float mx, my;
                        planned_coords(J.packed[p], J.packed_hi[p], false, mx, my);
                        uint32_t o[4];
                        cv_sample_linear<3>(J.src, (int64_t)J.src_stride, J.W, J.H, mx, my, P.cval, o);
                        spk = o[0] | (o[1] << 8) | (o[2] << 16);
                    }
                    p0 = (int)w0 < 0 ? spk : p0;
                }
                const uint32_t tw = w0, tp = p0;
                const int tf = f0, tn = n0;
                w0 = w1; w1 = w2; w2 = w3; w3 = tw;
                p0 = p1; p1 = p2; p2 = p3; p3 = tp;
                f0 = f1; f1 = f2; f2 = f3; f3 = tf;
                n0 = n1; n1 = n2; n2 = n3; n3 = tn;
            }
            pk[0] = p0; pk[1] = p1; pk[2] = p2; pk[3] = p3;
        }
        // The next tile's box into the other buffer (last read before the previous barrier) -- BEFORE this tile's stores are issued: the
        // wait for the box loads then has nothing younger in the wavefront's memory queue (a wait behind the four stores became
        // s_waitcnt vmcnt(0): 1-2 us per tile until the stores had left).  The next tile's words and the head after it are pinned here
        // as well, so that the top of the next iteration does not wait for this one's stores either.
        if (more) box_store(Tn, st, s_lds + ((g + 1) & 1) * P.buf_bytes);
        asm volatile("" : "+v"(pwn.a), "+v"(pwn.b), "+v"(pwn.c), "+v"(pwn.d), "+v"(hv2));
        __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks the LDS writes below the stores again)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // lanes 4m .. 4m + 3 hold a quad: lane k cuts dword k of its 12 bytes out of pixels k and k + 1; lane 3 repeats lane 2's store
            const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk[k], 0xF9, 0xf, 0xf, true);       // quad_perm [1,2,3,3]
            const uint32_t dw = __builtin_amdgcn_perm(nxt, pk[k], sel);
            const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, true);         // quad_perm [0,1,2,2]
            // (a segment shorter than 64 pixels -- the last tile column: its padded lanes repeat the last quad, values and addresses)
            const uint32_t loff = nq1[k] < 15 ? (uint32_t)(12 * min(lane >> 2, nq1[k]) + k4off) : lane_off;
            if (live[k]) *reinterpret_cast<uint32_t*>(__builtin_assume_aligned(dstp + (seg[k] + loff), 4)) = dwq;
        }
        T = Tn;
        hv1 = hv2;
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wavefront's box rows are in LDS (NOT vmcnt: the stores drain on their own)
        __builtin_amdgcn_s_barrier();                    // tile g + 1's box is in LDS AND every wavefront is done with tile g's
    }
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------------------------------------------------------
void ts_plan_free(TsPlan* p) {
    if (!p) return;
    if (p->d_recs) (void)hipFree(p->d_recs);
    delete p;
}

// The stage plan of a map plan for sources of W x H (the slow / border classes depend on the source size).  Synchronises `s` once (the
// largest box decides the launch's LDS size).  nullptr with *herr == hipSuccess: out of host memory.
TsPlan* ts_build_plan(const uint32_t* d_packed, const uint8_t* d_hi, int h, int w, int W, int H, int R, int use_valid, int box_budget, hipStream_t s,
                      hipError_t* herr) {
    *herr = hipSuccess;
    TsPlan* p = new (std::nothrow) TsPlan();
    if (!p) return nullptr;
    p->W = W; p->H = H; p->R = R; p->h = h; p->w = w; p->use_valid = use_valid;
    p->tiles_x = (w + 3 + 63) / 64;                      // (a span may be three pixels longer than a row)
    p->n_tiles = p->tiles_x * ((h + R - 1) / R);
    int* d_stats = nullptr;
    *herr = hipMalloc((void**)&p->d_recs, (size_t)p->n_tiles * (size_t)kTsRecBytes + 1024);      // (slack: a record is copied in whole 16-byte chunks)
    if (*herr == hipSuccess) *herr = hipMalloc((void**)&d_stats, 2 * sizeof(int));
    if (*herr == hipSuccess) *herr = hipMemsetAsync(d_stats, 0, 2 * sizeof(int), s);
    if (*herr == hipSuccess) {
        hipLaunchKernelGGL(table_stage_plan_kernel, dim3((unsigned)p->n_tiles), dim3(256), 0, s, d_packed, d_hi, h, w, W, H, R, p->tiles_x, use_valid,
                           box_budget, p->d_recs, d_stats);
        *herr = hipGetLastError();
    }
    int stats[2] = {0, 0};
    if (*herr == hipSuccess) *herr = hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, s);
    if (*herr == hipSuccess) *herr = hipStreamSynchronize(s);
    if (d_stats) (void)hipFree(d_stats);
    if (*herr != hipSuccess) { ts_plan_free(p); return nullptr; }
    p->max_box = stats[0];
    p->slow_tiles = stats[1];
    return p;
}

hipError_t ts_launch(const TsLaunch& L, int n_cu, size_t lds_per_cu, hipStream_t s) {
    TsArgs P;
    std::memset(&P, 0, sizeof(P));
    int base = 0, max_box = 0;
    for (int j = 0; j < L.n_jobs; ++j) {
        const TsJobDesc& D = L.job[j];
        TsJob& J = P.job[j];
        J.src = D.src; J.dst = D.dst; J.packed = D.packed; J.packed_hi = D.packed_hi;
        J.recs = D.plan->d_recs;
        J.W = D.plan->W; J.H = D.plan->H; J.h = D.plan->h; J.w = D.plan->w;
        J.src_stride = (int32_t)D.src_stride; J.dst_stride = (int32_t)D.dst_stride;
        J.tight = D.dst_stride == (int64_t)3 * D.plan->w ? 1 : 0;
        J.tile_base = base;
        J.fillpk = (uint32_t)D.fill * 0x010101u;
        base += D.plan->n_tiles;
        max_box = max_box > D.plan->max_box ? max_box : D.plan->max_box;
    }
    if (base == 0) return hipSuccess;
    P.n_jobs = L.n_jobs; P.total_tiles = base; P.chunk = (base + 7) / 8; P.R = L.R;
    P.buf_bytes = (max_box + 16 + 63) & ~63;
    P.cvalpk = (uint32_t)L.cval[0] | ((uint32_t)L.cval[1] << 8) | ((uint32_t)L.cval[2] << 16);
    for (int i = 0; i < 4; ++i) P.cval[i] = L.cval[i];
    const size_t lds = 2 * (size_t)P.buf_bytes;
    // persistent workgroups: as many as the CUs hold (LDS; at most three), never more than the tiles of an XCD's chunk
    int per_cu = (int)(lds_per_cu / (lds + 1024));
    per_cu = per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu);
    if (L.wg_per_cu > 0) per_cu = L.wg_per_cu;
    int nj = (n_cu * per_cu) / 8;
    nj = nj < 1 ? 1 : (nj > P.chunk ? P.chunk : nj);
    const void* kern = (const void*)table_staged_kernel<kTsWaves>;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    void* args[] = {(void*)&P};
    return hipLaunchKernel(kern, dim3((unsigned)(nj * 8)), dim3((unsigned)(64 * kTsWaves)), args, lds, s);
}

}  // namespace gs360
