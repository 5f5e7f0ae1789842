// Round-5 study harness for the source-major equirect kernel: a workgroup streams ONE source tile into LDS (global_load_lds_dwordx4)
// and renders every output pixel of every view whose tap pair starts inside it, from a static plan (make_plan.py).
// MODE 0: full kernel; 1: replay (DMA + LDS reads + stores, no blend arithmetic); 2: DMA only; 3: no DMA (plan + LDS + blend + stores)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o srcmajor_probe srcmajor_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#ifndef SM_DMA_AUX
#define SM_DMA_AUX 0
#endif
#ifndef SM_THIN
#define SM_THIN 63
#endif
#define SM_VMCNT(K) (((K) & 15) | (7 << 4) | (15 << 8) | (((K) >> 4) << 14))
#ifndef SM_NT_STORE
#define SM_NT_STORE 0
#endif
#ifndef SM_NOMASK
#define SM_NOMASK 0
#endif
#ifndef SM_LOADER_PRIO
#define SM_LOADER_PRIO 0
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_void_t;
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int kMaxFrames = 16, kMaxViews = 16;
struct SmTile { int32_t x0, y0, nrows, wch, ebeg, ecnt, pad0, pad1; };
struct SmLaunch {
    const uint8_t* src[kMaxFrames];
    uint8_t* dst[kMaxFrames * kMaxViews];
    const SmTile* tiles;
    const uint2* entries;
    int32_t W, H, N, w, h, PB, n_tiles, n_frames;
    int32_t per_frame, total, chunk;
    int32_t qmap[kMaxViews];      // ring position q -> view index of the call
    int64_t src_stride, dst_stride;
};

__device__ __forceinline__ int dot2_i16(uint32_t taps, uint32_t weights, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, taps), __builtin_bit_cast(s16x2, weights), acc, false);
}
#define PAIR(lo, hi) (0x0c000c00u | ((uint32_t)(hi) << 16) | (uint32_t)(lo))

template <int MODE>
__global__ __launch_bounds__(256) void srcmajor_kernel(const SmLaunch L) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    __shared__ uint8_t* s_dst[kMaxViews];
    const int b = blockIdx.x;
    const int t = (b & 7) * L.chunk + (b >> 3);
    if (t >= L.total) return;
    const int f = t / L.per_frame;
    int r = t - f * L.per_frame;
    const int n_img = 2 * L.N;
    const int ti = r / n_img, img = r - ti * n_img;
    const int k = img >> 1;
    const bool flip = img & 1;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < L.N) {
        int q = tid + k; if (q >= L.N) q -= L.N;
        s_dst[tid] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    // ---- stage the tile: chunk c = (row, col) -> LDS byte 16 c
    if (MODE != 3) {
        const int total = T.nrows * T.wch;
        int xk = T.x0 + k * L.PB;
        for (int c0 = wave * 64; c0 < total; c0 += 256) {
            const int c = c0 + lane;
            if (c < total) {
                const int row = c / T.wch, col = c - row * T.wch;
                int y = flip ? L.H - 1 - (T.y0 + row) : T.y0 + row;
                y = min(max(y, 0), L.H - 1);
                int x = xk + col * 16;
                if (x >= rowbytes) x -= rowbytes;
                if (x >= rowbytes) x -= rowbytes;
                const uint8_t* g = src + (size_t)y * L.src_stride + x;
                __builtin_amdgcn_global_load_lds((global_void_t*)g, (lds_void_t*)(s_tile + (size_t)c0 * 16), 16, 0, 0);
            }
        }
    }
    if (MODE == 2) { __builtin_amdgcn_s_waitcnt(0x0F70); return; }
    const uint2* __restrict__ ent = L.entries + T.ebeg;
    uint2 e = make_uint2(0, 0);
    if (tid < T.ecnt) e = ent[tid];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const int pitch = T.wch * 16;
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const int rowpx = L.w;
    for (int i0 = 0; i0 < T.ecnt; i0 += 256) {
        const uint2 cur = e;
        if (i0 + 256 + tid < T.ecnt) e = ent[i0 + 256 + tid];
        const bool valid = (cur.x >> 28) & 1u, full = (cur.x >> 29) & 1u, noflip = (cur.x >> 30) & 1u;
        const uint32_t o0 = cur.y & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
        const int fx = (cur.y >> 17) & 31, fy = (cur.y >> 22) & 31;
        const uint32_t* qa = reinterpret_cast<const uint32_t*>(s_tile + (o0 & ~3u));
        const uint32_t* qb = reinterpret_cast<const uint32_t*>(s_tile + (o1 & ~3u));
        const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
        uint32_t pk;
        if (MODE == 1) {
            pk = (a0 ^ a1 ^ a2 ^ b0 ^ b1 ^ b2) & 0xffffffu;
        } else {
            const uint32_t t0x = __builtin_amdgcn_alignbyte(a1, a0, o0), t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
            const uint32_t t1x = __builtin_amdgcn_alignbyte(b1, b0, o1), t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
            const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
            const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
            const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
            const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
            const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
            pk = c0 | (c1 << 8) | (c2 << 16);
        }
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
        const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
        const int i = cur.x & 0xfff, j = (cur.x >> 12) & 0xfff, vrel = (cur.x >> 24) & 15;
        const int jj = flip ? L.h - 1 - j : j;
        uint8_t* const d = s_dst[vrel];
        const bool live = valid && !(flip && noflip);
        const uint32_t off = (uint32_t)(jj * rowpx + i) * 3u;
        if (live) {
            if (full) {
                if (k4 < 3) *reinterpret_cast<uint32_t*>(d + off + k4) = dw;
            } else {
                d[off] = (uint8_t)pk; d[off + 1] = (uint8_t)(pk >> 8); d[off + 2] = (uint8_t)(pk >> 16);
            }
        }
    }
}


// v2: one LOADER wavefront streams the G images of a canonical tile into two alternating LDS buffers while four CONSUMER
// wavefronts render from the buffer that has landed (the consumers' own vmcnt queue never holds a DMA: their plan-entry waits
// do not wait for the next tile).  Entries are padded to whole wavefronts with copies of real quads: no predicate, one dword store
// per pixel slot, straight-line loop body (the compiler counts vmcnt exactly).
struct SmLaunch2 { SmLaunch L; int32_t G, groups_per_tile, groups_per_frame, total_groups, gchunk, buf_bytes; unsigned long long* dbg; };

template <int MODE, int LDSRD>
__global__ __launch_bounds__(320) void srcmajor2_kernel(const SmLaunch2 P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int total = T.nrows * T.wch;
    auto dma = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * L.PB;
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int c = c0 + lane;
            if (c < total) {
                const int row = c / T.wch, col = c - row * T.wch;
                int y = flip ? L.H - 1 - (T.y0 + row) : T.y0 + row;
                y = min(max(y, 0), L.H - 1);
                int x = xk + col * 16;
                if (x >= rowbytes) x -= rowbytes;
                if (x >= rowbytes) x -= rowbytes;
                const uint8_t* gp = src + (size_t)y * L.src_stride + x;
                __builtin_amdgcn_global_load_lds((global_void_t*)gp, (lds_void_t*)(buf + (size_t)c0 * 16), 16, 0, 0);
            }
        }
    };
    if (wave == 0) {
        if (MODE != 3) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    const int pitch = T.wch * 16;
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint2* __restrict__ ent = L.entries + T.ebeg;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (wave == 0) {
            if (g + 1 < G && MODE != 3) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            __builtin_amdgcn_s_waitcnt(0x0F70);
        } else if (MODE != 2) {
            const bool flip = (g0 + g) & 1;
            int i0 = (wave - 1) * 64;
            uint2 e = ent[i0 + lane];
            __builtin_amdgcn_s_waitcnt(0x0F70);      // nothing pending at loop entry: the waits inside the loop then count exactly
            for (; i0 < T.ecnt; i0 += 256) {
                const uint2 cur = e;
                e = ent[i0 + 256 + lane];
                const uint32_t o0 = cur.y & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                const int fx = (cur.y >> 17) & 31, fy = (cur.y >> 22) & 31;
                uint32_t t0x, t0y, t1x, t1y;
                if (LDSRD == 0) {
                    const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                    const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                    const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
                    t0x = __builtin_amdgcn_alignbyte(a1, a0, o0); t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                    t1x = __builtin_amdgcn_alignbyte(b1, b0, o1); t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                } else {
                    const uint2* qa = reinterpret_cast<const uint2*>(cur_buf + (o0 & ~7u));
                    const uint2* qb = reinterpret_cast<const uint2*>(cur_buf + (o1 & ~7u));
                    const uint2 a01 = qa[0], a23 = qa[1], b01 = qb[0], b23 = qb[1];
                    const bool ha = o0 & 4u, hb = o1 & 4u;
                    const uint32_t a0 = ha ? a01.y : a01.x, a1 = ha ? a23.x : a01.y, a2 = ha ? a23.y : a23.x;
                    const uint32_t b0 = hb ? b01.y : b01.x, b1 = hb ? b23.x : b01.y, b2 = hb ? b23.y : b23.x;
                    t0x = __builtin_amdgcn_alignbyte(a1, a0, o0); t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                    t1x = __builtin_amdgcn_alignbyte(b1, b0, o1); t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                }
                uint32_t pk;
                if (MODE == 1) {
                    pk = (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                } else {
                    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                    const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                    const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                    const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                    const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                    pk = c0 | (c1 << 8) | (c2 << 16);
                }
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
                const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                const int i = cur.x & 0xfff, j = (cur.x >> 12) & 0xfff, vrel = (cur.x >> 24) & 15;
                const int jj = flip ? L.h - 1 - j : j;
                uint8_t* const d = s_dst[g * kMaxViews + vrel];
                // lane 3 of a quad has no dword of its own: it repeats lane 2's store (same value, same address) so that the store is
                // unconditional and the loop body stays straight-line
                const uint32_t off = (uint32_t)(jj * L.w + i) * 3u + (uint32_t)k4;
                const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, false);    // quad_perm [0,1,2,2]
                const uint32_t ofq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)off, 0xA4, 0xf, 0xf, false);
                *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + ofq) = dwq;    // a GLOBAL store: a flat one (pointer from LDS) cannot be counted
            }
        }
        __builtin_amdgcn_s_barrier();
    }
}

// v3: pixel ownership (56 MB of tiles per frame), per tile a list of full output quads (one dword store per slot) and a list of single
// pixels (three byte stores), both padded to whole wavefronts with copies; one loader wavefront with scalar row bases (no per-lane
// address arithmetic per DMA instruction) + NCW consumer wavefronts; two alternating LDS buffers.
template <int MODE, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void srcmajor3_kernel(const SmLaunch2 P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int pitch = T.wch * 16;
    auto dma = [&](const int img, uint8_t* const buf) {          // loader wavefront: one instruction per (row, block of 64 chunks)
        const int k = img >> 1;
        const bool flip = img & 1;
        for (int cb = 0; cb < T.wch; cb += 64) {
            int x = T.x0 + k * L.PB + (cb + lane) * 16;
            if (x >= rowbytes) x -= rowbytes;
            if (x >= rowbytes) x -= rowbytes;
            if (cb + lane < T.wch) {
                for (int row = 0; row < T.nrows; ++row) {
                    int y = flip ? L.H - 1 - (T.y0 + row) : T.y0 + row;
                    y = min(max(y, 0), L.H - 1);
                    const uint8_t* rowp = src + (size_t)y * L.src_stride;
                    __builtin_amdgcn_global_load_lds((global_void_t*)(rowp + (uint32_t)x), (lds_void_t*)(buf + row * pitch + cb * 16), 16, 0, SM_DMA_AUX);
                }
            }
        }
    };
    if (wave == 0) {
        if (SM_LOADER_PRIO) __builtin_amdgcn_s_setprio(SM_LOADER_PRIO);
        if (MODE != 3) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint2* __restrict__ entq = L.entries + T.ebeg;
    const uint2* __restrict__ ents = L.entries + T.pad0;
    const int qcnt = T.ecnt, scnt = T.pad1;
    // every image of the tile replays the same lists: a consumer wavefront keeps its first two chunks of each in registers for the whole
    // workgroup (an image then starts without a memory round trip) and prefetches two turns ahead inside the loops
    uint2 q0 = make_uint2(0, 0), q1 = q0, z0 = q0, z1 = q0;
    if (wave > 0) {
        const int i0 = (wave - 1) * 64 + lane;
        q0 = entq[i0]; q1 = entq[i0 + 64 * NCW]; z0 = ents[i0]; z1 = ents[i0 + 64 * NCW];
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const bool dbg_on = P.dbg && (b % 61) == 0 && b / 61 < 256;
    unsigned long long* const dbg = P.dbg + (size_t)(b / 61) * 64;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 0] = __builtin_readcyclecounter();
        if (wave == 0) {
            if (g + 1 < G && MODE != 3) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 1] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 2] = __builtin_readcyclecounter();
        } else if (MODE != 2) {
            const bool flip = (g0 + g) & 1;
            auto sample = [&](const uint2 cur, uint32_t& off, uint8_t*& d) -> uint32_t {
                const uint32_t o0 = cur.y & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                const int fx = (cur.y >> 17) & 31, fy = (cur.y >> 22) & 31;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
                const int i = cur.x & 0xfff, j = (cur.x >> 12) & 0xfff, vrel = (cur.x >> 24) & 15;
                const int jj = flip ? L.h - 1 - j : j;
                d = s_dst[g * kMaxViews + vrel];
                off = (uint32_t)(jj * L.w + i) * 3u;
                const uint32_t t0x = __builtin_amdgcn_alignbyte(a1, a0, o0), t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                const uint32_t t1x = __builtin_amdgcn_alignbyte(b1, b0, o1), t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                if (MODE == 1) return (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                return c0 | (c1 << 8) | (c2 << 16);
            };
            {   // full quads: lanes 4m..4m+3 hold pixels 4q..4q+3 of one output row; lanes 0..2 of the quad write its three dwords
                int i0 = (wave - 1) * 64;
                if (i0 < qcnt) {
                    uint2 ea = q0, eb = q1;
                    auto quad_turn = [&](const uint2 cur) {
                        uint32_t off; uint8_t* d;
                        const uint32_t pk = sample(cur, off, d);
                        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
                        const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                        off += (uint32_t)k4;
                        const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, false);    // quad_perm [0,1,2,2]
                        const uint32_t ofq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)off, 0xA4, 0xf, 0xf, false);
                        *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + ofq) = dwq;
                    };
                    // two turns per trip, each register pair reloaded right after its use: entries arrive two turns ahead, no copies
                    for (; i0 < qcnt; i0 += 128 * NCW) {
                        const uint2 ca = ea;
                        ea = entq[i0 + 128 * NCW + lane];
                        quad_turn(ca);
                        if (i0 + 64 * NCW < qcnt) {
                            const uint2 cb = eb;
                            eb = entq[i0 + 192 * NCW + lane];
                            quad_turn(cb);
                        }
                    }
                }
            }
            {   // single pixels (quads cut by a tile boundary): three byte stores
                int i0 = (wave - 1) * 64;
                if (i0 < scnt) {
                    uint2 e = z0, e1 = z1;
                    for (; i0 < scnt; i0 += 64 * NCW) {
                        const uint2 cur = e;
                        e = e1;
                        e1 = ents[i0 + 128 * NCW + lane];
                        uint32_t off; uint8_t* d;
                        const uint32_t pk = sample(cur, off, d);
                        __attribute__((address_space(1))) uint8_t* q = (__attribute__((address_space(1))) uint8_t*)((uintptr_t)d + off);
                        q[0] = (uint8_t)pk; q[1] = (uint8_t)(pk >> 8); q[2] = (uint8_t)(pk >> 16);
                    }
                }
            }
        }
        if (dbg_on && lane == 0 && wave == 1 && g < 6) dbg[24 + g * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 3] = __builtin_readcyclecounter();
    }
}

// v5: a consumer lane renders a whole output QUAD per turn (four pixels = three dwords, one 12-byte store per lane, 768 contiguous bytes
// per wavefront: no cross-lane repack, no duplicate lanes), the quad's header (column, row, view) decoded once; plan entries as five dword
// planes (coalesced dword loads), prefetched one turn ahead.  Singles (quads cut by a tile edge) as before.
template <int MODE, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void srcmajor5_kernel(const SmLaunch2 P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int pitch = T.wch * 16;
    auto dma = [&](const int img, uint8_t* const buf) {          // loader wavefront: one instruction per (row, block of 64 chunks)
        const int k = img >> 1;
        const bool flip = img & 1;
        for (int cb = 0; cb < T.wch; cb += 64) {
            int x = T.x0 + k * L.PB + (cb + lane) * 16;
            if (x >= rowbytes) x -= rowbytes;
            if (x >= rowbytes) x -= rowbytes;
            if (cb + lane < T.wch) {
                for (int row = 0; row < T.nrows; ++row) {
                    int y = flip ? L.H - 1 - (T.y0 + row) : T.y0 + row;
                    y = min(max(y, 0), L.H - 1);
                    const uint8_t* rowp = src + (size_t)y * L.src_stride;
                    __builtin_amdgcn_global_load_lds((global_void_t*)(rowp + (uint32_t)x), (lds_void_t*)(buf + row * pitch + cb * 16), 16, 0, SM_DMA_AUX);
                }
            }
        }
    };
    if (wave == 0) {
        if (SM_LOADER_PRIO) __builtin_amdgcn_s_setprio(SM_LOADER_PRIO);
        if (MODE != 3) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    const uint32_t* __restrict__ pool = reinterpret_cast<const uint32_t*>(L.entries);
    const uint32_t* __restrict__ qpl = pool + T.ebeg;
    const uint2* __restrict__ ents = reinterpret_cast<const uint2*>(pool + T.pad0);
    const int nq = T.ecnt, scnt = T.pad1;
    // first turn of both lists stays in registers for every image of the tile
    uint32_t h0 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0; uint2 z0 = make_uint2(0, 0);
    if (wave > 0) {
        const int i0 = (wave - 1) * 64 + lane;
        h0 = qpl[i0]; a0 = qpl[nq + i0]; a1 = qpl[2 * nq + i0]; a2 = qpl[3 * nq + i0]; a3 = qpl[4 * nq + i0];
        z0 = ents[i0];
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const bool dbg_on = P.dbg && (b % 61) == 0 && b / 61 < 256;
    unsigned long long* const dbg = P.dbg + (size_t)(b / 61) * 64;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 0] = __builtin_readcyclecounter();
        if (wave == 0) {
            if (g + 1 < G && MODE != 3) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 1] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 2] = __builtin_readcyclecounter();
        } else if (MODE != 2) {
            const bool flip = (g0 + g) & 1;
            auto tap = [&](const uint32_t pe) -> uint32_t {          // one pixel: 24-bit packed result
                const uint32_t o0 = pe & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                const int fx = (pe >> 17) & 31, fy = (pe >> 22) & 31;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                const uint32_t r0 = qa[0], r1 = qa[1], r2 = qa[2], s0 = qb[0], s1 = qb[1], s2 = qb[2];
                const uint32_t t0x = __builtin_amdgcn_alignbyte(r1, r0, o0), t0y = __builtin_amdgcn_alignbyte(r2, r1, o0);
                const uint32_t t1x = __builtin_amdgcn_alignbyte(s1, s0, o1), t1y = __builtin_amdgcn_alignbyte(s2, s1, o1);
                if (MODE == 1) return (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                return c0 | (c1 << 8) | (c2 << 16);
            };
            {
                int i0 = (wave - 1) * 64;
                if (i0 < nq) {
                    uint32_t eh = h0, e0 = a0, e1 = a1, e2 = a2, e3 = a3;
                    for (; i0 < nq; i0 += 64 * NCW) {
                        const uint32_t ch = eh, c0 = e0, c1 = e1, c2 = e2, c3 = e3;
                        const int in = i0 + 64 * NCW + lane;
                        eh = qpl[in]; e0 = qpl[nq + in]; e1 = qpl[2 * nq + in]; e2 = qpl[3 * nq + in]; e3 = qpl[4 * nq + in];
                        const uint32_t p0 = tap(c0), p1 = tap(c1), p2 = tap(c2), p3 = tap(c3);
                        const int i = ch & 0xfff, j = (ch >> 12) & 0xfff, vrel = (ch >> 24) & 15;
                        const int jj = flip ? L.h - 1 - j : j;
                        uint8_t* const d = s_dst[g * kMaxViews + vrel];
                        const uint32_t off = (uint32_t)(jj * L.w + i) * 3u;
                        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
                        u32x3 o;
                        o.x = __builtin_amdgcn_perm(p1, p0, 0x04020100u);
                        o.y = __builtin_amdgcn_perm(p2, p1, 0x05040201u);
                        o.z = __builtin_amdgcn_perm(p3, p2, 0x06050402u);
                        *(__attribute__((address_space(1))) u32x3*)((uintptr_t)d + off) = o;
                    }
                }
            }
            {   // single pixels (quads cut by a tile boundary): byte stores
                int i0 = (wave - 1) * 64;
                if (i0 < scnt) {
                    uint2 e = z0;
                    for (; i0 < scnt; i0 += 64 * NCW) {
                        const uint2 cur = e;
                        e = ents[i0 + 64 * NCW + lane];
                        const uint32_t pk = tap(cur.y);
                        const int i = cur.x & 0xfff, j = (cur.x >> 12) & 0xfff, vrel = (cur.x >> 24) & 15;
                        const int jj = flip ? L.h - 1 - j : j;
                        uint8_t* const d = s_dst[g * kMaxViews + vrel];
                        const uint32_t off = (uint32_t)(jj * L.w + i) * 3u;
                        __attribute__((address_space(1))) uint8_t* q = (__attribute__((address_space(1))) uint8_t*)((uintptr_t)d + off);
                        q[0] = (uint8_t)pk; q[1] = (uint8_t)(pk >> 8); q[2] = (uint8_t)(pk >> 16);
                    }
                }
            }
        }
        if (dbg_on && lane == 0 && wave == 1 && g < 6) dbg[24 + g * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 3] = __builtin_readcyclecounter();
    }
}

// v6: quad ownership (every output quad rendered by ONE tile: no byte path), ragged rows (the loader copies per box row only the chunks
// some tap touches), a pixel per lane with 8-byte-aligned LDS reads (64 banks instead of 32: the tap windows of 32 neighbouring pixels
// spread over 110 dwords), quads re-sliced into dwords with two DPP moves, lanes 0..2 of a quad store.
template <int MODE, int NCW, int LDSRD>
__global__ __launch_bounds__(64 * (NCW + 1)) void srcmajor6_kernel(const SmLaunch2 P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_tile[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int pitch = T.wch * 16;
    const int2* __restrict__ rowtab = reinterpret_cast<const int2*>(reinterpret_cast<const int32_t*>(L.entries + T.pad1) + T.pad0);
    // The loader's issue rate decides the stream: one wavefront must put a 1-KiB copy in flight every ~200 cycles, so a row costs a dozen
    // scalar instructions, not forty.  Per image the row base addresses are computed ACROSS the lanes (lane r = box row r: flip, clamp,
    // 64-bit multiply once), the row table (first chunk | count << 16) sits in a register since the start of the workgroup, and a row
    // takes three v_readlane + a scalar add; the per-lane offset is the constant 16 lane unless the box crosses the 360-degree seam.
    int rt_lane = 0;
    if (wave == 0 && lane < T.nrows) { const int2 q = rowtab[lane]; rt_lane = q.x | (q.y << 16); }
    __builtin_amdgcn_s_waitcnt(0x0F70);          // before the first copy is issued: a later wait for the table would wait for the copies too
    auto dma = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * L.PB;                          // < rowbytes
        const bool seam = xk + pitch > rowbytes;                 // wave-uniform
        int y = flip ? L.H - 1 - (T.y0 + lane) : T.y0 + lane;
        y = min(max(y, 0), L.H - 1);
        const uint64_t rb = (uint64_t)(uintptr_t)src + (uint64_t)(uint32_t)y * (uint64_t)L.src_stride + (uint32_t)xk;
        const uint32_t rb_lo = (uint32_t)rb, rb_hi = (uint32_t)(rb >> 32);
        const uint32_t v16 = (uint32_t)lane * 16u;
        for (int row = 0; row < T.nrows; ++row) {
            const uint32_t rt = (uint32_t)__builtin_amdgcn_readlane(rt_lane, row);
            const uint32_t c0 = rt & 0xffffu, cn = rt >> 16;
            const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)rb_hi, row) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)rb_lo, row);
            uint8_t* const ldst = buf + row * pitch + c0 * 16;
            if (!seam) {
                const uint8_t* gp = reinterpret_cast<const uint8_t*>(base + c0 * 16u);
                for (uint32_t cb = 0; cb < cn; cb += 64)
                    if (SM_NOMASK || cb + lane < cn)
                        __builtin_amdgcn_global_load_lds((global_void_t*)(gp + cb * 16 + v16), (lds_void_t*)(ldst + cb * 16), 16, 0, SM_DMA_AUX);
            } else {
                const uint8_t* gp = reinterpret_cast<const uint8_t*>(base - (uint32_t)xk);
                for (uint32_t cb = 0; cb < cn; cb += 64) {
                    int x = xk + (int)(c0 + cb + lane) * 16;
                    if (x >= rowbytes) x -= rowbytes;
                    if (cb + lane < cn)
                        __builtin_amdgcn_global_load_lds((global_void_t*)(gp + (uint32_t)x), (lds_void_t*)(ldst + cb * 16), 16, 0, SM_DMA_AUX);
                }
            }
        }
    };
    if (wave == 0) {
        if (MODE != 3) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint2* __restrict__ entq = L.entries + T.ebeg;
    const int qcnt = T.ecnt;
    uint2 q0 = make_uint2(0, 0);
    if (wave > 0) {
        q0 = entq[(wave - 1) * 64 + lane];
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const bool dbg_on = P.dbg && (b % 61) == 0 && b / 61 < 256;
    unsigned long long* const dbg = P.dbg + (size_t)(b / 61) * 64;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 0] = __builtin_readcyclecounter();
        if (wave == 0) {
            if (g + 1 < G && MODE != 3) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 1] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 2] = __builtin_readcyclecounter();
        } else if (MODE != 2) {
            const bool flip = (g0 + g) & 1;
            const int jbase = flip ? L.h - 1 : 0, jsgn = flip ? -1 : 1;
            int i0 = (wave - 1) * 64;
            if (i0 < qcnt) {
                uint2 e = q0;
                for (; i0 < qcnt; i0 += 64 * NCW) {
                    const uint2 cur = e;
                    e = entq[i0 + 64 * NCW + lane];
                    const uint32_t o0 = cur.y & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                    const int fx = (cur.y >> 17) & 31, fy = (cur.y >> 22) & 31;
                    uint32_t t0x, t0y, t1x, t1y;
                    if (LDSRD == 0) {
                        const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                        const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                        const uint32_t a0 = qa[0], a1 = qa[1], a2 = qa[2], b0 = qb[0], b1 = qb[1], b2 = qb[2];
                        t0x = __builtin_amdgcn_alignbyte(a1, a0, o0); t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                        t1x = __builtin_amdgcn_alignbyte(b1, b0, o1); t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                    } else {
                        // (as inline assembly: the compiler's own 8-byte LDS loads come with an s_waitcnt vmcnt(0) in front -- it cannot tell
                        // that no LDS-DMA of THIS wavefront is in flight -- which would serialise the plan prefetch with every turn)
                        uint2 a01, a23, b01, b23;
                        {
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            u32x4 ra, rb;
                            const uint32_t la = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)(cur_buf) + (o0 & ~7u);
                            const uint32_t lb = la - (o0 & ~7u) + (o1 & ~7u);
                            asm volatile("ds_read2_b64 %0, %2 offset1:1\n\tds_read2_b64 %1, %3 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&v"(ra), "=&v"(rb) : "v"(la), "v"(lb) : "memory");
                            a01 = make_uint2(ra.x, ra.y); a23 = make_uint2(ra.z, ra.w); b01 = make_uint2(rb.x, rb.y); b23 = make_uint2(rb.z, rb.w);
                        }
                        const bool ha = o0 & 4u, hb = o1 & 4u;
                        const uint32_t a0 = ha ? a01.y : a01.x, a1 = ha ? a23.x : a01.y, a2 = ha ? a23.y : a23.x;
                        const uint32_t b0 = hb ? b01.y : b01.x, b1 = hb ? b23.x : b01.y, b2 = hb ? b23.y : b23.x;
                        t0x = __builtin_amdgcn_alignbyte(a1, a0, o0); t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                        t1x = __builtin_amdgcn_alignbyte(b1, b0, o1); t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                    }
                    uint32_t pk;
                    if (MODE == 1) {
                        pk = (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                    } else {
                        const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                        const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                        const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                        const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                        const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                        pk = c0 | (c1 << 8) | (c2 << 16);
                    }
                    const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
                    const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                    const int i = cur.x & 0xfff, j = (cur.x >> 12) & 0xfff, vrel = (cur.x >> 24) & 15;
                    const int jj = jbase + jsgn * j;
                    uint8_t* const d = s_dst[g * kMaxViews + vrel];
                    const uint32_t off = (uint32_t)(jj * L.w + i) * 3u + (uint32_t)k4;
                    // lane 3 of a quad repeats lane 2's store (same dword, same address): an unconditional store keeps the loop straight-line
                    const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, false);    // quad_perm [0,1,2,2]
                    const uint32_t ofq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)off, 0xA4, 0xf, 0xf, false);
                    *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + ofq) = dwq;
                }
            }
        }
        if (dbg_on && lane == 0 && wave == 1 && g < 6) dbg[24 + g * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 3] = __builtin_readcyclecounter();
    }
}

// v7: quad ownership (no byte path), the tile's plan entries LDS-resident (copied once per workgroup by the loader, replayed by every
// image of the tile: the consumers issue no memory reads at all, only their stores), constant exec mask in the loader's row loop.
template <int MODE, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void srcmajor7_kernel(const SmLaunch2 P, const int ent_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    uint8_t* const s_ent = s_lds;                        // [hdr: nq dwords][px words: 4 nq dwords]
    uint8_t* const s_tile = s_lds + ent_bytes;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int pitch = T.wch * 16;
    const int nq = T.ecnt;
    auto dma = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * L.PB;
        for (int cb = 0; cb < T.wch; cb += 64) {
            int x = xk + (cb + lane) * 16;
            if (x >= rowbytes) x -= rowbytes;
            if (x >= rowbytes) x -= rowbytes;
            if (cb + lane < T.wch) {
                int y = flip ? L.H - 1 - T.y0 : T.y0;
                const int ystep = flip ? -1 : 1;
                for (int row = 0; row < T.nrows; ++row, y += ystep) {
                    const int yc = min(max(y, 0), L.H - 1);
                    const uint8_t* rowp = src + (size_t)yc * L.src_stride;
                    __builtin_amdgcn_global_load_lds((global_void_t*)(rowp + (uint32_t)x), (lds_void_t*)(buf + row * pitch + cb * 16), 16, 0, SM_DMA_AUX);
                    if (SM_THIN < 63) __builtin_amdgcn_s_waitcnt(SM_VMCNT(SM_THIN));     // thinned loader: at most SM_THIN + 1 copies of this wavefront in the memory queue
                }
            }
        }
    };
    if (wave == 0) {
        const uint8_t* ge = reinterpret_cast<const uint8_t*>(reinterpret_cast<const uint32_t*>(L.entries) + T.ebeg);
        const int eb = 20 * nq;
        for (int o = 0; o < eb; o += 1024)
            if (o + lane * 16 < eb)
                __builtin_amdgcn_global_load_lds((global_void_t*)(ge + o + lane * 16), (lds_void_t*)(s_ent + o), 16, 0, 0);
        if (MODE != 3) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint32_t* const e_hdr = reinterpret_cast<const uint32_t*>(s_ent);
    const uint32_t* const e_px = e_hdr + nq;
    const int npx = 4 * nq;
    const bool dbg_on = P.dbg && (b % 61) == 0 && b / 61 < 256;
    unsigned long long* const dbg = P.dbg + (size_t)(b / 61) * 64;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 0] = __builtin_readcyclecounter();
        if (wave == 0) {
            if (g + 1 < G && MODE != 3) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 1] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 2] = __builtin_readcyclecounter();
        } else if (MODE != 2) {
            const bool flip = (g0 + g) & 1;
            const int jbase = flip ? L.h - 1 : 0, jsgn = flip ? -1 : 1;
            for (int i0 = (wave - 1) * 64; i0 < npx; i0 += 64 * NCW) {
                const uint32_t pw = e_px[i0 + lane];
                const uint32_t hd = e_hdr[(i0 + lane) >> 2];
                const uint32_t o0 = pw & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                const int fx = (pw >> 17) & 31, fy = (pw >> 22) & 31;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                uint32_t a0, a1, a2, b0, b1, b2;
                if (MODE == 5) { a0 = pw * 3u; a1 = pw * 5u; a2 = pw * 7u; b0 = hd * 3u; b1 = hd * 5u; b2 = hd * 7u; }      // probe: no tap reads from LDS
                else { a0 = qa[0]; a1 = qa[1]; a2 = qa[2]; b0 = qb[0]; b1 = qb[1]; b2 = qb[2]; }
                const uint32_t t0x = __builtin_amdgcn_alignbyte(a1, a0, o0), t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                const uint32_t t1x = __builtin_amdgcn_alignbyte(b1, b0, o1), t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                uint32_t pk;
                if (MODE == 1) {
                    pk = (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                } else {
                    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                    const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                    const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                    const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                    const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                    pk = c0 | (c1 << 8) | (c2 << 16);
                }
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
                const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                const int i = hd & 0xfff, j = (hd >> 12) & 0xfff, vrel = (hd >> 24) & 15;    // the quad's first column
                const int jj = jbase + jsgn * j;
                uint8_t* const d = s_dst[g * kMaxViews + vrel];
                // lane k of a quad writes dword k of its 12 bytes; lane 3 repeats lane 2's store (same value, same address)
                const uint32_t off = (uint32_t)(jj * L.w + i) * 3u + 4u * (uint32_t)min(k4, 2);
                const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, false);    // quad_perm [0,1,2,2]
                if (MODE == 4) { if (dwq == 0x12345678u && fx == 77) *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off) = dwq; }   // probe: no stores
                else if (SM_NT_STORE) __builtin_nontemporal_store(dwq, (__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off));
                else *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off) = dwq;
            }
        }
        if (dbg_on && lane == 0 && wave == 1 && g < 6) dbg[24 + g * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 3] = __builtin_readcyclecounter();
    }
}

// v9 = v7 with class-sorted ragged rows: per box row only the chunks some tap touches, rows sorted by chunk count so that the exec mask
// changes a handful of times per image instead of every row (boxes that cross the seam keep the full-row loader).
// (v7:) quad ownership (no byte path), the tile's plan entries LDS-resident (copied once per workgroup by the loader, replayed by every
// image of the tile: the consumers issue no memory reads at all, only their stores), constant exec mask in the loader's row loop.
template <int MODE, int NCW>
__global__ __launch_bounds__(64 * (NCW + 1)) void srcmajor9_kernel(const SmLaunch2 P, const int ent_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    uint8_t* const s_ent = s_lds;                        // [hdr: nq dwords][px words: 4 nq dwords]
    uint8_t* const s_tile = s_lds + ent_bytes;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int pitch = T.wch * 16;
    const int nq = T.ecnt;
    const uint32_t* __restrict__ pool32 = reinterpret_cast<const uint32_t*>(L.entries);
    const int ncls = T.pad1;
    int cls_lane = 0, row_lane = 0, nlist = 0;
    if (wave == 0) {
        if (lane < ncls) cls_lane = (int)pool32[T.pad0 + lane];
        for (int c = 0; c < ncls; ++c) nlist += __builtin_amdgcn_readlane(cls_lane, c) >> 16;
        if (lane < nlist) row_lane = (int)pool32[T.pad0 + ncls + lane];
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    auto dma_full = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * L.PB;
        for (int cb = 0; cb < T.wch; cb += 64) {
            int x = xk + (cb + lane) * 16;
            if (x >= rowbytes) x -= rowbytes;
            if (x >= rowbytes) x -= rowbytes;
            if (cb + lane < T.wch) {
                int y = flip ? L.H - 1 - T.y0 : T.y0;
                const int ystep = flip ? -1 : 1;
                for (int row = 0; row < T.nrows; ++row, y += ystep) {
                    const int yc = min(max(y, 0), L.H - 1);
                    const uint8_t* rowp = src + (size_t)yc * L.src_stride;
                    __builtin_amdgcn_global_load_lds((global_void_t*)(rowp + (uint32_t)x), (lds_void_t*)(buf + row * pitch + cb * 16), 16, 0, SM_DMA_AUX);
                }
            }
        }
    };
    auto dma = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * L.PB;
        if (xk + pitch > rowbytes || T.wch > 64) { dma_full(img, buf); return; }       // the box crosses the seam: full rows, per-lane wrap
        // lane p = p-th row of the sorted list: its global base (flip, clamp, 64-bit multiply ONCE per image across the lanes) and LDS offset
        const int row = row_lane & 0xff, c0 = row_lane >> 8;
        int y = flip ? L.H - 1 - (T.y0 + row) : T.y0 + row;
        y = min(max(y, 0), L.H - 1);
        const uint64_t rb = (uint64_t)(uintptr_t)src + (uint64_t)(uint32_t)y * (uint64_t)L.src_stride + (uint32_t)(xk + c0 * 16);
        const int rb_lo = (int)(uint32_t)rb, rb_hi = (int)(uint32_t)(rb >> 32);
        const int lo_lane = row * pitch + c0 * 16;
        const uint32_t v16 = (uint32_t)lane * 16u;
        int p = 0;
        for (int c = 0; c < ncls; ++c) {
            const int cw = __builtin_amdgcn_readlane(cls_lane, c);
            const int cnt = cw & 0xffff, kk = cw >> 16;
            if (lane < cnt) {                                     // ONE exec change per class
                for (int i = 0; i < kk; ++i) {
                    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(rb_hi, p + i) << 32) | (uint32_t)__builtin_amdgcn_readlane(rb_lo, p + i);
                    const int lo = __builtin_amdgcn_readlane(lo_lane, p + i);
                    __builtin_amdgcn_global_load_lds((global_void_t*)(reinterpret_cast<const uint8_t*>(base) + v16), (lds_void_t*)(buf + lo), 16, 0, SM_DMA_AUX);
                }
            }
            p += kk;
        }
    };
    if (wave == 0) {
        const uint8_t* ge = reinterpret_cast<const uint8_t*>(reinterpret_cast<const uint32_t*>(L.entries) + T.ebeg);
        const int eb = 20 * nq;
        for (int o = 0; o < eb; o += 1024)
            if (o + lane * 16 < eb)
                __builtin_amdgcn_global_load_lds((global_void_t*)(ge + o + lane * 16), (lds_void_t*)(s_ent + o), 16, 0, 0);
        if (MODE != 3) dma(g0, s_tile);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    __builtin_amdgcn_s_barrier();
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint32_t* const e_hdr = reinterpret_cast<const uint32_t*>(s_ent);
    const uint32_t* const e_px = e_hdr + nq;
    const int npx = 4 * nq;
    const bool dbg_on = P.dbg && (b % 61) == 0 && b / 61 < 256;
    unsigned long long* const dbg = P.dbg + (size_t)(b / 61) * 64;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 0] = __builtin_readcyclecounter();
        if (wave == 0) {
            if (g + 1 < G && MODE != 3) dma(g0 + g + 1, s_tile + ((g + 1) & 1) * P.buf_bytes);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 1] = __builtin_readcyclecounter();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (dbg_on && lane == 0 && g < 6) dbg[g * 4 + 2] = __builtin_readcyclecounter();
        } else if (MODE != 2) {
            const bool flip = (g0 + g) & 1;
            const int jbase = flip ? L.h - 1 : 0, jsgn = flip ? -1 : 1;
            for (int i0 = (wave - 1) * 64; i0 < npx; i0 += 64 * NCW) {
                const uint32_t pw = e_px[i0 + lane];
                const uint32_t hd = e_hdr[(i0 + lane) >> 2];
                const uint32_t o0 = pw & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                const int fx = (pw >> 17) & 31, fy = (pw >> 22) & 31;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                uint32_t a0, a1, a2, b0, b1, b2;
                if (MODE == 5) { a0 = pw * 3u; a1 = pw * 5u; a2 = pw * 7u; b0 = hd * 3u; b1 = hd * 5u; b2 = hd * 7u; }      // probe: no tap reads from LDS
                else { a0 = qa[0]; a1 = qa[1]; a2 = qa[2]; b0 = qb[0]; b1 = qb[1]; b2 = qb[2]; }
                const uint32_t t0x = __builtin_amdgcn_alignbyte(a1, a0, o0), t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                const uint32_t t1x = __builtin_amdgcn_alignbyte(b1, b0, o1), t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                uint32_t pk;
                if (MODE == 1) {
                    pk = (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                } else {
                    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                    const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                    const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                    const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                    const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                    pk = c0 | (c1 << 8) | (c2 << 16);
                }
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
                const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                const int i = hd & 0xfff, j = (hd >> 12) & 0xfff, vrel = (hd >> 24) & 15;    // the quad's first column
                const int jj = jbase + jsgn * j;
                uint8_t* const d = s_dst[g * kMaxViews + vrel];
                // lane k of a quad writes dword k of its 12 bytes; lane 3 repeats lane 2's store (same value, same address)
                const uint32_t off = (uint32_t)(jj * L.w + i) * 3u + 4u * (uint32_t)min(k4, 2);
                const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, false);    // quad_perm [0,1,2,2]
                if (MODE == 4) { if (dwq == 0x12345678u && fx == 77) *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off) = dwq; }   // probe: no stores
                else if (SM_NT_STORE) __builtin_nontemporal_store(dwq, (__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off));
                else *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off) = dwq;
            }
        }
        if (dbg_on && lane == 0 && wave == 1 && g < 6) dbg[24 + g * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (dbg_on && lane == 0 && wave < 2 && g < 6) dbg[wave * 24 + g * 4 + 3] = __builtin_readcyclecounter();
    }
}


// v8 = v7 with TWO loader wavefronts (even / odd images): a tile image is requested two turns ahead, so both LDS buffers are landing
// zones except while one is being rendered (twice the bytes in flight per CU).
// (v7:) quad ownership (no byte path), the tile's plan entries LDS-resident (copied once per workgroup by the loader, replayed by every
// image of the tile: the consumers issue no memory reads at all, only their stores), constant exec mask in the loader's row loop.
template <int MODE, int NCW>
__global__ __launch_bounds__(64 * (NCW + 2)) void srcmajor8_kernel(const SmLaunch2 P, const int ent_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_lds[];
    __shared__ uint8_t* s_dst[12 * kMaxViews];
    const SmLaunch& L = P.L;
    const int b = blockIdx.x;
    const int t = (b & 7) * P.gchunk + (b >> 3);
    if (t >= P.total_groups) return;
    const int f = t / P.groups_per_frame;
    const int r = t - f * P.groups_per_frame;
    const int ti = r / P.groups_per_tile, g0 = (r - ti * P.groups_per_tile) * P.G;
    const SmTile T = L.tiles[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = P.G;
    uint8_t* const s_ent = s_lds;                        // [hdr: nq dwords][px words: 4 nq dwords]
    uint8_t* const s_tile = s_lds + ent_bytes;
    if (tid < G * L.N) {
        const int g = tid / L.N, v = tid - g * L.N;
        int q = v + ((g0 + g) >> 1); if (q >= L.N) q -= L.N;
        s_dst[g * kMaxViews + v] = L.dst[f * L.N + L.qmap[q]];
    }
    const uint8_t* __restrict__ src = L.src[f];
    const int rowbytes = 3 * L.W;
    const int pitch = T.wch * 16;
    const int nq = T.ecnt;
    auto dma = [&](const int img, uint8_t* const buf) {
        const int k = img >> 1;
        const bool flip = img & 1;
        const int xk = T.x0 + k * L.PB;
        for (int cb = 0; cb < T.wch; cb += 64) {
            int x = xk + (cb + lane) * 16;
            if (x >= rowbytes) x -= rowbytes;
            if (x >= rowbytes) x -= rowbytes;
            if (cb + lane < T.wch) {
                int y = flip ? L.H - 1 - T.y0 : T.y0;
                const int ystep = flip ? -1 : 1;
                for (int row = 0; row < T.nrows; ++row, y += ystep) {
                    const int yc = min(max(y, 0), L.H - 1);
                    const uint8_t* rowp = src + (size_t)yc * L.src_stride;
                    __builtin_amdgcn_global_load_lds((global_void_t*)(rowp + (uint32_t)x), (lds_void_t*)(buf + row * pitch + cb * 16), 16, 0, SM_DMA_AUX);
                    if (SM_THIN < 63) __builtin_amdgcn_s_waitcnt(SM_VMCNT(SM_THIN));     // thinned loader: at most SM_THIN + 1 copies of this wavefront in the memory queue
                }
            }
        }
    };
    if (wave == 0) {
        const uint8_t* ge = reinterpret_cast<const uint8_t*>(reinterpret_cast<const uint32_t*>(L.entries) + T.ebeg);
        const int eb = 20 * nq;
        for (int o = 0; o < eb; o += 1024)
            if (o + lane * 16 < eb)
                __builtin_amdgcn_global_load_lds((global_void_t*)(ge + o + lane * 16), (lds_void_t*)(s_ent + o), 16, 0, 0);
        if (MODE != 3) dma(g0, s_tile);
    } else if (wave == 1) {
        if (MODE != 3 && G > 1) dma(g0 + 1, s_tile + P.buf_bytes);
    }
    const int k4 = lane & 3;
    const uint32_t sel = k4 == 0 ? 0x04020100u : (k4 == 1 ? 0x05040201u : 0x06050402u);
    const uint32_t* const e_hdr = reinterpret_cast<const uint32_t*>(s_ent);
    const uint32_t* const e_px = e_hdr + nq;
    const int npx = 4 * nq;
    const bool dbg_on = P.dbg && (b % 61) == 0 && b / 61 < 256;
    unsigned long long* const dbg = P.dbg + (size_t)(b / 61) * 64;
    for (int g = 0; g < G; ++g) {
        uint8_t* const cur_buf = s_tile + (g & 1) * P.buf_bytes;
        if (dbg_on && lane == 0 && wave < 3 && g < 6) dbg[(wave == 2) * 24 + g * 4 + 0] = __builtin_readcyclecounter();
        if (wave == (g & 1)) __builtin_amdgcn_s_waitcnt(0x0F70);          // this loader's image (and, the first time, the plan entries) has landed
        if (dbg_on && lane == 0 && wave == (g & 1) && g < 6) dbg[g * 4 + 2] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();
        if (wave >= 2 && MODE != 2) {
            const bool flip = (g0 + g) & 1;
            const int jbase = flip ? L.h - 1 : 0, jsgn = flip ? -1 : 1;
            for (int i0 = (wave - 2) * 64; i0 < npx; i0 += 64 * NCW) {
                const uint32_t pw = e_px[i0 + lane];
                const uint32_t hd = e_hdr[(i0 + lane) >> 2];
                const uint32_t o0 = pw & 0x1ffffu, o1 = o0 + (uint32_t)pitch;
                const int fx = (pw >> 17) & 31, fy = (pw >> 22) & 31;
                const uint32_t* qa = reinterpret_cast<const uint32_t*>(cur_buf + (o0 & ~3u));
                const uint32_t* qb = reinterpret_cast<const uint32_t*>(cur_buf + (o1 & ~3u));
                uint32_t a0, a1, a2, b0, b1, b2;
                if (MODE == 5) { a0 = pw * 3u; a1 = pw * 5u; a2 = pw * 7u; b0 = hd * 3u; b1 = hd * 5u; b2 = hd * 7u; }      // probe: no tap reads from LDS
                else { a0 = qa[0]; a1 = qa[1]; a2 = qa[2]; b0 = qb[0]; b1 = qb[1]; b2 = qb[2]; }
                const uint32_t t0x = __builtin_amdgcn_alignbyte(a1, a0, o0), t0y = __builtin_amdgcn_alignbyte(a2, a1, o0);
                const uint32_t t1x = __builtin_amdgcn_alignbyte(b1, b0, o1), t1y = __builtin_amdgcn_alignbyte(b2, b1, o1);
                uint32_t pk;
                if (MODE == 1) {
                    pk = (t0x ^ t0y ^ t1x ^ t1y) & 0xffffffu;
                } else {
                    const uint32_t ah = (uint32_t)(32 - fx) | ((uint32_t)fx << 16);
                    const uint32_t wr0 = __umul24(ah, (uint32_t)(32 - fy)), wr1 = __umul24(ah, (uint32_t)fy);
                    const uint32_t c0 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1x, t1x, PAIR(0, 3)), wr1, dot2_i16(__builtin_amdgcn_perm(t0x, t0x, PAIR(0, 3)), wr0, 512)) >> 10;
                    const uint32_t c1 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(1, 4)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(1, 4)), wr0, 512)) >> 10;
                    const uint32_t c2 = (uint32_t)dot2_i16(__builtin_amdgcn_perm(t1y, t1x, PAIR(2, 5)), wr1, dot2_i16(__builtin_amdgcn_perm(t0y, t0x, PAIR(2, 5)), wr0, 512)) >> 10;
                    pk = c0 | (c1 << 8) | (c2 << 16);
                }
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pk, 0xF9, 0xf, 0xf, false);   // quad_perm [1,2,3,3]
                const uint32_t dw = __builtin_amdgcn_perm(nxt, pk, sel);
                const int i = hd & 0xfff, j = (hd >> 12) & 0xfff, vrel = (hd >> 24) & 15;    // the quad's first column
                const int jj = jbase + jsgn * j;
                uint8_t* const d = s_dst[g * kMaxViews + vrel];
                // lane k of a quad writes dword k of its 12 bytes; lane 3 repeats lane 2's store (same value, same address)
                const uint32_t off = (uint32_t)(jj * L.w + i) * 3u + 4u * (uint32_t)min(k4, 2);
                const uint32_t dwq = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dw, 0xA4, 0xf, 0xf, false);    // quad_perm [0,1,2,2]
                if (MODE == 4) { if (dwq == 0x12345678u && fx == 77) *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off) = dwq; }   // probe: no stores
                else if (SM_NT_STORE) __builtin_nontemporal_store(dwq, (__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off));
                else *(__attribute__((address_space(1))) uint32_t*)((uintptr_t)d + off) = dwq;
            }
        }
        if (dbg_on && lane == 0 && wave == 2 && g < 6) dbg[24 + g * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();                                      // every consumer is done with cur_buf
        if (wave == (g & 1) && g + 2 < G && MODE != 3) dma(g0 + g + 2, cur_buf);
        if (dbg_on && lane == 0 && wave == (g & 1) && g < 6) dbg[g * 4 + 1] = __builtin_readcyclecounter();
        if (dbg_on && lane == 0 && wave < 3 && g < 6) dbg[(wave == 2) * 24 + g * 4 + 3] = __builtin_readcyclecounter();
    }
}

static uint8_t frame_byte(uint32_t f, uint32_t p) {      // reproducible in numpy (uint32 arithmetic)
    uint32_t x = p * 2654435761u + f * 40503u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return (uint8_t)(x >> 8);
}
__global__ void fill_kernel(uint8_t* d, uint32_t f, size_t n) {
    size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) { uint32_t x = (uint32_t)p * 2654435761u + f * 40503u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; d[p] = (uint8_t)(x >> 8); }
}

int main(int argc, char** argv) {
    const char* plan = argc > 1 ? argv[1] : "plan.bin";
    const int mode_only = argc > 2 ? atoi(argv[2]) : -1;
    const char* dump = (argc > 3 && argv[3][0]) ? argv[3] : nullptr;
    FILE* fp = fopen(plan, "rb");
    if (!fp) { printf("no plan %s\n", plan); return 1; }
    int32_t hdr[16];
    if (fread(hdr, 4, 16, fp) != 16) return 1;
    const int W = hdr[0], H = hdr[1], N = hdr[2], w = hdr[3], h = hdr[4], PB = hdr[5], n_tiles = hdr[8], n_ent = hdr[9], max_lds = hdr[10];
    std::vector<SmTile> tiles(n_tiles);
    std::vector<uint2> ents(n_ent);
    if (fread(tiles.data(), sizeof(SmTile), n_tiles, fp) != (size_t)n_tiles) return 1;
    if (fread(ents.data(), 8, n_ent, fp) != (size_t)n_ent) return 1;
    fclose(fp);
    printf("plan %s: W %d H %d N %d w %d h %d PB %d Bx %d R %d tiles %d entries %d max_lds %d\n", plan, W, H, N, w, h, PB, hdr[6], hdr[7], n_tiles, n_ent, max_lds);
    const int F = 16;
    const size_t fbytes = (size_t)W * H * 3, vbytes = (size_t)w * h * 3;
    SmLaunch L; memset(&L, 0, sizeof L);
    for (int f = 0; f < F; ++f) {
        uint8_t* d; CK(hipMalloc((void**)&d, fbytes + 256));
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((fbytes + 255) / 256)), dim3(256), 0, 0, d, (uint32_t)f, fbytes);
        L.src[f] = d;
        for (int v = 0; v < N; ++v) { uint8_t* o; CK(hipMalloc((void**)&o, vbytes + 256)); CK(hipMemset(o, 0xEE, vbytes)); L.dst[f * N + v] = o; }
    }
    SmTile* dt; uint2* de;
    CK(hipMalloc((void**)&dt, n_tiles * sizeof(SmTile))); CK(hipMemcpy(dt, tiles.data(), n_tiles * sizeof(SmTile), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&de, (size_t)n_ent * 8 + 65536)); CK(hipMemcpy(de, ents.data(), (size_t)n_ent * 8, hipMemcpyHostToDevice));
    L.tiles = dt; L.entries = de; L.W = W; L.H = H; L.N = N; L.w = w; L.h = h; L.PB = PB; L.n_tiles = n_tiles; L.n_frames = F;
    L.per_frame = n_tiles * 2 * N; L.total = L.per_frame * F; L.chunk = (L.total + 7) / 8;
    for (int q = 0; q < N; ++q) L.qmap[q] = q;
    L.src_stride = 3 * W; L.dst_stride = 3 * w;
    const int grid = L.chunk * 8;
    const size_t lds = (size_t)max_lds + 64;
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = argc > 4 ? atoi(argv[4]) : 6;
    const int ldsrd = argc > 5 ? atoi(argv[5]) : 1;
    SmLaunch2 P; P.L = L; P.G = G; P.groups_per_tile = 2 * N / G; P.groups_per_frame = n_tiles * P.groups_per_tile;
    P.total_groups = P.groups_per_frame * F; P.gchunk = (P.total_groups + 7) / 8; P.buf_bytes = (max_lds + 63) & ~63;
    P.dbg = nullptr;
    const int grid2 = P.gchunk * 8;
    const size_t lds2 = 2 * (size_t)P.buf_bytes + 64;
    const int ncw = argc > 6 ? atoi(argv[6]) : 0;
    const int ent_bytes = (hdr[11] + 63) & ~63;
    auto launch2 = [&](int mode) {
#define L3(M, C) { CK(hipFuncSetAttribute((const void*)srcmajor3_kernel<M, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); hipLaunchKernelGGL((srcmajor3_kernel<M, C>), dim3(grid2), dim3(64 * (C + 1)), lds2, 0, P); }
#define L5(M, C) { CK(hipFuncSetAttribute((const void*)srcmajor5_kernel<M, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); hipLaunchKernelGGL((srcmajor5_kernel<M, C>), dim3(grid2), dim3(64 * (C + 1)), lds2, 0, P); }
#define L6(M, C, R) { CK(hipFuncSetAttribute((const void*)srcmajor6_kernel<M, C, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); hipLaunchKernelGGL((srcmajor6_kernel<M, C, R>), dim3(grid2), dim3(64 * (C + 1)), lds2, 0, P); }
        if (ncw == 64) { if (ldsrd) { if (mode == 0) L6(0, 4, 1) if (mode == 1) L6(1, 4, 1) if (mode == 2) L6(2, 4, 1) if (mode == 3) L6(3, 4, 1) } else { if (mode == 0) L6(0, 4, 0) if (mode == 1) L6(1, 4, 0) if (mode == 2) L6(2, 4, 0) if (mode == 3) L6(3, 4, 0) } return; }
        if (ncw == 68) { if (ldsrd) { if (mode == 0) L6(0, 8, 1) if (mode == 1) L6(1, 8, 1) if (mode == 2) L6(2, 8, 1) if (mode == 3) L6(3, 8, 1) } else { if (mode == 0) L6(0, 8, 0) if (mode == 1) L6(1, 8, 0) if (mode == 2) L6(2, 8, 0) if (mode == 3) L6(3, 8, 0) } return; }
#define L7(M, C) { const size_t l7 = lds2 + ent_bytes; CK(hipFuncSetAttribute((const void*)srcmajor7_kernel<M, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l7)); hipLaunchKernelGGL((srcmajor7_kernel<M, C>), dim3(grid2), dim3(64 * (C + 1)), l7, 0, P, ent_bytes); }
        if (ncw == 74) { if (mode == 0) L7(0, 4) if (mode == 1) L7(1, 4) if (mode == 2) L7(2, 4) if (mode == 3) L7(3, 4) return; }
        if (ncw == 78) { if (mode == 0) L7(0, 8) if (mode == 1) L7(1, 8) if (mode == 2) L7(2, 8) if (mode == 3) L7(3, 8) if (mode == 4) L7(4, 8) if (mode == 5) L7(5, 8) return; }
#define L8(M, C) { const size_t l7 = lds2 + ent_bytes; CK(hipFuncSetAttribute((const void*)srcmajor8_kernel<M, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l7)); hipLaunchKernelGGL((srcmajor8_kernel<M, C>), dim3(grid2), dim3(64 * (C + 2)), l7, 0, P, ent_bytes); }
#define L9(M, C) { const size_t l7 = lds2 + ent_bytes; CK(hipFuncSetAttribute((const void*)srcmajor9_kernel<M, C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l7)); hipLaunchKernelGGL((srcmajor9_kernel<M, C>), dim3(grid2), dim3(64 * (C + 1)), l7, 0, P, ent_bytes); }
        if (ncw == 98) { if (mode == 0) L9(0, 8) if (mode == 1) L9(1, 8) if (mode == 2) L9(2, 8) if (mode == 3) L9(3, 8) if (mode == 4) L9(4, 8) if (mode == 5) L9(5, 8) return; }
        if (ncw == 84) { if (mode == 0) L8(0, 4) if (mode == 1) L8(1, 4) if (mode == 2) L8(2, 4) if (mode == 3) L8(3, 4) return; }
        if (ncw == 86) { if (mode == 0) L8(0, 6) if (mode == 1) L8(1, 6) if (mode == 2) L8(2, 6) if (mode == 3) L8(3, 6) return; }
        if (ncw == 88) { if (mode == 0) L8(0, 8) if (mode == 1) L8(1, 8) if (mode == 2) L8(2, 8) if (mode == 3) L8(3, 8) if (mode == 4) L8(4, 8) if (mode == 5) L8(5, 8) return; }
        if (ncw == 52) { if (mode == 0) L5(0, 2) if (mode == 1) L5(1, 2) if (mode == 2) L5(2, 2) if (mode == 3) L5(3, 2) return; }
        if (ncw == 53) { if (mode == 0) L5(0, 3) if (mode == 1) L5(1, 3) if (mode == 2) L5(2, 3) if (mode == 3) L5(3, 3) return; }
        if (ncw == 54) { if (mode == 0) L5(0, 4) if (mode == 1) L5(1, 4) if (mode == 2) L5(2, 4) if (mode == 3) L5(3, 4) return; }
        if (ncw == 4) { if (mode == 0) L3(0, 4) if (mode == 1) L3(1, 4) if (mode == 2) L3(2, 4) if (mode == 3) L3(3, 4) return; }
        if (ncw == 8) { if (mode == 0) L3(0, 8) if (mode == 1) L3(1, 8) if (mode == 2) L3(2, 8) if (mode == 3) L3(3, 8) return; }
#define L2(M, R) { CK(hipFuncSetAttribute((const void*)srcmajor2_kernel<M, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); hipLaunchKernelGGL((srcmajor2_kernel<M, R>), dim3(grid2), dim3(320), lds2, 0, P); }
        if (ldsrd == 0) { if (mode == 0) L2(0, 0) if (mode == 1) L2(1, 0) if (mode == 2) L2(2, 0) if (mode == 3) L2(3, 0) }
        else { if (mode == 0) L2(0, 1) if (mode == 1) L2(1, 1) if (mode == 2) L2(2, 1) if (mode == 3) L2(3, 1) }
    };
    { int nb = 0;
      if (ncw == 54) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)srcmajor5_kernel<0, 4>, 320, lds2));
      if (ncw == 52) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)srcmajor5_kernel<0, 2>, 192, lds2));
      if (ncw == 8) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)srcmajor3_kernel<0, 8>, 576, lds2));
      printf("occupancy API: %d workgroups per CU (dynamic LDS %zu)\n", nb, lds2); }
    for (int mode = 0; mode < 6; ++mode) {
        if (mode_only >= 0 && mode != mode_only) continue;
        if (mode >= 4 && ncw != 78 && ncw != 88 && ncw != 98) continue;
        const int NSET = getenv("SM_QUICK") ? 3 : 400, NIT = getenv("SM_QUICK") ? 5 : 100;
        for (int i = 0; i < NSET; ++i) launch2(mode);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < NIT; ++i) launch2(mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= NIT;
        printf("v%d ncw %d G %d ldsrd %d mode %d: %.1f us/launch = %.2f us/frame (grid %d, lds %zu)\n", ncw ? 3 : 2, ncw, G, ldsrd, mode, ms * 1e3, ms * 1e3 / F, grid2, lds2);
    }
    if (getenv("SM_DBG")) {
        unsigned long long* d; CK(hipMalloc((void**)&d, 256 * 64 * 8)); CK(hipMemset(d, 0, 256 * 64 * 8));
        for (int i = 0; i < 50; ++i) launch2(0);
        P.dbg = d;
        launch2(0);
        CK(hipDeviceSynchronize());
        P.dbg = nullptr;
        std::vector<unsigned long long> hb(256 * 64);
        CK(hipMemcpy(hb.data(), d, 256 * 64 * 8, hipMemcpyDeviceToHost));
        double s_issue = 0, s_land = 0, s_comp = 0, s_item = 0, s_lbar = 0, s_cbar = 0; int n = 0;
        for (int w = 0; w < 256; ++w) {
            const unsigned long long* q = hb.data() + (size_t)w * 64;
            for (int g = 1; g < 5 && g + 1 < G; ++g) {
                if (!q[g * 4] || !q[24 + g * 4 + 1]) continue;
                s_issue += (double)(q[g * 4 + 1] - q[g * 4]); s_land += (double)(q[g * 4 + 2] - q[g * 4]);
                s_comp += (double)(q[24 + g * 4 + 1] - q[24 + g * 4]); s_item += (double)(q[g * 4 + 3] - q[g * 4]);
                s_lbar += (double)(q[g * 4 + 3] - q[g * 4 + 2]); s_cbar += (double)(q[24 + g * 4 + 3] - q[24 + g * 4 + 1]);
                ++n;
            }
        }
        if (n) printf("dbg (%d samples, cycles of the 100 MHz?? counter -- see ratio): dma issue %.0f, dma landed %.0f, consumer wave compute %.0f, item %.0f, loader waits at barrier %.0f, consumer waits at barrier %.0f\n",
                      n, s_issue / n, s_land / n, s_comp / n, s_item / n, s_lbar / n, s_cbar / n);
    }
    if (dump) {
        // full kernel once more on clean outputs, dump frames 0 and F-1
        for (int i = 0; i < F * N; ++i) CK(hipMemset(L.dst[i], 0xEE, vbytes));
        launch2(0);
        CK(hipDeviceSynchronize());
        FILE* fo = fopen(dump, "wb");
        std::vector<uint8_t> hb(vbytes);
        for (int f : {0, F - 1})
            for (int v = 0; v < N; ++v) { CK(hipMemcpy(hb.data(), L.dst[f * N + v], vbytes, hipMemcpyDeviceToHost)); fwrite(hb.data(), 1, vbytes, fo); }
        fclose(fo);
        printf("dumped %s\n", dump);
    }
    (void)frame_byte;
    return 0;
}
