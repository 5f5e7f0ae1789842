// gs360_kernels.h -- device-side parameter blocks shared by the kernels and the C-ABI glue.
// gfx950 only; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "../../include/gs360.h"

namespace gs360 {

// Output tile of one 256-thread workgroup: 64 px wide (one wavefront = 64 consecutive pixels of a row),
// 16 rows tall (4 wavefronts x 4 rows each).
constexpr int kTileW = 64;
constexpr int kRowsPerWave = 4;
constexpr int kWaves = 4;                        // wavefronts per workgroup
constexpr int kTileH = kWaves * kRowsPerWave;
constexpr int kHalfRows = kRowsPerWave / 2;   // level views: half of a wavefront's rows are horizon mirrors

// ------------------------------------------------------------------------------------------------
// EQ-SPEC v1 per-view constants.  Host computes them in float64 and rounds once.
// ------------------------------------------------------------------------------------------------
struct EqView {
    float sxu, syv;      // tan(hfov/2)/out_w, tan(vfov/2)/out_h
    float sp, cp;        // sin / cos of pitch
    float x0f32;         // 32 * frac((yaw/360 + 1/2) * W - 1/2)
    int32_t out_w, out_h;
    int32_t tiles_x, tiles_y;
    int32_t tile_base;   // first tile index of this view's RING inside one frame
    int32_t level;       // pitch == 0 exactly (sp == 0, cp == 1): horizon-symmetric fast path
    int32_t fish;        // equidistant-fisheye output (GS360_EQ_FISHEYE_OUT): sxu/syv = fov/180/size, general row path
    int32_t blocked;     // RGB bilinear only: 1 = 4-row x 16-column gather patches instead of 64-pixel rows (strong minification);
                         // 2 = the view is rendered by eq_staged_kernel (LDS-staged source texels; all views of the launch then are)
    int32_t pad_;
    // the two per-member scalars of a yaw ring, adjacent and 8-byte aligned: the kernel reads them with one scalar load
    int32_t x0i32;       // 32 * (floor((yaw/360 + 1/2) * W - 1/2) mod W)
    int32_t flip;        // ring member whose pitch is MINUS the ring's first view's pitch: rows run bottom-up, latitude negated
};
static_assert(sizeof(EqView) == 64 && offsetof(EqView, x0i32) % 8 == 0, "EqView layout");

struct EqLaunch {
    const uint8_t* src[GS360_MAX_FRAMES];
    const uint8_t* mask[GS360_MAX_FRAMES];   // optional keep-BIT images ((H + 1) rows of mask_stride bytes, see mask_pack_kernel), all null when unused
    uint8_t* dst[GS360_MAX_FRAMES * GS360_MAX_VIEWS];
    EqView view[GS360_MAX_VIEWS];
    float kx32, ky32;    // 32*W/(2*pi), 32*H/pi
    int32_t W, H;
    int32_t y0i32;       // 16*H - 16
    int32_t n_views, n_frames;
    int32_t tiles_per_frame, total_tiles, chunk;  // chunk = ceil(total_tiles / 8) (XCD swizzle)
    const int16_t* cubic_tab;   // 32*32*16 int16 (device) when interp == cubic
    int64_t src_stride;
    int64_t mask_stride;
    int64_t dst_stride;  // 0 = tight (out_w * C)
    // Yaw rings: views [ring_first[g], ring_first[g] + ring_count[g]) share every EQ-SPEC constant except the integer
    // longitude offset x0i32 (and the sign of the pitch: `flip`), so one workgroup evaluates the coordinates of a tile once
    // and samples it for every member.  Tiles are numbered per RING (view[ring_first[g]].tile_base); the 16-bit kernel
    // runs full rings as well.
    int32_t n_rings;
    int32_t xcd_group_log2;   // >= 0: XCD x takes runs of 2^g consecutive tiles round-robin (rings of unequal size in one launch:
                              // contiguous chunks would hand whole rings to single XCDs); -1: contiguous chunks (`chunk`)
    int32_t ring_first[GS360_MAX_VIEWS];   // 32-bit on purpose: sub-dword fields of the kernel argument are fetched with VECTOR loads
    int32_t ring_count[GS360_MAX_VIEWS];   // (a memory round trip per wavefront), dwords with scalar loads
    // The cubic variants run persistent workgroups (one LDS weight table per workgroup): the caller sets `persist_blocks` (grid
    // cap, 0 = one tile per workgroup); the launcher fills `grid_total` (positions of the tile order to walk).
    int32_t persist_blocks, grid_total;
};

// ------------------------------------------------------------------------------------------------
// table-mode remap (cv2.remap semantics)
// ------------------------------------------------------------------------------------------------
struct TableLaunch {
    const uint8_t* src;
    const float* map_x;
    const float* map_y;
    const uint8_t* valid;  // may be null
    // a map plan instead of map_x / map_y / valid (map_pack_kernel; null = float maps): 5 bytes per pixel
    const uint32_t* packed;     // x + 8 (12 bits) | y + 8 (12 bits) | x phase (5 bits) | low 3 bits of the y phase
    const uint8_t* packed_hi;   // high 2 bits of the y phase | valid << 2
    int32_t use_valid;          // with a plan: apply its valid bit (the `valid` pointer of the float form)
    int32_t flat;               // row slots take spans of the flat output instead of row segments (table_remap_tile)
    uint8_t* dst;
    int32_t H, W, h, w;
    int64_t src_stride, dst_stride;
    int32_t interp;
    int32_t fill;
    uint8_t cval[4];
    const int16_t* cubic_tab;
    int32_t pipelined;   // W >= 8 and 32-bit tap offsets: split fetch/blend path allowed
    int32_t tiles_x;     // filled by launch_table_batch
    int32_t tile_base;   // first tile of this job inside the batched launch
    // Lanczos-4, RGB: the 1-D phase table (32 x 8 float32, the floats the 2-D table was built from) and, per 2-D phase, the weight
    // pairs (taps 4, 5) of window rows 4 and 5 of the 2-D table as two dwords -- the block its sum fix-up patches.  With both
    // present the kernel rebuilds the other weights per pixel (same float32 product, same rounding) instead of reading 128 B of
    // the 128 KiB table; null = read the table.  (gs360_ctx_create checks the rebuilt weights against the table, all phases.)
    const float* lz_c1;
    const uint32_t* lz_cen;
};

// One launch for several remaps (e.g. the views of a dual-fisheye pair): no per-view launch tails.
struct TableBatch {
    TableLaunch job[GS360_MAX_VIEWS];
    int32_t n_jobs, total_tiles, chunk;
    int32_t persist_blocks;   // grid cap for kernels whose workgroups walk several tiles (0: one tile per workgroup); set by the caller
};

// ------------------------------------------------------------------------------------------------
// FE-SPEC v1 (fused dual-fisheye -> perspective)
// ------------------------------------------------------------------------------------------------
struct FeView {
    const uint8_t* src;
    uint8_t* dst;
    uint8_t* valid_out;  // may be null
    float sxu, syv, sp, cp, sy, cy;
    float k1, k2, k3, k4, p1, p2, tp1, tp2, b1, b2, f, cx0, cy0, wmax, hmax, cos_tmax;
    int32_t tang;
    int32_t W, H;        // sensor size
    int32_t out_w, out_h;
    int32_t tiles_x, tiles_y, tile_base;
};

struct FeCommon {          // per-launch constants of fe_views_kernel
    int32_t total_tiles, chunk, n_views;
    int32_t interp, mask_outside, mask_value;
    int64_t src_stride, dst_stride;
    uint8_t cval[4];
    const int16_t* cubic_tab;
    int32_t pipelined;
    int32_t grid_total;      // positions of the tile order (the persistent bicubic RGB variant walks them with stride gridDim.x)
};

struct FeBatch {           // kernel argument: all views of one launch + the common block
    FeView view[GS360_MAX_VIEWS];
    FeCommon common;
};

struct FeLaunch {          // host-side batch description
    FeView view[GS360_MAX_VIEWS];
    int32_t n_views, total_tiles, chunk;
    int32_t interp, mask_outside, mask_value;
    int64_t src_stride, dst_stride;
    uint8_t cval[4];
    const int16_t* cubic_tab;
    int32_t pipelined;
    int32_t persist_blocks;  // grid cap of the persistent bicubic RGB variant (0: one tile per workgroup)
};

// ------------------------------------------------------------------------------------------------
// input colour stage (gs360_color.hip): DF:684-725 for 8-bit images
// ------------------------------------------------------------------------------------------------
struct ColorLaunch {
    const uint8_t* src;
    uint8_t* dst;
    const void* rtab;       // device float3[n*n*256]: LUT pre-interpolated along red for every red level
    const float* tables;    // device: level positions R,G,B x 256, 256 output thresholds, packed bin levels
    const void* cube;       // device uint32[2^24]: the stage evaluated for every 8-bit pixel (NULL: evaluate per pixel from rtab / tables)
    int32_t H, W, lut_size, red_index;
    int32_t fixups;         // 1 or 2 in-bin threshold compares, 0 = binary search (color_build_bins)
    int64_t src_stride, dst_stride;
};
struct Color16Launch {     // 16-bit images: everything per pixel, output thresholds in monotone pieces (gs360_color.hip)
    const void* src;
    void* dst;
    const float* lut;       // device [b][g][r][3]
    const float* thr;       // device, concatenated pieces
    const void* bins;       // device, color16_bins_bytes() (color16_build_bins); unused when n_pieces == 0
    float dmin[3], span[3];
    float start[4];
    int32_t base[4], off[5];
    int32_t n_pieces, H, W, lut_size, red_index;
    int64_t src_stride, dst_stride;
};
hipError_t launch_color16(const Color16Launch& L, int C, hipStream_t s);
size_t color16_bins_bytes();
void color16_build_bins(int n_pieces, const float* start, const int32_t* off, const float* thr, void* bins /* host, color16_bins_bytes() */);
size_t color_rtab_bytes(int lut_size);
size_t color_tables_floats();
int color_build_bins(const float* thresholds, uint8_t* bins);
hipError_t build_color_rtab(const float* d_lut, const float* d_pos_r, void* d_rtab, int lut_size, hipStream_t s);
size_t color_cube_bytes();
hipError_t build_color_cube(const ColorLaunch& L /* rtab, tables, lut_size, fixups */, void* d_cube, hipStream_t s);
hipError_t launch_color(const ColorLaunch& L, int C, hipStream_t s);

// kernel launchers (gs360_kernels.hip)
// keep-mask threshold + pack (gs360_equirect_views_masked_u8): byte masks -> bit images, one launch for all frames of a call
struct MaskPack {
    const uint8_t* src[GS360_MAX_FRAMES];
    uint32_t* dst[GS360_MAX_FRAMES];
    int32_t W, H, pitch_dw, n;   // pitch_dw = (W + 1 + 31) / 32 dwords per row; H + 1 rows are written
    int64_t stride;              // bytes per row of the byte masks
};
hipError_t launch_mask_pack(const MaskPack& P, hipStream_t s);
hipError_t launch_equirect(const EqLaunch& L, int C, hipStream_t s);
hipError_t launch_equirect_staged(const EqLaunch& L, hipStream_t s);   // bilinear RGB u8, every view with blocked == 2 (LDS-staged 16x16 wavefront tiles)
hipError_t launch_equirect_cubic(const EqLaunch& L, int C, hipStream_t s);
// source-major kernel (gs360_srcmajor.hip): one launch = one level yaw ring that fills its circle
struct SmTile { int32_t x0, y0, nrows, wch, eoff, nq, pad0, pad1; };   // box of a plan tile: first byte (in the period) / row, rows, 16-byte chunks per row; entries
struct SmPlan;
void sm_plan_free(SmPlan* p);
struct SmShape {                         // how a call's views fall into yaw rings of one size (filled by sm_eligible)
    int N, n_rings;                      // members per ring, rings
    int ref[GS360_MAX_VIEWS];            // ring -> view index of the member at the ring's origin
    int partner[GS360_MAX_VIEWS];        // ring -> the ring at minus its pitch (itself for a level ring)
    int qmap[GS360_MAX_VIEWS];           // ring * N + position -> view index
};
// the context's source-major plans: most recent geometries (a geometry may hold two: full- and half-height tiles); plans that were
// evicted wait in the graveyard for a moment at which the caller waits for the device anyway (hipFree synchronises it)
struct SmScratch {                       // the plan builder's device and host blocks, kept between builds (grow only)
    std::mutex mu;
    void *dev = nullptr, *host = nullptr;
    size_t dev_cap = 0, host_cap = 0;
    SmScratch() = default;
    SmScratch& operator=(const SmScratch& o) { dev = o.dev; host = o.host; dev_cap = o.dev_cap; host_cap = o.host_cap; return *this; }
};
struct SmCache {
    std::mutex mu;
    SmScratch scratch;
    std::vector<SmPlan*> plans, graveyard;
    size_t cap = 16;
    uint64_t builds = 0;                 // plans built by this context (hits build nothing: tests count instead of timing)
    uint64_t build_us = 0;               // ... and the wall time those builds took in all (coordinate kernels, copy back, ordering, upload)
    uint64_t inline_frees = 0;           // plans released inside a call because nobody synchronised for 64 evictions
};
void sm_cache_drain(SmCache& cache);     // releases the graveyard (gs360_sync, gs360_ctx_destroy)
void sm_cache_destroy(SmCache& cache);
bool sm_eligible(const EqLaunch& L, int C, int esize, int interp, bool masked, SmShape* S);
int sm_prepare(const EqLaunch& L, const SmShape& S, SmCache& cache, bool masked, int Bx, int R, int G_opt, bool adapt, int max_box_pct, size_t lds_limit, int n_cu,
               hipStream_t s, hipError_t* herr, SmPlan** out, int* box_pct);
int sm_launch(const EqLaunch& L, const SmShape& S, const SmPlan* plan, int G_opt, bool stage_regs, size_t lds_limit, int n_cu, hipStream_t s, hipError_t* herr,
              int* info /* [4]: box overhead %, tile rows, images per workgroup, 1 = the register-staging kernel */);
void sm_release(SmCache& cache, SmPlan* plan);
void build_cubic_table(int16_t* out);      // host: OpenCV initInterTab2D(INTER_CUBIC, fixpt) restated, 32*32*16
void build_lanczos4_table(int16_t* out);   // host: initInterTab2D(INTER_LANCZOS4, fixpt) restated, 32*32*64
void build_coef1d(float* out);             // host: the float32 1-D phase tables (linear, cubic, lanczos4) of the CV_16U samplers, 448
hipError_t launch_equirect_u16(const EqLaunch& L, int C, bool cubic, hipStream_t s);                       // gs360_u16.hip
hipError_t launch_bswap16(uint16_t* buf, size_t n, hipStream_t s);            // gs360_u16.hip: in-place byte swap of 16-bit samples
hipError_t launch_arith_selftest(uint32_t seed, int blocks, int iters, unsigned long long* d_bad, hipStream_t s);
hipError_t launch_table_u16_batch(TableBatch& B, int C, const float* coef, const uint16_t cval[4], hipStream_t s);   // all jobs share interp
hipError_t launch_table(const TableLaunch& L, int C, hipStream_t s);
// map plan: float maps (+ valid) -> the packed form; `nearest` packs cvRound(map) instead of the 1/32-pixel fixed point
constexpr int kMapPlanMaxDim = 4079;     // x + 8, y + 8 of any position that still touches the image fit 12 bits
// a packed position back as the floats k / 32 (or the integer position for nearest): what the samplers' own cvRound(. * 32) maps to k again
__device__ __forceinline__ void planned_coords(const uint32_t P, const uint32_t hb, const bool nearest, float& mx, float& my) {
    const int ix = (int)(P & 0xfffu) - 8, iy = (int)((P >> 12) & 0xfffu) - 8;
    if (nearest) {
        mx = (float)ix;
        my = (float)iy;
        return;
    }
    mx = (float)(ix * 32 + (int)((P >> 24) & 31u)) * 0.03125f;
    my = (float)(iy * 32 + (int)((P >> 29) | ((hb & 3u) << 3))) * 0.03125f;
}

hipError_t launch_map_pack(const float* map_x, const float* map_y, const uint8_t* valid, int64_t n, int nearest,
                           uint32_t* packed, uint8_t* packed_hi, hipStream_t s);
// LDS-staged table kernel (gs360_tablestage.hip): bilinear RGB u8 through a map plan's STAGE PLAN -- per source size the output cut into
// tiles of 64 pixels x R rows, per tile the box of source texels its taps touch, per pixel one dword (LDS offset | phases), tile-major
struct TsTile { int32_t x0b, y0, nrows, wch, magic, chunks, ty, tx; };       // box of a tile: first byte of a source row, first row, rows, 16-byte chunks
                                                                              // per row; ceil(2^20 / wch); chunks of the box; tile row / column in the output
struct TsPlan {
    int W = 0, H = 0, R = 0, h = 0, w = 0, use_valid = 0;            // key: source size, tile rows, whether the valid bit is applied
    uint8_t* d_recs = nullptr;                                        // per tile: 64-byte head (TsTile) + R x 64 plan words
    int n_tiles = 0, tiles_x = 0;
    int max_box = 0;                                                  // bytes of the largest box
    int slow_tiles = 0;                                               // tiles whose box exceeded the budget (every pixel redone from memory)
};
void ts_plan_free(TsPlan* p);
TsPlan* ts_build_plan(const uint32_t* d_packed, const uint8_t* d_hi, int h, int w, int W, int H, int R, int use_valid, int box_budget, hipStream_t s,
                      hipError_t* herr);
struct TsJobDesc {
    const uint8_t* src;
    uint8_t* dst;
    const uint32_t* packed;
    const uint8_t* packed_hi;
    const TsPlan* plan;
    int64_t src_stride, dst_stride;
    int32_t fill;
};
struct TsLaunch {
    TsJobDesc job[GS360_MAX_VIEWS];
    int32_t n_jobs, R;
    int32_t wg_per_cu;                   // 0 auto (probes)
    uint8_t cval[4];
};
hipError_t ts_launch(const TsLaunch& L, int n_cu, size_t lds_per_cu, hipStream_t s);
hipError_t launch_table_batch(TableBatch& B, int C, hipStream_t s);   // all jobs share C and interp (job[0].interp)
hipError_t launch_fisheye(const FeLaunch& L, int C, hipStream_t s);

}  // namespace gs360
