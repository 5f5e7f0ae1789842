/*
 * gs360.h -- C ABI of libgs360hip.so, the MI355X (gfx950) reprojection engine.
 *
 * This is the drop-in boundary for the 360PerspCut hot path.  The reference is pure Python and has
 * no FFI of its own; each entry point below replaces the native work the reference delegates to a
 * third-party binary/wheel at the cited call site (paths are into the reference repository):
 *
 *   gs360_equirect_views_u8      one `ffmpeg -vf v360=input=equirect:output=rectilinear:...` process per
 *                                (source, view): cli_tools/gs360_360PerspCut.py:310-314 (filter string),
 *                                :569-590 (run_one -> Popen), :830-836 (one job per view).
 *   gs360_remap_table_u8         cv2.remap(src, map_x, map_y, interp, borderMode=BORDER_CONSTANT,
 *                                borderValue=float(mask_value)) followed by `out[~valid] = mask_value`:
 *                                cli_tools/gs360_DualFisheyeDistortionCalibration.py:2001-2014 (image),
 *                                :2031-2043 (mask, INTER_NEAREST), :1198-1212 (undistort).
 *   gs360_fisheye_views_u8       the same call sites with the map of DF:1759-1823
 *                                (build_direct_perspective_map_for_lens) evaluated in-kernel instead of
 *                                being read from a table (FE-SPEC v1).
 *
 * Conventions: extern "C", plain pointers and sizes, POD structs, no exceptions across the boundary.
 * Every function returns 0 on success or a negative gs360_status; gs360_last_error() returns the
 * calling thread's last message.  The caller owns every buffer it passes in; the library never frees
 * caller memory.  One ctx per device; calls on different ctx are fully concurrent; calls on one ctx
 * are ordered per `slot` (each slot is a HIP stream) and are ASYNCHRONOUS unless stated otherwise --
 * gs360_sync(ctx, slot) waits.  Images are interleaved HWC uint8, C in {1,3,4}.
 */
#ifndef GS360_H
#define GS360_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): + gs360_ctx_set_option / _get_option / gs360_device_pci_bus_id (added in round 5 without a bump), the table_stage* options; the
 * library no longer reads GS360_RING, GS360_XCD_GROUP, GS360_EQ_PERSIST, GS360_TABLE_PERSIST, GS360_LANCZOS_TABLE, GS360_TABLE_ROWS from
 * the environment (context options of the same names do that: INTEGRATION.md section 2.1).  A binding checks gs360_abi_version() first. */
#define GS360_ABI_VERSION 2

typedef enum gs360_status {
    GS360_OK = 0,
    GS360_ERR_ARG = -1,         /* bad argument (NULL, size, channels, slot ...) */
    GS360_ERR_HIP = -2,         /* a HIP runtime call failed; text in gs360_last_error */
    GS360_ERR_NODEV = -3,       /* no usable GPU */
    GS360_ERR_UNSUPPORTED = -4, /* valid request this build does not implement */
    GS360_ERR_NOMEM = -5
} gs360_status;

/* values equal cv2.INTER_NEAREST / INTER_LINEAR / INTER_CUBIC / INTER_LANCZOS4 (DF:59-64) */
#define GS360_INTERP_NEAREST 0
#define GS360_INTERP_LINEAR 1
#define GS360_INTERP_CUBIC 2
#define GS360_INTERP_LANCZOS4 4 /* table remap and fused fisheye only (8x8 taps) */

/* flags of gs360_equirect_views_u8: the views are EQUIDISTANT-FISHEYE outputs instead of rectilinear ones -- the
 * `fisheyeXY` preset's `v360=...:output=fisheye:d_fov=...` jobs (cli_tools/gs360_360PerspCut.py:351-414).  hfov_deg /
 * vfov_deg of each view are then the full horizontal / vertical field of view of the fisheye image (image-plane radius
 * <-> off-axis angle, 90 degrees at radius 1; for v360's d_fov: hfov = d_fov * w / hypot(w, h)). */
#define GS360_EQ_FISHEYE_OUT 0x1u

/* limits of one batched launch (larger requests are split internally) */
#define GS360_MAX_VIEWS 16
#define GS360_MAX_FRAMES 16

typedef struct gs360_ctx gs360_ctx;

/* One output view: the numeric fields of ViewSpec (cli_tools/gs360_360PerspCut.py:32-45). */
typedef struct gs360_view {
    double yaw_deg;   /* + = look right */
    double pitch_deg; /* + = look up */
    double hfov_deg;
    double vfov_deg;
    int32_t width;
    int32_t height;
} gs360_view;

/* SensorCalibration (cli_tools/gs360_DualFisheyeDistortionCalibration.py:67-85), numeric fields. */
typedef struct gs360_calib {
    int32_t width;
    int32_t height;
    double f, cx, cy, k1, k2, k3, k4, p1, p2, b1, b2;
} gs360_calib;

/* ---- library / device ---------------------------------------------------------------------- */
int gs360_abi_version(void);
int gs360_device_count(void);
/* copies the calling thread's last error text (NUL terminated); returns its length */
int gs360_last_error(char *buf, size_t buf_len);
/* n_slots HIP streams are created (1..16) */
int gs360_ctx_create(int device, int n_slots, gs360_ctx **out);
int gs360_ctx_destroy(gs360_ctx *ctx);
int gs360_device_info(gs360_ctx *ctx, char *name, size_t name_len, int32_t *cu_count, uint64_t *hbm_bytes);
/* "domain:bus:device.function" of the context's GPU (>= 16 bytes): lets a multi-process job prove that its ranks sit on DISTINCT
 * devices (bench.py gathers it from every rank; the reference fans jobs out over one machine's workers, PC:830-836) */
int gs360_device_pci_bus_id(gs360_ctx *ctx, char *buf, size_t buf_len);
/* Context options: kernel-selection switches for tests, probes and A/B runs (the defaults are the measured optima; results never depend
 * on them).  The reference has no counterpart (it shells out to ffmpeg / calls cv2.remap: PC:310-314, DF:2001); a binding needs them
 * only to pin a kernel variant.  gs360_ctx_create seeds the documented user switches ONCE from the environment (GS360_STAGE,
 * GS360_LANEMAP, GS360_SRCMAJOR, GS360_COLOR_CUBE, GS360_TABLE_STAGE); after that the library never reads the environment -- set options here instead.
 *   "lanemap"        -1 auto | 0 rows | 1 blocked       gather kernels' lane map
 *   "stage"          -1 auto | 0 never | 1 always       LDS-staged equirect kernel
 *   "srcmajor"       -1 auto | 0 never | 1 always       source-major equirect kernel (yaw rings of one size, level or in +/- pitch pairs:
 *                                                       the default / full360coverage / fisheyelike presets); "srcmajor_bx" (bytes per tile row,
 *                                                       multiple of 16), "srcmajor_rows" (source rows per tile), "srcmajor_images"
 *                                                       (0 auto | 1..12 dividing twice the ring size: images of a tile one workgroup walks),
 *                                                       "srcmajor_adapt" (1 | 0: calls too small to fill the GPU take tiles of half the height)
 *   "ring"           0 auto | n                         at most n views share one coordinate evaluation
 *   "xcd_group"      -2 auto | -1 chunks | g            tile order across the XCDs
 *   "eq_persist", "table_persist"                       grid caps of the persistent kernels (table_persist: -1 auto)
 *   "lanczos_table", "table_rows"                       0 | 1: A/B references of the Lanczos-4 weight rebuild and the flat spans
 *   "color_cube"     -1 / 1 tabulate | 0 per pixel      8-bit colour stage (read by gs360_color_plan_create)
 *   "table_stage"    -1 auto | 0 never | 1 always       LDS-staged table kernel (bilinear RGB through map plans; auto: plans whose tiles have boxes);
 *                                                       "table_stage_rows" (8 | 16 | 32: rows of its 64-pixel tiles), "table_stage_wgs" (0 auto | 1..4
 *                                                       workgroups per CU)
 * Read-only (get): "last_eq_kernel" -- which kernel the last equirect call launched: 0 gather, 1 LDS-staged, 2 source-major, -1 none yet;
 * "last_srcmajor_box_pct" -- tile-box bytes of the last source-major plan in percent of the grid cells they stand for;
 * "last_srcmajor_rows", "last_srcmajor_images" -- tile rows and images per workgroup of the last source-major launch;
 * "srcmajor_plan_builds" -- source-major plans this context has built so far (a call on a cached geometry builds none), "srcmajor_plan_build_us"
 * -- the wall time those builds took in all, "srcmajor_plans" -- plans
 * it holds, "srcmajor_inline_frees" -- evicted plans it had to release inside a call (normally they wait for gs360_sync(ctx, -1) /
 * gs360_ctx_destroy: hipFree synchronises the device);
 * "last_table_kernel" -- jobs of the last 8-bit table call that took the LDS-staged kernel, "last_table_stage_slow_tiles" -- tiles of their
 * stage plans whose box exceeded the LDS budget.
 * Unknown keys and out-of-range values are GS360_ERR_ARG.  Thread-safe; a change applies to calls that start after it. */
int gs360_ctx_set_option(gs360_ctx *ctx, const char *key, int value);
int gs360_ctx_get_option(gs360_ctx *ctx, const char *key, int *value);

/* ---- memory (device buffers carry 64 B of readable slack after the requested size) ---------
 * Images handed to the hot-path entry points must come with that slack (the aligned 12- / 16-byte tap reads of the last
 * pixels of the last row run past the image) and, for 3-channel images, a 4-byte aligned base pointer; equirect sources
 * are at least 8 texels wide and less than 2^18 rows tall (the kernels form latitudes with 24-bit multiply-adds), H * stride < 2^32
 * and stride < 2^24 (32-bit tap offsets).  gs360_dev_alloc satisfies
 * the first two; the size limits are checked and reported as GS360_ERR_ARG / GS360_ERR_UNSUPPORTED. */
int gs360_dev_alloc(gs360_ctx *ctx, size_t bytes, void **dptr);
int gs360_dev_free(gs360_ctx *ctx, void *dptr);
int gs360_host_alloc(gs360_ctx *ctx, size_t bytes, void **hptr); /* pinned */
int gs360_host_free(gs360_ctx *ctx, void *hptr);
int gs360_upload(gs360_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, int slot);
int gs360_download(gs360_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, int slot);
int gs360_dev_memset(gs360_ctx *ctx, void *dst_dev, int value, size_t bytes, int slot);
/* in-place byte swap of n_samples 16-bit samples on the device (16-bit PPM frames of the video decode pipe are big-endian:
 * cli_tools/gs360_360PerspCut.py:343-347 keeps > 8-bit videos at 16 bits); asynchronous on `slot` */
int gs360_dev_bswap16(gs360_ctx *ctx, void *buf_dev, size_t n_samples, int slot);
int gs360_sync(gs360_ctx *ctx, int slot); /* slot < 0: every slot */

/* ---- timing: HIP events recorded on the slot's own stream (8 events per slot) -------------- */
int gs360_event_record(gs360_ctx *ctx, int slot, int event_idx);
int gs360_event_elapsed_ms(gs360_ctx *ctx, int slot, int event_from, int event_to, float *ms); /* syncs on event_to */
int gs360_event_sync(gs360_ctx *ctx, int slot, int event_idx);                                   /* host waits for the event */
/* work queued on waiting_slot after this call starts only when event (event_slot, event_idx) has completed: lets uploads,
 * kernels and downloads of one frame sit on DIFFERENT streams (one upload stream + one download stream is what makes
 * PCIe run full duplex: 97 GB/s against 57 GB/s with both directions on one stream, measured) */
int gs360_stream_wait_event(gs360_ctx *ctx, int waiting_slot, int event_slot, int event_idx);

/* ---- hot path: device-resident buffers, asynchronous --------------------------------------- */

/*
 * Equirectangular -> rectilinear views.  For every frame f < n_frames and view k < n_views writes
 * dst[f * n_views + k] (views[k].height x views[k].width x C, row stride dst_stride bytes, 0 = tight).
 * src_frames[f]: H x W x C equirect image, row stride src_stride bytes (0 = tight).
 * Geometry: EQ-SPEC v1: pinhole ray -> pitch about X -> yaw about Y -> lon/lat ->
 * 1/32-px fixed-point bilinear; horizontal border wraps, vertical border clamps.
 * interp: GS360_INTERP_LINEAR or GS360_INTERP_CUBIC (4x4 taps, OpenCV's fixed-point Keys A=-0.75 table; the
 * reference's own default is v360 interp=cubic, PC:730).  flags: 0 or GS360_EQ_FISHEYE_OUT.
 */
int gs360_equirect_views_u8(gs360_ctx *ctx, const void *const *src_frames, int n_frames,
                            int W, int H, int C, size_t src_stride,
                            const gs360_view *views, int n_views,
                            void *const *dst, size_t dst_stride,
                            int interp, uint32_t flags, int slot);

/*
 * Same, with a per-frame keep-mask fused into the store (BASELINE config 5; no reference counterpart): mask_frames[f]
 * is an H x W single-channel uint8 image in the SegmentationMaskTool convention (0 = masked target, 255 = keep,
 * reference cli_tools/gs360_SegmentationMaskTool.py:765-774).  The mask is sampled NEAREST at the same source
 * coordinate (texel ((sx+16)>>5 mod W, clamp((sy+16)>>5)) of EQ-SPEC v1) and the output pixel is written as 0 on all
 * channels where the mask value is < 128.  mask_stride in bytes (0 = W).
 * Only that comparison is ever used, so every call first thresholds its masks into bit images (context scratch of the slot, one
 * streaming pass over the mask bytes on the slot's stream, 6 us per 8K mask) and the kernels sample those; the masks themselves are
 * only read, and may be reused or released as soon as the call's stream work is complete.
 */
int gs360_equirect_views_masked_u8(gs360_ctx *ctx, const void *const *src_frames, const void *const *mask_frames,
                                   int n_frames, int W, int H, int C, size_t src_stride, size_t mask_stride,
                                   const gs360_view *views, int n_views,
                                   void *const *dst, size_t dst_stride,
                                   int interp, uint32_t flags, int slot);

/*
 * cv2.remap with float32 maps, BORDER_CONSTANT; then, if valid != NULL, dst[~valid] = fill_value
 * on all channels.  src: H x W x C; map_x/map_y/valid: h x w (tight); dst: h x w x C.
 * border_value: 4 doubles (cv::Scalar; Python's borderValue=float(v) is {v,0,0,0}).
 * interp: GS360_INTERP_NEAREST / LINEAR / CUBIC / LANCZOS4 (cv2's fixed-point tables for 8-bit images).
 * All pointers are device pointers.  H, W < 32767 (cv2.remap's own limit).
 */
int gs360_remap_table_u8(gs360_ctx *ctx, const void *src, int H, int W, int C, size_t src_stride,
                         const float *map_x, const float *map_y, const uint8_t *valid, int h, int w,
                         int interp, const double *border_value, int fill_value,
                         void *dst, size_t dst_stride, int slot);

/*
 * Several remaps in ONE launch (the 10 views of a dual-fisheye pair, DF:1993-2043: one cv2.remap per view): same
 * semantics per job as gs360_remap_table_u8, common channel count / interpolation / border value.  A batched launch
 * has no per-view launch tails (cfg4: -16 %).  n_jobs may exceed GS360_MAX_VIEWS (split internally).
 */
typedef struct gs360_remap_job {
    const void *src;        /* H x W x C */
    int32_t H, W;
    size_t src_stride;      /* 0 = tight */
    const float *map_x;     /* h x w */
    const float *map_y;
    const uint8_t *valid;   /* h x w or NULL */
    int32_t h, w;
    int32_t fill_value;     /* written where valid == 0 */
    void *dst;              /* h x w x C */
    size_t dst_stride;      /* 0 = tight */
} gs360_remap_job;
int gs360_remap_tables_u8(gs360_ctx *ctx, const gs360_remap_job *jobs, int n_jobs, int C, int interp,
                          const double *border_value, int slot);

/*
 * Map plans.  cv2.remap converts its float maps to 1/32-pixel fixed point on every call before it samples; the dual-fisheye
 * tool applies the SAME maps to every image pair of a run (DF:2582-2592).  A plan does the conversion once and keeps the result
 * in 5 bytes per output pixel (position, phases, valid bit) instead of two floats and a valid byte: less to read per call, no
 * conversion per pixel, identical results (integer positions are clamped to [-8, 4087], which leaves every position whose
 * Lanczos-4 window still touches a source of up to 4079 x 4079 pixels untouched; larger sources are refused with
 * GS360_ERR_UNSUPPORTED -- use the float maps).  map_x / map_y / valid are DEVICE pointers (h x w, valid may be NULL) and may be
 * released when the call returns.  `nearest` != 0 packs cvRound(map) for GS360_INTERP_NEAREST (mask cutting, DF:2031-2043);
 * a plan serves either nearest or the interpolating samplers (linear, cubic, Lanczos-4), 8- and 16-bit sources alike.
 * gs360_remap_plans_u8 = gs360_remap_tables_u8 with plans[j] in place of jobs[j].map_x / map_y (ignored); jobs[j].valid != NULL
 * asks for the plan's valid bit (fill_value where it is 0), the pointer itself is not read.
 */
typedef struct gs360_map_plan gs360_map_plan;
int gs360_map_plan_create(gs360_ctx *ctx, const float *map_x, const float *map_y, const uint8_t *valid, int h, int w,
                          int nearest, int slot, gs360_map_plan **out);
int gs360_map_plan_destroy(gs360_ctx *ctx, gs360_map_plan *plan);
int gs360_remap_plans_u8(gs360_ctx *ctx, const gs360_remap_job *jobs, const gs360_map_plan *const *plans, int n_jobs, int C,
                         int interp, const double *border_value, int slot);
/* the same on CV_16U sources (OpenCV converts the maps the same way for every depth): gs360_remap_tables_u16 with plans */
int gs360_remap_plans_u16(gs360_ctx *ctx, const gs360_remap_job *jobs, const gs360_map_plan *const *plans, int n_jobs, int C,
                          int interp, const double *border_value, int slot);

/*
 * Dual-fisheye -> perspective views with the map evaluated in-kernel (FE-SPEC v1).  View k samples
 * src_lens[k] (H x W x C of calibs[k]) with views[k].yaw_deg measured RELATIVE to that lens
 * (DF:1883).  Pixels outside the lens model / sensor get mask_value on all channels when
 * mask_outside != 0 (DF:2009-2014); border taps use {mask_value,0,0,0} as cv2 does.
 * valid_out[k] (h x w uint8, may be NULL / may contain NULLs) receives the validity mask.
 */
int gs360_fisheye_views_u8(gs360_ctx *ctx, const void *const *src_lens, const gs360_calib *calibs,
                           int C, size_t src_stride,
                           const gs360_view *views, int n_views, double lens_fov_deg,
                           int interp, int mask_outside, int mask_value,
                           void *const *dst, size_t dst_stride, uint8_t *const *valid_out, int slot);

/* ---- input colour stage (dual-fisheye tool, before resampling) -------------------------------- */

/*
 * The `--input-lut` stage of cli_tools/gs360_DualFisheyeDistortionCalibration.py: apply_input_color_pipeline
 * (DF:684-725) = image_to_float01 -> apply_cube_lut_trilinear (DF:620-681) -> optional rec709_to_srgb
 * (DF:565-600) -> float01_to_image, for 8-bit images.  Because the input has 256 levels per channel and the
 * output 256, the two scalar ends of that pipeline are passed in as tables the caller computes with the reference's
 * own float32 expressions (the Python host does this with NumPy, gs360/color.py), and the kernel evaluates the
 * trilinear interpolation between them in the reference's float32 operation order:
 *   level_pos[c*256 + v]  LUT-grid position `clip((v/255 - domain_min[c]) / span[c], 0, 1) * (size-1)` of input
 *                         level v on channel c (c = 0,1,2 = R,G,B)
 *   out_thresholds[k]     k = 1..255: the smallest float32 LUT output x for which the encoded 8-bit result is >= k
 *                         (entry 0 is ignored; entries are >= 0, non-decreasing; +inf = never reached).  The encode
 *                         step is monotone, so the output level is the number of thresholds <= clip(x, 0, 1).
 *   lut                   size^3 RGB float32 triples, red fastest ([b][g][r][3], the .cube order, DF:556-562)
 * All three are HOST pointers, copied at plan creation.
 *
 * A plan evaluates the stage once for every possible 8-bit pixel (2^24 values) and keeps the results in 64 MiB of device memory;
 * gs360_color_apply_u8 then reads one table entry per pixel (GS360_ERR_NOMEM when the table cannot be allocated).  With
 * GS360_COLOR_CUBE=0 in the environment at plan creation the plan keeps its interpolation tables instead (18 MB for a 33^3 LUT)
 * and every apply evaluates the stage per pixel -- same results, about half the speed on photographs.
 */
typedef struct gs360_color_plan gs360_color_plan;
int gs360_color_plan_create(gs360_ctx *ctx, const float *lut, int lut_size, const float *level_pos,
                            const float *out_thresholds, gs360_color_plan **out);
int gs360_color_plan_destroy(gs360_ctx *ctx, gs360_color_plan *plan);
/*
 * Applies the plan to an H x W x C uint8 image (C = 3 or 4; a 4th channel is copied), device pointers, dst may equal
 * src.  red_index = 0 for RGB(A) memory order, 2 for BGR(A) (cv2.imread order, DF:697-699).
 */
int gs360_color_apply_u8(gs360_ctx *ctx, const gs360_color_plan *plan, const void *src, int H, int W, int C,
                         size_t src_stride, int red_index, void *dst, size_t dst_stride, int slot);

/* ---- 16-bit images ---------------------------------------------------------------
 * Same calls on uint16 samples (interleaved H x W x C, strides in BYTES and even).  The reference keeps 16-bit inputs at
 * native depth: cv2.imread(IMREAD_UNCHANGED) in the dual-fisheye tool (DF:735) and 16-bit PNG/TIFF stills / > 8-bit
 * videos (rgb48le) in 360PerspCut (gs360_360PerspCut.py:327-347).
 *   gs360_equirect_views_u16  EQ-SPEC v1 coordinates; bilinear (sum S a b + 512) >> 10, bicubic with the fixed-point
 *                             Keys table, exact integer sum (the value a 64-bit accumulation gives), clamped to
 *                             [0, 65535].  No fused keep-mask.
 *   gs360_remap_table_u16     cv2.remap on CV_16U: OpenCV's float-weight samplers (weights cy[k1]*cx[k2] in float32,
 *                             float32 accumulation in OpenCV's expression order, cvRound + saturate), all four
 *                             interpolations; border_value saturates to [0, 65535]; fill_value is written as uint16.
 */
int gs360_equirect_views_u16(gs360_ctx *ctx, const void *const *src_frames, int n_frames,
                             int W, int H, int C, size_t src_stride,
                             const gs360_view *views, int n_views,
                             void *const *dst, size_t dst_stride,
                             int interp, uint32_t flags, int slot);
int gs360_remap_table_u16(gs360_ctx *ctx, const void *src, int H, int W, int C, size_t src_stride,
                          const float *map_x, const float *map_y, const uint8_t *valid, int h, int w,
                          int interp, const double *border_value, int fill_value,
                          void *dst, size_t dst_stride, int slot);
/* several CV_16U remaps in one launch (the views of a dual-fisheye pair), as gs360_remap_tables_u8 */
int gs360_remap_tables_u16(gs360_ctx *ctx, const gs360_remap_job *jobs, int n_jobs, int C, int interp,
                           const double *border_value, int slot);

/*
 * The same stage for 16-bit images (DF:603-618 treat uint16 like uint8 with 65535 levels).  Nothing is tabulated on the
 * input side (float01 conversion, domain mapping and all three interpolation stages run per pixel in the reference's
 * float32 order).  Output side: n_pieces = 0 means `passthrough` = rint(clip(x) * 65535) in-kernel; otherwise the encode
 * step (Rec.709 -> sRGB through NumPy's float32 power) arrives as thresholds in n_pieces <= 4 monotone pieces of clip(x):
 * piece q covers [piece_start[q], piece_start[q+1]) (piece_start[0] is taken as 0), its thresholds are
 * thresholds[piece_off[q] .. piece_off[q+1]) (sorted), and the level is piece_base[q] + the number of them <= clip(x).
 * All pointers are HOST pointers, copied at plan creation (which also derives a 1 MiB index of the thresholds: for each of
 * 65536 equal steps of clip(x) the count up to the step and the next three thresholds, so that a pixel's level costs one
 * read per channel; the count itself stays exact).
 */
typedef struct gs360_color_plan16 gs360_color_plan16;
int gs360_color_plan16_create(gs360_ctx *ctx, const float *lut, int lut_size, const float *domain_min, const float *domain_max,
                              int n_pieces, const float *piece_start, const int32_t *piece_base, const int32_t *piece_off,
                              const float *thresholds, gs360_color_plan16 **out);
int gs360_color_plan16_destroy(gs360_ctx *ctx, gs360_color_plan16 *plan);
int gs360_color_apply_u16(gs360_ctx *ctx, const gs360_color_plan16 *plan, const void *src, int H, int W, int C,
                          size_t src_stride, int red_index, void *dst, size_t dst_stride, int slot);

/* ---- host-buffer conveniences (synchronous: H2D -> kernel -> D2H on `slot`) ----------------- */
int gs360_equirect_views_u8_host(gs360_ctx *ctx, const uint8_t *src, int W, int H, int C, size_t src_stride,
                                 const gs360_view *views, int n_views,
                                 uint8_t *const *dst, size_t dst_stride, int interp, uint32_t flags, int slot);
int gs360_remap_table_u8_host(gs360_ctx *ctx, const uint8_t *src, int H, int W, int C, size_t src_stride,
                              const float *map_x, const float *map_y, const uint8_t *valid, int h, int w,
                              int interp, const double *border_value, int fill_value,
                              uint8_t *dst, size_t dst_stride, int slot);

int gs360_equirect_views_u16_host(gs360_ctx *ctx, const uint16_t *src, int W, int H, int C, size_t src_stride,
                                  const gs360_view *views, int n_views,
                                  uint16_t *const *dst, size_t dst_stride, int interp, uint32_t flags, int slot);
int gs360_remap_table_u16_host(gs360_ctx *ctx, const uint16_t *src, int H, int W, int C, size_t src_stride,
                               const float *map_x, const float *map_y, const uint8_t *valid, int h, int w,
                               int interp, const double *border_value, int fill_value,
                               uint16_t *dst, size_t dst_stride, int slot);

/* ---- self-test ------------------------------------------------------------------------------------------------
 * The kernels evaluate EQ-SPEC / FE-SPEC's divisions and square roots with shorter instruction sequences that are bit-identical
 * to IEEE `/` and sqrt on the specs' operand domains.  This compares both forms on the GPU for n_millions x 10^6 pseudo-random
 * operand sets (and the tiny-operand fallback); *n_mismatch must come back 0. */
int gs360_selftest_arith(gs360_ctx *ctx, uint32_t seed, int n_millions, uint64_t *n_checked, uint64_t *n_mismatch);

/* ---- image-codec helper (host only, no GPU) ------------------------------------------------------
 * In-place PNG scanline reconstruction (filter types 0-4) of h rows of (1 + stride) inflated bytes; bpp = bytes per
 * complete pixel.  Used by the package's own 16-bit PNG reader; image codecs are outside the measured path. */
int gs360_png_unfilter(uint8_t *data, int h, int stride, int bpp);
/* TIFF LZW (compression 5) strip decoder for the package's 16-bit TIFF reader: at most out_cap bytes are produced. */
int gs360_tiff_lzw_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif /* GS360_H */
