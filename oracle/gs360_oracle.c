/*
 * gs360_oracle.c -- TEST INFRASTRUCTURE.  CPU restatement of the 360PerspCut /
 * DualFisheye reprojection hot path, used ONLY as the checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
 * product package may import, link or call this file; the product path is the
 * HIP library (360cam-pgm-3dgs-tools_amd/csrc) and fails loudly without it.
 *
 * What is restated, and from where (citations into /root/reference):
 *   orc_remap_u8          the cv2.remap call of cli_tools/gs360_DualFisheyeDistortionCalibration.py
 *                         :1198-1205, :2001-2008 (image, INTER_LINEAR) and :2031-2038 (mask,
 *                         INTER_NEAREST), BORDER_CONSTANT.  cv2 is a third-party wheel
 *                         (requirements.txt:2 "opencv-python>=4.6", unpinned, absent from the
 *                         reference tree and from this image), so its published algorithm
 *                         (OpenCV 4.x imgproc remap: 1/32-px coordinate quantisation, int16
 *                         weights with 15 fractional bits, round-half-even cvRound) is
 *                         restated here.  PARITY UNPINNED at that boundary: the reference has
 *                         no tests or golden images for it; pinned instead by hand-derivable
 *                         integer known-answer tests (tests/test_oracle_remap.py).
 *   orc_remap_u16         the same call on 16-bit sources (DF:735 keeps them at native depth): OpenCV's float-weight
 *                         samplers for ushort, restated next to the function; parity unpinned as well.
 *   orc_valid_fill        DF:1207-1212, DF:2009-2014 (rendered[~valid] = mask_value).
 *   orc_fisheye_map       DF:1759-1823 build_direct_perspective_map_for_lens, with
 *                         DF:975-1005 (_apply_brown_distortion) and DF:1310-1339
 *                         (rotate_view_vectors): float32 arithmetic in the same order; pinned
 *                         against tests/golden/df_goldens.npz (captured by importing the
 *                         reference) to a few float32 ULP (libm vs NumPy SIMD transcendentals).
 *   orc_undistort_map     DF:1008-1051 _remap_for_zoom (as used by DF:1120-1170).
 *   orc_equirect_*        the equirect->rectilinear gather that the reference delegates to
 *                         ffmpeg's v360 filter (cli_tools/gs360_360PerspCut.py:310-314).  ffmpeg
 *                         is an external binary, absent here -> PARITY UNPINNED; geometry follows
 *                         the reference's own statement of the convention, gs360_GUI.py:377-395
 *                         (direction_from_uv) and :419-424 (lonlat_to_xy), evaluated by the
 *                         deterministic float32 formulation "EQ-SPEC v1" 
 *                         that the HIP kernel implements independently; pinned against a
 *                         float64 evaluation of the GUI formulas in tests/test_oracle_equirect.py.
 *   orc_fisheye_spec_*    "FE-SPEC v1": transcendental-free float32 formulation of DF:1759-1823
 *                         used by the fused analytic HIP kernel; pinned against the goldens to
 *                         a stated tolerance.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; see Makefile)
 */
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * POD descriptors.  Declared independently of include/gs360.h on purpose (the oracle is not
 * allowed to share code with the product); tests check the layouts agree.
 * ---------------------------------------------------------------------------------------- */
typedef struct orc_view {
    double yaw_deg, pitch_deg, hfov_deg, vfov_deg; /* ViewSpec fields, PC:32-45 */
    int32_t width, height;
} orc_view;

typedef struct orc_calib { /* SensorCalibration, DF:67-85 */
    int32_t width, height;
    double f, cx, cy, k1, k2, k3, k4, p1, p2, b1, b2;
} orc_calib;

ORC_API int orc_abi_sizes(int which) {
    return which == 0 ? (int)sizeof(orc_view) : (int)sizeof(orc_calib);
}

ORC_API int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int pick_threads(int n_threads) {
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    return n_threads;
#else
    (void)n_threads;
    return 1;
#endif
}

/* ==========================================================================================
 * 1. cv2.remap restatement (OpenCV imgproc, 8-bit, BORDER_CONSTANT)
 * ======================================================================================== */

/* cvRound(float): SSE cvtss2si -- round half to even; NaN / out of range -> INT_MIN. */
static inline int cv_round_f(float v) {
    if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
    return (int)lrintf(v);
}
static inline int sat_s16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline uint8_t sat_u8_d(double v) { /* saturate_cast<uchar>(double) */
    long r = lrint(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

/*
 * INTER_CUBIC weight table (OpenCV imgproc initInterTab2D, fixed point): for each of the 32x32 (fy, fx) sub-pixel
 * phases a 4x4 int16 kernel = saturate_cast<short>(cy[k1] * cx[k2] * 32768) with the Keys coefficients A = -0.75
 * evaluated in float32, followed by OpenCV's sum fix-up: if the 16 entries do not add up to 32768 the difference
 * is taken from the largest (sum too small) or smallest (sum too large) of the entries [2..3]x[2..3] -- the index
 * range the OpenCV source uses (ksize/2 .. ksize/2+1).  RESTATED FROM MEMORY OF THE OPENCV SOURCE, parity unpinned.
 */
static int16_t g_cubic_tab[32 * 32 * 16];
static int g_cubic_ready = 0;

static void cubic_coeffs(float x, float *c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* 2-D fixed-point table from a 1-D coefficient table (32 phases x ksize), with OpenCV's sum fix-up */
static void build_tab2d(const float *tab1, int ks, int16_t *out) {
    const int h = ks / 2;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int16_t *it = out + (i * 32 + j) * ks * ks;
            int isum = 0;
            for (int k1 = 0; k1 < ks; ++k1)
                for (int k2 = 0; k2 < ks; ++k2) {
                    float v = tab1[i * ks + k1] * tab1[j * ks + k2];
                    long r = lrintf(v * 32768.0f);
                    r = r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
                    it[k1 * ks + k2] = (int16_t)r;
                    isum += (int)r;
                }
            if (isum != 32768) {
                int diff = isum - 32768;
                int Mk1 = h, Mk2 = h, mk1 = h, mk2 = h;
                for (int k1 = h; k1 < h + 2; ++k1)
                    for (int k2 = h; k2 < h + 2; ++k2) {
                        if (it[k1 * ks + k2] < it[mk1 * ks + mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1 * ks + k2] > it[Mk1 * ks + Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1 * ks + Mk2] = (int16_t)(it[Mk1 * ks + Mk2] - diff);
                else it[mk1 * ks + mk2] = (int16_t)(it[mk1 * ks + mk2] - diff);
            }
        }
}

static void cubic_init(void) {
    if (g_cubic_ready) return;
    float tab1[32 * 4];
    for (int i = 0; i < 32; ++i) cubic_coeffs((float)i * (1.0f / 32.0f), tab1 + i * 4);
    build_tab2d(tab1, 4, g_cubic_tab);
    g_cubic_ready = 1;
}

/*
 * INTER_LANCZOS4 (OpenCV interpolateLanczos4 + the same 2-D table construction, ksize 8, taps ix-3 .. ix+4): the
 * eight 1-D weights of phase x are sin(pi(x+3-i)/4)-based values computed in double from one sin/cos pair and the
 * eighth-turn rotation table, divided by y^2, rounded to float, then normalised by their float sum; phase 0 is the
 * unit impulse on tap 3.  RESTATED FROM MEMORY OF THE OPENCV SOURCE, parity unpinned.
 */
static int16_t g_lanczos_tab[32 * 32 * 64];
static int g_lanczos_ready = 0;

static void lanczos4_coeffs(float x, float *c) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f) { /* FLT_EPSILON */
        for (int i = 0; i < 8; ++i) c[i] = 0.f;
        c[3] = 1.f;
        return;
    }
    float sum = 0.f;
    double y0 = -(x + 3) * 3.1415926535897932384626433832795 * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; ++i) {
        double y = -(x + 3 - i) * 3.1415926535897932384626433832795 * 0.25;
        c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += c[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; ++i) c[i] *= sum;
}

static void lanczos_init(void) {
    if (g_lanczos_ready) return;
    float tab1[32 * 8];
    for (int i = 0; i < 32; ++i) lanczos4_coeffs((float)i * (1.0f / 32.0f), tab1 + i * 8);
    build_tab2d(tab1, 8, g_lanczos_tab);
    g_lanczos_ready = 1;
}

/* copies the 32*32*64 int16 table (index (fy*32+fx)*64 + ky*8 + kx) */
ORC_API int orc_lanczos4_table(int16_t *out) {
    lanczos_init();
    memcpy(out, g_lanczos_tab, sizeof(g_lanczos_tab));
    return 0;
}

/* copies the 32*32*16 int16 table (index (fy*32+fx)*16 + ky*4 + kx) */
ORC_API int orc_cubic_table(int16_t *out) {
    cubic_init();
    memcpy(out, g_cubic_tab, sizeof(g_cubic_tab));
    return 0;
}

/*
 * interp: 0 = INTER_NEAREST, 1 = INTER_LINEAR, 2 = INTER_CUBIC, 4 = INTER_LANCZOS4.  border_val has 4 entries (cv::Scalar);
 * Python's borderValue=float(v) arrives as (v,0,0,0) -- channel c uses border_val[c & 3].
 * src: H x W x C interleaved u8, row stride src_stride bytes.  dst: h x w x C.
 */
ORC_API int orc_remap_u8(const uint8_t *src, int H, int W, int C, long src_stride,
                         const float *map_x, const float *map_y, int h, int w,
                         int interp, const double *border_val,
                         uint8_t *dst, long dst_stride, int n_threads) {
    if (!src || !map_x || !map_y || !dst || C < 1 || C > 4 || H < 1 || W < 1) return -1;
    if (H >= 32767 || W >= 32767) return -2; /* cv2.remap asserts on SHRT_MAX sizes */
    if (interp != 0 && interp != 1 && interp != 2 && interp != 4) return -3;
    if (interp == 2) cubic_init();
    if (interp == 4) lanczos_init();
    uint8_t cval[4];
    for (int c = 0; c < 4; ++c) cval[c] = sat_u8_d(border_val ? border_val[c] : 0.0);
    int nt = pick_threads(n_threads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *mx = map_x + (size_t)y * w, *my = map_y + (size_t)y * w;
        uint8_t *d = dst + (size_t)y * dst_stride;
        for (int x = 0; x < w; ++x, d += C) {
            if (interp == 0) {
                int ix = sat_s16(cv_round_f(mx[x])), iy = sat_s16(cv_round_f(my[x]));
                if ((unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) {
                    const uint8_t *s = src + (size_t)iy * src_stride + (size_t)ix * C;
                    for (int c = 0; c < C; ++c) d[c] = s[c];
                } else {
                    for (int c = 0; c < C; ++c) d[c] = cval[c];
                }
                continue;
            }
            int sx = cv_round_f(mx[x] * 32.0f), sy = cv_round_f(my[x] * 32.0f);
            int fx = sx & 31, fy = sy & 31;
            int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
            if (interp == 2 || interp == 4) { /* remapBicubic / remapLanczos4: ks x ks window at (ix-ks/2+1, iy-ks/2+1) */
                const int ks = interp == 2 ? 4 : 8;
                int x0 = ix - (ks / 2 - 1), y0 = iy - (ks / 2 - 1);
                if (x0 >= W || x0 + ks <= 0 || y0 >= H || y0 + ks <= 0) {
                    for (int c = 0; c < C; ++c) d[c] = cval[c];
                    continue;
                }
                const int16_t *wt = (interp == 2 ? g_cubic_tab : g_lanczos_tab) + (fy * 32 + fx) * ks * ks;
                for (int c = 0; c < C; ++c) {
                    int acc = 0;
                    for (int ky = 0; ky < ks; ++ky)
                        for (int kx = 0; kx < ks; ++kx) {
                            int xx = x0 + kx, yy = y0 + ky;
                            int v = (xx >= 0 && xx < W && yy >= 0 && yy < H)
                                        ? src[(size_t)yy * src_stride + (size_t)xx * C + c] : cval[c];
                            acc += v * wt[ky * ks + kx];
                        }
                    int r = (acc + (1 << 14)) >> 15;
                    d[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                }
                continue;
            }
            if (ix >= W || ix + 1 < 0 || iy >= H || iy + 1 < 0) {
                for (int c = 0; c < C; ++c) d[c] = cval[c];
                continue;
            }
            int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
            int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            int in_x0 = ix >= 0 && ix < W, in_x1 = ix + 1 >= 0 && ix + 1 < W;
            int in_y0 = iy >= 0 && iy < H, in_y1 = iy + 1 >= 0 && iy + 1 < H;
            const uint8_t *r0 = src + (ptrdiff_t)iy * src_stride + (ptrdiff_t)ix * C;
            const uint8_t *r1 = r0 + src_stride;
            for (int c = 0; c < C; ++c) {
                int v00 = (in_x0 && in_y0) ? r0[c] : cval[c];
                int v01 = (in_x1 && in_y0) ? r0[C + c] : cval[c];
                int v10 = (in_x0 && in_y1) ? r1[c] : cval[c];
                int v11 = (in_x1 && in_y1) ? r1[C + c] : cval[c];
                int acc = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
                int r = (acc + (1 << 14)) >> 15;
                d[c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
        }
    }
    return 0;
}

/*
 * cv2.remap on CV_16U sources (DF:735: cv2.imread(IMREAD_UNCHANGED) keeps 16-bit PNG/TIFF inputs at native depth).
 * OpenCV's dispatch tables send ushort to the FLOAT-weight samplers -- remapBilinear<Cast<float,ushort>, RemapNoVec, float>,
 * remapBicubic<Cast<float,ushort>, float, 1>, remapLanczos4<Cast<float,ushort>, float, 1> -- with the same 1/32-px
 * quantised coordinates as the 8-bit path: 2-D weights = cy[k1] * cx[k2] in float32 from the 1-D phase tables (no
 * fixed-point scaling, no sum fix-up), float32 accumulation in the source's expression order, then
 * saturate_cast<ushort>(float) = cvRound (half to even) clamped to [0, 65535].  Accumulation order as restated:
 *   inside the image   bilinear: ((S00*w0 + S01*w1) + S10*w2) + S11*w3;   bicubic / lanczos4: the ks products of a window
 *                      row are summed left to right, and the row sums are added to the running sum row by row
 *                      (`sum = row0; sum += row1; ...` / `sum = 0; sum += row_r` for Lanczos4);
 *   window on the border (BORDER_CONSTANT)   bilinear: taps outside are replaced by the border value, same expression;
 *                      bicubic / lanczos4: sum = cv, then sum += (S - cv) * w for every in-image tap in row-major order
 *                      (rows above/below the image are skipped).
 * RESTATED FROM MEMORY OF THE OPENCV 4.x SOURCE (imgwarp.cpp), parity unpinned like the 8-bit samplers; the
 * opportunistic cv2 cross-check (tests/test_crosscheck_external.py) covers u16 as well.
 * Strides are in BYTES.
 */
static inline uint16_t sat_u16_f(float v) {
    int iv = cv_round_f(v);
    return (uint16_t)(iv < 0 ? 0 : (iv > 65535 ? 65535 : iv));
}
static inline uint16_t sat_u16_d(double v) {
    long r = lrint(v);
    return (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
}

ORC_API int orc_remap_u16(const uint16_t *src, int H, int W, int C, long src_stride,
                          const float *map_x, const float *map_y, int h, int w,
                          int interp, const double *border_val,
                          uint16_t *dst, long dst_stride, int n_threads) {
    if (!src || !map_x || !map_y || !dst || C < 1 || C > 4 || H < 1 || W < 1) return -1;
    if (H >= 32767 || W >= 32767) return -2;
    if (interp != 0 && interp != 1 && interp != 2 && interp != 4) return -3;
    float lin1[32][2], cub1[32][4], lan1[32][8];
    for (int i = 0; i < 32; ++i) {
        float x = (float)i * (1.0f / 32.0f);
        lin1[i][0] = 1.f - x; lin1[i][1] = x;
        cubic_coeffs(x, cub1[i]);
        lanczos4_coeffs(x, lan1[i]);
    }
    uint16_t cval[4];
    for (int c = 0; c < 4; ++c) cval[c] = sat_u16_d(border_val ? border_val[c] : 0.0);
    const size_t ss = (size_t)src_stride / 2;
    int nt = pick_threads(n_threads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *mx = map_x + (size_t)y * w, *my = map_y + (size_t)y * w;
        uint16_t *d = (uint16_t *)((uint8_t *)dst + (size_t)y * dst_stride);
        for (int x = 0; x < w; ++x, d += C) {
            if (interp == 0) {
                int ix = sat_s16(cv_round_f(mx[x])), iy = sat_s16(cv_round_f(my[x]));
                if ((unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H) {
                    const uint16_t *s = src + (size_t)iy * ss + (size_t)ix * C;
                    for (int c = 0; c < C; ++c) d[c] = s[c];
                } else {
                    for (int c = 0; c < C; ++c) d[c] = cval[c];
                }
                continue;
            }
            int sx = cv_round_f(mx[x] * 32.0f), sy = cv_round_f(my[x] * 32.0f);
            int fx = sx & 31, fy = sy & 31;
            int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
            const int ks = interp == 1 ? 2 : (interp == 2 ? 4 : 8);
            const float *cy = interp == 1 ? lin1[fy] : (interp == 2 ? cub1[fy] : lan1[fy]);
            const float *cx = interp == 1 ? lin1[fx] : (interp == 2 ? cub1[fx] : lan1[fx]);
            int x0 = ix - (ks / 2 - 1), y0 = iy - (ks / 2 - 1);
            if (x0 >= W || x0 + ks <= 0 || y0 >= H || y0 + ks <= 0) {
                for (int c = 0; c < C; ++c) d[c] = cval[c];
                continue;
            }
            const int inside = x0 >= 0 && x0 + ks <= W && y0 >= 0 && y0 + ks <= H;
            for (int c = 0; c < C; ++c) {
                float sum;
                if (interp == 1) {
                    float v[4];
                    for (int ky = 0; ky < 2; ++ky)
                        for (int kx = 0; kx < 2; ++kx) {
                            int xx = x0 + kx, yy = y0 + ky;
                            v[ky * 2 + kx] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? (float)src[(size_t)yy * ss + (size_t)xx * C + c]
                                                                                      : (float)cval[c];
                        }
                    sum = v[0] * (cy[0] * cx[0]) + v[1] * (cy[0] * cx[1]) + v[2] * (cy[1] * cx[0]) + v[3] * (cy[1] * cx[1]);
                } else if (inside) {
                    sum = 0.f;
                    for (int ky = 0; ky < ks; ++ky) {
                        const uint16_t *row = src + (size_t)(y0 + ky) * ss + (size_t)x0 * C + c;
                        float r = (float)row[0] * (cy[ky] * cx[0]);
                        for (int kx = 1; kx < ks; ++kx) r += (float)row[(size_t)kx * C] * (cy[ky] * cx[kx]);
                        sum = (ky == 0 && interp == 2) ? r : sum + r;
                    }
                } else {
                    const float cv = (float)cval[c];
                    sum = cv;      /* cv * ONE with ONE = 1 for the float samplers */
                    for (int ky = 0; ky < ks; ++ky) {
                        int yy = y0 + ky;
                        if (yy < 0 || yy >= H) continue;
                        for (int kx = 0; kx < ks; ++kx) {
                            int xx = x0 + kx;
                            if (xx < 0 || xx >= W) continue;
                            sum += ((float)src[(size_t)yy * ss + (size_t)xx * C + c] - cv) * (cy[ky] * cx[kx]);
                        }
                    }
                }
                d[c] = sat_u16_f(sum);
            }
        }
    }
    return 0;
}

ORC_API int orc_valid_fill_u16(uint16_t *dst, long dst_stride, int h, int w, int C, const uint8_t *valid, int fill) {
    for (int y = 0; y < h; ++y) {
        uint16_t *row = (uint16_t *)((uint8_t *)dst + (size_t)y * dst_stride);
        for (int x = 0; x < w; ++x)
            if (!valid[(size_t)y * w + x])
                for (int c = 0; c < C; ++c) row[(size_t)x * C + c] = (uint16_t)fill;
    }
    return 0;
}

/* rendered[~valid] = fill (all channels), DF:2009-2014 */
ORC_API int orc_valid_fill(uint8_t *dst, long dst_stride, int h, int w, int C,
                           const uint8_t *valid, int fill) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            if (!valid[(size_t)y * w + x])
                for (int c = 0; c < C; ++c) dst[(size_t)y * dst_stride + (size_t)x * C + c] = (uint8_t)fill;
    return 0;
}

/* ==========================================================================================
 * 2. Dual-fisheye map builders (float32, operation order of the reference)
 * ======================================================================================== */

/* DF:975-1005.  Python-float coefficients are "weak" scalars -> float32 arithmetic. */
static inline void brown_f32(float x, float y, const float k[4], float p1, float p2, int tang,
                             float *xd, float *yd, float *r2o) {
    float r2 = (x * x) + (y * y);
    float r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
    float radial = 1.0f + (k[0] * r2);
    radial = radial + (k[1] * r4);
    radial = radial + (k[2] * r6);
    radial = radial + (k[3] * r8);
    float xy = x * y;
    float x_dist = x * radial, y_dist = y * radial;
    if (tang) {
        /* x_dist + (p1*(r2 + (2*x*x))) + (2*p2*xy): 2.0*x is float32, 2.0*calib.p2 is a Python
         * float product (float64) applied as a weak scalar -> rounded to float32 first. */
        float tp2 = (float)(2.0 * (double)p2), tp1 = (float)(2.0 * (double)p1);
        x_dist = (x_dist + (p1 * (r2 + ((2.0f * x) * x)))) + (tp2 * xy);
        y_dist = (y_dist + (p2 * (r2 + ((2.0f * y) * y)))) + (tp1 * xy);
    }
    *xd = x_dist; *yd = y_dist; *r2o = r2;
}

/*
 * DF:1759-1823.  NumPy >= 2 evaluates np.tan(hfov/2) * uu in float64 (np.float64 scalar times a
 * float32 array promotes), then stores to the float32 rays array; restated that way (numpy2 != 0).
 * With numpy2 == 0 the NumPy 1.x behaviour (float32 product) is used.
 */
ORC_API int orc_fisheye_map(const orc_calib *cal, double yaw_deg, double pitch_deg,
                            double hfov_deg, double vfov_deg, int out_w, int out_h,
                            double lens_fov_deg, int numpy2,
                            float *map_x, float *map_y, uint8_t *valid, int n_threads) {
    if (!cal || out_w < 1 || out_h < 1) return -1;
    const double PI = 3.14159265358979323846;
    double hf = fmax(1e-3, fmin(179.9, hfov_deg)) * PI / 180.0;
    double vf = fmax(1e-3, fmin(179.9, vfov_deg)) * PI / 180.0;
    double th = tan(hf * 0.5), tv = tan(vf * 0.5);
    double pitch = pitch_deg * PI / 180.0, yaw = yaw_deg * PI / 180.0;
    float cos_p = (float)cos(pitch), sin_p = (float)sin(pitch);
    float cos_y = (float)cos(yaw), sin_y = (float)sin(yaw);
    float nsin_p = (float)(-sin(pitch)), nsin_y = (float)(-sin(yaw));
    float theta_max = (float)((fmax(1.0, fmin(360.0, lens_fov_deg)) * 0.5) * PI / 180.0);
    float k[4] = {(float)cal->k1, (float)cal->k2, (float)cal->k3, (float)cal->k4};
    float p1 = (float)cal->p1, p2 = (float)cal->p2, b1 = (float)cal->b1, b2 = (float)cal->b2;
    int tang = (cal->p1 != 0.0) || (cal->p2 != 0.0);
    float fc = (float)cal->f;
    float cx0 = (float)((cal->width * 0.5) + cal->cx), cy0 = (float)((cal->height * 0.5) + cal->cy);
    float wmax = (float)(cal->width - 1), hmax = (float)(cal->height - 1);
    float fw = (float)out_w, fh = (float)out_h;
    int nt = pick_threads(n_threads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int j = 0; j < out_h; ++j) {
        float vv = (((float)j + 0.5f) / fh) * 2.0f - 1.0f;
        for (int i = 0; i < out_w; ++i) {
            float uu = (((float)i + 0.5f) / fw) * 2.0f - 1.0f;
            float x, y;
            if (numpy2) { x = (float)(th * (double)uu); y = (float)(tv * (double)(-vv)); }
            else { x = (float)th * uu; y = (float)tv * (-vv); }
            float z = 1.0f;
            /* np.linalg.norm(axis=2) on float32: sqrt(sum of squares) in float32 (pairwise add of 3) */
            float nrm = sqrtf(((x * x) + (y * y)) + (z * z));
            nrm = fmaxf(nrm, 1e-12f);
            x = x / nrm; y = y / nrm; z = z / nrm;
            float y1 = (cos_p * y) + (sin_p * z);
            float z1 = (nsin_p * y) + (cos_p * z);
            float rx = (cos_y * x) + (sin_y * z1);
            float rz = (nsin_y * x) + (cos_y * z1);
            float ry = y1;
            float rzc = fminf(fmaxf(rz, -1.0f), 1.0f);
            float theta = acosf(rzc);
            float rho = sqrtf((rx * rx) + (ry * ry));
            float scale = 0.0f;
            if (rho > 1e-12f) scale = (2.0f * sinf(theta * 0.5f)) / rho;
            float x_n = rx * scale, y_n = (-ry) * scale;
            float xd, yd, r2;
            brown_f32(x_n, y_n, k, p1, p2, tang, &xd, &yd, &r2);
            float mx = ((cx0 + (xd * fc)) + (xd * b1)) + (yd * b2);
            float my = cy0 + (yd * fc);
            size_t o = (size_t)j * out_w + i;
            map_x[o] = mx; map_y[o] = my;
            if (valid)
                valid[o] = (theta <= theta_max) && (mx >= 0.0f) && (mx <= wmax) && (my >= 0.0f) && (my <= hmax);
        }
    }
    return 0;
}

/* DF:1008-1051 for the full-resolution grid of DF:1138-1163 (dst = arange meshgrid). */
ORC_API int orc_undistort_map(const orc_calib *cal, double zoom, double lens_fov_deg,
                              float *map_x, float *map_y, uint8_t *valid, int n_threads) {
    if (!cal) return -1;
    const double PI = 3.14159265358979323846;
    int W = cal->width, H = cal->height;
    float cx0 = (float)((W * 0.5) + cal->cx), cy0 = (float)((H * 0.5) + cal->cy);
    float den_y = (float)cal->f, den_x = (float)(cal->f + cal->b1);
    float k[4] = {(float)cal->k1, (float)cal->k2, (float)cal->k3, (float)cal->k4};
    float p1 = (float)cal->p1, p2 = (float)cal->p2, b1 = (float)cal->b1, b2 = (float)cal->b2;
    int tang = (cal->p1 != 0.0) || (cal->p2 != 0.0);
    float fc = (float)cal->f, zf = (float)zoom;
    float theta_max = (float)((fmax(1.0, fmin(360.0, lens_fov_deg)) * 0.5) * PI / 180.0);
    float wmax = (float)(W - 1), hmax = (float)(H - 1);
    int nt = pick_threads(n_threads);
    (void)nt;
#pragma omp parallel for num_threads(nt) schedule(static)
    for (int j = 0; j < H; ++j) {
        float y0 = ((float)j - cy0) / den_y;
        for (int i = 0; i < W; ++i) {
            float x0 = (((float)i - cx0) - (y0 * b2)) / den_x;
            float x = x0 / zf, y = y0 / zf;
            float xd, yd, r2;
            brown_f32(x, y, k, p1, p2, tang, &xd, &yd, &r2);
            float sx = ((cx0 + (xd * fc)) + (xd * b1)) + (yd * b2);
            float sy = cy0 + (yd * fc);
            float r = sqrtf(fmaxf(r2, 0.0f));
            float theta = 2.0f * asinf(fminf(fmaxf(r * 0.5f, 0.0f), 1.0f));
            size_t o = (size_t)j * W + i;
            map_x[o] = sx; map_y[o] = sy;
            if (valid)
                valid[o] = (theta <= theta_max) && (sx >= 0.0f) && (sx <= wmax) && (sy >= 0.0f) && (sy <= hmax);
        }
    }
    return 0;
}

/* ==========================================================================================
 * 3. EQ-SPEC v1: deterministic float32 equirect map
 *
 *   Every operation below is a single IEEE-754 binary32 operation (+,-,*,/,sqrt,fma,rint);
 *   the file is compiled with -ffp-contract=off so nothing else is fused.  The angle is kept as
 *   sigma*r0 + K*(pi/4) with |r0| <= pi/8: the K*(pi/4) part is applied as an exact INTEGER
 *   offset in 1/32-pixel units (K*4W horizontally, K*8H vertically), only r0 goes through
 *   floating point.  Horizontal border = wrap, vertical border = clamp.
 * ======================================================================================== */
typedef struct eq_consts {
    float sxu, syv;     /* tan(hfov/2)/out_w, tan(vfov/2)/out_h */
    float sp, cp;       /* sin, cos of pitch */
    float kx32, ky32;   /* 32*W/(2*pi), 32*H/pi */
    float x0f32;        /* 32 * frac((yaw/360 + 1/2)*W - 1/2) */
    int32_t x0i32;      /* 32 * (floor(...) mod W) */
    int32_t y0i32;      /* 16*H - 16 */
    int32_t W, H, out_w, out_h;
    int32_t fish;       /* 0: rectilinear (pinhole) output; 1: equidistant fisheye output (PC:351-414, v360 output=fisheye) */
} eq_consts;

static void eq_make_consts_proj(const orc_view *v, int W, int H, int fish, eq_consts *c) {
    const double PI = 3.14159265358979323846;
    double hf = fmax(1e-3, fmin(179.9, v->hfov_deg)) * PI / 180.0; /* clamp as GUI:437-438 */
    double vf = fmax(1e-3, fmin(179.9, v->vfov_deg)) * PI / 180.0;
    c->sxu = (float)(tan(hf * 0.5) / (double)v->width);
    c->syv = (float)(tan(vf * 0.5) / (double)v->height);
    c->fish = fish;
    if (fish) { /* image-plane radius 1 <-> 90 degrees off axis; the view spans hfov x vfov degrees */
        c->sxu = (float)(fmax(1e-3, fmin(360.0, v->hfov_deg)) / 180.0 / (double)v->width);
        c->syv = (float)(fmax(1e-3, fmin(360.0, v->vfov_deg)) / 180.0 / (double)v->height);
    }
    double pitch = v->pitch_deg * PI / 180.0;
    c->sp = (float)sin(pitch);
    c->cp = (float)cos(pitch);
    c->kx32 = (float)(32.0 * (double)W / (2.0 * PI));
    c->ky32 = (float)(32.0 * (double)H / PI);
    double x0 = (v->yaw_deg / 360.0 + 0.5) * (double)W - 0.5; /* lonlat_to_xy, GUI:419-424, minus 1/2 px */
    double x0fl = floor(x0);
    c->x0f32 = (float)(32.0 * (x0 - x0fl));
    long xi = (long)x0fl % (long)W;
    if (xi < 0) xi += W;
    c->x0i32 = (int32_t)(32 * xi);
    c->y0i32 = 16 * H - 16;
    c->W = W; c->H = H; c->out_w = v->width; c->out_h = v->height;
}

static void eq_make_consts(const orc_view *v, int W, int H, eq_consts *c) { eq_make_consts_proj(v, W, H, 0, c); }

#define EQ_T8 0x1.a8279ap-2f /* tan(pi/8) rounded to float32 */
static const float EQ_C1 = -0.33333316445350647f, EQ_C2 = 0.199985072016716f,
                   EQ_C3 = -0.14244139194488525f, EQ_C4 = 0.10597943514585495f,
                   EQ_C5 = -0.06087981536984444f;

/* atan2(yy, xx) = sr0 + K*(pi/4);  returns sr0 (sign folded in), K in [-4, 4]. */
static inline float eq_atan2_red(float yy, float xx, int *Kout) {
    float ax = fabsf(xx), ay = fabsf(yy);
    float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    int big = mn > EQ_T8 * mx;
    float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    float t = den > 0.0f ? num / den : 0.0f;
    float z = t * t;
    float p = fmaf(EQ_C5, z, EQ_C4);
    p = fmaf(p, z, EQ_C3);
    p = fmaf(p, z, EQ_C2);
    p = fmaf(p, z, EQ_C1);
    float r0 = fmaf(p * z, t, t);
    int K = big;
    if (ay > ax) { r0 = -r0; K = 2 - K; }
    if (xx < 0.0f) { r0 = -r0; K = 4 - K; }
    if (yy < 0.0f) { r0 = -r0; K = -K; }
    *Kout = K;
    return r0;
}

/* Equidistant-fisheye output (EQ-SPEC v1, projection F): image-plane point (u, v), radius r = |(u, v)| <-> r*90 degrees
 * off the optical axis, ray = (u S, v S, C) with S = sin(pi r/2)/r and C = cos(pi r/2), both even in r and evaluated
 * as degree-8 polynomials in q = r^2 on [0, 4] (float32 Horner with fma; |error| < 4e-7, i.e. < 0.001 px at 8K). */
static const float EQ_FS[9] = {1.5707963705062866f, -0.6459640860557556f, 0.07969262450933456f, -0.004681753925979137f,
                               0.0001604411081643775f, -3.598792090997449e-06f, 5.689994608815141e-08f,
                               -6.633614213491512e-10f, 5.326020006968246e-12f};
static const float EQ_FC[9] = {1.0f, -1.2337005138397217f, 0.25366950035095215f, -0.020863480865955353f,
                               0.0009192594443447888f, -2.5201432436006144e-05f, 4.708266487796209e-07f,
                               -6.321354106830768e-09f, 5.675555858619674e-11f};
static inline float eq_poly8(const float *k, float q) {
    float p = k[8];
    for (int n = 7; n >= 0; --n) p = fmaf(p, q, k[n]);
    return p;
}

/* quantised source coordinate (1/32 px) for output pixel (i, j) */
static inline void eq_coord(const eq_consts *c, int i, int j, int *sxo, int *syo) {
    float x = (float)(2 * i + 1 - c->out_w) * c->sxu;
    float yv = (float)(2 * j + 1 - c->out_h) * c->syv;
    float b, cc;
    if (c->fish) {
        float q = fmaf(x, x, yv * yv);
        float S = eq_poly8(EQ_FS, q), Cz = eq_poly8(EQ_FC, q);
        x = x * S;
        yv = yv * S;
        b = fmaf(c->sp, yv, c->cp * Cz);
        cc = fmaf(-c->cp, yv, c->sp * Cz);
    } else {
        b = fmaf(c->sp, yv, c->cp);   /* forward component after pitch */
        cc = fmaf(-c->cp, yv, c->sp); /* up component after pitch */
    }
    float h = sqrtf(fmaf(x, x, b * b));
    int Kl, Kt;
    float rl = eq_atan2_red(x, b, &Kl);
    float rt = eq_atan2_red(cc, h, &Kt);
    int sx = (int)rintf(fmaf(rl, c->kx32, c->x0f32)) + c->x0i32 + Kl * 4 * c->W;
    int W32 = 32 * c->W;
    if (sx < 0) sx += W32;
    if (sx >= W32) sx -= W32;
    int sy = c->y0i32 - Kt * 8 * c->H - (int)rintf(rt * c->ky32);
    *sxo = sx; *syo = sy;
}

/* Export the quantised map and (optionally) the de-quantised float coordinates for tests. */
ORC_API int orc_equirect_map(const orc_view *v, int W, int H, int32_t *sx_out, int32_t *sy_out) {
    if (!v || W < 2 || H < 2 || v->width < 1 || v->height < 1) return -1;
    eq_consts c;
    eq_make_consts(v, W, H, &c);
    for (int j = 0; j < c.out_h; ++j)
        for (int i = 0; i < c.out_w; ++i) {
            int sx, sy;
            eq_coord(&c, i, j, &sx, &sy);
            sx_out[(size_t)j * c.out_w + i] = sx;
            sy_out[(size_t)j * c.out_w + i] = sy;
        }
    return 0;
}

ORC_API int orc_equirect_map_proj(const orc_view *v, int W, int H, int fish, int32_t *sx_out, int32_t *sy_out) {
    if (!v || W < 2 || H < 2 || v->width < 1 || v->height < 1) return -1;
    eq_consts c;
    eq_make_consts_proj(v, W, H, fish, &c);
    for (int j = 0; j < c.out_h; ++j)
        for (int i = 0; i < c.out_w; ++i) {
            int sx, sy;
            eq_coord(&c, i, j, &sx, &sy);
            sx_out[(size_t)j * c.out_w + i] = sx;
            sy_out[(size_t)j * c.out_w + i] = sy;
        }
    return 0;
}

static inline void eq_sample_px(const eq_consts *c, const uint8_t *src, long stride, int C,
                                int sx, int sy, uint8_t *d) {
    int fx = sx & 31, ix = sx >> 5;
    int fy = sy & 31, iy = sy >> 5;
    int ix1 = ix + 1 == c->W ? 0 : ix + 1;
    int y0 = iy < 0 ? 0 : (iy > c->H - 1 ? c->H - 1 : iy);
    int y1 = iy + 1 < 0 ? 0 : (iy + 1 > c->H - 1 ? c->H - 1 : iy + 1);
    const uint8_t *r0 = src + (size_t)y0 * stride, *r1 = src + (size_t)y1 * stride;
    int a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
    for (int ch = 0; ch < C; ++ch) {
        int acc = (r0[ix * C + ch] * a0 + r0[ix1 * C + ch] * a1) * b0 +
                  (r1[ix * C + ch] * a0 + r1[ix1 * C + ch] * a1) * b1;
        d[ch] = (uint8_t)((acc + 512) >> 10);
    }
}

/* EQ-SPEC cubic: same quantised coordinate; 4x4 window (ix-1.., iy-1..) with columns wrapping and rows clamping;
 * OpenCV's fixed-point Keys table; (sum + 2^14) >> 15 saturated. */
static inline void eq_sample_px_cubic(const eq_consts *c, const uint8_t *src, long stride, int C,
                                      int sx, int sy, uint8_t *d) {
    int fx = sx & 31, ix = sx >> 5, fy = sy & 31, iy = sy >> 5;
    const int16_t *wt = g_cubic_tab + (fy * 32 + fx) * 16;
    for (int ch = 0; ch < C; ++ch) {
        int acc = 0;
        for (int ky = 0; ky < 4; ++ky) {
            int yy = iy - 1 + ky;
            yy = yy < 0 ? 0 : (yy > c->H - 1 ? c->H - 1 : yy);
            for (int kx = 0; kx < 4; ++kx) {
                int xx = ix - 1 + kx;
                xx = xx < 0 ? xx + c->W : (xx >= c->W ? xx - c->W : xx);
                acc += src[(size_t)yy * stride + (size_t)xx * C + ch] * wt[ky * 4 + kx];
            }
        }
        int r = (acc + (1 << 14)) >> 15;
        d[ch] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}

ORC_API int orc_equirect_views_u8_interp(const uint8_t *src, int W, int H, int C, long src_stride,
                                         const orc_view *views, int n_views,
                                         uint8_t *const *dst, long dst_stride, int interp, int n_threads);

/* One frame, n views.  dst[k] -> view k (height x width x C, tight unless dst_stride given).
 * All (view, row) pairs form ONE parallel loop so that many host cores stay busy on small views. */
ORC_API int orc_equirect_views_u8(const uint8_t *src, int W, int H, int C, long src_stride,
                                  const orc_view *views, int n_views,
                                  uint8_t *const *dst, long dst_stride, int n_threads) {
    return orc_equirect_views_u8_interp(src, W, H, C, src_stride, views, n_views, dst, dst_stride, 1, n_threads);
}

ORC_API int orc_equirect_views_masked_u8(const uint8_t *src, const uint8_t *mask, int W, int H, int C, long src_stride,
                                         long mask_stride, const orc_view *views, int n_views,
                                         uint8_t *const *dst, long dst_stride, int interp, int n_threads);

ORC_API int orc_equirect_views_u8_interp(const uint8_t *src, int W, int H, int C, long src_stride,
                                         const orc_view *views, int n_views,
                                         uint8_t *const *dst, long dst_stride, int interp, int n_threads) {
    return orc_equirect_views_masked_u8(src, NULL, W, H, C, src_stride, 0, views, n_views, dst, dst_stride, interp, n_threads);
}

/* mask != NULL: keep-mask fused into the output (BASELINE config 5, build-defined; mask convention of the reference's
 * SegmentationMaskTool, SEG:765-774: 0 = masked, 255 = keep): nearest texel of the same quantised coordinate,
 * out = 0 where mask < 128. */
static int eq_views_impl(const uint8_t *src, const uint8_t *mask, int W, int H, int C, long src_stride,
                         long mask_stride, const orc_view *views, int n_views,
                         uint8_t *const *dst, long dst_stride, int interp, int n_threads, int fish);

ORC_API int orc_equirect_views_masked_u8(const uint8_t *src, const uint8_t *mask, int W, int H, int C, long src_stride,
                                         long mask_stride, const orc_view *views, int n_views,
                                         uint8_t *const *dst, long dst_stride, int interp, int n_threads) {
    return eq_views_impl(src, mask, W, H, C, src_stride, mask_stride, views, n_views, dst, dst_stride, interp, n_threads, 0);
}

/* equirect -> equidistant-fisheye views (the `fisheyeXY` preset's v360 output=fisheye jobs, PC:351-414):
 * views[k].hfov_deg / vfov_deg = full horizontal / vertical field of view of the fisheye image */
ORC_API int orc_equirect_fisheye_views_u8(const uint8_t *src, int W, int H, int C, long src_stride,
                                          const orc_view *views, int n_views,
                                          uint8_t *const *dst, long dst_stride, int interp, int n_threads) {
    return eq_views_impl(src, NULL, W, H, C, src_stride, 0, views, n_views, dst, dst_stride, interp, n_threads, 1);
}

static int eq_views_impl(const uint8_t *src, const uint8_t *mask, int W, int H, int C, long src_stride,
                         long mask_stride, const orc_view *views, int n_views,
                         uint8_t *const *dst, long dst_stride, int interp, int n_threads, int fish) {
    if (interp != 1 && interp != 2) return -3;
    if (mask && mask_stride == 0) mask_stride = W;
    if (interp == 2) cubic_init();
    if (!src || !views || !dst || C < 1 || C > 4 || W < 2 || H < 2 || n_views < 0) return -1;
    if (n_views == 0) return 0;
    if (src_stride == 0) src_stride = (long)W * C;
    int nt = pick_threads(n_threads);
    (void)nt;
    eq_consts *cs = (eq_consts *)malloc(sizeof(eq_consts) * (size_t)n_views);
    long *row0 = (long *)malloc(sizeof(long) * (size_t)(n_views + 1));
    if (!cs || !row0) { free(cs); free(row0); return -4; }
    row0[0] = 0;
    for (int k = 0; k < n_views; ++k) {
        eq_make_consts_proj(&views[k], W, H, fish, &cs[k]);
        row0[k + 1] = row0[k] + cs[k].out_h;
    }
    long total_rows = row0[n_views];
#pragma omp parallel for num_threads(nt) schedule(dynamic, 8)
    for (long rr = 0; rr < total_rows; ++rr) {
        int k = 0;
        while (rr >= row0[k + 1]) ++k;
        const eq_consts *c = &cs[k];
        int j = (int)(rr - row0[k]);
        long ds = dst_stride ? dst_stride : (long)c->out_w * C;
        uint8_t *out = dst[k] + (size_t)j * ds;
        for (int i = 0; i < c->out_w; ++i) {
            int sx, sy;
            eq_coord(c, i, j, &sx, &sy);
            if (interp == 2) eq_sample_px_cubic(c, src, src_stride, C, sx, sy, out + (size_t)i * C);
            else eq_sample_px(c, src, src_stride, C, sx, sy, out + (size_t)i * C);
            if (mask) {
                int xn = (sx + 16) >> 5, yn = (sy + 16) >> 5;
                if (xn >= W) xn -= W;
                yn = yn < 0 ? 0 : (yn > H - 1 ? H - 1 : yn);
                if (mask[(size_t)yn * mask_stride + xn] < 128)
                    for (int ch = 0; ch < C; ++ch) out[(size_t)i * C + ch] = 0;
            }
        }
    }
    free(cs); free(row0);
    return 0;
}

/*
 * EQ-SPEC v1 on 16-bit sources (16-bit stills keep their depth through the reference's ffmpeg path, PC:327-347 writes no
 * -pix_fmt for PNG/TIFF stills; > 8-bit videos leave as rgb48le, PC:343-347).  Same quantised coordinates; the samplers
 * are the 8-bit ones with 16-bit texels: bilinear (sum S a b + 512) >> 10, bicubic (sum S w + 2^14) >> 15 with the
 * fixed-point Keys table in 64-bit, clamped to [0, 65535].  Build-defined like the rest of EQ-SPEC.  Strides in BYTES.
 */
ORC_API int orc_equirect_views_u16(const uint16_t *src, int W, int H, int C, long src_stride,
                                   const orc_view *views, int n_views,
                                   uint16_t *const *dst, long dst_stride, int interp, int fish, int n_threads) {
    if (interp != 1 && interp != 2) return -3;
    if (interp == 2) cubic_init();
    if (!src || !views || !dst || C < 1 || C > 4 || W < 2 || H < 2 || n_views < 0) return -1;
    if (src_stride == 0) src_stride = (long)W * C * 2;
    const size_t ss = (size_t)src_stride / 2;
    int nt = pick_threads(n_threads);
    (void)nt;
    for (int k = 0; k < n_views; ++k) {
        eq_consts c;
        eq_make_consts_proj(&views[k], W, H, fish, &c);
        long ds = dst_stride ? dst_stride : (long)c.out_w * C * 2;
#pragma omp parallel for num_threads(nt) schedule(dynamic, 8)
        for (int j = 0; j < c.out_h; ++j) {
            uint16_t *out = (uint16_t *)((uint8_t *)dst[k] + (size_t)j * ds);
            for (int i = 0; i < c.out_w; ++i) {
                int sx, sy;
                eq_coord(&c, i, j, &sx, &sy);
                int fx = sx & 31, ix = sx >> 5, fy = sy & 31, iy = sy >> 5;
                if (interp == 1) {
                    int ix1 = ix + 1 == W ? 0 : ix + 1;
                    int y0 = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
                    int y1 = iy + 1 < 0 ? 0 : (iy + 1 > H - 1 ? H - 1 : iy + 1);
                    const uint16_t *r0 = src + (size_t)y0 * ss, *r1 = src + (size_t)y1 * ss;
                    uint32_t a0 = 32 - fx, a1 = fx, b0 = 32 - fy, b1 = fy;
                    for (int ch = 0; ch < C; ++ch) {
                        uint32_t acc = (r0[ix * C + ch] * a0 + r0[ix1 * C + ch] * a1) * b0 +
                                       (r1[ix * C + ch] * a0 + r1[ix1 * C + ch] * a1) * b1;
                        out[(size_t)i * C + ch] = (uint16_t)((acc + 512) >> 10);
                    }
                } else {
                    const int16_t *wt = g_cubic_tab + (fy * 32 + fx) * 16;
                    for (int ch = 0; ch < C; ++ch) {
                        int64_t acc = 0;
                        for (int ky = 0; ky < 4; ++ky) {
                            int yy = iy - 1 + ky;
                            yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
                            for (int kx = 0; kx < 4; ++kx) {
                                int xx = ix - 1 + kx;
                                xx = xx < 0 ? xx + W : (xx >= W ? xx - W : xx);
                                acc += (int64_t)src[(size_t)yy * ss + (size_t)xx * C + ch] * wt[ky * 4 + kx];
                            }
                        }
                        int64_t r = (acc + (1 << 14)) >> 15;
                        out[(size_t)i * C + ch] = (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
                    }
                }
            }
        }
    }
    return 0;
}

/*
 * Algorithmic-bytes helper: number of DISTINCT source texels referenced by any of the
 * four bilinear taps of one view (U_v), plus optionally OR-ing them into a caller bitmap (W*H bytes)
 * so the union over views can be counted too.
 */
ORC_API long orc_equirect_distinct_texels(const orc_view *v, int W, int H, uint8_t *union_bitmap) {
    eq_consts c;
    eq_make_consts(v, W, H, &c);
    uint8_t *bm = (uint8_t *)calloc((size_t)W * H, 1);
    if (!bm) return -1;
    for (int j = 0; j < c.out_h; ++j)
        for (int i = 0; i < c.out_w; ++i) {
            int sx, sy;
            eq_coord(&c, i, j, &sx, &sy);
            int ix = sx >> 5, iy = sy >> 5;
            int ix1 = ix + 1 == W ? 0 : ix + 1;
            int y0 = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
            int y1 = iy + 1 < 0 ? 0 : (iy + 1 > H - 1 ? H - 1 : iy + 1);
            bm[(size_t)y0 * W + ix] = 1; bm[(size_t)y0 * W + ix1] = 1;
            bm[(size_t)y1 * W + ix] = 1; bm[(size_t)y1 * W + ix1] = 1;
        }
    long n = 0;
    for (size_t t = 0; t < (size_t)W * H; ++t) {
        n += bm[t];
        if (union_bitmap && bm[t]) union_bitmap[t] = 1;
    }
    free(bm);
    return n;
}

/* distinct texels referenced by a table-mode bilinear remap (fisheye configs) */
ORC_API long orc_table_distinct_texels(const float *map_x, const float *map_y, long n_px, int W, int H) {
    uint8_t *bm = (uint8_t *)calloc((size_t)W * H, 1);
    if (!bm) return -1;
    for (long t = 0; t < n_px; ++t) {
        int sx = cv_round_f(map_x[t] * 32.0f), sy = cv_round_f(map_y[t] * 32.0f);
        int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                int xx = ix + dx, yy = iy + dy;
                if (xx >= 0 && xx < W && yy >= 0 && yy < H) bm[(size_t)yy * W + xx] = 1;
            }
    }
    long n = 0;
    for (size_t t = 0; t < (size_t)W * H; ++t) n += bm[t];
    free(bm);
    return n;
}

/* ==========================================================================================
 * 4. FE-SPEC v1: transcendental-free float32 formulation of DF:1759-1823 for the fused kernel.
 *
 *    With R = (X,Y,Z) the rotated, UN-normalised ray and N = |R|:  theta = acos(Z/N),
 *    2*sin(theta/2)/rho = sqrt(2/(1+Z/N)) / N  ==>  x_n = X*s, y_n = -Y*s, s = sqrt(2/(N*(N+Z))).
 *    valid_model  <=>  Z >= cos(theta_max)*N.
 * ======================================================================================== */
typedef struct fe_consts {
    float sxu, syv, sp, cp, sy_, cy_; /* sin/cos pitch, sin/cos yaw */
    float k1, k2, k3, k4, p1, p2, tp1, tp2, b1, b2, f, cx0, cy0, wmax, hmax, cos_tmax;
    int32_t tang, out_w, out_h;
} fe_consts;

static void fe_make_consts(const orc_calib *cal, double yaw_deg, double pitch_deg, double hfov_deg,
                           double vfov_deg, int out_w, int out_h, double lens_fov_deg, fe_consts *c) {
    const double PI = 3.14159265358979323846;
    double hf = fmax(1e-3, fmin(179.9, hfov_deg)) * PI / 180.0;
    double vf = fmax(1e-3, fmin(179.9, vfov_deg)) * PI / 180.0;
    c->sxu = (float)(tan(hf * 0.5) / (double)out_w);
    c->syv = (float)(tan(vf * 0.5) / (double)out_h);
    double pitch = pitch_deg * PI / 180.0, yaw = yaw_deg * PI / 180.0;
    c->sp = (float)sin(pitch); c->cp = (float)cos(pitch);
    c->sy_ = (float)sin(yaw); c->cy_ = (float)cos(yaw);
    c->k1 = (float)cal->k1; c->k2 = (float)cal->k2; c->k3 = (float)cal->k3; c->k4 = (float)cal->k4;
    c->p1 = (float)cal->p1; c->p2 = (float)cal->p2;
    c->tp1 = (float)(2.0 * cal->p1); c->tp2 = (float)(2.0 * cal->p2);
    c->b1 = (float)cal->b1; c->b2 = (float)cal->b2; c->f = (float)cal->f;
    c->cx0 = (float)((cal->width * 0.5) + cal->cx);
    c->cy0 = (float)((cal->height * 0.5) + cal->cy);
    c->wmax = (float)(cal->width - 1); c->hmax = (float)(cal->height - 1);
    c->cos_tmax = (float)cos((fmax(1.0, fmin(360.0, lens_fov_deg)) * 0.5) * PI / 180.0);
    c->tang = (cal->p1 != 0.0) || (cal->p2 != 0.0);
    c->out_w = out_w; c->out_h = out_h;
}

static inline int fe_coord(const fe_consts *c, int i, int j, float *mxo, float *myo) {
    float x = (float)(2 * i + 1 - c->out_w) * c->sxu;
    float yv = (float)(2 * j + 1 - c->out_h) * c->syv; /* ray y = -yv */
    float Y = fmaf(-c->cp, yv, c->sp);                  /* y1 = cp*y + sp*z */
    float z1 = fmaf(c->sp, yv, c->cp);                  /* z1 = -sp*y + cp*z */
    float X = fmaf(c->cy_, x, c->sy_ * z1);
    float Z = fmaf(-c->sy_, x, c->cy_ * z1);
    float N2 = fmaf(x, x, fmaf(yv, yv, 1.0f));
    float N = sqrtf(N2);
    float d = N * (N + Z);
    float s = d > 0.0f ? sqrtf(2.0f / d) : 0.0f;
    float xn = X * s, yn = -(Y * s);
    float r2 = fmaf(xn, xn, yn * yn);
    float r4 = r2 * r2;
    float radial = fmaf(c->k4, r4 * r4, fmaf(c->k3, r4 * r2, fmaf(c->k2, r4, fmaf(c->k1, r2, 1.0f))));
    float xd = xn * radial, yd = yn * radial;
    if (c->tang) {
        float xy = xn * yn;
        xd = fmaf(c->tp2, xy, fmaf(c->p1, fmaf(2.0f * xn, xn, r2), xd));
        yd = fmaf(c->tp1, xy, fmaf(c->p2, fmaf(2.0f * yn, yn, r2), yd));
    }
    float mx = fmaf(yd, c->b2, fmaf(xd, c->b1, fmaf(xd, c->f, c->cx0)));
    float my = fmaf(yd, c->f, c->cy0);
    *mxo = mx; *myo = my;
    return (Z >= c->cos_tmax * N) && (mx >= 0.0f) && (mx <= c->wmax) && (my >= 0.0f) && (my <= c->hmax);
}

ORC_API int orc_fisheye_spec_map(const orc_calib *cal, double yaw_deg, double pitch_deg,
                                 double hfov_deg, double vfov_deg, int out_w, int out_h,
                                 double lens_fov_deg, float *map_x, float *map_y, uint8_t *valid) {
    if (!cal || out_w < 1 || out_h < 1) return -1;
    fe_consts c;
    fe_make_consts(cal, yaw_deg, pitch_deg, hfov_deg, vfov_deg, out_w, out_h, lens_fov_deg, &c);
    for (int j = 0; j < out_h; ++j)
        for (int i = 0; i < out_w; ++i) {
            float mx, my;
            int ok = fe_coord(&c, i, j, &mx, &my);
            size_t o = (size_t)j * out_w + i;
            map_x[o] = mx; map_y[o] = my;
            if (valid) valid[o] = (uint8_t)ok;
        }
    return 0;
}
