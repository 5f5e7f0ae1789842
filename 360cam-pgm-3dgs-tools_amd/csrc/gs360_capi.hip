// gs360_capi.hip -- C-ABI glue of libgs360hip.so (declared in include/gs360.h).
// Host side only: context / streams / events / memory, per-view constant preparation (float64 -> one
// rounding to float32) and launch batching.  No CPU compute path exists here on purpose: without a GPU
// every entry point fails with GS360_ERR_NODEV / GS360_ERR_HIP.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "gs360_kernels.h"

using namespace gs360;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr size_t kSmLdsPerGroup = 80 * 1024;   // two workgroups of the source-major kernel per CU (160 KiB of LDS)
constexpr double kSmMinPixels = 3.5e6;           // automatic selection of the source-major kernel: output pixels of the call (smaller calls are launch-bound)
constexpr int kSmFamilyMinFrames = 4;          // automatic selection of the source-major kernel for calls of several rings: frames per call ...
constexpr int kTsBoxBudget = 26 * 1024 - 64;     // largest tile box of the LDS-staged table kernel: two of them per workgroup, three workgroups per CU
constexpr int kSmMaxBoxPct = 160;              // ... and tile boxes at most this large relative to their grid cells (profiles/r05/srcmajor_family_sweep.txt)

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(GS360_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kMaxSlots = 16;
constexpr int kEventsPerSlot = 8;
constexpr size_t kSlack = 64;
constexpr double kPi = 3.14159265358979323846;

// context options: name, default, range, environment seed (user switches only)
enum Opt { kOptLanemap, kOptStage, kOptRing, kOptXcdGroup, kOptEqPersist, kOptTablePersist, kOptLanczosTable, kOptTableRows, kOptColorCube,
           kOptSrcMajor, kOptSrcMajorBx, kOptSrcMajorRows, kOptSrcMajorImages, kOptSrcMajorAdapt, kOptSrcMajorStage, kOptTableStage, kOptTableStageRows,
           kOptTableStageWgs, kOptCount };
struct OptDesc { const char* key; int def, lo, hi; const char* env; };
const OptDesc kOpts[kOptCount] = {
    {"lanemap", -1, -1, 1, "GS360_LANEMAP"},          // -1 auto (per view, by minification), 0 rows, 1 blocked       (env: rows | blocked)
    {"stage", -1, -1, 1, "GS360_STAGE"},              // LDS-staged kernel: -1 auto, 0 never, 1 every call that can
    {"ring", 0, 0, GS360_MAX_VIEWS, nullptr},         // 0 auto; n: at most n views share a coordinate evaluation
    {"xcd_group", -2, -2, 12, nullptr},               // -2 auto; -1 contiguous chunks; g: runs of 2^g tiles
    {"eq_persist", 0, 0, 1 << 20, nullptr},           // grid cap of the cubic equirect kernels (0: one tile per workgroup)
    {"table_persist", -1, -1, 1 << 20, nullptr},      // -1 auto; grid cap of the bicubic table / fisheye kernels
    {"lanczos_table", 0, 0, 1, nullptr},              // 1: read the 128 KiB Lanczos-4 table instead of rebuilding weights per pixel
    {"table_rows", 0, 0, 1, nullptr},                 // 1: table kernel in row form even for tight outputs (A/B of the flat spans)
    {"color_cube", -1, -1, 1, "GS360_COLOR_CUBE"},    // -1 / 1: tabulate the 8-bit colour stage (64 MiB per plan); 0: evaluate per pixel
    {"srcmajor", -1, -1, 1, "GS360_SRCMAJOR"},        // source-major kernel: -1 auto (strongly minified level rings), 0 never, 1 whenever eligible
    {"srcmajor_bx", 768, 256, 4032, nullptr},         // its tile: bytes per box row (multiple of 16) ...
    {"srcmajor_rows", 32, 8, 128, nullptr},           // ... and source rows
    {"srcmajor_images", 0, 0, 12, nullptr},           // images of a tile one workgroup walks (0 auto; must divide twice the ring size)
    {"srcmajor_adapt", 1, 0, 1, nullptr},             // 1: jobs that do not fill the GPU take tiles of half the height; 0: srcmajor_rows as given (probes)
    {"srcmajor_stage", 0, 0, 1, nullptr},             // 0: a loader wavefront copies tiles with global_load_lds; 1: the consumers stage them through registers
    {"table_stage", -1, -1, 1, "GS360_TABLE_STAGE"},  // LDS-staged table kernel (bilinear RGB through map plans): -1 auto, 0 never, 1 every job that can
    {"table_stage_rows", 32, 8, 32, nullptr},         // its output tile: rows (multiple of 8) of 64 pixels
    {"table_stage_wgs", 0, 0, 4, nullptr},            // workgroups per CU (0 auto: what the LDS holds, at most three)
};

struct Staging {  // per-slot device staging used by the *_host conveniences
    void* d_src = nullptr; size_t src_cap = 0;
    void* d_dst = nullptr; size_t dst_cap = 0;
    void* d_aux = nullptr; size_t aux_cap = 0;
    void* d_maskbits = nullptr; size_t maskbits_cap = 0;   // keep-bit images of one masked equirect launch (<= GS360_MAX_FRAMES frames)
};

}  // namespace

struct gs360_ctx {
    int device = 0;
    int n_slots = 0;
    hipStream_t stream[kMaxSlots] = {};
    hipEvent_t event[kMaxSlots][kEventsPerSlot] = {};
    Staging stage[kMaxSlots];
    hipDeviceProp_t prop;
    int16_t* d_cubic = nullptr;   // 32*32*16 int16 cubic weight table, uploaded at context creation
    int16_t* d_lanczos = nullptr; // 32*32*64 int16 Lanczos4 weight table
    float* d_coef1d = nullptr;    // 448 float32 1-D phase coefficients for the 16-bit (float-weight) samplers
    uint32_t* d_lz_cen = nullptr; // 1024 x 2 dwords: the patched block of every Lanczos4 2-D phase (TableLaunch::lz_cen)
    bool lz_rebuild = false;      // the per-pixel weight rebuild reproduces d_lanczos (checked at context creation)
    // Options (gs360_ctx_set_option; seeded ONCE from the environment by gs360_ctx_create for the documented user switches).  The hot
    // path reads these atomics, never the environment: getenv racing a host thread's putenv is undefined behaviour.
    std::atomic<int> opt[kOptCount];
    std::atomic<int> last_sm_stage{0};        // read-only "last_srcmajor_stage": 1 = that launch staged its tiles through registers
    std::atomic<int> last_sm_rows{0}, last_sm_images{0};   // read-only "last_srcmajor_rows" / "last_srcmajor_images": tile rows and images per workgroup of that launch
    std::atomic<int> last_sm_box_pct{0};      // read-only option "last_srcmajor_box_pct": tile-box bytes of the last source-major plan in % of its grid cells
    std::atomic<int> last_eq_kernel{-1};      // read-only option "last_eq_kernel": 0 gather, 1 LDS-staged, 2 source-major (which kernel the last equirect call launched)
    std::atomic<int> last_table_kernel{-1};   // read-only option "last_table_kernel": jobs of the last 8-bit table call that took the LDS-staged kernel (-1 none yet)
    std::atomic<int> last_table_slow{0};      // read-only option "last_table_stage_slow_tiles": tiles of those jobs' stage plans without a box (redone from memory)
    // source-major plans of this context (gs360_srcmajor.hip), most recent calls' geometries
    gs360::SmCache sm;
};

namespace {

int check_ctx_slot(gs360_ctx* ctx, int slot) {
    if (!ctx) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (slot < 0 || slot >= ctx->n_slots) return fail(GS360_ERR_ARG, "slot %d out of range [0,%d)", slot, ctx->n_slots);
    return 0;
}

int ensure(gs360_ctx* ctx, void** p, size_t* cap, size_t need) {
    if (*cap >= need) return 0;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr; *cap = 0;
    size_t want = need + need / 4 + kSlack;
    HIP_TRY(hipMalloc(p, want));
    *cap = want - kSlack;
    (void)ctx;
    return 0;
}

double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// EQ-SPEC v1 per-view constants.  Convention: gs360_GUI.py:377-395 / :419-424 of the reference.
void make_eq_view(const gs360_view& v, int W, bool fisheye_out, int lanemap, EqView* o) {
    double hf = clampd(v.hfov_deg, 1e-3, 179.9) * kPi / 180.0;
    double vf = clampd(v.vfov_deg, 1e-3, 179.9) * kPi / 180.0;
    o->sxu = (float)(std::tan(hf * 0.5) / (double)v.width);
    o->syv = (float)(std::tan(vf * 0.5) / (double)v.height);
    double pitch = v.pitch_deg * kPi / 180.0;
    o->sp = (float)std::sin(pitch);
    o->cp = (float)std::cos(pitch);
    double x0 = (v.yaw_deg / 360.0 + 0.5) * (double)W - 0.5;
    double fl = std::floor(x0);
    o->x0f32 = (float)(32.0 * (x0 - fl));
    long xi = (long)fl % (long)W;
    if (xi < 0) xi += W;
    o->x0i32 = (int32_t)(32 * xi);
    o->out_w = v.width;
    o->out_h = v.height;
    // The kernel computes the left half of every row and mirrors it; level views also mirror top/bottom.
    o->level = (o->sp == 0.0f && o->cp == 1.0f) ? 1 : 0;
    // Lane map (gs360_kernels.hip): source pixels stepped per output pixel at the view centre.  Above ~3 the view
    // bends across so many source rows per 64-pixel output row that compact 4x16 gather patches touch fewer cache
    // lines (cfg2: 4.6 -> blocked, -3 %); below it full rows coalesce better and need less arithmetic (cfg1 1.7,
    // cfg3 2.0, cfg5 1.25: blocked would cost 7-19 %).  GS360_LANEMAP=rows|blocked overrides (tests, probes).
    const double step = (double)W / (2.0 * kPi) * 2.0 * std::tan(hf * 0.5) / (double)v.width;
    o->blocked = step >= 3.0 ? 1 : 0;
    if (lanemap >= 0) o->blocked = lanemap;           // option "lanemap": tests, probes
    o->fish = 0;
    if (fisheye_out) {   // image-plane radius 1 <-> 90 degrees off axis; hfov/vfov = full field of view of the fisheye image
        o->fish = 1;
        o->sxu = (float)(clampd(v.hfov_deg, 1e-3, 360.0) / 180.0 / (double)v.width);
        o->syv = (float)(clampd(v.vfov_deg, 1e-3, 360.0) / 180.0 / (double)v.height);
        o->level = 0;
        o->blocked = 0;
        o->tiles_y = (v.height + kTileH - 1) / kTileH;
    }
    const int half_w = (v.width + 1) / 2;
    o->tiles_x = (half_w + kTileW - 1) / kTileW;
    o->tiles_y = o->level ? ((v.height + 1) / 2 + kTileH / 2 - 1) / (kTileH / 2) : (v.height + kTileH - 1) / kTileH;
}

void make_fe_view(const gs360_calib& cal, const gs360_view& v, double lens_fov_deg, FeView* o) {
    double hf = clampd(v.hfov_deg, 1e-3, 179.9) * kPi / 180.0;
    double vf = clampd(v.vfov_deg, 1e-3, 179.9) * kPi / 180.0;
    o->sxu = (float)(std::tan(hf * 0.5) / (double)v.width);
    o->syv = (float)(std::tan(vf * 0.5) / (double)v.height);
    double pitch = v.pitch_deg * kPi / 180.0, yaw = v.yaw_deg * kPi / 180.0;
    o->sp = (float)std::sin(pitch); o->cp = (float)std::cos(pitch);
    o->sy = (float)std::sin(yaw); o->cy = (float)std::cos(yaw);
    o->k1 = (float)cal.k1; o->k2 = (float)cal.k2; o->k3 = (float)cal.k3; o->k4 = (float)cal.k4;
    o->p1 = (float)cal.p1; o->p2 = (float)cal.p2;
    o->tp1 = (float)(2.0 * cal.p1); o->tp2 = (float)(2.0 * cal.p2);
    o->b1 = (float)cal.b1; o->b2 = (float)cal.b2; o->f = (float)cal.f;
    o->cx0 = (float)((cal.width * 0.5) + cal.cx);   // DF:1812-1813
    o->cy0 = (float)((cal.height * 0.5) + cal.cy);
    o->wmax = (float)(cal.width - 1); o->hmax = (float)(cal.height - 1);
    o->cos_tmax = (float)std::cos(clampd(lens_fov_deg, 1.0, 360.0) * 0.5 * kPi / 180.0);  // DF:1800
    o->tang = (cal.p1 != 0.0 || cal.p2 != 0.0) ? 1 : 0;
    o->W = cal.width; o->H = cal.height;
    o->out_w = v.width; o->out_h = v.height;
    o->tiles_x = (v.width + kTileW - 1) / kTileW;
    o->tiles_y = (v.height + kTileH - 1) / kTileH;
}

}  // namespace

// OpenCV imgproc initInterTab2D(fixed point), restated: per-phase 1-D coefficients in float32, outer product scaled
// by 2^15 and rounded to short, then the entries are patched so each ks x ks kernel sums to 2^15 (the patch goes to
// the largest / smallest entry of rows/cols ks/2 .. ks/2+1, the block OpenCV inspects).
namespace {
void build_tab2d(const float* c1, int ks, int16_t* out) {
    const int h = ks / 2;
    for (int fy = 0; fy < 32; ++fy)
        for (int fx = 0; fx < 32; ++fx) {
            int16_t* k = out + (fy * 32 + fx) * ks * ks;
            int sum = 0;
            for (int a = 0; a < ks; ++a)
                for (int b = 0; b < ks; ++b) {
                    long r = std::lrintf(c1[fy * ks + a] * c1[fx * ks + b] * 32768.0f);
                    r = r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
                    k[a * ks + b] = (int16_t)r;
                    sum += (int)r;
                }
            if (sum != 32768) {
                int hi = h * ks + h, lo = hi;
                for (int a = h; a < h + 2; ++a)
                    for (int b = h; b < h + 2; ++b) {
                        const int idx = a * ks + b;
                        if (k[idx] < k[lo]) lo = idx;
                        else if (k[idx] > k[hi]) hi = idx;
                    }
                const int diff = sum - 32768;
                if (diff < 0) k[hi] = (int16_t)(k[hi] - diff);
                else k[lo] = (int16_t)(k[lo] - diff);
            }
        }
}
}  // namespace

namespace {
void cubic_coef1d(float* c1) {   // Keys kernel, A = -0.75: 32 phases x 4 taps
    const float A = -0.75f;
    for (int i = 0; i < 32; ++i) {
        const float x = (float)i * (1.0f / 32.0f);
        float* c = c1 + i * 4;
        c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
        c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
        c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
        c[3] = 1.f - c[0] - c[1] - c[2];
    }
}

void lanczos4_coef1d(float* c1) {   // OpenCV interpolateLanczos4: taps -3..+4, one sin/cos pair per phase; 32 phases x 8 taps
    static const double r = 0.70710678118654752440084436210485;
    static const double rot[8][2] = {{1, 0}, {-r, -r}, {0, 1}, {r, -r}, {-1, 0}, {r, r}, {0, -1}, {-r, r}};
    for (int i = 0; i < 32; ++i) {
        const float x = (float)i * (1.0f / 32.0f);
        float* c = c1 + i * 8;
        if (x < 1.1920928955078125e-07f) {
            for (int t = 0; t < 8; ++t) c[t] = (t == 3) ? 1.f : 0.f;
            continue;
        }
        const double a0 = -(x + 3) * kPi * 0.25, s0 = std::sin(a0), c0 = std::cos(a0);
        float sum = 0.f;
        for (int t = 0; t < 8; ++t) {
            const double a = -(x + 3 - t) * kPi * 0.25;
            c[t] = (float)((rot[t][0] * s0 + rot[t][1] * c0) / (a * a));
            sum += c[t];
        }
        sum = 1.f / sum;
        for (int t = 0; t < 8; ++t) c[t] *= sum;
    }
}
}  // namespace

void gs360::build_cubic_table(int16_t* out) {
    float c1[32 * 4];
    cubic_coef1d(c1);
    build_tab2d(c1, 4, out);
}

void gs360::build_lanczos4_table(int16_t* out) {
    float c1[32 * 8];
    lanczos4_coef1d(c1);
    build_tab2d(c1, 8, out);
}

// float32 1-D phase tables of the CV_16U samplers: [0,64) linear (1-x, x), [64,192) cubic, [192,448) lanczos4
void gs360::build_coef1d(float* out) {
    for (int i = 0; i < 32; ++i) {
        const float x = (float)i * (1.0f / 32.0f);
        out[i * 2] = 1.f - x;
        out[i * 2 + 1] = x;
    }
    cubic_coef1d(out + 64);
    lanczos4_coef1d(out + 192);
}

namespace {

uint8_t sat_u8(double v) {  // cv::saturate_cast<uchar>(double)
    long r = std::lrint(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

}  // namespace

extern "C" {

int gs360_abi_version(void) { return GS360_ABI_VERSION; }

int gs360_last_error(char* buf, size_t n) {
    size_t len = std::strlen(g_err);
    if (buf && n) {
        size_t c = len < n - 1 ? len : n - 1;
        std::memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return (int)len;
}

int gs360_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        fail(GS360_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
        return 0;
    }
    return n;
}

int gs360_ctx_create(int device, int n_slots, gs360_ctx** out) {
    if (!out) return fail(GS360_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n_slots < 1 || n_slots > kMaxSlots) return fail(GS360_ERR_ARG, "n_slots must be in [1,%d]", kMaxSlots);
    int n = gs360_device_count();
    if (n <= 0) return fail(GS360_ERR_NODEV, "no HIP device visible (libgs360hip has no CPU path)");
    if (device < 0 || device >= n) return fail(GS360_ERR_ARG, "device %d out of range [0,%d)", device, n);
    gs360_ctx* c = new (std::nothrow) gs360_ctx();
    if (!c) return fail(GS360_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->n_slots = n_slots;
    for (int k = 0; k < kOptCount; ++k) {
        int v = kOpts[k].def;
        if (kOpts[k].env)
            if (const char* e = std::getenv(kOpts[k].env)) {       // the ONLY place the library reads its switches from the environment
                if (k == kOptLanemap) v = !std::strcmp(e, "rows") ? 0 : (!std::strcmp(e, "blocked") ? 1 : -1);
                else v = std::atoi(e) != 0 ? 1 : 0;
            }
        c->opt[k].store(v, std::memory_order_relaxed);
    }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&c->prop, device);
    for (int s = 0; s < n_slots && e == hipSuccess; ++s) {
        e = hipStreamCreateWithFlags(&c->stream[s], hipStreamNonBlocking);
        for (int k = 0; k < kEventsPerSlot && e == hipSuccess; ++k) e = hipEventCreate(&c->event[s][k]);
    }
    if (e != hipSuccess) {
        int rc = fail(GS360_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
        gs360_ctx_destroy(c);
        return rc;
    }
    if (std::strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(GS360_ERR_NODEV, "device %d is %s; this library carries gfx950 code objects only", device,
                      c->prop.gcnArchName);
        gs360_ctx_destroy(c);
        return rc;
    }
    {
        std::vector<int16_t> tab(32 * 32 * 16);
        build_cubic_table(tab.data());
        e = hipMalloc((void**)&c->d_cubic, tab.size() * sizeof(int16_t));
        if (e == hipSuccess) e = hipMemcpy(c->d_cubic, tab.data(), tab.size() * sizeof(int16_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            tab.assign(32 * 32 * 64, 0);
            build_lanczos4_table(tab.data());
            e = hipMalloc((void**)&c->d_lanczos, tab.size() * sizeof(int16_t));
            if (e == hipSuccess) e = hipMemcpy(c->d_lanczos, tab.data(), tab.size() * sizeof(int16_t), hipMemcpyHostToDevice);
            // the Lanczos kernel rebuilds the 2-D weights per pixel from the 1-D table: w = low 16 bits of the float
            // (cy * (cx * 2^15)) + 1.5 * 2^23 (round-to-nearest-even into the mantissa).  That reproduces every table entry except
            // the block [4,5] x [4,5] the sum fix-up patches (shipped per phase: `cen`) and the one saturated entry of phase 0
            // (handled in the kernel) -- verified here for all 1024 phases; on any mismatch the kernel keeps reading the table.
            std::vector<uint32_t> cen(1024 * 2);
            float c1[32 * 8];
            lanczos4_coef1d(c1);
            bool rebuilt_ok = true;
            for (int p = 0; p < 1024; ++p) {
                const int16_t* k = tab.data() + p * 64;
                cen[2 * p] = (uint32_t)(uint16_t)k[4 * 8 + 4] | ((uint32_t)(uint16_t)k[4 * 8 + 5] << 16);
                cen[2 * p + 1] = (uint32_t)(uint16_t)k[5 * 8 + 4] | ((uint32_t)(uint16_t)k[5 * 8 + 5] << 16);
                const float* cy = c1 + (p >> 5) * 8;
                const float* cx = c1 + (p & 31) * 8;
                for (int a = 0; a < 8; ++a)
                    for (int b = 0; b < 8; ++b) {
                        if ((a == 4 || a == 5) && (b == 4 || b == 5)) continue;
                        volatile float cx32 = cx[b] * 32768.0f;
                        volatile float m = cy[a] * cx32;
                        volatile float t = m + 12582912.0f;
                        const float tf = t;
                        uint32_t bits;
                        std::memcpy(&bits, &tf, 4);
                        int16_t w = (int16_t)(uint16_t)(bits & 0xffffu);
                        if (p == 0 && a == 3 && b == 3) w = 32767;          // the kernel's phase-0 rule
                        if (w != k[a * 8 + b]) rebuilt_ok = false;
                    }
            }
            c->lz_rebuild = rebuilt_ok;
            if (e == hipSuccess) e = hipMalloc((void**)&c->d_lz_cen, cen.size() * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpy(c->d_lz_cen, cen.data(), cen.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            float coef[448];
            build_coef1d(coef);
            e = hipMalloc((void**)&c->d_coef1d, sizeof(coef));
            if (e == hipSuccess) e = hipMemcpy(c->d_coef1d, coef, sizeof(coef), hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            int rc = fail(GS360_ERR_HIP, "interpolation table upload failed: %s", hipGetErrorString(e));
            gs360_ctx_destroy(c);
            return rc;
        }
    }
    *out = c;
    return GS360_OK;
}

int gs360_ctx_destroy(gs360_ctx* c) {
    if (!c) return GS360_OK;
    (void)hipSetDevice(c->device);
    for (int s = 0; s < c->n_slots; ++s) {
        if (c->stream[s]) (void)hipStreamSynchronize(c->stream[s]);
        for (int k = 0; k < kEventsPerSlot; ++k)
            if (c->event[s][k]) (void)hipEventDestroy(c->event[s][k]);
        if (c->stage[s].d_src) (void)hipFree(c->stage[s].d_src);
        if (c->stage[s].d_dst) (void)hipFree(c->stage[s].d_dst);
        if (c->stage[s].d_aux) (void)hipFree(c->stage[s].d_aux);
        if (c->stage[s].d_maskbits) (void)hipFree(c->stage[s].d_maskbits);
        if (c->stream[s]) (void)hipStreamDestroy(c->stream[s]);
    }
    if (c->d_cubic) (void)hipFree(c->d_cubic);
    if (c->d_lanczos) (void)hipFree(c->d_lanczos);
    if (c->d_coef1d) (void)hipFree(c->d_coef1d);
    if (c->d_lz_cen) (void)hipFree(c->d_lz_cen);
    gs360::sm_cache_destroy(c->sm);
    delete c;
    return GS360_OK;
}

int gs360_ctx_set_option(gs360_ctx* c, const char* key, int value) {
    if (!c || !key) return fail(GS360_ERR_ARG, "NULL argument");
    for (int k = 0; k < kOptCount; ++k)
        if (!std::strcmp(key, kOpts[k].key)) {
            if (value < kOpts[k].lo || value > kOpts[k].hi)
                return fail(GS360_ERR_ARG, "option %s: %d outside [%d, %d]", key, value, kOpts[k].lo, kOpts[k].hi);
            if (k == kOptSrcMajorBx && value % 16) return fail(GS360_ERR_ARG, "option srcmajor_bx must be a multiple of 16");
            if (k == kOptTableStageRows && value % 8) return fail(GS360_ERR_ARG, "option table_stage_rows must be a multiple of 8");
            c->opt[k].store(value, std::memory_order_relaxed);
            return GS360_OK;
        }
    return fail(GS360_ERR_ARG, "unknown option '%s'", key);
}

int gs360_ctx_get_option(gs360_ctx* c, const char* key, int* value) {
    if (!c || !key || !value) return fail(GS360_ERR_ARG, "NULL argument");
    if (!std::strcmp(key, "last_eq_kernel")) {
        *value = c->last_eq_kernel.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    if (!std::strcmp(key, "last_table_kernel")) {
        *value = c->last_table_kernel.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    if (!std::strcmp(key, "last_table_stage_slow_tiles")) {
        *value = c->last_table_slow.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    if (!std::strcmp(key, "srcmajor_plan_build_us")) {
        std::lock_guard<std::mutex> lock(c->sm.mu);
        *value = (int)std::min<uint64_t>(c->sm.build_us, 0x7fffffffu);
        return GS360_OK;
    }
    if (!std::strcmp(key, "srcmajor_plan_builds") || !std::strcmp(key, "srcmajor_inline_frees") || !std::strcmp(key, "srcmajor_plans")) {
        std::lock_guard<std::mutex> lock(c->sm.mu);
        *value = !std::strcmp(key, "srcmajor_plan_builds") ? (int)c->sm.builds : (!std::strcmp(key, "srcmajor_inline_frees") ? (int)c->sm.inline_frees : (int)c->sm.plans.size());
        return GS360_OK;
    }
    if (!std::strcmp(key, "last_srcmajor_box_pct")) {
        *value = c->last_sm_box_pct.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    if (!std::strcmp(key, "last_srcmajor_stage")) {
        *value = c->last_sm_stage.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    if (!std::strcmp(key, "last_srcmajor_rows")) {
        *value = c->last_sm_rows.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    if (!std::strcmp(key, "last_srcmajor_images")) {
        *value = c->last_sm_images.load(std::memory_order_relaxed);
        return GS360_OK;
    }
    for (int k = 0; k < kOptCount; ++k)
        if (!std::strcmp(key, kOpts[k].key)) {
            *value = c->opt[k].load(std::memory_order_relaxed);
            return GS360_OK;
        }
    return fail(GS360_ERR_ARG, "unknown option '%s'", key);
}

int gs360_device_pci_bus_id(gs360_ctx* c, char* buf, size_t n) {
    if (!c || !buf || n < 16) return fail(GS360_ERR_ARG, "NULL argument or buffer shorter than 16 bytes");
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)n, c->device));
    return GS360_OK;
}

int gs360_device_info(gs360_ctx* c, char* name, size_t n, int32_t* cu_count, uint64_t* hbm_bytes) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (name && n) {
        // hipDeviceProp_t::name comes back empty on some driver stacks (the MI355X pool's): the amdgpu driver's product_name then
        char product[128] = "";
        std::snprintf(product, sizeof(product), "%s", c->prop.name);
        if (!product[0]) {
            char bus[32] = "", path[96];
            if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c->device) == hipSuccess) {
                for (char* q = bus; *q; ++q) *q = (char)std::tolower((unsigned char)*q);
                std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/product_name", bus);
                if (FILE* f = std::fopen(path, "r")) {
                    if (std::fgets(product, (int)sizeof(product), f)) product[std::strcspn(product, "\r\n")] = 0;
                    std::fclose(f);
                }
            }
            (void)hipGetLastError();
        }
        snprintf(name, n, "%s (%s)", product[0] ? product : "AMD GPU", c->prop.gcnArchName);
    }
    if (cu_count) *cu_count = c->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)c->prop.totalGlobalMem;
    return GS360_OK;
}

// ---- memory ------------------------------------------------------------------------------------
int gs360_dev_alloc(gs360_ctx* c, size_t bytes, void** dptr) {
    if (!c || !dptr) return fail(GS360_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const hipError_t e = hipMalloc(dptr, bytes + kSlack);
    if (e == hipErrorOutOfMemory) {               // its own code: a streaming caller (gs360/video.py) retires frames and tries again
        (void)hipGetLastError();
        return fail(GS360_ERR_NOMEM, "out of device memory (%zu bytes)", bytes);
    }
    HIP_TRY(e);
    return GS360_OK;
}
int gs360_dev_free(gs360_ctx* c, void* dptr) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!dptr) return GS360_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipFree(dptr));
    return GS360_OK;
}
int gs360_host_alloc(gs360_ctx* c, size_t bytes, void** hptr) {
    if (!c || !hptr) return fail(GS360_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return GS360_OK;
}
int gs360_host_free(gs360_ctx* c, void* hptr) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!hptr) return GS360_OK;
    HIP_TRY(hipHostFree(hptr));
    return GS360_OK;
}
int gs360_upload(gs360_ctx* c, void* dst_dev, const void* src_host, size_t bytes, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!dst_dev || !src_host) return fail(GS360_ERR_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream[slot]));
    return GS360_OK;
}
int gs360_download(gs360_ctx* c, void* dst_host, const void* src_dev, size_t bytes, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!dst_host || !src_dev) return fail(GS360_ERR_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream[slot]));
    return GS360_OK;
}
int gs360_dev_memset(gs360_ctx* c, void* dst_dev, int value, size_t bytes, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!dst_dev) return fail(GS360_ERR_ARG, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemsetAsync(dst_dev, value, bytes, c->stream[slot]));
    return GS360_OK;
}
int gs360_dev_bswap16(gs360_ctx* c, void* buf_dev, size_t n_samples, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!buf_dev || ((uintptr_t)buf_dev & 1)) return fail(GS360_ERR_ARG, "NULL or odd buffer address");
    if (n_samples == 0) return GS360_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_bswap16((uint16_t*)buf_dev, n_samples, c->stream[slot]));
    return GS360_OK;
}
int gs360_sync(gs360_ctx* c, int slot) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (slot < 0) {
        for (int s = 0; s < c->n_slots; ++s) HIP_TRY(hipStreamSynchronize(c->stream[s]));
        gs360::sm_cache_drain(c->sm);             // every stream is idle: plans the cache evicted since the last time are released here
        return GS360_OK;
    }
    if (int rc = check_ctx_slot(c, slot)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream[slot]));
    return GS360_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int gs360_event_record(gs360_ctx* c, int slot, int idx) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (idx < 0 || idx >= kEventsPerSlot) return fail(GS360_ERR_ARG, "event index %d out of range", idx);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->event[slot][idx], c->stream[slot]));
    return GS360_OK;
}
int gs360_event_elapsed_ms(gs360_ctx* c, int slot, int from, int to, float* ms) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!ms || from < 0 || to < 0 || from >= kEventsPerSlot || to >= kEventsPerSlot)
        return fail(GS360_ERR_ARG, "bad event arguments");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->event[slot][to]));
    HIP_TRY(hipEventElapsedTime(ms, c->event[slot][from], c->event[slot][to]));
    return GS360_OK;
}

int gs360_event_sync(gs360_ctx* c, int slot, int idx) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (idx < 0 || idx >= kEventsPerSlot) return fail(GS360_ERR_ARG, "bad event index");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->event[slot][idx]));
    return GS360_OK;
}
int gs360_stream_wait_event(gs360_ctx* c, int waiting_slot, int event_slot, int idx) {
    if (int rc = check_ctx_slot(c, waiting_slot)) return rc;
    if (int rc = check_ctx_slot(c, event_slot)) return rc;
    if (idx < 0 || idx >= kEventsPerSlot) return fail(GS360_ERR_ARG, "bad event index");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamWaitEvent(c->stream[waiting_slot], c->event[event_slot][idx], 0));
    return GS360_OK;
}

// Self-test: the kernels replace `/` and sqrtf by shorter instruction sequences that are bit-identical on the operand domains
// of EQ-SPEC / FE-SPEC (gs360_eqspec.h).  This runs both forms on `n_millions` x 10^6 pseudo-random operand sets per form.
int gs360_selftest_arith(gs360_ctx* c, uint32_t seed, int n_millions, uint64_t* n_checked, uint64_t* n_mismatch) {
    if (int rc = check_ctx_slot(c, 0)) return rc;
    if (!n_checked || !n_mismatch || n_millions < 1 || n_millions > 100000) return fail(GS360_ERR_ARG, "bad self-test arguments");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long* d_bad = nullptr;
    HIP_TRY(hipMalloc((void**)&d_bad, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(unsigned long long), c->stream[0]);
    const int iters = 1000, blocks = (int)(((long long)n_millions * 1000000 + 256LL * iters - 1) / (256LL * iters));
    if (e == hipSuccess) e = launch_arith_selftest(seed, blocks, iters, d_bad, c->stream[0]);
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, c->stream[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream[0]);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(GS360_ERR_HIP, "arithmetic self-test failed to run: %s", hipGetErrorString(e));
    *n_checked = (uint64_t)blocks * 256ull * (uint64_t)iters;
    *n_mismatch = bad;
    return GS360_OK;
}

// ---- equirect -> views -------------------------------------------------------------------------
namespace {
int equirect_views_impl(gs360_ctx* c, const void* const* src_frames, const void* const* mask_frames, int n_frames,
                        int W, int H, int C, size_t src_stride, size_t mask_stride, const gs360_view* views,
                        int n_views, void* const* dst, size_t dst_stride, int interp, uint32_t flags, int slot, int esize);
}

int gs360_equirect_views_u8(gs360_ctx* c, const void* const* src_frames, int n_frames, int W, int H, int C,
                            size_t src_stride, const gs360_view* views, int n_views, void* const* dst,
                            size_t dst_stride, int interp, uint32_t flags, int slot) {
    return equirect_views_impl(c, src_frames, nullptr, n_frames, W, H, C, src_stride, 0, views, n_views, dst,
                               dst_stride, interp, flags, slot, 1);
}

int gs360_equirect_views_u16(gs360_ctx* c, const void* const* src_frames, int n_frames, int W, int H, int C,
                             size_t src_stride, const gs360_view* views, int n_views, void* const* dst,
                             size_t dst_stride, int interp, uint32_t flags, int slot) {
    return equirect_views_impl(c, src_frames, nullptr, n_frames, W, H, C, src_stride, 0, views, n_views, dst,
                               dst_stride, interp, flags, slot, 2);
}

int gs360_equirect_views_masked_u8(gs360_ctx* c, const void* const* src_frames, const void* const* mask_frames, int n_frames,
                                   int W, int H, int C, size_t src_stride, size_t mask_stride, const gs360_view* views,
                                   int n_views, void* const* dst, size_t dst_stride, int interp, uint32_t flags, int slot) {
    return equirect_views_impl(c, src_frames, mask_frames, n_frames, W, H, C, src_stride, mask_stride, views, n_views, dst,
                               dst_stride, interp, flags, slot, 1);
}

namespace {
int equirect_views_impl(gs360_ctx* c, const void* const* src_frames, const void* const* mask_frames, int n_frames,
                        int W, int H, int C, size_t src_stride, size_t mask_stride, const gs360_view* views,
                        int n_views, void* const* dst, size_t dst_stride, int interp, uint32_t flags, int slot, int esize) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!src_frames || !views || !dst) return fail(GS360_ERR_ARG, "NULL argument");
    if (mask_frames) {
        if (mask_stride == 0) mask_stride = (size_t)W;
        if (mask_stride < (size_t)W) return fail(GS360_ERR_ARG, "mask_stride smaller than a row");
        if ((uint64_t)mask_stride * (uint64_t)H >= ((uint64_t)1 << 32)) return fail(GS360_ERR_UNSUPPORTED, "mask too large");
        if (H + 1 > 65535) return fail(GS360_ERR_UNSUPPORTED, "masked equirect calls take H < 65535 (the mask pack pass launches one grid row per mask row)");
        for (int f = 0; f < n_frames; ++f)
            if (!mask_frames[f]) return fail(GS360_ERR_ARG, "mask_frames[%d] is NULL", f);
    }
    if (n_frames < 0 || n_views < 0) return fail(GS360_ERR_ARG, "negative count");
    if (n_frames == 0 || n_views == 0) return GS360_OK;  // empty batch is a no-op
    if (C != 1 && C != 3 && C != 4) return fail(GS360_ERR_ARG, "C must be 1, 3 or 4 (got %d)", C);
    if (W < 8 || H < 2 || W > (1 << 21) || H > (1 << 21))
        return fail(GS360_ERR_ARG, "bad source size %dx%d (an equirect frame is at least 8 texels wide)", W, H);
    // the kernels form a flipped ring member's latitude with v_mad_i32_i24 (24-bit operands): 32 H must stay below 2^23
    if (H >= (1 << 18)) return fail(GS360_ERR_UNSUPPORTED, "source height %d: the equirect kernels take H < 262144", H);
    if (interp != GS360_INTERP_LINEAR && interp != GS360_INTERP_CUBIC)
        return fail(GS360_ERR_UNSUPPORTED, "equirect path implements INTER_LINEAR (1) and INTER_CUBIC (2), got %d", interp);
    if (flags & ~(uint32_t)GS360_EQ_FISHEYE_OUT) return fail(GS360_ERR_ARG, "unknown flags 0x%x", flags);
    if (src_stride == 0) src_stride = (size_t)W * C * esize;
    if (src_stride < (size_t)W * C * esize) return fail(GS360_ERR_ARG, "src_stride smaller than a row");
    if (esize == 2 && ((src_stride | dst_stride) & 1)) return fail(GS360_ERR_ARG, "16-bit images need even strides");
    if (src_stride >= ((size_t)1 << 24) || (uint64_t)src_stride * (uint64_t)H >= ((uint64_t)1 << 32))
        return fail(GS360_ERR_UNSUPPORTED, "frame too large for 32-bit tap offsets (stride %zu x %d rows)", src_stride, H);
    for (int k = 0; k < n_views; ++k) {
        if (views[k].width < 1 || views[k].height < 1 || views[k].width > 32768 || views[k].height > 32768)
            return fail(GS360_ERR_ARG, "view %d has bad size %dx%d", k, views[k].width, views[k].height);
        if (dst_stride && dst_stride < (size_t)views[k].width * C * esize) return fail(GS360_ERR_ARG, "dst_stride smaller than a row");
        if (!std::isfinite(views[k].yaw_deg) || !std::isfinite(views[k].pitch_deg) || !std::isfinite(views[k].hfov_deg) ||
            !std::isfinite(views[k].vfov_deg))
            return fail(GS360_ERR_ARG, "view %d has a non-finite angle", k);
    }
    for (int f = 0; f < n_frames; ++f) {
        if (!src_frames[f]) return fail(GS360_ERR_ARG, "src_frames[%d] is NULL", f);
        if (C == 3 && esize == 1 && ((uintptr_t)src_frames[f] & 3))
            return fail(GS360_ERR_ARG, "src_frames[%d] must be 4-byte aligned (the RGB tap reads are dword-aligned)", f);
    }
    for (int i = 0; i < n_frames * n_views; ++i)
        if (!dst[i]) return fail(GS360_ERR_ARG, "dst[%d] is NULL", i);
    HIP_TRY(hipSetDevice(c->device));

    // ---- yaw rings -----------------------------------------------------------------------------------------------
    // Views whose EQ-SPEC constants agree in everything but the integer longitude offset x0i32 -- and possibly the sign of the
    // pitch -- form a ring: the kernel evaluates a tile's coordinates once and samples it for every member (the presets'
    // yaw steps are whole texels: `yaw = i * 360 / count`, PC:794).  Float equality of the rounded constants is the criterion,
    // so the grouping can never change a result.
    static_assert(sizeof(EqLaunch) <= 4096, "EqLaunch travels as a kernel argument");
    const bool fish = (flags & GS360_EQ_FISHEYE_OUT) != 0;
    const int opt_lanemap = c->opt[kOptLanemap].load(std::memory_order_relaxed), opt_stage = c->opt[kOptStage].load(std::memory_order_relaxed);
    const int opt_ring = c->opt[kOptRing].load(std::memory_order_relaxed), opt_xcd = c->opt[kOptXcdGroup].load(std::memory_order_relaxed);
    const int opt_srcmajor = c->opt[kOptSrcMajor].load(std::memory_order_relaxed);
    try {
    std::vector<EqView> ev((size_t)n_views);
    for (int k = 0; k < n_views; ++k) {
        make_eq_view(views[k], W, fish, opt_lanemap, &ev[k]);
        ev[k].flip = 0;
        if (esize == 2) ev[k].blocked = 0;   // 16-bit samples: row-per-slot lane map only
    }
    // Source-major kernel (gs360_srcmajor.hip): a call whose views are yaw rings of one size filling their circle (`--count N`, PC:794; the
    // presets' pitched ring pairs, PC:616-680) streams every source tile once for all views instead of gathering per view.  Where it wins
    // (8K sources, N views per ring, s source texels per output pixel):
    //   * ONE level ring (profiles/r05/srcmajor_ring_sweep.txt, srcmajor_small_jobs.txt): N >= 6 at every s measured (1.5 .. 4.6) and every
    //     number of frames per call -- sixteen frames -8 .. -42 % (cfg2 19.0 -> 14.9 us per frame, cfg1 47.9 -> 33.6), one frame -3 .. -26 %
    //     (cfg2 21.7 -> 16.0: what the drop-in engine launches) -- as long as the call has work to fill the GPU (>= 3.5 M output pixels; a
    //     4K -> 6 x 400^2 frame is launch-bound either way); N = 5 from s = 2.25 and two frames; N = 4 never (neighbours overlap by a
    //     quarter of their field only: +14 .. +44 %);
    //   * SEVERAL rings (srcmajor_family_sweep.txt): from four frames per call, eight views and s = 1.75, unless the views reach so close to
    //     a pole that the tile boxes outgrow their grid cells (kSmMaxBoxPct).
    // Option "srcmajor": 0 never, 1 whenever the geometry fits (tests, probes).  Decided before the ring grouping below (which keeps blocked
    // views apart); a geometry that does not fit the plan format falls through to the gather kernels.
    // keep-masks: thresholded once per launch into bit images (the kernels only test `< 128`), behind the caller's upload on the launch
    // stream: a streaming pass over W x H bytes per frame, ~7 us for an 8K mask.  (The previous launch on this stream may still read the
    // images: a reallocation's hipFree synchronises the device.)
    const int mask_pitch_dw = (W + 1 + 31) / 32;
    const size_t mask_bits_bytes = (size_t)mask_pitch_dw * 4 * (size_t)(H + 1);
    auto pack_masks = [&](int f0, int nf) -> int {
        Staging& st = c->stage[slot];
        if (int rc = ensure(c, &st.d_maskbits, &st.maskbits_cap, mask_bits_bytes * (size_t)nf)) return rc;
        MaskPack P;
        std::memset(&P, 0, sizeof(P));
        for (int f = 0; f < nf; ++f) {
            P.src[f] = (const uint8_t*)mask_frames[f0 + f];
            P.dst[f] = (uint32_t*)((uint8_t*)st.d_maskbits + mask_bits_bytes * (size_t)f);
        }
        P.W = W; P.H = H; P.pitch_dw = mask_pitch_dw; P.n = nf;
        P.stride = (int64_t)mask_stride;
        HIP_TRY(launch_mask_pack(P, c->stream[slot]));
        return GS360_OK;
    };
    if (opt_srcmajor != 0 && esize == 1 && C == 3 && interp == GS360_INTERP_LINEAR && !fish && n_views >= 2 && n_views <= GS360_MAX_VIEWS) {
        bool ring = true;
        SmShape shape;
        std::vector<EqLaunch> Ls;
        for (int f0 = 0; f0 < n_frames && ring; f0 += GS360_MAX_FRAMES) {
            const int nf = n_frames - f0 < GS360_MAX_FRAMES ? n_frames - f0 : GS360_MAX_FRAMES;
            EqLaunch L;
            std::memset(&L, 0, sizeof(L));
            for (int k = 0; k < n_views; ++k) L.view[k] = ev[k];
            L.n_rings = 1; L.ring_first[0] = 0; L.ring_count[0] = n_views;
            for (int f = 0; f < nf; ++f) {
                L.src[f] = (const uint8_t*)src_frames[f0 + f];
                for (int k = 0; k < n_views; ++k) L.dst[f * n_views + k] = (uint8_t*)dst[(size_t)(f0 + f) * n_views + k];
            }
            L.kx32 = (float)(32.0 * (double)W / (2.0 * kPi));
            L.ky32 = (float)(32.0 * (double)H / kPi);
            L.W = W; L.H = H; L.y0i32 = 16 * H - 16;
            L.n_views = n_views; L.n_frames = nf;
            L.src_stride = (int64_t)src_stride; L.dst_stride = (int64_t)dst_stride;
            ring = sm_eligible(L, C, esize, interp, mask_frames != nullptr, &shape);     // (the shape depends on the views only: the same for every chunk)
            Ls.push_back(L);
        }
        if (ring && opt_srcmajor < 0) {
            const double hf = clampd(views[0].hfov_deg, 1e-3, 179.9) * kPi / 180.0;
            const double step = (double)W / (2.0 * kPi) * 2.0 * std::tan(hf * 0.5) / (double)views[0].width;
            const double out_px = (double)n_frames * n_views * views[0].width * views[0].height;
            if (shape.n_rings == 1) ring = out_px >= kSmMinPixels && (shape.N >= 6 ? step >= 1.5 : shape.N == 5 && n_frames >= 2 && step >= 2.25);
            else ring = n_frames >= kSmFamilyMinFrames && n_views >= 8 && step >= 1.75;
        }
        // the plan is decided ONCE per call (first chunk) and held until the last chunk is launched: a tail chunk of another size must not pick
        // another plan -- or find its plan evicted -- after earlier chunks have rendered
        gs360::SmPlan* plan = nullptr;
        if (ring) {
            hipError_t he = hipSuccess;
            int seen_box_pct = 0;
            const int rc = sm_prepare(Ls[0], shape, c->sm, mask_frames != nullptr, c->opt[kOptSrcMajorBx].load(std::memory_order_relaxed), c->opt[kOptSrcMajorRows].load(std::memory_order_relaxed),
                                      c->opt[kOptSrcMajorImages].load(std::memory_order_relaxed), c->opt[kOptSrcMajorAdapt].load(std::memory_order_relaxed) != 0,
                                      opt_srcmajor < 0 ? kSmMaxBoxPct : 0, kSmLdsPerGroup, c->prop.multiProcessorCount, c->stream[slot], &he, &plan, &seen_box_pct);
            c->last_sm_box_pct.store(seen_box_pct, std::memory_order_relaxed);
            if (rc < 0) return fail(he == hipErrorOutOfMemory ? GS360_ERR_NOMEM : GS360_ERR_HIP, "source-major plan failed: %s", hipGetErrorString(he));
            if (rc == 1) ring = false;                   // the geometry does not fit the plan format: the gather kernels (nothing launched yet)
        }
        for (size_t i = 0; i < Ls.size() && ring; ++i) {
            hipError_t he = hipSuccess;
            int info[4] = {0, 0, 0, 0};
            if (mask_frames) {                           // (packed per chunk of frames: the staging images are reused)
                if (int prc = pack_masks((int)i * GS360_MAX_FRAMES, Ls[i].n_frames)) { sm_release(c->sm, plan); return prc; }
                for (int f = 0; f < Ls[i].n_frames; ++f) Ls[i].mask[f] = (const uint8_t*)c->stage[slot].d_maskbits + mask_bits_bytes * (size_t)f;
                Ls[i].mask_stride = (int64_t)mask_pitch_dw * 4;
            }
            const int rc = sm_launch(Ls[i], shape, plan, c->opt[kOptSrcMajorImages].load(std::memory_order_relaxed),
                                     c->opt[kOptSrcMajorStage].load(std::memory_order_relaxed) != 0, kSmLdsPerGroup, c->prop.multiProcessorCount,
                                     c->stream[slot], &he, info);
            c->last_sm_box_pct.store(info[0], std::memory_order_relaxed);
            c->last_sm_rows.store(info[1], std::memory_order_relaxed);
            c->last_sm_images.store(info[2], std::memory_order_relaxed);
            c->last_sm_stage.store(info[3], std::memory_order_relaxed);
            if (rc < 0) {
                sm_release(c->sm, plan);
                return fail(he == hipErrorOutOfMemory ? GS360_ERR_NOMEM : GS360_ERR_HIP, "source-major launch failed: %s", hipGetErrorString(he));
            }
        }
        sm_release(c->sm, plan);
        if (ring) {
            c->last_eq_kernel.store(2, std::memory_order_relaxed);
            return GS360_OK;
        }
    }
    // LDS-staged kernel (eq_staged_kernel, north_star's "LDS-staged source texels"): bilinear RGB u8 views whose row stride keeps dword
    // alignment from row to row; its wavefront tiles are 16 x 16 pixels of the general (non-level) tiling.  When it is taken
    // (steady-state clocks, profiles/r04/stage_sweep.txt, settle_ab.txt, stage_auto_ab.txt): the gather form of PITCHED views that step
    // >= 1.75 source texels per output pixel at their centre is bound by the texture-address path, and staging wins there (8K ->
    // full360coverage: -2 % at step 1.75, -8 % at 1.96, -14 % at 2.6); level views keep the gather kernels' horizon sharing (an all-level
    // ring: level at step 2, -6 % at 2.6; cfg1 36.6 vs 44.3 us staged) and below 1.75 the arithmetic decides (cfg5 75.2 vs 88.7).
    // Splitting a call into a staged and a gather launch loses more in launch tails than it wins (cfg3 95.8 us against 84.4 all
    // gathered and 78.4 all staged), so the CALL is staged as a whole when such views write most of its pixels.
    // GS360_STAGE=0: never; GS360_STAGE=1: every call that can (tests, probes).
    {
        const int mode = opt_stage;                       // -1 auto
        // (the staged kernel forms destination row offsets in 32 bits with a 24-bit multiply: padded strides beyond that take the gather kernels)
        bool can = mode != 0 && C == 3 && esize == 1 && interp == GS360_INTERP_LINEAR && (src_stride & 3) == 0 && dst_stride < ((size_t)1 << 24);
        double px_all = 0.0, px_win = 0.0;
        for (int k = 0; k < n_views && can; ++k) {
            can = ev[k].blocked == 0;
            const double hf = clampd(views[k].hfov_deg, 1e-3, 179.9) * kPi / 180.0;
            const double step = (double)W / (2.0 * kPi) * 2.0 * std::tan(hf * 0.5) / (double)views[k].width;
            const double px = (double)views[k].width * (double)views[k].height;
            px_all += px;
            // (views whose rows are not whole dwords: the staged kernel would write them byte by byte, the gather kernels have a dword path)
            const size_t row_bytes = dst_stride ? dst_stride : (size_t)views[k].width * 3;
            if ((uint64_t)views[k].height * (uint64_t)row_bytes >= ((uint64_t)1 << 32)) can = false;
            if (!ev[k].level && !ev[k].fish && step >= 1.75 && (row_bytes & 3) == 0 && (views[k].width & 3) == 0) px_win += px;
        }
        if (can && (mode == 1 || 2.0 * px_win > px_all))
            for (int k = 0; k < n_views; ++k) {
                ev[k].blocked = 2;
                ev[k].level = 0;
                ev[k].tiles_y = (ev[k].out_h + kTileH - 1) / kTileH;
            }
    }
    // Ring size: unlimited for the row-per-slot lane map (arithmetic-bound views: cfg3 119 -> 99 -> 95 -> 93 us per frame for
    // rings of 1 / 2 / 3 / 4-8 views).  Views on the blocked lane map are memory-bound and gain nothing from shared arithmetic,
    // while a workgroup that walks six views in a row lengthens the launch's tail (cfg2 20.3 -> 22.6 us per frame): no sharing.
    int ring_max = GS360_MAX_VIEWS, ring_max_blocked = 1;
    if (opt_ring >= 1) ring_max = ring_max_blocked = opt_ring;      // option "ring" (tests / probes): 1 = no sharing anywhere, n = at most n views per ring
    std::vector<std::vector<int>> rings;
    const bool ring_forced = opt_ring >= 1;
    for (;;) {
        rings.clear();
        for (int k = 0; k < n_views; ++k) {
            const EqView& b = ev[k];
            int hit = -1;
            for (size_t r = 0; r < rings.size() && hit < 0; ++r) {
                const EqView& a = ev[rings[r][0]];
                if ((int)rings[r].size() < (b.blocked == 1 ? ring_max_blocked : ring_max) && a.sxu == b.sxu && a.syv == b.syv && a.cp == b.cp && (a.sp == b.sp || a.sp == -b.sp) &&
                    a.x0f32 == b.x0f32 && a.out_w == b.out_w && a.out_h == b.out_h && a.level == b.level && a.fish == b.fish &&
                    a.blocked == b.blocked)
                    hit = (int)r;
            }
            if (hit < 0) { rings.emplace_back(); hit = (int)rings.size() - 1; }
            rings[hit].push_back(k);
            ev[k].flip = ev[rings[hit][0]].sp != b.sp ? 1 : 0;
        }
        // A ring's workgroup walks all its members, so a SMALL job in long rings is too few workgroups to fill the chip twice over
        // (one 5.7K frame -> `default`: one ring of 8 = 1300 workgroups for 1280 resident slots: 61 us against 56 us as two rings of
        // 4; the engine's product path launches one frame at a time).  Halve the ring cap until the job has two rounds of workgroups.
        size_t longest = 1;
        long long wgs = 0;
        for (const auto& r : rings) {
            longest = r.size() > longest ? r.size() : longest;
            wgs += (long long)ev[r[0]].tiles_x * ev[r[0]].tiles_y;
        }
        wgs *= n_frames < GS360_MAX_FRAMES ? n_frames : GS360_MAX_FRAMES;
        const long long two_rounds = 2ll * c->prop.multiProcessorCount * 5;
        if (ring_forced || wgs >= two_rounds || longest <= 2 || ring_max <= 2) break;
        ring_max = (int)((longest + 1) / 2);
    }

    // members with the ring's own pitch sign first, the upside-down ones behind them: the kernel's member loop re-derives its
    // latitude-dependent row offsets once per change of sign (results do not depend on the order)
    for (auto& r : rings) std::stable_partition(r.begin(), r.end(), [&](int k) { return ev[k].flip == 0; });
    std::stable_partition(rings.begin(), rings.end(), [&](const std::vector<int>& r) { return ev[r[0]].blocked != 2; });   // gather rings, then staged ones
    size_t r0 = 0;
    while (r0 < rings.size()) {
        size_t r1 = r0;
        int nv = 0;
        const bool staged = ev[rings[r0][0]].blocked == 2;          // staged rings and gather rings never share a launch
        while (r1 < rings.size() && nv + (int)rings[r1].size() <= GS360_MAX_VIEWS && (ev[rings[r1][0]].blocked == 2) == staged)
            nv += (int)rings[r1++].size();
        for (int f0 = 0; f0 < n_frames; f0 += GS360_MAX_FRAMES) {
            int nf = n_frames - f0 < GS360_MAX_FRAMES ? n_frames - f0 : GS360_MAX_FRAMES;
            EqLaunch L;
            std::memset(&L, 0, sizeof(L));
            int order[GS360_MAX_VIEWS];
            int base = 0, j = 0;
            for (size_t r = r0; r < r1; ++r) {
                const EqView& lead = ev[rings[r][0]];
                L.ring_first[r - r0] = j;
                L.ring_count[r - r0] = (int32_t)rings[r].size();
                for (int idx : rings[r]) {
                    L.view[j] = ev[idx];
                    L.view[j].tile_base = base;
                    order[j++] = idx;
                }
                base += lead.tiles_x * lead.tiles_y;
            }
            L.n_rings = (int)(r1 - r0);
            // tiles of rings with different member counts differ in cost: deal them to the XCDs in short runs instead of chunks
            L.xcd_group_log2 = -1;
            for (size_t r = r0; r < r1; ++r)
                if (rings[r].size() != rings[r0].size()) L.xcd_group_log2 = 5;
            if (opt_xcd >= -1) L.xcd_group_log2 = opt_xcd;          // option "xcd_group" (probes): -1 = chunks, g = runs of 2^g tiles
            // Persistent workgroups for the cubic variants (one LDS weight-table fill per workgroup instead of per tile) are OFF:
            // an equirect tile already spreads the fill over its mirrored halves and ring members (2048-32768 pixels), and the
            // static walk costs more in balance than the fill saves (cfg2 / cfg1 / cfg3 cubic: 33.4 / 86.5 / 161.5 us per frame
            // with one tile per workgroup, 35.5 / 90.2 / 177.5 with 2048 persistent ones; profiles/r03/persistent_cubic_ab.txt).
            // The cv2 table kernel, 1024 pixels per tile, gains 12 % from it (launch_table_batch).
            L.persist_blocks = c->opt[kOptEqPersist].load(std::memory_order_relaxed);      // option "eq_persist" (probes): grid cap
            const size_t bits_bytes = mask_bits_bytes;
            const int pitch_dw = mask_pitch_dw;
            if (mask_frames && (r0 == 0 || n_frames > GS360_MAX_FRAMES)) {   // (one frame chunk: later ring groups reuse the images)
                if (int rc = pack_masks(f0, nf)) return rc;
            }
            for (int f = 0; f < nf; ++f) {
                L.src[f] = (const uint8_t*)src_frames[f0 + f];
                L.mask[f] = mask_frames ? (const uint8_t*)c->stage[slot].d_maskbits + bits_bytes * (size_t)f : nullptr;
                for (int k = 0; k < nv; ++k) L.dst[f * nv + k] = (uint8_t*)dst[(size_t)(f0 + f) * n_views + order[k]];
            }
            L.kx32 = (float)(32.0 * (double)W / (2.0 * kPi));
            L.ky32 = (float)(32.0 * (double)H / kPi);
            L.W = W; L.H = H;
            L.y0i32 = 16 * H - 16;
            L.n_views = nv; L.n_frames = nf;
            L.tiles_per_frame = base;
            L.total_tiles = base * nf;
            L.chunk = (L.total_tiles + 7) / 8;
            L.src_stride = (int64_t)src_stride;
            L.mask_stride = (int64_t)pitch_dw * 4;      // of the bit images
            L.dst_stride = (int64_t)dst_stride;
            L.cubic_tab = c->d_cubic;
            if (esize == 2) {
                HIP_TRY(launch_equirect_u16(L, C, interp == GS360_INTERP_CUBIC, c->stream[slot]));
            } else if (interp == GS360_INTERP_CUBIC) {   // same tiling and symmetry reuse, 4x4 taps
                HIP_TRY(launch_equirect_cubic(L, C, c->stream[slot]));
            } else if (staged) {
                HIP_TRY(launch_equirect_staged(L, c->stream[slot]));
            } else {
                HIP_TRY(launch_equirect(L, C, c->stream[slot]));
            }
            c->last_eq_kernel.store(staged ? 1 : 0, std::memory_order_relaxed);
        }
        r0 = r1;
    }
    } catch (const std::bad_alloc&) {
        return fail(GS360_ERR_NOMEM, "out of host memory while planning %d views", n_views);
    }
    return GS360_OK;
}
}  // namespace

// ---- table remap -------------------------------------------------------------------------------
struct gs360_map_plan {         // float maps packed once (gs360_kernels.hip, map plans)
    int device = 0;
    int h = 0, w = 0;
    int nearest = 0;
    int has_valid = 0;
    uint32_t* d_packed = nullptr;
    uint8_t* d_hi = nullptr;
    // stage plans of this map (gs360_tablestage.hip), one per (source size, tile rows, valid bit applied): built at the first call that asks
    mutable std::mutex ts_mutex;
    mutable std::vector<gs360::TsPlan*> ts_plans;
};

namespace {

int check_map_plan(gs360_ctx* c, const gs360_remap_job& J, const gs360_map_plan* plan, int interp) {
    if (plan->device != c->device) return fail(GS360_ERR_ARG, "map plan belongs to device %d, ctx is device %d", plan->device, c->device);
    if (plan->h != J.h || plan->w != J.w) return fail(GS360_ERR_ARG, "map plan is %dx%d, the job asks for %dx%d", plan->w, plan->h, J.w, J.h);
    if (plan->nearest != (interp == GS360_INTERP_NEAREST ? 1 : 0))
        return fail(GS360_ERR_ARG, "map plan was packed for %s sampling", plan->nearest ? "nearest" : "interpolated");
    if (J.W > kMapPlanMaxDim || J.H > kMapPlanMaxDim)
        return fail(GS360_ERR_UNSUPPORTED, "map plans address sources up to %d x %d (got %dx%d): use the float maps", kMapPlanMaxDim,
                    kMapPlanMaxDim, J.W, J.H);
    if (J.valid && !plan->has_valid) return fail(GS360_ERR_ARG, "the job asks for a valid fill, the plan was made without a valid map");
    return 0;
}

int fill_table_job(gs360_ctx* c, const gs360_remap_job& J, const gs360_map_plan* plan, int C, int interp, const double* border_value,
                   TableLaunch* L) {
    if (!J.src || !J.dst || (!plan && (!J.map_x || !J.map_y))) return fail(GS360_ERR_ARG, "NULL argument");
    if (plan)
        if (int rc = check_map_plan(c, J, plan, interp)) return rc;
    if (J.H < 1 || J.W < 1 || J.H >= 32767 || J.W >= 32767) return fail(GS360_ERR_ARG, "source size %dx%d outside cv2.remap limits", J.W, J.H);
    if (J.h < 0 || J.w < 0 || J.h >= 32767 || J.w >= 32767) return fail(GS360_ERR_ARG, "bad map size %dx%d", J.w, J.h);
    size_t src_stride = J.src_stride ? J.src_stride : (size_t)J.W * C;
    size_t dst_stride = J.dst_stride ? J.dst_stride : (size_t)J.w * C;
    if (src_stride < (size_t)J.W * C || dst_stride < (size_t)J.w * C) return fail(GS360_ERR_ARG, "stride smaller than a row");
    std::memset(L, 0, sizeof(*L));
    L->src = (const uint8_t*)J.src; L->map_x = J.map_x; L->map_y = J.map_y; L->valid = J.valid; L->dst = (uint8_t*)J.dst;
    if (plan) {                    // job.valid != NULL asks for the plan's valid bit (the pointer itself is not read)
        L->packed = plan->d_packed; L->packed_hi = plan->d_hi; L->use_valid = J.valid ? 1 : 0;
        L->map_x = L->map_y = nullptr; L->valid = nullptr;
    }
    L->H = J.H; L->W = J.W; L->h = J.h; L->w = J.w;
    L->src_stride = (int64_t)src_stride; L->dst_stride = (int64_t)dst_stride;
    L->interp = interp;
    L->fill = J.fill_value < 0 ? 0 : (J.fill_value > 255 ? 255 : J.fill_value);
    for (int k = 0; k < 4; ++k) L->cval[k] = sat_u8(border_value ? border_value[k] : 0.0);
    L->cubic_tab = interp == GS360_INTERP_LANCZOS4 ? c->d_lanczos : c->d_cubic;
    if (interp == GS360_INTERP_LANCZOS4 && c->lz_rebuild && !c->opt[kOptLanczosTable].load(std::memory_order_relaxed)) {   // (option "lanczos_table": probes / A-B runs)
        L->lz_c1 = c->d_coef1d + 192;
        L->lz_cen = c->d_lz_cen;
    }
    L->pipelined = (J.W >= 8 && src_stride < ((size_t)1 << 24) && (uint64_t)src_stride * (uint64_t)J.H < ((uint64_t)1 << 32)) ? 1 : 0;
    // a tight output whose rows are not whole dwords (the default 1750-pixel views), float maps: spans of the flat output, dword stores
    // (cfg4 70.4-72.8 -> 54.9-57.4 us per pair; with a map plan the byte stores of the row form are as fast: 52.7 vs 54.8, so plans keep it)
    const bool rows_only = c->opt[kOptTableRows].load(std::memory_order_relaxed) != 0;      // (option "table_rows": A/B)
    L->flat = (!plan && dst_stride == (size_t)J.w * C && (dst_stride & 3) != 0 && (reinterpret_cast<uintptr_t>(J.dst) & 3) == 0 &&
               !rows_only) ? 1 : 0;
    return 0;
}

}  // namespace

int gs360_map_plan_create(gs360_ctx* c, const float* map_x, const float* map_y, const uint8_t* valid, int h, int w,
                          int nearest, int slot, gs360_map_plan** out) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!map_x || !map_y || !out) return fail(GS360_ERR_ARG, "NULL argument");
    if (h < 1 || w < 1 || h >= 32767 || w >= 32767) return fail(GS360_ERR_ARG, "bad map size %dx%d", w, h);
    HIP_TRY(hipSetDevice(c->device));
    gs360_map_plan* p = new (std::nothrow) gs360_map_plan();
    if (!p) return fail(GS360_ERR_NOMEM, "out of host memory");
    p->device = c->device; p->h = h; p->w = w; p->nearest = nearest ? 1 : 0; p->has_valid = valid ? 1 : 0;
    const size_t n = (size_t)h * (size_t)w;
    hipError_t e = hipMalloc((void**)&p->d_packed, n * sizeof(uint32_t) + kSlack);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_hi, n + kSlack);
    if (e == hipSuccess) e = launch_map_pack(map_x, map_y, valid, (int64_t)n, p->nearest, p->d_packed, p->d_hi, c->stream[slot]);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream[slot]);      // the caller may release its maps on return
    if (e != hipSuccess) {
        if (p->d_packed) (void)hipFree(p->d_packed);
        if (p->d_hi) (void)hipFree(p->d_hi);
        delete p;
        return fail(e == hipErrorOutOfMemory ? GS360_ERR_NOMEM : GS360_ERR_HIP, "map plan setup failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return GS360_OK;
}

int gs360_map_plan_destroy(gs360_ctx* c, gs360_map_plan* p) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!p) return GS360_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    if (p->d_packed) HIP_TRY(hipFree(p->d_packed));
    if (p->d_hi) HIP_TRY(hipFree(p->d_hi));
    for (gs360::TsPlan* t : p->ts_plans) gs360::ts_plan_free(t);
    delete p;
    return GS360_OK;
}

static int remap_batches_u8(gs360_ctx* c, const gs360_remap_job* jobs, const gs360_map_plan* const* plans, int n_jobs, int C, int interp,
                            const double* border_value, int slot);

int gs360_remap_plans_u8(gs360_ctx* c, const gs360_remap_job* jobs, const gs360_map_plan* const* plans, int n_jobs, int C,
                         int interp, const double* border_value, int slot) {
    if (n_jobs > 0 && !plans) return fail(GS360_ERR_ARG, "plans is NULL");
    return remap_batches_u8(c, jobs, plans, n_jobs, C, interp, border_value, slot);
}

int gs360_remap_tables_u8(gs360_ctx* c, const gs360_remap_job* jobs, int n_jobs, int C, int interp,
                          const double* border_value, int slot) {
    return remap_batches_u8(c, jobs, nullptr, n_jobs, C, interp, border_value, slot);
}

static int remap_batches_u8(gs360_ctx* c, const gs360_remap_job* jobs, const gs360_map_plan* const* plans, int n_jobs, int C, int interp,
                            const double* border_value, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(GS360_ERR_ARG, "bad job list");
    if (C != 1 && C != 3 && C != 4) return fail(GS360_ERR_ARG, "C must be 1, 3 or 4 (got %d)", C);
    if (interp != GS360_INTERP_LINEAR && interp != GS360_INTERP_NEAREST && interp != GS360_INTERP_CUBIC &&
        interp != GS360_INTERP_LANCZOS4)
        return fail(GS360_ERR_UNSUPPORTED, "interp %d not implemented (nearest=0, linear=1, cubic=2, lanczos4=4)", interp);
    HIP_TRY(hipSetDevice(c->device));
    for (int j0 = 0; j0 < n_jobs; j0 += GS360_MAX_VIEWS) {
        TableBatch B;
        B.n_jobs = 0;
        B.persist_blocks = c->prop.multiProcessorCount * 8;      // two rounds of the four workgroups a CU holds (bicubic RGB)
        if (const int v = c->opt[kOptTablePersist].load(std::memory_order_relaxed); v >= 0) B.persist_blocks = v;   // option "table_persist" (probes): 0 = one tile per workgroup
        // LDS-staged kernel (gs360_tablestage.hip) for the jobs that can take it: bilinear RGB through a map plan, dword-aligned source rows,
        // an output whose quads start on dword boundaries (tight, or rows of whole dwords).  Their stage plans are built at the first call
        // (one launch + one synchronisation of the slot's stream per map plan and source size).  Option "table_stage": 0 never, 1 every job
        // that can, -1 (default) those whose plan has boxes for at least 7/8 of its tiles (a map that scatters its taps -- random test maps --
        // would be redone pixel by pixel from memory).
        TsLaunch S;
        std::memset(&S, 0, sizeof(S));
        const int opt_stage = c->opt[kOptTableStage].load(std::memory_order_relaxed);
        S.R = c->opt[kOptTableStageRows].load(std::memory_order_relaxed);
        S.wg_per_cu = c->opt[kOptTableStageWgs].load(std::memory_order_relaxed);
        int slow_tiles = 0;
        for (int j = j0; j < n_jobs && j < j0 + GS360_MAX_VIEWS; ++j) {
            if (jobs[j].h == 0 || jobs[j].w == 0) continue;
            TableLaunch& L = B.job[B.n_jobs];
            if (int rc = fill_table_job(c, jobs[j], plans ? plans[j] : nullptr, C, interp, border_value, &L)) return rc;
            const gs360_map_plan* plan = plans ? plans[j] : nullptr;
            const bool quads_ok = ((uintptr_t)L.dst & 3) == 0 && (L.dst_stride == (int64_t)3 * L.w ? ((int64_t)L.h * L.w) % 4 == 0 : (L.w % 4 == 0 && L.dst_stride % 4 == 0));
            if (opt_stage != 0 && plan && C == 3 && interp == GS360_INTERP_LINEAR && L.pipelined && quads_ok && ((uintptr_t)L.src & 3) == 0 &&
                L.src_stride % 4 == 0 && (int64_t)L.H * L.src_stride < ((int64_t)1 << 31) && (int64_t)L.h * L.dst_stride < ((int64_t)1 << 32) &&
                (int64_t)L.h * L.w < ((int64_t)1 << 30)) {
                TsPlan* tp = nullptr;
                {
                    std::lock_guard<std::mutex> lock(plan->ts_mutex);
                    for (TsPlan* q : plan->ts_plans)
                        if (q->W == L.W && q->H == L.H && q->R == S.R && q->use_valid == L.use_valid) { tp = q; break; }
                    if (!tp) {
                        hipError_t he = hipSuccess;
                        tp = ts_build_plan(plan->d_packed, plan->d_hi, L.h, L.w, L.W, L.H, S.R, L.use_valid, kTsBoxBudget,
                                           c->stream[slot], &he);
                        if (!tp) return fail(he == hipSuccess || he == hipErrorOutOfMemory ? GS360_ERR_NOMEM : GS360_ERR_HIP, "stage plan setup failed: %s", hipGetErrorString(he));
                        plan->ts_plans.push_back(tp);
                    }
                }
                if (opt_stage == 1 || tp->slow_tiles * 8 <= tp->n_tiles) {
                    TsJobDesc& D = S.job[S.n_jobs++];
                    D.src = L.src; D.dst = L.dst; D.packed = L.packed; D.packed_hi = L.packed_hi; D.plan = tp;
                    D.src_stride = L.src_stride; D.dst_stride = L.dst_stride; D.fill = L.fill;
                    for (int k = 0; k < 4; ++k) S.cval[k] = L.cval[k];
                    slow_tiles += tp->slow_tiles;
                    continue;                            // (B.job[B.n_jobs] is overwritten by the next job)
                }
            }
            ++B.n_jobs;
        }
        if (S.n_jobs) HIP_TRY(ts_launch(S, c->prop.multiProcessorCount, 160 * 1024, c->stream[slot]));
        if (B.n_jobs) HIP_TRY(launch_table_batch(B, C, c->stream[slot]));
        c->last_table_kernel.store(S.n_jobs, std::memory_order_relaxed);
        c->last_table_slow.store(slow_tiles, std::memory_order_relaxed);
    }
    return GS360_OK;
}

int gs360_remap_table_u8(gs360_ctx* c, const void* src, int H, int W, int C, size_t src_stride, const float* map_x,
                         const float* map_y, const uint8_t* valid, int h, int w, int interp,
                         const double* border_value, int fill_value, void* dst, size_t dst_stride, int slot) {
    gs360_remap_job J;
    J.src = src; J.H = H; J.W = W; J.src_stride = src_stride; J.map_x = map_x; J.map_y = map_y; J.valid = valid;
    J.h = h; J.w = w; J.fill_value = fill_value; J.dst = dst; J.dst_stride = dst_stride;
    return gs360_remap_tables_u8(c, &J, 1, C, interp, border_value, slot);
}

static int remap_batches_u16(gs360_ctx* c, const gs360_remap_job* jobs, const gs360_map_plan* const* plans, int n_jobs, int C, int interp,
                             const double* border_value, int slot);

int gs360_remap_tables_u16(gs360_ctx* c, const gs360_remap_job* jobs, int n_jobs, int C, int interp,
                           const double* border_value, int slot) {
    return remap_batches_u16(c, jobs, nullptr, n_jobs, C, interp, border_value, slot);
}

int gs360_remap_plans_u16(gs360_ctx* c, const gs360_remap_job* jobs, const gs360_map_plan* const* plans, int n_jobs, int C,
                          int interp, const double* border_value, int slot) {
    if (n_jobs > 0 && !plans) return fail(GS360_ERR_ARG, "plans is NULL");
    return remap_batches_u16(c, jobs, plans, n_jobs, C, interp, border_value, slot);
}

static int remap_batches_u16(gs360_ctx* c, const gs360_remap_job* jobs, const gs360_map_plan* const* plans, int n_jobs, int C, int interp,
                             const double* border_value, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return fail(GS360_ERR_ARG, "bad job list");
    if (C != 1 && C != 3 && C != 4) return fail(GS360_ERR_ARG, "C must be 1, 3 or 4 (got %d)", C);
    if (interp != GS360_INTERP_LINEAR && interp != GS360_INTERP_NEAREST && interp != GS360_INTERP_CUBIC &&
        interp != GS360_INTERP_LANCZOS4)
        return fail(GS360_ERR_UNSUPPORTED, "interp %d not implemented (nearest=0, linear=1, cubic=2, lanczos4=4)", interp);
    uint16_t cval[4];
    for (int k = 0; k < 4; ++k) {      // cv::saturate_cast<ushort>(double)
        long r = std::lrint(border_value ? border_value[k] : 0.0);
        cval[k] = (uint16_t)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
    }
    HIP_TRY(hipSetDevice(c->device));
    for (int j0 = 0; j0 < n_jobs; j0 += GS360_MAX_VIEWS) {
        TableBatch B;
        B.n_jobs = 0;
        B.persist_blocks = 0;
        for (int j = j0; j < n_jobs && j < j0 + GS360_MAX_VIEWS; ++j) {
            const gs360_remap_job& J = jobs[j];
            const gs360_map_plan* plan = plans ? plans[j] : nullptr;
            if (!J.src || !J.dst || (!plan && (!J.map_x || !J.map_y))) return fail(GS360_ERR_ARG, "NULL argument");
            if (plan)
                if (int rc = check_map_plan(c, J, plan, interp)) return rc;
            if (J.H < 1 || J.W < 1 || J.H >= 32767 || J.W >= 32767) return fail(GS360_ERR_ARG, "source size %dx%d outside cv2.remap limits", J.W, J.H);
            if (J.h < 0 || J.w < 0 || J.h >= 32767 || J.w >= 32767) return fail(GS360_ERR_ARG, "bad map size %dx%d", J.w, J.h);
            if (J.h == 0 || J.w == 0) continue;
            const size_t src_stride = J.src_stride ? J.src_stride : (size_t)J.W * C * 2;
            const size_t dst_stride = J.dst_stride ? J.dst_stride : (size_t)J.w * C * 2;
            if (src_stride < (size_t)J.W * C * 2 || dst_stride < (size_t)J.w * C * 2) return fail(GS360_ERR_ARG, "stride smaller than a row");
            if ((src_stride | dst_stride) & 1) return fail(GS360_ERR_ARG, "16-bit images need even strides");
            TableLaunch& L = B.job[B.n_jobs++];
            std::memset(&L, 0, sizeof(L));
            L.src = (const uint8_t*)J.src; L.map_x = J.map_x; L.map_y = J.map_y; L.valid = J.valid; L.dst = (uint8_t*)J.dst;
            L.H = J.H; L.W = J.W; L.h = J.h; L.w = J.w;
            L.src_stride = (int64_t)src_stride; L.dst_stride = (int64_t)dst_stride;
            L.interp = interp;
            L.fill = J.fill_value < 0 ? 0 : (J.fill_value > 65535 ? 65535 : J.fill_value);
            if (plan) {
                L.packed = plan->d_packed; L.packed_hi = plan->d_hi; L.use_valid = J.valid ? 1 : 0;
                L.map_x = L.map_y = nullptr; L.valid = nullptr;
            }
        }
        if (B.n_jobs) HIP_TRY(launch_table_u16_batch(B, C, c->d_coef1d, cval, c->stream[slot]));
    }
    return GS360_OK;
}

int gs360_remap_table_u16(gs360_ctx* c, const void* src, int H, int W, int C, size_t src_stride, const float* map_x,
                          const float* map_y, const uint8_t* valid, int h, int w, int interp,
                          const double* border_value, int fill_value, void* dst, size_t dst_stride, int slot) {
    gs360_remap_job J;
    J.src = src; J.H = H; J.W = W; J.src_stride = src_stride; J.map_x = map_x; J.map_y = map_y; J.valid = valid;
    J.h = h; J.w = w; J.fill_value = fill_value; J.dst = dst; J.dst_stride = dst_stride;
    return gs360_remap_tables_u16(c, &J, 1, C, interp, border_value, slot);
}

// ---- fused fisheye -> views --------------------------------------------------------------------
int gs360_fisheye_views_u8(gs360_ctx* c, const void* const* src_lens, const gs360_calib* calibs, int C, size_t src_stride,
                           const gs360_view* views, int n_views, double lens_fov_deg, int interp, int mask_outside,
                           int mask_value, void* const* dst, size_t dst_stride, uint8_t* const* valid_out, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!src_lens || !calibs || !views || !dst) return fail(GS360_ERR_ARG, "NULL argument");
    if (n_views < 0) return fail(GS360_ERR_ARG, "negative count");
    if (n_views == 0) return GS360_OK;
    if (C != 1 && C != 3 && C != 4) return fail(GS360_ERR_ARG, "C must be 1, 3 or 4 (got %d)", C);
    if (interp != GS360_INTERP_LINEAR && interp != GS360_INTERP_NEAREST && interp != GS360_INTERP_CUBIC &&
        interp != GS360_INTERP_LANCZOS4)
        return fail(GS360_ERR_UNSUPPORTED, "interp %d not implemented (nearest=0, linear=1, cubic=2, lanczos4=4)", interp);
    for (int k = 0; k < n_views; ++k) {
        if (!src_lens[k] || !dst[k]) return fail(GS360_ERR_ARG, "NULL image pointer for view %d", k);
        if (calibs[k].width < 1 || calibs[k].height < 1 || calibs[k].width >= 32767 || calibs[k].height >= 32767)
            return fail(GS360_ERR_ARG, "bad sensor size for view %d", k);
        if (calibs[k].width != calibs[0].width && src_stride != 0)
            return fail(GS360_ERR_ARG, "explicit src_stride needs equal sensor widths");
        if (views[k].width < 1 || views[k].height < 1 || views[k].width > 32768 || views[k].height > 32768)
            return fail(GS360_ERR_ARG, "view %d has bad size", k);
    }
    HIP_TRY(hipSetDevice(c->device));
    mask_value = mask_value < 0 ? 0 : (mask_value > 255 ? 255 : mask_value);
    for (int v0 = 0; v0 < n_views; v0 += GS360_MAX_VIEWS) {
        int nv = n_views - v0 < GS360_MAX_VIEWS ? n_views - v0 : GS360_MAX_VIEWS;
        // one launch per group of equal-width sensors keeps a single src_stride in the parameter block
        FeLaunch L;
        std::memset(&L, 0, sizeof(L));
        int base = 0;
        for (int k = 0; k < nv; ++k) {
            make_fe_view(calibs[v0 + k], views[v0 + k], lens_fov_deg, &L.view[k]);
            L.view[k].src = (const uint8_t*)src_lens[v0 + k];
            L.view[k].dst = (uint8_t*)dst[v0 + k];
            L.view[k].valid_out = valid_out ? valid_out[v0 + k] : nullptr;
            L.view[k].tile_base = base;
            base += L.view[k].tiles_x * L.view[k].tiles_y;
            if (calibs[v0 + k].width != calibs[v0].width)
                return fail(GS360_ERR_UNSUPPORTED, "views of one call must share the sensor width");
        }
        L.n_views = nv;
        L.total_tiles = base;
        L.chunk = (base + 7) / 8;
        L.interp = interp; L.mask_outside = mask_outside ? 1 : 0; L.mask_value = mask_value;
        L.src_stride = (int64_t)(src_stride ? src_stride : (size_t)calibs[v0].width * C);
        L.dst_stride = (int64_t)dst_stride;
        L.cval[0] = (uint8_t)mask_value;  // borderValue=float(mask_value) -> Scalar(v,0,0,0), DF:2007
        L.cubic_tab = interp == GS360_INTERP_LANCZOS4 ? c->d_lanczos : c->d_cubic;
        L.pipelined = 1;
        for (int k = 0; k < nv; ++k)
            if (calibs[v0 + k].width < 8 || (uint64_t)L.src_stride * (uint64_t)calibs[v0 + k].height >= ((uint64_t)1 << 32)) L.pipelined = 0;
        if ((uint64_t)L.src_stride >= ((uint64_t)1 << 24)) L.pipelined = 0;
        L.persist_blocks = c->prop.multiProcessorCount * 8;
        if (const int v = c->opt[kOptTablePersist].load(std::memory_order_relaxed); v >= 0) L.persist_blocks = v;
        HIP_TRY(launch_fisheye(L, C, c->stream[slot]));
    }
    return GS360_OK;
}

// ---- input colour stage ------------------------------------------------------------------------
struct gs360_color_plan {
    int device = 0;
    int lut_size = 0;
    int fixups = 0;
    void* d_rtab = nullptr;     // cell-major red-interpolated LUT, see gs360_color.hip (released once the cube is built)
    float* d_tables = nullptr;  // level positions, thresholds, bin levels
    void* d_cube = nullptr;     // uint32[2^24]: the stage evaluated for every 8-bit pixel (NULL with GS360_COLOR_CUBE=0)
};

int gs360_color_plan_create(gs360_ctx* c, const float* lut, int lut_size, const float* level_pos,
                            const float* out_thresholds, gs360_color_plan** out) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!lut || !level_pos || !out_thresholds || !out) return fail(GS360_ERR_ARG, "NULL argument");
    if (lut_size < 2 || lut_size > 256) return fail(GS360_ERR_ARG, "LUT size %d outside [2,256]", lut_size);
    const int nmax = lut_size - 1;
    for (int i = 0; i < 768; ++i)   // positions index the table: refuse anything that would read outside it
        if (!(level_pos[i] >= 0.0f && level_pos[i] <= (float)nmax))
            return fail(GS360_ERR_ARG, "level_pos[%d] = %g outside [0,%d]", i, (double)level_pos[i], nmax);
    for (int k = 1; k < 256; ++k) {
        if (!(out_thresholds[k] >= 0.0f)) return fail(GS360_ERR_ARG, "out_thresholds[%d] is negative or NaN", k);
        if (k > 1 && !(out_thresholds[k] >= out_thresholds[k - 1]))
            return fail(GS360_ERR_ARG, "out_thresholds must be non-decreasing (entry %d)", k);
    }
    HIP_TRY(hipSetDevice(c->device));
    gs360_color_plan* p = new (std::nothrow) gs360_color_plan();
    if (!p) return fail(GS360_ERR_NOMEM, "out of host memory");
    p->device = c->device;
    p->lut_size = lut_size;
    const size_t n3 = (size_t)lut_size * lut_size * lut_size;
    std::vector<float> tables(color_tables_floats(), 0.0f);
    std::memcpy(tables.data(), level_pos, 768 * sizeof(float));
    std::memcpy(tables.data() + 768, out_thresholds, 256 * sizeof(float));
    p->fixups = color_build_bins(out_thresholds, (uint8_t*)(tables.data() + 1024));
    float* d_lut = nullptr;
    hipError_t e = hipMalloc((void**)&d_lut, n3 * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&p->d_rtab, color_rtab_bytes(lut_size) + kSlack);
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_tables, tables.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_lut, lut, n3 * 3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_tables, tables.data(), tables.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = build_color_rtab(d_lut, p->d_tables /* red positions come first */, p->d_rtab, lut_size, c->stream[0]);
    const bool want_cube = c->opt[kOptColorCube].load(std::memory_order_relaxed) != 0;      // option "color_cube" (-1 / 1: yes)
    if (e == hipSuccess && want_cube) {
        e = hipMalloc(&p->d_cube, color_cube_bytes());
        if (e == hipErrorOutOfMemory) {          // a crowded device: the per-pixel evaluation gives the same results from the 18 MB it already has
            (void)hipGetLastError();
            p->d_cube = nullptr;
            e = hipSuccess;
        } else if (e == hipSuccess) {
            ColorLaunch B{};
            B.rtab = p->d_rtab; B.tables = p->d_tables; B.lut_size = lut_size; B.fixups = p->fixups;
            e = build_color_cube(B, p->d_cube, c->stream[0]);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream[0]);
    if (d_lut) (void)hipFree(d_lut);
    if (e == hipSuccess && p->d_cube) {          // the cube replaces the tables it was built from
        (void)hipFree(p->d_rtab);
        p->d_rtab = nullptr;
    }
    if (e != hipSuccess) {
        if (p->d_rtab) (void)hipFree(p->d_rtab);
        if (p->d_tables) (void)hipFree(p->d_tables);
        if (p->d_cube) (void)hipFree(p->d_cube);
        delete p;
        return fail(e == hipErrorOutOfMemory ? GS360_ERR_NOMEM : GS360_ERR_HIP, "colour plan setup failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return GS360_OK;
}

int gs360_color_plan_destroy(gs360_ctx* c, gs360_color_plan* p) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!p) return GS360_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    if (p->d_rtab) HIP_TRY(hipFree(p->d_rtab));
    if (p->d_tables) HIP_TRY(hipFree(p->d_tables));
    if (p->d_cube) HIP_TRY(hipFree(p->d_cube));
    delete p;
    return GS360_OK;
}

int gs360_color_apply_u8(gs360_ctx* c, const gs360_color_plan* p, const void* src, int H, int W, int C, size_t src_stride,
                         int red_index, void* dst, size_t dst_stride, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!p || !src || !dst) return fail(GS360_ERR_ARG, "NULL argument");
    if (p->device != c->device) return fail(GS360_ERR_ARG, "colour plan belongs to device %d, ctx is device %d", p->device, c->device);
    if (C != 3 && C != 4) return fail(GS360_ERR_ARG, "the LUT stage needs 3 or 4 channels (got %d)", C);   // DF:693-697
    if (red_index != 0 && red_index != 2) return fail(GS360_ERR_ARG, "red_index must be 0 (RGB) or 2 (BGR)");
    if (H < 0 || W < 0) return fail(GS360_ERR_ARG, "bad size");
    if (H == 0 || W == 0) return GS360_OK;
    if (H > 65535) return fail(GS360_ERR_UNSUPPORTED, "image height %d above 65535", H);
    if (src_stride == 0) src_stride = (size_t)W * C;
    if (dst_stride == 0) dst_stride = (size_t)W * C;
    if (src_stride < (size_t)W * C || dst_stride < (size_t)W * C) return fail(GS360_ERR_ARG, "stride smaller than a row");
    HIP_TRY(hipSetDevice(c->device));
    ColorLaunch L;
    L.src = (const uint8_t*)src; L.dst = (uint8_t*)dst; L.rtab = p->d_rtab; L.tables = p->d_tables; L.cube = p->d_cube;
    L.H = H; L.W = W; L.lut_size = p->lut_size; L.red_index = red_index; L.fixups = p->fixups;
    L.src_stride = (int64_t)src_stride; L.dst_stride = (int64_t)dst_stride;
    HIP_TRY(launch_color(L, C, c->stream[slot]));
    return GS360_OK;
}

struct gs360_color_plan16 {
    int device = 0;
    gs360::Color16Launch L;
    float* d_lut = nullptr;
    float* d_thr = nullptr;
    void* d_bins = nullptr;
};

int gs360_color_plan16_create(gs360_ctx* c, const float* lut, int lut_size, const float* domain_min, const float* domain_max,
                              int n_pieces, const float* piece_start, const int32_t* piece_base, const int32_t* piece_off,
                              const float* thresholds, gs360_color_plan16** out) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!lut || !domain_min || !domain_max || !out) return fail(GS360_ERR_ARG, "NULL argument");
    if (lut_size < 2 || lut_size > 256) return fail(GS360_ERR_ARG, "LUT size %d outside [2,256]", lut_size);
    if (n_pieces < 0 || n_pieces > 4) return fail(GS360_ERR_ARG, "n_pieces must be in [0,4]");
    if (n_pieces && (!piece_start || !piece_base || !piece_off || !thresholds)) return fail(GS360_ERR_ARG, "NULL piece tables");
    gs360_color_plan16* p = new (std::nothrow) gs360_color_plan16();
    if (!p) return fail(GS360_ERR_NOMEM, "out of host memory");
    std::memset(&p->L, 0, sizeof(p->L));
    for (int k = 0; k < 3; ++k) {
        p->L.dmin[k] = domain_min[k];
        p->L.span[k] = domain_max[k] - domain_min[k];                 // float32 subtraction, DF:641
        if (!(p->L.span[k] > 0.0f)) { delete p; return fail(GS360_ERR_ARG, "invalid LUT domain on channel %d", k); }
    }
    int total = 0;
    for (int q = 0; q < n_pieces; ++q) {
        const int lo = piece_off[q], hi = piece_off[q + 1];
        if (lo != total || hi < lo || hi > (1 << 20)) { delete p; return fail(GS360_ERR_ARG, "piece_off must be contiguous and ascending"); }
        for (int i = lo + 1; i < hi; ++i)
            if (!(thresholds[i] >= thresholds[i - 1])) { delete p; return fail(GS360_ERR_ARG, "thresholds of piece %d are not sorted (entry %d)", q, i); }
        if (piece_base[q] < 0 || piece_base[q] + (hi - lo) > 65535) { delete p; return fail(GS360_ERR_ARG, "piece %d would produce levels above 65535", q); }
        if (q > 0 && !(piece_start[q] >= piece_start[q - 1])) { delete p; return fail(GS360_ERR_ARG, "piece_start must be ascending"); }
        p->L.start[q] = q ? piece_start[q] : 0.0f;
        p->L.base[q] = piece_base[q];
        p->L.off[q] = lo;
        total = hi;
    }
    p->L.off[n_pieces] = total;
    p->L.n_pieces = n_pieces;
    p->L.lut_size = lut_size;
    p->device = c->device;
    const size_t n3 = (size_t)lut_size * lut_size * lut_size * 3;
    hipError_t e = hipSetDevice(c->device);                     // (a failure here must release the host plan too)
    if (e == hipSuccess) e = hipMalloc((void**)&p->d_lut, n3 * sizeof(float) + kSlack);
    if (e == hipSuccess) e = hipMemcpy(p->d_lut, lut, n3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && total) e = hipMalloc((void**)&p->d_thr, (size_t)total * sizeof(float));
    if (e == hipSuccess && total) e = hipMemcpy(p->d_thr, thresholds, (size_t)total * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess && n_pieces) {
        std::vector<uint8_t> bins(color16_bins_bytes());
        color16_build_bins(n_pieces, p->L.start, p->L.off, thresholds, bins.data());
        e = hipMalloc((void**)&p->d_bins, bins.size());
        if (e == hipSuccess) e = hipMemcpy(p->d_bins, bins.data(), bins.size(), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (p->d_lut) (void)hipFree(p->d_lut);
        if (p->d_thr) (void)hipFree(p->d_thr);
        if (p->d_bins) (void)hipFree(p->d_bins);
        delete p;
        return fail(GS360_ERR_HIP, "colour plan setup failed: %s", hipGetErrorString(e));
    }
    p->L.lut = p->d_lut;
    p->L.thr = p->d_thr;
    p->L.bins = p->d_bins;
    *out = p;
    return GS360_OK;
}

int gs360_color_plan16_destroy(gs360_ctx* c, gs360_color_plan16* p) {
    if (!c) return fail(GS360_ERR_ARG, "ctx is NULL");
    if (!p) return GS360_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    if (p->d_lut) HIP_TRY(hipFree(p->d_lut));
    if (p->d_thr) HIP_TRY(hipFree(p->d_thr));
    if (p->d_bins) HIP_TRY(hipFree(p->d_bins));
    delete p;
    return GS360_OK;
}

int gs360_color_apply_u16(gs360_ctx* c, const gs360_color_plan16* p, const void* src, int H, int W, int C, size_t src_stride,
                          int red_index, void* dst, size_t dst_stride, int slot) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!p || !src || !dst) return fail(GS360_ERR_ARG, "NULL argument");
    if (p->device != c->device) return fail(GS360_ERR_ARG, "colour plan belongs to device %d, ctx is device %d", p->device, c->device);
    if (C != 3 && C != 4) return fail(GS360_ERR_ARG, "the LUT stage needs 3 or 4 channels (got %d)", C);
    if (red_index != 0 && red_index != 2) return fail(GS360_ERR_ARG, "red_index must be 0 (RGB) or 2 (BGR)");
    if (H < 0 || W < 0) return fail(GS360_ERR_ARG, "bad size");
    if (H == 0 || W == 0) return GS360_OK;
    if (H > 65535) return fail(GS360_ERR_UNSUPPORTED, "image height %d above 65535", H);
    if (src_stride == 0) src_stride = (size_t)W * C * 2;
    if (dst_stride == 0) dst_stride = (size_t)W * C * 2;
    if (src_stride < (size_t)W * C * 2 || dst_stride < (size_t)W * C * 2 || ((src_stride | dst_stride) & 1))
        return fail(GS360_ERR_ARG, "16-bit images need even strides of at least one row");
    HIP_TRY(hipSetDevice(c->device));
    gs360::Color16Launch L = p->L;
    L.src = src; L.dst = dst; L.H = H; L.W = W; L.red_index = red_index;
    L.src_stride = (int64_t)src_stride; L.dst_stride = (int64_t)dst_stride;
    HIP_TRY(launch_color16(L, C, c->stream[slot]));
    return GS360_OK;
}

// ---- image-codec helper (host only) ---------------------------------------------------------------
// PNG scanline reconstruction (filter types 0-4) in place: `data` holds h rows of (1 + stride) bytes as inflated from the
// IDAT stream; on return row y's pixels sit at data + y * (stride + 1) + 1.  Both directions of a PNG filter are
// sequential (left neighbour and previous row), so the Python-side codec (gs360/imageio.py, used for 16-bit PNG, which
// Pillow cannot deliver at full depth for RGB) calls this instead of looping over bytes.  No GPU involved.
int gs360_png_unfilter(uint8_t* data, int h, int stride, int bpp) {
    if (!data || h < 0 || stride < 1 || bpp < 1 || bpp > 8) return fail(GS360_ERR_ARG, "bad PNG geometry");
    const size_t pitch = (size_t)stride + 1;
    for (int y = 0; y < h; ++y) {
        uint8_t* cur = data + (size_t)y * pitch + 1;
        const uint8_t* up = y ? cur - pitch : nullptr;
        const int ft = cur[-1];
        switch (ft) {
            case 0: break;
            case 1: for (int i = bpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
            case 2: if (up) for (int i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + up[i]); break;
            case 3:
                for (int i = 0; i < stride; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                    cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (int i = 0; i < stride; ++i) {
                    const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                    const int pp = a + b - c, pa = std::abs(pp - a), pb = std::abs(pp - b), pc = std::abs(pp - c);
                    cur[i] = (uint8_t)(cur[i] + ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c)));
                }
                break;
            default: return fail(GS360_ERR_ARG, "PNG row %d has unknown filter type %d", y, ft);
        }
    }
    return GS360_OK;
}

// TIFF LZW strip decoder (compression 5: MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, "early change").
// Host helper like gs360_png_unfilter: 16-bit TIFF panoramas are commonly LZW-compressed and the Python-side codec cannot loop
// over codes at image scale.  Writes at most out_cap bytes; *out_len receives the number produced.
int gs360_tiff_lzw_decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap, size_t* out_len) {
    if (!in || !out || !out_len) return fail(GS360_ERR_ARG, "NULL argument");
    struct Entry { uint32_t pos, len; };                 // every string is a slice of the output written so far
    std::vector<Entry> tab(4096);
    size_t op = 0, bitpos = 0;
    int next = 258, width = 9;
    int64_t prev = -1;
    const size_t nbits = in_len * 8;
    auto emit = [&](const uint8_t* srcp, uint32_t len) -> bool {
        if (op + len > out_cap) len = (uint32_t)(out_cap - op);
        for (uint32_t i = 0; i < len; ++i) out[op + i] = srcp[i];      // may overlap forwards: byte copy
        op += len;
        return op < out_cap;
    };
    while (bitpos + width <= nbits) {
        uint32_t code = 0;
        for (int b = 0; b < width; ++b) {
            const size_t bp = bitpos + b;
            code = (code << 1) | ((in[bp >> 3] >> (7 - (bp & 7))) & 1u);
        }
        bitpos += width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; prev = -1; continue; }
        const uint32_t start = (uint32_t)op;
        if (prev < 0) {                                   // first code after a clear: a literal
            if (code > 255) return fail(GS360_ERR_ARG, "corrupt LZW stream (code %u after clear)", code);
            const uint8_t lit = (uint8_t)code;
            tab[code] = Entry{start, 1};
            if (!emit(&lit, 1)) break;
            prev = code;
            continue;
        }
        const Entry pe = prev < 256 ? Entry{0, 1} : tab[prev];
        uint8_t plit = (uint8_t)prev;
        const uint8_t* pstr = prev < 256 ? &plit : out + pe.pos;
        bool more;
        if (code < 256) {
            const uint8_t lit = (uint8_t)code;
            more = emit(&lit, 1);
        } else if ((int)code < next) {
            const Entry e = tab[code];
            more = emit(out + e.pos, e.len);
        } else if ((int)code == next) {                  // KwKwK: previous string + its own first byte
            const uint32_t plen = prev < 256 ? 1u : pe.len;
            const uint8_t first = pstr[0];
            more = emit(pstr, plen);
            if (more) more = emit(&first, 1);
        } else {
            return fail(GS360_ERR_ARG, "corrupt LZW stream (code %u, table size %d)", code, next);
        }
        if (next < 4096) {                                // new entry = previous string + first byte of this one; it is
            const uint32_t plen = prev < 256 ? 1u : pe.len;   // exactly the bytes [start - plen, start + 1) of the output
            tab[next] = Entry{start - plen, plen + 1};
            ++next;
            if (next + 1 >= (1 << width) && width < 12) ++width;       // early change
        }
        prev = code;
        if (!more) break;
    }
    *out_len = op;
    return GS360_OK;
}

// ---- host-buffer conveniences ------------------------------------------------------------------
namespace {
int equirect_views_host_impl(gs360_ctx* c, const void* src, int W, int H, int C, size_t src_stride,
                             const gs360_view* views, int n_views, void* const* dst, size_t dst_stride,
                             int interp, uint32_t flags, int slot, int esize) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!src || !views || !dst) return fail(GS360_ERR_ARG, "NULL argument");
    if (n_views <= 0) return n_views == 0 ? GS360_OK : fail(GS360_ERR_ARG, "negative count");
    if (C != 1 && C != 3 && C != 4) return fail(GS360_ERR_ARG, "C must be 1, 3 or 4 (got %d)", C);
    if (W < 2 || H < 2) return fail(GS360_ERR_ARG, "bad source size");
    if (src_stride == 0) src_stride = (size_t)W * C * esize;
    HIP_TRY(hipSetDevice(c->device));
    Staging& S = c->stage[slot];
    size_t src_bytes = src_stride * (size_t)H;
    std::vector<size_t> off(n_views);
    size_t total = 0;
    for (int k = 0; k < n_views; ++k) {
        if (views[k].width < 1 || views[k].height < 1) return fail(GS360_ERR_ARG, "view %d has bad size", k);
        size_t ds = dst_stride ? dst_stride : (size_t)views[k].width * C * esize;
        off[k] = total;
        total += (ds * (size_t)views[k].height + 255) & ~(size_t)255;
    }
    if (int rc = ensure(c, &S.d_src, &S.src_cap, src_bytes)) return rc;
    if (int rc = ensure(c, &S.d_dst, &S.dst_cap, total)) return rc;
    hipStream_t st = c->stream[slot];
    HIP_TRY(hipMemcpyAsync(S.d_src, src, src_bytes, hipMemcpyHostToDevice, st));
    std::vector<void*> dptr(n_views);
    for (int k = 0; k < n_views; ++k) dptr[k] = (uint8_t*)S.d_dst + off[k];
    const void* frames[1] = {S.d_src};
    if (int rc = equirect_views_impl(c, frames, nullptr, 1, W, H, C, src_stride, 0, views, n_views, dptr.data(), dst_stride, interp,
                                     flags, slot, esize))
        return rc;
    for (int k = 0; k < n_views; ++k) {
        if (!dst[k]) return fail(GS360_ERR_ARG, "dst[%d] is NULL", k);
        size_t ds = dst_stride ? dst_stride : (size_t)views[k].width * C * esize;
        HIP_TRY(hipMemcpyAsync(dst[k], dptr[k], ds * (size_t)views[k].height, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return GS360_OK;
}
}  // namespace

int gs360_equirect_views_u8_host(gs360_ctx* c, const uint8_t* src, int W, int H, int C, size_t src_stride,
                                 const gs360_view* views, int n_views, uint8_t* const* dst, size_t dst_stride,
                                 int interp, uint32_t flags, int slot) {
    return equirect_views_host_impl(c, src, W, H, C, src_stride, views, n_views, (void* const*)dst, dst_stride, interp, flags, slot, 1);
}
int gs360_equirect_views_u16_host(gs360_ctx* c, const uint16_t* src, int W, int H, int C, size_t src_stride,
                                  const gs360_view* views, int n_views, uint16_t* const* dst, size_t dst_stride,
                                  int interp, uint32_t flags, int slot) {
    return equirect_views_host_impl(c, src, W, H, C, src_stride, views, n_views, (void* const*)dst, dst_stride, interp, flags, slot, 2);
}

namespace {
int remap_table_host_impl(gs360_ctx* c, const void* src, int H, int W, int C, size_t src_stride, const float* map_x,
                          const float* map_y, const uint8_t* valid, int h, int w, int interp,
                          const double* border_value, int fill_value, void* dst, size_t dst_stride, int slot, int esize) {
    if (int rc = check_ctx_slot(c, slot)) return rc;
    if (!src || !map_x || !map_y || !dst) return fail(GS360_ERR_ARG, "NULL argument");
    if (C != 1 && C != 3 && C != 4) return fail(GS360_ERR_ARG, "C must be 1, 3 or 4 (got %d)", C);
    if (H < 1 || W < 1 || h < 0 || w < 0) return fail(GS360_ERR_ARG, "bad size");
    if (h == 0 || w == 0) return GS360_OK;
    if (src_stride == 0) src_stride = (size_t)W * C * esize;
    if (dst_stride == 0) dst_stride = (size_t)w * C * esize;
    HIP_TRY(hipSetDevice(c->device));
    Staging& S = c->stage[slot];
    size_t src_bytes = src_stride * (size_t)H, dst_bytes = dst_stride * (size_t)h;
    size_t npx = (size_t)h * w, map_bytes = npx * sizeof(float);
    size_t map_al = (map_bytes + 255) & ~(size_t)255;
    if (int rc = ensure(c, &S.d_src, &S.src_cap, src_bytes)) return rc;
    if (int rc = ensure(c, &S.d_dst, &S.dst_cap, dst_bytes)) return rc;
    if (int rc = ensure(c, &S.d_aux, &S.aux_cap, 2 * map_al + npx)) return rc;
    hipStream_t st = c->stream[slot];
    float* dmx = (float*)S.d_aux;
    float* dmy = (float*)((uint8_t*)S.d_aux + map_al);
    uint8_t* dva = (uint8_t*)S.d_aux + 2 * map_al;
    HIP_TRY(hipMemcpyAsync(S.d_src, src, src_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dmx, map_x, map_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dmy, map_y, map_bytes, hipMemcpyHostToDevice, st));
    if (valid) HIP_TRY(hipMemcpyAsync(dva, valid, npx, hipMemcpyHostToDevice, st));
    if (int rc = (esize == 2 ? gs360_remap_table_u16 : gs360_remap_table_u8)(c, S.d_src, H, W, C, src_stride, dmx, dmy, valid ? dva : nullptr,
                                                                             h, w, interp, border_value, fill_value, S.d_dst, dst_stride, slot))
        return rc;
    HIP_TRY(hipMemcpyAsync(dst, S.d_dst, dst_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GS360_OK;
}
}  // namespace

int gs360_remap_table_u8_host(gs360_ctx* c, const uint8_t* src, int H, int W, int C, size_t src_stride, const float* map_x,
                              const float* map_y, const uint8_t* valid, int h, int w, int interp,
                              const double* border_value, int fill_value, uint8_t* dst, size_t dst_stride, int slot) {
    return remap_table_host_impl(c, src, H, W, C, src_stride, map_x, map_y, valid, h, w, interp, border_value, fill_value, dst, dst_stride, slot, 1);
}
int gs360_remap_table_u16_host(gs360_ctx* c, const uint16_t* src, int H, int W, int C, size_t src_stride, const float* map_x,
                               const float* map_y, const uint8_t* valid, int h, int w, int interp,
                               const double* border_value, int fill_value, uint16_t* dst, size_t dst_stride, int slot) {
    return remap_table_host_impl(c, src, H, W, C, src_stride, map_x, map_y, valid, h, w, interp, border_value, fill_value, dst, dst_stride, slot, 2);
}

}  // extern "C"
