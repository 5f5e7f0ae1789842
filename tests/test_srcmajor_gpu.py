"""-m gpu: the source-major equirect kernel (gs360_srcmajor.hip) through the C ABI against the CPU oracle, every byte.

It takes calls whose views are yaw rings of one size filling their circle (`--count N`, gs360_360PerspCut.py:794; one ffmpeg v360
process per (frame, view) in the reference, PC:310-314), each ring level or paired with the ring at minus its pitch (the
`full360coverage` and `fisheyelike` presets, PC:616-680).  Forced on with the context option "srcmajor" = 1 so that weakly minified
rings run through it too; `last_eq_kernel` proves which kernel a call launched."""
import numpy as np
import pytest

import gs360
from util import HFOV_12MM, HFOV_14MM, HFOV_17MM, PRESET_FISHEYELIKE, PRESET_FULL360, rand_image, ring_views

pytestmark = pytest.mark.gpu


def _check(ctx, orc, src, specs, what, expect_kernel=2, **kw):
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], **kw)
    assert ctx.get_option("last_eq_kernel") == expect_kernel, f"{what}: kernel {ctx.get_option('last_eq_kernel')}"
    want = orc.equirect_views_u8(src, [orc.make_view(*s) for s in specs], threads=0)
    for k, (g, w) in enumerate(zip(got, want)):
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)
            raise AssertionError(f"{what}: view {k}: {len(bad)} mismatching bytes of {g.size}, first at {bad[0].tolist()}")


@pytest.fixture(params=["lds-copies", "registers"])
def forced(ctx, request):
    """the source-major kernel forced onto every call whose geometry fits it, in both of its staging forms: a loader wavefront copying
    tiles with global_load_lds, and the consumers staging them through registers (option srcmajor_stage)"""
    with ctx.options(srcmajor=1, srcmajor_bx=768, srcmajor_rows=32, srcmajor_stage=1 if request.param == "registers" else 0):
        yield ctx


@pytest.mark.parametrize("count", [2, 3, 4, 5, 6, 8, 12, 16])
def test_ring_counts(forced, orc, count):
    W = 240 * 16                      # divisible by every count above, 3 W / count a multiple of 16
    src = rand_image(W // 2, W, seed=200 + count)
    _check(forced, orc, src, ring_views(count, 120, 100.0), f"ring of {count}")


@pytest.mark.parametrize("w,h", [(8, 2), (12, 7), (200, 201), (64, 333), (400, 96)])
def test_view_sizes_incl_odd_heights(forced, orc, w, h):
    src = rand_image(480, 960, seed=210)
    _check(forced, orc, src, [(i * 60.0, 0.0, 95.0, 70.0, w, h) for i in range(6)], f"{w}x{h} views")


@pytest.mark.parametrize("off", [0.0, 7.3, -123.456, 360.0 / 960 * 17, 179.99])
def test_yaw_offsets_and_seam(forced, orc, off):
    """rings rotated by arbitrary (sub-texel) angles: boxes that run across the 360-degree seam, quads that straddle period borders"""
    src = rand_image(480, 960, seed=211)
    _check(forced, orc, src, [(off + i * 90.0, 0.0, 112.0, 112.0, 160, 120) for i in range(4)], f"offset {off}")


def test_shuffled_view_order_and_frames(forced, orc):
    rng = np.random.default_rng(5)
    W, H, N = 1920, 960, 6
    specs = [ring_views(N, 200, HFOV_12MM)[k] for k in rng.permutation(N)]
    frames = [rand_image(H, W, seed=220 + f) for f in range(3)]
    d_src = [forced.to_device(f) for f in frames]
    dstride = 200 * 3 + 8
    d_out = [forced.alloc(dstride * 200 + 64) for _ in range(3 * N)]
    for b in d_out:
        forced.memset(b, 0xCD)
    forced.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out, dst_stride=dstride)
    forced.sync(0)
    assert forced.get_option("last_eq_kernel") == 2
    for f in range(3):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0)
        for k in range(N):
            raw = forced.download(d_out[f * N + k], (200, dstride))
            assert np.array_equal(raw[:, :600].reshape(200, 200, 3), want[k]), (f, k)
            assert np.all(raw[:, 600:] == 0xCD), "row padding written"
    for b in d_src + d_out:
        forced.free(b)


@pytest.mark.parametrize("bx,rows", [(256, 8), (512, 16), (768, 48), (1024, 24), (2048, 8)])
def test_tile_shapes(ctx, orc, bx, rows):
    src = rand_image(960, 1920, seed=230)
    with ctx.options(srcmajor=1, srcmajor_bx=bx, srcmajor_rows=rows, srcmajor_adapt=0):      # (exactly this shape: no half-height tiles)
        _check(ctx, orc, src, ring_views(6, 240, HFOV_12MM), f"tile {bx} x {rows}")


def test_small_jobs_take_half_height_tiles(ctx, orc):
    """a call that does not fill the GPU once is rendered from tiles of half the height (a second plan of the same geometry); the
    result is the same bytes either way, and both plans stay cached next to each other"""
    src = rand_image(1920, 3840, seed=231)
    specs = ring_views(6, 400, HFOV_12MM)
    for adapt in (1, 0, 1):
        with ctx.options(srcmajor=1, srcmajor_adapt=adapt):
            _check(ctx, orc, src, specs, f"adapt {adapt}")


def test_plan_cache_eviction_and_reuse(forced, orc):
    """more geometries than the context keeps plans for, then the first ones again"""
    src = rand_image(480, 960, seed=240)
    geoms = [[(i * 60.0 + o, 0.0, 100.0 + o, 100.0, 96 + 4 * o, 64) for i in range(6)] for o in range(6)]
    for rnd in range(2):
        for g in geoms:
            _check(forced, orc, src, g, f"geometry round {rnd}")


def test_shapes_it_must_leave_to_the_gather_kernels(forced, orc):
    src = rand_image(480, 960, seed=250)
    # width not a multiple of four; a pitched ring; a ring that does not fill its circle; one ring + a stray view
    _check(forced, orc, src, [(i * 60.0, 0.0, 100.0, 100.0, 98, 64) for i in range(6)], "width 98", expect_kernel=0)
    _check(forced, orc, src, [(i * 60.0, 30.0, 100.0, 100.0, 96, 64) for i in range(6)], "pitched ring", expect_kernel=0)
    _check(forced, orc, src, [(i * 60.0, 0.0, 100.0, 100.0, 96, 64) for i in range(5)], "5 of 6", expect_kernel=0)
    _check(forced, orc, src, [(i * 60.0, 0.0, 100.0, 100.0, 96, 64) for i in range(6)] + [(10.0, 0.0, 100.0, 100.0, 96, 64)], "ring + stray",
           expect_kernel=0)
    # 7 does not divide 960
    _check(forced, orc, src, [(i * 360.0 / 7, 0.0, 100.0, 100.0, 96, 64) for i in range(7)], "count 7", expect_kernel=0)


def test_geometry_that_does_not_fit_is_remembered(ctx, orc):
    """a tile whose two buffers exceed the LDS budget even at a quarter of the asked rows falls back to the gather kernels -- and is not
    planned again on the next call (planning costs tens of milliseconds)"""
    import time
    src = rand_image(1024, 2048, seed=270)
    specs = ring_views(8, 512, 100.0)
    with ctx.options(srcmajor=1, srcmajor_bx=4032, srcmajor_rows=128):
        _check(ctx, orc, src, specs, "oversized tiles", expect_kernel=0)
        views = [gs360.View.make(*s) for s in specs]
        d_src = ctx.to_device(src)
        d_out = [ctx.alloc(512 * 512 * 3) for _ in specs]
        ctx.equirect_views_dev([d_src], 2048, 1024, 3, views, d_out)
        ctx.sync(0)
        builds = ctx.get_option("srcmajor_plan_builds")
        for _ in range(5):
            ctx.equirect_views_dev([d_src], 2048, 1024, 3, views, d_out)
        ctx.sync(0)
        # (a counter, not a clock: the refused geometry is remembered as an empty plan, later calls plan nothing)
        assert ctx.get_option("last_eq_kernel") == 0 and ctx.get_option("srcmajor_plan_builds") == builds
    for b in [d_src] + d_out:
        ctx.free(b)


def _family(pairs, hfov, size):
    return [(float(y), float(p), hfov, hfov, size, size) for y, p in pairs]


RING_FAMILIES = {
    "full360coverage": (1920, 960, _family(PRESET_FULL360, HFOV_14MM, 200)),                 # a level ring of 4 + the +30 / -30 pair
    "fisheyelike": (1920, 960, _family(PRESET_FISHEYELIKE, HFOV_17MM, 256)),                 # 5 rings of 2
    "mirror pair alone": (960, 480, _family([(10 + 90 * i, s * 25) for i in range(4) for s in (1, -1)], 90.0, 120)),
    "level + two pairs, own yaw phases": (1440, 720, _family([(120 * i, 0) for i in range(3)] + [(17.5 + 120 * i, s * 20) for i in range(3) for s in (1, -1)]
                                                              + [(60 + 120 * i, s * 50) for i in range(3) for s in (-1, 1)], 60.0, 96)),
    "rings of 4 + 6 + 6 views = eight rings of two": (960, 480, [(90.0 * i, 0.0, 100.0, 100.0, 96, 96) for i in range(4)]
                                                       + [(60.0 * i, s * 30.0, 100.0, 100.0, 96, 96) for i in range(6) for s in (1, -1)]),
    "odd source height": (960, 479, _family(PRESET_FULL360, 100.0, 100)),
    "rectangular views": (1920, 960, [(float(y), float(p), 100.0, 70.0, 160, 90) for y, p in PRESET_FULL360]),
    "weak minification (tiles cut into several plan tiles)": (960, 480, _family(PRESET_FULL360, HFOV_14MM, 400)),
}


@pytest.mark.parametrize("name", list(RING_FAMILIES))
def test_ring_families(forced, orc, name):
    W, H, specs = RING_FAMILIES[name]
    _check(forced, orc, rand_image(H, W, seed=400 + len(name)), specs, name)


def test_single_ring_at_weak_minification_is_cut_not_refused(forced, orc):
    """0.6 source texels per output pixel: twenty times cfg2's plan entries per tile -- the builder cuts such tiles into plan tiles of
    at most 32 KiB of entries instead of giving up"""
    _check(forced, orc, rand_image(512, 1024, seed=271), ring_views(8, 1024, 150.0), "8 x 1024^2 from 1024 x 512")


def test_ring_family_shuffled_frames_and_row_padding(forced, orc):
    rng = np.random.default_rng(6)
    W, H = 1920, 960
    base = _family(PRESET_FULL360, HFOV_14MM, 200)
    specs = [base[k] for k in rng.permutation(len(base))]
    NV = len(specs)
    frames = [rand_image(H, W, seed=420 + f) for f in range(2)]
    d_src = [forced.to_device(f) for f in frames]
    dstride = 200 * 3 + 12
    d_out = [forced.alloc(dstride * 200 + 64) for _ in range(2 * NV)]
    for b in d_out:
        forced.memset(b, 0xCD)
    forced.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out, dst_stride=dstride)
    forced.sync(0)
    assert forced.get_option("last_eq_kernel") == 2
    for f in range(2):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0)
        for k in range(NV):
            raw = forced.download(d_out[f * NV + k], (200, dstride))
            assert np.array_equal(raw[:, :600].reshape(200, 200, 3), want[k]), (f, k)
            assert np.all(raw[:, 600:] == 0xCD), "row padding written"
    for b in d_src + d_out:
        forced.free(b)


@pytest.mark.parametrize("shape", ["ring", "family"])
def test_more_frames_than_one_launch_holds(forced, orc, shape):
    """18 frames in one call: the library cuts them into launches of GS360_MAX_FRAMES (16 + 2), one plan"""
    W, H = 960, 480
    specs = ring_views(6, 64, 100.0) if shape == "ring" else _family(PRESET_FULL360, 100.0, 64)
    NV, nf = len(specs), 18
    frames = [rand_image(H, W, seed=500 + f) for f in range(nf)]
    d_src = [forced.to_device(f) for f in frames]
    d_out = [forced.alloc(64 * 64 * 3) for _ in range(nf * NV)]
    forced.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out)
    forced.sync(0)
    assert forced.get_option("last_eq_kernel") == 2
    for f in range(nf):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0)
        for k in range(NV):
            assert np.array_equal(forced.download(d_out[f * NV + k], (64, 64, 3)), want[k]), (f, k)
    for b in d_src + d_out:
        forced.free(b)


def test_seventeen_frames_in_auto_mode_keep_one_plan(ctx, orc):
    """a call of 16 + 1 frames with the library's own choices (option srcmajor = -1): the plan is decided once, before the first chunk, and
    held to the last -- the one-frame tail chunk must not pick the half-height plan (whose boxes may exceed the automatic limit) after the
    first sixteen frames have been rendered; every frame of the call against the oracle"""
    W, H = 3840, 1920
    specs = _family(PRESET_FULL360, HFOV_14MM, 400)       # 4K -> full360coverage at 400^2: 2.0 texels per pixel, a ring family
    NV, nf = len(specs), 17
    base = rand_image(H, W, seed=530)
    frames = [np.ascontiguousarray(np.roll(base, 131 * f, axis=1)) for f in range(nf)]
    d_src = [ctx.to_device(f) for f in frames]
    d_out = [ctx.alloc(400 * 400 * 3) for _ in range(nf * NV)]
    with ctx.options(srcmajor=-1):
        ctx.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out)
        ctx.sync(0)
        assert ctx.get_option("last_eq_kernel") == 2
    for f in (0, 7, 15, 16):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0)
        for k in range(NV):
            assert np.array_equal(ctx.download(d_out[f * NV + k], (400, 400, 3)), want[k]), (f, k)
    for b in d_src + d_out:
        ctx.free(b)


def test_rotating_twelve_geometries_from_four_threads(orc):
    """the GUI case: the geometry changes from call to call (a preview slider), from several host threads on their own stream slots, more
    geometries than the cache keeps: evicted plans wait in the graveyard for gs360_sync(ctx, -1) instead of a hipFree (a device-wide
    synchronisation) inside the call -- counted by the read-only option srcmajor_inline_frees -- and every result stays bit-exact"""
    import threading
    ctx4 = gs360.Context(0, n_slots=4)
    src = rand_image(480, 960, seed=540)
    d_src = ctx4.to_device(src)
    geoms = [[(i * 60.0 + 2.5 * t, 0.0, 80.0 + 3 * t, 85.0, 64 + 8 * (t % 5), 72) for i in range(6)] for t in range(24)]
    wants = [orc.equirect_views_u8(src, [orc.make_view(*s) for s in g], threads=0) for g in geoms]
    errors = []
    barrier = threading.Barrier(4)

    def work(t):
        try:
            for rnd in range(3):
                for gi in range(t, len(geoms), 4):
                    g = geoms[gi]
                    d_out = [ctx4.alloc(s[4] * s[5] * 3) for s in g]
                    ctx4.equirect_views_dev([d_src], 960, 480, 3, [gs360.View.make(*s) for s in g], d_out, slot=t)
                    ctx4.sync(t)
                    for k, s in enumerate(g):
                        if not np.array_equal(ctx4.download(d_out[k], (s[5], s[4], 3), slot=t), wants[gi][k]):
                            errors.append((t, gi, k))
                    for b in d_out:
                        ctx4.free(b)
                barrier.wait()
                if t == 0:
                    ctx4.sync(-1)                          # every stream idle: the graveyard is released here
                barrier.wait()
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))
            barrier.abort()
    with ctx4.options(srcmajor=1):
        threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        builds, inline, held = ctx4.get_option("srcmajor_plan_builds"), ctx4.get_option("srcmajor_inline_frees"), ctx4.get_option("srcmajor_plans")
    ctx4.close()
    assert not errors, errors
    assert inline == 0, inline                             # no plan was released inside a call
    assert held <= 16 + 4                                  # the cache's size (+ plans held by calls in flight when it was full)
    assert 24 <= builds <= 3 * 2 * 24, builds              # (each geometry may hold two plans: full- and half-height tiles)


def test_concurrent_callers_share_the_plan_cache(orc):
    """four host threads, each on its own stream slot, render different ring geometries through one context at the same time (plans are
    built, looked up and evicted under the context's plan lock)"""
    import threading
    ctx4 = gs360.Context(0, n_slots=4)
    src = rand_image(480, 960, seed=520)
    d_src = ctx4.to_device(src)
    geoms = [[(i * 60.0 + 3 * t, 0.0, 90.0 + 5 * t, 90.0, 96 + 8 * t, 80) for i in range(6)] for t in range(4)]
    geoms[3] = _family(PRESET_FULL360, 100.0, 64)
    wants = [orc.equirect_views_u8(src, [orc.make_view(*s) for s in g], threads=0) for g in geoms]
    errors = []

    def work(t):
        try:
            g = geoms[t]
            d_out = [ctx4.alloc(s[4] * s[5] * 3) for s in g]
            for _ in range(20):
                ctx4.equirect_views_dev([d_src], 960, 480, 3, [gs360.View.make(*s) for s in g], d_out, slot=t)
            ctx4.sync(t)
            for k, s in enumerate(g):
                if not np.array_equal(ctx4.download(d_out[k], (s[5], s[4], 3), slot=t), wants[t][k]):
                    errors.append((t, k))
        except Exception as e:      # noqa: BLE001
            errors.append((t, repr(e)))
    with ctx4.options(srcmajor=1):
        threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    ctx4.close()
    assert not errors, errors


def _check_masked(ctx, orc, frames, masks, W, H, specs, what, expect_kernel=2):
    """frames / masks: lists of H x W x 3 / H x W uint8 arrays (0 = masked, 255 = keep, any value compared with 128)"""
    nf, NV = len(frames), len(specs)
    d_src = [ctx.to_device(f) for f in frames]
    d_msk = [ctx.to_device(m) for m in masks]
    d_out = [ctx.alloc(s[4] * s[5] * 3) for _ in range(nf) for s in specs]
    ctx.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out, masks=d_msk)
    ctx.sync(0)
    assert ctx.get_option("last_eq_kernel") == expect_kernel, f"{what}: kernel {ctx.get_option('last_eq_kernel')}"
    for f in range(nf):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0, mask=masks[f])
        for k, s in enumerate(specs):
            got = ctx.download(d_out[f * NV + k], (s[5], s[4], 3))
            if not np.array_equal(got, want[k]):
                bad = np.argwhere(got != want[k])
                raise AssertionError(f"{what}: frame {f} view {k}: {len(bad)} mismatching bytes, first at {bad[0].tolist()}")
    for b in d_src + d_msk + d_out:
        ctx.free(b)


def _noise_mask(H, W, seed, p_keep=0.5):
    """per-texel noise: any error in WHICH texel is the nearest one shows"""
    return np.where(np.random.default_rng(seed).random((H, W)) < p_keep, 255, 0).astype(np.uint8)


MASKED_SHAPES = {
    "ring of 6": (1920, 960, ring_views(6, 200, HFOV_12MM)),
    "ring of 6, weak minification": (960, 480, ring_views(6, 240, 100.0)),
    "ring of 4 rotated off the texel grid": (1280, 640, [(7.3 + 90.0 * i, 0.0, 110.0, 110.0, 160, 121) for i in range(4)]),
    "full360coverage": (1920, 960, _family(PRESET_FULL360, HFOV_14MM, 200)),
    "fisheyelike": (1920, 960, _family(PRESET_FISHEYELIKE, HFOV_17MM, 256)),
    "odd source height": (1920, 959, _family(PRESET_FULL360, 100.0, 120)),
}


@pytest.mark.parametrize("name", list(MASKED_SHAPES))
def test_masked_calls(forced, orc, name):
    """the fused keep-mask (BASELINE config 5's fusion, gs360_equirect_views_masked_u8) through the source-major kernel: the keep bits of
    a tile box are staged next to its texels; nearest texel incl. the half-way rows of upside-down images"""
    W, H, specs = MASKED_SHAPES[name]
    frames = [rand_image(H, W, seed=600 + f) for f in range(2)]
    masks = [_noise_mask(H, W, 610 + f) for f in range(2)]
    _check_masked(forced, orc, frames, masks, W, H, specs, name)


def test_masked_calls_it_must_leave_to_the_gather_kernels(forced, orc):
    # ring period not a whole number of keep dwords for any ring size the views fall into (2400 / 4 = 600, / 2 = 1200 texels); source
    # width not a multiple of 32
    for W, n in ((2400, 4), (1200, 3)):
        H = W // 2
        specs = [(i * 360.0 / n, 0.0, 100.0, 100.0, 96, 80) for i in range(n)]
        assert (W // n) % 32 or W % 32
        _check_masked(forced, orc, [rand_image(H, W, seed=620)], [_noise_mask(H, W, 621)], W, H, specs, f"{W} / {n}", expect_kernel=0)


def test_ring_families_it_must_leave_to_the_gather_kernels(forced, orc):
    src = rand_image(480, 960, seed=251)
    lvl = [(90.0 * i, 0.0, 100.0, 100.0, 96, 96) for i in range(4)]
    up = [(45 + 90.0 * i, 30.0, 100.0, 100.0, 96, 96) for i in range(4)]
    # a pitched ring without its mirror (the evenPlus30 preset's shape, PC:616-644)
    _check(forced, orc, src, lvl + up, "level ring + unpaired +30 ring", expect_kernel=0)
    # the mirror ring on other yaws
    _check(forced, orc, src, up + [(90.0 * i, -30.0, 100.0, 100.0, 96, 96) for i in range(4)], "pair on different yaws", expect_kernel=0)
    # rings whose sizes share no divisor (4 + 3 + 3)
    _check(forced, orc, src, lvl + [(120.0 * i, s * 30.0, 100.0, 100.0, 96, 96) for i in range(3) for s in (1, -1)], "4 + 3 + 3", expect_kernel=0)
    # views over the poles (their quads are not monotone in longitude; rows clamp)
    _check(forced, orc, src, [(90.0 * i, s * 60.0, 100.0, 100.0, 96, 96) for i in range(4) for s in (1, -1)], "pair over the poles", expect_kernel=0)
    # one view of the pair with another field of view
    odd = [(45 + 90.0 * i, -30.0, 100.0, 100.0 if i else 90.0, 96, 96) for i in range(4)]
    _check(forced, orc, src, up + odd, "one vfov differs", expect_kernel=0)


def test_auto_selection(ctx, orc):
    """left to itself the library takes it for ONE level ring of >= 6 views from 1.5 source texels per output pixel at any number of frames
    (5 views: from 2.25 and two frames; 4: never) once the call writes >= 3.5 M pixels (profiles/r05/srcmajor_ring_sweep.txt,
    srcmajor_small_jobs.txt)"""
    small = [rand_image(960, 1920, seed=260 + f) for f in range(15)]
    big = [rand_image(1920, 3840, seed=290 + f) for f in range(2)]
    d_small = [ctx.to_device(f) for f in small]
    d_big = [ctx.to_device(f) for f in big]

    def kernel_for(specs, n_frames, big_source=False):
        frames, d_src = (big, d_big) if big_source else (small, d_small)
        H, W = frames[0].shape[:2]
        d_out = [ctx.alloc(s[4] * s[5] * 3) for _ in range(n_frames) for s in specs]
        ctx.equirect_views_dev(d_src[:n_frames], W, H, 3, [gs360.View.make(*s) for s in specs], d_out)
        ctx.sync(0)
        want = orc.equirect_views_u8(frames[n_frames - 1], [orc.make_view(*s) for s in specs], threads=0)
        for k, s in enumerate(specs):
            assert np.array_equal(ctx.download(d_out[(n_frames - 1) * len(specs) + k], (s[5], s[4], 3)), want[k])
        for b in d_out:
            ctx.free(b)
        return ctx.get_option("last_eq_kernel")
    with ctx.options(srcmajor=-1):
        assert kernel_for(ring_views(6, 200, HFOV_12MM), 15) == 2         # 4.6 texels per pixel, 3.6 M pixels
        assert kernel_for(ring_views(6, 200, HFOV_12MM), 14) == 0         # 3.36 M pixels: too small a call
        assert kernel_for(ring_views(6, 400, HFOV_12MM), 4) == 2          # 2.3 texels
        assert kernel_for(ring_views(6, 800, HFOV_12MM), 1) == 0          # 1.15: below 1.5
        assert kernel_for(ring_views(6, 800, HFOV_12MM), 1, True) == 2    # ONE frame (what the engine hands over): 2.3 texels, 3.84 M pixels
        assert ctx.get_option("last_srcmajor_rows") == 16                 # ... on tiles of half the height (the job does not fill the GPU)
        assert kernel_for(ring_views(6, 800, HFOV_12MM), 2, True) == 2
        five = [(72.0 * i, 0.0, 130.0, 130.0, 840, 840) for i in range(5)]
        assert kernel_for(five, 1, True) == 0                             # five views at 3.1 texels: from two frames
        assert kernel_for(five, 2, True) == 2
        assert kernel_for([(72.0 * i, 0.0, 112.0, 112.0, 900, 900) for i in range(5)], 2, True) == 0    # five views at 2.0
        assert kernel_for(ring_views(4, 400, HFOV_12MM), 8) == 0          # four views: never
    with ctx.options(srcmajor=0):
        assert kernel_for(ring_views(6, 200, HFOV_12MM), 15) == 0
    for b in d_small + d_big:
        ctx.free(b)


def test_auto_selection_of_ring_families(ctx, orc):
    """several rings in one call (profiles/r05/srcmajor_family_sweep.txt): taken from four frames per call, eight views and 1.75 source
    texels per output pixel, unless the views reach so close to a pole that the tile boxes outgrow their grid cells"""
    W, H = 1920, 960
    frames = [rand_image(H, W, seed=280 + f) for f in range(4)]
    d_src = [ctx.to_device(f) for f in frames]

    def kernel_for(specs, n_frames):
        d_out = [ctx.alloc(s[4] * s[5] * 3) for _ in range(n_frames) for s in specs]
        ctx.equirect_views_dev(d_src[:n_frames], W, H, 3, [gs360.View.make(*s) for s in specs], d_out)
        ctx.sync(0)
        want = orc.equirect_views_u8(frames[n_frames - 1], [orc.make_view(*s) for s in specs], threads=0)
        for k, s in enumerate(specs):
            assert np.array_equal(ctx.download(d_out[(n_frames - 1) * len(specs) + k], (s[5], s[4], 3)), want[k])
        for b in d_out:
            ctx.free(b)
        return ctx.get_option("last_eq_kernel")
    full = _family(PRESET_FULL360, HFOV_14MM, 400)            # 1.96 texels per pixel
    with ctx.options(srcmajor=-1):
        assert kernel_for(full, 4) == 2
        assert kernel_for(full, 3) != 2                      # fewer than four frames
        assert kernel_for(_family(PRESET_FULL360, HFOV_14MM, 520), 4) != 2                      # 1.51 texels per pixel
        assert kernel_for(_family(PRESET_FISHEYELIKE, HFOV_17MM, 256), 4) == 2                  # five rings of two at 2.5
        assert kernel_for(_family([(120 * i, s * 30) for i in range(3) for s in (1, -1)], 100.0, 200), 4) != 2                 # a pair of three: six views
        poles = [(90.0 * i, 0.0, 90.0, 90.0, 256, 256) for i in range(4)] + [(45 + 90.0 * i, s * 45.0, 90.0, 90.0, 256, 256) for i in range(4) for s in (1, -1)]
        with ctx.options(srcmajor_rows=8):                   # (at 8K the builder itself ends at 8-row tiles for such views: 169 %)
            assert kernel_for(poles, 4) != 2                 # top edges on the poles: boxes ~180 % of their cells
            assert ctx.get_option("last_srcmajor_box_pct") > 160
        assert kernel_for(poles, 4) == 2 and ctx.get_option("last_srcmajor_box_pct") <= 160      # 16-row tiles at this size: ~140 %
    for b in d_src:
        ctx.free(b)


def test_cfg2_full_size_sixteen_frames_every_byte(ctx, orc):
    """BASELINE configs[1] as bench.py launches it: 16 distinct 8K frames x 6 x 800^2 in one call, every byte of all 96 views"""
    W, H, N = 7680, 3840, 6
    specs = ring_views(N, 800, HFOV_12MM)
    frames = [rand_image(H, W, seed=300 + f) for f in range(16)]
    d_src = [ctx.to_device(f) for f in frames]
    d_out = [ctx.alloc(800 * 800 * 3) for _ in range(16 * N)]
    with ctx.options(srcmajor=-1):
        ctx.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out)
        ctx.sync(0)
        assert ctx.get_option("last_eq_kernel") == 2
    for f in range(16):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0)
        for k in range(N):
            got = ctx.download(d_out[f * N + k], (800, 800, 3))
            assert np.array_equal(got, want[k]), f"frame {f} view {k}"
    for b in d_src + d_out:
        ctx.free(b)


def test_cfg3_full_size_four_frames_every_byte(ctx, orc):
    """BASELINE configs[2]'s view set as the resident-job bench launches it: 8K frames x full360coverage 12 x 1600^2 (a level ring of
    four and the +30 / -30 pair), automatic selection, every byte of all 48 views"""
    W, H = 7680, 3840
    specs = _family(PRESET_FULL360, HFOV_14MM, 1600)
    NV = len(specs)
    frames = [rand_image(H, W, seed=320 + f) for f in range(4)]
    d_src = [ctx.to_device(f) for f in frames]
    d_out = [ctx.alloc(1600 * 1600 * 3) for _ in range(4 * NV)]
    with ctx.options(srcmajor=-1):
        ctx.equirect_views_dev(d_src, W, H, 3, [gs360.View.make(*s) for s in specs], d_out)
        ctx.sync(0)
        assert ctx.get_option("last_eq_kernel") == 2
    for f in range(4):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0)
        for k in range(NV):
            got = ctx.download(d_out[f * NV + k], (1600, 1600, 3))
            assert np.array_equal(got, want[k]), f"frame {f} view {k}"
    for b in d_src + d_out:
        ctx.free(b)
