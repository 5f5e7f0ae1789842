"""-m gpu parity tests: HIP path (through the C ABI) vs the CPU oracle, bit-exact uint8."""
import numpy as np
import pytest

import gs360
from util import (FULL_CALIB, HFOV_12MM, HFOV_14MM, HFOV_17MM, PRESET_FISHEYELIKE, PRESET_FULL360, TEMPLATE_CALIB,
                  rand_image, ring_views)

pytestmark = pytest.mark.gpu


def _eq_both(ctx, orc, src, specs):
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs])
    want = orc.equirect_views_u8(src, [orc.make_view(*s) for s in specs], threads=0)
    return got, want


def _assert_same(got, want, what):
    assert len(got) == len(want)
    for k, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape
        if not np.array_equal(g, w):
            bad = np.argwhere(g != w)
            d = np.abs(g.astype(int) - w.astype(int))
            raise AssertionError(f"{what}: view {k}: {len(bad)} mismatching bytes of {g.size}, max |d|={d.max()}, "
                                 f"first at {bad[0].tolist()}")


# ---- equirect ---------------------------------------------------------------------------------
@pytest.fixture(params=["rows", "blocked", "auto", "staged", "srcmajor"])
def lanemap(request, ctx):
    """the equirect kernel has two lane maps (64-pixel rows / 4x16 patches) chosen per view on the host by the minification; the context
    option "lanemap" forces one so that every shape below is checked under both.  "staged" = the LDS-staged kernel forced on (option
    "stage" = 1; 16 x 16 wavefront tiles, boxes copied into LDS, gather form where a box does not qualify); "srcmajor" = the source-major
    kernel forced onto every call whose geometry fits it (level yaw rings that fill their circle), everything else as in "auto"."""
    want = {"rows": dict(lanemap=0), "blocked": dict(lanemap=1), "auto": dict(srcmajor=-1), "staged": dict(lanemap=0, stage=1),
            "srcmajor": dict(srcmajor=1)}[request.param]
    base = dict(lanemap=-1, stage=-1, srcmajor=0)      # the named gather / staged variants are not pre-empted by the source-major kernel
    base.update(want)
    with ctx.options(**base):
        yield request.param


def test_equirect_cfg2_ring_small_source(ctx, orc, lanemap):
    src = rand_image(480, 960)
    got, want = _eq_both(ctx, orc, src, ring_views(6, 200, HFOV_12MM))
    _assert_same(got, want, "cfg2-shaped ring on 960x480")
    # which kernel ran: the source-major one when forced (left to itself the library takes it for calls of >= 2 frames; this is one)
    assert ctx.get_option("last_eq_kernel") == {"srcmajor": 2, "staged": 1}.get(lanemap, 0)


@pytest.mark.parametrize("name,layout,hfov", [("full360coverage", PRESET_FULL360, HFOV_14MM),
                                              ("fisheyelike", PRESET_FISHEYELIKE, HFOV_17MM)])
def test_equirect_presets_pitched(ctx, orc, name, layout, hfov, lanemap):
    src = rand_image(512, 1024, seed=7)
    specs = [(y, p, hfov, hfov, 160, 160) for y, p in layout]
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, name)


def test_equirect_poles_seam_and_odd_shapes(ctx, orc, lanemap):
    src = rand_image(301, 602, seed=3)  # odd height, width not a multiple of 4
    specs = [(0, 90, 100, 100, 96, 96), (0, -90, 100, 100, 96, 96), (180, 0, 120, 90, 130, 70),
             (-179.9, 45, 60, 60, 33, 47), (37.3, -62.1, 150, 140, 101, 99), (12, 5, 1, 1, 16, 16),
             (0, 0, 179.9, 179.9, 64, 64), (720.5, 0, 90, 90, 31, 5), (90, 89.999, 90, 90, 40, 40)]
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, "poles/seam/odd")


def test_equirect_tallest_source_the_kernels_take(ctx, orc):
    """a flipped ring member's latitude is formed with a 24-bit multiply-add (32 H < 2^23): the tallest source the C ABI admits,
    H = 2^18 - 1, one view pair that forms a ring with a flipped member -- and one row more is refused, not mis-sampled
    (round-3 ADVICE, gs360_kernels.hip:1130)"""
    H, W = (1 << 18) - 1, 8
    src = np.random.default_rng(5).integers(0, 256, (H, W, 1), dtype=np.uint8)
    specs = [(0, 40, 50, 50, 48, 48), (0, -40, 50, 50, 48, 48), (90, 89, 70, 70, 32, 32)]
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, "H = 2^18 - 1")
    with pytest.raises(gs360.Gs360Error) as exc:
        ctx.equirect_views(np.zeros((H + 1, W, 1), np.uint8), [gs360.View.make(*specs[0])])
    assert exc.value.code == -4


@pytest.mark.parametrize("W,H", [(8, 4), (8, 64), (12, 7), (20, 10), (44, 22)])
def test_equirect_sources_narrower_than_a_staged_box_row(ctx, orc, W, H, lanemap):
    """sources whose row (24-132 bytes) is shorter than the 32-byte-granular pitch of a staged box: the staged kernel must fall back
    to the gather form rather than copy past the frame; exact-size device buffers so an overrun would fault or mis-sample"""
    src = rand_image(H, W, seed=W * 100 + H)
    specs = [(0, 0, 90, 90, 48, 48), (30, 35, 8, 8, 33, 17), (-100, -60, 2, 2, 16, 16), (179, 0, 1, 1, 64, 64), (10, 80, 40, 40, 16, 32)]
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, f"{W}x{H} source")


@pytest.mark.parametrize("channels", [1, 4])
def test_equirect_channels(ctx, orc, channels):
    src = rand_image(256, 512, c=channels, seed=11)
    specs = [(30, 10, 100, 80, 120, 88), (-150, -40, 90, 90, 64, 64)]
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, f"C={channels}")


def test_equirect_full_size_8k_default6(ctx, orc, lanemap):
    """BASELINE cfg2 at full size: 7680x3840 -> 6 x 800^2, every byte."""
    src = rand_image(3840, 7680)
    got, want = _eq_both(ctx, orc, src, ring_views(6, 800, HFOV_12MM))
    _assert_same(got, want, "cfg2 full size")


def test_equirect_mirror_symmetry_edge_shapes(ctx, orc, lanemap):
    """the kernel computes half of each row and mirrors it (and the top half of level views): odd widths/heights,
    widths around the 64/128 tile edges, 1-pixel views, level and pitched"""
    src = rand_image(257, 514, seed=17)
    specs = []
    for w, h in [(1, 1), (2, 3), (3, 2), (63, 5), (64, 7), (65, 9), (127, 16), (128, 17), (129, 33), (130, 31), (255, 8), (257, 15)]:
        specs.append((25.0, 0.0, 110.0, 95.0, w, h))       # level
        specs.append((-160.0, 22.5, 110.0, 95.0, w, h))    # pitched
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, "mirror edge shapes")
    for channels in (1, 4):
        srcc = rand_image(130, 260, c=channels, seed=18)
        sp = [(10.0, 0.0, 100.0, 100.0, 67, 21), (10.0, -35.0, 100.0, 100.0, 66, 20), (180.0, 0.0, 120.0, 60.0, 129, 3)]
        got, want = _eq_both(ctx, orc, srcc, sp)
        _assert_same(got, want, f"mirror edge shapes C={channels}")


@pytest.mark.parametrize("interp", [1, 2])
def test_equirect_fused_keep_mask(ctx, orc, interp, lanemap):
    """BASELINE config 5's fused mask multiply (SegmentationMaskTool convention: 0 = masked, 255 = keep)"""
    H, W = 240, 480
    frames = [rand_image(H, W, seed=500 + f) for f in range(2)]
    rng = np.random.default_rng(501)
    masks = []
    for f in range(2):
        m = np.full((H, W), 255, np.uint8)
        for _ in range(12):                                   # disks of zeros, seed-fixed
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(8, 40)
            yy, xx = np.ogrid[:H, :W]
            m[(yy - cy) ** 2 + (np.minimum(abs(xx - cx), W - abs(xx - cx))) ** 2 <= r * r] = 0
        m[rng.integers(0, H, 50), rng.integers(0, W, 50)] = rng.integers(100, 160, 50).astype(np.uint8)   # around the threshold
        masks.append(m)
    specs = [(0, 0, 110, 110, 130, 70), (180, 0, 100, 100, 65, 65), (-75.5, 33, 90, 120, 67, 129), (0, 90, 120, 120, 64, 64)]
    views = [gs360.View.make(*s) for s in specs]
    dfr = [ctx.to_device(f) for f in frames]
    dms = [ctx.to_device(m) for m in masks]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in frames for s in specs]
    ctx.equirect_views_dev(dfr, W, H, 3, views, dsts, interp=interp, masks=dms)
    ctx.sync(0)
    for f in range(2):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], interp=interp, mask=masks[f])
        got = [ctx.download(dsts[f * len(specs) + k], (s[5], s[4], 3)) for k, s in enumerate(specs)]
        _assert_same(got, want, f"masked frame {f} interp={interp}")
        assert any((g == 0).all(axis=2).mean() > 0.02 for g in got)      # the mask really removed pixels
    for b in dfr + dms + dsts:
        ctx.free(b)


def test_equirect_batched_frames_device_api(ctx, orc, lanemap):
    """n_frames x n_views in ONE launch through the device-pointer entry point."""
    H, W = 300, 600
    frames = [rand_image(H, W, seed=100 + f) for f in range(3)]
    specs = ring_views(5, 96, 100.0) + [(10, 40, 100, 100, 96, 96)]
    views = [gs360.View.make(*s) for s in specs]
    dfr = [ctx.to_device(f) for f in frames]
    dsts = [ctx.alloc(96 * 96 * 3) for _ in range(len(frames) * len(views))]
    ctx.equirect_views_dev(dfr, W, H, 3, views, dsts, slot=1)
    ctx.sync(1)
    for f, fr in enumerate(frames):
        want = orc.equirect_views_u8(fr, [orc.make_view(*s) for s in specs])
        got = [ctx.download(dsts[f * len(views) + k], (96, 96, 3), slot=1) for k in range(len(views))]
        _assert_same(got, want, f"frame {f}")
    for b in dfr + dsts:
        ctx.free(b)


@pytest.mark.parametrize("channels,interp", [(3, 1), (3, 2), (1, 1), (4, 2)])
def test_equirect_fisheye_output(ctx, orc, channels, interp):
    """GS360_EQ_FISHEYE_OUT: the fisheyeXY preset's equidistant-fisheye views (PC:351-414), incl. fov > 180 and odd sizes"""
    src = rand_image(300, 600, c=channels, seed=29)
    d180 = 180.0 / np.sqrt(2.0)
    specs = [(0.0, 0.0, d180, d180, 150, 150), (180.0, 0.0, d180, d180, 150, 150), (37.0, -25.0, 200.0, 120.0, 131, 77),
             (-100.0, 60.0, 254.0, 254.0, 64, 64), (5.0, 0.0, 90.0, 45.0, 1, 3)]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], interp=interp, flags=gs360.EQ_FISHEYE_OUT)
    want = orc.equirect_fisheye_views_u8(src, [orc.make_view(*s) for s in specs], interp=interp)
    _assert_same(got, want, f"fisheye output C={channels} interp={interp}")
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(src, [gs360.View.make(*specs[0])], flags=2)


def test_equirect_many_views_split(ctx, orc):
    """more than GS360_MAX_VIEWS views -> split into several launches internally"""
    src = rand_image(200, 400, seed=5)
    specs = ring_views(30, 40, 80.0)
    got, want = _eq_both(ctx, orc, src, specs)
    _assert_same(got, want, "30 views")


def test_equirect_empty_and_errors(ctx):
    src = rand_image(64, 128)
    assert ctx.equirect_views(src, []) == []
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(src[:, :, :2], [gs360.View.make(0, 0, 90, 90, 8, 8)])  # C == 2
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(src, [gs360.View.make(0, 0, 90, 90, 0, 8)])
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(src, [gs360.View.make(float("nan"), 0, 90, 90, 8, 8)])
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(src, [gs360.View.make(0, 0, 90, 90, 8, 8)], interp=gs360.INTERP_NEAREST)


@pytest.mark.parametrize("channels", [1, 3, 4])
def test_equirect_cubic(ctx, orc, channels, lanemap):
    """cubic sampler (the reference's default interp, PC:730): wrap columns / clamp rows, OpenCV fixed-point table"""
    src = rand_image(193, 386, c=channels, seed=71)
    specs = [(0, 0, 110, 110, 130, 70), (180, 0, 100, 100, 65, 65), (-75.5, 33, 90, 120, 67, 129), (0, 90, 120, 120, 64, 64),
             (10, -89, 60, 60, 33, 31)]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], interp=gs360.INTERP_CUBIC)
    want = orc.equirect_views_u8(src, [orc.make_view(*s) for s in specs], interp=2)
    _assert_same(got, want, f"equirect cubic C={channels}")


def test_equirect_cubic_persistent_walk_probe(ctx, orc):
    """option "eq_persist" (a probe: the C ABI ships the cubic equirect kernels with one tile per workgroup): a capped
    grid whose workgroups walk the tile order must render the same bytes, 8- and 16-bit."""
    with ctx.options(eq_persist=8):
        _cubic_persistent_walk(ctx, orc)


def _cubic_persistent_walk(ctx, orc):
    src = rand_image(193, 386, c=3, seed=72)
    specs = [(0, 0, 110, 110, 330, 170), (90, 0, 110, 110, 330, 170), (-75.5, 33, 90, 120, 167, 229), (10, -89, 60, 60, 133, 131)]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], interp=gs360.INTERP_CUBIC)
    want = orc.equirect_views_u8(src, [orc.make_view(*s) for s in specs], interp=2)
    _assert_same(got, want, "equirect cubic, persistent probe")
    src16 = (np.random.default_rng(73).integers(0, 65536, size=(97, 194, 3))).astype(np.uint16)
    got16 = ctx.equirect_views(src16, [gs360.View.make(*s) for s in specs[:2]], interp=gs360.INTERP_CUBIC)
    want16 = orc.equirect_views_u16(src16, [orc.make_view(*s) for s in specs[:2]], interp=2)
    _assert_same(got16, want16, "equirect cubic u16, persistent probe")


# ---- table remap (cv2.remap semantics) ----------------------------------------------------------
def _rand_maps(h, w, H, W, seed, spread=12.0):
    rng = np.random.default_rng(seed)
    mx = rng.uniform(-spread, W + spread, size=(h, w)).astype(np.float32)
    my = rng.uniform(-spread, H + spread, size=(h, w)).astype(np.float32)
    # exact half-bucket ties, integers, borders
    mx[0, : min(w, 8)] = np.array([0.0, -1.0, W - 1.0, W - 0.5, 1 / 64, 3 / 64, -0.015625, W + 5.0], np.float32)[: min(w, 8)]
    my[0, : min(w, 8)] = np.array([0.0, -1.0, H - 1.0, H - 0.5, 1 / 64, 3 / 64, -0.015625, 2.0], np.float32)[: min(w, 8)]
    return mx, my


@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("interp", [0, 1, 2, 4])
def test_table_remap_random_maps(ctx, orc, channels, interp):
    H, W, h, w = 97, 131, 75, 108
    src = rand_image(H, W, c=channels, seed=21)
    mx, my = _rand_maps(h, w, H, W, seed=22)
    mx[3, 5] = np.nan
    my[4, 6] = np.inf
    mx[5, 7] = -3e9
    my[6, 8] = 1e30
    valid = np.random.default_rng(23).random((h, w)) > 0.1
    bv = (37.0, 0.0, 0.0, 0.0)
    got = ctx.remap(src, mx, my, interpolation=interp, border_value=bv, valid=valid, fill_value=200)
    want = orc.remap_u8(src, mx, my, interp=interp, border_value=bv)
    want = orc.valid_fill(want.copy(), valid, 200)
    _assert_same([got.reshape(h, w, channels)], [want.reshape(h, w, channels)], f"table C={channels} interp={interp}")


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("interp", [0, 1, 2, 4])
def test_table_remap_map_plans(ctx, orc, channels, interp, dtype):
    """map plans (the float maps packed once: clamped 1/32-pixel fixed point + valid bit, 5 bytes per pixel) give cv2.remap's results:
    random maps reaching 40 pixels outside the source, NaN / inf / huge values, exact ties, with and without the valid fill, two
    jobs in one launch; plus what a plan refuses"""
    H, W = 97, 131
    src = rand_image(H, W, c=channels, seed=31)
    if dtype == np.uint16:
        src = (src.astype(np.uint16) << 8) | rand_image(H, W, c=channels, seed=30)
    esz = np.dtype(dtype).itemsize
    oracle_remap = orc.remap_u16 if dtype == np.uint16 else orc.remap_u8
    d_src = ctx.to_device(src)
    bv = (37.0, 0.0, 0.0, 0.0)
    jobs, plans, wants, bufs = [], [], [], []
    # (widths that are / are not multiples of four, a last group of 1-3 pixels, a map narrower than one group of four pixels)
    for k, (h, w, use_valid) in enumerate([(75, 108, True), (33, 200, False), (41, 77, True), (23, 254, True), (9, 3, False)]):
        mx, my = _rand_maps(h, w, H, W, seed=32 + k, spread=40.0)
        if w > 8:
            mx[3, 5] = np.nan
            my[4, 6] = np.inf
            mx[5, 7] = -3e9
            my[6, 8] = 1e30
            mx[7, :8] = np.array([-8.0, -8.03125, -9.0, W + 7.96875, W + 8.0, W + 9.0, 4087.0, 4088.5], np.float32)   # around the clamp
        valid = np.random.default_rng(40 + k).random((h, w)) > 0.1
        d = [ctx.to_device(mx), ctx.to_device(my), ctx.to_device(valid.astype(np.uint8))]
        plan = ctx.map_plan(d[0], d[1], d[2], h, w, nearest=(interp == 0))
        for b in d:
            ctx.free(b)                                    # a plan keeps nothing of its inputs
        dst = ctx.alloc(h * w * channels * esz)
        bufs.append(dst)
        plans.append(plan)
        jobs.append((d_src, H, W, plan, use_valid, h, w, 200, dst))
        want = oracle_remap(src, mx, my, interp=interp, border_value=bv)
        wants.append(orc.valid_fill(want.copy(), valid, 200) if use_valid else want)
    ctx.remap_plans_dev(jobs, channels, interp=interp, border_value=bv, dtype=dtype)
    for k, (job, want) in enumerate(zip(jobs, wants)):
        got = ctx.download(job[8], (job[5], job[6], channels), dtype=dtype)
        _assert_same([got], [want.reshape(got.shape)], f"map plan job {k} C={channels} interp={interp} {np.dtype(dtype).name}")
    # refusals: a plan packed for the other sampler class, another map size, a source too large for the packed positions
    other = 1 if interp == 0 else 0
    with pytest.raises(gs360.Gs360Error):
        ctx.remap_plans_dev(jobs[:1], channels, interp=other, border_value=bv, dtype=dtype)
    with pytest.raises(gs360.Gs360Error):
        ctx.remap_plans_dev([(d_src, H, W, plans[0], False, 74, 108, 0, bufs[0])], channels, interp=interp, border_value=bv, dtype=dtype)
    with pytest.raises(gs360.Gs360Error) as exc:
        ctx.remap_plans_dev([(d_src, 2, 4080, plans[0], False, 75, 108, 0, bufs[0])], channels, interp=interp, border_value=bv, dtype=dtype)
    assert exc.value.code == -4
    for pl in plans:
        ctx.map_plan_free(pl)
    for b in bufs:
        ctx.free(b)
    ctx.free(d_src)


@pytest.mark.parametrize("persist", ["0", "8", "24"])
def test_bicubic_persistent_workgroups_walk_every_tile(ctx, orc, persist):
    """The bicubic RGB kernels (table + fused fisheye) cap their grid and let each workgroup walk tiles b, b + gridDim.x, ...;
    the option "table_persist" forces the cap (0 = one tile per workgroup, 8 / 24 = 13-40 tiles per workgroup here)."""
    with ctx.options(table_persist=int(persist)):
        _bicubic_persistent(ctx, orc, persist)


def _bicubic_persistent(ctx, orc, persist):
    H, W, h, w = 211, 300, 150, 333                      # 6 x 10 tiles per job
    src = rand_image(H, W, c=3, seed=71)
    d_src = ctx.to_device(src)
    jobs, want, keep = [], [], []
    for k in range(5):
        mx, my = _rand_maps(h - 7 * k, w - 11 * k, H, W, seed=72 + k)
        d = (ctx.to_device(mx), ctx.to_device(my), ctx.alloc(mx.size * 3))
        keep.append(d)
        jobs.append((d_src, H, W, d[0], d[1], None, mx.shape[0], mx.shape[1], 0, d[2]))
        want.append(orc.remap_u8(src, mx, my, interp=2, border_value=(5, 0, 0, 0)))
    ctx.remap_tables_dev(jobs, 3, interp=2, border_value=(5, 0, 0, 0))
    for k, (job, w_) in enumerate(zip(jobs, want)):
        assert np.array_equal(ctx.download(job[9], w_.shape), w_), f"table job {k} persist={persist}"
    for d in keep:
        for b in d:
            ctx.free(b)
    ctx.free(d_src)
    # the fused kernel shares the switch
    kw = dict(TEMPLATE_CALIB)
    kw.update(width=480, height=480, f=kw["f"] / 8)
    ocal, gcal = orc.make_calib(**kw), gs360.Calib.make(**kw)
    fsrc = rand_image(480, 480, seed=77)
    specs = [(0, 0, HFOV_14MM, HFOV_14MM, 200, 170), (40, 10, HFOV_14MM, HFOV_14MM, 130, 250), (-72.5, 33.25, 75.0, 110.0, 190, 66)]
    dsrc = ctx.to_device(fsrc)
    dsts = [ctx.alloc(s[4] * s[5] * 3) for s in specs]
    ctx.fisheye_views_dev([dsrc] * len(specs), [gcal] * len(specs), 3, [gs360.View.make(*s) for s in specs], 190.0, dsts,
                          interp=2, mask_outside=True, mask_value=9)
    ctx.sync(0)
    for k, s_ in enumerate(specs):
        mx, my, valid = orc.fisheye_spec_map(ocal, s_[0], s_[1], s_[2], s_[3], s_[4], s_[5], 190.0)
        want_f = orc.valid_fill(orc.remap_u8(fsrc, mx, my, interp=2, border_value=9.0), valid, 9)
        _assert_same([ctx.download(dsts[k], (s_[5], s_[4], 3))], [want_f], f"fused view {k} persist={persist}")
    for b in [dsrc] + dsts:
        ctx.free(b)


def test_table_remap_batched_jobs(ctx, orc):
    """gs360_remap_tables_u8: 19 jobs (more than one launch's 16) of different sizes, two sources, empty map included"""
    rng = np.random.default_rng(23)
    srcs = [rand_image(90, 120, seed=1), rand_image(33, 260, seed=2)]
    d_srcs = [ctx.to_device(s) for s in srcs]
    jobs, want, keep = [], [], []
    for k in range(19):
        si = k % 2
        H, W, _ = srcs[si].shape
        h, w = (0, 5) if k == 7 else (int(rng.integers(1, 70)), int(rng.integers(1, 150)))
        mx = rng.uniform(-6, W + 6, (h, w)).astype(np.float32)
        my = rng.uniform(-6, H + 6, (h, w)).astype(np.float32)
        valid = (rng.random((h, w)) > 0.3) if k % 3 else None
        fill = int(rng.integers(0, 256))
        d = (ctx.to_device(mx) if h else ctx.alloc(4), ctx.to_device(my) if h else ctx.alloc(4),
             ctx.to_device(np.ascontiguousarray(valid, np.uint8)) if valid is not None and h else None, ctx.alloc(max(h * w * 3, 4)))
        keep.append(d)
        jobs.append((d_srcs[si], H, W, d[0], d[1], d[2], h, w, fill, d[3]))
        ref = orc.remap_u8(srcs[si], mx, my, interp=1, border_value=(9, 8, 7, 0)) if h else np.zeros((0, w, 3), np.uint8)
        want.append(orc.valid_fill(ref.copy(), valid, fill) if valid is not None and h else ref)
    ctx.remap_tables_dev(jobs, 3, interp=1, border_value=(9, 8, 7, 0))
    for k, (job, w_) in enumerate(zip(jobs, want)):
        if job[6]:
            assert np.array_equal(ctx.download(job[9], w_.shape), w_), k
    for d in keep:
        for b in d:
            if b is not None:
                ctx.free(b)
    for b in d_srcs:
        ctx.free(b)


def test_table_remap_gray_2d_and_scalar_border(ctx, orc):
    src = rand_image(64, 80, c=1, seed=31)[:, :, 0]
    mx, my = _rand_maps(50, 44, 64, 80, seed=32)
    got = ctx.remap(src, mx, my, interpolation=1, border_value=255.0)
    want = orc.remap_u8(src, mx, my, interp=1, border_value=255.0)
    assert got.shape == (50, 44)
    assert np.array_equal(got, want)


def test_table_remap_identity_and_shifts(ctx):
    """hand-derivable known answers straight on the GPU path"""
    src = rand_image(40, 64, seed=41)
    yy, xx = np.meshgrid(np.arange(40, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
    assert np.array_equal(ctx.remap(src, xx, yy), src)
    assert np.array_equal(ctx.remap(src, xx + np.float32(1 / 64), yy), src)          # 32x+0.5 -> even -> fx=0
    half = ctx.remap(src, xx + np.float32(0.5), yy, border_value=(0, 0, 0, 0))
    want = (src[:, :-1].astype(int) + src[:, 1:].astype(int) + 1) >> 1
    assert np.array_equal(half[:, :-1], want)
    assert np.array_equal(half[:, -1], (src[:, -1].astype(int) + 1) >> 1)             # right tap = border 0


@pytest.mark.parametrize("interp", [2, 4])
def test_table_remap_every_phase_and_integer_grid(ctx, orc, interp):
    """bicubic / Lanczos-4 RGB on windows inside the image: every one of the 32 x 32 sub-pixel phases (the Lanczos kernel rebuilds
    its 2-D weights per pixel from the 1-D table; phase 0 and the patched block of each phase are its special cases), the integer
    grid (phase 0 everywhere: identity), and the same through the table-reading path (option "lanczos_table")."""
    H, W = 96, 131
    src = rand_image(H, W, c=3, seed=91)
    fy, fx = np.meshgrid(np.arange(32, dtype=np.float32), np.arange(32, dtype=np.float32), indexing="ij")
    reps = 5                                              # 160 x 160 pixels: five anchors per phase
    ay = np.tile(np.repeat(np.arange(reps), 32), (reps * 32, 1)).T.astype(np.float32)
    ax = np.tile(np.repeat(np.arange(reps), 32), (reps * 32, 1)).astype(np.float32)
    mx = (10.0 + 17.0 * ax + np.tile(fx, (reps, reps)) / 32.0).astype(np.float32)
    my = (9.0 + 13.0 * ay + np.tile(fy, (reps, reps)) / 32.0).astype(np.float32)
    want = orc.remap_u8(src, mx, my, interp=interp, border_value=(3, 0, 0, 0))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    ident = orc.remap_u8(src, xx, yy, interp=interp, border_value=(3, 0, 0, 0))
    inner = (slice(4, H - 5), slice(4, W - 7))
    assert np.array_equal(ident[inner], src[inner])       # OpenCV's table keeps integer positions exact
    for table_path in (0, 1):
        with ctx.options(lanczos_table=table_path):
            got = ctx.remap(src, mx, my, interpolation=interp, border_value=(3, 0, 0, 0))
            _assert_same([got], [want], f"all phases interp={interp} table_path={table_path!r}")
            assert np.array_equal(ctx.remap(src, xx, yy, interpolation=interp, border_value=(3, 0, 0, 0)), ident)


def test_table_remap_fisheye_maps_from_oracle(ctx, orc):
    """cfg4-shaped: template calibration, oracle-built DF maps, 3 SFM10 views at reduced size"""
    cal = orc.make_calib(**{**TEMPLATE_CALIB, "width": 960, "height": 960, "f": TEMPLATE_CALIB["f"] / 4})
    src = rand_image(960, 960, seed=51)
    for yaw, pitch in [(0, 0), (0, 40), (40, 0), (140, 0)]:
        mx, my, valid = orc.fisheye_map(cal, yaw, pitch, HFOV_14MM, HFOV_14MM, 350, 350, 190.0)
        got = ctx.remap(src, mx, my, interpolation=1, border_value=0.0, valid=valid, fill_value=0)
        want = orc.valid_fill(orc.remap_u8(src, mx, my, interp=1, border_value=0.0), valid, 0)
        _assert_same([got], [want], f"fisheye table yaw={yaw} pitch={pitch}")


# ---- fused fisheye (FE-SPEC v1) ---------------------------------------------------------------
@pytest.mark.parametrize("calib_kw,size", [(TEMPLATE_CALIB, 3840), (FULL_CALIB, None)])
@pytest.mark.parametrize("interp", [0, 1, 2, 4])
def test_fisheye_fused_vs_oracle_spec(ctx, orc, calib_kw, size, interp):
    kw = dict(calib_kw)
    if size:  # shrink the template sensor 8x to keep the test light
        kw.update(width=480, height=480, f=kw["f"] / 8)
    ocal = orc.make_calib(**kw)
    gcal = gs360.Calib.make(**kw)
    H, W = kw["height"], kw["width"]
    src = rand_image(H, W, seed=61)
    specs = [(0, 0, HFOV_14MM, HFOV_14MM, 120, 120), (0, 40, HFOV_14MM, HFOV_14MM, 120, 120),
             (40, 0, HFOV_14MM, HFOV_14MM, 120, 120), (140, 0, HFOV_14MM, HFOV_14MM, 120, 120),
             (-72.5, 33.25, 75.0, 110.0, 130, 66), (10, 5, 30, 20, 47, 33)]
    dsrc = ctx.to_device(src)
    dsts = [ctx.alloc(s[4] * s[5] * 3) for s in specs]
    vouts = [ctx.alloc(s[4] * s[5]) for s in specs]
    ctx.fisheye_views_dev([dsrc] * len(specs), [gcal] * len(specs), 3, [gs360.View.make(*s) for s in specs], 190.0, dsts,
                          valid_outs=vouts, interp=interp, mask_outside=True, mask_value=9)
    ctx.sync(0)
    for k, s in enumerate(specs):
        mx, my, valid = orc.fisheye_spec_map(ocal, s[0], s[1], s[2], s[3], s[4], s[5], 190.0)
        want = orc.valid_fill(orc.remap_u8(src, mx, my, interp=interp, border_value=9.0), valid, 9)
        got = ctx.download(dsts[k], (s[5], s[4], 3))
        gv = ctx.download(vouts[k], (s[5], s[4]))
        assert np.array_equal(gv.astype(bool), valid), f"valid mask view {k}"
        _assert_same([got], [want], f"fused fisheye view {k} interp={interp}")
    for b in [dsrc] + dsts + vouts:
        ctx.free(b)


def test_frame_pipeline_pinned_streams(ctx, orc):
    """host-fed pipeline (pinned H2D -> kernel -> D2H, 2 frames in flight) returns every frame's views intact"""
    from gs360.stream import FramePipeline
    H, W = 256, 512
    specs = ring_views(4, 64, 100.0) + [(20, -30, 100, 100, 64, 64)]
    views = [gs360.View.make(*s) for s in specs]
    pipe = FramePipeline(ctx, W, H, 3, views, n_slots=2)
    frames = [rand_image(H, W, seed=300 + k) for k in range(5)]
    got = {}
    for k, f in enumerate(frames):
        done = pipe.submit(f, tag=k)
        if done:
            got[done[0]] = done[1]
    for tag, outs in pipe.drain():
        got[tag] = outs
    pipe.close()
    assert sorted(got) == list(range(5))
    for k, f in enumerate(frames):
        want = orc.equirect_views_u8(f, [orc.make_view(*s) for s in specs])
        _assert_same(got[k], want, f"pipeline frame {k}")


def test_frame_pipeline_batches_the_kernel_for_ring_families(ctx, orc):
    """batch = 4: frames are uploaded as they arrive, four of them are rendered by one launch -- the `full360coverage` ring family then
    takes the source-major kernel -- and their views come back per frame; 11 frames through 6 slots (a last batch of three)"""
    from gs360.stream import FramePipeline
    from util import PRESET_FULL360, HFOV_14MM
    H, W = 512, 1024
    specs = [(float(y), float(p), HFOV_14MM, HFOV_14MM, 128, 128) for y, p in PRESET_FULL360]
    views = [gs360.View.make(*s) for s in specs]
    with ctx.options(srcmajor=-1):
        pipe = FramePipeline(ctx, W, H, 3, views, n_slots=6, batch=4)
        frames = [rand_image(H, W, seed=320 + k) for k in range(11)]
        got, kernels = {}, set()
        for k, f in enumerate(frames):
            done = pipe.submit(f, tag=k)
            if k % 4 == 3:
                kernels.add(ctx.get_option("last_eq_kernel"))
            if done:
                got[done[0]] = done[1]
        for tag, outs in pipe.drain():
            got[tag] = outs
        pipe.close()
    assert kernels == {2}                                  # every four-frame launch ran eq_srcmajor_kernel
    assert sorted(got) == list(range(11))
    for k, f in enumerate(frames):
        want = orc.equirect_views_u8(f, [orc.make_view(*s) for s in specs], threads=0)
        _assert_same(got[k], want, f"batched pipeline frame {k}")


# ---- full BASELINE sizes: bit-exact where the oracle is fast enough, size-independent properties otherwise -------
def test_full_size_cfg3_full360coverage_checks(ctx, orc):
    """BASELINE cfg3 frame: 7680x3840 -> 12 x 1600^2 (full360coverage).  Every byte of all 12 views against the oracle."""
    src = rand_image(3840, 7680, seed=33)
    specs = [(y, p, HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs])
    want = orc.equirect_views_u8(src, [orc.make_view(*s) for s in specs], threads=0)
    _assert_same(got, want, "cfg3 full size")


def test_full_size_properties_roll_constant_symmetry(ctx):
    """size-independent properties at 8K / 2048^2 (BASELINE cfg5 shape), GPU against itself:
    (a) rolling the panorama by k texels == yawing the camera by k texels; (b) a constant image stays constant;
    (c) a left-right mirrored panorama gives the mirrored view for yaw 0."""
    H, W = 3840, 7680
    src = rand_image(H, W, seed=34)
    v0 = gs360.View.make(0.0, 30.0, HFOV_17MM, HFOV_17MM, 2048, 2048)
    k = 640
    vk = gs360.View.make(360.0 * k / W, 30.0, HFOV_17MM, HFOV_17MM, 2048, 2048)     # exactly k texels: 30 degrees
    a = ctx.equirect_views(np.ascontiguousarray(np.roll(src, -k, axis=1)), [v0])[0]
    b = ctx.equirect_views(src, [vk])[0]
    assert np.array_equal(a, b)
    const = np.full((H, W, 3), 93, np.uint8)
    assert (ctx.equirect_views(const, [vk])[0] == 93).all()
    # mirror: texel centres sit at half-integers, so flipping the panorama left-right maps column c -> W-1-c and the
    # yaw-0 view onto its own mirror image; the 1/32-px quantisation is symmetric except on exact .5 ties
    lvl = gs360.View.make(0.0, 0.0, HFOV_17MM, HFOV_17MM, 2048, 2048)
    m = ctx.equirect_views(np.ascontiguousarray(src[:, ::-1]), [lvl])[0][:, ::-1]
    d = np.abs(m.astype(int) - ctx.equirect_views(src, [lvl])[0].astype(int))
    assert (d > 0).mean() < 0.08 and d.max() <= 24        # only rounding-tie buckets may differ, by one 1/32-px step


def test_full_size_cfg4_table_remap_template_sensor(ctx, orc):
    """BASELINE cfg4 shape with the shipped template calibration: 3840^2 lens -> 1750^2 view, reference-identical
    host tables sampled on the GPU vs the oracle's cv2.remap restatement on the same tables (linear + cubic)."""
    from gs360 import fisheye as fe
    c = fe.SensorCalibration("0", "equisolid_fisheye", **TEMPLATE_CALIB)
    mx, my, valid = fe.perspective_tables(c, 40.0, 0.0, HFOV_14MM, HFOV_14MM, 1750, 1750, 190.0)
    src = rand_image(3840, 3840, seed=35)
    for interp in (1, 2):
        got = ctx.remap(src, mx, my, interpolation=interp, border_value=0.0, valid=valid, fill_value=0)
        want = orc.valid_fill(orc.remap_u8(src, mx, my, interp=interp, border_value=0.0, threads=0), valid, 0)
        _assert_same([got], [want], f"cfg4 full size interp={interp}")


def test_api_limits_are_reported_not_crashed(ctx):
    """maximum sizes / unsupported shapes come back as Gs360Error with a message (the boundary never aborts)"""
    tiny = rand_image(4, 4)
    with pytest.raises(gs360.Gs360Error) as e:
        ctx.equirect_views(tiny, [gs360.View.make(0, 0, 90, 90, 4, 4)])         # narrower than 8 texels
    assert e.value.code == -1
    src = rand_image(8, 16)
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(src, [gs360.View.make(0, 0, 90, 90, 40000, 2)])       # view wider than 32768
    big_w = np.zeros((1, 32767, 1), np.uint8)                                      # cv2.remap's own SHRT_MAX limit
    m = np.zeros((2, 2), np.float32)
    with pytest.raises(gs360.Gs360Error):
        ctx.remap(big_w, m, m)
    with pytest.raises(gs360.Gs360Error):
        ctx.remap(rand_image(8, 8), m, m, interpolation=3)                        # INTER_AREA is not a remap mode
    with pytest.raises(gs360.Gs360Error):
        ctx.equirect_views(rand_image(8, 16), [gs360.View.make(0, 0, 90, 90, 8, 8)], interp=gs360.INTERP_LANCZOS4)
    with pytest.raises(gs360.Gs360Error):
        ctx.remap(rand_image(8, 8, c=1)[:, :, 0].reshape(8, 4, 2), m, m)          # 2 channels
    assert ctx.remap(rand_image(8, 8), np.zeros((0, 5), np.float32), np.zeros((0, 5), np.float32)).shape == (0, 5, 3)
    # 1x1 source, every interpolation: all taps are border or the single texel
    one = np.array([[[10, 20, 30]]], np.uint8)
    mm = np.array([[0.0, 0.5, -0.5, 3.0]], np.float32)
    for interp in (0, 1, 2, 4):
        out = ctx.remap(one, mm, np.zeros_like(mm), interpolation=interp, border_value=(1, 2, 3, 4))
        assert out[0, 0].tolist() == [10, 20, 30] and out[0, 3].tolist() == [1, 2, 3]


def test_reduced_division_and_sqrt_sequences_equal_ieee(ctx):
    """gs360_eqspec.h replaces `/` and sqrtf by shorter sequences that must be bit-identical on the specs' operand domains:
    1.5 x 10^9 pseudo-random divisions and as many square roots on the GPU, both forms, zero mismatches"""
    for seed in (1, 20260424, 0xDEADBEEF):
        n, bad = ctx.selftest_arith(seed=seed, n_millions=512)
        assert n >= 512_000_000 and bad == 0, (seed, n, bad)
