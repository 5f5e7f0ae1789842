"""-m gpu: the LDS-staged table kernel (csrc/gs360_tablestage.hip: bilinear RGB through a map plan's stage plan) against the CPU oracle,
every byte.  Reference call sites: cv2.remap + valid fill, cli_tools/gs360_DualFisheyeDistortionCalibration.py:2001-2014, :1198-1212."""
import numpy as np
import pytest

import gs360
from util import TEMPLATE_CALIB, rand_image

pytestmark = pytest.mark.gpu


def _diff(got, want, what):
    assert got.shape == want.shape, what
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        d = np.abs(got.astype(int) - want.astype(int))
        raise AssertionError(f"{what}: {len(bad)} mismatching bytes of {got.size}, max |d| = {d.max()}, first at {bad[0].tolist()}")


def _smooth_maps(h, w, H, W, kind, seed=0):
    """maps a lens / view geometry would produce: affine + a gentle radial term; `kind` moves them across the source's borders"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    u, v = xx / max(w - 1, 1) - 0.5, yy / max(h - 1, 1) - 0.5
    ang = {"inside": 0.1, "rot": 0.6, "cross": -0.25, "far": 0.0, "edge": 0.0}[kind]
    sc = {"inside": 0.8, "rot": 0.62, "cross": 1.25, "far": 1.0, "edge": 1.0}[kind]
    r2 = u * u + v * v
    k = 1.0 + 0.18 * r2
    ur, vr = (u * np.cos(ang) - v * np.sin(ang)) * k, (u * np.sin(ang) + v * np.cos(ang)) * k
    mx = (ur * sc + 0.5) * (W - 1)
    my = (vr * sc + 0.5) * (H - 1)
    if kind == "far":
        mx += 3.0 * W
    if kind == "edge":                                   # exactly onto the last column / row (weight-zero taps), and half a pixel past
        mx = np.clip(mx, 0, W - 1) + (np.arange(w)[None, :] % 7 == 0) * 0.5
        my = np.clip(my, 0, H - 1) + (np.arange(h)[:, None] % 5 == 0) * 0.5
    mx += rng.uniform(-0.02, 0.02, size=mx.shape)
    return mx.astype(np.float32), my.astype(np.float32)


def _run_plans(ctx, jobs_host, src, bv, **opts):
    """jobs_host: list of (mx, my, valid or None, use_valid, fill, dst_pad) -> list of outputs through gs360_remap_plans_u8"""
    H, W = src.shape[:2]
    d_src = ctx.to_device(src)
    plans, bufs, jobs = [], [], []
    for mx, my, valid, use_valid, fill, _ in jobs_host:
        h, w = mx.shape
        d = [ctx.to_device(mx), ctx.to_device(my), ctx.to_device(valid.astype(np.uint8)) if valid is not None else None]
        plans.append(ctx.map_plan(d[0], d[1], d[2], h, w))
        for b in d:
            if b is not None:
                ctx.free(b)
        bufs.append(ctx.alloc(h * w * 3))
        jobs.append((d_src, H, W, plans[-1], use_valid, h, w, fill, bufs[-1]))
    with ctx.options(**opts):
        ctx.remap_plans_dev(jobs, 3, interp=1, border_value=bv)
        ctx.sync(0)
        staged, slow = ctx.get_option("last_table_kernel"), ctx.get_option("last_table_stage_slow_tiles")
        # a second call reuses the stage plans
        ctx.remap_plans_dev(jobs, 3, interp=1, border_value=bv)
        ctx.sync(0)
    outs = [ctx.download(b, (j[5], j[6], 3)) for b, j in zip(bufs, jobs)]
    for pl in plans:
        ctx.map_plan_free(pl)
    for b in bufs + [d_src]:
        ctx.free(b)
    return outs, staged, slow


SHAPES = [(75, 108), (33, 200), (41, 76), (64, 64), (23, 252), (130, 70), (7, 1028)]      # h w % 4 == 0: tight outputs whose quads are whole


@pytest.mark.parametrize("rows", [8, 16, 32])
@pytest.mark.parametrize("kind", ["inside", "rot", "cross", "far", "edge"])
def test_staged_smooth_maps_every_class(ctx, orc, kind, rows):
    """smooth maps (boxes are used) that stay inside, rotate, cross all four borders (slow + border pixels), lie wholly outside (border
    constant only) and sit exactly on the last column / row (weight-zero taps served from the box); valid fill on and off; widths that
    are / are not multiples of four; all jobs in ONE launch; tiles of 8 / 16 / 32 rows"""
    H, W = 211, 316
    src = rand_image(H, W, seed=91)
    bv = (37.0, 11.0, 5.0, 0.0)
    host = []
    for k, (h, w) in enumerate(SHAPES):
        mx, my = _smooth_maps(h, w, H, W, kind, seed=k)
        valid = np.random.default_rng(50 + k).random((h, w)) > 0.15 if k % 2 == 0 else None
        host.append((mx, my, valid, valid is not None and k % 4 == 0, 200, 0))
    outs, staged, slow = _run_plans(ctx, host, src, bv, table_stage=1, table_stage_rows=rows)
    assert staged == len(SHAPES)
    if kind == "far":
        assert slow == 0                                 # (strongly minifying jobs of the other kinds outgrow the box budget: per-tile fallback)
    for k, ((mx, my, valid, use_valid, fill, _), got) in enumerate(zip(host, outs)):
        want = orc.remap_u8(src, mx, my, interp=1, border_value=bv)
        if use_valid:
            want = orc.valid_fill(want.copy(), valid, fill)
        _diff(got, want.reshape(got.shape), f"staged table {kind} job {k} {mx.shape} rows={rows}")


@pytest.mark.parametrize("mode", [1, -1])
def test_staged_random_maps_go_the_slow_way(ctx, orc, mode):
    """maps that scatter a tile's taps over the whole source: forced (table_stage = 1) every pixel is redone from memory through the plan's
    own positions -- NaN / inf / huge values, exact ties, the clamp range included; left to itself (-1) the library keeps the gather kernel"""
    H, W = 97, 132
    src = rand_image(H, W, seed=31)
    bv = (37.0, 0.0, 0.0, 0.0)
    host = []
    for k, (h, w) in enumerate([(75, 108), (33, 200), (41, 76)]):
        rng = np.random.default_rng(32 + k)
        mx = rng.uniform(-40, W + 40, size=(h, w)).astype(np.float32)
        my = rng.uniform(-40, H + 40, size=(h, w)).astype(np.float32)
        mx[3, 5] = np.nan
        my[4, 6] = np.inf
        mx[5, 7] = -3e9
        my[6, 8] = 1e30
        mx[7, :8] = np.array([-8.0, -8.03125, -9.0, W + 7.96875, W + 8.0, W + 9.0, 4087.0, 4088.5], np.float32)
        mx[0, :8] = np.array([0.0, -1.0, W - 1.0, W - 0.5, 1 / 64, 3 / 64, -0.015625, W + 5.0], np.float32)
        my[0, :8] = np.array([0.0, -1.0, H - 1.0, H - 0.5, 1 / 64, 3 / 64, -0.015625, 2.0], np.float32)
        valid = rng.random((h, w)) > 0.1
        host.append((mx, my, valid, k != 1, 200, 0))
    # a budget-sized box needs taps further apart than this source is tall: shrink the tiles' chance by using the smallest rows
    outs, staged, slow = _run_plans(ctx, host, src, bv, table_stage=mode, table_stage_rows=32)
    for k, ((mx, my, valid, use_valid, fill, _), got) in enumerate(zip(host, outs)):
        want = orc.remap_u8(src, mx, my, interp=1, border_value=bv)
        if use_valid:
            want = orc.valid_fill(want.copy(), valid, fill)
        _diff(got, want.reshape(got.shape), f"random maps job {k} table_stage={mode}")
    assert staged == (3 if mode == 1 else staged)


def test_staged_large_scatter_tiles_fall_back_per_tile(ctx, orc):
    """a map that is smooth except for a band of rows that jump across a 1500-row source: those tiles' boxes exceed the LDS budget and
    are redone from memory, the others are staged"""
    H, W = 1500, 900
    src = rand_image(H, W, seed=77)
    h, w = 96, 256
    mx, my = _smooth_maps(h, w, H, W, "inside", seed=3)
    my[40:44, :] = (np.arange(w)[None, :] * 5.7) % (H - 2)
    outs, staged, slow = _run_plans(ctx, [(mx, my, None, False, 0, 0)], src, (0, 0, 0, 0), table_stage=1, table_stage_rows=16)
    assert staged == 1 and slow > 0
    _diff(outs[0], orc.remap_u8(src, mx, my, interp=1).reshape(outs[0].shape), "scatter band")


def test_staged_padded_rows_and_ineligible_jobs(ctx, orc):
    """outputs with a padded row stride take the staged kernel when their rows are whole dwords; jobs it cannot take (odd widths with
    padding, h w not a multiple of four, single channel, bicubic) keep the gather kernel inside the same call"""
    H, W = 150, 260
    src = rand_image(H, W, seed=5)
    d_src = ctx.to_device(src)
    cases = [(40, 64, 64 * 3 + 16, True), (40, 100, 100 * 3 + 4, True), (33, 75, 75 * 3, False), (40, 70, 70 * 3 + 2, False)]
    with ctx.options(table_stage=1):
        for h, w, stride, expect in cases:
            mx, my = _smooth_maps(h, w, H, W, "cross", seed=h + w)
            d = [ctx.to_device(mx), ctx.to_device(my)]
            plan = ctx.map_plan(d[0], d[1], None, h, w)
            dst = ctx.alloc(h * stride)
            ctx.memset(dst, 0xAB)
            arr = (gs360.capi.RemapJob * 1)(gs360.capi.RemapJob(d_src.ptr, H, W, 0, None, None, None, h, w, 0, dst.ptr, stride))
            import ctypes as C
            pl = (C.c_void_p * 1)(plan)
            bv = (C.c_double * 4)(9.0, 8.0, 7.0, 0.0)
            gs360.capi._check(ctx.L.gs360_remap_plans_u8(ctx.handle, arr, pl, 1, 3, 1, bv, 0), ctx.L)
            ctx.sync(0)
            assert (ctx.get_option("last_table_kernel") == 1) == expect, (h, w, stride)
            raw = ctx.download(dst, (h, stride))
            got = raw[:, : w * 3].reshape(h, w, 3)
            _diff(got, orc.remap_u8(src, mx, my, interp=1, border_value=(9.0, 8.0, 7.0, 0.0)).reshape(h, w, 3), f"padded {h}x{w} stride {stride}")
            assert (raw[:, w * 3:] == 0xAB).all(), "the padding of a row was written"
            ctx.map_plan_free(plan)
            for b in d + [dst]:
                ctx.free(b)
        # bicubic and single-channel calls never stage
        mx, my = _smooth_maps(40, 64, H, W, "inside")
        d = [ctx.to_device(mx), ctx.to_device(my)]
        plan = ctx.map_plan(d[0], d[1], None, 40, 64)
        dst = ctx.alloc(40 * 64 * 3)
        ctx.remap_plans_dev([(d_src, H, W, plan, False, 40, 64, 0, dst)], 3, interp=2)
        ctx.sync(0)
        assert ctx.get_option("last_table_kernel") == 0
        _diff(ctx.download(dst, (40, 64, 3)), orc.remap_u8(src, mx, my, interp=2).reshape(40, 64, 3), "bicubic through a plan")
        ctx.map_plan_free(plan)
        for b in d + [dst]:
            ctx.free(b)
    ctx.free(d_src)


@pytest.mark.parametrize("sensor", [4000, 3840])
def test_cfg4_full_size_all_sfm10_views_staged(ctx, orc, sensor):
    """BASELINE configs[3] as the tool runs it: a 2 x sensor^2 pair -> all 10 SFM10 views (1750^2) through map plans, ONE call, the
    library's own kernel choice -- the LDS-staged kernel for every view -- every byte against the oracle; then the same with the stage
    plans rebuilt for 16-row tiles and three workgroups per CU"""
    from gs360 import fisheye as fe
    kw = dict(TEMPLATE_CALIB, width=sensor, height=sensor)
    c = fe.SensorCalibration("0", "equisolid_fisheye", kw["width"], kw["height"], kw["f"], kw["cx"], kw["cy"], kw["k1"], kw["k2"], kw["k3"])
    specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)
    tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
    imgs = {"X": rand_image(sensor, sensor, seed=104), "Y": rand_image(sensor, sensor, seed=105)}
    dev = {k: ctx.to_device(v) for k, v in imgs.items()}
    plans = {}
    for v, t in tables.items():
        d = (ctx.to_device(t["map_x"]), ctx.to_device(t["map_y"]), ctx.to_device(np.ascontiguousarray(t["valid"], np.uint8)))
        plans[v] = ctx.map_plan(*d, 1750, 1750)
        for b in d:
            ctx.free(b)
    d_out = {v: ctx.alloc(1750 * 1750 * 3) for v in tables}
    jobs = [(dev[tables[s["view_id"]]["lens_key"]], sensor, sensor, plans[s["view_id"]], True, 1750, 1750, 0, d_out[s["view_id"]]) for s in specs]
    wants = {}
    for opts in (dict(), dict(table_stage_rows=16, table_stage_wgs=3)):
        for b in d_out.values():
            ctx.memset(b, 0x5A)
        with ctx.options(**opts):
            ctx.remap_plans_dev(jobs, 3, interp=1, border_value=(0, 0, 0, 0))
            ctx.sync(0)
            assert ctx.get_option("last_table_kernel") == 10
        for s in specs:
            t = tables[s["view_id"]]
            got = ctx.download(d_out[s["view_id"]], (1750, 1750, 3))
            if s["view_id"] not in wants:
                wants[s["view_id"]] = orc.valid_fill(orc.remap_u8(imgs[t["lens_key"]], t["map_x"], t["map_y"], interp=1, threads=0), t["valid"], 0)
            _diff(got, wants[s["view_id"]], f"cfg4 {sensor}^2 view {s['view_id']} staged {opts}")
    for pl in plans.values():
        ctx.map_plan_free(pl)
    for b in list(dev.values()) + list(d_out.values()):
        ctx.free(b)


def test_undistort_shaped_maps_staged(ctx, orc):
    """the tool's other cv2.remap call (DF:1198-1212: the undistorted fisheye image, sensor-sized maps, zoomed so that the corners leave
    the model): a 1000 x 1000 lens image through its own undistort tables"""
    from gs360 import fisheye as fe
    kw = dict(TEMPLATE_CALIB, width=1000, height=1000)
    kw["f"] = kw["f"] * 1000 / 3840
    c = fe.SensorCalibration("0", "equisolid_fisheye", kw["width"], kw["height"], kw["f"], kw["cx"], kw["cy"], kw["k1"], kw["k2"], kw["k3"])
    ut = fe.undistort_tables(c, 1.0, 190.0)
    mx, my, valid = ut.map_x, ut.map_y, ut.valid_mask
    src = rand_image(1000, 1000, seed=12)
    outs, staged, slow = _run_plans(ctx, [(np.ascontiguousarray(mx), np.ascontiguousarray(my), np.asarray(valid, bool), True, 0, 0)], src, (0, 0, 0, 0))
    assert staged == 1
    want = orc.valid_fill(orc.remap_u8(src, mx, my, interp=1), np.asarray(valid, bool), 0)
    _diff(outs[0], want.reshape(outs[0].shape), "undistort maps")
