"""Pins EQ-SPEC v1 (oracle side) on the reference's own convention: `tests/golden/eq_convention_goldens.npz` holds the outputs of
the reference GUI's direction_from_uv / rotate_pitch / rotate_yaw / lonlat_to_xy (gs360_GUI.py:342-424, lifted with `ast` and RUN
by tests/golden/make_eq_convention_goldens.py in the build container) for every view of the presets; a float64 re-statement of the
same formulas covers the poles and odd yaws on dense grids.  Parity is UNPINNED against ffmpeg v360's sampler (absent, unpinned
third-party binary); what is pinned is the geometry convention and the quantisation error."""
import json
import pathlib

import numpy as np
import pytest

GOLD = pathlib.Path(__file__).resolve().parent / "golden"

from util import HFOV_12MM, PRESET_FULL360, HFOV_14MM, rand_image, ring_views


def truth_xy(spec, W, H):
    yaw, pitch, hfov, vfov, w, h = spec
    i = np.arange(w)[None, :]
    j = np.arange(h)[:, None]
    u = ((i + 0.5) / w) * 2 - 1
    v = ((j + 0.5) / h) * 2 - 1
    hf = np.radians(min(max(hfov, 1e-3), 179.9))
    vf = np.radians(min(max(vfov, 1e-3), 179.9))
    x = np.tan(hf / 2) * u + 0 * v
    y = np.tan(vf / 2) * (-v) + 0 * u
    z = np.ones_like(x)
    n = np.sqrt(x * x + y * y + z * z)
    x, y, z = x / n, y / n, z / n
    p, yw = np.radians(pitch), np.radians(yaw)
    y1 = np.cos(p) * y + np.sin(p) * z
    z1 = -np.sin(p) * y + np.cos(p) * z
    x2 = np.cos(yw) * x + np.sin(yw) * z1
    z2 = -np.sin(yw) * x + np.cos(yw) * z1
    lon = np.arctan2(x2, z2)
    lat = np.arcsin(np.clip(y1, -1, 1))
    return (lon / (2 * np.pi) + 0.5) * W - 0.5, (0.5 - lat / np.pi) * H - 0.5, lat


def _convention_views():
    meta = json.loads((GOLD / "eq_convention_goldens.json").read_text())
    return meta["views"]


@pytest.mark.parametrize("view", _convention_views(), ids=lambda v: v["key"])
def test_quantised_map_matches_reference_gui_functions(orc, view):
    """oracle sx/32, sy/32 == round(32 (x - 1/2)) / 32 of the REFERENCE's lonlat_to_xy(direction_from_uv(...)) at pixel centres
    (same tolerances as the float64 re-statement below): every view of default / fisheyelike / full360coverage / --count 6
    --size 800, 8K and 5.7K panoramas."""
    g = np.load(GOLD / "eq_convention_goldens.npz")[view["key"]]
    W, H = view["W"], view["H"]
    sx, sy = orc.equirect_map(orc.make_view(view["yaw_deg"], view["pitch_deg"], view["hfov_deg"], view["vfov_deg"],
                                            view["width"], view["height"]), W, H)
    i, j = g[:, 0].astype(int), g[:, 1].astype(int)
    lat, X, Y = g[:, 3], g[:, 4] - 0.5, g[:, 5] - 0.5          # texel centres sit at half-integers of the GUI's pixel coordinates
    qx, qy = sx[j, i], sy[j, i]
    assert qx.min() >= 0 and qx.max() < 32 * W
    dx = (qx / 32.0 - X + W / 2) % W - W / 2
    dy = qy / 32.0 - Y
    cosl = np.maximum(np.cos(lat), 1e-3)
    assert np.abs(dy).max() <= 1 / 64 + 2e-4
    assert (np.abs(dx) * cosl).max() <= 1 / 64 + 3e-4
    # the quantised value IS the rounded reference coordinate except within 2e-4 px (on the sphere) of a bucket edge
    wrong_x = (np.rint(32 * X).astype(np.int64) % (32 * W)) != qx
    wrong_y = np.rint(32 * Y).astype(np.int64) != qy
    fx = np.abs((32 * X) % 1 - 0.5); fy = np.abs((32 * Y) % 1 - 0.5)
    assert (fx[wrong_x] * cosl[wrong_x] <= 32 * 3e-4 + 1e-9).all() and (fy[wrong_y] <= 32 * 2e-4 + 1e-9).all()
    assert wrong_x.mean() + wrong_y.mean() < 0.02


CASES = [(0, 0), (60, 0), (180, 0), (-120, 0), (45, 30), (135, -30), (-45, -30), (0, 90), (0, -90), (17.3, -62.1),
         (359.5, 12.0)]


@pytest.mark.parametrize("yaw,pitch", CASES)
def test_quantised_map_is_the_rounded_truth(orc, yaw, pitch):
    """sx/32, sy/32 must equal the float64 truth rounded to 1/32 px, except where the truth sits within
    2e-4 px of a bucket boundary (float32 evaluation error; < 0.2 % of pixels)."""
    W, H = 7680, 3840
    spec = (yaw, pitch, HFOV_12MM, HFOV_12MM, 400, 400)
    sx, sy = orc.equirect_map(orc.make_view(*spec), W, H)
    X, Y, lat = truth_xy(spec, W, H)
    assert sx.min() >= 0 and sx.max() < 32 * W
    dx = (sx / 32.0 - X + W / 2) % W - W / 2
    dy = sy / 32.0 - Y
    away_from_pole = np.abs(lat) < np.radians(89.0)     # d(lon) is amplified by 1/cos(lat) at the poles
    assert np.abs(dy).max() <= 1 / 64 + 2e-4
    assert np.abs(dx[away_from_pole]).max() <= 1 / 64 + 2e-3
    cosl = np.maximum(np.cos(lat), 1e-3)
    assert (np.abs(dx) * cosl).max() <= 1 / 64 + 3e-4    # error measured on the sphere
    flips = (np.abs(dx[away_from_pole]) > 1 / 64 + 1e-9).mean() + (np.abs(dy) > 1 / 64 + 1e-9).mean()
    assert flips < 0.004


def test_view_centre_lands_on_expected_texel(orc):
    W, H = 2048, 1024
    for yaw, pitch in [(0, 0), (90, 0), (-90, 0), (180, 0), (0, 45), (30, -20)]:
        v = orc.make_view(yaw, pitch, 1, 1, 2, 2)        # 1-degree view: all 4 pixels hug the view centre
        sx, sy = orc.equirect_map(v, W, H)
        cx = ((sx / 32.0 - (((yaw + 180) % 360) / 360.0 * W - 0.5) + W / 2) % W) - W / 2
        assert np.abs(cx).max() < 3.0
        cy = (sy / 32.0).mean()
        assert abs(cy - ((0.5 - pitch / 180.0) * H - 0.5)) < 0.05
        # +yaw looks right (larger x), +pitch looks up (smaller y): SURVEY 8(c)
    a = orc.equirect_map(orc.make_view(10, 0, 60, 60, 2, 2), W, H)[0].mean()
    b = orc.equirect_map(orc.make_view(20, 0, 60, 60, 2, 2), W, H)[0].mean()
    assert b > a
    a = orc.equirect_map(orc.make_view(0, 10, 60, 60, 2, 2), W, H)[1].mean()
    b = orc.equirect_map(orc.make_view(0, 20, 60, 60, 2, 2), W, H)[1].mean()
    assert b < a


def test_sampler_properties(orc):
    """constant image -> constant output; horizontal roll of the source == yaw shift by whole texels"""
    const = np.full((64, 128, 3), 77, np.uint8)
    out = orc.equirect_views_u8(const, [orc.make_view(33, 21, 100, 80, 50, 40)])[0]
    assert (out == 77).all()
    src = rand_image(128, 256, seed=9)
    W = 256
    v0 = orc.make_view(0, 15, 90, 90, 64, 64)
    v1 = orc.make_view(360.0 * 8 / W, 15, 90, 90, 64, 64)       # yaw by exactly 8 texels
    a = orc.equirect_views_u8(np.roll(src, -8, axis=1), [v0])[0]
    b = orc.equirect_views_u8(src, [v1])[0]
    assert np.array_equal(a, b)
    assert np.array_equal(orc.equirect_views_u8(src, [v0], threads=1)[0], orc.equirect_views_u8(src, [v0], threads=3)[0])


def test_vertical_clamp_and_horizontal_wrap(orc):
    H, W = 16, 32
    src = np.zeros((H, W, 1), np.uint8)
    src[0] = 200          # north pole row
    src[-1] = 100         # south pole row
    up = orc.equirect_views_u8(src, [orc.make_view(0, 90, 2, 2, 8, 8)])[0]
    down = orc.equirect_views_u8(src, [orc.make_view(0, -90, 2, 2, 8, 8)])[0]
    assert (up == 200).all() and (down == 100).all()
    seam = np.zeros((H, W, 1), np.uint8)
    seam[:, 0] = 255
    seam[:, -1] = 255
    back = orc.equirect_views_u8(seam, [orc.make_view(180, 0, 10, 10, 16, 16)])[0]
    assert back.max() == 255 and back[:, 7:9].min() > 200     # the seam columns blend into a solid band


def test_algorithmic_bytes_constant_of_bench(orc):
    """bench.py's ALGO_BYTES_PER_FRAME: sum of distinct texels touched per view for BASELINE cfg2."""
    import bench
    W, H = 7680, 3840
    union = np.zeros((H, W), np.uint8)
    total = sum(orc.equirect_distinct_texels(orc.make_view(*s), W, H, union) for s in ring_views(6, 800, HFOV_12MM))
    assert total == 14_325_324
    assert int(union.sum()) == 11_685_864
    assert bench.ALGO_BYTES_PER_FRAME == 6 * 800 * 800 * 3 + total * 3 == 54_495_972
    assert [tuple(v) for v in bench.view_table()] == [tuple(s) for s in ring_views(6, 800, HFOV_12MM)]


def test_line_bound_constant_of_bench(orc):
    """bench.py's LINE_BYTES_PER_FRAME: distinct 128-B source lines the bilinear taps of each cfg2 view touch (frame base
    128-B aligned, tight rows), summed over the six views, x 128 B + the stores."""
    import bench
    W, H = 7680, 3840
    per_view, every = [], []
    for spec in ring_views(6, 800, HFOV_12MM):
        sx, sy = orc.equirect_map(orc.make_view(*spec), W, H)
        ix, iy = (sx >> 5).astype(np.int64), (sy >> 5).astype(np.int64)
        ix1 = np.where(ix + 1 == W, 0, ix + 1)
        lines = []
        for yy in (np.clip(iy, 0, H - 1), np.clip(iy + 1, 0, H - 1)):
            for xx in (ix, ix1):
                first = (yy * W + xx) * 3
                lines += [(first >> 7).ravel(), ((first + 2) >> 7).ravel()]
        per_view.append(int(np.unique(np.concatenate(lines)).size))
        every.append(np.unique(np.concatenate(lines)))
    union = int(np.unique(np.concatenate(every)).size)
    assert union == 413_172
    assert per_view == [119_680] * 6
    assert bench.LINE_BYTES_PER_FRAME == sum(per_view) * 128 + 6 * 800 * 800 * 3 == 103_434_240
    # the source-major kernel's bound: every distinct line ONCE for all six views
    assert bench.UNION_LINE_BYTES_PER_FRAME == union * 128 + 6 * 800 * 800 * 3 == 64_406_016


# ---- equidistant-fisheye OUTPUT (the fisheyeXY preset's v360 output=fisheye jobs, PC:351-414) ---------------------
def truth_fisheye_xy(spec, W, H):
    """float64: image-plane radius r <-> 90 r degrees off axis (equidistant), pitch about X, yaw about Y, lon/lat"""
    yaw, pitch, hfov, vfov, w, h = spec
    u = ((2 * np.arange(w) + 1 - w) / w)[None, :] * (hfov / 180.0)
    v = ((2 * np.arange(h) + 1 - h) / h)[:, None] * (vfov / 180.0)
    r = np.hypot(u, v)
    ang = np.pi / 2 * r
    s = np.where(r > 0, np.sin(ang) / np.where(r > 0, r, 1), np.pi / 2)
    x, yd, z = u * s, v * s, np.cos(ang) + 0 * u
    p, yw = np.radians(pitch), np.radians(yaw)
    fwd = np.sin(p) * yd + np.cos(p) * z
    up = -np.cos(p) * yd + np.sin(p) * z
    lon = np.arctan2(x, fwd) + yw
    lat = np.arctan2(up, np.hypot(x, fwd))
    return np.mod((lon / (2 * np.pi) + 0.5) * W - 0.5, W), (0.5 - lat / np.pi) * H - 0.5, lat


@pytest.mark.parametrize("yaw,pitch,fov", [(0, 0, 127.27922061357856), (180, 0, 127.27922061357856), (33.0, -20.0, 100.0),
                                           (-90.0, 45.0, 200.0), (10.0, 0.0, 254.0)])
def test_fisheye_output_map_is_the_rounded_truth(orc, yaw, pitch, fov):
    W, H = 7680, 3840
    spec = (yaw, pitch, fov, fov, 360, 360)
    sx, sy = orc.equirect_fisheye_map(orc.make_view(*spec), W, H)
    X, Y, lat = truth_fisheye_xy(spec, W, H)
    assert sx.min() >= 0 and sx.max() < 32 * W
    dx = (sx / 32.0 - X + W / 2) % W - W / 2
    dy = sy / 32.0 - Y
    cosl = np.maximum(np.cos(lat), 1e-3)
    assert np.abs(dy).max() <= 1 / 64 + 2e-3                  # polynomial sin/cos: < 0.001 px at 8K on top of the rounding
    assert (np.abs(dx) * cosl).max() <= 1 / 64 + 2e-3


def test_fisheye_output_centre_and_rim(orc):
    """view centre looks along (yaw, pitch); at d_fov = 180 the corners of a square image are 90 degrees off axis"""
    W, H = 4096, 2048
    n = 201
    fov = 180.0 / np.sqrt(2.0)
    sx, sy = orc.equirect_fisheye_map(orc.make_view(90.0, 0.0, fov, fov, n, n), W, H)
    c = n // 2
    assert abs(sx[c, c] / 32.0 - ((90 / 360 + 0.5) * W - 0.5)) <= 1 / 64 + 1e-3 and abs(sy[c, c] / 32.0 - (H / 2 - 0.5)) <= 1 / 64 + 1e-3
    # pixel centres nearest the corner: r = (n-1)/n * sqrt(2) * fov/180 = (n-1)/n -> 90 (n-1)/n degrees off axis, on the diagonal
    off = np.radians(90.0 * (n - 1) / n)
    d = np.array([np.sin(off) / np.sqrt(2), np.sin(off) / np.sqrt(2), np.cos(off)])        # (right, up, forward) of the top-right corner
    lon = np.arctan2(d[0], d[2]) + np.pi / 2
    lat = np.arcsin(d[1])
    assert abs(sx[0, n - 1] / 32.0 - np.mod((lon / (2 * np.pi) + 0.5) * W - 0.5, W)) < 0.05
    assert abs(sy[0, n - 1] / 32.0 - ((0.5 - lat / np.pi) * H - 0.5)) < 0.05
    src = rand_image(128, 256)
    a = orc.equirect_fisheye_views_u8(src, [orc.make_view(0, 0, fov, fov, 33, 21)])[0]
    b = orc.equirect_fisheye_views_u8(np.roll(src, 64, axis=1), [orc.make_view(90.0, 0, fov, fov, 33, 21)])[0]
    assert np.array_equal(a, b)                          # yaw by 90 degrees == rolling the panorama by W/4
