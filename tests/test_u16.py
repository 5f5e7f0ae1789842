"""16-bit (uint16) samplers (reference DF:735 and PC:327-347 keep 16-bit sources at native depth).

CPU: the oracle's CV_16U restatement against hand-derivable known answers and against the 8-bit samplers.
GPU (-m gpu): gs360_remap_table_u16 / gs360_equirect_views_u16 through the C ABI, bit-exact against the oracle."""
import numpy as np
import pytest

import gs360
from util import HFOV_12MM, ring_views


def rand16(h, w, c=3, seed=5):
    return np.random.default_rng(seed).integers(0, 65536, size=(h, w, c), dtype=np.uint16)


def rand_maps(h, w, H, W, seed):
    rng = np.random.default_rng(seed)
    mx = rng.uniform(-12, W + 12, (h, w)).astype(np.float32)
    my = rng.uniform(-12, H + 12, (h, w)).astype(np.float32)
    mx[0, :8] = np.array([0.0, -1.0, W - 1.0, W - 0.5, 1 / 64, 3 / 64, -0.015625, W + 5.0], np.float32)
    my[0, :8] = np.array([0.0, -1.0, H - 1.0, H - 0.5, 1 / 64, 3 / 64, -0.015625, 2.0], np.float32)
    mx[3, 5] = np.nan
    my[4, 6] = np.inf
    mx[5, 7] = -3e9
    my[6, 8] = 1e30
    return mx, my


# ---- oracle known answers (CPU) ------------------------------------------------------------------------------
def test_oracle_u16_identity_halfshift_and_constant(orc):
    src = rand16(40, 64)
    yy, xx = np.meshgrid(np.arange(40, dtype=np.float32), np.arange(64, dtype=np.float32), indexing="ij")
    for interp in (0, 1, 2, 4):
        assert np.array_equal(orc.remap_u16(src, xx, yy, interp=interp), src), interp       # phase 0 = unit impulse
    half = orc.remap_u16(src, xx + np.float32(0.5), yy, interp=1)
    a, b = src[:, :-1].astype(np.int64), src[:, 1:].astype(np.int64)
    s = a + b
    want = s // 2 + ((s & 1) & ((s // 2) & 1))              # (a+b)/2 rounded half to even = cvRound of the exact float value
    assert np.array_equal(half[:, :-1], want.astype(np.uint16))
    const = np.full((30, 30, 3), 51234, np.uint16)
    mx, my = rand_maps(20, 25, 30, 30, 3)
    inside = (mx > 4) & (mx < 24) & (my > 4) & (my < 24)
    for interp in (1, 2, 4):
        out = orc.remap_u16(const, mx, my, interp=interp, border_value=(51234, 51234, 51234, 0))
        assert np.abs(out[inside].astype(int) - 51234).max() <= 1     # float weights sum to 1 within rounding


def test_oracle_u16_tracks_the_8bit_sampler(orc):
    """a 16-bit image that is 257 x an 8-bit one: the float-weight result / 257 stays within one 8-bit level of cv2's
    fixed-point 8-bit result (different arithmetic, same geometry and borders)"""
    s8 = np.random.default_rng(9).integers(0, 256, (50, 70, 3), dtype=np.uint8)
    s16 = s8.astype(np.uint16) * 257
    mx, my = rand_maps(40, 45, 50, 70, 10)
    for interp in (0, 1, 2, 4):
        o8 = orc.remap_u8(s8, mx, my, interp=interp, border_value=(7, 0, 0, 0))
        o16 = orc.remap_u16(s16, mx, my, interp=interp, border_value=(7 * 257, 0, 0, 0))
        d = np.abs(o16.astype(np.float64) / 257.0 - o8)
        assert d.max() <= (0.0 if interp == 0 else 1.01), (interp, d.max())
    views = [orc.make_view(30, 10, 100, 80, 50, 40), orc.make_view(-170, -35, 90, 90, 33, 31)]
    for interp in (1, 2):
        o8 = orc.equirect_views_u8(s8, views, interp=interp)
        o16 = orc.equirect_views_u16(s16, views, interp=interp)
        for a, b in zip(o8, o16):
            assert np.abs(b.astype(np.float64) / 257.0 - a).max() <= 1.0


def test_cubic_table_properties_the_16bit_kernel_relies_on(orc):
    """eq16_cubic_blend_rgb accumulates sum w (S - 32768) in 32 bits and adds 32768 * 32768 back: every phase of the fixed-point
    Keys table must sum to 32768, and 32768 * sum |w| must stay below 2^31."""
    t = orc.cubic_table().astype(np.int64).reshape(1024, 16)
    assert (t.sum(axis=1) == 32768).all()
    assert int(np.abs(t).sum(axis=1).max()) * 32768 < 2 ** 31
    # worst cases through the oracle's 64-bit accumulation: all-0 / all-65535 images and a +-extreme checker stay in range
    for img in (np.zeros((9, 16, 3), np.uint16), np.full((9, 16, 3), 65535, np.uint16),
                (np.indices((9, 16)).sum(0) % 2 * 65535).astype(np.uint16)[..., None].repeat(3, 2)):
        out = orc.equirect_views_u16(img, [orc.make_view(10.0, 5.0, 80.0, 80.0, 24, 24)], interp=2)[0]
        assert out.dtype == np.uint16


# ---- GPU parity -----------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("interp", [1, 2])
def test_gpu_equirect_u16_rings_extremes_and_odd_shapes(ctx, orc, interp):
    """the 16-bit path on the ring skeleton: level + flipped-pitch rings, odd widths (centre column, byte-store fallback),
    0 / 65535 checkers (the cubic sampler's 32-bit accumulation at its limits), strided rows"""
    rng = np.random.default_rng(91)
    src = rng.integers(0, 65536, (180, 360, 3), dtype=np.uint16)
    src[::2, ::2] = 65535
    src[1::2, 1::2] = 0
    specs = [(y, p, 100.0, 100.0, 75, 61) for y in (0.0, 90.0, 180.0, -90.0) for p in (30.0, -30.0)]
    specs += [(y, 0.0, 112.62, 112.62, 80, 80) for y in (0.0, 45.0, 90.0)] + [(12.3, 77.0, 60.0, 60.0, 33, 47), (0.0, 90.0, 105.0, 105.0, 64, 64)]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], interp=interp)
    want = orc.equirect_views_u16(src, [orc.make_view(*s) for s in specs], interp=interp)
    for k, (g, w_) in enumerate(zip(got, want)):
        assert np.array_equal(g, w_), (k, specs[k], interp)


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("interp", [0, 1, 2, 4])
def test_gpu_table_remap_u16(ctx, orc, channels, interp):
    H, W, h, w = 97, 131, 75, 108
    src = rand16(H, W, channels, seed=21)
    mx, my = rand_maps(h, w, H, W, 22)
    valid = np.random.default_rng(23).random((h, w)) > 0.1
    bv = (40000.0, 123.0, 0.0, 70000.0)
    got = ctx.remap(src, mx, my, interpolation=interp, border_value=bv, valid=valid, fill_value=51000)
    want = orc.valid_fill(orc.remap_u16(src, mx, my, interp=interp, border_value=bv).copy(), valid, 51000)
    assert got.dtype == np.uint16 and got.shape == want.shape
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"C={channels} interp={interp}: {len(bad)} mismatches, first {bad[0].tolist()}"


@pytest.mark.gpu
@pytest.mark.parametrize("interp", [1, 2])
def test_gpu_equirect_u16(ctx, orc, interp):
    src = rand16(301, 602, 3, seed=3)
    specs = [(0, 90, 100, 100, 96, 96), (180, 0, 120, 90, 130, 70), (-179.9, 45, 60, 60, 33, 47), (37.3, -62.1, 150, 140, 101, 99),
             (0, 0, 112.6, 112.6, 130, 130), (720.5, 0, 90, 90, 31, 5)]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], interp=interp)
    want = orc.equirect_views_u16(src, [orc.make_view(*s) for s in specs], interp=interp)
    for k, (g, w_) in enumerate(zip(got, want)):
        assert g.dtype == np.uint16 and np.array_equal(g, w_), (k, interp)
    for channels in (1, 4):
        s2 = rand16(128, 256, channels, seed=4)
        sp = [(10.0, 0.0, 100.0, 100.0, 67, 21), (10.0, -35.0, 100.0, 100.0, 66, 20)]
        got = ctx.equirect_views(s2, [gs360.View.make(*s) for s in sp], interp=interp)
        want = orc.equirect_views_u16(s2, [orc.make_view(*s) for s in sp], interp=interp)
        assert all(np.array_equal(g, w_) for g, w_ in zip(got, want)), channels
    fish = [(0.0, 0.0, 127.28, 127.28, 90, 90)]
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in fish], interp=interp, flags=gs360.EQ_FISHEYE_OUT)
    want = orc.equirect_views_u16(src, [orc.make_view(*s) for s in fish], interp=interp, fisheye=True)
    assert np.array_equal(got[0], want[0])


@pytest.mark.gpu
def test_gpu_equirect_u16_full_size_8k(ctx, orc):
    """rgb48 8K frame -> 6 x 800^2 (cfg2 shape at 16 bits), every sample"""
    src = rand16(3840, 7680, 3, seed=8)
    specs = ring_views(6, 800, HFOV_12MM)
    got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs])
    want = orc.equirect_views_u16(src, [orc.make_view(*s) for s in specs], threads=0)
    for k, (g, w_) in enumerate(zip(got, want)):
        assert np.array_equal(g, w_), k


@pytest.mark.gpu
def test_gpu_bswap16_in_place(ctx):
    """gs360_dev_bswap16 (big-endian 16-bit PPM frames of the video pipe are swapped on the device): every size parity"""
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 64, 1001, 3 * 7680 * 5 + 1):
        a = rng.integers(0, 65536, n, dtype=np.uint16)
        d = ctx.to_device(a)
        ctx.bswap16(d, n, slot=0)
        got = ctx.download(d, (n,), dtype=np.uint16)
        assert np.array_equal(got, a.byteswap()), n
        ctx.free(d)
