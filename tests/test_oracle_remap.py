"""Pins the oracle's cv2.remap restatement (parity unpinned at the cv2 boundary: no cv2 here, the reference has
no tests) with hand-derivable integer known answers and an independent NumPy restatement."""
import numpy as np
import pytest

from util import rand_image


def np_remap_linear(src, mx, my, cval):
    """Independent vectorised restatement of remapBilinear/BORDER_CONSTANT for u8."""
    H, W, C = src.shape
    with np.errstate(invalid="ignore", over="ignore"):
        vx, vy = mx * np.float32(32), my * np.float32(32)
        okx = (vx >= -2147483648.0) & (vx < 2147483648.0)
        oky = (vy >= -2147483648.0) & (vy < 2147483648.0)
        sx = np.where(okx, np.rint(np.where(okx, vx, 0)), -2147483648).astype(np.int64)
        sy = np.where(oky, np.rint(np.where(oky, vy, 0)), -2147483648).astype(np.int64)
    fx, fy = sx & 31, sy & 31
    ix, iy = np.clip(sx >> 5, -32768, 32767), np.clip(sy >> 5, -32768, 32767)
    acc = np.zeros(mx.shape + (C,), np.int64)
    cv = np.array(cval[:C], np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = ix + dx, iy + dy
            inb = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            val = np.where(inb[..., None], src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64), cv)
            wx = fx if dx else 32 - fx
            wy = fy if dy else 32 - fy
            acc += val * (wx * wy * 32)[..., None]
    res = (acc + 16384) >> 15
    outside = (ix >= W) | (ix + 1 < 0) | (iy >= H) | (iy + 1 < 0)
    return np.where(outside[..., None], cv, res).astype(np.uint8)


def grid(h, w):
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    return xx, yy


def test_identity_is_exact(orc):
    src = rand_image(37, 53)
    xx, yy = grid(37, 53)
    assert np.array_equal(orc.remap_u8(src, xx, yy, interp=1), src)
    assert np.array_equal(orc.remap_u8(src, xx, yy, interp=0), src)


def test_half_pixel_shift_is_rounded_mean(orc):
    src = rand_image(20, 40)
    xx, yy = grid(20, 40)
    out = orc.remap_u8(src, xx + np.float32(0.5), yy, interp=1, border_value=(0, 0, 0, 0))
    assert np.array_equal(out[:, :-1], (src[:, :-1].astype(int) + src[:, 1:].astype(int) + 1) >> 1)
    assert np.array_equal(out[:, -1], (src[:, -1].astype(int) + 1) >> 1)   # right tap is the (zero) border
    out = orc.remap_u8(src, xx, yy + np.float32(0.5), interp=1, border_value=(255, 255, 255, 255))
    assert np.array_equal(out[:-1], (src[:-1].astype(int) + src[1:].astype(int) + 1) >> 1)
    assert np.array_equal(out[-1], (src[-1].astype(int) + 255 + 1) >> 1)


def test_1_64_tie_rounds_to_even_bucket(orc):
    """map = x + 1/64 -> 32x + 0.5 -> half-to-even -> fx = 0 (32x is even) -> exact copy;
    map = x + 3/64 -> 32x + 1.5 -> 32x + 2 -> fx = 2."""
    src = rand_image(8, 32)
    xx, yy = grid(8, 32)
    assert np.array_equal(orc.remap_u8(src, xx + np.float32(1 / 64), yy, interp=1), src)
    out = orc.remap_u8(src, xx + np.float32(3 / 64), yy, interp=1)
    a, b = src[:, :-1].astype(int), src[:, 1:].astype(int)
    assert np.array_equal(out[:, :-1], (a * 30 * 32 * 32 + b * 2 * 32 * 32 + 16384) >> 15)


def test_quarter_weights_known_answer(orc):
    src = np.zeros((2, 2, 1), np.uint8)
    src[..., 0] = [[10, 50], [90, 250]]
    mx = np.array([[0.25]], np.float32)
    my = np.array([[0.75]], np.float32)
    # fx = 8, fy = 24: w00 = 24*8, w01 = 8*8, w10 = 24*24, w11 = 8*24  (/1024)
    want = (10 * 24 * 8 + 50 * 8 * 8 + 90 * 24 * 24 + 250 * 8 * 24 + 512) >> 10
    assert int(orc.remap_u8(src, mx, my, interp=1)[0, 0, 0]) == want == 103


def test_border_constant_rules(orc):
    src = np.full((4, 4, 3), 200, np.uint8)
    mx = np.array([[-1.0, -0.5, 3.5, 4.0, -2.0, 1.0, 3.0]], np.float32)
    my = np.array([[1.0, 1.0, 1.0, 1.0, 1.0, -1.5, 3.0]], np.float32)
    out = orc.remap_u8(src, mx, my, interp=1, border_value=37.0)   # Scalar(37,0,0,0)
    assert out[0, 0].tolist() == [37, 0, 0]                        # ix=-1, fx=0: all weight on the border tap
    assert out[0, 6].tolist() == [200, 200, 200]                   # last row/col, fx=fy=0: zero-weight border taps
    assert out[0, 1].tolist() == [(37 * 16 + 200 * 16 + 16) >> 5, 100, 100]
    assert out[0, 2].tolist() == [(200 * 16 + 37 * 16 + 16) >> 5, 100, 100]
    assert out[0, 3].tolist() == [37, 0, 0]                        # ix = W: fully outside
    assert out[0, 4].tolist() == [37, 0, 0]                        # ix + 1 < 0
    assert out[0, 5].tolist() == [37, 0, 0]                        # iy = -2


def test_nearest_rounds_half_to_even_and_border(orc):
    src = np.arange(5, dtype=np.uint8).reshape(1, 5, 1) * 10 + 10
    mx = np.array([[0.5, 1.5, 2.5, 3.5, 4.5, -0.5, -0.51]], np.float32)
    my = np.zeros_like(mx)
    out = orc.remap_u8(src, mx, my, interp=0, border_value=7.0)[0, :, 0].tolist()
    assert out == [10, 30, 30, 50, 50, 10, 7]    # 0.5->0, 1.5->2, 2.5->2, 3.5->4, 4.5->4, -0.5->0, -0.51->-1


def test_nan_inf_and_huge_coordinates_are_border(orc):
    src = rand_image(6, 6)
    mx = np.array([[np.nan, np.inf, -np.inf, 3e9, -3e9, 1e30]], np.float32)
    my = np.full_like(mx, 2.0)
    for interp in (0, 1):
        out = orc.remap_u8(src, mx, my, interp=interp, border_value=(9, 8, 7, 6))
        assert (out == np.array([9, 8, 7], np.uint8)).all()
        out = orc.remap_u8(src, my, mx, interp=interp, border_value=(9, 8, 7, 6))
        assert (out == np.array([9, 8, 7], np.uint8)).all()


@pytest.mark.parametrize("channels", [1, 3, 4])
def test_matches_independent_numpy_restatement(orc, channels):
    H, W, h, w = 61, 83, 70, 90
    src = rand_image(H, W, c=channels, seed=5)
    rng = np.random.default_rng(6)
    mx = rng.uniform(-10, W + 10, (h, w)).astype(np.float32)
    my = rng.uniform(-10, H + 10, (h, w)).astype(np.float32)
    mx[0, :4] = [np.nan, 1e20, -1e20, W - 1]
    got = orc.remap_u8(src, mx, my, interp=1, border_value=(37, 1, 2, 3))
    assert np.array_equal(got, np_remap_linear(src, mx, my, [37, 1, 2, 3]))


def test_valid_fill_and_threads(orc):
    src = rand_image(40, 40)
    rng = np.random.default_rng(7)
    mx = rng.uniform(0, 39, (33, 35)).astype(np.float32)
    my = rng.uniform(0, 39, (33, 35)).astype(np.float32)
    a = orc.remap_u8(src, mx, my, threads=1)
    b = orc.remap_u8(src, mx, my, threads=4)
    assert np.array_equal(a, b)
    valid = rng.random((33, 35)) > 0.5
    c = orc.valid_fill(a.copy(), valid, 123)
    assert (c[~valid] == 123).all() and np.array_equal(c[valid], a[valid])


def test_size_limits(orc):
    with pytest.raises(RuntimeError):
        orc.remap_u8(np.zeros((1, 32767, 1), np.uint8), np.zeros((1, 1), np.float32), np.zeros((1, 1), np.float32))


# ---- INTER_CUBIC (restated from the OpenCV source; parity unpinned, see oracle header) ---------------------------
def test_cubic_table_properties(orc):
    t = orc.cubic_table().astype(np.int64)
    assert (t.reshape(1024, 16).sum(axis=1) == 32768).all()          # every phase kernel sums to 2^15 after the fix-up
    assert t[0, 0, 1, 1] == 32767 and t[0, 0, 2, 2] == 1             # phase 0: short saturation of 1.0 + the +1 patch
    assert t[0, 0].sum() == 32768 and (t[0, 0] != 0).sum() == 2
    # Keys A=-0.75 at x = 1/2: (-0.09375, 0.59375, 0.59375, -0.09375) -> outer product * 32768
    w = np.array([-0.09375, 0.59375, 0.59375, -0.09375])
    assert np.array_equal(t[16, 16], np.rint(np.outer(w, w) * 32768).astype(np.int64))
    assert np.array_equal(t[16, 16], t[16, 16].T)


def test_cubic_identity_and_constant(orc):
    src = rand_image(23, 31)
    xx, yy = grid(23, 31)
    out = orc.remap_u8(src, xx, yy, interp=2, border_value=(9, 9, 9, 9))
    assert np.array_equal(out, src)                  # (S*32767 + S22*1 + 16384) >> 15 == S for 8-bit S
    const = np.full((12, 12, 3), 200, np.uint8)
    rng = np.random.default_rng(3)
    mx = rng.uniform(1.5, 9.5, (20, 20)).astype(np.float32)
    my = rng.uniform(1.5, 9.5, (20, 20)).astype(np.float32)
    assert (orc.remap_u8(const, mx, my, interp=2) == 200).all()       # kernels sum to exactly 2^15


def test_cubic_half_shift_known_answer_and_border(orc):
    row = np.array([[10, 20, 40, 80, 160, 250]], np.uint8)[:, :, None]
    src = np.repeat(row, 6, axis=0)
    mx = np.array([[2.5]], np.float32)
    my = np.array([[2.0]], np.float32)
    # fy = 0 -> vertical kernel (0, 32767.., 1 patch) collapses to the row; horizontal (-3072, 19456, 19456, -3072)/32768
    t = orc.cubic_table().astype(np.int64)[0, 16]
    want = int(((src[1:5, 1:5, 0].astype(np.int64) * t).sum() + 16384) >> 15)
    assert int(orc.remap_u8(src, mx, my, interp=2)[0, 0, 0]) == want
    far = orc.remap_u8(src, np.array([[-3.0, 6.0, 2.0]], np.float32), np.array([[2.0, 2.0, 9.0]], np.float32), interp=2, border_value=77.0)
    assert far[0, :, 0].tolist() == [77, 77, 77]      # window fully outside: x0+4 <= 0, x0 >= W, y0 >= H


# ---- INTER_LANCZOS4 (restated from the OpenCV source; parity unpinned, see oracle header) -------------------------
def lanczos4_weights_f64(x):
    """the defining formula: sinc(t)*sinc(t/4) on taps -3..4, normalised (float64, independent of the trig identity
    the restatement uses)"""
    t = x + 3 - np.arange(8)
    w = np.sinc(t) * np.sinc(t / 4.0)
    return w / w.sum()


def test_lanczos4_table_properties(orc):
    t = orc.lanczos4_table().astype(np.int64)
    assert (t.reshape(1024, 64).sum(axis=1) == 32768).all()          # every phase kernel sums to 2^15 after the fix-up
    assert t[0, 0, 3, 3] == 32767 and t[0, 0, 4, 4] == 1 and (t[0, 0] != 0).sum() == 2
    for f in (1, 7, 16, 31):                                         # 1-D weights agree with the sinc definition
        w = lanczos4_weights_f64(f / 32.0)
        row = t[0, f, 3].astype(np.float64) / 32767.0                # fy = 0: the vertical impulse selects tap row 3
        assert np.abs(row - w).max() < 2e-4, f
    assert np.array_equal(t[16, 16], t[16, 16].T)
    assert np.array_equal(t[5, 9], t[9, 5].T)


def test_lanczos4_identity_constant_and_border(orc):
    src = rand_image(23, 31)
    xx, yy = grid(23, 31)
    assert np.array_equal(orc.remap_u8(src, xx, yy, interp=4, border_value=(9, 9, 9, 9)), src)
    const = np.full((20, 20, 3), 200, np.uint8)
    rng = np.random.default_rng(3)
    mx = rng.uniform(3.5, 14.5, (20, 20)).astype(np.float32)
    my = rng.uniform(3.5, 14.5, (20, 20)).astype(np.float32)
    assert (orc.remap_u8(const, mx, my, interp=4) == 200).all()       # kernels sum to exactly 2^15
    far = orc.remap_u8(src, np.array([[-5.0, 34.0, 2.0, -4.5]], np.float32), np.array([[2.0, 2.0, 27.0, 2.0]], np.float32),
                       interp=4, border_value=77.0)
    assert far[0, :3, 0].tolist() == [77, 77, 77]     # window fully outside: x0+8 <= 0, x0 >= W, y0 >= H
    assert far[0, 3, 0] == 77                          # ix = floor(-4.5) = -5: x0 + 8 = 0, still fully outside


def test_lanczos4_known_answer_from_table(orc):
    src = rand_image(16, 16, c=1)
    mx = np.array([[7.25, 1.5]], np.float32)
    my = np.array([[8.75, 0.25]], np.float32)
    t = orc.lanczos4_table().astype(np.int64)
    out = orc.remap_u8(src[:, :, 0], mx, my, interp=4, border_value=50.0)
    win = src[8 - 3:8 + 5, 7 - 3:7 + 5, 0].astype(np.int64)
    want = int(np.clip(((win * t[24, 8]).sum() + 16384) >> 15, 0, 255))
    assert int(out[0, 0]) == want
    pad = np.full((16 + 8, 16 + 8), 50, np.int64)                    # taps outside the image read the border constant
    pad[4:20, 4:20] = src[:, :, 0]
    win = pad[0 - 3 + 4:0 + 5 + 4, 1 - 3 + 4:1 + 5 + 4]
    want = int(np.clip(((win * t[8, 16]).sum() + 16384) >> 15, 0, 255))
    assert int(out[0, 1]) == want
