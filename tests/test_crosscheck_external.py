"""Opportunistic cross-checks against the REAL third-party samplers (round-1 VERDICT "missing" #4).

Neither `cv2` nor `ffmpeg` exists in the build / GPU images, so every test here SKIPS there.  On the first box that has
either one they turn "parity unpinned"  into a measured fact:

* cv2.remap (DF:2001-2014): the oracle's restatement (CPU) and the HIP path (GPU) must be bit-identical to OpenCV for
  INTER_NEAREST / LINEAR / CUBIC / LANCZOS4 with BORDER_CONSTANT, including out-of-image, NaN/inf and huge coordinates.
* ffmpeg v360 (PC:310-314): EQ-SPEC is a build-defined map (documented sub-pixel deviation), so
  the check is orientation + a loose photometric bound on a smooth panorama, and the measured difference is printed.
"""
import shutil
import subprocess

import numpy as np
import pytest

from util import rand_image

INTERPS = [0, 1, 2, 4]


def _maps(h, w, H, W, seed):
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


@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("interp", INTERPS)
def test_oracle_remap_equals_cv2(orc, channels, interp):
    cv2 = pytest.importorskip("cv2")
    H, W, h, w = 97, 131, 75, 108
    src = rand_image(H, W, c=channels, seed=21)
    mx, my = _maps(h, w, H, W, 22)
    bv = (37.0, 0.0, 0.0, 0.0)
    want = cv2.remap(src, mx, my, interpolation=interp, borderMode=cv2.BORDER_CONSTANT, borderValue=bv)
    got = orc.remap_u8(src, mx, my, interp=interp, border_value=bv)
    assert np.array_equal(got.reshape(want.shape), want), f"oracle != cv2.remap (C={channels}, interp={interp}, cv2 {cv2.__version__})"


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("interp", INTERPS)
def test_hip_remap_equals_cv2(ctx, channels, interp):
    cv2 = pytest.importorskip("cv2")
    H, W, h, w = 97, 131, 75, 108
    src = rand_image(H, W, c=channels, seed=21)
    mx, my = _maps(h, w, H, W, 22)
    want = cv2.remap(src, mx, my, interpolation=interp, borderMode=cv2.BORDER_CONSTANT, borderValue=float(9))
    got = ctx.remap(src, mx, my, interpolation=interp, border_value=9.0)
    assert np.array_equal(got.reshape(want.shape), want), f"HIP != cv2.remap (C={channels}, interp={interp}, cv2 {cv2.__version__})"


@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("interp", INTERPS)
def test_oracle_remap_u16_equals_cv2(orc, channels, interp):
    """CV_16U: OpenCV's float-weight samplers (the accumulation order is the part restated from memory)"""
    cv2 = pytest.importorskip("cv2")
    H, W, h, w = 97, 131, 75, 108
    src = np.random.default_rng(24).integers(0, 65536, (H, W, channels), dtype=np.uint16)
    mx, my = _maps(h, w, H, W, 22)
    bv = (40000.0, 123.0, 0.0, 70000.0)
    want = cv2.remap(src, mx, my, interpolation=interp, borderMode=cv2.BORDER_CONSTANT, borderValue=bv)
    got = orc.remap_u16(src, mx, my, interp=interp, border_value=bv)
    assert np.array_equal(got.reshape(want.shape), want), f"oracle != cv2.remap CV_16U (C={channels}, interp={interp}, cv2 {cv2.__version__})"


@pytest.mark.gpu
@pytest.mark.parametrize("interp", INTERPS)
def test_hip_remap_u16_equals_cv2(ctx, interp):
    cv2 = pytest.importorskip("cv2")
    H, W, h, w = 97, 131, 75, 108
    src = np.random.default_rng(25).integers(0, 65536, (H, W, 3), dtype=np.uint16)
    mx, my = _maps(h, w, H, W, 22)
    want = cv2.remap(src, mx, my, interpolation=interp, borderMode=cv2.BORDER_CONSTANT, borderValue=float(9))
    got = ctx.remap(src, mx, my, interpolation=interp, border_value=9.0)
    assert np.array_equal(got, want), f"HIP != cv2.remap CV_16U (interp={interp}, cv2 {cv2.__version__})"


def _smooth_pano(H, W):
    x = np.arange(W, dtype=np.float64)[None, :] / W
    y = np.arange(H, dtype=np.float64)[:, None] / H
    img = np.empty((H, W, 3), np.uint8)
    img[..., 0] = (127.5 + 127.5 * np.sin(2 * np.pi * (3 * x + y))).astype(np.uint8)
    img[..., 1] = (255 * y * np.ones_like(x)).astype(np.uint8)
    img[..., 2] = (127.5 + 127.5 * np.cos(2 * np.pi * 2 * x) * np.ones_like(y)).astype(np.uint8)
    return img


def test_oracle_equirect_vs_ffmpeg_v360(orc, tmp_path):
    ffmpeg = shutil.which("ffmpeg")
    if not ffmpeg:
        pytest.skip("no ffmpeg on this box")
    H, W, size = 512, 1024, 200
    src = _smooth_pano(H, W)
    (tmp_path / "in.ppm").write_bytes(b"P6\n%d %d\n255\n" % (W, H) + src.tobytes())
    for yaw, pitch in [(0.0, 0.0), (60.0, 0.0), (-135.0, 30.0)]:
        out = tmp_path / "out.ppm"
        vf = (f"v360=input=equirect:output=rectilinear:w={size}:h={size}:yaw={yaw}:pitch={pitch}:roll=0:"
              f"h_fov=100:v_fov=100:interp=linear")
        subprocess.run([ffmpeg, "-hide_banner", "-loglevel", "error", "-y", "-i", str(tmp_path / "in.ppm"), "-vf", vf,
                        "-frames:v", "1", "-pix_fmt", "rgb24", str(out)], check=True)
        raw = out.read_bytes()
        ref = np.frombuffer(raw[-size * size * 3:], np.uint8).reshape(size, size, 3)
        got = orc.equirect_views_u8(src, [orc.make_view(yaw, pitch, 100.0, 100.0, size, size)])[0]
        d = np.abs(got.astype(int) - ref.astype(int))
        print(f"v360 vs EQ-SPEC yaw={yaw} pitch={pitch}: mean |d| = {d.mean():.3f}, max = {d.max()}")
        assert d.mean() < 3.0, "orientation / convention mismatch against ffmpeg v360"
