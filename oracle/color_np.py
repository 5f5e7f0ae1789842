"""CPU oracle of the dual-fisheye tool's input colour stage -- TEST INFRASTRUCTURE ONLY.

A NumPy restatement of reference cli_tools/gs360_DualFisheyeDistortionCalibration.py:565-725 (float01 conversion,
.cube trilinear lookup, Rec.709 -> sRGB re-encode, quantisation), evaluated per pixel with the reference's float32
operation order.  Only tests/ may import it; the product (gs360/color.py + gs360_color.hip) never does.

Pinned: bit-exact against vectors captured by running the reference's own functions in the build container
(tests/golden/make_color_goldens.py -> color_goldens.npz), for uint8 and uint16 images, both output spaces, LUT sizes
2/5/9/17, with and without a DOMAIN_MIN/MAX.  The sRGB re-encode goes through NumPy's float32 `power`, which is
implementation-defined (SIMD routines differ from libm by an ulp on ~20 % of inputs); the vectors carry a probe of it
and the sRGB golden comparison is exact only where the probe matches (else <= 1 level).
"""
import numpy as np

F32 = np.float32


def to_float01(img):                                   # DF:603-613
    if img.dtype == np.uint8:
        return img.astype(F32) / 255.0
    if img.dtype == np.uint16:
        return img.astype(F32) / 65535.0
    return np.clip(img.astype(F32), 0.0, 1.0)


def from_float01(v, dtype):                            # DF:616-628
    v = np.clip(v.astype(F32), 0.0, 1.0)
    if dtype == np.uint8:
        return np.rint(v * 255.0).astype(np.uint8)
    if dtype == np.uint16:
        return np.rint(v * 65535.0).astype(np.uint16)
    return v.astype(dtype)


def rec709_to_linear(x):                               # DF:565-574
    v = np.clip(x.astype(F32), 0.0, 1.0)
    out = np.empty_like(v)
    knee = v < 0.081
    out[knee] = v[knee] / 4.5
    out[~knee] = np.power((v[~knee] + 0.099) / 1.099, 1.0 / 0.45).astype(F32)
    return out


def linear_to_srgb(x):                                 # DF:577-587
    v = np.clip(x.astype(F32), 0.0, 1.0)
    out = np.empty_like(v)
    toe = v <= 0.0031308
    out[toe] = 12.92 * v[toe]
    out[~toe] = (1.055 * np.power(v[~toe], 1.0 / 2.4) - 0.055).astype(F32)
    return np.clip(out, 0.0, 1.0)


def trilinear(rgb, table, dmin, dmax):                 # DF:620-681
    """rgb: float32 (..., 3); table: float32 [b][g][r][3]."""
    flat = rgb.reshape(-1, 3).astype(F32)
    n1 = table.shape[0] - 1
    span = (dmax - dmin).reshape(1, 3)
    pos = np.clip((flat - dmin.reshape(1, 3)) / span, 0.0, 1.0) * float(n1)
    i0 = np.floor(pos).astype(np.int32)
    i1 = np.minimum(i0 + 1, n1)
    t = pos - i0.astype(F32)
    tr, tg, tb = t[:, 0:1], t[:, 1:2], t[:, 2:3]

    def along_r(bi, gi):
        lo, hi = table[bi, gi, i0[:, 0]], table[bi, gi, i1[:, 0]]
        return lo + (hi - lo) * tr

    def along_g(bi):
        lo, hi = along_r(bi, i0[:, 1]), along_r(bi, i1[:, 1])
        return lo + (hi - lo) * tg

    lo, hi = along_g(i0[:, 2]), along_g(i1[:, 2])
    return (lo + (hi - lo) * tb).reshape(rgb.shape)


def color_pipeline(image, table, dmin, dmax, space, red_index=2):
    """apply_input_color_pipeline (DF:684-725).  red_index=2: channels are B,G,R(,A) as cv2.imread delivers them
    (what the reference assumes); red_index=0: R,G,B(,A)."""
    if image.ndim < 3 or image.shape[2] < 3:
        raise ValueError("LUT-based input conversion requires at least 3-channel RGB image input")
    order = [0, 1, 2] if red_index == 0 else [2, 1, 0]
    rgb = image[..., :3][..., order]
    x = trilinear(to_float01(rgb), np.asarray(table, F32), np.asarray(dmin, F32), np.asarray(dmax, F32))
    if space == "srgb":
        x = linear_to_srgb(rec709_to_linear(x))
    elif space == "passthrough":
        x = np.clip(x, 0.0, 1.0)
    else:
        raise ValueError("Unexpected LUT output color space")
    q = from_float01(x, image.dtype)[..., order]
    if image.shape[2] == 3:
        return np.ascontiguousarray(q)
    out = image.copy()
    out[..., :3] = q
    return out
