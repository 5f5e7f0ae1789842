#!/usr/bin/env python3
"""Capture colour-stage golden vectors by IMPORTING the reference (container-only, needs /root/reference).

The dual-fisheye tool's input colour pipeline (.cube 3D LUT, trilinear; optional Rec.709 -> sRGB re-encode;
DF:494-712) is pure NumPy, so the reference's own functions are run here on seeded inputs and the inputs/outputs are
stored as data.  As in make_df_goldens.py a constants-only `cv2` module object satisfies the import guard
(DF:32-39); no cv2 function is emulated or called.

Note recorded with the vectors: `np.power` on float32 is implementation-defined (this container's NumPy uses an
AVX-512 SIMD routine whose results differ from the correctly rounded power for ~20 % of inputs), so the sRGB vectors
are tied to NumPy's float32 power; `power_probe_*` stores a probe of it so a test can tell whether the host it runs on
computes the same function.

    python tests/golden/make_color_goldens.py      ->  tests/golden/color_goldens.npz (+ .json)
"""
import json
import pathlib
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/cli_tools")
_flags = types.ModuleType("cv2")
_flags.INTER_NEAREST, _flags.INTER_LINEAR, _flags.INTER_CUBIC, _flags.INTER_LANCZOS4 = 0, 1, 2, 4
_flags.BORDER_CONSTANT = 0
sys.modules["cv2"] = _flags
import gs360_DualFisheyeDistortionCalibration as df  # noqa: E402  (reference; container-only)

HERE = pathlib.Path(__file__).resolve().parent
rng = np.random.default_rng(20260424)
arrays, meta = {}, {"_meta": {"numpy": np.__version__, "source": "reference DF:494-712 run on seeded inputs"}}


def cube_text(size, fn, dmin=None, dmax=None, title=True):
    """A .cube file body: red fastest (the layout DF:556-562 reshapes to [b][g][r])."""
    lines = []
    if title:
        lines += ['TITLE "synthetic {}"'.format(size), "# comment line", ""]
    lines.append("LUT_3D_SIZE {}".format(size))
    if dmin is not None:
        lines.append("DOMAIN_MIN {} {} {}".format(*dmin))
        lines.append("DOMAIN_MAX {} {} {}".format(*dmax))
    g = np.linspace(0.0, 1.0, size)
    for b in g:
        for gg in g:
            for r in g:
                lines.append("{:.6f} {:.6f} {:.6f}".format(*fn(r, gg, b)))
    return "\n".join(lines) + "\n"


def log_like(r, g, b):           # a contrast curve + channel mixing, mildly out of [0,1] so the clips are exercised
    m = 0.8 * r + 0.15 * g + 0.05 * b
    return (1.08 * m ** 0.6 - 0.03, 1.05 * (0.1 * r + 0.85 * g + 0.05 * b) ** 0.7 - 0.02,
            1.1 * (0.05 * r + 0.1 * g + 0.85 * b) ** 0.5 - 0.04)


def identity(r, g, b):
    return (r, g, b)


LUTS = {
    "mix17": cube_text(17, log_like),
    "id5": cube_text(5, identity, title=False),
    "dom9": cube_text(9, log_like, dmin=(0.0625, 0.0, 0.125), dmax=(0.9375, 1.0, 0.75)),
    "id2": cube_text(2, identity, title=False),
}

images = {
    "rgb": rng.integers(0, 256, (48, 64, 3), dtype=np.uint8),
    "rgba": rng.integers(0, 256, (24, 40, 4), dtype=np.uint8),
    "u16": rng.integers(0, 65536, (16, 24, 3), dtype=np.uint16),
}
# every 8-bit level on every channel, against a few fixed partner values
ramp = np.zeros((12, 256, 3), np.uint8)
lv = np.arange(256, dtype=np.uint8)
for row, (a, b) in enumerate(((0, 0), (255, 255), (128, 37), (9, 200))):
    ramp[row * 3 + 0] = np.stack([lv, np.full(256, a, np.uint8), np.full(256, b, np.uint8)], -1)
    ramp[row * 3 + 1] = np.stack([np.full(256, a, np.uint8), lv, np.full(256, b, np.uint8)], -1)
    ramp[row * 3 + 2] = np.stack([np.full(256, a, np.uint8), np.full(256, b, np.uint8), lv], -1)
images["ramp"] = ramp
for k, v in images.items():
    arrays["img_" + k] = v

with tempfile.TemporaryDirectory() as td:
    for name, text in LUTS.items():
        p = pathlib.Path(td) / (name + ".cube")
        p.write_text(text)
        lut = df.load_cube_lut(p)
        arrays["lut_{}_text".format(name)] = np.frombuffer(text.encode(), np.uint8)
        arrays["lut_{}_table".format(name)] = lut.table
        arrays["lut_{}_dmin".format(name)] = lut.domain_min
        arrays["lut_{}_dmax".format(name)] = lut.domain_max
        meta["lut_" + name] = {"size": int(lut.size)}
        for iname, img in images.items():
            if name in ("id2", "id5") and iname not in ("rgb", "ramp"):
                continue
            for space in ("srgb", "passthrough"):
                # the reference treats channel 0..2 as B,G,R (cv2.imread order, DF:697-699)
                out = df.apply_input_color_pipeline(img, lut, space)
                arrays["out_{}_{}_{}".format(name, iname, space)] = np.ascontiguousarray(out)
        # the float stage alone (DF:632-679) on a few hundred float triples
        f = rng.random((20, 25, 3), dtype=np.float32) * np.float32(1.2) - np.float32(0.1)
        arrays["tri_{}_in".format(name)] = f
        arrays["tri_{}_out".format(name)] = df.apply_cube_lut_trilinear(f, lut)

    # loader error behaviour (DF:494-562): message text per malformed file
    bad = {
        "nosize": "0 0 0\n1 1 1\n",
        "size1": "LUT_3D_SIZE 1\n0 0 0\n",
        "rows": "LUT_3D_SIZE 2\n0 0 0\n1 1 1\n",
        "domain": "LUT_3D_SIZE 2\nDOMAIN_MIN 0 0 0\nDOMAIN_MAX 1 0 1\n" + "0 0 0\n" * 8,
        "domain_short": "LUT_3D_SIZE 2\nDOMAIN_MIN 0 0\n" + "0 0 0\n" * 8,
    }
    meta["loader_errors"] = {}
    for k, text in bad.items():
        p = pathlib.Path(td) / "bad.cube"
        p.write_text(text)
        try:
            df.load_cube_lut(p)
            meta["loader_errors"][k] = None
        except Exception as exc:  # noqa: BLE001
            meta["loader_errors"][k] = [type(exc).__name__, str(exc).replace(str(p), "<path>")]
        arrays["bad_{}_text".format(k)] = np.frombuffer(text.encode(), np.uint8)

# transfer functions alone (DF:565-600)
x = np.concatenate([np.linspace(-0.25, 1.25, 3001, dtype=np.float32),
                    np.float32([0.081, np.nextafter(np.float32(0.081), np.float32(0)), 0.0031308, 0.018, 0.5, 1.0, 0.0])])
arrays["tf_in"] = x
arrays["tf_rec709_to_linear"] = df.rec709_to_linear(x)
arrays["tf_linear_to_srgb"] = df.linear_to_srgb(x)
arrays["tf_rec709_to_srgb"] = df.rec709_to_srgb(x)
arrays["q8_in"] = x
arrays["q8_out"] = df.float01_to_image(x, np.dtype(np.uint8))
arrays["q16_out"] = df.float01_to_image(x, np.dtype(np.uint16))
arrays["f01_u8"] = df.image_to_float01(np.arange(256, dtype=np.uint8))
# probe of this host's float32 power (see module docstring)
probe = rng.random(4096, dtype=np.float32)
arrays["power_probe_in"] = probe
arrays["power_probe_out_045"] = np.power(probe, 1.0 / 0.45)
arrays["power_probe_out_24"] = np.power(probe, 1.0 / 2.4)
meta["normalize"] = {v: df.normalize_lut_output_color_space(v) for v in ("native", "passthrough", "srgb", "SRGB ", "")}

np.savez_compressed(HERE / "color_goldens.npz", **arrays)
(HERE / "color_goldens.json").write_text(json.dumps(meta, indent=1, sort_keys=True) + "\n")
print("wrote", len(arrays), "arrays,", (HERE / "color_goldens.npz").stat().st_size, "bytes")
