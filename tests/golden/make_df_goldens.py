#!/usr/bin/env python3
"""Capture dual-fisheye geometry golden vectors by IMPORTING the reference.

Container-only (needs /root/reference).  The reference module refuses to import
without OpenCV (DF:32-39); none of the NumPy map builders use it, so a
constants-only stand-in module object is registered for the duration of the
import (it carries the five integer flags DF:59-64 reads, nothing else -- no
cv2 function is emulated and none is called here).

Output (data only, no reference text):
  df_goldens.npz   float32/bool arrays produced by the reference's NumPy builders
  df_goldens.json  scalars: calibration, specs, lens choices, zooms, metadata

    python tests/golden/make_df_goldens.py
"""
import json
import math
import pathlib
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = pathlib.Path("/root/reference/cli_tools")
sys.path.insert(0, str(REF))

_flags = types.ModuleType("cv2")
_flags.INTER_NEAREST, _flags.INTER_LINEAR, _flags.INTER_CUBIC, _flags.INTER_LANCZOS4 = 0, 1, 2, 4
_flags.BORDER_CONSTANT = 0
sys.modules["cv2"] = _flags
import gs360_DualFisheyeDistortionCalibration as df  # noqa: E402  (reference; container-only)

HERE = pathlib.Path(__file__).resolve().parent
arrays = {}
meta = {"_meta": {"numpy": np.__version__, "python": sys.version.split()[0],
                  "source": "reference cli_tools/gs360_DualFisheyeDistortionCalibration.py"}}


def calib_dict(c):
    return {k: (getattr(c, k) if isinstance(getattr(c, k), (int, str)) else repr(getattr(c, k)))
            for k in ("sensor_id", "model_type", "width", "height", "f", "cx", "cy",
                      "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2")}


# ---- calibration (DF:767-828) -------------------------------------------------------
sensor_map, cam_to_sensor = df.load_metashape_calibration(df.DEFAULT_CAMERA_XML)
calib = sensor_map["0"]
meta["template_calibration"] = calib_dict(calib)
meta["template_camera_to_sensor_count"] = len(cam_to_sensor)
meta["template_camera_to_sensor_first"] = sorted(cam_to_sensor.items())[:4]

# a second, synthetic calibration exercising k4 / tangential / affinity terms
calib_full = df.SensorCalibration(
    sensor_id="syn", model_type="equisolid_fisheye", width=640, height=480,
    f=170.25, cx=3.5, cy=-2.25, k1=0.08, k2=-0.01, k3=0.002, k4=-0.0003,
    p1=0.0007, p2=-0.0004, b1=1.75, b2=-0.6)
meta["synthetic_calibration"] = calib_dict(calib_full)

# ---- SFM10 layout (DF:1243-1307) ----------------------------------------------------
specs = df.build_sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)
meta["sfm10_specs_default"] = [{k: (repr(v) if isinstance(v, float) else v) for k, v in s.items()} for s in specs]
specs_alt = df.build_sfm10_specs(512, 18.0, "36x24", 35.0, 25.0)
meta["sfm10_specs_alt"] = [{k: (repr(v) if isinstance(v, float) else v) for k, v in s.items()} for s in specs_alt]
meta["compute_view_fov_deg"] = [[f, s, [repr(x) for x in df.compute_view_fov_deg(f, s)]]
                                for f, s in [(14.0, "36 36"), (18.0, "36x24"), (0.05, "36 36"), (4000.0, "36 36")]]
meta["wrap_angle_deg"] = [[a, repr(df.wrap_angle_deg(a))] for a in [0.0, 180.0, -180.0, 190.0, 540.0, -181.0, 320.0]]

# ---- Brown distortion (DF:975-1005) -------------------------------------------------
rng = np.random.default_rng(20260424)
bx = rng.uniform(-1.6, 1.6, size=(48, 40)).astype(np.float32)
by = rng.uniform(-1.6, 1.6, size=(48, 40)).astype(np.float32)
arrays["brown_in_x"], arrays["brown_in_y"] = bx, by
for tag, c in (("tmpl", calib), ("full", calib_full)):
    xd, yd, r2 = df._apply_brown_distortion(bx, by, c)
    arrays[f"brown_{tag}_xd"], arrays[f"brown_{tag}_yd"], arrays[f"brown_{tag}_r2"] = xd, yd, r2
    assert xd.dtype == np.float32

# ---- rotation (DF:1310-1339) --------------------------------------------------------
vec = rng.normal(size=(33, 17, 3)).astype(np.float32)
arrays["rot_in"] = vec
rot_cases = [(0.0, 0.0), (40.0, 0.0), (0.0, 40.0), (-140.0, -40.0), (180.0, 30.0), (37.5, -89.0)]
meta["rot_cases"] = rot_cases
for i, (yw, pt) in enumerate(rot_cases):
    arrays[f"rot_out_{i}"] = df.rotate_view_vectors(vec, yw, pt)

# ---- direct perspective maps, small full shapes (DF:1759-1823) ----------------------
small_cases = [
    # name, calib, yaw, pitch, hfov, vfov, w, h, lens_fov
    ("s_A", "tmpl", 0.0, 0.0, 104.2500326978036, 104.2500326978036, 64, 64, 190.0),
    ("s_AU", "tmpl", 0.0, 40.0, 104.2500326978036, 104.2500326978036, 64, 64, 190.0),
    ("s_B", "tmpl", 40.0, 0.0, 104.2500326978036, 104.2500326978036, 96, 64, 190.0),
    ("s_J", "tmpl", -40.0, 0.0, 104.2500326978036, 104.2500326978036, 64, 96, 190.0),
    ("s_back", "tmpl", 140.0, 0.0, 104.2500326978036, 104.2500326978036, 64, 64, 190.0),
    ("s_side", "tmpl", 90.0, -40.0, 90.0, 60.0, 80, 48, 190.0),
    ("s_narrow", "tmpl", 10.0, 5.0, 30.0, 20.0, 48, 32, 120.0),
    ("s_wide", "tmpl", 0.0, 0.0, 179.9, 179.9, 64, 64, 190.0),
    ("s_full_A", "full", 0.0, 0.0, 100.0, 80.0, 72, 56, 190.0),
    ("s_full_tilt", "full", 25.0, -30.0, 100.0, 80.0, 72, 56, 180.0),
    ("s_odd", "tmpl", -72.5, 33.25, 75.0, 110.0, 257, 129, 200.0),
]
meta["small_cases"] = [[c[0], c[1]] + list(c[2:]) for c in small_cases]
for name, cname, yaw, pitch, hf, vf, w, h, lf in small_cases:
    c = calib if cname == "tmpl" else calib_full
    mx, my, valid = df.build_direct_perspective_map_for_lens(c, yaw, pitch, hf, vf, w, h, lf)
    arrays[f"{name}_mx"], arrays[f"{name}_my"], arrays[f"{name}_valid"] = mx, my, valid
    assert mx.dtype == np.float32 and valid.dtype == np.bool_

# ---- real-size maps (1750^2), strided samples + anchors -----------------------------
HF = float(specs[0]["hfov_deg"])
STRIDE = 25
anchor_rc = [(0, 0), (0, 1749), (874, 874), (875, 875), (1749, 0), (1000, 300), (1749, 1749), (3, 1234)]
meta["real_stride"] = STRIDE
meta["real_anchor_rc"] = anchor_rc
real_views = [("A", 0.0, 0.0), ("A_U", 0.0, 40.0), ("B", 40.0, 0.0), ("A_D", 0.0, -40.0), ("J", -40.0, 0.0)]
meta["real_views"] = real_views
for vid, yaw, pitch in real_views:
    mx, my, valid = df.build_direct_perspective_map_for_lens(calib, yaw, pitch, HF, HF, 1750, 1750, 190.0)
    arrays[f"real_{vid}_mx_s"] = mx[::STRIDE, ::STRIDE].copy()
    arrays[f"real_{vid}_my_s"] = my[::STRIDE, ::STRIDE].copy()
    arrays[f"real_{vid}_valid_s"] = valid[::STRIDE, ::STRIDE].copy()
    arrays[f"real_{vid}_anchor"] = np.array([[mx[r, c], my[r, c]] for r, c in anchor_rc], dtype=np.float32)
    arrays[f"real_{vid}_rows"] = np.stack([mx[0], my[0], mx[875], my[875], mx[1749], my[1749]])
    meta[f"real_{vid}_valid_ratio"] = repr(float(np.mean(valid)))

# ---- lens selection (DF:1857-1907) --------------------------------------------------
sel_specs = df.build_sfm10_specs(175, 14.0, "36 36", 40.0, 40.0)
sel = df.build_perspective_spec_maps({"0": calib}, "0", "0", sel_specs, 0.0, 180.0, 190.0)
meta["lens_choice_175"] = {k: v["lens_key"] for k, v in sel.items()}
meta["lens_valid_ratio_175"] = {k: repr(float(np.mean(v["valid"]))) for k, v in sel.items()}
for k, v in sel.items():
    arrays[f"sel175_{k}_mx"] = v["map_x"][::5, ::5].copy()
    arrays[f"sel175_{k}_my"] = v["map_y"][::5, ::5].copy()
# a rig whose lenses are not 0/180 and a narrow lens, to exercise the tie-break and partial validity
sel2 = df.build_perspective_spec_maps({"0": calib}, "0", "0", sel_specs, 20.0, -150.0, 150.0)
meta["lens_choice_175_rig2"] = {k: v["lens_key"] for k, v in sel2.items()}
meta["lens_valid_ratio_175_rig2"] = {k: repr(float(np.mean(v["valid"]))) for k, v in sel2.items()}
for k in ("B", "E", "A_U"):
    arrays[f"sel175rig2_{k}_valid"] = sel2[k]["valid"].copy()

# ---- undistort map + auto zoom (DF:1008-1170) ---------------------------------------
cache = df.build_remap_cache(calib, None, 190.0)
meta["undistort_zoom_template"] = repr(cache.undistort_zoom)
meta["undistort_valid_fraction_template"] = repr(float(np.mean(cache.valid_mask)))
US = 60
meta["undistort_stride"] = US
arrays["undist_tmpl_mx_s"] = cache.map_x[::US, ::US].copy()
arrays["undist_tmpl_my_s"] = cache.map_y[::US, ::US].copy()
arrays["undist_tmpl_valid_s"] = cache.valid_mask[::US, ::US].copy()
arrays["undist_tmpl_row1920"] = np.stack([cache.map_x[1920], cache.map_y[1920]])
# explicit zoom
cache2 = df.build_remap_cache(calib_full, 1.35, 170.0)
arrays["undist_full_mx"], arrays["undist_full_my"], arrays["undist_full_valid"] = (
    cache2.map_x, cache2.map_y, cache2.valid_mask)
meta["undistort_zoom_full_explicit"] = repr(cache2.undistort_zoom)
# auto zoom on calibrations that overflow at zoom 1 (growth + bisection branch)
zoom_cases = []
for f_scale, lf in ((1.0, 120.0), (1.6, 100.0), (2.5, 60.0), (1.0, 190.0)):
    c = df.SensorCalibration(**{**calib_full.__dict__, "f": calib_full.f * f_scale})
    z = df.estimate_auto_undistort_zoom(c, lens_fov_deg=lf)
    zoom_cases.append([f_scale, lf, repr(z)])
meta["auto_zoom_cases_full"] = zoom_cases
cache3 = df.build_remap_cache(df.SensorCalibration(**{**calib_full.__dict__, "f": calib_full.f * 1.6}), None, 100.0)
meta["undistort_zoom_full_auto_f1.6_lf100"] = repr(cache3.undistort_zoom)
arrays["undist_full_auto_mx"], arrays["undist_full_auto_my"], arrays["undist_full_auto_valid"] = (
    cache3.map_x[::4, ::4].copy(), cache3.map_y[::4, ::4].copy(), cache3.valid_mask[::4, ::4].copy())

np.savez_compressed(HERE / "df_goldens.npz", **arrays)
(HERE / "df_goldens.json").write_text(json.dumps(meta, indent=1, sort_keys=True) + "\n")
tot = sum(a.nbytes for a in arrays.values())
print("arrays:", len(arrays), "raw bytes:", tot, "file:", (HERE / "df_goldens.npz").stat().st_size)
