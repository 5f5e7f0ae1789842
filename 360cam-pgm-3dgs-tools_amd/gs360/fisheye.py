"""Host-side geometry of the dual-fisheye path: calibration, SFM10 view layout and the float32 remap tables.

These tables are what the reference feeds to cv2.remap (gs360_DualFisheyeDistortionCalibration.py, "DF"), so
the "exact" mode of the drop-in builds them the same way -- NumPy float32 arithmetic in the reference's
operation order -- and samples them on the GPU with gs360_remap_table_u8.  They are pinned bit-for-bit against
vectors captured from the reference (tests/golden/df_goldens.npz, NumPy >= 2).

  SensorCalibration / load_metashape_calibration    DF:67-85, DF:754-828
  brown_distort                                     DF:975-1005
  undistort_tables / auto_undistort_zoom            DF:1008-1170
  sfm10_specs / view_fov_deg                        DF:1243-1307
  rotate_pitch_yaw                                  DF:1310-1339
  perspective_tables                                DF:1759-1823
  choose_lens_tables                                DF:1857-1907
"""
import math
import os
import pathlib
import xml.etree.ElementTree as ET
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import hostmem

SUPPORTED_MODELS = {"equisolid_fisheye"}
F32 = np.float32
# threads for the independent remap-table builds of choose_lens_tables (each holds a few 1750^2 float32 temporaries)
_TABLE_BUILD_THREADS = int(os.environ.get("GS360_TABLE_BUILD_THREADS", "0")) or min(16, hostmem.effective_cpus())


@dataclass
class SensorCalibration:
    sensor_id: str
    model_type: str
    width: int
    height: int
    f: float
    cx: float = 0.0
    cy: float = 0.0
    k1: float = 0.0
    k2: float = 0.0
    k3: float = 0.0
    k4: float = 0.0
    p1: float = 0.0
    p2: float = 0.0
    b1: float = 0.0
    b2: float = 0.0

    @property
    def centre(self) -> Tuple[float, float]:
        return (self.width * 0.5) + self.cx, (self.height * 0.5) + self.cy


@dataclass
class UndistortTables:
    map_x: np.ndarray
    map_y: np.ndarray
    valid_mask: np.ndarray
    undistort_zoom: float


# ---- calibration XML ------------------------------------------------------------------------------
def _child_float(node, tag: str, default: float = 0.0) -> float:
    hit = node.find(tag) if node is not None else None
    return float(hit.text) if hit is not None and hit.text is not None else default


def _pick_calibration(sensor):
    """adjusted > initial > first (DF:754-764)"""
    nodes = sensor.findall("calibration")
    for wanted in ("adjusted", "initial"):
        for n in nodes:
            if n.attrib.get("class", "").strip().lower() == wanted:
                return n
    return nodes[0] if nodes else None


def load_metashape_calibration(xml_path) -> Tuple[Dict[str, SensorCalibration], Dict[str, str]]:
    """-> ({sensor_id: calibration}, {camera label: sensor_id})"""
    root = ET.parse(str(xml_path)).getroot()
    sensors: Dict[str, SensorCalibration] = {}
    for sensor in root.findall(".//sensors/sensor"):
        sid = sensor.attrib.get("id", "").strip()
        calib = _pick_calibration(sensor) if sid else None
        if calib is None:
            continue
        model = (calib.attrib.get("type") or sensor.attrib.get("type") or "").strip().lower()
        res = calib.find("resolution")
        if res is None or len(res) == 0:
            # the reference writes `calib.find(..) or sensor.find(..)` (DF:792): a childless Element is falsy, so the
            # sensor-level node is what gets used (the shipped template carries identical values in both)
            res = sensor.find("resolution")
        if res is None:
            continue
        w, h = int(res.attrib.get("width", "0")), int(res.attrib.get("height", "0"))
        if w <= 0 or h <= 0:
            continue
        c = SensorCalibration(sid, model, w, h, *[_child_float(calib, t) for t in
                                                   ("f", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2")])
        if c.f > 0.0:
            sensors[sid] = c
    labels = {}
    for cam in root.findall(".//cameras/camera"):
        label, sid = cam.attrib.get("label", "").strip(), cam.attrib.get("sensor_id", "").strip()
        if label and sid:
            labels[label] = sid
    return sensors, labels


# ---- view layout ----------------------------------------------------------------------------------
def parse_sensor_mm(text: str) -> Tuple[float, float]:
    vals = []
    for tok in str(text or "").strip().replace("x", " ").replace("X", " ").replace(",", " ").split():
        try:
            vals.append(float(tok))
        except ValueError:
            pass
    if not vals:
        raise ValueError("Invalid --perspective-sensor-mm: '{}'".format(text))
    w, h = float(vals[0]), float(vals[1] if len(vals) > 1 else vals[0])
    if w <= 0.0 or h <= 0.0:
        raise ValueError("Sensor dimensions must be positive: '{}'".format(text))
    return w, h


def view_fov_deg(focal_mm: float, sensor_mm: str) -> Tuple[float, float]:
    f = float(focal_mm)
    if f <= 0.0:
        raise ValueError("--perspective-focal-mm must be > 0")
    sw, sh = parse_sensor_mm(sensor_mm)
    clampf = lambda d: max(1.0, min(179.9, d))   # noqa: E731
    return clampf(math.degrees(2.0 * math.atan(sw / (2.0 * f)))), clampf(math.degrees(2.0 * math.atan(sh / (2.0 * f))))


SFM10_LAYOUT = (("A", 0, 0.0, 0), ("A_U", 0, 0.0, +1), ("A_D", 0, 0.0, -1), ("B", +1, 0.0, 0), ("E", -1, 180.0, 0),
                ("F", 0, 180.0, 0), ("F_U", 0, 180.0, +1), ("F_D", 0, 180.0, -1), ("G", +1, 180.0, 0), ("J", -1, 360.0, 0))


def sfm10_specs(output_size: int, focal_mm: float, sensor_mm: str, yaw_delta_deg: float,
                pitch_delta_deg: float) -> List[Dict[str, object]]:
    """10 views around the front (yaw 0) and back (yaw 180) lens axes (DF:1281-1292)."""
    size = int(output_size)
    if size <= 0:
        raise ValueError("--perspective-size must be > 0")
    dy, dp = float(yaw_delta_deg), float(pitch_delta_deg)
    if dy <= 0.0 or dy >= 180.0:
        raise ValueError("--perspective-yaw-delta-deg must be in (0, 180)")
    if dp <= 0.0 or dp >= 89.9:
        raise ValueError("--perspective-pitch-delta-deg must be in (0, 89.9)")
    hfov, vfov = view_fov_deg(focal_mm, sensor_mm)
    return [{"view_id": vid, "yaw_deg": float(base + ysign * dy), "pitch_deg": float(psign * dp),
             "hfov_deg": float(hfov), "vfov_deg": float(vfov), "width": size, "height": size}
            for vid, ysign, base, psign in SFM10_LAYOUT]


def wrap_angle_deg(a: float) -> float:
    return ((float(a) + 180.0) % 360.0) - 180.0


# ---- float32 table builders -----------------------------------------------------------------------
def brown_distort(x: np.ndarray, y: np.ndarray, c: SensorCalibration):
    """Brown radial (k1..k4) + tangential (p1, p2) in normalised coordinates; float32 in, float32 out."""
    r2 = (x * x) + (y * y)
    r4 = r2 * r2
    r6 = r4 * r2
    r8 = r4 * r4
    radial = 1.0 + (c.k1 * r2) + (c.k2 * r4) + (c.k3 * r6) + (c.k4 * r8)
    xy = x * y
    xd, yd = x * radial, y * radial
    if c.p1 != 0.0 or c.p2 != 0.0:
        xd = xd + (c.p1 * (r2 + (2.0 * x * x))) + (2.0 * c.p2 * xy)
        yd = yd + (c.p2 * (r2 + (2.0 * y * y))) + (2.0 * c.p1 * xy)
    return xd, yd, r2


def sensor_coords(xd, yd, c: SensorCalibration):
    cx0, cy0 = c.centre
    return cx0 + (xd * c.f) + (xd * c.b1) + (yd * c.b2), cy0 + (yd * c.f)


def rotate_pitch_yaw(v: np.ndarray, yaw_deg: float, pitch_deg: float) -> np.ndarray:
    """pitch about +X, then yaw about +Y (camera frame x right, y up, z forward)"""
    p, yw = math.radians(pitch_deg), math.radians(yaw_deg)
    cp, sp, cy, sy = math.cos(p), math.sin(p), math.cos(yw), math.sin(yw)
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    y1 = (cp * y) + (sp * z)
    z1 = (-sp * y) + (cp * z)
    out = np.empty_like(v)
    out[..., 0] = (cy * x) + (sy * z1)
    out[..., 1] = y1
    out[..., 2] = (-sy * x) + (cy * z1)
    return out


def _pixel_centre_axis(n: int) -> np.ndarray:
    return ((np.arange(n, dtype=F32) + 0.5) / float(n)) * 2.0 - 1.0


def perspective_tables(c: SensorCalibration, yaw_deg, pitch_deg, hfov_deg, vfov_deg, out_w: int, out_h: int,
                       lens_fov_deg: float):
    """(map_x, map_y, valid) from one equisolid lens straight to a pinhole view."""
    uu, vv = np.meshgrid(_pixel_centre_axis(out_w), _pixel_centre_axis(out_h))
    hf = math.radians(max(1e-3, min(179.9, hfov_deg)))
    vf = math.radians(max(1e-3, min(179.9, vfov_deg)))
    rays = np.empty((out_h, out_w, 3), dtype=F32)
    rays[..., 0] = np.tan(hf * 0.5) * uu          # np.float64 scalar * float32 array (NumPy >= 2: float64 product)
    rays[..., 1] = np.tan(vf * 0.5) * (-vv)
    rays[..., 2] = 1.0
    rays = rays / np.maximum(np.linalg.norm(rays, axis=2, keepdims=True), 1e-12)
    rays = rotate_pitch_yaw(rays, yaw_deg, pitch_deg)
    rx, ry, rz = rays[..., 0], rays[..., 1], rays[..., 2]
    theta = np.arccos(np.clip(rz, -1.0, 1.0))
    theta_max = math.radians(max(1.0, min(360.0, lens_fov_deg)) * 0.5)
    rho = np.sqrt((rx * rx) + (ry * ry))
    scale = np.zeros_like(rho, dtype=F32)
    nz = rho > 1e-12
    scale[nz] = (2.0 * np.sin(theta[nz] * 0.5) / rho[nz]).astype(F32)   # equisolid: r = 2 sin(theta / 2)
    xd, yd, _ = brown_distort(rx * scale, -ry * scale, c)                # image y grows downwards
    mx, my = sensor_coords(xd, yd, c)
    valid = (theta <= theta_max) & (mx >= 0.0) & (mx <= (c.width - 1)) & (my >= 0.0) & (my <= (c.height - 1))
    return mx.astype(F32), my.astype(F32), valid


def _undistort_for_zoom(c: SensorCalibration, dst_x, dst_y, zoom: float, lens_fov_deg: float):
    cx0, cy0 = c.centre
    den_y, den_x = c.f, c.f + c.b1
    if abs(den_y) < 1e-12 or abs(den_x) < 1e-12:
        raise ValueError("Invalid focal/b1 configuration caused division by zero.")
    y0 = (dst_y - cy0) / den_y
    x0 = (dst_x - cx0 - (y0 * c.b2)) / den_x
    xd, yd, r2 = brown_distort(x0 / zoom, y0 / zoom, c)
    sx, sy = sensor_coords(xd, yd, c)
    theta = 2.0 * np.arcsin(np.clip(np.sqrt(np.maximum(r2, 0.0)) * 0.5, 0.0, 1.0))
    in_model = theta <= math.radians(max(1.0, min(360.0, float(lens_fov_deg))) * 0.5)
    in_bounds = (sx >= 0.0) & (sx <= (c.width - 1)) & (sy >= 0.0) & (sy <= (c.height - 1))
    return sx, sy, in_model & in_bounds, in_model


def auto_undistort_zoom(c: SensorCalibration, sample_count: int = 192, lens_fov_deg: float = 190.0) -> float:
    """smallest zoom whose model-valid samples all stay inside the sensor (x1.2 growth, then 20 bisections)"""
    w, h = int(c.width), int(c.height)
    n = max(32, int(sample_count))
    gx, gy = np.meshgrid(np.linspace(0.0, float(w - 1), n, dtype=F32), np.linspace(0.0, float(h - 1), n, dtype=F32))

    def overflow(zoom: float) -> float:
        sx, sy, _v, model = _undistort_for_zoom(c, gx, gy, zoom, lens_fov_deg)
        if not np.any(model):
            return 0.0
        sx, sy = sx[model], sy[model]
        return float(max(float(np.max(np.maximum(0.0, -sx))), float(np.max(np.maximum(0.0, sx - (w - 1)))),
                         float(np.max(np.maximum(0.0, -sy))), float(np.max(np.maximum(0.0, sy - (h - 1))))))

    if overflow(1.0) <= 0.0:
        return 1.0
    low = high = 1.0
    for _ in range(20):
        high *= 1.2
        if overflow(high) <= 0.0:
            break
    if overflow(high) > 0.0:
        return high
    for _ in range(20):
        mid = (low + high) * 0.5
        if overflow(mid) <= 0.0:
            high = mid
        else:
            low = mid
    return high


def undistort_tables(c: SensorCalibration, undistort_zoom: Optional[float], lens_fov_deg: float) -> UndistortTables:
    if c.model_type not in SUPPORTED_MODELS:
        raise ValueError("Unsupported sensor model '{}' (supported: {}).".format(
            c.model_type, ", ".join(sorted(SUPPORTED_MODELS))))
    gx, gy = np.meshgrid(np.arange(int(c.width), dtype=F32), np.arange(int(c.height), dtype=F32))
    zoom = float(undistort_zoom) if undistort_zoom is not None else auto_undistort_zoom(c, lens_fov_deg=float(lens_fov_deg))
    zoom = max(1e-6, zoom)
    sx, sy, valid, _m = _undistort_for_zoom(c, gx, gy, zoom, float(lens_fov_deg))
    return UndistortTables(sx.astype(F32), sy.astype(F32), valid, zoom)


def choose_lens_tables(sensors: Dict[str, SensorCalibration], sensor_id_x: str, sensor_id_y: str,
                       specs: Sequence[Dict[str, object]], lens_x_yaw_deg: float, lens_y_yaw_deg: float,
                       lens_fov_deg: float) -> Dict[str, Dict[str, object]]:
    """per view: evaluate both lenses, keep the one with the larger valid ratio (ties: smaller |yaw_rel|).
    The 2 x len(specs) table builds are independent NumPy passes (0.1-0.4 s each at 1750^2, and NumPy releases the GIL): they run on a
    small thread pool -- the reference builds them one after the other (DF:1857-1907), which is most of the start-up time of a run."""
    lenses = (("X", lens_x_yaw_deg, sensor_id_x), ("Y", lens_y_yaw_deg, sensor_id_y))

    def build(spec, lens):
        key, lens_yaw, sid = lens
        yaw_rel = wrap_angle_deg(float(spec["yaw_deg"]) - lens_yaw)
        mx, my, valid = perspective_tables(sensors[sid], yaw_rel, float(spec["pitch_deg"]), float(spec["hfov_deg"]),
                                           float(spec["vfov_deg"]), int(spec["width"]), int(spec["height"]), lens_fov_deg)
        return (float(np.mean(valid)), -abs(yaw_rel)), key, mx, my, valid, yaw_rel, sid

    tasks = [(spec, lens) for spec in specs for lens in lenses]
    n_threads = max(1, min(len(tasks), _TABLE_BUILD_THREADS))
    if n_threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=n_threads, thread_name_prefix="gs360-maps") as pool:
            built = list(pool.map(lambda t: build(*t), tasks))
    else:
        built = [build(*t) for t in tasks]
    out: Dict[str, Dict[str, object]] = {}
    for i, spec in enumerate(specs):
        best = None
        for cand in built[2 * i:2 * i + 2]:               # X first: a tie keeps X, as the reference's loop does
            if best is None or cand[0] > best[0]:
                best = cand
        out[str(spec["view_id"])] = {"lens_key": best[1], "map_x": best[2], "map_y": best[3], "valid": best[4],
                                     "yaw_rel_deg": best[5], "sensor_id": best[6]}
    return out
