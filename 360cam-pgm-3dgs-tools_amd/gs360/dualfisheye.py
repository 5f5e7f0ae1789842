"""GPU execution of the dual-fisheye pair pipeline (the per-pair work of DF:1910-2064).

One PairRenderer per device holds the remap tables of a sensor pair resident in HBM (the reference builds them
once per run and shares them between its worker threads, DF:2582-2592, DF:2776-2804) and renders all views of one
X/Y pair: upload the two lens images once, one table-remap launch per view (cv2.remap semantics + fused
`out[~valid] = mask_value`), download.  `fused=True` evaluates the map in-kernel instead (FE-SPEC v1, no table
traffic; <= 0.01 px from the reference tables).

The tables of a run never change, so pairs go through MAP PLANS (include/gs360.h): each table converted once to the fixed
point cv2.remap derives from it on every call, 5 bytes per pixel instead of 9 -- same results, less to read per pair.  Plans are made
on first use per (view, sampler class); sources beyond 4079 pixels keep the float tables.  GS360_MAP_PLANS=0
turns them off (A/B).
"""
import os
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import capi
from .fisheye import SensorCalibration, UndistortTables

INTERPOLATIONS = {"nearest": 0, "linear": 1, "cubic": 2, "lanczos4": 4}   # cv2.INTER_* values (DF:59-64)


def engine_interpolation(flag: int) -> int:
    """cv2 flag -> what the engine implements: all four of the tool's choices (nearest, linear, cubic, lanczos4)."""
    if flag not in (capi.INTERP_NEAREST, capi.INTERP_LINEAR, capi.INTERP_CUBIC, capi.INTERP_LANCZOS4):
        raise ValueError("unsupported interpolation flag {}".format(flag))
    return flag


class PairRenderer:
    def __init__(self, ctx: capi.Context, sensors: Dict[str, SensorCalibration], specs: Sequence[Dict[str, object]],
                 tables: Optional[Dict[str, Dict[str, object]]], undistort: Optional[Dict[str, UndistortTables]],
                 lens_fov_deg: float, fused: bool = False):
        self.ctx = ctx
        self.sensors = sensors
        self.specs = list(specs)
        self.tables = tables
        self.lens_fov_deg = float(lens_fov_deg)
        self.fused = bool(fused)
        self.lock = threading.Lock()       # one pair at a time per device context (slot 0)
        self.dev_tables = {}
        if tables and not fused:
            for vid, t in tables.items():
                self.dev_tables[vid] = (ctx.to_device(t["map_x"]), ctx.to_device(t["map_y"]),
                                        ctx.to_device(np.ascontiguousarray(t["valid"], np.uint8)))
        self._scratch = None                # grow-only output buffer shared by the per-view launches of a pair
        self._plans = {}                    # (kind, id, nearest) -> map plan handle
        self.use_plans = os.environ.get("GS360_MAP_PLANS", "1") not in ("0", "off", "no")
        self.dev_undistort = {}
        for sid, u in (undistort or {}).items():
            self.dev_undistort[sid] = (ctx.to_device(u.map_x), ctx.to_device(u.map_y),
                                       ctx.to_device(np.ascontiguousarray(u.valid_mask, np.uint8)))

    # ---------------------------------------------------------------------------------------------
    def _plan(self, kind, key, maps, out_hw, interp, shape, dtype):
        """-> the map plan for these tables and this sampler class, or None where plans do not apply"""
        H, W = shape[0], shape[1]
        if not self.use_plans or max(H, W) > self.ctx.MAP_PLAN_MAX_DIM:
            return None
        nearest = interp == capi.INTERP_NEAREST
        k = (kind, key, nearest)
        if k not in self._plans:
            self._plans[k] = self.ctx.map_plan(maps[0], maps[1], maps[2], out_hw[0], out_hw[1], nearest=nearest, slot=0)
        return self._plans[k]

    def close(self):
        """release the plans (the context's own close releases everything else)"""
        with self.lock:
            for plan in self._plans.values():
                if self.ctx.handle:
                    self.ctx.map_plan_free(plan)
            self._plans.clear()

    def _remap(self, d_src, shape, maps, out_hw, interp, border, valid_fill, dtype=np.uint8, plan_key=None):
        H, W, C = shape
        h, w = out_hw
        need = h * w * C * np.dtype(dtype).itemsize
        if self._scratch is None or self._scratch.nbytes < need:
            if self._scratch is not None:
                self.ctx.free(self._scratch)
            self._scratch = self.ctx.alloc(need)
        plan = self._plan("undistort", plan_key, maps, out_hw, interp, shape, dtype) if plan_key is not None else None
        if plan is not None:
            self.ctx.remap_plans_dev([(d_src, H, W, plan, valid_fill is not None, h, w, valid_fill if valid_fill is not None else 0,
                                       self._scratch)], C, interp=interp, border_value=border, slot=0, dtype=dtype)
        else:
            self.ctx.remap_table_dev(d_src, H, W, C, maps[0], maps[1], maps[2] if valid_fill is not None else None, h, w,
                                     self._scratch, interp=interp, border_value=border,
                                     fill_value=valid_fill if valid_fill is not None else 0, slot=0, dtype=dtype)
        return self.ctx.download(self._scratch, (h, w, C), dtype=dtype, slot=0)

    def _remap_views(self, dev, imgs, dmask, interp, border, valid_fill):
        """All views of the pair in ONE batched launch (no per-view launch tails), then the downloads.  `dmask` set:
        the per-lens mask images are the sources (DF:2031-2043), else the lens images."""
        dtype = np.uint8
        if dmask is None and any(v.dtype == np.uint16 for v in imgs.values()):
            if any(v.dtype != np.uint16 for v in imgs.values()):
                raise RuntimeError("the two lens images differ in bit depth")
            dtype = np.uint16                 # CV_16U samplers, the same batched launch (gs360_remap_tables_u16)
        esz = np.dtype(dtype).itemsize
        jobs, shapes, bufs, planned = [], [], [], []
        for spec in self.specs:
            vid = str(spec["view_id"])
            key = self.tables[vid]["lens_key"]
            if dmask is not None:
                if key not in dmask:
                    raise RuntimeError("Mask source missing for lens {}".format(key))
                d_src, (H, W, C) = dmask[key]
            else:
                d_src, (H, W, C) = dev[key], imgs[key].shape
            h, w = int(spec["height"]), int(spec["width"])
            d_dst = self.ctx.alloc(h * w * C * esz)
            bufs.append(d_dst)
            mx, my, va = self.dev_tables[vid]
            jobs.append((d_src, H, W, mx, my, va if valid_fill is not None else None, h, w,
                         valid_fill if valid_fill is not None else 0, d_dst))
            plan = self._plan("view", vid, (mx, my, va), (h, w), interp, (H, W, C), dtype)
            planned.append(None if plan is None else (d_src, H, W, plan, valid_fill is not None, h, w,
                                                      valid_fill if valid_fill is not None else 0, d_dst))
            shapes.append((vid, (h, w, C)))
        try:
            channels = {s[1][2] for s in shapes}
            if len(channels) != 1:
                raise RuntimeError("the two lens images differ in channel count")
            if all(p is not None for p in planned):
                self.ctx.remap_plans_dev(planned, channels.pop(), interp=interp, border_value=border, slot=0, dtype=dtype)
            else:
                self.ctx.remap_tables_dev(jobs, channels.pop(), interp=interp, border_value=border, slot=0, dtype=dtype)
            return {vid: self.ctx.download(b, shape, dtype=dtype, slot=0) for (vid, shape), b in zip(shapes, bufs)}
        finally:
            for b in bufs:
                self.ctx.free(b)

    def render_pair(self, image_x: np.ndarray, image_y: np.ndarray, sensor_id_x: str, sensor_id_y: str, *,
                    interpolation: int, mask_outside_model: bool, mask_value: int,
                    mask_x: Optional[np.ndarray] = None, mask_y: Optional[np.ndarray] = None,
                    want_fisheye: bool = False, want_perspective: bool = True,
                    color_stage=None, want_color: bool = False):
        """-> dict(perspective={view_id: img}, masks={view_id: img}, fisheye={'X': img, 'Y': img}, color={'X','Y'})

        `color_stage` (gs360.color.ColorStage) converts both lens images on the device right after the upload, i.e.
        before every resampling step, as load_prepared_input_image does on the host (DF:728-743); `want_color`
        returns the converted images (the --save-color-corrected-output files, DF:1953-1959)."""
        interp = engine_interpolation(interpolation)
        out = {"perspective": {}, "masks": {}, "fisheye": {}, "color": {}}
        with self.lock:
            imgs = {"X": _hwc(image_x), "Y": _hwc(image_y)}
            if color_stage is not None:
                for v in imgs.values():
                    color_stage.check_image(v.shape, v.dtype)
            dev = {k: self.ctx.to_device(v) for k, v in imgs.items()}
            dmask = {}
            for k, m in (("X", mask_x), ("Y", mask_y)):
                if m is not None:
                    dmask[k] = (self.ctx.to_device(_hwc(m)), _hwc(m).shape)
            try:
                if color_stage is not None:
                    for k, v in imgs.items():
                        color_stage.apply_dev(self.ctx, dev[k], v.shape, red_index=0, slot=0, dtype=v.dtype)   # arrays here are RGB(A)
                if want_color:
                    for k, v in imgs.items():
                        out["color"][k] = self.ctx.download(dev[k], v.shape, dtype=v.dtype, slot=0) if color_stage is not None else v
                # borderValue=float(mask_value) -> cv::Scalar(v,0,0,0): only channel 0 gets v, and channel 0 of a
                # cv2.imread image is BLUE.  Arrays here are RGB(A), so the value goes to index 2 for colour images.
                C_in = imgs["X"].shape[2]
                border = (0.0, 0.0, float(mask_value), 0.0) if C_in >= 3 else (float(mask_value), 0.0, 0.0, 0.0)
                fill = int(mask_value) if mask_outside_model else None
                if want_fisheye:
                    for key, sid in (("X", sensor_id_x), ("Y", sensor_id_y)):
                        c = self.sensors[sid]
                        if imgs[key].shape[:2] != (c.height, c.width):
                            raise RuntimeError("Resolution mismatch for {} lens: got {}x{}, expected {}x{}".format(
                                key, imgs[key].shape[1], imgs[key].shape[0], c.width, c.height))
                        out["fisheye"][key] = self._remap(dev[key], imgs[key].shape, self.dev_undistort[sid],
                                                          (c.height, c.width), interp, border, fill, dtype=imgs[key].dtype, plan_key=sid)
                if want_perspective:
                    if self.fused and imgs["X"].dtype == np.uint16:
                        raise RuntimeError("16-bit lens images need table mode (the fused-map kernel is 8-bit)")
                    if self.fused:
                        self._render_fused(out, imgs, dev, sensor_id_x, sensor_id_y, interp, mask_outside_model, mask_value)
                    else:
                        out["perspective"] = self._remap_views(dev, imgs, None, interp, border, fill)
                    if dmask:
                        if self.fused:   # masks always go through the table path (nearest, border 0, invalid -> 0)
                            raise RuntimeError("mask rendering needs table mode")
                        out["masks"] = self._remap_views(None, None, dmask, capi.INTERP_NEAREST, (0.0, 0.0, 0.0, 0.0),
                                                         0 if mask_outside_model else None)
            finally:
                for b in dev.values():
                    self.ctx.free(b)
                for b, _s in dmask.values():
                    self.ctx.free(b)
        return out

    def _render_fused(self, out, imgs, dev, sid_x, sid_y, interp, mask_outside, mask_value):
        views, calibs, srcs, dsts = [], [], [], []
        C = imgs["X"].shape[2]
        for key, sid in (("X", sid_x), ("Y", sid_y)):
            c = self.sensors[sid]
            if imgs[key].shape[:2] != (c.height, c.width):
                raise RuntimeError("fused map mode needs images of the calibrated size {}x{} (lens {} is {}x{})".format(
                    c.width, c.height, key, imgs[key].shape[1], imgs[key].shape[0]))
        for spec in self.specs:
            vid = str(spec["view_id"])
            t = self.tables[vid]
            key = t["lens_key"]
            c = self.sensors[sid_x if key == "X" else sid_y]
            views.append(capi.View.make(t["yaw_rel_deg"], spec["pitch_deg"], spec["hfov_deg"], spec["vfov_deg"],
                                        spec["width"], spec["height"]))
            calibs.append(capi.Calib.make(c.width, c.height, c.f, c.cx, c.cy, c.k1, c.k2, c.k3, c.k4, c.p1, c.p2, c.b1, c.b2))
            srcs.append(dev[key])
            dsts.append(self.ctx.alloc(int(spec["width"]) * int(spec["height"]) * C))
        try:
            self.ctx.fisheye_views_dev(srcs, calibs, C, views, self.lens_fov_deg, dsts, interp=interp,
                                       mask_outside=mask_outside, mask_value=mask_value, slot=0)
            for spec, d in zip(self.specs, dsts):
                out["perspective"][str(spec["view_id"])] = self.ctx.download(
                    d, (int(spec["height"]), int(spec["width"]), C), slot=0)
        finally:
            for d in dsts:
                self.ctx.free(d)


def _hwc(a: np.ndarray) -> np.ndarray:
    """uint8 stays uint8; uint16 sources (cv2.imread(IMREAD_UNCHANGED), DF:735) stay uint16"""
    a = np.ascontiguousarray(a, dtype=np.uint16 if np.asarray(a).dtype == np.uint16 else np.uint8)
    return a if a.ndim == 3 else a[:, :, None]
