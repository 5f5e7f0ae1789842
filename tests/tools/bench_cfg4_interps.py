#!/usr/bin/env python3
"""cfg4 (2 x 4000^2 fisheye -> 6 x 1750^2, table mode, one batched launch) for every cv2 interpolation, 8- and 16-bit: ms per pair.  Informational."""
import sys, pathlib, numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent; sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT/'360cam-pgm-3dgs-tools_amd')); sys.path.insert(0, str(ROOT/'tests')); sys.path.insert(0, str(ROOT/'tests'/'tools'))
import gs360
from gs360 import fisheye as fe
from bench_configs import synth, time_steps
from util import TEMPLATE_CALIB
ctx = gs360.Context(0, n_slots=1)
cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
c = fe.SensorCalibration("0", "equisolid_fisheye", 4000, 4000, cal_kw["f"], cal_kw["cx"], cal_kw["cy"], cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)[:6]
tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
imgs = {"X": synth(4000, 4000, 1), "Y": synth(4000, 4000, 2)}
dev = {k: ctx.to_device(v) for k, v in imgs.items()}
d_tab = {v: (ctx.to_device(t["map_x"]), ctx.to_device(t["map_y"]), ctx.to_device(np.ascontiguousarray(t["valid"], np.uint8))) for v, t in tables.items()}
d_out = {v: ctx.alloc(1750 * 1750 * 3) for v in tables}
jobs = [(dev[tables[s["view_id"]]["lens_key"]], 4000, 4000) + tuple(d_tab[s["view_id"]]) + (1750, 1750, 0, d_out[s["view_id"]]) for s in specs]
for interp in (0, 1, 2, 4):
    ms = time_steps(ctx, lambda: ctx.remap_tables_dev(jobs, 3, interp=interp, border_value=(0, 0, 0, 0), slot=0), 20)
    print("interp", interp, "ms_per_pair", round(ms, 4))
# the same pair as 16-bit lens images (CV_16U samplers, gs360_remap_tables_u16)
dev16 = {k: ctx.to_device((v.astype(np.uint16) * 257) ^ np.uint16(3)) for k, v in imgs.items()}
d_out16 = {v: ctx.alloc(1750 * 1750 * 6) for v in tables}
jobs16 = [(dev16[tables[s["view_id"]]["lens_key"]], 4000, 4000) + tuple(d_tab[s["view_id"]]) + (1750, 1750, 0, d_out16[s["view_id"]]) for s in specs]
for interp in (0, 1, 2, 4):
    ms = time_steps(ctx, lambda: ctx.remap_tables_dev(jobs16, 3, interp=interp, border_value=(0, 0, 0, 0), slot=0, dtype=np.uint16), 10)
    print("u16 interp", interp, "ms_per_pair", round(ms, 4))
ctx.close()
