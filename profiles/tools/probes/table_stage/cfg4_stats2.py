import sys, numpy as np
sys.path.insert(0, "/root/repo/360cam-pgm-3dgs-tools_amd"); sys.path.insert(0, "/root/repo/tests")
from gs360 import fisheye as fe
from util import TEMPLATE_CALIB
cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
c = fe.SensorCalibration("0", "equisolid_fisheye", 4000, 4000, cal_kw["f"], cal_kw["cx"], cal_kw["cy"], cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)
tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
W = H = 4000
for TW, TH in ((64, 4), (64, 8), (64, 2), (128, 4)):
    boxes = []
    for s in specs[:6]:
        t = tables[s["view_id"]]
        mx, my = t["map_x"], t["map_y"]
        sx = np.rint(mx * 32).astype(np.int64); sy = np.rint(my * 32).astype(np.int64)
        ix, iy = sx >> 5, sy >> 5
        h, w = mx.shape
        hh, ww = (h // TH) * TH, (w // TW) * TW
        X = ix[:hh, :ww].reshape(hh // TH, TH, ww // TW, TW); Y = iy[:hh, :ww].reshape(hh // TH, TH, ww // TW, TW)
        xmin, xmax = X.min(axis=(1, 3)), X.max(axis=(1, 3)); ymin, ymax = Y.min(axis=(1, 3)), Y.max(axis=(1, 3))
        x0 = (3 * xmin) & ~15
        wb = ((((3 * xmax - x0) & ~3) + 12) + 15) // 16 * 16
        nr = ymax - ymin + 2
        boxes.append((wb * nr).ravel())
    b = np.concatenate(boxes)
    print(f"tile {TW}x{TH}: n {b.size} total {b.sum()/1e6*18.375e6/(b.size*TW*TH):.1f} MB (scaled to all px) mean {b.mean():.0f} p50 {np.percentile(b,50):.0f} p90 {np.percentile(b,90):.0f} p99 {np.percentile(b,99):.0f} max {b.max()}")
