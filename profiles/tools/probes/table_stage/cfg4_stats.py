import sys, numpy as np
sys.path.insert(0, "/root/repo/360cam-pgm-3dgs-tools_amd"); sys.path.insert(0, "/root/repo/tests")
from gs360 import fisheye as fe
from util import TEMPLATE_CALIB
cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
c = fe.SensorCalibration("0", "equisolid_fisheye", 4000, 4000, cal_kw["f"], cal_kw["cx"], cal_kw["cy"], cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)
tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
np.savez("/root/repo/scratch/r06/cfg4_tables.npz", **{f"{v}_{k}": np.asarray(t[k]) for v, t in tables.items() for k in ("map_x", "map_y", "valid")})
W = H = 4000
for TW, TH in ((64, 32), (64, 16), (128, 16), (32, 32), (64, 64)):
    tot_box = tot_px = tot_valid = 0; maxbox = 0; nt = 0; empty = 0
    for s in specs[:6]:
        t = tables[s["view_id"]]
        mx, my, valid = t["map_x"], t["map_y"], np.asarray(t["valid"], bool)
        sx = np.rint(mx * 32).astype(np.int64); sy = np.rint(my * 32).astype(np.int64)
        ix, iy = sx >> 5, sy >> 5
        ok = valid & (ix >= 0) & (iy >= 0) & (ix <= W - 2) & (iy <= H - 2)
        h, w = mx.shape
        for ty in range(0, h, TH):
            for tx in range(0, w, TW):
                o = ok[ty:ty+TH, tx:tx+TW]
                nt += 1
                tot_px += o.size
                if not o.any():
                    empty += 1; continue
                xs = ix[ty:ty+TH, tx:tx+TW][o]; ys = iy[ty:ty+TH, tx:tx+TW][o]
                x0 = (3 * xs.min()) & ~15
                wb = (((3 * xs.max() - x0) & ~3) + 12 + 15) // 16 * 16
                nr = ys.max() - ys.min() + 2
                b = wb * nr
                tot_box += b; maxbox = max(maxbox, b); tot_valid += o.sum()
    print(f"tile {TW}x{TH}: tiles {nt} empty {empty} box MB {tot_box/1e6:.1f} max box {maxbox} B, valid px {tot_valid/1e6:.2f} M of {tot_px/1e6:.2f} M; box B/valid px {tot_box/max(tot_valid,1):.2f}")
