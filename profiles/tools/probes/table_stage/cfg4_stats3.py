import sys, numpy as np
sys.path.insert(0, "/root/repo/360cam-pgm-3dgs-tools_amd"); sys.path.insert(0, "/root/repo/tests")
from gs360 import fisheye as fe
from util import TEMPLATE_CALIB
cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
c = fe.SensorCalibration("0", "equisolid_fisheye", 4000, 4000, cal_kw["f"], cal_kw["cx"], cal_kw["cy"], cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)
tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
TW, TH = 64, 32
for s in specs[:6]:
    t = tables[s["view_id"]]
    mx, my = t["map_x"], t["map_y"]
    sx = np.rint(mx * 32).astype(np.int64); sy = np.rint(my * 32).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    h, w = mx.shape
    bb = rag = shear = 0
    for ty in range(0, h, TH):
        for tx in range(0, w, TW):
            X = ix[ty:ty+TH, tx:tx+TW].ravel(); Y = iy[ty:ty+TH, tx:tx+TW].ravel()
            y0, y1 = Y.min(), Y.max() + 1
            nr = y1 - y0 + 1
            # per box row: x range of taps in that row (taps at rows iy and iy+1)
            lo = np.full(nr, 1 << 30); hi = np.full(nr, -1)
            for dy in (0, 1):
                r = Y - y0 + dy
                np.minimum.at(lo, r, 3 * X); np.maximum.at(hi, r, 3 * X + 11)
            x0 = (lo.min()) & ~15
            bb += nr * ((hi.max() - x0 + 16) // 16 * 16)
            rag += (((hi - (lo & ~15)) + 16) // 16 * 16).sum()
            # affine shear: row start = a + b*r quantised to 16 B; choose b by fitting lo
            rr = np.arange(nr)
            b = np.polyfit(rr, lo, 1)[0]
            start = lo.min() + 0  # find intercept so that start_r <= lo_r for all r
            line = b * rr
            a = (lo - line).min()
            st = (np.floor((a + line) / 16) * 16).astype(np.int64)
            wmax = (hi - st).max()
            shear += nr * ((wmax + 16) // 16 * 16)
    print(s["view_id"], f"bbox {bb/1e6:.1f} MB ragged {rag/1e6:.1f} MB shear {shear/1e6:.1f} MB; texels*3 {len(np.unique(iy*4000+ix))*3/1e6:.1f}")
