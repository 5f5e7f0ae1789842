import sys, pathlib, numpy as np
ROOT = pathlib.Path("/root/repo")
for p in (ROOT, ROOT / "360cam-pgm-3dgs-tools_amd", ROOT / "tests", ROOT / "tests" / "tools"):
    sys.path.insert(0, str(p))
import gs360
from gs360 import fisheye as fe
import bench_configs as bc
from util import TEMPLATE_CALIB
ctx = gs360.Context(0, n_slots=1)
cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
c = fe.SensorCalibration("0", "equisolid_fisheye", 4000, 4000, cal_kw["f"], cal_kw["cx"], cal_kw["cy"], cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)[:6]
tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
imgs = {"X": bc.synth(4000, 4000, 1), "Y": bc.synth(4000, 4000, 2)}
dev = {k: ctx.to_device(v) for k, v in imgs.items()}
plans, outs = {}, {}
for s in specs:
    t = tables[s["view_id"]]
    d = (ctx.to_device(t["map_x"]), ctx.to_device(t["map_y"]), ctx.to_device(np.ascontiguousarray(t["valid"], np.uint8)))
    plans[s["view_id"]] = ctx.map_plan(*d, 1750, 1750)
    outs[s["view_id"]] = ctx.alloc(1750 * 1750 * 3)
jobs = [(dev[tables[s["view_id"]]["lens_key"]], 4000, 4000, plans[s["view_id"]], True, 1750, 1750, 0, outs[s["view_id"]]) for s in specs]
with ctx.options(table_stage=1, table_stage_rows=int(sys.argv[1]) if len(sys.argv) > 1 else 32, table_stage_wgs=int(sys.argv[2]) if len(sys.argv) > 2 else 0):
    for _ in range(50):
        ctx.remap_plans_dev(jobs, 3, interp=1)
    ctx.sync(0)
raw = ctx.download(outs[specs[1]["view_id"]], (1750 * 1750 * 3,))
q = raw[:8 * 1000].view(np.uint64)
for who, base in (("b0 w0", 0), ("b0 w5", 200), ("b9 w0", 400), ("b9 w5", 600)):
    r = q[8 + base: 8 + base + 100].reshape(20, 5).astype(np.int64)
    print(who)
    t0 = r[0, 0]
    for g in range(18):
        a = r[g]
        if "w0" in who:
            print(f"  g{g:2d} start {a[0]-t0:8d} | issue {a[1]-a[0]:6d} land {a[2]-a[1]:6d} - {a[3]-a[2]:6d} barrier {a[4]-a[3]:6d} | iter {a[4]-a[0]:6d}")
        else:
            print(f"  g{g:2d} start {a[0]-t0:8d} | prologue {a[1]-a[0]:6d} render {a[3]-a[1]:6d} barrier {a[4]-a[3]:6d} | iter {a[4]-a[0]:6d}")
