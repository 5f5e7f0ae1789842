"""cfg1 / cfg2 / cfg3 in a loop of 16-frame calls (source-major forced or off): the program rocprofv3 runs for per-config counters (pmc_cfg.sh)"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np
import gs360
from util import HFOV_14MM, PRESET_FULL360, ring_views, HFOV_12MM
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sm = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
if which == "cfg3":
    W, H = 7680, 3840; specs = [(float(y), float(p), HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360]
elif which == "cfg1":
    W, H = 5760, 2880; specs = ring_views(8, 1600, HFOV_12MM)
else:
    W, H = 7680, 3840; specs = ring_views(6, 800, HFOV_12MM)
F = 16
frames = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
views = [gs360.View.make(*s) for s in specs]
dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
with ctx.options(srcmajor=sm, srcmajor_rows=16 if which == "cfg3" else 32):
    for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
    ctx.sync(0)
print("done", which, ctx.get_option("last_eq_kernel"))
