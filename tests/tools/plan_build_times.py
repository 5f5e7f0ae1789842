"""Source-major plan builds per geometry (the library's own clock, read-only options srcmajor_plan_builds / srcmajor_plan_build_us) next to
the wall time of the call that builds: first build of a context (scratch blocks + first use of the sort kernels), then further geometries."""
import sys, pathlib, time, numpy as np
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "360cam-pgm-3dgs-tools_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import gs360
from util import PRESET_FULL360, HFOV_14MM, HFOV_12MM, ring_views
ctx = gs360.Context(0, n_slots=1)
W, H = 7680, 3840
frames = [ctx.to_device(np.zeros((H, W, 3), np.uint8)) for _ in range(4)]
def build(specs, label):
    views = [gs360.View.make(*s) for s in specs]
    outs = [ctx.alloc(s[4] * s[5] * 3) for _ in range(4) for s in specs]
    b0, u0 = ctx.get_option("srcmajor_plan_builds"), ctx.get_option("srcmajor_plan_build_us")
    t0 = time.perf_counter()
    ctx.equirect_views_dev(frames, W, H, 3, views, outs)
    ctx.sync(0)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"{label}: builds {ctx.get_option('srcmajor_plan_builds') - b0}, build {(ctx.get_option('srcmajor_plan_build_us') - u0) / 1e3:.2f} ms, call {dt:.2f} ms, kernel {ctx.get_option('last_eq_kernel')}")
    for b in outs:
        ctx.free(b)
with ctx.options(srcmajor=1):
    build([(y, p, HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360], "cfg3 first")
    build([(y, p, HFOV_14MM, HFOV_14MM, 1596, 1596) for y, p in PRESET_FULL360], "cfg3-like second (1596^2)")
    build([(y, p, HFOV_14MM + 1.0, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360], "cfg3-like third (fov + 1)")
    build(ring_views(6, 800, HFOV_12MM), "cfg2")
    build(ring_views(6, 804, HFOV_12MM), "cfg2-like (804^2)")
    build(ring_views(8, 1600, HFOV_12MM), "default 8 x 1600^2 on 8K")
ctx.close()
