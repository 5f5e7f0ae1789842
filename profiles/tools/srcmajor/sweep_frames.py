"""gather vs source-major (forced) for calls of 1, 2 and 16 frames over level rings: the evidence for the frame-count condition of the auto rule"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np, math
import gs360
from util import ring_views, HFOV_12MM
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
def bench(W, H, F, specs, label):
    frames = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
    views = [gs360.View.make(*s) for s in specs]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
    step = W / (2 * math.pi) * 2 * math.tan(math.radians(specs[0][2]) / 2) / specs[0][4]
    res = []
    for name, opts in [("gather", dict(srcmajor=0)), ("srcmajor", dict(srcmajor=1))]:
        with ctx.options(**opts):
            def run(n):
                for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
            run(2); ctx.sync(0)
            t0 = time.time()
            while time.time() - t0 < 0.12: run(10)
            ctx.sync(0)
            ctx.event_record(0, 0); run(40); ctx.event_record(0, 1)
            res.append(ctx.event_elapsed_ms(0, 0, 1) / 40 * 1e3 / F)
            shape = (ctx.get_option("last_srcmajor_rows"), ctx.get_option("last_srcmajor_images"))
    print(f"{label} N={len(specs)} step {step:.2f} F={F}: gather {res[0]:.2f} srcmajor {res[1]:.2f} ratio {res[1]/res[0]:.2f}  (tile rows {shape[0]}, images {shape[1]})", flush=True)
    for b in frames + dsts: ctx.free(b)
W, H = 7680, 3840
for F in (int(a) for a in (sys.argv[1:] or ["1", "2", "3", "4", "16"])):
    for n, size in ((6, 800), (6, 1200), (6, 1600), (6, 2096), (8, 1024), (8, 1600), (12, 800), (5, 1224)):
        if F == 16 and n * size * size > 6 * 1700 * 1700: continue
        bench(W, H, F, ring_views(n, size, HFOV_12MM), f"8K->{n}x{size}")
    bench(5760, 2880, F, ring_views(8, 1600, HFOV_12MM), "cfg1")
    bench(3840, 1920, F, ring_views(6, 400, HFOV_12MM), "4K->6x400")
