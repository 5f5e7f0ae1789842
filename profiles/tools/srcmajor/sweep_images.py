"""images per workgroup (option srcmajor_images) against frames per call: the evidence for the rule in sm_launch"""
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
    res = []
    for name, opts in [("gather", dict(srcmajor=0))] + [(f"G{g}", dict(srcmajor=1, srcmajor_images=g)) for g in (0, 1, 2, 3, 4, 6, 12) if g == 0 or (2 * len(specs)) % g == 0]:
        with ctx.options(**opts):
            def run(n):
                for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
            run(2); ctx.sync(0)
            t0 = time.time()
            while time.time() - t0 < 0.12: run(10)
            ctx.sync(0)
            ctx.event_record(0, 0); run(40); ctx.event_record(0, 1)
            res.append((name, ctx.event_elapsed_ms(0, 0, 1) / 40 * 1e3 / F))
    print(f"{label} F={F}: " + "; ".join(f"{n} {t:.2f}" for n, t in res), flush=True)
    for b in frames + dsts: ctx.free(b)
for F in (1, 2, 3, 4, 8):
    bench(7680, 3840, F, ring_views(6, 800, HFOV_12MM), "cfg2")
for F in (1, 2, 4):
    bench(5760, 2880, F, ring_views(8, 1600, HFOV_12MM), "cfg1")
