"""tile height (srcmajor_rows, adapt off) x images per workgroup x frames per call for single level rings: the evidence for the half-height
tiles of small jobs (profiles/r05/srcmajor_small_jobs.txt; its first section swept rows 32/28/24/20/16 x images 0/4/6/12 at 1-2 frames the same way)"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np, math
import gs360
from util import ring_views, HFOV_12MM, HFOV_14MM, PRESET_FULL360
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
pool = {}
def bench(W, H, F, specs, label, variants):
    if (W, H) not in pool: pool[(W, H)] = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(8)]
    frames = pool[(W, H)][:F]
    views = [gs360.View.make(*s) for s in specs]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
    res = []
    for name, opts in variants:
        with ctx.options(**opts):
            def run(n):
                for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
            run(1); ctx.sync(0)
            t0 = time.time()
            while time.time() - t0 < 0.12: run(8)
            ctx.sync(0)
            ctx.event_record(0, 0); run(40); ctx.event_record(0, 1)
            res.append(f"{name}[{ctx.get_option('last_srcmajor_rows')},{ctx.get_option('last_srcmajor_images')}] {ctx.event_elapsed_ms(0, 0, 1) / 40 * 1e3 / F:.2f}")
    print(f"{label} F={F}: " + "; ".join(res), flush=True)
    for b in dsts: ctx.free(b)
G = ("gather", dict(srcmajor=0))
def sm(r, g=0): return (f"r{r}g{g}", dict(srcmajor=1, srcmajor_rows=r, srcmajor_images=g, srcmajor_adapt=0))
W, H = 7680, 3840
for F in (1, 2, 3, 4, 6, 8):
    bench(W, H, F, ring_views(6, 800, HFOV_12MM), "cfg2", [G, sm(32), sm(16), sm(16, 12), sm(16, 6)])
for F in (1, 2, 3, 4, 6):
    bench(5760, 2880, F, ring_views(8, 1600, HFOV_12MM), "cfg1", [G, sm(16), sm(8), sm(8, 8), sm(8, 4)])
for F in (1, 2, 4):
    bench(W, H, F, ring_views(6, 1200, HFOV_12MM), "6x1200", [G, sm(32), sm(16), sm(8)])
    bench(W, H, F, ring_views(8, 1600, HFOV_12MM), "8x1600", [G, sm(32), sm(16), sm(8)])
    bench(W, H, F, ring_views(12, 800, HFOV_12MM), "12x800", [G, sm(32), sm(16), sm(8)])
