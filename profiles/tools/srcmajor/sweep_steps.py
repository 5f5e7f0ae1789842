"""gather vs source-major (forced) over ring size N and minification step: the evidence for the N / step conditions of the auto rule"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np, math
import gs360
from util import ring_views, HFOV_12MM, HFOV_14MM
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
def bench(W, H, F, specs, label):
    frames = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
    views = [gs360.View.make(*s) for s in specs]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
    step = W / (2 * math.pi) * 2 * math.tan(math.radians(specs[0][2]) / 2) / specs[0][4]
    out = []
    for name, opts in (("gather", dict(srcmajor=0)), ("srcmajor", dict(srcmajor=1))):
        with ctx.options(**opts):
            def run(n):
                for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
            run(2); ctx.sync(0)
            t0 = time.time()
            while time.time() - t0 < 0.12: run(5)
            ctx.sync(0)
            ctx.event_record(0, 0); run(30); ctx.event_record(0, 1)
            out.append(ctx.event_elapsed_ms(0, 0, 1) / 30 * 1e3 / F)
    print(f"{label} F={F}: step {step:.2f} overlap {len(specs)*specs[0][2]/360:.2f}  gather {out[0]:.2f}  srcmajor {out[1]:.2f} us/frame  ratio {out[1]/out[0]:.2f}", flush=True)
    for b in frames + dsts: ctx.free(b)
W, H = 7680, 3840
for n in (4, 5, 6, 8, 12):
    for step in (1.5, 1.75, 2.0, 2.5, 3.0):
        size = int(round(W / (2 * math.pi) * 2 * math.tan(math.radians(HFOV_12MM) / 2) / step / 4)) * 4
        if n * size * size > 6 * 2600 * 2600: continue
        bench(W, H, 4, ring_views(n, size, HFOV_12MM), f"8K -> {n}x{size}^2")
for n, size in ((4, 1600), (6, 800), (6, 1600), (8, 1600)):
    bench(W, H, 1, ring_views(n, size, HFOV_12MM if n != 4 else HFOV_14MM), f"8K -> {n}x{size}^2")
bench(5760, 2880, 1, ring_views(8, 1600, HFOV_12MM), "cfg1 single frame")
bench(5760, 2880, 8, ring_views(8, 1600, HFOV_12MM), "cfg1")
bench(7680, 3840, 4, ring_views(4, 1600, HFOV_14MM), "cfg3 level ring")
bench(7680, 3840, 4, [(y, 0.0, 93.0, 93.0, 2048, 2048) for y in (0, 36, 144, 180, -144, -36)], "cfg5 level views (not a ring)")
