"""level rings at weak minification through the source-major kernel (forced) vs the gather kernels"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np
import gs360
from util import ring_views, HFOV_12MM, HFOV_14MM
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
def bench(W, H, F, specs, label, opts_list):
    frames = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
    views = [gs360.View.make(*s) for s in specs]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
    def run(n):
        for _ in range(n):
            ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
    for name, opts in opts_list:
        with ctx.options(**opts):
            try:
                run(2); ctx.sync(0)
            except Exception as e:
                print(label, name, "failed", e); continue
            k = ctx.get_option("last_eq_kernel")
            t0 = time.time()
            while time.time() - t0 < 0.15: run(5)
            ctx.sync(0)
            ctx.event_record(0, 0); run(30); ctx.event_record(0, 1)
            ms = ctx.event_elapsed_ms(0, 0, 1) / 30
            print(f"{label} [{name}] kernel {k}: {ms*1e3/F:.2f} us/frame", flush=True)
    for b in frames + dsts: ctx.free(b)
variants = [("gather", dict(srcmajor=0))] + [(f"srcmajor {bx}x{r}", dict(srcmajor=1, srcmajor_bx=bx, srcmajor_rows=r)) for bx, r in
            [(768, 32), (768, 16), (768, 8), (512, 16), (512, 8), (1024, 8), (384, 16)]]
bench(5760, 2880, 8, ring_views(8, 1600, HFOV_12MM), "cfg1 5.7K -> 8x1600^2", variants)
bench(7680, 3840, 4, ring_views(4, 1600, HFOV_14MM), "cfg3 level ring 8K -> 4x1600^2", variants)
bench(7680, 3840, 8, ring_views(6, 1200, HFOV_12MM), "8K -> 6x1200^2 (step 3.05)", variants)
bench(7680, 3840, 8, ring_views(8, 1024, HFOV_12MM), "8K -> 8x1024^2 (step 3.6)", variants)
