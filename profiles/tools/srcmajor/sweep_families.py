"""ring families (a level ring + pitched mirror pairs of one ring size) through the source-major kernel (forced) vs the gather / staged
kernels, over ring size, minification and frames per call: the evidence for the auto rule of multi-ring calls"""
import sys, time, math
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np
import gs360
from util import HFOV_14MM, HFOV_17MM, PRESET_FULL360, PRESET_FISHEYELIKE
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
W, H = 7680, 3840
pool = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(16)]
def bench(F, specs, label):
    views = [gs360.View.make(*s) for s in specs]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
    frames = pool[:F]
    step = W / (2 * math.pi) * 2 * math.tan(math.radians(specs[0][2]) / 2) / specs[0][4]
    res = []
    for opts in (dict(srcmajor=0), dict(srcmajor=1)):
        with ctx.options(**opts):
            def run(n):
                for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
            run(1); ctx.sync(0)
            k = ctx.get_option("last_eq_kernel")
            pct = ctx.get_option("last_srcmajor_box_pct")
            t0 = time.time()
            while time.time() - t0 < 0.12: run(4)
            ctx.sync(0)
            reps = max(4, 64 // F)
            ctx.event_record(0, 0); run(reps); ctx.event_record(0, 1)
            res.append((k, ctx.event_elapsed_ms(0, 0, 1) / reps * 1e3 / F))
    print(f"{label} views {len(specs)} step {step:.2f} F={F}: kernel {res[0][0]} {res[0][1]:.2f}  source-major (kernel {res[1][0]}) {res[1][1]:.2f} us/frame  ratio {res[1][1]/res[0][1]:.2f}  boxes {pct} %", flush=True)
    for b in dsts: ctx.free(b)
def family(n, pitches, hfov, size, phase=0.5):
    v = [(i * 360.0 / n, 0.0, hfov, hfov, size, size) for i in range(n)]
    for p in pitches:
        v += [((i + phase) * 360.0 / n, s * p, hfov, hfov, size, size) for i in range(n) for s in (1, -1)]
    return v
for F in (int(a) for a in (sys.argv[1:] or ['1', '2', '4', '16'])):
    for size in (2096, 1600, 1200, 800):
        if F * size * size > 16 * 1700 * 1700: continue
        bench(F, [(float(y), float(p), HFOV_14MM, HFOV_14MM, size, size) for y, p in PRESET_FULL360], f"full360coverage {size}")
    bench(F, [(float(y), float(p), HFOV_17MM, HFOV_17MM, 2048, 2048) for y, p in PRESET_FISHEYELIKE], "fisheyelike 2048") if F < 16 else None
    bench(F, [(float(y), float(p), HFOV_17MM, HFOV_17MM, 1024, 1024) for y, p in PRESET_FISHEYELIKE], "fisheyelike 1024")
    bench(F, family(3, [35], 100.0, 1600), "3 + 3 + 3")
    bench(F, family(5, [30], HFOV_14MM, 1200), "5 + 5 + 5")
    bench(F, family(4, [45], 90.0, 1200), "4 + 4 + 4 at +/-45")
    bench(F, family(8, [], 0, 0)[:0] + [((i + 0.5) * 45.0, s * 30.0, HFOV_14MM, HFOV_14MM, 1200, 1200) for i in range(8) for s in (1, -1)], "pair of 8 alone")
    bench(F, [((i) * 60.0, s * 25.0, 100.0, 100.0, 1200, 1200) for i in range(6) for s in (1, -1)], "pair of 6 alone")
    bench(F, family(4, [40], 90.0, 1200), "4 + 4 + 4 at +/-40")
    bench(F, family(4, [35], 90.0, 1200), "4 + 4 + 4 at +/-35")
