"""calls with the fused keep-mask (40 random disks, pack pass inside the timed loop) through the kernels the library picks with srcmajor=0
(gather / LDS-staged) and through the source-major kernel (forced), next to the unmasked source-major call; [n] = tile rows of the last plan"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np
import gs360
from util import ring_views, HFOV_12MM, HFOV_14MM, HFOV_17MM, PRESET_FULL360, PRESET_FISHEYELIKE
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
def bench(W, H, F, specs, label, variants):
    frames = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
    yy, xx = np.ogrid[:H, :W]
    m = np.full((H, W), 255, np.uint8)
    for _ in range(40):
        cy, cx, r = int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(40, 400))
        m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0
    masks = [ctx.to_device(np.ascontiguousarray(np.roll(m, 97 * k, axis=1))) for k in range(F)]
    views = [gs360.View.make(*s) for s in specs]
    dsts = [ctx.alloc(s[4] * s[5] * 3) for _ in range(F) for s in specs]
    for rep in range(2):
        res = []
        for name, opts, msk in variants:
            with ctx.options(**opts):
                def run(n):
                    for _ in range(n): ctx.equirect_views_dev(frames, W, H, 3, views, dsts, masks=masks if msk else None)
                run(1); ctx.sync(0)
                k = ctx.get_option("last_eq_kernel")
                t0 = time.time()
                while time.time() - t0 < 0.15: run(4)
                ctx.sync(0)
                ctx.event_record(0, 0); run(20); ctx.event_record(0, 1)
                res.append(f"{name} k{k}[{ctx.get_option('last_srcmajor_rows')}] {ctx.event_elapsed_ms(0, 0, 1) / 20 * 1e3 / F:.2f}")
        print(f"{label} F={F}: " + "; ".join(res), flush=True)
    for b in frames + dsts + masks: ctx.free(b)
full = [(float(y), float(p), HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360]
V = [("mask default", dict(srcmajor=0), True), ("mask srcmajor", dict(srcmajor=1), True), ("mask srcmajor r16", dict(srcmajor=1, srcmajor_rows=16), True), ("nomask srcmajor", dict(srcmajor=1), False)]
W, H = 7680, 3840
bench(W, H, 4, full, "cfg3", V)
bench(W, H, 16, ring_views(6, 800, HFOV_12MM), "cfg2", V)
bench(W, H, 4, [(float(y), float(p), HFOV_14MM, HFOV_14MM, 1200, 1200) for y, p in PRESET_FULL360], "full360 1200", V)
bench(W, H, 4, [(float(y), float(p), HFOV_17MM, HFOV_17MM, 1024, 1024) for y, p in PRESET_FISHEYELIKE], "fisheyelike 1024", V)
bench(W, H, 8, ring_views(8, 1600, HFOV_12MM), "8K 8x1600", V)
fish = [(float(y), float(p), HFOV_17MM, HFOV_17MM, 2048, 2048) for y, p in PRESET_FISHEYELIKE]
bench(W, H, 4, fish, "cfg5", V)
bench(W, H, 16, full, "cfg3 F=16", V)
bench(5760, 2880, 8, ring_views(8, 1600, HFOV_12MM), "cfg1 (5760: period 720, not whole keep dwords)", V)
