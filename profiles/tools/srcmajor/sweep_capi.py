"""cfg2 through the C ABI with different source-major tile shapes (context options), HIP-event timing behind a settle phase"""
import sys, time
import pathlib; R = pathlib.Path(__file__).resolve().parents[3]; sys.path[:0] = [str(R / '360cam-pgm-3dgs-tools_amd'), str(R / 'tests'), str(R)]
import numpy as np
import gs360
from util import ring_views, HFOV_12MM
W, H, F = 7680, 3840, 16
ctx = gs360.Context(0, n_slots=1)
rng = np.random.default_rng(1)
frames = [ctx.to_device(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
views = [gs360.View.make(*s) for s in ring_views(6, 800, HFOV_12MM)]
dsts = [ctx.alloc(800 * 800 * 3) for _ in range(F * 6)]
def run(n):
    for _ in range(n):
        ctx.equirect_views_dev(frames, W, H, 3, views, dsts)
def timed(label):
    run(3); ctx.sync(0)
    t0 = time.time()
    while time.time() - t0 < 0.15: run(20)
    ctx.sync(0)
    ctx.event_record(0, 0); run(100); ctx.event_record(0, 1)
    ms = ctx.event_elapsed_ms(0, 0, 1) / 100
    print(f"{label}: {ms*1e3:.1f} us/launch = {ms*1e3/F:.2f} us/frame", flush=True)
ctx.set_option("srcmajor", 0); timed("gather kernel")
ctx.set_option("srcmajor", -1)
for bx, rows in [(768, 32), (768, 24), (768, 16), (768, 20), (640, 32), (640, 24), (896, 32), (896, 24), (1008, 32), (1008, 24), (1008, 16), (512, 32), (768, 40)]:
    ctx.set_option("srcmajor_bx", bx); ctx.set_option("srcmajor_rows", rows)
    timed(f"srcmajor bx {bx} rows {rows}")
