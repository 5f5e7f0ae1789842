#!/usr/bin/env python3
"""BASELINE config 3 at full scale on ONE GPU: 600 decoded 8K frames resident in HBM (53 GB of the 288 GB), every frame
cut into the 12 `full360coverage` views (1600^2).  This is the state gs360/video.py leaves the device in after the
shared decode of a video; here the frames are synthetic (image B rolled 13 px per frame) and uploaded
once.  Timed: the 600 x 12 view renders (batched 16 frames per launch, HIP events on the launch stream); a few frames
are checked against the oracle.  Informational -- bench.py (cfg2) is the headline.

    python tests/tools/bench_cfg3_resident.py [--frames 600]
"""
import argparse
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

import bench  # noqa: E402
import gs360  # noqa: E402
from oracle import orc  # noqa: E402  (checker)
from util import HFOV_14MM, PRESET_FULL360  # noqa: E402

W, H, C, SIZE = 7680, 3840, 3, 1600
BATCH = 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600)
    args = ap.parse_args()
    ctx = gs360.Context(0, n_slots=2)
    info = ctx.info()
    specs = [(y, p, HFOV_14MM, HFOV_14MM, SIZE, SIZE) for y, p in PRESET_FULL360]
    views = [gs360.View.make(*s) for s in specs]
    t0 = time.perf_counter()
    base = bench.synth_frame(np, 0)
    d_frames = []
    for k in range(args.frames):
        d_frames.append(ctx.to_device(np.roll(base, 13 * k, axis=1)))
    t_up = time.perf_counter() - t0
    d_out = [ctx.alloc(SIZE * SIZE * C) for _ in range(BATCH * len(views))]
    calls = []
    for b0 in range(0, args.frames, BATCH):
        fr = d_frames[b0:b0 + BATCH]
        calls.append(ctx.make_equirect_call(fr, W, H, C, views, d_out[:len(fr) * len(views)], slot=0))
    for c in calls[:2]:
        c()
    ctx.sync(-1)
    ctx.event_record(0, 0)
    t0 = time.perf_counter()
    for c in calls:
        c()
    ctx.event_record(0, 1)
    ctx.sync(-1)
    wall = time.perf_counter() - t0
    ms = ctx.event_elapsed_ms(0, 0, 1)
    # parity: the last batch is still in d_out -> check two of its frames, two views each
    last0 = (len(calls) - 1) * BATCH
    ok = True
    for j in (0, min(BATCH, args.frames - last0) - 1):
        k = last0 + j
        frame = np.roll(base, 13 * k, axis=1)
        for v in (1, 6):
            got = ctx.download(d_out[j * len(views) + v], (SIZE, SIZE, C))
            want = orc.equirect_views_u8(frame, [orc.make_view(*specs[v])], threads=0)[0]
            ok = ok and bool(np.array_equal(got, want))
    px = args.frames * len(views) * SIZE * SIZE
    print(json.dumps({"what": "cfg3 full scale on one GPU: HBM-resident 8K frames -> full360coverage 12x1600^2",
                      "frames": args.frames, "resident_GB": round(args.frames * W * H * C / 1e9, 1), "hbm_GB": round(info["hbm_bytes"] / 1e9),
                      "upload_s": round(t_up, 1), "render_ms_total": round(ms, 2), "us_per_frame": round(ms / args.frames * 1e3, 1),
                      "frames_per_s_kernel_only": round(args.frames / (ms * 1e-3)), "GPix_per_s": round(px / (ms * 1e-3) / 1e9, 1),
                      "wall_s": round(wall, 3), "parity_vs_oracle": ok}))
    ctx.close()


if __name__ == "__main__":
    main()
