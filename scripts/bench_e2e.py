#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the cfg2 workload: pinned host frame -> H2D -> 6-view launch -> D2H, N frames
in flight per GPU (gs360.stream.FramePipeline).  Informational -- bench.py's `value` is the device-resident rate.

    python scripts/bench_e2e.py --frames 60 --slots 3
"""
import argparse
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))

import numpy as np  # noqa: E402

import bench  # noqa: E402
import gs360  # noqa: E402
from gs360.stream import FramePipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--in-place", action="store_true", help="frames are produced directly in the pinned slot buffers "
                    "(what a decoder's readinto does): no staging copy, PCIe is the bound")
    args = ap.parse_args()
    ctx = gs360.Context(0, n_slots=args.slots)
    views = [gs360.View.make(*v) for v in bench.view_table()]
    pipe = FramePipeline(ctx, bench.W, bench.H, bench.C, views, n_slots=args.slots)
    src = [bench.synth_frame(np, k) for k in range(4)]
    for k in range(args.slots):                       # warm-up
        pipe.submit(src[k % 4])
    pipe.drain()
    t0 = time.perf_counter()
    n_out = 0
    for k in range(args.frames):
        if args.in_place:
            done, buf = pipe.acquire()
            buf[:4096] = src[k % 4].reshape(-1)[:4096]     # a token write; the slot keeps the frame staged during warm-up
            pipe.commit(tag=k)
        else:
            done = pipe.submit(src[k % 4], tag=k)
        if done:
            n_out += 1
    n_out += len(pipe.drain())
    dt = time.perf_counter() - t0
    px = args.frames * bench.N_VIEWS * bench.SIZE * bench.SIZE
    in_bytes = args.frames * bench.W * bench.H * bench.C
    print(json.dumps({"what": "cfg2 end-to-end, pinned host -> H2D -> kernel -> D2H", "frames": args.frames, "slots": args.slots,
                      "frames_per_s": round(args.frames / dt, 1), "MPix_per_s_out": round(px / dt / 1e6, 1),
                      "h2d_GB_per_s": round(in_bytes / dt / 1e9, 2), "ms_per_frame": round(dt / args.frames * 1e3, 3),
                      "note": ("frames produced in place in the pinned slot buffers (decoder readinto): H2D + kernel + D2H only"
                               if args.in_place else "includes the host memcpy into the pinned staging buffer (single Python thread)")}))
    pipe.close()
    ctx.close()


if __name__ == "__main__":
    main()
