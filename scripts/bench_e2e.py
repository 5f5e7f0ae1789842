#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the cfg2 workload: pinned host frame -> H2D -> 6-view launch -> D2H, N frames
in flight per GPU (gs360.stream.FramePipeline).  Informational -- bench.py's `value` is the device-resident rate.

    python scripts/bench_e2e.py --frames 60 --slots 3
    python scripts/bench_e2e.py --in-place --frames 600 --devices 0,1,2,3,4,5,6,7    # ONE process, one feeder thread + context per GPU

`--devices` is the in-process fan-out north_star describes ("pinned-host decoded frames fanned out on per-GPU HIP streams"):
the frames of the job are dealt round-robin to the listed devices (gs360.sharding.frames_for_rank), every device has its own
context, upload/download streams and pinned slot buffers, and one Python thread per device drives it (the ctypes calls
release the GIL).  A device may be listed twice (two contexts on one GPU) to exercise the path on a 1-GPU box.
"""
import argparse
import json
import pathlib
import sys
import threading
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))

import numpy as np  # noqa: E402

import bench  # noqa: E402
import gs360  # noqa: E402
from gs360.stream import FramePipeline  # noqa: E402


def fan_out(args):
    from gs360.sharding import frames_for_rank
    devs = [int(t) for t in args.devices.split(",") if t.strip() != ""]
    views = [gs360.View.make(*v) for v in bench.view_table()]
    src = [bench.synth_frame(np, k) for k in range(4)]
    ctxs = [gs360.Context(d, n_slots=2) for d in devs]
    pipes = [FramePipeline(c, bench.W, bench.H, bench.C, views, n_slots=args.slots, copy_out=False) for c in ctxs]
    for p in pipes:                                        # stage a frame in every pinned slot
        for k in range(args.slots):
            _d, buf = p.acquire()
            buf[:] = src[k % 4].reshape(-1)
            p.commit(tag=-1)
        p.drain()
    done = [0] * len(devs)
    per_dev_s = [0.0] * len(devs)
    start = threading.Barrier(len(devs) + 1)

    def feeder(r):
        mine = frames_for_rank(args.frames, len(devs), r)
        start.wait()
        t0 = time.perf_counter()
        n = 0
        for k in mine:
            d, _buf = pipes[r].acquire()
            pipes[r].commit(tag=k)
            n += 1 if d else 0
        n += len(pipes[r].drain())
        per_dev_s[r] = time.perf_counter() - t0
        done[r] = n
    threads = [threading.Thread(target=feeder, args=(r,)) for r in range(len(devs))]
    for t in threads:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    assert sum(done) == args.frames, (done, args.frames)
    print(json.dumps({"what": "cfg2 end-to-end, ONE process fanning frames out to per-device contexts (pinned host -> H2D -> kernel -> D2H)",
                      "devices": devs, "frames": args.frames, "slots": args.slots, "frames_per_s": round(args.frames / dt, 1),
                      "MPix_per_s_out": round(args.frames * bench.N_VIEWS * bench.SIZE * bench.SIZE / dt / 1e6, 1),
                      "h2d_GB_per_s_total": round(args.frames * bench.W * bench.H * bench.C / dt / 1e9, 2),
                      "per_device_frames": done, "per_device_seconds": [round(x, 4) for x in per_dev_s]}))
    for p in pipes:
        p.close()
    for c in ctxs:
        c.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--slots", type=int, default=3)
    ap.add_argument("--in-place", action="store_true", help="frames are produced directly in the pinned slot buffers "
                    "(what a decoder's readinto does): no staging copy, PCIe is the bound")
    ap.add_argument("--devices", default="", help="comma list of device ordinals: frames are dealt to one feeder thread + context per entry")
    args = ap.parse_args()
    if args.devices:
        return fan_out(args)
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
