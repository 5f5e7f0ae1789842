#!/usr/bin/env python3
"""End-to-end rate of the DROP-IN CLI itself (round-1 VERDICT item 5): gs360_360PerspCut.main() on

  stills  a cfg1-style folder: N synthetic 5760x2880 PNG panoramas -> `--preset default` (8 x 1600^2 views each)
  video   the cfg3 path: an 8K clip through the shared-decode video session (tests/fake_ffmpeg.py stands in for the
          absent ffmpeg binary as the PPM-pipe decoder) -> `--preset full360coverage` (12 x 1600^2 per frame)

and reports frames/s, views/s, how the wall time splits into image decode / GPU / image encode (thread-summed), the
number of batched launches the engine issued (gs360/engine.py coalesces the view jobs of a frame) and "gpu_busy_share" =
thread-summed wall time spent inside launch + copy + sync sections (waiting for a free stream slot included) / wall time
-- an UPPER bound of the share of time the GPU is busy (with many workers the sections of different frames overlap).  Informational: the image codecs run on
the host and bound both figures; bench.py's `value` is the device-resident hot path.

    python scripts/bench_cli_e2e.py [--frames 6] [--video-frames 8] [--jobs 16] [--ext jpg] [--only stills|video]
"""
import argparse
import io
import json
import os
import pathlib
import stat
import sys
import tempfile
import threading
import time
from contextlib import redirect_stdout

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "360cam-pgm-3dgs-tools_amd"
for p in (str(ROOT), str(PKG), str(PKG / "cli_tools")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

import gs360_360PerspCut as cut  # noqa: E402
from gs360 import engine, imageio  # noqa: E402


def synth(h, w, k):
    x = np.arange(w, dtype=np.uint32)[None, :]
    y = np.arange(h, dtype=np.uint32)[:, None]
    n = (((x * np.uint32(2654435761)) ^ (y * np.uint32(40503 + 977 * k))) >> np.uint32(29)).astype(np.uint8)
    img = np.empty((h, w, 3), np.uint8)
    img[..., 0] = ((x * 255) // w).astype(np.uint8) + n
    img[..., 1] = ((y * 255) // h).astype(np.uint8) + n
    img[..., 2] = ((((x >> 6) + (y >> 6)) & 1) * 96).astype(np.uint8) + n
    return img


class Timers:
    """thread-summed seconds spent inside the host codecs"""

    def __init__(self):
        self.t = {"decode_s": 0.0, "encode_s": 0.0}
        self.lock = threading.Lock()
        self._read, self._write = imageio.read_image, imageio.write_image

    def __enter__(self):
        def read(path):
            t0 = time.perf_counter()
            try:
                return self._read(path)
            finally:
                with self.lock:
                    self.t["decode_s"] += time.perf_counter() - t0

        def write(path, arr, jpeg_q=None):
            t0 = time.perf_counter()
            try:
                return self._write(path, arr, jpeg_q=jpeg_q)
            finally:
                with self.lock:
                    self.t["encode_s"] += time.perf_counter() - t0
        imageio.read_image, imageio.write_image = read, write
        return self

    def __exit__(self, *exc):
        imageio.read_image, imageio.write_image = self._read, self._write


def run_cli(argv):
    old = sys.argv
    sys.argv = ["gs360_360PerspCut.py"] + argv
    buf = io.StringIO()
    eng = engine.get_engine()
    before = eng.stats()
    t0 = time.perf_counter()
    try:
        with Timers() as tm, redirect_stdout(buf):
            try:
                cut.main()
            except SystemExit as exc:
                if exc.code not in (0, None):
                    raise RuntimeError(f"CLI exited with {exc.code}: {buf.getvalue()[-400:]}")
    finally:
        sys.argv = old
    wall = time.perf_counter() - t0
    after = eng.stats()
    st = {k: after.get(k, 0) - before.get(k, 0) for k in ("launches", "views", "gpu_s")}
    tail = [ln for ln in buf.getvalue().splitlines() if ln.startswith("[OK]")]
    return wall, st, tm.t, (tail[-1] if tail else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--video-frames", type=int, default=8)
    ap.add_argument("--jobs", type=int, default=16)
    ap.add_argument("--ext", default="jpg")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory(prefix="gs360_e2e_") as td:
        td = pathlib.Path(td)
        if args.only in ("", "stills"):
            d = td / "stills"
            d.mkdir()
            for k in range(args.frames):
                imageio.write_image(d / f"pano_{k:03d}.png", synth(2880, 5760, k))
            wall, st, tm, ok = run_cli(["-i", str(d), "--preset", "default", "--ext", args.ext, "-j", str(args.jobs)])
            nv = args.frames * 8
            rows.append({"what": f"CLI stills: {args.frames} x 5760x2880 PNG -> default preset 8 x 1600^2 .{args.ext}, -j {args.jobs}",
                         "wall_s": round(wall, 3), "frames_per_s": round(args.frames / wall, 2), "views_per_s": round(nv / wall, 1),
                         "out_MPix_per_s": round(nv * 2.56 / wall, 1), "batched_launches": st["launches"], "views_rendered": st["views"],
                         "views_per_launch": round(st["views"] / max(1, st["launches"]), 2),
                         "gpu_section_s": round(st["gpu_s"], 4), "gpu_busy_share": round(st["gpu_s"] / wall, 4),
                         "decode_s_thread_sum": round(tm["decode_s"], 2), "encode_s_thread_sum": round(tm["encode_s"], 2), "cli": ok})
        if args.only in ("", "video"):
            clip = np.stack([synth(3840, 7680, 100 + k) for k in range(args.video_frames)])
            np.save(td / "clip.npy", clip)
            del clip
            prog = td / "ffmpeg_double"
            prog.write_text("#!/bin/sh\nexec {} {} \"$@\"\n".format(sys.executable, ROOT / "tests" / "fake_ffmpeg.py"))
            prog.chmod(prog.stat().st_mode | stat.S_IXUSR)
            os.environ["GS360_INTERP"] = "linear"      # BASELINE configs[2] is the bilinear workload
            wall, st, tm, ok = run_cli(["-i", str(td / "clip.npy"), "--ffmpeg", str(prog), "-f", "1", "--preset", "full360coverage",
                                        "--ext", args.ext, "-o", str(td / "vout"), "-j", str(args.jobs)])
            os.environ.pop("GS360_INTERP")
            nv = args.video_frames * 12
            rows.append({"what": f"CLI video: {args.video_frames} x 7680x3840 frames (PPM pipe from the decoder double) -> full360coverage 12 x 1600^2 "
                                 f".{args.ext}, -j {args.jobs}", "wall_s": round(wall, 3), "frames_per_s": round(args.video_frames / wall, 2),
                         "views_per_s": round(nv / wall, 1), "out_MPix_per_s": round(nv * 2.56 / wall, 1), "batched_launches": st["launches"],
                         "views_rendered": st["views"], "views_per_launch": round(st["views"] / max(1, st["launches"]), 2),
                         "gpu_section_s": round(st["gpu_s"], 4), "gpu_busy_share": round(st["gpu_s"] / wall, 4),
                         "encode_s_thread_sum": round(tm["encode_s"], 2), "cli": ok})
    for r in rows:
        print(json.dumps(r))
    engine.shutdown()


if __name__ == "__main__":
    main()
