#!/usr/bin/env python3
"""End-to-end rate of the dual-fisheye DROP-IN CLI: gs360_DualFisheyeDistortionCalibration.main() on N synthetic pairs of 4000 x 4000
fisheye JPEGs with the template calibration (no -x) -> the tool's defaults: SFM10 (10 x 1750^2 views per pair), bicubic, JPEG output.
Reports pairs/s and views/s of the whole command (map building, image decode, GPU, image encode, file writes).  Informational: the
image codecs run on the host; bench.py / tests/tools/bench_configs.py measure the device-resident remap (cfg4: ~0.11 ms per 6 views).

    python scripts/bench_df_cli_e2e.py [--pairs 8] [--workers 16] [--interpolation cubic] [--ext jpg]
"""
import argparse
import io
import json
import pathlib
import sys
import tempfile
import time
from contextlib import redirect_stdout

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "360cam-pgm-3dgs-tools_amd"
for p in (str(ROOT), str(PKG), str(PKG / "cli_tools")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

import gs360_DualFisheyeDistortionCalibration as df  # noqa: E402
from gs360 import imageio  # noqa: E402


def synth(h, w, k):
    x = np.arange(w, dtype=np.uint32)[None, :]
    y = np.arange(h, dtype=np.uint32)[:, None]
    n = (((x * np.uint32(2654435761)) ^ (y * np.uint32(40503 + 977 * k))) >> np.uint32(29)).astype(np.uint8)
    img = np.empty((h, w, 3), np.uint8)
    img[..., 0] = ((x * 255) // w).astype(np.uint8) + n
    img[..., 1] = ((y * 255) // h).astype(np.uint8) + n
    img[..., 2] = ((((x >> 6) + (y >> 6)) & 1) * 96).astype(np.uint8) + n
    return img


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--interpolation", default="cubic")
    ap.add_argument("--ext", default="jpg")
    ap.add_argument("--size", type=int, default=4000)
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        d = pathlib.Path(tmp) / "shots"
        d.mkdir()
        for k in range(args.pairs):
            for j, lens in enumerate("XY"):
                imageio.write_image(d / f"frame_{k:04d}_{lens}.jpg", synth(args.size, args.size, 2 * k + j))
        argv = ["prog", "-i", str(d), "--interpolation", args.interpolation, "--perspective-ext", args.ext, "--workers", str(args.workers)]
        out = io.StringIO()
        old = sys.argv
        sys.argv = argv
        t0 = time.perf_counter()
        try:
            with redirect_stdout(out):
                try:
                    df.main()
                except SystemExit as e:
                    if e.code not in (0, None):
                        raise
        finally:
            sys.argv = old
        dt = time.perf_counter() - t0
        lines = out.getvalue().splitlines()
        done = [l for l in lines if l.startswith("[DONE]")]
        n_out = len(list((d.resolve().with_name("shots_perspective_colmap") / "Images").glob("*")))
    print(json.dumps({"pairs": args.pairs, "workers": args.workers, "interpolation": args.interpolation, "ext": args.ext,
                      "seconds": round(dt, 2), "pairs_per_s": round(args.pairs / dt, 2), "views_per_s": round(n_out / dt, 1),
                      "views_written": n_out, "done_line": done[-1] if done else None}))


if __name__ == "__main__":
    main()
