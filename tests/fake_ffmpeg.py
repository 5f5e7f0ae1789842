#!/usr/bin/env python3
"""Test double for the external `ffmpeg` BINARY (absent from the build and GPU images) -- decoder role only.

Understands exactly the command line gs360/video.py composes for its shared decoder:
    <prog> -hide_banner -loglevel error -nostdin [-copyts] [-ss S] -i clip.npy [-ss S] [-to T] [-vsync vfr]
           -vf <chain>,format=rgb24 -an -f image2pipe -c:v ppm pipe:1
`clip.npy` is a uint8 (or uint16) array [N, H, W, 3]; frame n has timestamp n seconds.  `fps=F` keeps frames whose timestamp is
a multiple of 1/F (F <= 1 in the tests), `select='eq(n\\,i)+...'` keeps the listed indices, -ss/-to bound the timestamps.
Anything that looks like the reference's per-view invocation (a v360 filter, an image file pattern as output) is
refused with exit code 3, so a test notices when a job was routed to the subprocess path by mistake.
"""
import re
import sys

import numpy as np


def main(argv):
    args = argv[1:]
    if not args or args[-1] != "pipe:1":
        sys.stderr.write("fake_ffmpeg: only the PPM pipe decoder role is emulated\n")
        return 3
    opts, src, i = {}, None, 0
    while i < len(args) - 1:
        tok = args[i]
        if tok in ("-hide_banner", "-nostdin", "-copyts", "-an", "-y"):
            i += 1
            continue
        val = args[i + 1]
        if tok == "-i":
            src = val
        else:
            opts[tok] = val
        i += 2
    chain = opts.get("-vf", "")
    deep = chain.endswith("format=rgb48be")
    if "v360=" in chain or opts.get("-f") != "image2pipe" or opts.get("-c:v") != "ppm" or not (chain.endswith("format=rgb24") or deep):
        sys.stderr.write("fake_ffmpeg: unexpected decoder command line\n")
        return 3
    if src is None or src.endswith("broken.npy"):
        sys.stderr.write("fake_ffmpeg: cannot open input\n")
        return 1
    clip = np.load(src)
    keep = list(range(clip.shape[0]))
    lo = float(opts["-ss"]) if "-ss" in opts else None
    hi = float(opts["-to"]) if "-to" in opts else None
    keep = [n for n in keep if (lo is None or n >= lo) and (hi is None or n < hi)]
    m = re.search(r"select='([^']*)'", chain)
    if m:
        want = {int(t) for t in re.findall(r"eq\(n\\,(\d+)\)", m.group(1))}
        keep = [n for n in keep if n in want]
    m = re.search(r"fps=([0-9.]+)", chain)
    if m:
        step = max(1, int(round(1.0 / float(m.group(1)))))
        keep = [n for k, n in enumerate(keep) if k % step == 0]
    out = sys.stdout.buffer
    for n in keep:
        fr = np.ascontiguousarray(clip[n])
        if deep:          # 16-bit PPM: maxval 65535, big-endian samples (an 8-bit clip is widened like ffmpeg's format filter)
            fr16 = fr.astype(np.uint16) * (257 if fr.dtype == np.uint8 else 1)
            out.write(b"P6\n%d %d\n65535\n" % (fr.shape[1], fr.shape[0]))
            out.write(fr16.astype(">u2").tobytes())
            continue
        out.write(b"P6\n%d %d\n255\n" % (fr.shape[1], fr.shape[0]))
        out.write(fr.tobytes())
    out.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
