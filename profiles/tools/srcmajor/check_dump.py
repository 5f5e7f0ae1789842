"""Compare the harness' dumped views (frames 0 and 15) with the CPU oracle on the same synthetic frames."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import orc

W, H, N, S, HF = 7680, 3840, 6, 800, 112.61986494804043


def frame(f):
    p = np.arange(W * H * 3, dtype=np.uint32)
    x = p * np.uint32(2654435761) + np.uint32(f * 40503)
    x ^= x >> np.uint32(15); x *= np.uint32(2246822519); x ^= x >> np.uint32(13)
    return ((x >> np.uint32(8)) & np.uint32(255)).astype(np.uint8).reshape(H, W, 3)


def main():
    dump = np.fromfile(sys.argv[1], np.uint8).reshape(2, N, S, S, 3)
    views = [orc.make_view(((i * 360.0 / N + 180) % 360) - 180, 0.0, HF, HF, S, S) for i in range(N)]
    bad = 0
    for n, f in enumerate((0, 15)):
        want = orc.equirect_views_u8(frame(f), views, threads=16)
        for v in range(N):
            d = np.abs(dump[n, v].astype(int) - want[v].astype(int))
            nb = int((d != 0).sum())
            bad += nb
            print(f"frame {f} view {v}: mismatching bytes {nb} max diff {int(d.max())}" + ("" if nb == 0 else f" first at {np.argwhere(d != 0)[0].tolist()}"))
    print("PARITY", "OK" if bad == 0 else "FAIL")


main()
