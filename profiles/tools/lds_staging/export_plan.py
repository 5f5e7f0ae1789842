"""Export the exact per-(tile, pass) line lists a staged eq_views kernel would load for cfg2 (kernel's own tiling)."""
import sys, numpy as np
import pathlib
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from oracle import orc
from util import *
W,H=7680,3840; stride=W*3
def lines_of(sx, sy):
    ix = sx >> 5; iy = sy >> 5
    y0 = np.clip(iy,0,H-1); y1 = np.clip(iy+1,0,H-1)
    ixl = np.minimum(ix, W-5)
    out=[]
    rows = np.concatenate([y0.ravel(), y1.ravel()]).astype(np.int64)
    xb = np.concatenate([ixl.ravel(), ixl.ravel()]).astype(np.int64)*3
    g0 = (rows*stride + xb) >> 7; g1 = (rows*stride + xb + 5) >> 7
    order = np.argsort(rows, kind='stable')
    rows=rows[order]; g0=g0[order]; g1=g1[order]
    ur, start = np.unique(rows, return_index=True)
    lo = np.minimum.reduceat(g0, start); hi = np.maximum.reduceat(g1, start)
    # seam tiles: a row whose range is huge -> mark direct (empty list)
    if (hi-lo).max() > 64: return None
    return np.concatenate([np.arange(a,b+1) for a,b in zip(lo,hi)]).astype(np.uint32)
passes=[]
specs = ring_views(6, 800, HFOV_12MM)
for spec in specs:
    sx, sy = orc.equirect_map(orc.make_view(*spec), W, H)
    h,w = sx.shape; half=(w+1)//2; top=(h+1)//2
    for ty in range(0, top, 8):
        for tx in range(0, half, 64):
            for mirror in (0,1):
                cols = slice(tx, min(tx+64,half)) if not mirror else slice(w-min(tx+64,half), w-tx)
                rt = slice(ty, min(ty+8, top)); rb = slice(h-min(ty+8,top), h-ty)
                a = lines_of(sx[rt,cols], sy[rt,cols]); b = lines_of(sx[rb,cols], sy[rb,cols])
                passes.append(np.zeros(0,np.uint32) if a is None or b is None else np.concatenate([a,b]))
n = np.array([len(p) for p in passes], np.uint32)
off = np.concatenate([[0], np.cumsum(n)]).astype(np.uint32)
with open(sys.argv[1] if len(sys.argv) > 1 else 'cfg2_plan.bin','wb') as f:
    np.array([len(passes)], np.uint32).tofile(f); off.tofile(f); np.concatenate(passes).astype(np.uint32).tofile(f)
print(len(passes), 'passes', n.sum(), 'lines/frame', 'empty(seam)', (n==0).sum(), 'max', n.max())
allv = np.concatenate(passes); print('distinct lines/frame', len(np.unique(allv)))
