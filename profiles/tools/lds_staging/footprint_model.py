import sys, numpy as np
import pathlib
ROOT = pathlib.Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
from oracle import orc
from util import *
LINE=128
def stage_lines(sx, sy, W, H, stride):
    ix = sx >> 5; iy = sy >> 5
    y0 = np.clip(iy,0,H-1); y1 = np.clip(iy+1,0,H-1)
    ixl = np.minimum(ix, W-5)
    tot=0; rows=set()
    lo = {}; hi = {}
    for y in (y0,y1):
        g0 = (y.astype(np.int64)*stride + ixl*3) >> 7
        g1 = (y.astype(np.int64)*stride + ixl*3 + 5) >> 7
        for yy,a,b in zip(y.ravel(), g0.ravel(), g1.ravel()):
            if yy in lo:
                if a<lo[yy]: lo[yy]=a
                if b>hi[yy]: hi[yy]=b
            else: lo[yy]=a; hi[yy]=b
    n = sum(hi[r]-lo[r]+1 for r in lo)
    return n, len(lo), (max(lo)-min(lo)+1)
def run(name, W, H, spec, tw=64, th=16):
    v = orc.make_view(*spec)
    sx, sy = orc.equirect_map(v, W, H)
    h, w = sx.shape
    stride = W*3
    level = spec[1]==0
    half = (w+1)//2
    res=[]
    if level:
        top=(h+1)//2
        for ty in range(0, top, th//2):
            for tx in range(0, half, tw):
                for mirror in (0,1):
                    cols = slice(tx, min(tx+tw,half)) if not mirror else slice(w-min(tx+tw,half), w-tx)
                    rt = slice(ty, min(ty+th//2, top))
                    rb = slice(h-min(ty+th//2,top), h-ty)
                    a = stage_lines(sx[rt,cols], sy[rt,cols], W,H,stride)
                    b = stage_lines(sx[rb,cols], sy[rb,cols], W,H,stride)
                    res.append((a[0]+b[0], a[1]+b[1], a[0], a[2]))
    else:
        for ty in range(0,h,th):
            for tx in range(0,half,tw):
                for mirror in (0,1):
                    cols = slice(tx, min(tx+tw,half)) if not mirror else slice(w-min(tx+tw,half), w-tx)
                    a = stage_lines(sx[ty:ty+th,cols], sy[ty:ty+th,cols], W,H,stride)
                    res.append((a[0], a[1], a[0], a[2]))
    r = np.array(res)
    # seam tiles have giant ranges: report separately
    big = r[:,0] > 1024
    rr = r[~big]
    print(f"{name}: passes {len(r)}, seam/huge {big.sum()}, lines/pass mean {rr[:,0].mean():.0f} p50 {np.median(rr[:,0]):.0f} p90 {np.percentile(rr[:,0],90):.0f} max {rr[:,0].max()};"
          f" rows mean {rr[:,1].mean():.0f} max {rr[:,1].max()}; rowspan max {rr[:,3].max()}; frac>384: {(rr[:,0]>384).mean():.2f} frac>320 {(rr[:,0]>320).mean():.2f} frac>256 {(rr[:,0]>256).mean():.2f}; lines/px {rr[:,0].sum()/ (w*h):.3f}")
run("cfg2 level 800", 7680,3840,(0,0,HFOV_12MM,HFOV_12MM,800,800))
run("cfg2 yaw60", 7680,3840,(60,0,HFOV_12MM,HFOV_12MM,800,800))
run("cfg1 level 1600 on 5760", 5760,2880,(45,0,HFOV_12MM,HFOV_12MM,1600,1600))
run("cfg3 pitch30 1600", 7680,3840,(45,30,HFOV_14MM,HFOV_14MM,1600,1600))
run("cfg3 level 1600", 7680,3840,(90,0,HFOV_14MM,HFOV_14MM,1600,1600))
run("cfg5 level 2048", 7680,3840,(36,0,HFOV_17MM,HFOV_17MM,2048,2048))
run("cfg5 pitch30 2048", 7680,3840,(0,30,HFOV_17MM,HFOV_17MM,2048,2048))
