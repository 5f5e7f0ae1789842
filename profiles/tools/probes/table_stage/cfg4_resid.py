import numpy as np
d = np.load("/root/repo/scratch/r06/cfg4_tables.npz")
views = sorted({k.rsplit("_map_x", 1)[0] for k in d.files if k.endswith("_map_x")})
print(views)
for v in views:
    mx, my = d[v + "_map_x"], d[v + "_map_y"]
    sx = np.rint(mx * 32).astype(np.int64); sy = np.rint(my * 32).astype(np.int64)
    h, w = sx.shape
    worst = {}
    for name, s in (("x", sx), ("y", sy)):
        mlin = 0; mquad = 0; md2 = 0
        for x0 in range(0, w, 64):
            seg = s[:, x0:x0+64].astype(np.float64)
            n = seg.shape[1]
            if n < 3: continue
            i = np.arange(n)[None, :]
            a = seg[:, :1]; b = seg[:, -1:]
            lin = a + (b - a) * i / (n - 1)
            r = seg - lin
            mlin = max(mlin, np.abs(r).max())
            # quadratic through first, mid, last
            m = seg[:, n // 2:n // 2 + 1]; im = n // 2
            # Lagrange
            L0 = (i - im) * (i - (n - 1)) / ((0 - im) * (0 - (n - 1)))
            L1 = (i - 0) * (i - (n - 1)) / ((im - 0) * (im - (n - 1)))
            L2 = (i - 0) * (i - im) / (((n - 1) - 0) * ((n - 1) - im))
            q = a * L0 + m * L1 + b * L2
            mquad = max(mquad, np.abs(seg - q).max())
        d1 = np.diff(s, axis=1)
        worst[name] = (mlin, mquad, d1.min(), d1.max())
    print(v, {k: tuple(round(float(t), 1) for t in vv) for k, vv in worst.items()})
