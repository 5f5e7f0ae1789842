#!/usr/bin/env python3
"""Randomised parity campaign: HIP kernels (through the C ABI) against the CPU oracle on random shapes, strides,
channel counts, views, maps and interpolation modes.  Bit-exact uint8 is the bar; the first mismatch is printed with
the seed that reproduces it and the exit code is 1.

    python tests/tools/fuzz_parity.py --seconds 120 [--seed 1]
"""
import argparse
import ctypes as C
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))

import numpy as np  # noqa: E402

import gs360  # noqa: E402
from oracle import orc  # noqa: E402  (checker)


def padded(rng, img, pad):
    """copy of img with `pad` junk bytes after every row; returns (buffer, stride)"""
    h, w, c = img.shape
    stride = w * c + pad
    buf = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    buf[:, :w * c] = img.reshape(h, w * c)
    return buf, stride


def fuzz_equirect(ctx, rng, case):
    c = int(rng.choice([1, 3, 3, 3, 4]))
    W = int(rng.integers(8, 700))
    H = int(rng.integers(2, 360))
    src = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
    nv = int(rng.integers(1, 5))
    specs = []
    for _ in range(nv):
        level = rng.random() < 0.4
        specs.append((float(rng.uniform(-400, 400)), 0.0 if level else float(rng.uniform(-95, 95)),
                      float(rng.uniform(5, 179)), float(rng.uniform(5, 179)), int(rng.integers(1, 200)), int(rng.integers(1, 120))))
    if rng.random() < 0.5:
        # yaw rings: the presets' shape (PC:794) -- one pitch magnitude / fov / size, yaw = i * 360 / count so that members differ by
        # whole texels when count divides W, members with the pitch sign flipped, a duplicate, and (half the time) a member
        # whose yaw breaks the common sub-texel phase; up to 20 views so that rings are split across launches
        if rng.random() < 0.5:
            W = int(rng.choice([8, 12, 24, 48, 96, 240, 360, 480, 600]))
            src = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
        count = int(rng.choice([2, 3, 4, 6, 8, 12]))
        base = specs[0]
        pitch = 0.0 if rng.random() < 0.35 else float(rng.choice([30.0, -30.0, 90.0, float(rng.uniform(-95, 95))]))
        off = float(rng.choice([0.0, 0.0, 45.0, float(rng.uniform(-180, 180))]))
        nv = int(rng.integers(2, 21))
        specs = []
        for i in range(nv):
            yaw = off + (i % count) * 360.0 / count
            if rng.random() < 0.15:
                yaw += float(rng.uniform(-3, 3))
            sign = -1.0 if rng.random() < 0.4 else 1.0
            specs.append((yaw, sign * pitch, base[2], base[3], base[4], base[5]))
        if rng.random() < 0.3:
            specs[int(rng.integers(0, nv))] = (float(rng.uniform(-400, 400)), float(rng.uniform(-95, 95)), float(rng.uniform(5, 179)),
                                               float(rng.uniform(5, 179)), int(rng.integers(1, 200)), int(rng.integers(1, 120)))
    interp = int(rng.choice([1, 1, 2]))
    spad = int(rng.choice([0, 0, 1, 3, 4, 64]))
    dpad = int(rng.choice([0, 0, 1, 2, 4]))
    use_mask = rng.random() < 0.3
    fish = (not use_mask) and rng.random() < 0.3          # equidistant-fisheye outputs (GS360_EQ_FISHEYE_OUT)
    if fish:
        specs = [(s[0], s[1], float(rng.uniform(20, 300)), float(rng.uniform(20, 300)), s[4], s[5]) for s in specs]
    sbuf, sstride = padded(rng, src, spad)
    d_src = ctx.to_device(sbuf)
    views = [gs360.View.make(*s) for s in specs]
    # one common dst stride for all views of the call (ABI): widest view row + pad
    dstride = max(s[4] for s in specs) * c + dpad if dpad else 0
    d_out = [ctx.alloc((dstride or s[4] * c) * s[5] + 64) for s in specs]
    for b in d_out:
        ctx.memset(b, 0xAB)
    mask = d_mask = None
    if use_mask:
        mask = (rng.integers(0, 2, (H, W), dtype=np.uint8) * 255)
        d_mask = ctx.to_device(mask)
    ctx.equirect_views_dev([d_src], W, H, c, views, d_out, src_stride=sstride if spad else 0, dst_stride=dstride, interp=interp,
                           masks=[d_mask] if use_mask else None, flags=gs360.EQ_FISHEYE_OUT if fish else 0)
    if fish:
        want = orc.equirect_fisheye_views_u8(src, [orc.make_view(*s) for s in specs], interp=interp, threads=0)
    else:
        want = orc.equirect_views_u8(src, [orc.make_view(*s) for s in specs], interp=interp, mask=mask, threads=0)
    ok = True
    for k, s in enumerate(specs):
        row = dstride or s[4] * c
        raw = ctx.download(d_out[k], (s[5], row))
        got = raw[:, :s[4] * c].reshape(s[5], s[4], c)
        if not np.array_equal(got, want[k]):
            ok = False
            bad = np.argwhere(got != want[k])
            print(f"[equirect] case {case}: view {k} {s} C={c} src {W}x{H} interp={interp} spad={spad} dpad={dpad} mask={use_mask} fish={fish}: "
                  f"{len(bad)} bytes differ, first at {bad[0].tolist()}")
        if dstride and not np.all(raw[:, s[4] * c:] == 0xAB):
            ok = False
            print(f"[equirect] case {case}: view {k} wrote into the row padding")
    for b in d_out + [d_src] + ([d_mask] if d_mask else []):
        ctx.free(b)
    return ok


def fuzz_srcmajor(ctx, rng, case):
    """Yaw rings that fill their circle -- the shapes the source-major kernel takes (count | W, (3 W / count) % 16 == 0, view width
    % 4 == 0): one level ring, or a FAMILY of rings of one size (a level ring and / or pitched rings that come with their mirror ring at
    minus the pitch, each pair on its own yaw phase; sometimes a ring left WITHOUT its mirror: the call must then fall back).  Random
    counts, panorama sizes, fields of view (strong and weak minification), pitches up to views that touch the poles, view sizes incl.
    odd heights, yaw offsets that are NOT whole texels, view orders, several frames, padded destination rows.  The option "srcmajor"
    decides whether the kernel is actually taken (1: whenever the geometry fits); the result must be the oracle's either way."""
    family = rng.random() < 0.5
    count = int(rng.choice([2, 3, 4, 5] if family else [2, 3, 4, 5, 6, 8, 12, 16]))
    unit = 16 * count                                    # W multiple of 16 * count keeps both divisibility rules
    W = unit * int(rng.integers(1, max(2, 2200 // unit)))
    H = int(rng.choice([W // 2, int(rng.integers(max(2, W // 4), W))]))
    w = 4 * int(rng.integers(2, 60))
    h = int(rng.integers(2, 200))
    hf = float(rng.uniform(20, 170)); vf = float(rng.choice([hf, float(rng.uniform(20, 170))]))
    off = float(rng.choice([0.0, 0.0, 360.0 / W * int(rng.integers(0, W)), float(rng.uniform(-180, 180))]))
    specs = [(off + q * 360.0 / count, 0.0, hf, vf, w, h) for q in range(count)] if (not family or rng.random() < 0.6) else []
    if family:
        pairs = int(rng.integers(1, max(2, (16 - len(specs)) // (2 * count) + 1)))
        for _ in range(pairs):
            if len(specs) + 2 * count > 16:
                break
            pitch = float(rng.choice([30.0, float(rng.uniform(1, 85))]))
            ph = float(rng.choice([0.0, 180.0 / count, float(rng.uniform(0, 360))]))
            signs = (1, -1) if rng.random() < 0.9 else (1,)       # (now and then a ring without its mirror)
            specs += [(off + ph + q * 360.0 / count, sg * pitch, hf, vf, w, h) for q in range(count) for sg in signs]
    specs = [specs[int(q)] for q in rng.permutation(len(specs))]
    count = len(specs)
    nf = int(rng.choice([1, 1, 2, 3]))
    frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(nf)]
    d_src = [ctx.to_device(f) for f in frames]
    # a third of the cases with fused keep-masks (per-texel noise or blobs; source-major only when W and the ring period are whole keep dwords)
    masks = d_msk = None
    if rng.random() < 0.33:
        masks = [np.where(rng.random((H, W)) < rng.choice([0.5, 0.9]), 255, int(rng.integers(0, 128))).astype(np.uint8) for _ in range(nf)]
        d_msk = [ctx.to_device(m) for m in masks]
    views = [gs360.View.make(*s) for s in specs]
    dpad = int(rng.choice([0, 0, 4, 8, 3]))
    dstride = w * 3 + dpad if dpad else 0
    d_out = [ctx.alloc((dstride or w * 3) * h + 64) for _ in range(nf * count)]
    for b in d_out:
        ctx.memset(b, 0xAB)
    ctx.equirect_views_dev(d_src, W, H, 3, views, d_out, dst_stride=dstride, masks=d_msk)
    ok = True
    for f in range(nf):
        want = orc.equirect_views_u8(frames[f], [orc.make_view(*s) for s in specs], threads=0, mask=masks[f] if masks else None)
        for k, s in enumerate(specs):
            raw = ctx.download(d_out[f * count + k], (h, dstride or w * 3))
            got = raw[:, :w * 3].reshape(h, w, 3)
            if not np.array_equal(got, want[k]):
                ok = False
                bad = np.argwhere(got != want[k])
                print(f"[srcmajor] case {case}: frame {f} view {k} {s} src {W}x{H} count={count} dpad={dpad}: {len(bad)} bytes differ, first at {bad[0].tolist()}")
            if dstride and not np.all(raw[:, w * 3:] == 0xAB):
                ok = False
                print(f"[srcmajor] case {case}: view {k} wrote into the row padding")
    for b in d_out + d_src + (d_msk or []):
        ctx.free(b)
    return ok


def fuzz_table(ctx, rng, case):
    c = int(rng.choice([1, 3, 3, 4]))
    W = int(rng.integers(1, 300))
    H = int(rng.integers(1, 200))
    w = int(rng.integers(1, 260))
    h = int(rng.integers(1, 100))
    src = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
    kind = rng.integers(0, 4)
    if kind == 0:      # smooth affine-ish map
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        a = rng.uniform(-1.5, 1.5, 6).astype(np.float32)
        mx = a[0] * xx + a[1] * yy + np.float32(rng.uniform(-20, W))
        my = a[2] * xx + a[3] * yy + np.float32(rng.uniform(-20, H))
    elif kind == 1:    # uniform random incl. outside
        mx = rng.uniform(-8, W + 8, (h, w)).astype(np.float32)
        my = rng.uniform(-8, H + 8, (h, w)).astype(np.float32)
    elif kind == 2:    # exact 1/32 grid points and half-way ties
        mx = (rng.integers(-64, 32 * W + 64, (h, w)) / 32.0 + rng.choice([0.0, 1 / 64.0], (h, w))).astype(np.float32)
        my = (rng.integers(-64, 32 * H + 64, (h, w)) / 32.0 + rng.choice([0.0, 1 / 64.0], (h, w))).astype(np.float32)
    else:              # specials
        mx = rng.uniform(-2, W + 2, (h, w)).astype(np.float32)
        my = rng.uniform(-2, H + 2, (h, w)).astype(np.float32)
        sel = rng.random((h, w))
        mx[sel < 0.05] = np.nan
        my[(sel > 0.05) & (sel < 0.1)] = np.inf
        mx[(sel > 0.1) & (sel < 0.15)] = -np.inf
        mx[(sel > 0.15) & (sel < 0.2)] = 1e9
        my[(sel > 0.2) & (sel < 0.25)] = -40000.0
    interp = int(rng.choice([0, 1, 1, 2, 4]))
    bv = tuple(float(v) for v in rng.integers(0, 256, 4))
    valid = (rng.random((h, w)) > 0.2) if rng.random() < 0.5 else None
    fill = int(rng.integers(0, 256))
    planned = rng.random() < 0.5       # half of the cases through a map plan (the tables packed once; gs360_remap_plans_u8)
    if planned:
        got = _remap_planned(ctx, src, mx, my, valid, interp, bv, fill, np.uint8)
    else:
        got = ctx.remap(src, mx, my, interpolation=interp, border_value=bv, valid=valid, fill_value=fill)
    want = orc.remap_u8(src, mx, my, interp=interp, border_value=bv, threads=0)
    if valid is not None:
        want = orc.valid_fill(want.copy(), valid, fill)
    if not np.array_equal(got.reshape(want.shape), want):
        bad = np.argwhere(got.reshape(want.shape) != want)
        print(f"[table] case {case}: src {W}x{H}x{c} map {w}x{h} kind={kind} interp={interp} planned={planned}: {len(bad)} bytes differ, "
              f"first at {bad[0].tolist()}")
        return False
    return True


def fuzz_tablestage(ctx, rng, case):
    """the LDS-staged table kernel (bilinear RGB through map plans, csrc/gs360_tablestage.hip): smooth maps of random scale / rotation /
    radial term that cross the source's borders, valid fills, 1-4 jobs per call on one source, tight outputs (any width with h w % 4 == 0)
    and padded ones (rows of whole dwords), tiles of 8 / 16 / 32 rows; forced onto every job (option table_stage = 1)"""
    W = 4 * int(rng.integers(2, 160))                    # rows of whole dwords: what the kernel's box loads need
    H = int(rng.integers(2, 400))
    src = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    d_src = ctx.to_device(src)
    rows = int(rng.choice([8, 16, 32]))
    bv = tuple(float(v) for v in rng.integers(0, 256, 4))
    n_jobs = int(rng.integers(1, 5))
    jobs, wants, plans, bufs, strides = [], [], [], [], []
    for _ in range(n_jobs):
        pad = int(rng.choice([0, 0, 4, 16]))
        w = int(rng.integers(1, 70)) * 4 if pad else int(rng.integers(1, 280))
        h = int(rng.integers(1, 120))
        if not pad and (h * w) % 4:
            h = 4 * ((h + 3) // 4)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        u, v = xx / max(w - 1, 1) - 0.5, yy / max(h - 1, 1) - 0.5
        ang, sc, k2 = rng.uniform(-3.2, 3.2), rng.uniform(0.05, 1.6), rng.uniform(-0.3, 0.3)
        kk = 1.0 + k2 * (u * u + v * v)
        mx = (((u * np.cos(ang) - v * np.sin(ang)) * kk * sc + rng.uniform(0.2, 0.8)) * (W - 1)).astype(np.float32)
        my = (((u * np.sin(ang) + v * np.cos(ang)) * kk * sc + rng.uniform(0.2, 0.8)) * (H - 1)).astype(np.float32)
        if rng.random() < 0.3:                           # exact grid points: weight-zero taps on the last column / row
            mx, my = np.rint(mx).astype(np.float32), np.rint(my).astype(np.float32)
        valid = (rng.random((h, w)) > 0.2) if rng.random() < 0.5 else None
        fill = int(rng.integers(0, 256))
        d = [ctx.to_device(mx), ctx.to_device(my), ctx.to_device(valid.astype(np.uint8)) if valid is not None else None]
        plans.append(ctx.map_plan(d[0], d[1], d[2], h, w))
        for b in d:
            if b is not None:
                ctx.free(b)
        stride = w * 3 + pad
        dst = ctx.alloc(h * stride)
        ctx.memset(dst, 0xAB)
        bufs.append(dst)
        strides.append(stride)
        jobs.append(gs360.capi.RemapJob(d_src.ptr, H, W, 0, None, None, dst.ptr if valid is not None else None, h, w, fill, dst.ptr, stride))
        want = orc.remap_u8(src, mx, my, interp=1, border_value=bv, threads=0)
        wants.append((orc.valid_fill(want.copy(), valid, fill) if valid is not None else want).reshape(h, w, 3))
    import ctypes as C
    arr = (gs360.capi.RemapJob * n_jobs)(*jobs)
    pl = (C.c_void_p * n_jobs)(*plans)
    cbv = (C.c_double * 4)(*bv)
    with ctx.options(table_stage=1, table_stage_rows=rows):
        gs360.capi._check(ctx.L.gs360_remap_plans_u8(ctx.handle, arr, pl, n_jobs, 3, 1, cbv, 0), ctx.L)
        ctx.sync(0)
        staged = ctx.get_option("last_table_kernel")
    ok = staged == n_jobs
    if not ok:
        print(f"[tablestage] case {case}: {staged} of {n_jobs} jobs took the staged kernel")
    for k in range(n_jobs):
        h, w = wants[k].shape[:2]
        raw = ctx.download(bufs[k], (h, strides[k]))
        got = raw[:, :w * 3].reshape(h, w, 3)
        if not np.array_equal(got, wants[k]):
            ok = False
            bad = np.argwhere(got != wants[k])
            print(f"[tablestage] case {case}: job {k} src {W}x{H} map {w}x{h} rows={rows} stride={strides[k]}: {len(bad)} bytes differ, first at {bad[0].tolist()}")
        if strides[k] > w * 3 and not np.all(raw[:, w * 3:] == 0xAB):
            ok = False
            print(f"[tablestage] case {case}: job {k} wrote into the row padding")
    for pl_ in plans:
        ctx.map_plan_free(pl_)
    for b in bufs + [d_src]:
        ctx.free(b)
    return ok


def _remap_planned(ctx, src, mx, my, valid, interp, bv, fill, dtype):
    H, W, c = src.shape
    h, w = mx.shape
    d = [ctx.to_device(np.ascontiguousarray(src)), ctx.to_device(np.ascontiguousarray(mx)), ctx.to_device(np.ascontiguousarray(my)),
         ctx.to_device(np.ascontiguousarray(valid, dtype=np.uint8)) if valid is not None else None,
         ctx.alloc(h * w * c * np.dtype(dtype).itemsize)]
    plan = ctx.map_plan(d[1], d[2], d[3], h, w, nearest=(interp == 0))
    try:
        ctx.remap_plans_dev([(d[0], H, W, plan, valid is not None, h, w, fill, d[4])], c, interp=interp, border_value=bv, dtype=dtype)
        return ctx.download(d[4], (h, w, c), dtype=dtype)
    finally:
        ctx.map_plan_free(plan)
        for b in d:
            if b is not None:
                ctx.free(b)


def fuzz_fisheye(ctx, rng, case):
    c = int(rng.choice([1, 3, 3, 4]))
    W = int(rng.integers(40, 400))
    H = int(rng.integers(40, 400))
    f = float(rng.uniform(0.2, 0.5) * min(W, H))
    kw = dict(width=W, height=H, f=f, cx=float(rng.uniform(-5, 5)), cy=float(rng.uniform(-5, 5)), k1=float(rng.uniform(-0.1, 0.15)),
              k2=float(rng.uniform(-0.02, 0.02)), k3=float(rng.uniform(-0.003, 0.003)), k4=float(rng.uniform(-0.0005, 0.0005)))
    if rng.random() < 0.5:
        kw.update(p1=float(rng.uniform(-0.002, 0.002)), p2=float(rng.uniform(-0.002, 0.002)), b1=float(rng.uniform(-2, 2)), b2=float(rng.uniform(-1, 1)))
    src = rng.integers(0, 256, (H, W, c), dtype=np.uint8)
    spec = (float(rng.uniform(-200, 200)), float(rng.uniform(-80, 80)), float(rng.uniform(20, 150)), float(rng.uniform(20, 150)),
            int(rng.integers(1, 150)), int(rng.integers(1, 100)))
    lens_fov = float(rng.uniform(120, 220))
    interp = int(rng.choice([0, 1, 1, 2, 4]))
    mask_outside = bool(rng.random() < 0.7)
    mval = int(rng.integers(0, 256))
    d_src = ctx.to_device(src)
    d_out = ctx.alloc(spec[4] * spec[5] * c)
    d_val = ctx.alloc(spec[4] * spec[5])
    ctx.fisheye_views_dev([d_src], [gs360.Calib.make(**kw)], c, [gs360.View.make(*spec)], lens_fov, [d_out], valid_outs=[d_val],
                          interp=interp, mask_outside=mask_outside, mask_value=mval)
    got = ctx.download(d_out, (spec[5], spec[4], c))
    gval = ctx.download(d_val, (spec[5], spec[4]))
    mx, my, valid = orc.fisheye_spec_map(orc.make_calib(**kw), *spec, lens_fov)
    want = orc.remap_u8(src, mx, my, interp=interp, border_value=float(mval), threads=0)
    if mask_outside:
        want = orc.valid_fill(want.copy(), valid, mval)
    for b in (d_src, d_out, d_val):
        ctx.free(b)
    if not np.array_equal(got, want.reshape(got.shape)) or not np.array_equal(gval.astype(bool), valid):
        bad = np.argwhere(got != want.reshape(got.shape))
        print(f"[fisheye] case {case}: lens {W}x{H}x{c} view {spec} fov={lens_fov:.1f} interp={interp} mask={mask_outside}: "
              f"{len(bad)} bytes differ, valid differs at {int((gval.astype(bool) != valid).sum())} px")
        return False
    return True


def fuzz_color(ctx, rng, case):
    from gs360 import color
    from oracle import color_np
    n = int(rng.integers(2, 34))
    table = (rng.random((n, n, n, 3), dtype=np.float32) * np.float32(1.3) - np.float32(0.15)).astype(np.float32)
    if rng.random() < 0.5:     # smooth LUT (realistic), else noise
        g = np.linspace(0, 1, n, dtype=np.float32)
        bb, gg, rr = np.meshgrid(g, g, g, indexing="ij")
        table = np.stack([rr ** np.float32(rng.uniform(0.4, 2.0)), gg * np.float32(0.8) + bb * np.float32(0.2),
                          bb ** np.float32(rng.uniform(0.4, 2.0))], -1).astype(np.float32)
    dmin = np.float32(rng.uniform(0, 0.2, 3)) if rng.random() < 0.4 else np.zeros(3, np.float32)
    dmax = np.float32(rng.uniform(0.7, 1.0, 3)) if rng.random() < 0.4 else np.ones(3, np.float32)
    space = str(rng.choice(["srgb", "passthrough"]))
    c = int(rng.choice([3, 3, 4]))
    img = rng.integers(0, 256, (int(rng.integers(1, 90)), int(rng.integers(1, 300)), c), dtype=np.uint8)
    red = int(rng.choice([0, 2]))
    stage = color.ColorStage(color.CubeLUT(n, table, dmin, dmax), space)
    got = stage.apply(ctx, img, red_index=red)
    stage.close()
    want = color_np.color_pipeline(img, table, dmin, dmax, space, red_index=red)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        print(f"[color] case {case}: lut {n}^3 {space} img {img.shape} red={red}: {len(bad)} bytes differ, first at {bad[0].tolist()}")
        return False
    return True


def fuzz_u16(ctx, rng, case):
    """16-bit samplers and colour stage: random shapes, strides are tight (host conveniences), all interpolations"""
    from gs360 import color
    from oracle import color_np
    kind = int(rng.integers(0, 3))
    c = int(rng.choice([1, 3, 3, 4]))
    if kind == 0:        # equirect u16
        W, H = int(rng.integers(8, 500)), int(rng.integers(2, 260))
        src = rng.integers(0, 65536, (H, W, c), dtype=np.uint16)
        specs = [(float(rng.uniform(-400, 400)), 0.0 if rng.random() < 0.4 else float(rng.uniform(-95, 95)), float(rng.uniform(5, 179)),
                  float(rng.uniform(5, 179)), int(rng.integers(1, 160)), int(rng.integers(1, 100))) for _ in range(int(rng.integers(1, 4)))]
        if rng.random() < 0.5:   # yaw rings (whole-texel yaw steps, flipped pitch) and extreme samples for the 32-bit cubic accumulation
            if rng.random() < 0.5:
                W = int(rng.choice([8, 24, 48, 96, 240, 360]))
                src = rng.integers(0, 65536, (H, W, c), dtype=np.uint16)
            count = int(rng.choice([2, 3, 4, 6, 8]))
            base = specs[0]
            pitch = 0.0 if rng.random() < 0.35 else float(rng.choice([30.0, -30.0, 90.0, float(rng.uniform(-95, 95))]))
            specs = [((i % count) * 360.0 / count + (float(rng.uniform(-3, 3)) if rng.random() < 0.1 else 0.0),
                      (-pitch if rng.random() < 0.4 else pitch), base[2], base[3], base[4], base[5]) for i in range(int(rng.integers(2, 19)))]
            if rng.random() < 0.5:
                src[rng.random(src.shape[:2]) < 0.3] = 65535
                src[rng.random(src.shape[:2]) < 0.3] = 0
        interp = int(rng.choice([1, 2]))
        fish = rng.random() < 0.25
        got = ctx.equirect_views(src, [gs360.View.make(*s) for s in specs], interp=interp, flags=gs360.EQ_FISHEYE_OUT if fish else 0)
        want = orc.equirect_views_u16(src, [orc.make_view(*s) for s in specs], interp=interp, fisheye=fish, threads=0)
        ok = all(np.array_equal(g, w) for g, w in zip(got, want))
        what = f"equirect src {W}x{H}x{c} interp={interp} fish={fish} views={specs}"
    elif kind == 1:      # table remap u16
        W, H, w, h = int(rng.integers(1, 260)), int(rng.integers(1, 160)), int(rng.integers(1, 200)), int(rng.integers(1, 80))
        src = rng.integers(0, 65536, (H, W, c), dtype=np.uint16)
        mx = rng.uniform(-8, W + 8, (h, w)).astype(np.float32)
        my = rng.uniform(-8, H + 8, (h, w)).astype(np.float32)
        if rng.random() < 0.4:
            mx = (rng.integers(-64, 32 * W + 64, (h, w)) / 32.0 + rng.choice([0.0, 1 / 64.0], (h, w))).astype(np.float32)
        sel = rng.random((h, w))
        mx[sel < 0.02] = np.nan
        my[(sel > 0.02) & (sel < 0.04)] = np.inf
        interp = int(rng.choice([0, 1, 2, 4]))
        bv = tuple(float(v) for v in rng.integers(0, 70000, 4))
        valid = (rng.random((h, w)) > 0.2) if rng.random() < 0.5 else None
        fill = int(rng.integers(0, 65536))
        if rng.random() < 0.5:
            got = _remap_planned(ctx, src, mx, my, valid, interp, bv, fill, np.uint16)
        else:
            got = ctx.remap(src, mx, my, interpolation=interp, border_value=bv, valid=valid, fill_value=fill)
        want = orc.remap_u16(src, mx, my, interp=interp, border_value=bv, threads=0)
        if valid is not None:
            want = orc.valid_fill(want.copy(), valid, fill)
        ok = np.array_equal(got.reshape(want.shape), want)
        what = f"table src {W}x{H}x{c} map {w}x{h} interp={interp}"
    else:                # colour stage u16
        n = int(rng.integers(2, 20))
        g = np.linspace(0, 1, n, dtype=np.float32)
        bb, gg, rr = np.meshgrid(g, g, g, indexing="ij")
        table = np.stack([rr ** np.float32(rng.uniform(0.4, 2.0)), gg * np.float32(0.8) + bb * np.float32(0.2),
                          bb ** np.float32(rng.uniform(0.4, 2.0))], -1).astype(np.float32)
        if rng.random() < 0.3:
            table = (rng.random((n, n, n, 3), dtype=np.float32) * np.float32(1.3) - np.float32(0.15)).astype(np.float32)
        dmin = np.float32(rng.uniform(0, 0.2, 3)) if rng.random() < 0.4 else np.zeros(3, np.float32)
        dmax = np.float32(rng.uniform(0.7, 1.0, 3)) if rng.random() < 0.4 else np.ones(3, np.float32)
        space = str(rng.choice(["srgb", "passthrough"]))
        cc = int(rng.choice([3, 4]))
        img = rng.integers(0, 65536, (int(rng.integers(1, 60)), int(rng.integers(1, 200)), cc), dtype=np.uint16)
        red = int(rng.choice([0, 2]))
        stage = _stage16(n, table, dmin, dmax, space)
        got = stage.apply(ctx, img, red_index=red)
        stage.close()
        want = color_np.color_pipeline(img, table, dmin, dmax, space, red_index=red)
        ok = np.array_equal(got, want)
        what = f"colour lut {n}^3 {space} img {img.shape} red={red}"
    if not ok:
        print(f"[u16] case {case}: {what}: differs from the oracle")
    return ok


_PIECES16 = {}


def _stage16(n, table, dmin, dmax, space):
    """ColorStage whose 16-bit output tables (a pure function of the colour space) are computed once per run"""
    from gs360 import color
    stage = color.ColorStage(color.CubeLUT(n, table, dmin, dmax), space)
    if space not in _PIECES16:
        _PIECES16[space] = color.output_pieces16(space)
    stage._pieces16 = _PIECES16[space]
    return stage


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT",
                    help="context option (include/gs360.h: gs360_ctx_set_option), e.g. lanemap=1, stage=1, ring=3, srcmajor=1")
    ap.add_argument("--only", default="", help="comma list of case families (equirect,table,fisheye,color,u16,srcmajor,tablestage)")
    args = ap.parse_args()
    ctx = gs360.Context(0, n_slots=2)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    t0 = time.time()
    fns = {"equirect": fuzz_equirect, "table": fuzz_table, "fisheye": fuzz_fisheye, "color": fuzz_color, "u16": fuzz_u16, "srcmajor": fuzz_srcmajor, "tablestage": fuzz_tablestage}
    names = tuple(n for n in fns if not args.only or n in args.only.split(","))
    counts = {n: 0 for n in names}
    case, failures = 0, 0
    while time.time() - t0 < args.seconds and failures < 5:
        name = names[case % len(names)]
        rng = np.random.default_rng([args.seed, case])
        if not fns[name](ctx, rng, f"{args.seed}:{case}"):
            failures += 1
        counts[name] += 1
        case += 1
    print(f"fuzz_parity: {case} cases in {time.time() - t0:.1f} s {counts}, failures={failures}")
    ctx.close()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
