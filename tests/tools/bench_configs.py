#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs (device-resident, HIP-event timed, parity-checked on one
view each).  Informational -- bench.py (cfg2) is the headline.  Prints one JSON object per config.

    python tests/tools/bench_configs.py [--steps 50]
"""
import argparse
import json
import os
import pathlib
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the oracle's OpenMP team must not spin on the host between timed loops

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))
sys.path.insert(0, str(ROOT / "tests"))

import numpy as np  # noqa: E402

import gs360  # noqa: E402
from gs360 import fisheye as fe  # noqa: E402
from oracle import orc  # noqa: E402  (checker + algorithmic-byte counter only)
from util import PRESET_FISHEYELIKE, PRESET_FULL360, HFOV_12MM, HFOV_14MM, HFOV_17MM, TEMPLATE_CALIB, ring_views  # noqa: E402


def synth(h, w, k=0, c=3):
    x = np.arange(w, dtype=np.uint32)[None, :]
    y = np.arange(h, dtype=np.uint32)[:, None]
    n = (((x * np.uint32(2654435761)) ^ (y * np.uint32(40503 + 977 * k))) >> np.uint32(27)).astype(np.uint8)
    img = np.empty((h, w, c), np.uint8)
    for ch in range(c):
        img[..., ch] = (((x + 31 * ch) * 255) // w).astype(np.uint8) + n
    return img


# untimed launches before every timed loop until the device has been busy this long: its clocks ramp over ~100 ms of load, and a
# 5-launch warm-up (2 ms) in front of a 10-20 ms timed loop measures the ramp (cfg5: 89 us per frame against 74 in steady state).
# bench.py's headline has done this since round 3 (--settle-ms); since round 4 every row here does (GS360_BENCH_SETTLE_MS, 0 = off).
SETTLE_MS = float(os.environ.get("GS360_BENCH_SETTLE_MS", "100"))


def time_steps(ctx, call, steps, warmup=5):
    call()                                      # first call apart: it may build a plan on the host (source-major: up to 0.2 s), which must
    ctx.sync(-1)                                # not count as device-busy time of the settle phase
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < SETTLE_MS:
        for _ in range(4):
            call()
        ctx.sync(-1)
    for _ in range(warmup):
        call()
    ctx.sync(-1)
    ctx.event_record(0, 0)
    for _ in range(steps):
        call()
    ctx.event_record(0, 1)
    return ctx.event_elapsed_ms(0, 0, 1) / steps


def equirect_cfg(ctx, name, W, H, specs, n_frames, steps, interp=gs360.INTERP_LINEAR, with_mask=False, dtype=np.uint8, hot=False):
    if dtype == np.uint16:
        return equirect_u16_cfg(ctx, name, W, H, specs, n_frames, steps, interp)
    base = synth(H, W, 0)                       # frame k = the base image rolled 97 k texels (distinct HBM-resident frames, one synthesis)
    if hot:                                     # SURVEY 8(d)(i)'s "hot" number: ONE frame in every slot of the launch (88.5 MB: Infinity-Cache-resident)
        frames = [base] * n_frames
        d0 = ctx.to_device(base)
        d_fr = [d0] * n_frames
    else:
        frames = [base] + [np.ascontiguousarray(np.roll(base, 97 * k, axis=1)) for k in range(1, n_frames)]
        d_fr = [ctx.to_device(f) for f in frames]
    views = [gs360.View.make(*s) for s in specs]
    d_out = [ctx.alloc(s[4] * s[5] * 3) for _ in range(n_frames) for s in specs]
    masks = d_masks = None
    if with_mask:   # disks of value 0 on a 255 background, seed-fixed
        rng = np.random.default_rng(20260424)
        yy, xx = np.ogrid[:H, :W]
        m = np.full((H, W), 255, np.uint8)
        for _ in range(40):
            cy, cx, r = int(rng.integers(0, H)), int(rng.integers(0, W)), int(rng.integers(40, 400))
            m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0
        masks = [np.ascontiguousarray(np.roll(m, 97 * k, axis=1)) for k in range(n_frames)]
        d_masks = [ctx.to_device(mm) for mm in masks]

        def fp_call():
            ctx.equirect_views_dev(d_fr, W, H, 3, views, d_out, slot=0, interp=interp, masks=d_masks)
    else:
        fp_call = ctx.make_equirect_call(d_fr, W, H, 3, views, d_out, slot=0, interp=interp)
    ms = time_steps(ctx, fp_call, steps)
    eq_kernel = {0: "gather", 1: "lds-staged", 2: "source-major"}.get(ctx.get_option("last_eq_kernel"), "?")
    # parity of one view of frame 0, algorithmic bytes from the oracle
    k = len(specs) // 2
    got = ctx.download(d_out[k], (specs[k][5], specs[k][4], 3))
    want = orc.equirect_views_u8(frames[0], [orc.make_view(*specs[k])], threads=0, interp=2 if interp == 2 else 1,
                                 mask=masks[0] if masks else None)[0]
    union = np.zeros((H, W), np.uint8)          # SURVEY 8(d)'s stricter figure: every distinct texel of the frame ONCE for all views
    uv = sum(orc.equirect_distinct_texels(orc.make_view(*s), W, H, union) for s in specs)
    out_px = sum(s[4] * s[5] for s in specs)
    algo = (out_px * 3 + uv * 3) * n_frames
    algo_union = (out_px * 3 + int(union.sum()) * 3) * n_frames
    for b in (d_fr[:1] if hot else d_fr) + d_out + (d_masks or []):
        ctx.free(b)
    return {"config": name, "frames_per_launch": n_frames, "views": len(specs), "out_MPix_per_frame": round(out_px / 1e6, 2),
            "ms_per_launch": round(ms, 4), "us_per_frame": round(ms / n_frames * 1e3, 1),
            "MPix_per_s": round(out_px * n_frames / ms / 1e3, 0), "algorithmic_MB_per_frame": round(algo / n_frames / 1e6, 1),
            "achieved_GB_per_s": round(algo / ms / 1e6, 0), "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3),
            "union_MB_per_frame": round(algo_union / n_frames / 1e6, 1), "frac_union": round(algo_union / ms / 1e6 / 8000, 3),
            "eq_kernel": eq_kernel, "parity_vs_oracle": bool(np.array_equal(got, want))}


def equirect_u16_cfg(ctx, name, W, H, specs, n_frames, steps, interp):
    """the first-cut 16-bit kernel (rgb48 frames): same workload shape, 2-byte samples"""
    frames = [(synth(H, W, k).astype(np.uint16) * 257) ^ np.uint16(k + 1) for k in range(n_frames)]
    d_fr = [ctx.to_device(f) for f in frames]
    views = [gs360.View.make(*s) for s in specs]
    d_out = [ctx.alloc(s[4] * s[5] * 6) for _ in range(n_frames) for s in specs]

    def call():
        ctx.equirect_views_dev(d_fr, W, H, 3, views, d_out, slot=0, interp=interp, dtype=np.uint16)
    ms = time_steps(ctx, call, steps)
    k = len(specs) // 2
    got = ctx.download(d_out[k], (specs[k][5], specs[k][4], 3), dtype=np.uint16)
    want = orc.equirect_views_u16(frames[0], [orc.make_view(*specs[k])], threads=0, interp=2 if interp == 2 else 1)[0]
    uv = sum(orc.equirect_distinct_texels(orc.make_view(*s), W, H) for s in specs)
    out_px = sum(s[4] * s[5] for s in specs)
    algo = (out_px * 6 + uv * 6) * n_frames
    for b in d_fr + d_out:
        ctx.free(b)
    return {"config": name, "frames_per_launch": n_frames, "views": len(specs), "ms_per_launch": round(ms, 4),
            "us_per_frame": round(ms / n_frames * 1e3, 1), "MPix_per_s": round(out_px * n_frames / ms / 1e3, 0),
            "algorithmic_MB_per_frame": round(algo / n_frames / 1e6, 1), "achieved_GB_per_s": round(algo / ms / 1e6, 0),
            "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3), "parity_vs_oracle": bool(np.array_equal(got, want))}


def fisheye_cfg(ctx, steps):
    cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
    c = fe.SensorCalibration("0", "equisolid_fisheye", cal_kw["width"], cal_kw["height"], cal_kw["f"], cal_kw["cx"], cal_kw["cy"],
                             cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
    specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)[:6]          # BASELINE cfg4: 6 views
    tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
    imgs = {"X": synth(4000, 4000, 1), "Y": synth(4000, 4000, 2)}
    dev = {k: ctx.to_device(v) for k, v in imgs.items()}
    res = []
    d_tab = {v: (ctx.to_device(t["map_x"]), ctx.to_device(t["map_y"]), ctx.to_device(np.ascontiguousarray(t["valid"], np.uint8)))
             for v, t in tables.items()}
    d_out = {v: ctx.alloc(1750 * 1750 * 3) for v in tables}

    jobs = [(dev[tables[s["view_id"]]["lens_key"]], 4000, 4000) + tuple(d_tab[s["view_id"]]) + (1750, 1750, 0, d_out[s["view_id"]])
            for s in specs]

    def table_call():          # the six views of the pair in one batched launch (gs360_remap_tables_u8)
        ctx.remap_tables_dev(jobs, 3, interp=1, border_value=(0, 0, 0, 0), slot=0)
    ms = time_steps(ctx, table_call, steps)
    v0 = specs[1]["view_id"]
    t = tables[v0]
    got = ctx.download(d_out[v0], (1750, 1750, 3))
    want = orc.valid_fill(orc.remap_u8(imgs[t["lens_key"]], t["map_x"], t["map_y"], interp=1, threads=0), t["valid"], 0)
    uv = sum(orc.table_distinct_texels(tables[s["view_id"]]["map_x"], tables[s["view_id"]]["map_y"], 4000, 4000) for s in specs)
    px = 6 * 1750 * 1750
    # algorithmic = stores + distinct source texels; the maps (8 B per pixel + 1 B valid; 4 + 1 through a plan) are OVERHEAD
    algo_px = px * 3 + uv * 3
    algo_table = algo_px + px * (8 + 1)
    res.append({"config": "cfg4 dual-fisheye 2x4000^2 -> 6x1750^2, TABLE mode (reference-identical maps), same pair every step (Infinity-Cache-warm)",
                "ms_per_pair": round(ms, 4),
                "MPix_per_s": round(px / ms / 1e3, 0), "algorithmic_MB_per_pair": round(algo_px / 1e6, 1), "overhead_MB_per_pair": round(px * 9 / 1e6, 1),
                "achieved_GB_per_s": round(algo_px / ms / 1e6, 0), "frac_of_8TBps": round(algo_px / ms / 1e6 / 8000, 3),
                "frac_incl_map_bytes": round(algo_table / ms / 1e6 / 8000, 3),
                "parity_vs_oracle": bool(np.array_equal(got, want))})
    # the same launch through MAP PLANS: the float tables packed once (5 bytes per pixel instead of 9), as the drop-in CLI runs 8-bit pairs
    plans = {s["view_id"]: {nearest: None for nearest in (False, True)} for s in specs}
    for interp, label in ((1, "linear"), (2, "cubic")):
        for s in specs:
            if plans[s["view_id"]][False] is None:
                plans[s["view_id"]][False] = ctx.map_plan(*d_tab[s["view_id"]], 1750, 1750, nearest=False)
        pjobs = [(dev[tables[s["view_id"]]["lens_key"]], 4000, 4000, plans[s["view_id"]][False], True, 1750, 1750, 0, d_out[s["view_id"]])
                 for s in specs]

        def plan_call():
            ctx.remap_plans_dev(pjobs, 3, interp=interp, border_value=(0, 0, 0, 0), slot=0)
        ms = time_steps(ctx, plan_call, steps)
        got = ctx.download(d_out[v0], (1750, 1750, 3))
        want_i = orc.valid_fill(orc.remap_u8(imgs[t["lens_key"]], t["map_x"], t["map_y"], interp=interp, threads=0), t["valid"], 0)
        algo_plan = algo_px + px * (4 + 1)
        res.append({"config": f"cfg4 dual-fisheye 2x4000^2 -> 6x1750^2, TABLE mode through map plans, {label}, same pair every step (Infinity-Cache-warm)",
                    "ms_per_pair": round(ms, 4),
                    "MPix_per_s": round(px / ms / 1e3, 0), "algorithmic_MB_per_pair": round(algo_px / 1e6, 1), "overhead_MB_per_pair": round(px * 5 / 1e6, 1),
                    "achieved_GB_per_s": round(algo_px / ms / 1e6, 0), "frac_of_8TBps": round(algo_px / ms / 1e6 / 8000, 3),
                    "frac_incl_map_bytes": round(algo_plan / ms / 1e6 / 8000, 3),
                    "parity_vs_oracle": bool(np.array_equal(got, want_i))})
    # ... and with FOUR lens pairs taken in turn (384 MB of sources: each pair's images come from HBM, as in a run over many pairs;
    # the rows above render the same pair every step, Infinity-Cache-warm)
    more = [{k: ctx.to_device(np.ascontiguousarray(np.roll(v, 211 * (r + 1), axis=1))) for k, v in imgs.items()} for r in range(3)]
    pair_devs = [dev] + more
    rot_jobs = [[(pd[tables[s["view_id"]]["lens_key"]], 4000, 4000, plans[s["view_id"]][False], True, 1750, 1750, 0, d_out[s["view_id"]])
                 for s in specs] for pd in pair_devs]
    turn = [0]

    def rot_call():
        ctx.remap_plans_dev(rot_jobs[turn[0] & 3], 3, interp=1, border_value=(0, 0, 0, 0), slot=0)
        turn[0] += 1
    ms = time_steps(ctx, rot_call, steps)
    last = (turn[0] - 1) & 3
    src_last = imgs[t["lens_key"]] if last == 0 else np.roll(imgs[t["lens_key"]], 211 * last, axis=1)
    got = ctx.download(d_out[v0], (1750, 1750, 3))
    want_r = orc.valid_fill(orc.remap_u8(np.ascontiguousarray(src_last), t["map_x"], t["map_y"], interp=1, threads=0), t["valid"], 0)
    res.append({"config": "cfg4 dual-fisheye 2x4000^2 -> 6x1750^2, TABLE mode through map plans, linear, four pairs in turn (sources from HBM)",
                "ms_per_pair": round(ms, 4), "MPix_per_s": round(px / ms / 1e3, 0), "algorithmic_MB_per_pair": round(algo_px / 1e6, 1),
                "overhead_MB_per_pair": round(px * 5 / 1e6, 1),
                "achieved_GB_per_s": round(algo_px / ms / 1e6, 0), "frac_of_8TBps": round(algo_px / ms / 1e6 / 8000, 3),
                "frac_incl_map_bytes": round(algo_plan / ms / 1e6 / 8000, 3),
                "parity_vs_oracle": bool(np.array_equal(got, want_r))})
    for pd in more:
        for b in pd.values():
            ctx.free(b)
    # the same pair as 16-bit images: CV_16U samplers, all six views in one batched launch (gs360_remap_tables_u16)
    imgs16 = {k: (v.astype(np.uint16) * 257) ^ np.uint16(3) for k, v in imgs.items()}
    dev16 = {k: ctx.to_device(v) for k, v in imgs16.items()}
    d_out16 = {s["view_id"]: ctx.alloc(1750 * 1750 * 6) for s in specs}
    jobs16 = [(dev16[tables[s["view_id"]]["lens_key"]], 4000, 4000) + tuple(d_tab[s["view_id"]]) + (1750, 1750, 0, d_out16[s["view_id"]])
              for s in specs]
    pjobs16 = [(dev16[tables[s["view_id"]]["lens_key"]], 4000, 4000, plans[s["view_id"]][False], True, 1750, 1750, 0, d_out16[s["view_id"]])
               for s in specs]
    timed16 = []
    for planned in (False, True):
        for interp, label in ((1, "linear"), (2, "cubic")):     # time all first: the oracle's OpenMP team spins on the host afterwards
            def u16_call():
                if planned:
                    ctx.remap_plans_dev(pjobs16, 3, interp=interp, border_value=(0, 0, 0, 0), slot=0, dtype=np.uint16)
                else:
                    ctx.remap_tables_dev(jobs16, 3, interp=interp, border_value=(0, 0, 0, 0), slot=0, dtype=np.uint16)
            ms16 = time_steps(ctx, u16_call, steps)
            timed16.append((interp, label + (", map plans" if planned else ""), ms16, 4 if planned else 8,
                            ctx.download(d_out16[specs[-1]["view_id"]], (1750, 1750, 3), dtype=np.uint16)))
    for s in specs:
        ctx.map_plan_free(plans[s["view_id"]][False])
    for interp, label, ms16, map_bytes, got16 in timed16:
        t_last = tables[specs[-1]["view_id"]]
        want16 = orc.valid_fill(orc.remap_u16(imgs16[t_last["lens_key"]], t_last["map_x"], t_last["map_y"], interp=interp, threads=0), t_last["valid"], 0)
        algo16 = px * 6 + uv * 6
        res.append({"config": f"cfg4 shape on 16-bit lens images, TABLE mode, CV_16U {label}, one batched launch (Infinity-Cache-warm)", "ms_per_pair": round(ms16, 4),
                    "MPix_per_s": round(px / ms16 / 1e3, 0), "algorithmic_MB_per_pair": round(algo16 / 1e6, 1), "overhead_MB_per_pair": round(px * (map_bytes + 1) / 1e6, 1),
                    "achieved_GB_per_s": round(algo16 / ms16 / 1e6, 0), "frac_of_8TBps": round(algo16 / ms16 / 1e6 / 8000, 3),
                    "frac_incl_map_bytes": round((algo16 + px * (map_bytes + 1)) / ms16 / 1e6 / 8000, 3),
                    "parity_vs_oracle": bool(np.array_equal(got16, want16))})
    calib = gs360.Calib.make(c.width, c.height, c.f, c.cx, c.cy, c.k1, c.k2, c.k3)
    views = [gs360.View.make(tables[s["view_id"]]["yaw_rel_deg"], s["pitch_deg"], s["hfov_deg"], s["vfov_deg"], 1750, 1750) for s in specs]
    srcs = [dev[tables[s["view_id"]]["lens_key"]] for s in specs]
    outs = [d_out[s["view_id"]] for s in specs]

    def fused_call():
        ctx.fisheye_views_dev(srcs, [calib] * 6, 3, views, 190.0, outs, interp=1, mask_outside=True, mask_value=0, slot=0)
    ms = time_steps(ctx, fused_call, steps)
    s1 = specs[1]
    mx, my, valid = orc.fisheye_spec_map(orc.make_calib(c.width, c.height, c.f, c.cx, c.cy, c.k1, c.k2, c.k3),
                                         tables[v0]["yaw_rel_deg"], s1["pitch_deg"], s1["hfov_deg"], s1["vfov_deg"], 1750, 1750, 190.0)
    want = orc.valid_fill(orc.remap_u8(imgs[t["lens_key"]], mx, my, interp=1, threads=0), valid, 0)
    got = ctx.download(d_out[v0], (1750, 1750, 3))
    algo_fused = px * 3 + uv * 3
    res.append({"config": "cfg4 dual-fisheye 2x4000^2 -> 6x1750^2, FUSED mode (FE-SPEC v1, no map traffic)", "ms_per_pair": round(ms, 4),
                "MPix_per_s": round(px / ms / 1e3, 0), "algorithmic_MB_per_pair": round(algo_fused / 1e6, 1),
                "achieved_GB_per_s": round(algo_fused / ms / 1e6, 0), "frac_of_8TBps": round(algo_fused / ms / 1e6 / 8000, 3),
                "parity_vs_oracle": bool(np.array_equal(got, want))})
    return res


def color_cfg(ctx, steps):
    """--input-lut stage on one 4000x4000 lens image, 33^3 cube, sRGB re-encode: a smooth image (colour locality, the
    photographic case) and i.i.d. noise (every pixel in another LUT cell: the L2-gather worst case)."""
    from gs360 import color
    from oracle import color_np
    n = 33
    g = np.linspace(0, 1, n, dtype=np.float32)
    bb, gg, rr = np.meshgrid(g, g, g, indexing="ij")
    table = np.stack([rr ** 0.8, 0.9 * gg + 0.1 * bb, np.sqrt(bb)], -1).astype(np.float32)
    stage = color.ColorStage(color.CubeLUT(n, table, np.zeros(3, np.float32), np.ones(3, np.float32)), "srgb")
    res = []
    t0 = time.perf_counter()
    stage._plan(ctx)                               # plan creation: host tables, upload, the 2^24-entry cube (synchronous)
    plan_ms = (time.perf_counter() - t0) * 1e3
    which = os.environ.get("GS360_BENCH_COLOR_IMAGES", "smooth,noise,u16")     # one kind per run for counter passes
    images = []
    if "smooth" in which:
        images.append(("smooth+hash image", synth(4000, 4000, 3)))
    if "noise" in which:
        images.append(("i.i.d. noise image", np.random.default_rng(1).integers(0, 256, (4000, 4000, 3), dtype=np.uint8)))
    for label, img in images:
        d_in, d_out = ctx.to_device(img), ctx.alloc(img.nbytes)
        plan = stage._plan(ctx)

        def call():
            ctx.color_apply_dev(plan, d_in, 4000, 4000, 3, dst=d_out, slot=0)
        ms = time_steps(ctx, call, steps)
        got = ctx.download(d_out, img.shape)
        rows = np.r_[0:4000:131]
        want = color_np.color_pipeline(img[rows], table, stage.lut.domain_min, stage.lut.domain_max, "srgb", red_index=0)
        algo = 2 * img.nbytes                      # read + write of the image; the LUT (575 KB) is cache-resident
        res.append({"config": "colour stage 4000x4000x3, 33^3 LUT + Rec.709->sRGB, " + label, "ms_per_image": round(ms, 4),
                    "MPix_per_s": round(16.0 / ms * 1e3, 0), "algorithmic_MB_per_image": round(algo / 1e6, 1),
                    "achieved_GB_per_s": round(algo / ms / 1e6, 0), "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3),
                    "plan_create_ms": round(plan_ms, 2), "parity_vs_oracle": bool(np.array_equal(got[rows], want))})
        ctx.free(d_in)
        ctx.free(d_out)
    if "u16" in which:
        img16 = (synth(4000, 4000, 3).astype(np.uint16) << 8) | synth(4000, 4000, 5)        # smooth high byte, busy low byte
        d_in, d_out = ctx.to_device(img16), ctx.alloc(img16.nbytes)
        plan16 = stage._plan16(ctx)

        def call16():
            ctx.color_apply16_dev(plan16, d_in, 4000, 4000, 3, dst=d_out, slot=0)
        ms = time_steps(ctx, call16, max(3, steps // 4))
        got = ctx.download(d_out, img16.shape, dtype=np.uint16)
        rows = np.r_[0:4000:499]
        want = color_np.color_pipeline(img16[rows], table, stage.lut.domain_min, stage.lut.domain_max, "srgb", red_index=0)
        algo = 2 * img16.nbytes
        res.append({"config": "colour stage 4000x4000x3 uint16, 33^3 LUT + Rec.709->sRGB (per-pixel evaluation)", "ms_per_image": round(ms, 4),
                    "MPix_per_s": round(16.0 / ms * 1e3, 0), "algorithmic_MB_per_image": round(algo / 1e6, 1),
                    "achieved_GB_per_s": round(algo / ms / 1e6, 0), "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3),
                    "parity_vs_oracle": bool(np.array_equal(got[rows], want))})
        ctx.free(d_in)
        ctx.free(d_out)
    stage.close()
    return res


def cfg4_rows(ctx, steps, interps=((1, "linear"), (2, "cubic")), rotate=4):
    """cfg4 (2 x 4000^2 fisheye -> 6 x 1750^2, table mode, one batched launch per pair) for the given cv2 interpolations, HBM-COLD: `rotate`
    distinct lens pairs, each with its own copy of the tables / map plans, are rendered in turn (4 x 96 MB of images + 4 x 165 MB of float
    tables = 1 GB against the 256 MiB Infinity Cache), as the headline rotates 16 frames.  Bytes per SURVEY section 8(d): ALGORITHMIC =
    stores + distinct source texels; the maps' own traffic (8 B per pixel as float tables, 4 B through a plan, + 1 B valid) is reported
    beside it as OVERHEAD ("table mode's 8 B/px map reads are overhead, not algorithmic"); `frac` is on the algorithmic bytes."""
    cal_kw = dict(TEMPLATE_CALIB, width=4000, height=4000)
    c = fe.SensorCalibration("0", "equisolid_fisheye", 4000, 4000, cal_kw["f"], cal_kw["cx"], cal_kw["cy"], cal_kw["k1"], cal_kw["k2"], cal_kw["k3"])
    specs = fe.sfm10_specs(1750, 14.0, "36 36", 40.0, 40.0)[:6]
    tables = fe.choose_lens_tables({"0": c}, "0", "0", specs, 0.0, 180.0, 190.0)
    imgs = {"X": synth(4000, 4000, 1), "Y": synth(4000, 4000, 2)}
    pairs_host = [imgs] + [{k: np.ascontiguousarray(np.roll(v, 211 * r, axis=1)) for k, v in imgs.items()} for r in range(1, rotate)]
    devs = [{k: ctx.to_device(v) for k, v in ph.items()} for ph in pairs_host]
    d_tabs = [{v: (ctx.to_device(t["map_x"]), ctx.to_device(t["map_y"]), ctx.to_device(np.ascontiguousarray(t["valid"], np.uint8)))
               for v, t in tables.items()} for _ in range(rotate)]
    d_out = {v: ctx.alloc(1750 * 1750 * 3) for v in tables}
    uv = sum(orc.table_distinct_texels(tables[s["view_id"]]["map_x"], tables[s["view_id"]]["map_y"], 4000, 4000) for s in specs)
    px = 6 * 1750 * 1750
    algo = px * 3 + uv * 3                                    # stores + distinct texels
    over_float, over_plan = px * (8 + 1), px * (4 + 1)        # the maps (+ valid): overhead of table mode
    v0 = specs[1]["view_id"]
    t = tables[v0]
    plans = [{s["view_id"]: ctx.map_plan(*d_tabs[r][s["view_id"]], 1750, 1750, nearest=False) for s in specs} for r in range(rotate)]
    jobs = [[(devs[r][tables[s["view_id"]]["lens_key"]], 4000, 4000) + tuple(d_tabs[r][s["view_id"]]) + (1750, 1750, 0, d_out[s["view_id"]])
             for s in specs] for r in range(rotate)]
    pjobs = [[(devs[r][tables[s["view_id"]]["lens_key"]], 4000, 4000, plans[r][s["view_id"]], True, 1750, 1750, 0, d_out[s["view_id"]])
              for s in specs] for r in range(rotate)]
    timed = []
    for interp, label in interps:           # time everything first: the oracle's OpenMP team spins on the host afterwards
        for planned in (False, True):
            turn = [0]

            def call():
                r = turn[0] % rotate
                turn[0] += 1
                if planned:
                    ctx.remap_plans_dev(pjobs[r], 3, interp=interp, border_value=(0, 0, 0, 0), slot=0)
                else:
                    ctx.remap_tables_dev(jobs[r], 3, interp=interp, border_value=(0, 0, 0, 0), slot=0)
            ms = time_steps(ctx, call, steps)
            last = (turn[0] - 1) % rotate
            staged = ctx.get_option("last_table_kernel")            # jobs of the last call that took the LDS-staged kernel (tile records: 4.03 B/px)
            over = (int(px * (4 + 64 / 2048)) if staged else over_plan) if planned else over_float
            timed.append((interp, label + (", map plans" if planned else ""), ms, over, last,
                          ctx.download(d_out[v0], (1750, 1750, 3)), "table_staged_kernel" if planned and staged else "table_remap_kernel"))
    res = []
    for interp, label, ms, over, last, got, kern in timed:
        want = orc.valid_fill(orc.remap_u8(pairs_host[last][t["lens_key"]], t["map_x"], t["map_y"], interp=interp, threads=0), t["valid"], 0)
        res.append({"config": f"cfg4 dual-fisheye 2x4000^2 -> 6x1750^2, table mode, {label}, {rotate} pairs in turn (HBM-cold)",
                    "key": label.replace(", map plans", "-plans"), "kernel": kern,
                    "ms_per_pair": round(ms, 4), "algorithmic_MB_per_pair": round(algo / 1e6, 1), "overhead_MB_per_pair": round(over / 1e6, 1),
                    "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3), "frac_incl_map_bytes": round((algo + over) / ms / 1e6 / 8000, 3),
                    "parity_vs_oracle": bool(np.array_equal(got, want))})
    for pr in plans:
        for pl in pr.values():
            ctx.map_plan_free(pl)
    for b in [x for d in devs for x in d.values()] + [x for dt in d_tabs for tup in dt.values() for x in tup] + list(d_out.values()):
        ctx.free(b)
    return res


def secondary_rows(ctx, steps=20):
    """The other BASELINE configs in compact form for bench.py's `secondary` array: same timing method as the rows above (HIP events
    around `steps` batched launches of 4 resident frames / one lens pair), one view of each checked against the oracle."""
    full360 = [(y, p, HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360]
    fishlike = [(y, p, HFOV_17MM, HFOV_17MM, 2048, 2048) for y, p in PRESET_FISHEYELIKE]
    plan = [
        ("cfg1", "5760x2880 -> default 8x1600^2, linear", 5760, 2880, ring_views(8, 1600, HFOV_12MM), gs360.INTERP_LINEAR, False),
        ("cfg1-cubic", "5760x2880 -> default 8x1600^2, cubic (the tool's default)", 5760, 2880, ring_views(8, 1600, HFOV_12MM), gs360.INTERP_CUBIC, False),
        ("cfg2-cubic", "8K -> 6x800^2, cubic", 7680, 3840, ring_views(6, 800, HFOV_12MM), gs360.INTERP_CUBIC, False),
        ("cfg3", "8K -> full360coverage 12x1600^2, linear", 7680, 3840, full360, gs360.INTERP_LINEAR, False),
        ("cfg5", "8K -> fisheyelike 10x2048^2, linear", 7680, 3840, fishlike, gs360.INTERP_LINEAR, False),
        ("cfg5+mask", "8K -> fisheyelike 10x2048^2, linear, fused keep-mask (threshold + pack pass inside the timed region)", 7680, 3840,
         fishlike, gs360.INTERP_LINEAR, True),
        ("cfg3+mask", "8K -> full360coverage 12x1600^2, linear, fused keep-mask (pack pass inside the timed region)", 7680, 3840,
         full360, gs360.INTERP_LINEAR, True),
    ]
    plan.append(("cfg2-hot", "8K -> 6x800^2, linear, the SAME frame in all 16 slots of a launch (Infinity-Cache-warm: SURVEY 8(d)'s hot number; "
                 "the headline is the cold one)", 7680, 3840, ring_views(6, 800, HFOV_12MM), gs360.INTERP_LINEAR, False))
    out = []
    for key, name, w, h, specs, interp, with_mask in plan:
        # frames per launch as in the full rows of main(): 8 for the 5.7K / 6 x 800^2 shapes, 4 for the 8K large-view presets
        r = equirect_cfg(ctx, name, w, h, specs, 16 if key == "cfg2-hot" else 8 if key in ("cfg1", "cfg1-cubic", "cfg2-cubic") else 4, steps, interp=interp,
                         with_mask=with_mask, hot=key == "cfg2-hot")
        out.append({"config": key, "workload": name, "unit": "frame", "us_per_unit": r["us_per_frame"], "frac": r["frac_of_8TBps"],
                    "algorithmic_MB_per_unit": r["algorithmic_MB_per_frame"], "frac_union": r["frac_union"], "union_MB_per_unit": r["union_MB_per_frame"],
                    "eq_kernel": r["eq_kernel"], "parity_vs_oracle": r["parity_vs_oracle"]})
    for key, name, interp in (("cfg2-u16", "8K rgb48 (uint16) -> 6x800^2, linear", gs360.INTERP_LINEAR),
                              ("cfg2-u16-cubic", "8K rgb48 (uint16) -> 6x800^2, cubic", gs360.INTERP_CUBIC)):
        r = equirect_u16_cfg(ctx, name, 7680, 3840, ring_views(6, 800, HFOV_12MM), 4, steps, interp)
        out.append({"config": key, "workload": name, "unit": "frame", "us_per_unit": r["us_per_frame"], "frac": r["frac_of_8TBps"],
                    "algorithmic_MB_per_unit": r["algorithmic_MB_per_frame"], "parity_vs_oracle": r["parity_vs_oracle"]})
    for r in cfg4_rows(ctx, steps):
        out.append({"config": "cfg4-" + r["key"], "workload": r["config"], "unit": "lens pair",
                    "us_per_unit": round(r["ms_per_pair"] * 1e3, 1), "frac": r["frac_of_8TBps"],
                    "algorithmic_MB_per_unit": r["algorithmic_MB_per_pair"], "overhead_MB_per_unit": r["overhead_MB_per_pair"],
                    "frac_incl_overhead": r["frac_incl_map_bytes"], "kernel": r["kernel"], "parity_vs_oracle": r["parity_vs_oracle"]})
    for r in color_cfg(ctx, steps):
        kind = "color-u16" if "uint16" in r["config"] else ("color-noise" if "noise" in r["config"] else "color-smooth")
        out.append({"config": kind, "workload": r["config"], "unit": "4000^2 image", "us_per_unit": round(r["ms_per_image"] * 1e3, 1),
                    "frac": r["frac_of_8TBps"], "algorithmic_MB_per_unit": r["algorithmic_MB_per_image"], "parity_vs_oracle": r["parity_vs_oracle"]})
    # The arithmetic-bound rows (cubic: the reference tools' default interpolation, PC:730 / DF:229-234; cfg5) carry the vector-ALU busy
    # fraction of their kernel next to the HBM fraction -- their real yardstick.  Measured with rocprofv3 PMC passes
    # (profiles/tools/prof_valu.sh -> profiles/valu_busy.json), not in this run: attached only when the kernel family matches.
    try:
        vb = json.loads((ROOT / "profiles" / "valu_busy.json").read_text())["configs"]
    except Exception:
        vb = {}
    for row in out:
        rec = vb.get(row["config"])
        if rec:
            row["valu_busy"] = rec["valu_busy"]
            row["valu_busy_kernel"] = rec["kernel"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--only", default="", help="comma list of: equirect, fisheye, color (default all)")
    ap.add_argument("--secondary", action="store_true", help="print bench.py's compact `secondary` rows instead")
    ap.add_argument("--eq", default="", help="comma list of equirect rows: cfg1,cfg2,cfg2cubic,cfg2u16,cfg2u16cubic,cfg3,cfg1cubic,cfg3cubic,cfg3mask,cfg5,cfg5mask (default all)")
    args = ap.parse_args()
    ctx = gs360.Context(0, n_slots=1)
    if args.secondary:
        for r in secondary_rows(ctx, min(args.steps, 20)):
            print(json.dumps(r))
        ctx.close()
        return
    rows = []
    only = {t for t in args.only.split(",") if t}
    if only and "equirect" not in only:
        if "fisheye" in only:
            rows += fisheye_cfg(ctx, args.steps)
        if "color" in only:
            rows += color_cfg(ctx, args.steps)
        for r in rows:
            print(json.dumps(r))
        ctx.close()
        return
    eq = {
        "cfg1": lambda: equirect_cfg(ctx, "cfg1 5760x2880 -> default preset 8x1600^2", 5760, 2880, ring_views(8, 1600, HFOV_12MM), 8, args.steps),
        "cfg2": lambda: equirect_cfg(ctx, "cfg2 7680x3840 -> 6x800^2 (headline, bench.py)", 7680, 3840, ring_views(6, 800, HFOV_12MM), 8, args.steps),
        "cfg2cubic": lambda: equirect_cfg(ctx, "cfg2 with INTER_CUBIC (reference default interp)", 7680, 3840, ring_views(6, 800, HFOV_12MM), 8,
                                          args.steps, interp=gs360.INTERP_CUBIC),
        "cfg2u16": lambda: equirect_cfg(ctx, "cfg2 shape on rgb48 (uint16) frames, bilinear", 7680, 3840, ring_views(6, 800, HFOV_12MM), 4, args.steps,
                                        dtype=np.uint16),
        "cfg2u16cubic": lambda: equirect_cfg(ctx, "cfg2 shape on rgb48 (uint16) frames, cubic", 7680, 3840, ring_views(6, 800, HFOV_12MM), 4,
                                             args.steps, interp=gs360.INTERP_CUBIC, dtype=np.uint16),
        "cfg3": lambda: equirect_cfg(ctx, "cfg3 7680x3840 -> full360coverage 12x1600^2", 7680, 3840,
                                     [(y, p, HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360], 4, args.steps),
        "cfg1cubic": lambda: equirect_cfg(ctx, "cfg1 with INTER_CUBIC (the tool's own default interp, PC:730)", 5760, 2880, ring_views(8, 1600, HFOV_12MM), 8,
                                          args.steps, interp=gs360.INTERP_CUBIC),
        "cfg3cubic": lambda: equirect_cfg(ctx, "cfg3 with INTER_CUBIC", 7680, 3840,
                                          [(y, p, HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360], 4, args.steps,
                                          interp=gs360.INTERP_CUBIC),
        "cfg5": lambda: equirect_cfg(ctx, "cfg5 7680x3840 -> fisheyelike 10x2048^2 (u8, no fp16/mask fusion)", 7680, 3840,
                                     [(y, p, HFOV_17MM, HFOV_17MM, 2048, 2048) for y, p in PRESET_FISHEYELIKE], 4, args.steps),
        "cfg3mask": lambda: equirect_cfg(ctx, "cfg3 + fused keep-mask multiply", 7680, 3840,
                                         [(y, p, HFOV_14MM, HFOV_14MM, 1600, 1600) for y, p in PRESET_FULL360], 4, args.steps, with_mask=True),
        "cfg5mask": lambda: equirect_cfg(ctx, "cfg5 + fused keep-mask multiply (u8 mask, nearest, threshold 128)", 7680, 3840,
                                         [(y, p, HFOV_17MM, HFOV_17MM, 2048, 2048) for y, p in PRESET_FISHEYELIKE], 4, args.steps,
                                         with_mask=True),
    }
    for name, fn in eq.items():
        if not args.eq or name in args.eq.split(","):
            rows.append(fn())
    if not only or "fisheye" in only:
        rows += fisheye_cfg(ctx, args.steps)
    if not only or "color" in only:
        rows += color_cfg(ctx, args.steps)
    for r in rows:
        print(json.dumps(r))
    ctx.close()


if __name__ == "__main__":
    main()
