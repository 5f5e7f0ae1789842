"""-m gpu: the gs360_360PerspCut drop-in end to end -- CLI on a synthetic folder, outputs compared with the oracle."""
import pathlib
import subprocess
import sys
import threading

import numpy as np
import pytest

import gs360_360PerspCut as cut
from conftest import PKG
from gs360 import imageio

pytestmark = pytest.mark.gpu
EXE = [sys.executable, str(PKG / "cli_tools" / "gs360_360PerspCut.py")]


def make_panos(d, n=2, w=512, h=256):
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(77)
    out = {}
    for k in range(n):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        imageio.write_image(d / f"pano_{k:03d}.png", a)
        out[f"pano_{k:03d}"] = a
    return out


def test_cli_renders_full360coverage_folder(tmp_path, orc):
    src = make_panos(tmp_path / "in")
    r = subprocess.run(EXE + ["-i", str(tmp_path / "in"), "--preset", "full360coverage", "--size", "96", "--ext", "png", "-j", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[INFO] parallel jobs: 3 / total: 24" in r.stdout
    assert "[OK] Completed: success=24, failed=0, total=24" in r.stdout
    assert "[INFO] View summary (pano_000.png): 12 views - A, B_U, B_D, C, D_U, D_D, E, F_U, F_D, G, H_U, H_D" in r.stdout
    out_dir = tmp_path / "in" / "_geometry"
    files = sorted(p.name for p in out_dir.iterdir())
    assert len(files) == 24 and files[0] == "pano_000_A.png" and "pano_001_H_D.png" in files
    # plan the same job in-process to get the view table, then compare every output with the oracle
    args = cut.create_arg_parser().parse_args(["-i", str(tmp_path / "in"), "--preset", "full360coverage", "--size", "96", "--ext", "png"])
    for a in ("size", "hfov", "focal_mm"):
        setattr(args, a + "_explicit", getattr(args, a + "_explicit", False))
    args.input_is_video, args.video_bit_depth = False, 8
    plan = cut.build_view_jobs(args, sorted((tmp_path / "in").glob("*.png")), out_dir)
    for v in plan.view_specs:
        want = orc.equirect_views_u8(src[v.source_path.stem], [orc.make_view(v.yaw_deg, v.pitch_deg, v.hfov_deg, v.vfov_deg, v.width, v.height)],
                                     interp=2)[0]      # the planned argv says interp=cubic (PC:730)
        got = imageio.read_image(out_dir / v.output_name)
        assert np.array_equal(got, want), v.output_name


def test_cli_renders_fisheyexy_preset(tmp_path, orc):
    """the fisheyeXY preset (equirect -> X/Y fisheye pair, v360 output=fisheye d_fov=180) runs on the GPU, no ffmpeg"""
    src = make_panos(tmp_path / "in")
    r = subprocess.run(EXE + ["-i", str(tmp_path / "in"), "--preset", "fisheyeXY", "--size", "120", "--ext", "png"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "success=4, failed=0, total=4" in r.stdout
    out_dir = tmp_path / "in" / "_geometry"
    assert sorted(p.name for p in out_dir.iterdir()) == ["pano_000_X.png", "pano_000_Y.png", "pano_001_X.png", "pano_001_Y.png"]
    fov = 180.0 * 120 / float(np.hypot(120, 120))
    for stem, img in src.items():
        for tag, yaw in (("X", 0.0), ("Y", 180.0)):
            want = orc.equirect_fisheye_views_u8(img, [orc.make_view(yaw, 0.0, fov, fov, 120, 120)], interp=2)[0]
            assert np.array_equal(imageio.read_image(out_dir / f"{stem}_{tag}.png"), want), (stem, tag)


def test_run_one_is_thread_safe_and_cancellable(tmp_path, orc):
    src = make_panos(tmp_path / "in", n=1)
    args = cut.create_arg_parser().parse_args(["-i", str(tmp_path / "in"), "--count", "6", "--size", "64", "--ext", "png"])
    for a in ("size", "hfov", "focal_mm"):
        setattr(args, a + "_explicit", getattr(args, a + "_explicit", False))
    args.input_is_video, args.video_bit_depth = False, 8
    out_dir = tmp_path / "out"
    plan = cut.build_view_jobs(args, sorted((tmp_path / "in").glob("*.png")), out_dir)
    results = [None] * len(plan.jobs)

    def work(i):
        results[i] = cut.run_one(plan.jobs[i][0])
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(plan.jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert results == [(0, "")] * 6
    for v in plan.view_specs:
        want = orc.equirect_views_u8(src["pano_000"], [orc.make_view(v.yaw_deg, v.pitch_deg, v.hfov_deg, v.vfov_deg, v.width, v.height)], interp=2)[0]
        assert np.array_equal(imageio.read_image(out_dir / v.output_name), want)
    # missing input -> rc != 0 with text, no exception; cancelled -> 130
    bad = list(plan.jobs[0][0])
    bad[bad.index("-i") + 1] = str(tmp_path / "nope.png")
    rc, text = cut.run_one(bad)
    assert rc == 1 and "gs360" in text
    cut.stop_event.set()
    try:
        assert cut.run_one(plan.jobs[0][0]) == (130, "")
    finally:
        cut.stop_event.clear()


def test_engine_sharding_is_deterministic_across_device_counts(tmp_path, orc):
    """frames shard across devices by source path; the output set must not depend on how many devices take part.
    Two engine contexts on the one visible GPU stand in for two devices."""
    from gs360 import engine as eng
    from gs360.jobspec import parse_job_argv
    src = make_panos(tmp_path / "in", n=9, w=256, h=128)
    args = cut.create_arg_parser().parse_args(["-i", str(tmp_path / "in"), "--count", "3", "--size", "48", "--ext", "png"])
    for a in ("size", "hfov", "focal_mm"):
        setattr(args, a + "_explicit", getattr(args, a + "_explicit", False))
    args.input_is_video, args.video_bit_depth = False, 8
    results = {}
    for tag, devices in (("one", [0]), ("two", [0, 0]), ("three", [0, 0, 0])):
        out_dir = tmp_path / f"out_{tag}"
        plan = cut.build_view_jobs(args, sorted((tmp_path / "in").glob("*.png")), out_dir)
        e = eng.Engine(devices=devices)
        per_dev = {}
        for argv, _s, _d in plan.jobs:
            job = parse_job_argv(argv)
            per_dev.setdefault(e.device_for(job.src), set()).add(str(job.src))
            e.run_job(job)
        e.close()
        results[tag] = {p.name: imageio.read_image(p) for p in sorted(out_dir.iterdir())}
        assert len(results[tag]) == 27
        loads = [len(per_dev.get(d, ())) for d in range(len(devices))]
        assert sum(loads) == 9 and max(loads) - min(loads) <= 1, loads     # balanced: at most one frame of spread
    for name, img in results["one"].items():
        assert np.array_equal(img, results["two"][name]) and np.array_equal(img, results["three"][name])


def test_cli_16bit_stills_keep_their_depth(tmp_path, orc):
    """a 16-bit TIFF panorama -> 16-bit views (the reference's image mode sets no -pix_fmt for PNG/TIFF, PC:327-347, so ffmpeg
    keeps the depth); JPEG output is the 8-bit container"""
    d = tmp_path / "in"
    d.mkdir()
    a = np.random.default_rng(3).integers(0, 65536, (256, 512, 3), dtype=np.uint16)
    imageio.write_image(d / "pano.tif", a)
    r = subprocess.run(EXE + ["-i", str(d), "--count", "4", "--size", "72", "--ext", "tif"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "success=4, failed=0, total=4" in r.stdout
    args = cut.create_arg_parser().parse_args(["-i", str(d), "--count", "4", "--size", "72", "--ext", "tif"])
    for x in ("size", "hfov", "focal_mm"):
        setattr(args, x + "_explicit", getattr(args, x + "_explicit", False))
    args.input_is_video, args.video_bit_depth = False, 8
    plan = cut.build_view_jobs(args, [d / "pano.tif"], d / "_geometry")
    for v in plan.view_specs:
        want = orc.equirect_views_u16(a, [orc.make_view(v.yaw_deg, v.pitch_deg, v.hfov_deg, v.vfov_deg, v.width, v.height)], interp=2)[0]
        got = imageio.read_image(d / "_geometry" / v.output_name)
        assert got.dtype == np.uint16 and np.array_equal(got, want), v.output_name
    r = subprocess.run(EXE + ["-i", str(d), "--count", "2", "--size", "40", "--ext", "jpg", "-o", str(tmp_path / "j")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and imageio.read_image(tmp_path / "j" / "pano_A.jpg").dtype == np.uint8
