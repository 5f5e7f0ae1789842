"""Dual-fisheye drop-in: CLI contract on CPU (flags, dry-run lines, exit codes) and, on the GPU, the whole pair
pipeline in table mode against the oracle's cv2.remap restatement applied to the same reference-identical tables."""
import pathlib
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG
from gs360 import fisheye as fe, imageio

EXE = [sys.executable, str(PKG / "cli_tools" / "gs360_DualFisheyeDistortionCalibration.py")]

SMALL_XML = """<document><chunk><sensors><sensor id="0" type="equisolid_fisheye"><resolution width="240" height="240"/>
<calibration type="equisolid_fisheye" class="adjusted"><resolution width="240" height="240"/><f>65.62</f><cx>-0.05</cx><cy>0.04</cy>
<k1>0.1019</k1><k2>0.0008</k2><k3>-0.0003</k3></calibration></sensor></sensors></chunk></document>"""


def make_pairs(d, n=2, size=240, seed=5):
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    imgs = {}
    for k in range(n):
        for lens in "XY":
            a = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
            imageio.write_image(d / f"frame_{k:04d}_{lens}.png", a)
            imgs[(k, lens)] = a
    (d / "stray_Z.png").write_bytes(b"x")
    return imgs


def run(args, **kw):
    return subprocess.run(EXE + args, capture_output=True, text=True, timeout=300, **kw)


def test_default_template_calibration_values():
    import gs360_DualFisheyeDistortionCalibration as df
    sensors, labels = fe.load_metashape_calibration(df.DEFAULT_CAMERA_XML)
    c = sensors["0"]
    assert (c.width, c.height, c.model_type) == (3840, 3840, "equisolid_fisheye")
    assert (c.f, c.cx, c.cy) == (1049.9268186384606, -0.053481903280599763, -0.040449115818567277)
    assert (c.k1, c.k2, c.k3, c.k4, c.p1, c.p2, c.b1, c.b2) == (0.10190869149858893, 0.00079808296648272998,
                                                                 -0.00031893309097734927, 0.0, 0.0, 0.0, 0.0, 0.0)
    assert labels == {}


def test_parser_accepts_every_reference_flag():
    import gs360_DualFisheyeDistortionCalibration as df
    a = df.build_parser().parse_args(
        ["-i", "d", "-x", "c.xml", "-o", "o", "--suffixes", "_L,_R", "--ext", "png", "--input-lut", "l.cube",
         "--lut-output-color-space", "passthrough", "--input-color-profile", "native", "--dlogm-lut", "z", "--sensor-id-x", "0",
         "--sensor-id-y", "1", "--interpolation", "linear", "--undistort-zoom", "1.1", "--no-mask-outside-model", "--mask-value", "7",
         "--limit", "3", "--workers", "2", "--memory-throttle-percent", "50", "--dry-run", "--report-json", "r.json", "--no-perspective",
         "--save-fisheye-output", "--save-color-corrected-output", "--color-corrected-output-dir", "c", "--fisheye-output-dir", "f",
         "--no-fisheye-output", "--perspective-output-dir", "p", "--perspective-ext", "png", "--perspective-mask-ext", "png",
         "--perspective-size", "100", "--perspective-focal-mm", "10", "--perspective-sensor-mm", "36x24", "--perspective-yaw-delta-deg", "30",
         "--perspective-pitch-delta-deg", "20", "--perspective-jpeg-quality", "90", "--lens-fov-deg", "185", "--lens-x-yaw-deg", "5",
         "--lens-y-yaw-deg", "175", "--camera-extrinsics-xml", "e.xml", "--pointcloud-ply", "p.ply", "--mask-input-dir", "m",
         "--perspective-metashape-xml-name", "n.xml", "--metadata-only"])
    assert a.mask_outside_model is False and a.interpolation == "linear" and a.perspective_size == 100
    d = df.build_parser().parse_args([])
    assert (d.interpolation, d.perspective_size, d.perspective_focal_mm, d.lens_fov_deg, d.mask_value, d.undistort_zoom,
            d.perspective_ext, d.suffixes, d.lens_y_yaw_deg, d.map_mode) == ("cubic", 1750, 14.0, 190.0, 0, "auto", "jpg", "_X,_Y",
                                                                             180.0, "table")


def test_usage_errors_exit_1(tmp_path):
    assert "[ERR] --input-dir is required" in run([]).stderr and run([]).returncode == 1
    f = tmp_path / "v.mp4"
    f.write_bytes(b"")
    r = run(["-i", str(f)])
    assert r.returncode == 1 and "not a video file" in r.stderr
    r = run(["-i", str(tmp_path / "missing")])
    assert r.returncode == 1 and "[ERR] Input path not found:" in r.stderr
    d = tmp_path / "in"
    d.mkdir()
    r = run(["-i", str(d)])
    assert r.returncode == 1 and "[ERR] No target images found in" in r.stderr
    (d / "a_X.png").write_bytes(b"")
    r = run(["-i", str(d)])
    assert r.returncode == 1 and "[ERR] No valid X/Y fisheye pairs found in" in r.stderr
    for bad, msg in ((["--no-perspective"], "[ERR] All outputs are disabled."), (["--suffixes", "_X"], "[ERR] --suffixes must include"),
                     (["--undistort-zoom", "-1"], "[ERR] --undistort-zoom:"), (["--workers", "0"], "[ERR] --workers must be >= 1."),
                     (["-x", str(tmp_path / "no.xml")], "[ERR] Calibration XML not found:"),
                     (["--mask-input-dir", str(tmp_path / "nomask")], "[ERR] Mask input directory not found:"),
                     (["--lut-output-color-space", "xyz"], "[ERR] Unsupported --lut-output-color-space")):
        (d / "a_Y.png").write_bytes(b"")
        r = run(["-i", str(d)] + bad)
        assert r.returncode == 1 and msg in r.stderr, (bad, r.stderr)


def test_dry_run_lines_and_layout(tmp_path):
    d = tmp_path / "shots"
    make_pairs(d, n=2)
    xml = tmp_path / "c.xml"
    xml.write_text(SMALL_XML)
    r = run(["-i", str(d), "-x", str(xml), "--dry-run", "--save-fisheye-output", "--perspective-ext", "PNG"])
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert f"[INFO] input:  {d.resolve()}" in out
    assert f"[INFO] fisheye output: {d.resolve().with_name('shots_undistorted')}" in out
    assert f"[INFO] perspective images dir: {d.resolve().with_name('shots_perspective_colmap') / 'Images'}" in out
    assert "[INFO] pairs:  2" in out and "[INFO] files:  4" in out and "[INFO] undistort zoom: auto" in out
    assert "[DRY]    1/   2 frame_0000_X.png -> frame_0000_X.png (sensor_id=0)" in out
    persp = [l for l in out if l.startswith("[DRY][PERSP]")]
    assert len(persp) == 20 and persp[0] == "[DRY][PERSP]    1/   2 frame_0000_A.png" and persp[-1].endswith("frame_0001_J.png")
    assert out[-1] == "[DONE] processed=4 skipped=0 total=4 persp_outputs=20 mask_outputs=0 color_outputs=0 errors=0"
    assert not d.resolve().with_name("shots_perspective_colmap").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["table", "fused"])
def test_pair_pipeline_end_to_end_on_gpu(tmp_path, orc, mode):
    d = tmp_path / "shots"
    imgs = make_pairs(d, n=2)
    masks = tmp_path / "masks"
    masks.mkdir()
    rng = np.random.default_rng(9)
    mimgs = {}
    for p in sorted(d.glob("frame_*.png")):
        m = (rng.random((240, 240)) > 0.3).astype(np.uint8) * 255
        imageio.write_image(masks / p.name, m)
        mimgs[p.name] = m
    xml = tmp_path / "c.xml"
    xml.write_text(SMALL_XML)
    args = ["-i", str(d), "-x", str(xml), "--interpolation", "linear", "--perspective-size", "96", "--perspective-ext", "png",
            "--workers", "2", "--mask-value", "5", "--map-mode", mode]
    if mode == "table":
        args += ["--save-fisheye-output", "--mask-input-dir", str(masks)]
    r = run(args)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[OK ][PERSP]    1/   2 frame_0000 -> 10 views" in r.stdout
    assert r.stdout.splitlines()[-1].startswith("[DONE] processed=4 skipped=0 total=4 persp_outputs=20")
    sensors, _ = fe.load_metashape_calibration(xml)
    specs = fe.sfm10_specs(96, 14.0, "36 36", 40.0, 40.0)
    tables = fe.choose_lens_tables(sensors, "0", "0", specs, 0.0, 180.0, 190.0)
    root = d.resolve().with_name("shots_perspective_colmap")
    for k in range(2):
        for spec in specs:
            vid = spec["view_id"]
            t = tables[vid]
            src = imgs[(k, t["lens_key"])]
            got = imageio.read_image(root / "Images" / f"frame_{k:04d}_{vid}.png")
            if mode == "table":
                want = orc.valid_fill(orc.remap_u8(src, t["map_x"], t["map_y"], interp=1, border_value=(0, 0, 5, 0)), t["valid"], 5)
                assert np.array_equal(got, want), (k, vid)
                gm = imageio.read_image(root / "Masks" / f"frame_{k:04d}_{vid}.png")[:, :, 0]
                wm = orc.valid_fill(orc.remap_u8(mimgs[f"frame_{k:04d}_{t['lens_key']}.png"], t["map_x"], t["map_y"], interp=0,
                                                 border_value=0.0).copy(), t["valid"], 0)
                assert np.array_equal(gm, wm), (k, vid)
            else:   # fused map: <= 0.01 px from the tables -> a small fraction of pixels may differ by a few grey levels
                want = orc.valid_fill(orc.remap_u8(src, t["map_x"], t["map_y"], interp=1, border_value=(0, 0, 5, 0)), t["valid"], 5)
                diff = np.abs(got.astype(int) - want.astype(int))
                assert (diff > 0).mean() < 0.25 and np.percentile(diff, 99.9) <= 40
    if mode == "table":
        und = fe.undistort_tables(sensors["0"], None, 190.0)
        got = imageio.read_image(d.resolve().with_name("shots_undistorted") / "frame_0001_Y.png")
        want = orc.valid_fill(orc.remap_u8(imgs[(1, "Y")], und.map_x, und.map_y, interp=1, border_value=(0, 0, 5, 0)), und.valid_mask, 5)
        assert np.array_equal(got, want)


# ---- --input-lut (.cube colour stage, DF:494-725 / DF:2098-2118 / DF:2453-2458) -------------------------------------
def write_cube(path, n=5):
    g = np.linspace(0.0, 1.0, n)
    rows = ["LUT_3D_SIZE {}".format(n)]
    for b in g:
        for gg in g:
            for r in g:
                rows.append("{:.6f} {:.6f} {:.6f}".format(0.9 * r ** 0.7 + 0.05 * gg, 0.8 * gg + 0.1 * b ** 2, 1.02 * b ** 0.5 - 0.01))
    path.write_text("\n".join(rows) + "\n")


def test_input_lut_flag_errors_and_info_lines(tmp_path):
    d = tmp_path / "shots"
    make_pairs(d, n=1)
    xml = tmp_path / "c.xml"
    xml.write_text(SMALL_XML)
    r = run(["-i", str(d), "-x", str(xml), "--dry-run", "--input-lut", str(tmp_path / "none.cube")])
    assert r.returncode == 1 and "[ERR] Failed to load input LUT: LUT file not found:" in r.stderr
    bad = tmp_path / "bad.cube"
    bad.write_text("LUT_3D_SIZE 2\n0 0 0\n")
    r = run(["-i", str(d), "-x", str(xml), "--dry-run", "--input-lut", str(bad)])
    assert r.returncode == 1 and "[ERR] Failed to load input LUT: LUT row count mismatch" in r.stderr
    cube = tmp_path / "look.cube"
    write_cube(cube)
    r = run(["-i", str(d), "-x", str(xml), "--dry-run", "--input-lut", str(cube), "--lut-output-color-space", "native",
             "--save-color-corrected-output"])
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert f"[INFO] input LUT: {cube.resolve()}" in out and "[INFO] LUT output color space: passthrough" in out
    assert f"[INFO] color-corrected output: {d.resolve().with_name('shots_colorcorrected')}" in out
    assert "[DRY][COLOR]    1/   1 frame_0000_X.png -> frame_0000_X.png" in out
    assert out[-1] == "[DONE] processed=2 skipped=0 total=2 persp_outputs=10 mask_outputs=0 color_outputs=2 errors=0"


@pytest.mark.gpu
@pytest.mark.parametrize("space", ["srgb", "passthrough"])
def test_pair_pipeline_with_input_lut_on_gpu(tmp_path, orc, space):
    """colour stage first, resampling second (DF:1938-1947): outputs equal oracle-remap(oracle-colour(image))"""
    from gs360 import color
    from oracle import color_np
    d = tmp_path / "shots"
    imgs = make_pairs(d, n=1)
    xml = tmp_path / "c.xml"
    xml.write_text(SMALL_XML)
    cube = tmp_path / "look.cube"
    write_cube(cube)
    r = run(["-i", str(d), "-x", str(xml), "--interpolation", "linear", "--perspective-size", "64", "--perspective-ext", "png",
             "--input-lut", str(cube), "--lut-output-color-space", space, "--save-color-corrected-output", "--save-fisheye-output"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[OK ][COLOR]    1/   1 frame_0000_X.png -> frame_0000_X.png" in r.stdout
    assert r.stdout.splitlines()[-1] == "[DONE] processed=2 skipped=0 total=2 persp_outputs=10 mask_outputs=0 color_outputs=2 errors=0"
    lut = color.load_cube_lut(cube)
    conv = {k: color_np.color_pipeline(v, lut.table, lut.domain_min, lut.domain_max, space, red_index=0) for k, v in imgs.items()}
    for lens in "XY":
        got = imageio.read_image(d.resolve().with_name("shots_colorcorrected") / f"frame_0000_{lens}.png")
        assert np.array_equal(got, conv[(0, lens)])
    sensors, _ = fe.load_metashape_calibration(xml)
    specs = fe.sfm10_specs(64, 14.0, "36 36", 40.0, 40.0)
    tables = fe.choose_lens_tables(sensors, "0", "0", specs, 0.0, 180.0, 190.0)
    for spec in specs:
        t = tables[spec["view_id"]]
        got = imageio.read_image(d.resolve().with_name("shots_perspective_colmap") / "Images" / f"frame_0000_{spec['view_id']}.png")
        want = orc.valid_fill(orc.remap_u8(conv[(0, t["lens_key"])], t["map_x"], t["map_y"], interp=1, border_value=0.0), t["valid"], 0)
        assert np.array_equal(got, want), spec["view_id"]
    und = fe.undistort_tables(sensors["0"], None, 190.0)
    got = imageio.read_image(d.resolve().with_name("shots_undistorted") / "frame_0000_X.png")
    assert np.array_equal(got, orc.valid_fill(orc.remap_u8(conv[(0, "X")], und.map_x, und.map_y, interp=1, border_value=0.0), und.valid_mask, 0))


@pytest.mark.gpu
def test_pair_pipeline_16bit_sources_keep_their_depth(tmp_path, orc):
    """16-bit lens images (cv2.imread(IMREAD_UNCHANGED) keeps them uint16, DF:735) leave as 16-bit PNGs, sampled by the CV_16U
    cv2.remap restatement -- cubic, the tool's default interpolation (DF:229-234)"""
    d = tmp_path / "shots"
    d.mkdir()
    rng = np.random.default_rng(15)
    imgs = {}
    for lens in "XY":
        a = rng.integers(0, 65536, (240, 240, 3), dtype=np.uint16)
        imageio.write_image(d / f"frame_0000_{lens}.png", a)
        imgs[lens] = a
    xml = tmp_path / "c.xml"
    xml.write_text(SMALL_XML)
    r = run(["-i", str(d), "-x", str(xml), "--perspective-size", "80", "--perspective-ext", "png", "--mask-value", "9", "--save-fisheye-output"])
    assert r.returncode == 0, r.stdout + r.stderr
    sensors, _ = fe.load_metashape_calibration(xml)
    specs = fe.sfm10_specs(80, 14.0, "36 36", 40.0, 40.0)
    tables = fe.choose_lens_tables(sensors, "0", "0", specs, 0.0, 180.0, 190.0)
    root = d.resolve().with_name("shots_perspective_colmap")
    for spec in specs:
        t = tables[spec["view_id"]]
        got = imageio.read_image(root / "Images" / f"frame_0000_{spec['view_id']}.png")
        want = orc.valid_fill(orc.remap_u16(imgs[t["lens_key"]], t["map_x"], t["map_y"], interp=2, border_value=(0, 0, 9, 0)), t["valid"], 9)
        assert got.dtype == np.uint16 and np.array_equal(got, want), spec["view_id"]
    und = fe.undistort_tables(sensors["0"], None, 190.0)
    got = imageio.read_image(d.resolve().with_name("shots_undistorted") / "frame_0000_Y.png")
    want = orc.valid_fill(orc.remap_u16(imgs["Y"], und.map_x, und.map_y, interp=2, border_value=(0, 0, 9, 0)), und.valid_mask, 9)
    assert got.dtype == np.uint16 and np.array_equal(got, want)


@pytest.mark.gpu
def test_pair_pipeline_16bit_with_input_lut(tmp_path, orc):
    """16-bit lens images through --input-lut (16-bit colour stage) and then the CV_16U samplers: colour first, resampling
    second (DF:1938-1947), everything at native depth"""
    from gs360 import color
    from oracle import color_np
    d = tmp_path / "shots"
    d.mkdir()
    rng = np.random.default_rng(16)
    imgs = {}
    for lens in "XY":
        a = rng.integers(0, 65536, (240, 240, 3), dtype=np.uint16)
        imageio.write_image(d / f"frame_0000_{lens}.png", a)
        imgs[lens] = a
    xml = tmp_path / "c.xml"
    xml.write_text(SMALL_XML)
    cube = tmp_path / "look.cube"
    write_cube(cube)
    r = run(["-i", str(d), "-x", str(xml), "--interpolation", "linear", "--perspective-size", "64", "--perspective-ext", "png",
             "--input-lut", str(cube), "--lut-output-color-space", "srgb", "--save-color-corrected-output"])
    assert r.returncode == 0, r.stdout + r.stderr
    lut = color.load_cube_lut(cube)
    conv = {k: color_np.color_pipeline(v, lut.table, lut.domain_min, lut.domain_max, "srgb", red_index=0) for k, v in imgs.items()}
    for lens in "XY":
        got = imageio.read_image(d.resolve().with_name("shots_colorcorrected") / f"frame_0000_{lens}.png")
        assert got.dtype == np.uint16 and np.array_equal(got, conv[lens])
    sensors, _ = fe.load_metashape_calibration(xml)
    specs = fe.sfm10_specs(64, 14.0, "36 36", 40.0, 40.0)
    tables = fe.choose_lens_tables(sensors, "0", "0", specs, 0.0, 180.0, 190.0)
    for spec in specs:
        t = tables[spec["view_id"]]
        got = imageio.read_image(d.resolve().with_name("shots_perspective_colmap") / "Images" / f"frame_0000_{spec['view_id']}.png")
        want = orc.valid_fill(orc.remap_u16(conv[t["lens_key"]], t["map_x"], t["map_y"], interp=1, border_value=0.0), t["valid"], 0)
        assert np.array_equal(got, want), spec["view_id"]
