"""Host logic of the drop-in: argv round trip (planner -> jobspec), GUI-style argv rewriting, CLI dry-run output,
image codecs."""
import json
import pathlib
import shlex
import subprocess
import sys

import numpy as np
import pytest

import gs360_360PerspCut as cut
from conftest import GOLDEN, PKG
from gs360 import imageio
from gs360.jobspec import JobParseError, parse_job_argv, split_filter_chain

G = json.loads((GOLDEN / "planner_goldens.json").read_text())


@pytest.mark.parametrize("name", sorted(G["cases"]))
def test_every_reference_argv_parses_back(name):
    exp = G["cases"][name]["expect"]
    for (argv, src, dst), spec in zip(exp["jobs"], exp["view_specs"]):
        job = parse_job_argv(argv)
        assert job.src.name == src and job.dst.name == dst
        assert job.width == spec["width"] and job.height == spec["height"]
        assert repr(job.fnum("yaw")) == spec["yaw_deg"] and repr(job.fnum("pitch")) == spec["pitch_deg"]
        if spec["projection"] == "perspective":
            assert job.output_projection == "rectilinear"
            assert repr(job.fnum("h_fov")) == spec["hfov_deg"] and repr(job.fnum("v_fov")) == spec["vfov_deg"]
        else:
            assert job.output_projection == "fisheye" and repr(job.fnum("d_fov")) == spec["hfov_deg"]
        assert job.interp == "cubic" and job.input_projection == "equirect"
        assert job.is_still_image == (not G["cases"][name]["video"])


GUI = json.loads((GOLDEN / "gui_select_goldens.json").read_text())["cases"]


@pytest.mark.parametrize("case", sorted(GUI))
def test_gui_rewritten_argv_still_parses(case):
    """argv lists produced by the reference GUI's own _apply_frame_selection_to_jobs (gs360_GUI.py:19081-19148, captured by
    tests/golden/make_gui_select_goldens.py): select filter with escaped commas, -frame_pts, -copyts, seeks moved behind -i"""
    g = GUI[case]
    for planned, argv in zip(g["planned"], g["rewritten"]):
        job, ref = parse_job_argv(argv), parse_job_argv(planned)
        want = "select='" + "+".join("eq(n\\,{})".format(i) for i in g["indices"]) + "'"
        assert job.filters[0] == want and not any(f.startswith("fps=") for f in job.filters)
        assert job.filter_named("colorspace") == ref.filter_named("colorspace")
        assert "-copyts" in job.flags and job.options.get("-frame_pts") == "1" and "-start_number" not in job.options
        assert job.v360 == ref.v360 and job.dst == ref.dst and job.src == ref.src and not job.is_still_image
        for flag in ("-ss", "-to"):                       # seeks survive, but now on the output side
            assert job.options.get(flag) == ref.options.get(flag)
            if flag in job.options:
                assert flag in job.output_options and flag not in job.input_options


def test_filter_chain_splitter():
    assert split_filter_chain("fps=2.0,v360=a=b:c=d") == ["fps=2.0", "v360=a=b:c=d"]
    assert split_filter_chain("select='a,b',x") == ["select='a,b'", "x"]
    assert split_filter_chain("select=eq(n\\,1),x") == ["select=eq(n\\,1)", "x"]


@pytest.mark.parametrize("argv", [[], ["ffmpeg", "out.jpg"], ["ffmpeg", "-vf", "v360=input=equirect", "-y", "o.jpg"],
                                  ["ffmpeg", "-i", "a.png", "-threads", "1", "o.jpg"],
                                  ["ffmpeg", "-i", "a.png", "-vf", "scale=2:2", "o.jpg"],
                                  ["ffmpeg", "-i", "a.png", "-vf", "v360=input", "o.jpg"]])
def test_malformed_argv_is_rejected(argv):
    with pytest.raises(JobParseError):
        parse_job_argv(argv)


def test_cli_dry_run_matches_reference_plan(tmp_path):
    """`python gs360_360PerspCut.py -i DIR --preset ... --dry-run` prints `$ <argv>` lines + the [DRY] footer
    (reference PC:1023-1027); argv tokens equal the reference's with the paths substituted."""
    in_dir = tmp_path / "in"
    in_dir.mkdir()
    (in_dir / "pano_0001.png").write_bytes(b"")
    (in_dir / "pano_0002.jpg").write_bytes(b"")
    (in_dir / "notes.txt").write_bytes(b"")
    out_dir = tmp_path / "out"
    case = G["cases"]["full360coverage"]
    res = subprocess.run([sys.executable, str(PKG / "cli_tools" / "gs360_360PerspCut.py"), "-i", str(in_dir), "-o", str(out_dir),
                          "--dry-run"] + case["argv"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.splitlines()
    want = []
    for argv, _s, _d in case["expect"]["jobs"]:
        toks = [t.replace("/data/in", str(in_dir.resolve())).replace("/data/out", str(out_dir.resolve())) for t in argv]
        want.append("$ " + " ".join(shlex.quote(t) for t in toks))
    assert lines[:len(want)] == want
    assert lines[len(want):] == ["", f"[DRY] Exiting without execution (total {len(want)} commands)"]
    assert out_dir.is_dir()


def test_cli_missing_input_and_video_without_fps(tmp_path):
    exe = [sys.executable, str(PKG / "cli_tools" / "gs360_360PerspCut.py")]
    r = subprocess.run(exe + ["-i", str(tmp_path / "nope")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "[ERR] Input path not found:" in r.stderr
    vid = tmp_path / "clip.mp4"
    vid.write_bytes(b"")
    r = subprocess.run(exe + ["-i", str(vid)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "[ERR] -f/--fps must be specified for video inputs" in r.stderr
    empty = tmp_path / "empty"
    empty.mkdir()
    r = subprocess.run(exe + ["-i", str(empty)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "[WARN] No target images found" in r.stderr
    r = subprocess.run(exe + ["-i", str(vid), "-f", "2", "--dry-run", "--count", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and f"{tmp_path.resolve()}/clip_geometry/clip_%07d_A.jpg" in r.stdout   # PC:1008-1010


def test_png_codec_roundtrip_builtin_and_pillow(tmp_path):
    rng = np.random.default_rng(1)
    for ch in (1, 3, 4):
        a = rng.integers(0, 256, (17, 23, ch), dtype=np.uint8)
        p = tmp_path / f"a{ch}.png"
        imageio._png_write(p, a)
        assert np.array_equal(imageio._png_read(p.read_bytes()), a)
        assert np.array_equal(imageio.read_image(p), a)
        q = tmp_path / f"b{ch}.png"
        imageio.write_image(q, a)
        assert np.array_equal(imageio._png_read(q.read_bytes()), a)     # built-in reader handles Pillow's filters
    a = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    imageio.write_image(tmp_path / "x.jpg", a, jpeg_q=1)
    assert imageio.read_image(tmp_path / "x.jpg").shape == (16, 16, 3)
    imageio.write_image(tmp_path / "x.tif", a)
    assert np.array_equal(imageio.read_image(tmp_path / "x.tif"), a)
