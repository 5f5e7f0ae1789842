"""Planner parity: the drop-in gs360_360PerspCut must reproduce the reference planner's outputs exactly.

Goldens were captured by importing the reference (tests/golden/make_planner_goldens.py)."""
import contextlib
import io
import json
import pathlib

import pytest

import gs360_360PerspCut as cut
from conftest import GOLDEN

G = json.loads((GOLDEN / "planner_goldens.json").read_text())


def _run(case):
    parser = cut.create_arg_parser()
    if case["video"]:
        src = [pathlib.Path("/data/clip.mp4")]
        args = parser.parse_args(["-i", str(src[0])] + case["argv"])
    else:
        src = [pathlib.Path("/data/in/pano_0001.png"), pathlib.Path("/data/in/pano_0002.jpg")]
        args = parser.parse_args(["-i", "/data/in"] + case["argv"])
    for attr in ("size", "hfov", "focal_mm"):
        setattr(args, attr + "_explicit", getattr(args, attr + "_explicit", False))
    args.input_is_video = case["video"]
    args.video_bit_depth = case["bit_depth"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = cut.build_view_jobs(args, src, pathlib.Path("/data/out"))
    return args, res, buf.getvalue()


@pytest.mark.parametrize("name", sorted(G["cases"]))
def test_case_matches_reference(name):
    case = G["cases"][name]
    exp = case["expect"]
    args, res, out = _run(case)
    assert out == exp["stdout"]
    assert [[list(c), s, d] for c, s, d in res.jobs] == exp["jobs"]
    got_specs = [{"source_path": str(v.source_path), "output_name": v.output_name, "view_id": v.view_id,
                  "yaw_deg": repr(v.yaw_deg), "pitch_deg": repr(v.pitch_deg), "hfov_deg": repr(v.hfov_deg),
                  "vfov_deg": repr(v.vfov_deg), "width": v.width, "height": v.height, "projection": v.projection}
                 for v in res.view_specs]
    assert got_specs == exp["view_specs"]
    assert repr(res.focal_used_mm) == exp["focal_used_mm"]
    assert repr(res.focal_35mm_equiv) == exp["focal_35mm_equiv"]
    assert repr(res.hfov_deg) == exp["hfov_deg"] and repr(res.vfov_deg) == exp["vfov_deg"]
    for k in ("preview_views_line", "sensor_line", "realityscan_line", "metashape_line"):
        assert getattr(res, k) == exp[k], k
    assert res.total == len(exp["jobs"])
    after = exp["args_after"]
    assert (args.count, args.size, repr(args.focal_mm), args.add_top, args.add_bottom) == (
        after["count"], after["size"], after["focal_mm"], after["add_top"], after["add_bottom"])


@pytest.mark.parametrize("name", sorted(G["errors"]))
def test_error_cases(name):
    case = {"argv": G["errors"][name]["argv"], "video": False, "bit_depth": 8}
    assert G["errors"][name]["raises"] == "ValueError"
    with pytest.raises(ValueError):
        _run(case)


def test_count_zero_exits_1(capsys):
    with pytest.raises(SystemExit) as e:
        _run({"argv": ["--count", "0"], "video": False, "bit_depth": 8})
    assert e.value.code == 1
    assert "[ERR] --count must be >= 1" in capsys.readouterr().err


def test_helpers_match_reference():
    h = G["helpers"]
    for f, s, want in h["fov_from_focal_mm"]:
        assert repr(cut.fov_from_focal_mm(f, s)) == want
    for a, s, want in h["focal_from_hfov_deg"]:
        assert repr(cut.focal_from_hfov_deg(a, s)) == want
    for a, w, hh, want in h["v_fov_from_hfov"]:
        assert repr(cut.v_fov_from_hfov(a, w, hh)) == want
    for a, want in h["normalize_angle_deg"]:
        assert repr(cut.normalize_angle_deg(a)) == want
    for i, want in h["letter_tag"]:
        assert cut.letter_tag(i) == want
    for d, dd, want in h["extra_suffix"]:
        assert cut.extra_suffix(d, dd) == want
    for s, want in h["parse_jobs"]:
        assert cut.parse_jobs(s) == want
    for s, want in h["parse_sensor"]:
        assert repr(cut.parse_sensor(s)) == want
    assert sorted(cut.EXTS) == h["EXTS"]
    assert cut.PROGRESS_INTERVAL == h["PROGRESS_INTERVAL"]


def test_parser_defaults_superset_of_reference():
    ours = {k: (repr(v) if isinstance(v, float) else v)
            for k, v in vars(cut.create_arg_parser().parse_args(["-i", "x"])).items()}
    for k, v in G["helpers"]["parser_defaults"].items():
        assert ours[k] == v, k
    assert set(ours) - set(G["helpers"]["parser_defaults"]) == {"engine"}   # the only additive flag


def test_module_surface_used_by_gui():
    for name in ("create_arg_parser", "ViewSpec", "BuildResult", "build_view_jobs", "run_one", "stop_event",
                 "procs_lock", "running_procs", "parse_jobs", "detect_input_bit_depth", "EXTS", "PROGRESS_INTERVAL",
                 "fov_from_focal_mm", "v_fov_from_hfov", "StoreWithFlag", "main"):
        assert hasattr(cut, name), name


def test_run_one_early_out_when_cancelled():
    cut.stop_event.set()
    try:
        assert cut.run_one(["ffmpeg", "-i", "x", "-vf", "v360=input=equirect", "y.jpg"]) == (130, "")
    finally:
        cut.stop_event.clear()
