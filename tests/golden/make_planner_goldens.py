#!/usr/bin/env python3
"""Capture planner golden vectors by IMPORTING the reference planner.

Runs only in the build container (needs /root/reference).  The reference module
cli_tools/gs360_360PerspCut.py is stdlib-only, so it imports as-is.  Output is
data only: for each flag combination the view table, the job argv lists, and the
four info lines produced by the reference's build_view_jobs (PC:593-980).

    python tests/golden/make_planner_goldens.py      # rewrites planner_goldens.json
"""
import io
import json
import pathlib
import sys
import contextlib

sys.dont_write_bytecode = True
REF = pathlib.Path("/root/reference/cli_tools")
sys.path.insert(0, str(REF))

import gs360_360PerspCut as ref  # noqa: E402  (reference; container-only)

HERE = pathlib.Path(__file__).resolve().parent

# Each case: (name, argv after "-i <in>", video?, bit_depth)
CASES = [
    ("default", [], False, 8),
    ("default_png", ["--ext", "png"], False, 8),
    ("cfg2_count6_size800", ["--preset", "default", "--count", "6", "--size", "800"], False, 8),
    ("fisheyelike", ["--preset", "fisheyelike"], False, 8),
    ("fisheyelike_2048", ["--preset", "fisheyelike", "--size", "2048"], False, 8),
    ("full360coverage", ["--preset", "full360coverage"], False, 8),
    ("2views", ["--preset", "2views"], False, 8),
    ("2views_size_explicit", ["--preset", "2views", "--size", "1000", "--focal-mm", "9"], False, 8),
    ("evenMinus30", ["--preset", "evenMinus30"], False, 8),
    ("evenPlus30", ["--preset", "evenPlus30", "--ext", "tif"], False, 8),
    ("fisheyeXY", ["--preset", "fisheyeXY"], False, 8),
    ("fisheyeXY_hfov", ["--preset", "fisheyeXY", "--hfov", "170", "--size", "1200"], False, 8),
    ("cube105", ["--count", "4", "--hfov", "105", "--add-top", "--add-bottom"], False, 8),
    ("topdown_hidden", ["--count", "4", "--add-topdown", "--sensor-mm", "36x24"], False, 8),
    ("addcam_mix", ["--addcam", "B,D:U20,F:D", "--delcam", "C", "--addcam-deg", "25"], False, 8),
    ("setcam_abs_rel", ["--setcam", "A=U10,B:+5,C=-12.5,D=D", "--addcam", "A", "--size", "640"], False, 8),
    ("setcam_extra", ["--addcam", "A,B:U15", "--setcam", "A_U:+5,A_D=-45,B_U15=20"], False, 8),
    ("full360_user_addcam", ["--preset", "full360coverage", "--addcam", "A:U"], False, 8),
    ("fisheyelike_user_delcam", ["--preset", "fisheyelike", "--delcam", "B"], False, 8),
    ("sensor_36x24_focal", ["--sensor-mm", "36x24", "--focal-mm", "18", "--size", "1024"], False, 8),
    ("sensor_apsc", ["--sensor-mm", "23.5 15.6", "--focal-mm", "10"], False, 8),
    ("jpeg95", ["--jpeg-quality-95", "--count", "3"], False, 8),
    ("count30", ["--count", "30", "--size", "256"], False, 8),
    ("video_jpg", ["-f", "2", "--preset", "full360coverage"], True, 8),
    ("video_png_10bit", ["-f", "1.5", "--ext", "png", "--start", "3", "--end", "12.5"], True, 10),
    ("video_tif_keep709", ["-f", "5", "--ext", "tif", "--keep-rec709", "--count", "4"], True, 8),
    ("video_jpg95_start", ["-f", "0.5", "--jpeg-quality-95", "--start", "-2", "--count", "2"], True, 8),
]

ERROR_CASES = [
    ("addcam_bad_token", ["--addcam", "B:+10"]),
    ("setcam_bar_sign", ["--setcam", "A:|5"]),
    ("setcam_no_sep", ["--setcam", "A"]),
    ("delcam_bad_key", ["--delcam", "?"]),
]


def run_case(argv, video, depth):
    in_dir = pathlib.Path("/data/in")
    out_dir = pathlib.Path("/data/out")
    parser = ref.create_arg_parser()
    if video:
        src = [pathlib.Path("/data/clip.mp4")]
        args = parser.parse_args(["-i", str(src[0])] + argv)
    else:
        src = [in_dir / "pano_0001.png", in_dir / "pano_0002.jpg"]
        args = parser.parse_args(["-i", str(in_dir)] + argv)
    for attr in ("size", "hfov", "focal_mm"):
        setattr(args, attr + "_explicit", getattr(args, attr + "_explicit", False))
    args.input_is_video = video
    args.video_bit_depth = depth
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = ref.build_view_jobs(args, src, out_dir)
    return {
        "stdout": buf.getvalue(),
        "jobs": [[list(cmd), s, d] for cmd, s, d in res.jobs],
        "view_specs": [
            {
                "source_path": str(v.source_path),
                "output_name": v.output_name,
                "view_id": v.view_id,
                "yaw_deg": repr(v.yaw_deg),
                "pitch_deg": repr(v.pitch_deg),
                "hfov_deg": repr(v.hfov_deg),
                "vfov_deg": repr(v.vfov_deg),
                "width": v.width,
                "height": v.height,
                "projection": v.projection,
            }
            for v in res.view_specs
        ],
        "focal_used_mm": repr(res.focal_used_mm),
        "focal_35mm_equiv": repr(res.focal_35mm_equiv),
        "hfov_deg": repr(res.hfov_deg),
        "vfov_deg": repr(res.vfov_deg),
        "preview_views_line": res.preview_views_line,
        "sensor_line": res.sensor_line,
        "realityscan_line": res.realityscan_line,
        "metashape_line": res.metashape_line,
        "args_after": {
            "count": args.count,
            "size": args.size,
            "focal_mm": repr(args.focal_mm),
            "add_top": args.add_top,
            "add_bottom": args.add_bottom,
        },
    }


def main():
    out = {"_meta": {"source": "reference cli_tools/gs360_360PerspCut.py build_view_jobs",
                     "python": sys.version.split()[0]},
           "cases": {}, "errors": {}, "helpers": {}}
    for name, argv, video, depth in CASES:
        out["cases"][name] = {"argv": argv, "video": video, "bit_depth": depth,
                              "expect": run_case(argv, video, depth)}
    for name, argv in ERROR_CASES:
        try:
            run_case(argv, False, 8)
            out["errors"][name] = {"argv": argv, "raises": None}
        except Exception as exc:  # noqa: BLE001
            out["errors"][name] = {"argv": argv, "raises": type(exc).__name__}
    h = out["helpers"]
    h["fov_from_focal_mm"] = [[f, s, repr(ref.fov_from_focal_mm(f, s))]
                              for f, s in [(12.0, 36.0), (17.0, 36.0), (14.0, 36.0), (6.0, 36.0), (10.0, 23.5)]]
    h["focal_from_hfov_deg"] = [[a, s, repr(ref.focal_from_hfov_deg(a, s))]
                                for a, s in [(105.0, 36.0), (90.0, 36.0), (60.0, 24.0)]]
    h["v_fov_from_hfov"] = [[a, w, hh, repr(ref.v_fov_from_hfov(a, w, hh))]
                            for a, w, hh in [(90.0, 1600, 1600), (100.0, 1920, 1080), (60.0, 800, 1200)]]
    h["normalize_angle_deg"] = [[a, repr(ref.normalize_angle_deg(a))]
                                for a in [0.0, 180.0, -180.0, 540.0, 225.0, 359.999999, -179.9999999, 720.5]]
    h["letter_tag"] = [[i, ref.letter_tag(i)] for i in [0, 7, 25, 26, 29]]
    h["extra_suffix"] = [[d, dd, ref.extra_suffix(d, dd)]
                         for d, dd in [(30.0, 30.0), (-30.0, 30.0), (20.0, 30.0), (-12.5, 30.0), (25.0, 25.0)]]
    h["parse_jobs"] = [[s, ref.parse_jobs(s)] for s in ["1", "7", "0", "-3"]]
    h["parse_sensor"] = [[s, repr(ref.parse_sensor(s))] for s in ["36 36", "36x24", "23.5, 15.6", "36×24"]]
    h["EXTS"] = sorted(ref.EXTS)
    h["PROGRESS_INTERVAL"] = ref.PROGRESS_INTERVAL
    h["parser_defaults"] = {k: (repr(v) if isinstance(v, float) else v)
                            for k, v in vars(ref.create_arg_parser().parse_args(["-i", "x"])).items()}
    (HERE / "planner_goldens.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("wrote", HERE / "planner_goldens.json", "cases:", len(out["cases"]))


if __name__ == "__main__":
    main()
