#!/usr/bin/env python3
"""Capture what the unmodified GUI does to a planned video argv for a CSV frame selection (round-1 VERDICT item 7).

Runs only in the build container (needs /root/reference).  `gs360_GUI.py` cannot be imported here (tkinter, PIL.ImageTk),
but `_apply_frame_selection_to_jobs` (gs360_GUI.py:19081-19148) is pure list / str manipulation that never touches `self`:
this script finds the method's FunctionDef with `ast`, compiles THAT NODE alone (no source text is stored or copied) and
calls it on argv lists planned by the reference's own planner (imported as in make_planner_goldens.py).  Output is data
only: the planned argv, the selected indices and the rewritten argv.

    python tests/golden/make_gui_select_goldens.py     # rewrites gui_select_goldens.json
"""
import ast
import json
import pathlib
import sys
import typing

sys.dont_write_bytecode = True
REF = pathlib.Path("/root/reference")
sys.path.insert(0, str(REF / "cli_tools"))

import gs360_360PerspCut as ref  # noqa: E402  (reference; container-only)

HERE = pathlib.Path(__file__).resolve().parent


def lift_method(path: pathlib.Path, name: str):
    tree = ast.parse(path.read_text(encoding="utf-8"))
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ast.fix_missing_locations(mod)
            ns = {k: getattr(typing, k) for k in ("List", "Tuple", "Sequence", "Optional", "Dict", "Any")}
            exec(compile(mod, str(path), "exec"), ns)      # noqa: S102  (the reference's own function, container-only)
            return ns[name], node.lineno, node.end_lineno
    raise SystemExit(f"{name} not found in {path}")


def plan(extra):
    args = ref.create_arg_parser().parse_args(["-i", "/videos/clip.mp4"] + extra)
    for attr in ("size", "hfov", "focal_mm"):
        setattr(args, f"{attr}_explicit", getattr(args, f"{attr}_explicit", False))
    args.input_is_video, args.video_bit_depth = True, 8
    return ref.build_view_jobs(args, [pathlib.Path("/videos/clip.mp4")], pathlib.Path("/out"))


CASES = [
    ("png_fps1_count2", ["-f", "1", "--ext", "png", "--count", "2"], [4, 0, 2]),
    ("jpg_fps1_count2", ["-f", "1", "--count", "2"], [40, 7, 19]),
    ("png_seek_both", ["-f", "2", "--ext", "png", "--count", "4", "--start", "3", "--end", "12.5"], [1, 4, 7, 20]),
    ("tif_seek_start_full360", ["-f", "5", "--ext", "tif", "--preset", "full360coverage", "--start", "1.5"], [3, 17, 40]),
    ("jpg95_single_index", ["-f", "0.5", "--jpeg-quality-95", "--count", "3"], [0]),
]


def main():
    fn, lo, hi = lift_method(REF / "gs360_GUI.py", "_apply_frame_selection_to_jobs")
    out = {"source": f"gs360_GUI.py:{lo}-{hi} (_apply_frame_selection_to_jobs), planner gs360_360PerspCut.build_view_jobs", "cases": {}}
    for name, extra, indices in CASES:
        jobs = plan(extra).jobs
        rewritten = fn(None, [(list(c), s, d) for c, s, d in jobs], indices)
        out["cases"][name] = {"cli": extra, "indices": indices,
                              "planned": [list(c) for c, _s, _d in jobs],
                              "rewritten": [list(c) for c, _s, _d in rewritten]}
    (HERE / "gui_select_goldens.json").write_text(json.dumps(out, indent=1) + "\n")
    print("wrote", HERE / "gui_select_goldens.json", {k: len(v["planned"]) for k, v in out["cases"].items()})


if __name__ == "__main__":
    main()
