#!/usr/bin/env python3
"""Pin the equirect pixel convention on the reference's OWN functions (round-4 VERDICT item 2).

Runs only in the build container (needs /root/reference).  `gs360_GUI.py` cannot be imported (tkinter), but its geometry helpers
-- normalize_vector, rotate_pitch, rotate_yaw, direction_from_uv, lonlat_to_xy (gs360_GUI.py:342-424) -- are pure `math`
functions: this script finds their FunctionDef nodes with `ast`, compiles THOSE NODES alone (no source text is stored or copied),
and evaluates them on a strided grid of pixel centres of every perspective view the reference PLANNER (imported, as in
make_planner_goldens.py) emits for `default`, `fisheyelike`, `full360coverage` and `--count 6 --size 800`, for 7680x3840 and
5760x2880 panoramas.  Output is data only: per view its parameters and, per sample, (i, j, lon, lat, x, y) in float64.

    python tests/golden/make_eq_convention_goldens.py     # rewrites eq_convention_goldens.npz / .json
"""
import ast
import json
import math
import pathlib
import sys
import typing

import numpy as np

sys.dont_write_bytecode = True
REF = pathlib.Path("/root/reference")
sys.path.insert(0, str(REF / "cli_tools"))

import gs360_360PerspCut as ref  # noqa: E402  (reference; container-only)

HERE = pathlib.Path(__file__).resolve().parent
NAMES = ("normalize_vector", "rotate_pitch", "rotate_yaw", "direction_from_uv", "lonlat_to_xy")


def lift_functions(path: pathlib.Path, names):
    tree = ast.parse(path.read_text(encoding="utf-8"))
    nodes = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names}
    missing = [n for n in names if n not in nodes]
    if missing:
        raise SystemExit(f"{missing} not found in {path}")
    mod = ast.Module(body=[nodes[n] for n in names], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {k: getattr(typing, k) for k in ("List", "Tuple", "Sequence", "Optional", "Dict", "Any", "Iterable")}
    ns["math"] = math
    exec(compile(mod, str(path), "exec"), ns)      # noqa: S102  (the reference's own functions, container-only)
    return ns, {n: (nodes[n].lineno, nodes[n].end_lineno) for n in names}


def plan(extra):
    args = ref.create_arg_parser().parse_args(["-i", "/in/pano.png"] + extra)
    for attr in ("size", "hfov", "focal_mm"):
        setattr(args, f"{attr}_explicit", getattr(args, f"{attr}_explicit", False))
    args.input_is_video, args.video_bit_depth = False, 8
    return ref.build_view_jobs(args, [pathlib.Path("/in/pano.png")], pathlib.Path("/out"))


CASES = [("default", []), ("fisheyelike", ["--preset", "fisheyelike"]), ("full360coverage", ["--preset", "full360coverage"]),
         ("count6_size800", ["--count", "6", "--size", "800"])]
PANOS = [(7680, 3840), (5760, 2880)]
GRID = 13


def main():
    ns, lines = lift_functions(REF / "gs360_GUI.py", NAMES)
    direction_from_uv, lonlat_to_xy = ns["direction_from_uv"], ns["lonlat_to_xy"]
    arrays, meta = {}, {"source": {n: f"gs360_GUI.py:{lo}-{hi}" for n, (lo, hi) in lines.items()},
                        "note": "u = (i + 1/2) / w * 2 - 1, v = (j + 1/2) / h * 2 - 1 (pixel centres); fov clamped to [1e-3, 179.9] deg as gs360_GUI.py:437-438",
                        "views": []}
    for cname, extra in CASES:
        specs = [s for s in plan(extra).view_specs if s.projection == "perspective"]
        for k, s in enumerate(specs):
            w, h = int(s.width), int(s.height)
            ii = sorted(set([0, 1, w // 2 - 1, w // 2, w - 2, w - 1] + [int(round(t)) for t in np.linspace(0, w - 1, GRID)]))
            jj = sorted(set([0, 1, h // 2 - 1, h // 2, h - 2, h - 1] + [int(round(t)) for t in np.linspace(0, h - 1, GRID)]))
            hf = math.radians(min(max(s.hfov_deg, 1e-3), 179.9))
            vf = math.radians(min(max(s.vfov_deg, 1e-3), 179.9))
            yaw, pitch = math.radians(s.yaw_deg), math.radians(s.pitch_deg)
            for W, H in PANOS:
                rows = []
                for j in jj:
                    for i in ii:
                        u = ((i + 0.5) / w) * 2.0 - 1.0
                        v = ((j + 0.5) / h) * 2.0 - 1.0
                        lon, lat = direction_from_uv(u, v, hf, vf, yaw, pitch)
                        x, y = lonlat_to_xy(lon, lat, W, H)
                        rows.append((i, j, lon, lat, x, y))
                key = f"{cname}/{k}/{W}x{H}"
                arrays[key] = np.array(rows, np.float64)
                meta["views"].append({"key": key, "case": cname, "view_id": s.view_id, "yaw_deg": s.yaw_deg, "pitch_deg": s.pitch_deg,
                                      "hfov_deg": s.hfov_deg, "vfov_deg": s.vfov_deg, "width": w, "height": h, "W": W, "H": H})
    np.savez_compressed(HERE / "eq_convention_goldens.npz", **arrays)
    (HERE / "eq_convention_goldens.json").write_text(json.dumps(meta, indent=1) + "\n")
    print("wrote", len(arrays), "views,", sum(len(a) for a in arrays.values()), "samples;", meta["source"])


if __name__ == "__main__":
    main()
