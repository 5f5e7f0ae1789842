#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""gs360_DualFisheyeDistortionCalibration -- MI355X drop-in for the reference tool of the same name.

The GUI drives this tool as a subprocess (reference gs360_GUI.py:9971-10147), so the contract is the command
line: the same flags and defaults (reference cli_tools/gs360_DualFisheyeDistortionCalibration.py:124-450), the same
log prefixes ([INFO] / [SKIP] / [DRY] / [OK ][PERSP] / [DONE] ...), the same exit codes (0, 1 = usage, 2 = errors)
and the same output layout (<dir>_perspective_colmap/Images|Masks, <dir>_undistorted).

What changed underneath: the per-pair `cv2.remap` calls (reference :2001-2014, :2031-2043, :1198-1212) run on the
GPU through libgs360hip.so.  Default `--map-mode table` samples the reference-identical NumPy tables
(gs360/fisheye.py) with a restatement of cv2's arithmetic (8-bit fixed point; float weights for 16-bit images, which
keep their depth) -> bit-identical to the CPU checker's restatement of cv2.remap for all four interpolations (against a
real cv2 the parity is unpinned: none exists in the build or test images, tests/test_crosscheck_external.py runs where
one does); `--map-mode fused` is the MEMORY-SAVING mode: the map is evaluated in-kernel (8-bit images), no 8 B/px tables are built
or kept (165 MB for ten 1750^2 views), at the price of arithmetic -- it is not faster than table mode (82 vs 77 us per pair of six
1750^2 views) and follows the reference's float32 maps only to 0.01 px (median 0-1 ULP).  `--input-lut` (.cube 3D LUT + optional Rec.709 -> sRGB re-encode,
reference :494-725) also runs on the GPU, on the uploaded lens images before any resampling (8- and 16-bit images).
Not built here (outside the pixel path): the COLMAP / Metashape metadata export -- the flags are
accepted, and asking for that stage is reported as an error instead of being silently skipped.
"""
import argparse
import os
import pathlib
import sys
from concurrent.futures import FIRST_COMPLETED, ThreadPoolExecutor, wait
from typing import Dict, List, Optional, Sequence, Set, Tuple

_HERE = pathlib.Path(__file__).resolve().parent
if str(_HERE.parent) not in sys.path:
    sys.path.insert(0, str(_HERE.parent))

from gs360 import color  # noqa: E402
from gs360 import fisheye as fe  # noqa: E402
from gs360.dualfisheye import INTERPOLATIONS  # noqa: E402

SUPPORTED_EXTS = (".jpg", ".jpeg", ".png", ".tif", ".tiff")
SUPPORTED_MODELS = fe.SUPPORTED_MODELS
SCRIPT_DIR = _HERE
DEFAULT_CAMERA_XML = SCRIPT_DIR / "templates" / "Osmo360-Fisheye-Distortion.xml"
DEFAULT_DLOGM_LUT = SCRIPT_DIR / "templates" / "DJI Osmo 360 D-Log M to Rec.709 V1.cube"
# The reference defaults to one worker per CPU core (DF:2680-2701: its workers sit in cv2, outside the interpreter lock).  Here a worker
# is a thread that decodes two images, queues ten GPU views and encodes them, and os.cpu_count() ignores a container's CPU quota (the
# MI355X boxes of the build pool: 256 hardware threads, 16 CPUs allowed; 48 pairs run at 22 pairs/s with 12-16 workers, 17 with 32,
# 14.7 with 256).  Default: the CPUs the process may actually use, at most 32; --workers N is taken as given.
def _default_workers() -> int:
    from gs360 import hostmem
    return max(1, min(32, hostmem.effective_cpus()))


DEFAULT_WORKERS = _default_workers()
DEFAULT_PERSPECTIVE_METASHAPE_XML_NAME = "perspective_cams.xml"
INTERPOLATION_MAP = dict(INTERPOLATIONS)


# (flags, keywords): option names / destinations / types / defaults are the reference's command line (DF:124-450) -- the
# GUI composes exactly these flags (gs360_GUI.py:9971-10147); the help wording is ours.  H = hidden legacy option.
H = argparse.SUPPRESS
_OPTIONS = (
    (("-i", "--input-dir"), dict(required=False, help="folder with the *_X / *_Y fisheye stills (not needed with --metadata-only)")),
    (("--metadata-only",), dict(action="store_true", help="export camera metadata only (not available in this build)")),
    (("-x", "--camera-xml"), dict(default=str(DEFAULT_CAMERA_XML), help="Metashape XML holding the lens calibration")),
    (("-o", "--output-dir"), dict(default=None, help="folder for undistorted fisheye images (default <input>_undistorted)")),
    (("--suffixes",), dict(default="_X,_Y", help="stem suffixes of the two lenses, comma separated")),
    (("--ext",), dict(default="jpg,jpeg,png,tif,tiff", help="input extensions to pick up, comma separated")),
    (("--input-lut",), dict(default=None, help=".cube 3D LUT applied to the lens images on the GPU before resampling (8- and 16-bit images)")),
    (("--lut-output-color-space",), dict(metavar="{passthrough,srgb}", default="srgb", help="colour space written after the LUT")),
    (("--input-color-profile",), dict(choices=("native", "osmo360-dlogm"), default="native", help=H)),
    (("--dlogm-lut",), dict(default=str(DEFAULT_DLOGM_LUT), help=H)),
    (("--sensor-id-x",), dict(default=None, help="force this sensor id for the X lens")),
    (("--sensor-id-y",), dict(default=None, help="force this sensor id for the Y lens")),
    (("--interpolation",), dict(choices=tuple(INTERPOLATION_MAP.keys()), default="cubic",
                                help="resampling kernel (GPU cost per pair of 6 x 1750^2 views on 8-bit RGB: nearest / linear ~0.08 ms, "
                                     "cubic ~0.11 ms, lanczos4 ~0.36 ms)")),
    (("--undistort-zoom",), dict(default="auto", help="zoom of the undistorted fisheye output: a positive number or 'auto'")),
    (("--mask-outside-model",), dict(dest="mask_outside_model", action="store_true", help="paint pixels outside the lens model with --mask-value")),
    (("--no-mask-outside-model",), dict(dest="mask_outside_model", action="store_false", help="leave pixels outside the lens model as sampled")),
    (("--mask-value",), dict(type=int, default=0, help="grey level 0-255 used for masked pixels and the resampling border")),
    (("--limit",), dict(type=int, default=0, help=H)),
    (("--workers",), dict(type=int, default=DEFAULT_WORKERS, help="pairs processed concurrently (default: usable CPUs, at most 32 = {})".format(DEFAULT_WORKERS))),
    (("--memory-throttle-percent",), dict(type=float, default=80.0, help="accepted for compatibility (host-memory throttle threshold)")),
    (("--dry-run",), dict(action="store_true", help="list what would be written and stop")),
    (("--report-json",), dict(default=None, help=H)),
    (("--no-perspective",), dict(action="store_true", help="skip the perspective views")),
    (("--save-fisheye-output",), dict(action="store_true", help="also write undistorted fisheye images")),
    (("--save-color-corrected-output",), dict(action="store_true", help="also write the inputs after the colour stage only")),
    (("--color-corrected-output-dir",), dict(default=None, help="folder for those (default <input>_colorcorrected)")),
    (("--fisheye-output-dir",), dict(default=None, help=H)),
    (("--no-fisheye-output",), dict(action="store_true", help=H)),
    (("--perspective-output-dir",), dict(default=None, help="COLMAP-style root for the views (default <input>_perspective_colmap)")),
    (("--perspective-ext",), dict(default="jpg", help="file type of the views")),
    (("--perspective-mask-ext",), dict(default="png", help="file type of the cut masks")),
    (("--perspective-size",), dict(type=int, default=1750, help="edge length of the square views")),
    (("--perspective-focal-mm",), dict(type=float, default=14.0, help="focal length of the views in mm")),
    (("--perspective-sensor-mm",), dict(default="36 36", help="virtual sensor size of the views in mm")),
    (("--perspective-yaw-delta-deg",), dict(type=float, default=40.0, help="yaw offset of the side views")),
    (("--perspective-pitch-delta-deg",), dict(type=float, default=40.0, help="pitch offset of the up/down views")),
    (("--perspective-jpeg-quality",), dict(type=int, default=95, help="JPEG quality of the views")),
    (("--lens-fov-deg",), dict(type=float, default=190.0, help="usable field of view of each fisheye lens")),
    (("--lens-x-yaw-deg",), dict(type=float, default=0.0, help="rig yaw of the X lens")),
    (("--lens-y-yaw-deg",), dict(type=float, default=180.0, help="rig yaw of the Y lens")),
    (("--camera-extrinsics-xml",), dict(default=None, help="Metashape alignment XML (enables metadata export in the reference tool)")),
    (("--pointcloud-ply",), dict(default=None, help="Metashape point cloud used by the metadata export")),
    (("--mask-input-dir",), dict(default=None, help="folder with masks named like the inputs; they are cut with the same maps")),
    (("--perspective-metashape-xml-name",), dict(default=DEFAULT_PERSPECTIVE_METASHAPE_XML_NAME, help="name of the exported camera XML")),
    # additive
    (("--map-mode",), dict(choices=("table", "fused"), default="table",
                           help="table (default, the parity mode) = reference-identical NumPy remap tables sampled on the GPU; "
                                "fused = memory-saving mode: map evaluated in-kernel, no tables kept, within 0.01 px of the reference maps, "
                                "not faster")),
)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Dual-fisheye pairs -> SFM10 perspective views (and optional undistorted fisheyes) "
                                            "on the GPU; drop-in for the cv2-based tool.")
    for flags, kw in _OPTIONS:
        p.add_argument(*flags, **kw)
    p.set_defaults(mask_outside_model=True)
    return p


def parse_arguments() -> argparse.Namespace:
    return build_parser().parse_args()


def parse_undistort_zoom_arg(value: str) -> Optional[float]:
    text = (value or "").strip().lower()
    if not text or text == "auto":
        return None
    zoom = float(text)
    if zoom <= 0.0:
        raise ValueError("undistort zoom must be > 0")
    return zoom


normalize_lut_output_color_space = color.normalize_lut_output_color_space
load_cube_lut = color.load_cube_lut


# ---- pair discovery (reference :831-914) ---------------------------------------------------------------
def gather_input_images(input_dir: pathlib.Path, ext_filter: Sequence[str], suffix_filter: Sequence[str]) -> List[pathlib.Path]:
    return [p for p in sorted(input_dir.iterdir())
            if p.is_file() and p.suffix.lower().lstrip(".") in ext_filter
            and (not suffix_filter or any(p.stem.endswith(s) for s in suffix_filter))]


def split_stem_suffix(stem: str, x_suffix: str, y_suffix: str) -> Tuple[str, str]:
    if stem.endswith(x_suffix):
        return stem[:-len(x_suffix)], "X"
    if stem.endswith(y_suffix):
        return stem[:-len(y_suffix)], "Y"
    return stem, ""


def build_pair_records(image_paths: Sequence[pathlib.Path], x_suffix: str, y_suffix: str):
    table: Dict[str, Dict[str, pathlib.Path]] = {}
    for path in image_paths:
        base, key = split_stem_suffix(path.stem, x_suffix, y_suffix)
        if key:
            table.setdefault(base, {})[key] = path
    return [(base, table[base]["X"], table[base]["Y"]) for base in sorted(table) if "X" in table[base] and "Y" in table[base]]


def resolve_sensor_id_for_file(image_path, camera_to_sensor, sensor_map, sensor_id_x, sensor_id_y, x_suffix="_X", y_suffix="_Y"):
    stem = image_path.stem
    if camera_to_sensor.get(stem) in sensor_map:
        return camera_to_sensor[stem]
    if sensor_id_x and stem.endswith(x_suffix) and sensor_id_x in sensor_map:
        return sensor_id_x
    if sensor_id_y and stem.endswith(y_suffix) and sensor_id_y in sensor_map:
        return sensor_id_y
    if len(sensor_map) == 1:
        return next(iter(sensor_map))
    return None


def collect_mask_pair_paths(mask_dir: pathlib.Path, resolved_pairs):
    names = {p.name: p for p in sorted(mask_dir.iterdir()) if p.is_file()}
    matched, missing = {}, []
    for _idx, base, x_path, y_path, _sx, _sy in resolved_pairs:
        mx, my = names.get(x_path.name), names.get(y_path.name)
        missing += [n for n, m in ((x_path.name, mx), (y_path.name, my)) if m is None]
        if mx is not None and my is not None:
            matched[base] = (mx, my)
    if missing:
        uniq = sorted(set(missing))
        raise ValueError("Missing mask images in {}: {}".format(mask_dir, ", ".join(uniq[:8]) + (", ..." if len(uniq) > 8 else "")))
    return matched


def get_perspective_images_dir(root): return pathlib.Path(root) / "Images"          # noqa: E704
def get_perspective_masks_dir(root): return pathlib.Path(root) / "Masks"            # noqa: E704
def get_perspective_sparse_dir(root): return pathlib.Path(root) / "Sparse" / "0"    # noqa: E704


def _die(msg: str, code: int = 1):
    print(msg, file=sys.stderr)
    sys.exit(code)


def _write_image(path: pathlib.Path, image, jpeg_quality: Optional[int]):
    """cv2.imwrite stand-in: arrays here are in file channel order already (decoded and encoded by the same codec)."""
    from gs360 import imageio
    path.parent.mkdir(parents=True, exist_ok=True)
    if path.suffix.lower() in (".jpg", ".jpeg") and imageio.Image is not None:
        import numpy as np
        a = imageio.to_uint8(np.ascontiguousarray(image))       # JPEG is an 8-bit container
        mode = {1: "L", 3: "RGB", 4: "RGBA"}[1 if a.ndim == 2 else a.shape[2]]
        im = imageio.Image.fromarray(a[:, :, 0] if (a.ndim == 3 and a.shape[2] == 1) else a, mode)
        if mode == "RGBA":
            im = im.convert("RGB")
        im.save(path, "JPEG", quality=int(max(1, min(100, jpeg_quality if jpeg_quality is not None else 95))))
        return
    imageio.write_image(path, image)


def main() -> None:
    args = parse_arguments()
    try:
        zoom_override = parse_undistort_zoom_arg(args.undistort_zoom)
    except Exception as exc:
        _die("[ERR] --undistort-zoom: {}".format(exc))

    metadata_only = bool(args.metadata_only)
    input_value = str(args.input_dir or "").strip()
    input_path = pathlib.Path(input_value).expanduser().resolve() if input_value else None
    if input_path is None and not metadata_only:
        _die("[ERR] --input-dir is required unless --metadata-only is used.")

    legacy_profile = str(args.input_color_profile).strip().lower()
    input_lut_path = None
    if args.input_lut:
        input_lut_path = pathlib.Path(args.input_lut).expanduser().resolve()
    elif legacy_profile == "osmo360-dlogm":
        input_lut_path = pathlib.Path(args.dlogm_lut).expanduser().resolve()
    elif legacy_profile != "native":
        _die("[ERR] Unsupported --input-color-profile: {}".format(legacy_profile))
    input_lut = None
    if input_lut_path is not None:
        try:
            input_lut = color.load_cube_lut(input_lut_path)
        except Exception as exc:
            _die("[ERR] Failed to load input LUT: {}".format(exc))
    try:
        lut_space = normalize_lut_output_color_space(str(args.lut_output_color_space).strip().lower())
    except Exception as exc:
        _die("[ERR] {}".format(exc))

    camera_xml_value = str(args.camera_xml or "").strip()
    camera_xml_path = pathlib.Path(camera_xml_value).expanduser().resolve() if camera_xml_value else None
    suffix_filter = [t.strip() for t in args.suffixes.split(",") if t.strip()]
    if len(suffix_filter) < 2:
        _die("[ERR] --suffixes must include at least two values like '_X,_Y'.")
    x_suffix, y_suffix = suffix_filter[0], suffix_filter[1]

    fisheye_dir = None
    if input_path is not None:
        if input_path.is_file():
            _die("[ERR] Input must be a directory of fisheye frames, not a video file.\n"
                 "Use gs360_Video2Frames.py to extract *_X/*_Y images first.")
        if not input_path.is_dir():
            _die("[ERR] Input path not found: {}".format(input_path))
        fisheye_dir = input_path

    write_fisheye = bool(args.save_fisheye_output) and not metadata_only
    save_color = bool(args.save_color_corrected_output) and not metadata_only
    write_persp = (not bool(args.no_perspective)) and not metadata_only
    if not metadata_only and not (write_fisheye or write_persp or save_color):
        _die("[ERR] All outputs are disabled. Enable perspective, --save-fisheye-output, or --save-color-corrected-output.")

    extrinsics_value = str(args.camera_extrinsics_xml or "").strip()
    extrinsics_xml_path = None
    if extrinsics_value:
        extrinsics_xml_path = pathlib.Path(extrinsics_value).expanduser().resolve()
        if not extrinsics_xml_path.is_file():
            _die("[ERR] Camera extrinsics XML not found: {}".format(extrinsics_xml_path))
        if not write_persp and not metadata_only:
            _die("[ERR] --camera-extrinsics-xml requires perspective output.")

    out_arg = args.output_dir or args.fisheye_output_dir
    if out_arg:
        output_dir = pathlib.Path(out_arg).expanduser().resolve()
    elif fisheye_dir is not None:
        output_dir = fisheye_dir.with_name(fisheye_dir.name + "_undistorted")
    else:
        output_dir = pathlib.Path.cwd() / "_unused_dualfisheye_undistorted"
    if args.perspective_output_dir:
        persp_dir = pathlib.Path(args.perspective_output_dir).expanduser().resolve()
    elif fisheye_dir is not None:
        persp_dir = fisheye_dir.with_name(fisheye_dir.name + "_perspective_colmap")
    elif extrinsics_xml_path is not None:
        persp_dir = extrinsics_xml_path.with_name(extrinsics_xml_path.stem + "_perspective_colmap")
    else:
        persp_dir = pathlib.Path.cwd() / "perspective_colmap"
    if args.color_corrected_output_dir:
        color_dir = pathlib.Path(args.color_corrected_output_dir).expanduser().resolve()
    elif fisheye_dir is not None:
        color_dir = fisheye_dir.with_name(fisheye_dir.name + "_colorcorrected")
    else:
        color_dir = pathlib.Path.cwd() / "_unused_colorcorrected"

    ply_value = str(args.pointcloud_ply or "").strip()
    ply_path = None
    if ply_value:
        ply_path = pathlib.Path(ply_value).expanduser().resolve()
        if not ply_path.is_file():
            _die("[ERR] Point cloud PLY not found: {}".format(ply_path))
    if metadata_only:
        if extrinsics_xml_path is None:
            _die("[ERR] --metadata-only requires --camera-extrinsics-xml.")
        if ply_path is None:
            _die("[ERR] --metadata-only requires --pointcloud-ply.")
        _die("[ERR] --metadata-only: the COLMAP / Metashape metadata export is not part of the gs360 engine build "
             "(pixel path only); use the reference tool for metadata")

    calibration_xml_path = extrinsics_xml_path or camera_xml_path
    if calibration_xml_path is None:
        _die("[ERR] Specify --camera-extrinsics-xml or --camera-xml.")
    if not calibration_xml_path.is_file():
        _die("[ERR] Calibration XML not found: {}".format(calibration_xml_path))

    mask_dir_value = str(args.mask_input_dir or "").strip()
    mask_dir_path = None
    if mask_dir_value:
        mask_dir_path = pathlib.Path(mask_dir_value).expanduser().resolve()
        if not mask_dir_path.is_dir():
            _die("[ERR] Mask input directory not found: {}".format(mask_dir_path))
        if not write_persp:
            _die("[ERR] --mask-input-dir requires perspective output.")

    ext_filter = [t.strip().lower().lstrip(".") for t in args.ext.split(",") if t.strip()] or [e.lstrip(".") for e in SUPPORTED_EXTS]
    sensor_map, camera_to_sensor = fe.load_metashape_calibration(calibration_xml_path)
    if not sensor_map:
        _die("[ERR] No usable calibration found in XML.")
    unsupported = [s.sensor_id for s in sensor_map.values() if s.model_type not in SUPPORTED_MODELS]
    if unsupported:
        _die("[ERR] Unsupported model types in sensors: {}".format(", ".join(sorted(unsupported))))

    images = gather_input_images(fisheye_dir, ext_filter, suffix_filter)
    if not images:
        _die("[ERR] No target images found in {}".format(fisheye_dir))
    pairs = build_pair_records(images, x_suffix, y_suffix)
    if not pairs:
        _die("[ERR] No valid X/Y fisheye pairs found in {}".format(fisheye_dir))
    if args.limit:
        print("[WARN] --limit is deprecated and ignored. Processing all pairs.")
    if args.report_json:
        print("[WARN] --report-json is deprecated and ignored.")
    pair_images = [p for _b, x, y in pairs for p in (x, y)]

    interpolation = INTERPOLATION_MAP[args.interpolation]
    mask_value = int(max(0, min(255, args.mask_value)))
    workers = int(args.workers)
    if workers < 1:
        _die("[ERR] --workers must be >= 1.")
    mem_threshold = float(args.memory_throttle_percent) / 100.0
    if mem_threshold <= 0.0 or mem_threshold > 1.0:
        _die("[ERR] --memory-throttle-percent must be > 0 and <= 100.")

    say = print
    say("[INFO] input:  {}".format(fisheye_dir))
    say("[INFO] fisheye output: {}".format(output_dir) if write_fisheye else "[INFO] fisheye output: disabled")
    if write_persp:
        say("[INFO] perspective output: {}".format(persp_dir))
        say("[INFO] perspective xml: {}".format(persp_dir / args.perspective_metashape_xml_name))
        say("[INFO] perspective images dir: {}".format(get_perspective_images_dir(persp_dir)))
        say("[INFO] perspective sparse dir: {}".format(get_perspective_sparse_dir(persp_dir)))
        say("[INFO] perspective masks dir: {}".format(get_perspective_masks_dir(persp_dir)))
    else:
        say("[INFO] perspective output: disabled")
    say("[INFO] color-corrected output: {}".format(color_dir) if save_color else "[INFO] color-corrected output: disabled")
    say("[INFO] calibration xml: {}".format(calibration_xml_path))
    say("[INFO] pairs:  {}".format(len(pairs)))
    say("[INFO] files:  {}".format(len(pair_images)))
    say("[INFO] camera extrinsics xml: {}".format(extrinsics_xml_path) if extrinsics_xml_path else "[INFO] camera extrinsics xml: disabled")
    say("[INFO] pointcloud ply: {}".format(ply_path) if ply_path else "[INFO] pointcloud ply: disabled")
    say("[INFO] mask input dir: {}".format(mask_dir_path) if mask_dir_path else "[INFO] mask input dir: disabled")
    say("[INFO] workers: {} (memory auto-throttle > {}%)".format(workers, "{:.1f}".format(mem_threshold * 100.0)))
    say("[INFO] pair worker mode: enabled")
    if input_lut_path is not None:
        say("[INFO] input LUT: {}".format(input_lut_path))
        say("[INFO] LUT output color space: {}".format(lut_space))
    else:
        say("[INFO] input LUT: disabled")
    if write_fisheye:
        say("[INFO] undistort zoom: auto" if zoom_override is None else "[INFO] undistort zoom: {:.6f}".format(zoom_override))
    else:
        say("[INFO] undistort zoom: unused (direct perspective path)")

    processed = skipped = 0
    errors: List[str] = []
    color_count = persp_count = mask_count = 0
    resolved = []
    used_ids: Set[str] = set()
    used_pairs: Set[Tuple[str, str]] = set()
    for idx, (base, x_path, y_path) in enumerate(pairs, start=1):
        sx = resolve_sensor_id_for_file(x_path, camera_to_sensor, sensor_map, args.sensor_id_x, args.sensor_id_y, x_suffix, y_suffix)
        sy = resolve_sensor_id_for_file(y_path, camera_to_sensor, sensor_map, args.sensor_id_x, args.sensor_id_y, x_suffix, y_suffix)
        if sx is None or sy is None:
            skipped += 2
            say("[SKIP] {}: sensor_id unresolved".format(base))
            continue
        resolved.append((idx, base, x_path, y_path, sx, sy))
        used_ids.update([sx, sy])
        used_pairs.add((sx, sy))

    pair_masks = {}
    if mask_dir_path is not None:
        try:
            pair_masks = collect_mask_pair_paths(mask_dir_path, resolved)
        except Exception as exc:
            _die("[ERR] {}".format(exc))

    undistort = {}
    if write_fisheye and not args.dry_run:
        for sid in sorted(used_ids):
            try:
                undistort[sid] = fe.undistort_tables(sensor_map[sid], zoom_override, float(args.lens_fov_deg))
                say("[INFO] sensor {} undistort_zoom={:.6f}".format(sid, undistort[sid].undistort_zoom))
            except Exception as exc:
                err = "[ERR] sensor {}: remap build failed ({})".format(sid, exc)
                say(err)
                errors.append(err)
        if errors:
            sys.exit(2)

    specs: List[Dict[str, object]] = []
    tables: Dict[Tuple[str, str], Dict[str, Dict[str, object]]] = {}
    if write_persp:
        try:
            specs = fe.sfm10_specs(int(args.perspective_size), float(args.perspective_focal_mm), str(args.perspective_sensor_mm),
                                   float(args.perspective_yaw_delta_deg), float(args.perspective_pitch_delta_deg))
        except ValueError as exc:
            _die("[ERR] {}".format(exc))
        if not args.dry_run:
            for sp in sorted(used_pairs):
                try:
                    tables[sp] = fe.choose_lens_tables(sensor_map, sp[0], sp[1], specs, float(args.lens_x_yaw_deg),
                                                       float(args.lens_y_yaw_deg), float(args.lens_fov_deg))
                except Exception as exc:
                    err = "[ERR] perspective remap build failed for sensor pair {} / {} ({})".format(sp[0], sp[1], exc)
                    say(err)
                    errors.append(err)
            if errors:
                sys.exit(2)

    persp_ext = "." + args.perspective_ext.strip().lstrip(".").lower()
    mask_ext = "." + args.perspective_mask_ext.strip().lstrip(".").lower()
    jpeg_q = int(args.perspective_jpeg_quality)

    if args.dry_run:
        total = max(1, len(resolved))
        for idx, base, x_path, y_path, sx, sy in resolved:
            if save_color:
                for p in (x_path, y_path):
                    say("[DRY][COLOR] {:4d}/{:4d} {} -> {}".format(idx, total, p.name, p.name))
                color_count += 2
            if write_fisheye:
                say("[DRY] {:4d}/{:4d} {} -> {} (sensor_id={})".format(idx, total, x_path.name, x_path.name, sx))
                say("[DRY] {:4d}/{:4d} {} -> {} (sensor_id={})".format(idx, total, y_path.name, y_path.name, sy))
            if write_persp:
                for spec in specs:
                    say("[DRY][PERSP] {:4d}/{:4d} {}".format(idx, total, "{}_{}{}".format(base, spec["view_id"], persp_ext)))
                    if mask_dir_path is not None:
                        say("[DRY][MASK ] {:4d}/{:4d} {}".format(idx, total, "{}_{}{}".format(base, spec["view_id"], mask_ext)))
                persp_count += len(specs)
                if mask_dir_path is not None:
                    mask_count += len(specs)
            processed += 2
    else:
        from gs360 import capi, hostmem, imageio
        from gs360.dualfisheye import PairRenderer
        n_dev = capi.device_count()
        if n_dev <= 0:
            _die("[ERR] no MI355X visible: the gs360 engine has no CPU fallback", 2)
        hostmem.tune_malloc()                     # the worker threads' large short-lived image buffers (gs360/hostmem.py)
        contexts = [capi.Context(device=d, n_slots=1) for d in range(n_dev)]
        stage = color.ColorStage(input_lut, lut_space) if input_lut is not None else None
        renderers = {}
        for d, ctx in enumerate(contexts):
            for sp in sorted(used_pairs):
                renderers[(d, sp)] = PairRenderer(ctx, sensor_map, specs, tables.get(sp),
                                                  {sid: undistort[sid] for sid in sp if sid in undistort},
                                                  float(args.lens_fov_deg), fused=(args.map_mode == "fused"))

        def pair_task(idx, base, x_path, y_path, sx, sy):
            r = renderers[((idx - 1) % n_dev, (sx, sy))]           # pairs shard across GPUs, no exchange step
            img_x, img_y = imageio.read_image(x_path), imageio.read_image(y_path)
            mx = my = None
            if base in pair_masks:
                mx, my = imageio.read_image(pair_masks[base][0]), imageio.read_image(pair_masks[base][1])
            res = r.render_pair(img_x, img_y, sx, sy, interpolation=interpolation, mask_outside_model=bool(args.mask_outside_model),
                                mask_value=mask_value, mask_x=mx, mask_y=my, want_fisheye=write_fisheye, want_perspective=write_persp,
                                color_stage=stage, want_color=save_color)
            names = {"color": [], "fisheye": [], "persp": [], "mask": []}
            if save_color:
                for key, p in (("X", x_path), ("Y", y_path)):
                    _write_image(color_dir / p.name, res["color"][key].reshape(
                        (img_x if key == "X" else img_y).shape), None)
                    names["color"].append(p.name)
            for key, p in (("X", x_path), ("Y", y_path)):
                if key in res["fisheye"]:
                    _write_image(output_dir / p.name, res["fisheye"][key], None)
                    names["fisheye"].append(p.name)
            for spec in specs if write_persp else ():
                vid = str(spec["view_id"])
                name = "{}_{}{}".format(base, vid, persp_ext)
                _write_image(get_perspective_images_dir(persp_dir) / name, res["perspective"][vid], jpeg_q)
                names["persp"].append(name)
                if vid in res["masks"]:
                    mname = "{}_{}{}".format(base, vid, mask_ext)
                    _write_image(get_perspective_masks_dir(persp_dir) / mname, res["masks"][vid], jpeg_q)
                    names["mask"].append(mname)
            return names

        pending, meta = set(), {}

        def drain(done):
            nonlocal processed, skipped, color_count, persp_count, mask_count
            for fut in done:
                idx, base = meta.pop(fut)
                try:
                    names = fut.result()
                except Exception as exc:  # noqa: BLE001
                    skipped += 2
                    err = "[ERR] {}: {}".format(base, exc)
                    say(err)
                    errors.append(err)
                    continue
                for n in names["color"]:
                    say("[OK ][COLOR] {:4d}/{:4d} {} -> {}".format(idx, len(pairs), n, n))
                for n in names["fisheye"]:
                    say("[OK ][FISH] {:4d}/{:4d} {} -> {}".format(idx, len(pairs), n, n))
                if names["persp"]:
                    say("[OK ][PERSP] {:4d}/{:4d} {} -> {} views".format(idx, len(pairs), base, len(names["persp"])))
                if names["mask"]:
                    say("[OK ][MASK ] {:4d}/{:4d} {} -> {} masks".format(idx, len(pairs), base, len(names["mask"])))
                processed += 2
                color_count += len(names["color"])
                persp_count += len(names["persp"])
                mask_count += len(names["mask"])

        with ThreadPoolExecutor(max_workers=workers) as pool:
            for item in resolved:
                while pending and len(pending) >= workers:
                    done, pending = wait(pending, return_when=FIRST_COMPLETED)
                    drain(list(done))
                fut = pool.submit(pair_task, *item)
                pending.add(fut)
                meta[fut] = (item[0], item[1])
            while pending:
                done, pending = wait(pending, return_when=FIRST_COMPLETED)
                drain(list(done))
        if stage is not None:
            stage.close()
        for r in renderers.values():
            r.close()
        for ctx in contexts:
            ctx.close()

    if extrinsics_xml_path is not None and not args.dry_run:
        err = ("[ERR] perspective camera metadata export failed (COLMAP / Metashape export is not part of the gs360 "
               "engine build; images were rendered)")
        print(err, file=sys.stderr)
        errors.append(err)

    say("[DONE] processed={} skipped={} total={} persp_outputs={} mask_outputs={} color_outputs={} errors={}".format(
        processed, skipped, len(pair_images), persp_count, mask_count, color_count, len(errors)))
    if errors:
        sys.exit(2)


if __name__ == "__main__":
    main()
