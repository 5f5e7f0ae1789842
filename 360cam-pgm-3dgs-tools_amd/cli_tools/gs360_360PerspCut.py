#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""gs360_360PerspCut -- MI355X drop-in for the reference tool of the same name.

Same module surface the GUI and the other tools bind to (reference gs360_GUI.py:54, :301, :2090, :18850,
:19299; gs360_Video2Frames.py:26): create_arg_parser, build_view_jobs, run_one, stop_event, procs_lock,
running_procs, parse_jobs, detect_input_bit_depth, EXTS, PROGRESS_INTERVAL, ViewSpec, BuildResult,
fov_from_focal_mm, v_fov_from_hfov.  Same flags, preset names, stdout lines, exit codes and output layout
(reference cli_tools/gs360_360PerspCut.py:417-532, :983-1087).

What changed underneath: run_one() no longer spawns one single-threaded `ffmpeg -vf v360=...` process per
(source, view) (reference :569-590).  It parses the ffmpeg-shaped job argv and executes equirect->rectilinear
jobs in-process on the GPU through libgs360hip.so (hand-written HIP, gfx950).  Video inputs are decoded ONCE per
video by a single ffmpeg process feeding device memory; every view job then samples the HBM-resident frames
(gs360/video.py) instead of decoding the video again.  The `fisheyeXY` preset's `output=fisheye` jobs run on the GPU too.
Jobs the engine does not cover (16-bit output, other v360 projections) and `--engine ffmpeg` / GS360_ENGINE=ffmpeg keep the reference's subprocess path.
"""
import argparse
import json
import os
import pathlib
import shlex
import shutil
import signal
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor, as_completed
from typing import List, Tuple

_HERE = pathlib.Path(__file__).resolve().parent
if str(_HERE.parent) not in sys.path:
    sys.path.insert(0, str(_HERE.parent))

from gs360 import planner as _planner  # noqa: E402
from gs360.jobspec import JobParseError, parse_job_argv  # noqa: E402
from gs360.planner import (BuildResult, ViewSpec, build_ffmpeg_cmd, build_ffmpeg_equisolid_cmd,  # noqa: E402,F401
                           clamp, extra_suffix, focal_from_hfov_deg, fov_from_focal_mm, letter_tag,
                           letter_to_index1, map_interp_for_v360, normalize_angle_deg, parse_addcam_spec,
                           parse_delcam_spec, parse_sensor, parse_sensor_dimensions, parse_setcam_spec,
                           v_fov_from_hfov)

EXTS = {".tif", ".tiff", ".jpg", ".jpeg", ".png"}
PROGRESS_INTERVAL = 5


class StoreWithFlag(argparse.Action):
    """Stores the value and marks `<dest>_explicit` so presets only override untouched options (PC:24-29)."""

    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, values)
        setattr(namespace, f"{self.dest}_explicit", True)


def update_progress(label: str, completed: int, total: int, last_pct: int) -> int:
    if total <= 0:
        return last_pct
    pct = int((completed * 100) / total)
    if last_pct < 0 or pct >= 100 or (pct - last_pct) >= PROGRESS_INTERVAL:
        sys.stdout.write(f"{label}... {pct:3d}% ({completed}/{total})\r")
        sys.stdout.flush()
        return pct
    return last_pct


_HIGH_DEPTH_TOKENS = ("p10", "p12", "p14", "p16", "p010", "p012", "p016", "gbrp10", "gbrp12", "gbrp14", "gbrp16",
                      "rgb48", "rgba64")


def detect_input_bit_depth(in_path: pathlib.Path) -> int:
    """Nominal bit depth of a video's first stream via ffprobe; 8 when unknown (PC:111-149)."""
    if not shutil.which("ffprobe"):
        return 8
    try:
        res = subprocess.run(["ffprobe", "-v", "error", "-select_streams", "v:0", "-show_entries",
                              "stream=bits_per_raw_sample,pix_fmt", "-of", "json", str(in_path)],
                             check=True, capture_output=True, text=True)
        stream = (json.loads(res.stdout or "{}").get("streams") or [{}])[0]
        raw = stream.get("bits_per_raw_sample")
        if isinstance(raw, str) and raw.isdigit():
            return int(raw) if int(raw) >= 9 else 8
        if any(tok in (stream.get("pix_fmt") or "") for tok in _HIGH_DEPTH_TOKENS):
            return 10
    except Exception:
        pass
    return 8


# Option table: (flags, argparse keywords).  Names, destinations, types and defaults are the reference's CLI surface
# (PC:432-531) because the GUI builds Namespaces from this parser (gs360_GUI.py:301); the help wording is ours.
_OPTIONS = (
    (("-i", "--in"), dict(dest="input_dir", required=True, help="folder of equirectangular stills, or one equirectangular video file")),
    (("-o", "--out"), dict(dest="out_dir", default=None, help="where the views go (default: <input>/_geometry)")),
    (("--preset",), dict(choices=list(_planner.PRESET_NAMES), default="default",
                         help="view layout: default = 8-view ring; fisheyelike = 10 views at 17 mm; full360coverage = 12 views at "
                              "14 mm with +/-30 deg extras; 2views = front/back at 6 mm, 3600 px; evenMinus30 / evenPlus30 = even "
                              "slots pitched; fisheyeXY = the X/Y fisheye pair only")),
    (("--count",), dict(type=int, default=8, help="number of yaw slots around the horizon")),
    (("--addcam",), dict(default="", help="extra pitched views per slot letter, e.g. 'B', 'B:U', 'D:D20' (comma separated)")),
    (("--addcam-deg",), dict(type=float, default=30.0, help="pitch magnitude used when U/D carry no number")),
    (("--add-top",), dict(action="store_true", help="also emit a straight-up view (pitch +90)")),
    (("--add-bottom",), dict(action="store_true", help="also emit a straight-down view (pitch -90)")),
    (("--add-topdown",), dict(action="store_true", dest="add_topdown", help=argparse.SUPPRESS)),
    (("--delcam",), dict(default="", help="drop ring views by letter, e.g. 'B,D'")),
    (("--setcam",), dict(default="", help="set ('A=30', 'A=U', 'A=D20') or shift ('A:+10') the pitch of a view")),
    (("--size",), dict(type=int, default=1600, action="flagged", help="edge length of the square views in pixels")),
    (("--ext",), dict(default="jpg", help="output image type (jpg / png / tif)")),
    (("--jpeg-quality-95",), dict(action="store_true", help="write JPEGs at ~95 %% quality instead of the maximum")),
    (("-f", "--fps"), dict(type=float, default=None, help="frames per second to extract when the input is a video")),
    (("--start",), dict(type=float, default=None, help="video start time in seconds")),
    (("--end",), dict(type=float, default=None, help="video end time in seconds")),
    (("--keep-rec709",), dict(action="store_true", help="keep the Rec.709 transfer curve of video inputs (default converts to sRGB)")),
    (("--hfov",), dict(type=float, default=None, action="flagged", help="horizontal field of view in degrees; wins over --focal-mm")),
    (("--focal-mm",), dict(type=float, default=12.0, action="flagged", help="focal length in mm on the virtual sensor")),
    (("--sensor-mm",), dict(default="36 36", help="virtual sensor size in mm, '36 36' or '36x24'")),
    (("-j", "--jobs"), dict(default="auto", help="concurrent jobs (a number, or 'auto' = half the cores)")),
    (("--print-cmd",), dict(choices=["once", "none", "all"], default="once", help="how many planned job lines to echo")),
    (("--ffmpeg",), dict(default="ffmpeg", help="ffmpeg executable used for jobs that stay on the subprocess path")),
    (("--dry-run",), dict(action="store_true", help="print the plan and exit")),
    # additive: executor selection (default = GS360_ENGINE or the HIP engine)
    (("--engine",), dict(choices=["hip", "ffmpeg"], default=None, help="hip = in-process MI355X engine (default), ffmpeg = one subprocess per job")),
)


def create_arg_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(
        description="Cut equirectangular panoramas into preset perspective views on the GPU (drop-in for the ffmpeg/v360 tool).",
        formatter_class=argparse.ArgumentDefaultsHelpFormatter,
        epilog="Presets only replace --size / --focal-mm / --hfov values you did not pass yourself.")
    for flags, kw in _OPTIONS:
        kw = dict(kw)
        if kw.get("action") == "flagged":
            kw["action"] = StoreWithFlag
        ap.add_argument(*flags, **kw)
    return ap


# ---- cancellation and the subprocess executor (contract: PC:535-590) -----------------------------------------------
stop_event = threading.Event()      # names below are read by the GUI (gs360_GUI.py:9170, :18811)
procs_lock = threading.Lock()
running_procs = set()
sig_hits = 0
_engine_choice = None               # set by main(); None -> environment / default


def on_signal(sig, frame):
    """First signal: stop handing out work and ask children to terminate; second: kill them."""
    global sig_hits
    sig_hits += 1
    first = not stop_event.is_set()
    stop_event.set()
    if first:
        print("\n[INFO] Cancel requested. Stopping new jobs and terminating running processes...", file=sys.stderr)
    with procs_lock:
        victims = list(running_procs)
    for proc in victims:
        try:
            (proc.terminate if sig_hits == 1 else proc.kill)()
        except Exception:
            pass
    if sig_hits >= 2:
        print("[INFO] Force exiting", file=sys.stderr)


def _install_signal_handlers():
    if threading.current_thread() is not threading.main_thread():
        return
    names = ["SIGINT", "SIGTERM"] + (["SIGBREAK"] if os.name == "nt" else [])
    for name in names:
        try:
            signal.signal(getattr(signal, name), on_signal)
        except (AttributeError, ValueError, OSError):
            pass


_install_signal_handlers()


def parse_jobs(s: str) -> int:
    if str(s).lower() == "auto":
        return max(1, (os.cpu_count() or 1) // 2)
    return max(1, int(s))


def _selected_engine() -> str:
    return (_engine_choice or os.environ.get("GS360_ENGINE") or "hip").lower()


def _run_subprocess(cmd: List[str]) -> Tuple[int, str]:
    """One external process for the job, polled twice a second so a cancel request reaches it quickly."""
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    except OSError as exc:
        return 127, f"{cmd[0]}: {exc}"
    with procs_lock:
        running_procs.add(proc)
    rc = None
    try:
        while rc is None:
            try:
                rc = proc.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                if stop_event.is_set():
                    try:
                        proc.terminate()
                    except Exception:
                        pass
        text = (proc.stderr.read() or b"").decode(errors="ignore")
        return rc, text
    finally:
        with procs_lock:
            running_procs.discard(proc)


def run_one(cmd: List[str]) -> Tuple[int, str]:
    """Execute one planned job.  Never raises: returns (rc, stderr_text); rc 0 ok, 130 cancelled (PC:569-590)."""
    if stop_event.is_set():
        return 130, ""
    if _selected_engine() == "ffmpeg":
        return _run_subprocess(cmd)
    try:
        job = parse_job_argv(list(cmd))
    except JobParseError as exc:
        return 2, f"gs360: cannot interpret job argv: {exc}"
    if job.output_projection not in ("rectilinear", "fisheye"):
        return _run_subprocess(cmd)          # anything else v360 can do stays on the reference path
    plan = None
    if not job.is_still_image:
        # video: one shared decoder process + HBM-resident frames (gs360/video.py); argv shapes it does not
        # understand (16-bit output, foreign options) run as the reference's per-view subprocess
        from gs360 import video as _video
        plan = _video.build_decode_plan(job)
        if plan is None:
            return _run_subprocess(cmd)
    try:
        from gs360 import engine as _engine
        if plan is None:
            _engine.get_engine().run_job(job, stop_event=stop_event)
        else:
            _engine.get_engine().run_video_job(job, plan, stop_event=stop_event, register_proc=_track_proc,
                                               expected_jobs=_planned_video_jobs.get(str(job.src)))
    except Exception as exc:  # noqa: BLE001  (boundary: HIP / IO / decoder errors become rc + text)
        return 1, f"gs360: {type(exc).__name__}: {exc}"
    if stop_event.is_set():
        return 130, ""
    return 0, ""


_planned_video_jobs: dict = {}     # video path -> view jobs planned for it (lets the engine free a video's frames)


def _track_proc(proc, add: bool) -> None:
    """The shared video decoder is registered like the reference's per-view processes so a cancel reaches it."""
    with procs_lock:
        (running_procs.add if add else running_procs.discard)(proc)


def build_view_jobs(args, files: List[pathlib.Path], out_dir: pathlib.Path) -> BuildResult:
    """Compose job definitions and view specifications (planning only, PC:593-980)."""
    result = _planner.build_view_jobs(args, files, out_dir, stop_event=stop_event)
    if getattr(args, "input_is_video", False):
        counts: dict = {}
        for cmd, _src, _dst in result.jobs:
            if "-i" in cmd:
                key = str(pathlib.Path(cmd[cmd.index("-i") + 1]))
                counts[key] = counts.get(key, 0) + 1
        _planned_video_jobs.update(counts)
    return result


def _announce_jobs(jobs_list, workers: int) -> None:
    """Tell the engine how many view jobs of each source can arrive together so that it batches them into one launch
    without waiting out its linger window (gs360/engine.py).  Purely a hint: run_one works without it (the GUI path)."""
    try:
        from gs360 import engine as _engine
        specs = []
        for cmd, _src, _dst in jobs_list:
            try:
                specs.append(parse_job_argv(list(cmd)))
            except JobParseError:
                pass
        if specs:
            _engine.get_engine().announce(specs, workers=workers)
    except Exception:  # noqa: BLE001  (no GPU / no library: run_one reports that per job, exactly as before)
        pass


def _retire_jobs() -> None:
    """End of a run: the engine forgets what the announced list left behind (jobs of a cancelled or failed run never arrive)."""
    try:
        from gs360 import engine as _engine
        if _engine._engine is not None:
            _engine._engine.retire()
    except Exception:  # noqa: BLE001
        pass


# ---- main (PC:983-1087) --------------------------------------------------------------------------------
def _print_cmd(cmd):
    print("$ " + " ".join(shlex.quote(c) for c in cmd))


def main():
    global _engine_choice
    args = create_arg_parser().parse_args()
    for attr in ("size", "hfov", "focal_mm"):
        setattr(args, f"{attr}_explicit", getattr(args, f"{attr}_explicit", False))
    _engine_choice = args.engine

    input_path = pathlib.Path(args.input_dir).expanduser().resolve()
    if input_path.is_dir():
        args.input_is_video, args.video_bit_depth = False, 8
        out_dir = pathlib.Path(args.out_dir).resolve() if args.out_dir else (input_path / "_geometry")
        out_dir.mkdir(parents=True, exist_ok=True)
        files = [p for p in sorted(input_path.iterdir()) if p.is_file() and p.suffix.lower() in EXTS]
        if not files:
            print("[WARN] No target images found (tif/jpg/png)", file=sys.stderr)
            sys.exit(0)
    elif input_path.is_file():
        args.input_is_video = True
        if args.fps is None or args.fps <= 0:
            print("[ERR] -f/--fps must be specified for video inputs", file=sys.stderr)
            sys.exit(1)
        out_dir = pathlib.Path(args.out_dir).resolve() if args.out_dir else (
            input_path.parent / f"{input_path.stem}_geometry")
        out_dir.mkdir(parents=True, exist_ok=True)
        args.video_bit_depth = detect_input_bit_depth(input_path)
        files = [input_path]
    else:
        print("[ERR] Input path not found:", input_path, file=sys.stderr)
        sys.exit(1)

    result = build_view_jobs(args, files, out_dir)
    jobs_list, total = result.jobs, result.total

    if args.dry_run:
        for cmd, _, _ in jobs_list:
            _print_cmd(cmd)
        print(f"\n[DRY] Exiting without execution (total {total} commands)")
        return

    if args.print_cmd == "all":
        for cmd, _, _ in jobs_list:
            _print_cmd(cmd)
    elif args.print_cmd == "once" and jobs_list:
        _print_cmd(jobs_list[0][0])

    jobs = parse_jobs(args.jobs)
    print(f"[INFO] parallel jobs: {jobs} / total: {total}")
    if result.preview_views_line:
        print(result.preview_views_line)
        for line in (result.sensor_line, result.realityscan_line, result.metashape_line):
            if line:
                print(line)

    # `-j auto` is cores // 2 ffmpeg processes in the reference (PC:563-567).  The HIP engine's view jobs are threads of this process
    # that spend their time in the image codecs, and os.cpu_count() ignores a container's CPU quota (the MI355X boxes of the build
    # pool show 256 hardware threads and allow 16 CPUs: 48 panoramas run at 46 frames/s with 32 workers, 35 with 128).  `auto` is
    # therefore capped at twice the CPUs the process may actually use; an explicit -j is taken as given.
    workers = jobs
    if _selected_engine() != "ffmpeg":
        from gs360 import hostmem
        hostmem.tune_malloc()                     # the codec threads' large short-lived buffers: this process is the tool itself
        if str(args.jobs).lower() == "auto":
            cap = int(os.environ.get("GS360_AUTO_WORKERS", "0")) or max(8, 2 * hostmem.effective_cpus())
            workers = min(jobs, max(1, cap))
        _announce_jobs(jobs_list, workers)
    ok = fail = done = 0
    last_pct = -1
    with ThreadPoolExecutor(max_workers=workers) as pool:
        futures = [pool.submit(run_one, cmd) for cmd, _, _ in jobs_list]
        for fut, (_, _src, dst) in zip(as_completed(futures), jobs_list):
            rc, err = fut.result()
            done += 1
            if rc == 0:
                ok += 1
                last_pct = update_progress("Progress", done, total, last_pct)
                continue
            fail += 1
            if stop_event.is_set():
                continue
            if total:
                last_pct = update_progress("Progress", done, total, last_pct)
                sys.stdout.write("\n")
                sys.stdout.flush()
            print(f"[{done}/{total}] {dst} {'canceled' if rc == 130 else 'failed'}", file=sys.stderr)
            if err.strip():
                print(err.strip(), file=sys.stderr)
    if total and last_pct >= 0:
        sys.stdout.write("\n")
        sys.stdout.flush()
    _retire_jobs()

    if stop_event.is_set():
        print(f"[STOPPED] Interrupted: success={ok}, failed={fail}, total={total}")
        sys.exit(130)
    print(f"[OK] Completed: success={ok}, failed={fail}, total={total}")


if __name__ == "__main__":
    main()
