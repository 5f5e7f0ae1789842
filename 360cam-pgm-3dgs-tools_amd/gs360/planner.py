"""View planner of the 360PerspCut drop-in: flags + preset -> ViewSpec list + ffmpeg-shaped job argv.

Behavioural contract: cli_tools/gs360_360PerspCut.py:77-283 (helpers, camera grammars), :286-414 (argv shape)
and :593-980 (build_view_jobs) of the reference; pinned by tests/golden/planner_goldens.json, which was
captured by importing the reference planner.  The implementation is table-driven (PRESETS below) rather
than a transcription: the GUI edits the argv between planning and execution (gs360_GUI.py:19081-19148), so
jobs stay ffmpeg-shaped and gs360.jobspec parses them back for the HIP engine.
"""
import math
import pathlib
import re
import sys
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Set, Tuple

# ------------------------------------------------------------------------------------------------
# public dataclasses (same field names/order as the reference: PC:32-65)
# ------------------------------------------------------------------------------------------------


@dataclass
class ViewSpec:
    source_path: pathlib.Path
    output_name: str
    view_id: str
    yaw_deg: float
    pitch_deg: float
    hfov_deg: float
    vfov_deg: float
    width: int
    height: int
    projection: str = "perspective"


@dataclass
class BuildResult:
    jobs: List[Tuple[List[str], str, str]]
    view_specs: List[ViewSpec]
    focal_used_mm: float
    focal_35mm_equiv: Optional[float]
    hfov_deg: float
    vfov_deg: float
    preview_views_line: str
    sensor_line: str
    realityscan_line: str
    metashape_line: str

    @property
    def total(self) -> int:
        return len(self.jobs)


# ------------------------------------------------------------------------------------------------
# preset table (PC:616-644, :654-680)
# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Preset:
    force_count: Optional[int] = None   # ring size imposed by the preset
    focal_mm: Optional[float] = None    # used unless --hfov / --focal-mm were given explicitly
    size: Optional[int] = None          # used unless --size was given explicitly
    auto_del: str = ""                  # ring slots dropped unless the user passed --delcam
    auto_add: str = ""                  # slots that get +/- addcam-deg extras unless the user passed --addcam
    hard_del: str = ""                  # slots always dropped
    even_pitch: Optional[float] = None  # pitch offset of even slots
    fisheye_pair: bool = False          # emit only the X/Y equisolid pair (slots 1 and 5)
    announce_count: bool = False


PRESETS: Dict[str, Preset] = {
    "default": Preset(),
    "fisheyelike": Preset(force_count=10, focal_mm=17.0, auto_del="CDHI", auto_add="AF"),
    "full360coverage": Preset(force_count=8, focal_mm=14.0, auto_del="BDFH", auto_add="BDFH"),
    "2views": Preset(focal_mm=6.0, size=3600, hard_del="BCDFGH"),
    "evenMinus30": Preset(even_pitch=-30.0),
    "evenPlus30": Preset(even_pitch=+30.0),
    "fisheyeXY": Preset(force_count=8, fisheye_pair=True, announce_count=True),
}
PRESET_NAMES = ["default", "fisheyelike", "full360coverage", "2views", "evenMinus30", "evenPlus30", "fisheyeXY"]
FISHEYE_PAIR_SLOTS = {1: "X", 5: "Y"}

# ------------------------------------------------------------------------------------------------
# scalar helpers (PC:77-109, :151-180)
# ------------------------------------------------------------------------------------------------


def fov_from_focal_mm(f_mm: float, sensor_w_mm: float) -> float:
    return math.degrees(2.0 * math.atan(sensor_w_mm / (2.0 * f_mm)))


def focal_from_hfov_deg(hfov_deg: float, sensor_w_mm: float) -> float:
    return sensor_w_mm / (2.0 * math.tan(math.radians(hfov_deg) / 2.0))


def v_fov_from_hfov(hfov_deg: float, w: int, h: int) -> float:
    half = math.tan(math.radians(hfov_deg) / 2.0) * (h / float(w))
    return math.degrees(2.0 * math.atan(half))


def letter_tag(idx: int) -> str:
    return chr(ord("A") + idx) if idx < 26 else f"{idx + 1:02d}"


def letter_to_index1(s: str) -> int:
    s = s.strip()
    if not s:
        raise ValueError("empty key")
    if s.isdigit():
        return int(s)
    first = s.upper()[0]
    if "A" <= first <= "Z":
        return ord(first) - ord("A") + 1
    raise ValueError("invalid key: " + s)


def normalize_angle_deg(a: float) -> float:
    a = ((a + 180.0) % 360.0) - 180.0
    return 180.0 if abs(a + 180.0) < 1e-6 else a


def clamp(v: float, lo: float, hi: float) -> float:
    return max(lo, min(hi, v))


def map_interp_for_v360(name: str) -> str:
    return {"bicubic": "cubic", "bilinear": "linear", "lanczos": "lanczos"}.get((name or "").lower(), "cubic")


def _sensor_text(s: str) -> str:
    return s.lower().replace("×", "x").replace(",", " ").strip()


def parse_sensor(s: str) -> float:
    t = _sensor_text(s)
    head = t.split("x")[0].strip() if "x" in t else t.split()[0]
    return float(head)


def parse_sensor_dimensions(s: str) -> Tuple[float, ...]:
    t = _sensor_text(s)
    parts = [p.strip() for p in t.split("x") if p.strip()] if "x" in t else t.split()
    dims = []
    for p in parts:
        try:
            dims.append(float(p))
        except ValueError:
            pass
    return tuple(dims)


def extra_suffix(delta_pitch: float, default_deg: float = 30.0) -> str:
    head = "_U" if delta_pitch > 0 else "_D"
    mag = abs(delta_pitch)
    if abs(mag - default_deg) < 1e-6:
        return head
    if float(mag).is_integer():
        return f"{head}{int(round(mag))}"
    return f"{head}{mag:g}"


# ------------------------------------------------------------------------------------------------
# --addcam / --delcam / --setcam grammars (PC:183-283)
# ------------------------------------------------------------------------------------------------
_KV = re.compile(r"[:=]")
_ADD_VALUE = re.compile(r"^([UD])\s*([+-]?\d+(?:\.\d+)?)?$")
_SET_RELATIVE = re.compile(r"^[+|-]\s*\d+(?:\.\d+)?$")   # the literal '|' is accepted as a sign, then float() rejects it
_SET_UP = re.compile(r"^[Uu]\s*(\d+(?:\.\d+)?)?$")
_SET_DOWN = re.compile(r"^[Dd]\s*(\d+(?:\.\d+)?)?$")


def _tokens(spec: str):
    for tok in (spec or "").split(","):
        tok = tok.strip()
        if tok:
            yield tok


def parse_addcam_spec(spec: str, default_deg: float) -> Dict[int, List[float]]:
    extras: Dict[int, List[float]] = {}
    for tok in _tokens(spec):
        if _KV.search(tok):
            key, value = _KV.split(tok, maxsplit=1)
            slot = letter_to_index1(key)
            m = _ADD_VALUE.match(value.strip().upper())
            if not m:
                raise ValueError("invalid --addcam token: " + tok)
            deg = float(m.group(2)) if m.group(2) else default_deg
            extras.setdefault(slot, []).append(deg if m.group(1) == "U" else -deg)
        else:
            extras.setdefault(letter_to_index1(tok), []).extend([+default_deg, -default_deg])
    return extras


def parse_delcam_spec(spec: str) -> Set[int]:
    return {letter_to_index1(tok) for tok in _tokens(spec)}


def parse_setcam_spec(spec: str, default_deg: float):
    """-> (abs_map, delta_map, extra_abs_map, extra_delta_map); extra maps keyed by (slot, suffix)."""
    abs_map: Dict[int, float] = {}
    delta_map: Dict[int, float] = {}
    extra_abs: Dict[Tuple[int, str], float] = {}
    extra_delta: Dict[Tuple[int, str], float] = {}
    for tok in _tokens(spec):
        if not _KV.search(tok):
            raise ValueError("invalid --setcam token: " + tok)
        raw_key, raw_val = _KV.split(tok, maxsplit=1)
        raw_key = raw_key.strip()
        suffix = None
        base = raw_key
        if "_" in raw_key:
            base, tail = raw_key.split("_", 1)
            suffix = "_" + tail.strip()
        slot = letter_to_index1(base)
        key = (slot, suffix) if suffix else slot
        into_abs, into_delta = (extra_abs, extra_delta) if suffix else (abs_map, delta_map)
        val = raw_val.strip()
        if _SET_RELATIVE.match(val):
            into_delta[key] = float(val.replace(" ", ""))
            continue
        up, down = _SET_UP.match(val), _SET_DOWN.match(val)
        if up:
            into_abs[key] = +(float(up.group(1)) if up.group(1) else default_deg)
        elif down:
            into_abs[key] = -(float(down.group(1)) if down.group(1) else default_deg)
        else:
            try:
                into_abs[key] = float(val.replace(" ", ""))
            except Exception as exc:
                raise ValueError("invalid --setcam token: " + tok) from exc
    return abs_map, delta_map, extra_abs, extra_delta


# ------------------------------------------------------------------------------------------------
# ffmpeg-shaped argv (PC:286-414).  Kept byte-identical so the unmodified GUI can rewrite it.
# ------------------------------------------------------------------------------------------------
@dataclass
class MediaOptions:
    video_mode: bool = False
    fps: Optional[float] = None
    keep_rec709: bool = False
    bit_depth: int = 8
    jpeg_quality_95: bool = False
    start_time: Optional[float] = None
    end_time: Optional[float] = None


def _job_argv(ffmpeg: str, inp: pathlib.Path, out: pathlib.Path, v360_filter: str, ext: str, m: MediaOptions) -> List[str]:
    ext = ext.lower()
    is_jpeg = ext in (".jpg", ".jpeg")
    chain: List[str] = []
    if m.video_mode:
        if m.fps is None or m.fps <= 0:
            raise ValueError("fps must be specified and > 0 when processing a video input")
        cs = "colorspace=iall=bt709:all=smpte170m" + ("" if m.keep_rec709 else ":trc=iec61966-2-1")
        chain += [f"fps={m.fps}", cs + (":range=jpeg" if is_jpeg else "") + ":format=yuv444p"]
    chain.append(v360_filter)

    argv = [ffmpeg, "-hide_banner", "-loglevel", "error", "-y"]
    if m.video_mode and m.start_time is not None:
        argv += ["-ss", f"{max(0.0, float(m.start_time))}"]
    argv += ["-i", str(inp)]
    if m.video_mode and m.end_time is not None:
        argv += ["-to", f"{max(0.0, float(m.end_time))}"]
    argv += ["-vf", ",".join(chain), "-threads", "1"]
    argv += ["-vsync", "vfr", "-start_number", "0"] if m.video_mode else ["-frames:v", "1"]
    if is_jpeg:
        q = "2" if m.jpeg_quality_95 else "1"
        argv += ["-c:v", "mjpeg", "-q:v", q, "-qmin", q, "-qmax", q, "-pix_fmt", "yuvj444p", "-huffman", "optimal"]
        if m.video_mode:
            argv += ["-colorspace", "smpte170m", "-color_primaries", "smpte170m", "-color_trc", "smpte170m"]
    elif m.video_mode and ext in (".png", ".tif", ".tiff"):
        argv += ["-pix_fmt", "rgb48le" if m.bit_depth > 8 else "rgb24"]
    argv.append(str(out))
    return argv


def _media(video_mode, fps, keep_rec709, bit_depth, jpeg_quality_95, start_time, end_time) -> MediaOptions:
    return MediaOptions(video_mode, fps, keep_rec709, bit_depth, jpeg_quality_95, start_time, end_time)


def build_ffmpeg_cmd(ffmpeg, inp, out, w, h, yaw, pitch, hfov, vfov, interp_v360, ext, *, video_mode=False, fps=None,
                     keep_rec709=False, bit_depth=8, jpeg_quality_95=False, start_time=None, end_time=None):
    flt = (f"v360=input=equirect:output=rectilinear:w={w}:h={h}:yaw={yaw}:pitch={pitch}:roll=0"
           f":h_fov={hfov}:v_fov={vfov}:interp={interp_v360}")
    return _job_argv(ffmpeg, inp, out, flt, ext,
                     _media(video_mode, fps, keep_rec709, bit_depth, jpeg_quality_95, start_time, end_time))


def build_ffmpeg_equisolid_cmd(ffmpeg, inp, out, w, h, yaw, pitch, fov_deg, interp_v360, ext, *, video_mode=False,
                               fps=None, keep_rec709=False, bit_depth=8, jpeg_quality_95=False, start_time=None,
                               end_time=None):
    flt = (f"v360=input=equirect:output=fisheye:w={w}:h={h}:yaw={yaw}:pitch={pitch}:roll=0"
           f":d_fov={fov_deg}:interp={interp_v360}")
    return _job_argv(ffmpeg, inp, out, flt, ext,
                     _media(video_mode, fps, keep_rec709, bit_depth, jpeg_quality_95, start_time, end_time))


# ------------------------------------------------------------------------------------------------
# the planner proper (PC:593-980)
# ------------------------------------------------------------------------------------------------
@dataclass
class _Optics:
    focal_mm: float
    focal_35: Optional[float]
    hfov: float
    vfov: float
    sensor_w: float


def _resolve_optics(args, size: int) -> _Optics:
    sensor_w = parse_sensor(args.sensor_mm)
    dims = parse_sensor_dimensions(args.sensor_mm)
    sensor_long = max(dims) if dims else sensor_w
    sensor_h = float(dims[1]) if len(dims) >= 2 else sensor_w
    if sensor_h <= 0:
        sensor_h = None
    if args.hfov is not None:
        hfov = float(args.hfov)
        focal = focal_from_hfov_deg(hfov, sensor_w)
    else:
        focal = float(args.focal_mm)
        hfov = fov_from_focal_mm(focal, sensor_w)
    focal_35 = None
    if sensor_long and sensor_long > 0 and abs(sensor_long - 36.0) > 1e-6:
        focal_35 = focal * (36.0 / sensor_long)
    if sensor_h and focal > 1e-6:
        vfov = clamp(math.degrees(2.0 * math.atan(sensor_h / (2.0 * focal))), 1.0, 179.9)
    else:
        vfov = v_fov_from_hfov(hfov, size, size)
    return _Optics(focal, focal_35, hfov, vfov, sensor_w)


@dataclass
class _CamEdits:
    extras: Dict[int, List[float]] = field(default_factory=dict)
    dropped: Set[int] = field(default_factory=set)
    set_abs: Dict[int, float] = field(default_factory=dict)
    set_delta: Dict[int, float] = field(default_factory=dict)
    set_extra_abs: Dict[Tuple[int, str], float] = field(default_factory=dict)
    set_extra_delta: Dict[Tuple[int, str], float] = field(default_factory=dict)

    def pitch_for(self, slot: int, base_pitch: float, suffix: Optional[str] = None) -> float:
        p = base_pitch
        if suffix:
            key = (slot, suffix)
            if key in self.set_extra_abs:
                p = float(self.set_extra_abs[key])
            elif slot in self.set_abs:
                p = float(self.set_abs[slot])
            if key in self.set_extra_delta:
                p += float(self.set_extra_delta[key])
            elif slot in self.set_delta:
                p += float(self.set_delta[slot])
            return p
        if slot in self.set_abs:
            p = float(self.set_abs[slot])
        if slot in self.set_delta:
            p += float(self.set_delta[slot])
        return p


def _collect_edits(args, preset: Preset) -> _CamEdits:
    ed = _CamEdits()
    ed.extras = parse_addcam_spec(args.addcam, args.addcam_deg)
    ed.dropped = parse_delcam_spec(args.delcam)
    user_add = bool(str(getattr(args, "addcam", "")).strip()) or bool(getattr(args, "addcam_explicit", False))
    user_del = bool(str(getattr(args, "delcam", "")).strip()) or bool(getattr(args, "delcam_explicit", False))
    if preset.auto_del and not user_del:
        ed.dropped.update(letter_to_index1(ch) for ch in preset.auto_del)
    if preset.auto_add and not user_add:
        deg = float(args.addcam_deg)
        for ch in preset.auto_add:
            have = ed.extras.setdefault(letter_to_index1(ch), [])
            for want in (+deg, -deg):
                if not any(abs(v - want) < 1e-6 for v in have):
                    have.append(want)
    ed.dropped.update(letter_to_index1(ch) for ch in preset.hard_del)
    ed.set_abs, ed.set_delta, ed.set_extra_abs, ed.set_extra_delta = parse_setcam_spec(args.setcam, args.addcam_deg)
    return ed


def _view_id_from_name(out_name: str, stem: str, video_mode: bool) -> str:
    out_stem = pathlib.Path(out_name).stem
    if video_mode and out_stem.startswith(f"{stem}_%07d_"):
        return out_stem[len(stem) + 6:]
    if out_stem.startswith(f"{stem}_"):
        return out_stem[len(stem) + 1:]
    return out_stem


def build_view_jobs(args, files: List[pathlib.Path], out_dir: pathlib.Path, stop_event=None) -> BuildResult:
    """Plan every (source, view) job.  Pure planning: no I/O, mutates args the way the reference does."""
    explicit = {k: bool(getattr(args, f"{k}_explicit", False)) for k in ("size", "hfov", "focal_mm")}
    video_mode = bool(getattr(args, "input_is_video", False))
    media = MediaOptions(
        video_mode=video_mode, fps=getattr(args, "fps", None), keep_rec709=bool(getattr(args, "keep_rec709", False)),
        bit_depth=int(getattr(args, "video_bit_depth", 8)), jpeg_quality_95=args.jpeg_quality_95,
        start_time=getattr(args, "start", None), end_time=getattr(args, "end", None))

    add_top = bool(getattr(args, "add_top", False)) or bool(getattr(args, "add_topdown", False))
    add_bottom = bool(getattr(args, "add_bottom", False)) or bool(getattr(args, "add_topdown", False))
    args.add_top, args.add_bottom = add_top, add_bottom

    preset = PRESETS[args.preset]
    if preset.force_count is not None:
        if preset.announce_count and args.count != preset.force_count:
            print(f"[INFO] preset '{args.preset}' forces count={preset.force_count}")
        args.count = preset.force_count
    if preset.size is not None and not explicit["size"]:
        args.size = preset.size
    if preset.focal_mm is not None and not explicit["hfov"] and not explicit["focal_mm"]:
        args.focal_mm = preset.focal_mm

    edits = _collect_edits(args, preset)
    size = int(args.size)
    optics = _resolve_optics(args, size)

    pair_size, pair_fov = size, optics.hfov
    if preset.fisheye_pair:
        pair_size = size if explicit["size"] else 3600
        pair_fov = optics.hfov if explicit["hfov"] else 180.0

    count = int(args.count)
    if count <= 0:
        print("[ERR] --count must be >= 1", file=sys.stderr)
        sys.exit(1)
    step = 360.0 / count
    ext_dot = "." + args.ext.lower().lstrip(".")
    interp = map_interp_for_v360("bicubic")   # the reference hard-wires cubic (PC:730)

    jobs: List[Tuple[List[str], str, str]] = []
    specs: List[ViewSpec] = []
    taken: Set[str] = set()

    def emit(img, stem, tag, yaw, pitch, *, fisheye=False):
        pattern = f"{stem}_%07d_{tag}{ext_dot}" if video_mode else f"{stem}_{tag}{ext_dot}"
        if pattern in taken:
            return
        out_path = out_dir / pattern
        if fisheye:
            argv = _job_argv(args.ffmpeg, img, out_path,
                             f"v360=input=equirect:output=fisheye:w={pair_size}:h={pair_size}:yaw={yaw}:pitch={pitch}"
                             f":roll=0:d_fov={pair_fov}:interp={interp}", ext_dot, media)
            dims, fovs, proj = (pair_size, pair_size), (pair_fov, pair_fov), "equisolid"
        else:
            argv = _job_argv(args.ffmpeg, img, out_path,
                             f"v360=input=equirect:output=rectilinear:w={size}:h={size}:yaw={yaw}:pitch={pitch}"
                             f":roll=0:h_fov={optics.hfov}:v_fov={optics.vfov}:interp={interp}", ext_dot, media)
            dims, fovs, proj = (size, size), (optics.hfov, optics.vfov), "perspective"
        jobs.append((argv, img.name, pattern))
        taken.add(pattern)
        specs.append(ViewSpec(img, pattern, _view_id_from_name(pattern, stem, video_mode), yaw, pitch,
                              fovs[0], fovs[1], dims[0], dims[1], proj))

    for img in files:
        stem = img.stem
        pair: List[Tuple[str, float, float]] = []
        for slot0 in range(count):
            if stop_event is not None and stop_event.is_set():
                break
            slot = slot0 + 1
            tag = letter_tag(slot0)
            yaw = normalize_angle_deg(slot0 * step)
            pitch = 0.0
            if slot % 2 == 0 and not preset.fisheye_pair and preset.even_pitch is not None:
                pitch += float(preset.even_pitch)
            pitch = clamp(edits.pitch_for(slot, pitch), -90.0, 90.0)
            if preset.fisheye_pair:
                if slot in FISHEYE_PAIR_SLOTS:
                    pair.append((FISHEYE_PAIR_SLOTS[slot], yaw, pitch))
                continue
            if slot not in edits.dropped:
                emit(img, stem, tag, yaw, pitch)
            for delta in edits.extras.get(slot, ()):
                suffix = extra_suffix(delta, args.addcam_deg)
                p = edits.pitch_for(slot, clamp(pitch + delta, -90.0, 90.0), suffix=suffix)
                emit(img, stem, f"{tag}{suffix}", yaw, p)
        for tag, yaw, pitch in pair:
            emit(img, stem, tag, yaw, pitch, fisheye=True)
        next_slot0 = count
        for wanted, pole_pitch in ((add_top, 90.0), (add_bottom, -90.0)):
            if not wanted:
                continue
            tag = letter_tag(next_slot0)
            next_slot0 += 1
            p = edits.pitch_for(letter_to_index1(tag), clamp(pole_pitch, -90.0, 90.0))
            emit(img, stem, tag, 0.0, p)

    lines = _info_lines(args, jobs, video_mode, preset, optics, size, pair_size, pair_fov)
    return BuildResult(jobs, specs, optics.focal_mm, optics.focal_35, optics.hfov, optics.vfov, *lines)


def _info_lines(args, jobs, video_mode, preset, optics, size, pair_size, pair_fov):
    """The four user-facing summary lines (PC:914-967); RealityScan/Metashape read the focal values."""
    views_line = sensor_line = rs_line = ms_line = ""
    if not jobs:
        return views_line, sensor_line, rs_line, ms_line
    first_src = jobs[0][1]
    ref_stem = pathlib.Path(first_src).stem
    seen: List[str] = []
    for _argv, src_name, dst_name in jobs:
        if src_name != first_src:
            break
        vid = _view_id_from_name(dst_name, ref_stem, bool(getattr(args, "input_is_video", False)))
        if vid and vid not in seen:
            seen.append(vid)
    if not seen:
        return views_line, sensor_line, rs_line, ms_line
    n = len(seen)
    views_line = f"[INFO] View summary ({first_src}): {n} view{'s' if n != 1 else ''} - " + ", ".join(seen)
    if preset.fisheye_pair:
        views_line += f" | fisheye_fov={pair_fov:.1f}deg | size={pair_size}x{pair_size}"
        return views_line, sensor_line, rs_line, ms_line
    sensor_line = f"[INFO] Sensor={args.sensor_mm} mm | size={size}x{size}"
    focal_txt = f"focal length=  {optics.focal_mm:.3f} mm"
    if optics.focal_35 is not None:
        focal_txt += f" (35mm eq=  {optics.focal_35:.3f} mm)"
    rs_line = f"[INFO] For RealityScan: {focal_txt}"
    if size > 0:
        pixel_mm = optics.sensor_w / float(size)
        if pixel_mm > 0:
            ms_line = "[INFO] For Metashape: Precalibrated f=  {:.5f}  | pixel_size=  {:.4f} mm".format(
                optics.focal_mm / pixel_mm, pixel_mm)
    return views_line, sensor_line, rs_line, ms_line
