"""Video inputs: ONE decode per video, frames resident in HBM, every view job samples the resident frames.

The reference plans one ffmpeg process per (video, view) (PC:830-836); each of those decodes the whole video again and
runs v360 on one thread ("N_views single-thread ffmpeg procs, each re-decoding the input").  Here the
first view job of a video starts a single decoder process -- the job's own ffmpeg command line with the v360 filter and
the encoder options removed and `format=rgb24 ... -f image2pipe -c:v ppm pipe:1` appended -- and a reader thread
uploads every frame to device memory (frames are dealt round-robin over the visible GPUs, so all views of a frame share
a device and there is no exchange step).  An 8K RGB frame is 88.5 MB: a 288 GB MI355X keeps > 2000 of them, i.e. the
whole 600-frame workload of BASELINE config 3 stays resident while the 12 view jobs run over it.

Two argv shapes are understood (anything else returns None and the caller falls back to the reference's subprocess):
  * the planner's:        -ss S -i V -to T -vf fps=F,colorspace=...,v360=... -vsync vfr -start_number 0 ... out_%07d_X.png
  * the GUI's selection:  -copyts -i V -ss S -to T -vf select='eq(n\\,i)+...',colorspace=...,v360=... -frame_pts 1 ...
    (gs360_GUI.py:19081-19148): output numbers are the selected source frame indices.
Geometry/sampling are the engine's (EQ-SPEC v1 on RGB); ffmpeg's v360 interpolates the YUV planes, so outputs are not
bit-comparable with the reference's -- parity at this seam is unpinned (no ffmpeg in the build or test images).
"""
import os
import re
import subprocess
import tempfile
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .jobspec import JobSpec

# options that belong to the encoder / muxer of the per-view process and have no meaning for the shared decoder
_ENCODER_OPTIONS = {"-c:v", "-q:v", "-qmin", "-qmax", "-pix_fmt", "-huffman", "-colorspace", "-color_primaries",
                    "-color_trc", "-start_number", "-frame_pts", "-threads", "-frames:v", "-loglevel"}
_DECODER_OPTIONS = {"-ss", "-to", "-t", "-vsync", "-fps_mode", "-r"}
_BUDGET_BYTES = int(float(os.environ.get("GS360_VIDEO_CACHE_GB", "200")) * (1 << 30))


@dataclass(frozen=True)
class DecodePlan:
    argv: Tuple[str, ...]                 # the decoder command line
    key: Tuple                            # identical for every view job of the same video + decode settings
    numbers: Optional[Tuple[int, ...]]    # output number of the k-th decoded frame; None = start_number + k
    start_number: int


def _select_indices(flt: str) -> Optional[List[int]]:
    """select='eq(n\\,3)+eq(n\\,17)' -> [3, 17]; None when the expression is anything else."""
    body = flt[len("select="):].strip()
    if len(body) >= 2 and body[0] == "'" and body[-1] == "'":
        body = body[1:-1]
    out = []
    for clause in body.split("+"):
        m = re.fullmatch(r"eq\(n\\?,(\d+)\)", clause.strip())
        if not m:
            return None
        out.append(int(m.group(1)))
    return out


def build_decode_plan(job: JobSpec) -> Optional[DecodePlan]:
    """Decoder command for a video view job, or None when the argv is not one of the two understood shapes."""
    if job.is_still_image or job.filters_after:
        return None
    if "%" not in job.dst.name:
        return None
    pix = job.options.get("-pix_fmt", "")
    if pix and pix not in ("rgb24", "yuvj444p", "rgb48le"):
        return None
    deep = pix == "rgb48le"                # bit depth > 8 (PC:343-347): frames travel as 16-bit PPM, views are uint16
    known = _ENCODER_OPTIONS | _DECODER_OPTIONS
    for tok in job.input_options[0::2] + job.output_options[0::2]:
        if tok not in known:
            return None
    numbers = None
    pre = [f for f in job.filters if f not in job.filters_after]
    for f in pre:
        if f.startswith("select="):
            idx = _select_indices(f)
            if idx is None:
                return None
            numbers = tuple(sorted(set(idx)))
    if numbers is not None and job.options.get("-frame_pts") != "1":
        numbers = None                     # select without frame_pts: plain sequential numbering
    if numbers is None and "-frame_pts" in job.options:
        return None                        # pts-driven numbering without a select list cannot be reproduced
    if numbers is not None and any(k in job.options for k in ("-ss", "-to", "-t")):
        # The GUI keeps -ss/-to on the OUTPUT side next to -copyts (gs360_GUI.py:19094-19148): ffmpeg then drops every
        # selected frame whose timestamp lies outside [S, T], and -frame_pts still names the survivors by their own
        # index.  The PPM pipe carries no timestamps, so which selected indices survive cannot be known here without
        # probing the stream; pairing the k-th decoded frame with the k-th selected index would misnumber every output.
        # Such jobs stay on the reference's per-view subprocess path.
        return None
    argv = [job.program, "-hide_banner", "-loglevel", "error", "-nostdin"]
    argv += [f for f in job.flags if f == "-copyts"]

    def keep(tokens: Sequence[str]) -> List[str]:
        out = []
        for k, v in zip(tokens[0::2], tokens[1::2]):
            if k in _DECODER_OPTIONS:
                out += [k, v]
        return out
    argv += keep(job.input_options) + ["-i", str(job.src)] + keep(job.output_options)
    argv += ["-vf", ",".join(pre + ["format=rgb48be" if deep else "format=rgb24"]), "-an", "-f", "image2pipe", "-c:v", "ppm", "pipe:1"]
    key = (str(job.src), tuple(argv[1:]))
    start = int(job.options.get("-start_number", "0")) if numbers is None else 0
    return DecodePlan(tuple(argv), key, numbers, start)


def output_path(job: JobSpec, plan: DecodePlan, k: int) -> str:
    n = plan.numbers[k] if plan.numbers is not None and k < len(plan.numbers) else plan.start_number + k
    return str(job.dst) % n


# ---- PPM stream ------------------------------------------------------------------------------------------------
class PpmError(RuntimeError):
    pass


def read_ppm_header(stream) -> Optional[Tuple[int, int, int]]:
    """Reads one binary-PPM header from a buffered byte stream -> (width, height, maxval), None at a clean EOF."""
    tokens, cur, in_comment = [], b"", False
    while len(tokens) < 4:
        ch = stream.read(1)
        if not ch:
            if not tokens and not cur:
                return None
            raise PpmError("truncated PPM header")
        if in_comment:
            in_comment = ch != b"\n"
            continue
        if ch == b"#":
            in_comment = True
        elif ch.isspace():
            if cur:
                tokens.append(cur)
                cur = b""
        else:
            cur += ch
            if len(cur) > 16:
                raise PpmError("not a PPM stream")
    if tokens[0] != b"P6":
        raise PpmError("expected a binary PPM (P6) frame, got {!r}".format(tokens[0][:8]))
    try:
        w, h, maxval = (int(t) for t in tokens[1:])
    except ValueError as exc:
        raise PpmError("bad PPM header") from exc
    if w < 1 or h < 1 or w > 65535 or h > 65535 or maxval < 1 or maxval > 65535:
        raise PpmError("bad PPM geometry {}x{} maxval {}".format(w, h, maxval))
    return w, h, maxval


def read_exact_into(stream, mv: memoryview) -> None:
    got = 0
    while got < len(mv):
        n = stream.readinto(mv[got:])
        if not n:
            raise PpmError("truncated PPM frame ({} of {} bytes)".format(got, len(mv)))
        got += n


# ---- session ---------------------------------------------------------------------------------------------------
class VideoSession:
    """Decoded frames of one video, resident on the engine's devices.  Thread-safe; view jobs call frame(k)."""

    def __init__(self, states, plan: DecodePlan, stop_event=None, register_proc=None, budget=_BUDGET_BYTES):
        self.states = states
        self.plan = plan
        self.stop_event = stop_event
        self.register_proc = register_proc       # callable(proc, add: bool): lets the caller's cancel handler see the decoder
        self.budget = budget * max(1, len(states))    # the budget is per device; frames are dealt round-robin
        self.frames = []                          # (state, DeviceBuffer, H, W)
        self.bytes = 0
        self.finished = False
        self.error: Optional[str] = None
        self.cond = threading.Condition()
        self.active_jobs = 0
        self.done_jobs = 0
        self.proc = None
        self.thread = threading.Thread(target=self._reader, name="gs360-video-decode", daemon=True)
        self.thread.start()

    # reader thread ---------------------------------------------------------------------------------------------
    def _reader(self):
        pinned = {}
        errlog = None
        try:
            # stderr goes to an unnamed temporary file, not a pipe: a damaged stream can make ffmpeg print more than a pipe
            # buffer holds while this thread is blocked on stdout, which would stall decoder and view jobs alike
            errlog = tempfile.TemporaryFile()
            try:
                self.proc = subprocess.Popen(list(self.plan.argv), stdout=subprocess.PIPE, stderr=errlog, bufsize=1 << 20)
            except OSError as exc:
                raise PpmError("{}: {}".format(self.plan.argv[0], exc)) from exc
            if self.register_proc:
                self.register_proc(self.proc, True)
            out = self.proc.stdout
            k = 0
            while True:
                if self.stop_event is not None and self.stop_event.is_set():
                    raise PpmError("cancelled")
                head = read_ppm_header(out)
                if head is None:
                    break
                w, h, maxval = head
                if maxval not in (255, 65535):
                    raise PpmError("decoder delivered {}-level samples; the engine takes 8- or 16-bit frames".format(maxval + 1))
                fdtype = np.uint16 if maxval == 65535 else np.uint8
                nbytes = w * h * 3 * np.dtype(fdtype).itemsize
                if self.bytes + nbytes > self.budget:
                    raise PpmError("decoded frames exceed the HBM budget of {:.0f} GB per GPU (GS360_VIDEO_CACHE_GB); lower "
                                   "--fps, cut the range with --start/--end, or use --engine ffmpeg".format(
                                       self.budget / len(self.states) / (1 << 30)))
                st = self.states[k % len(self.states)]
                stage = pinned.get((id(st), nbytes))
                if stage is None:
                    stage = pinned[(id(st), nbytes)] = st.ctx.pinned(nbytes)
                host = np.frombuffer(stage.view, dtype=np.uint8, count=nbytes)      # the pinned block as an array
                read_exact_into(out, memoryview(host))
                if fdtype == np.uint16:
                    host.view(np.uint16).byteswap(inplace=True)       # PPM samples are big-endian
                buf = st.ctx.alloc(nbytes)
                st.ctx.upload(buf, host, slot=st.upload_slot, sync=True)
                with self.cond:
                    self.frames.append((st, buf, h, w, fdtype))
                    self.bytes += nbytes
                    self.cond.notify_all()
                k += 1
            rc = self.proc.wait()
            if rc != 0:
                errlog.seek(max(0, errlog.seek(0, os.SEEK_END) - 2000))
                text = errlog.read().decode(errors="ignore").strip()
                raise PpmError("decoder exited with code {}: {}".format(rc, text[-400:]))
        except Exception as exc:  # noqa: BLE001  (reported to every waiting view job)
            with self.cond:
                self.error = str(exc)
            if self.proc is not None and self.proc.poll() is None:
                try:
                    self.proc.kill()
                except Exception:
                    pass
        finally:
            for stage in pinned.values():
                stage.free()
            if errlog is not None:
                errlog.close()
            if self.proc is not None and self.register_proc:
                self.register_proc(self.proc, False)
            with self.cond:
                self.finished = True
                self.cond.notify_all()

    # view jobs -------------------------------------------------------------------------------------------------
    def frame(self, k: int):
        """k-th decoded frame as (state, DeviceBuffer, H, W, dtype); None after the last one.  Raises on decoder failure."""
        with self.cond:
            while k >= len(self.frames) and not self.finished:
                self.cond.wait(timeout=0.25)
                if self.stop_event is not None and self.stop_event.is_set():
                    return None
            if k < len(self.frames):
                return self.frames[k]
            if self.error:
                raise PpmError(self.error)
            return None

    def close(self):
        if self.proc is not None and self.proc.poll() is None:
            try:
                self.proc.kill()
            except Exception:
                pass
        self.thread.join(timeout=5.0)
        with self.cond:
            for st, buf, _h, _w, _dt in self.frames:
                if st.ctx.handle:
                    st.ctx.free(buf)
            self.frames = []
            self.bytes = 0
