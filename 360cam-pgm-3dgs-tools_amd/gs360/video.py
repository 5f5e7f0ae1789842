"""Video inputs: ONE decode per video, frames resident in HBM, every view job samples the resident frames.

The reference plans one ffmpeg process per (video, view) (PC:830-836); each of those decodes the whole video again and
runs v360 on one thread ("N_views single-thread ffmpeg procs, each re-decoding the input").  Here the
first view job of a video starts a single decoder process -- the job's own ffmpeg command line with the v360 filter and
the encoder options removed and `format=rgb24 ... -f image2pipe -c:v ppm pipe:1` appended -- and a reader thread
uploads every frame to device memory through two pinned staging blocks per device (the pipe read of one frame overlaps
the H2D copy of the previous one; frames are dealt round-robin over the visible GPUs, so all views of a frame share a
device and there is no exchange step).  An 8K RGB frame is 88.5 MB: a 288 GB MI355X keeps > 2000 of them, i.e. the
whole 600-frame workload of BASELINE config 3 stays resident while the 12 view jobs run over it.  Longer videos stream:
past the budget (GS360_VIDEO_CACHE_GB per device) the frames every registered view job has consumed are retired and the
reader waits for the view jobs instead of failing (ffmpeg itself streams, PC:318, PC:746-749).

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
import time
import tempfile
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import capi
from .jobspec import JobSpec

# options that belong to the encoder / muxer of the per-view process and have no meaning for the shared decoder
_ENCODER_OPTIONS = {"-c:v", "-q:v", "-qmin", "-qmax", "-pix_fmt", "-huffman", "-colorspace", "-color_primaries",
                    "-color_trc", "-start_number", "-frame_pts", "-threads", "-frames:v", "-loglevel"}
_DECODER_OPTIONS = {"-ss", "-to", "-t", "-vsync", "-fps_mode", "-r"}
_BUDGET_BYTES = int(float(os.environ.get("GS360_VIDEO_CACHE_GB", "200")) * (1 << 30))
_EVENT_BASE = 6                            # HIP events 6 / 7 of the upload slot mark the two staging blocks' copies


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
class SharedBudget:
    """Device memory all live sessions of one engine may hold together (GS360_VIDEO_CACHE_GB per device x devices).  A session
    superseded by a second decode of the same video stays alive until its last job leaves; with a budget of its own each, two
    sessions could ask for twice the budget of a device."""

    def __init__(self, total):
        self.total = int(total)
        self.used = 0
        self.lock = threading.Lock()

    def add(self, n):
        with self.lock:
            self.used += n


# View jobs take the frames of a video in WINDOWS of this many (one batched launch per window and device: ring families -- the
# `full360coverage` preset, PC:616-680 -- reach the source-major kernel from four frames per call, csrc/gs360_capi.hip); frames are dealt to
# the devices in blocks of a window, so that a window's frames sit on one device.  GS360_VIDEO_WINDOW=1: one frame per launch (A/B).
_WINDOW = max(1, min(16, int(os.environ.get("GS360_VIDEO_WINDOW", "4"))))
_WINDOW_WAIT_S = 0.05                             # longest a view job waits for the frames behind the first one of its window


class VideoSession:
    """Decoded frames of one video on the engine's devices, streamed: the reader keeps at most `budget` bytes resident, view
    jobs walk the frames with a cursor each, and a frame every registered job has passed is retired when the reader needs
    the room (before that it stays, so that a view job that joins later still finds the video from its first frame).  The
    reader blocks -- back-pressure on the decoder's pipe -- while nothing can be retired.  A job that arrives after frames
    were retired cannot join (join() returns None): the engine starts a second session for it and its fellow late-comers.
    Thread-safe."""

    def __init__(self, states, plan: DecodePlan, stop_event=None, register_proc=None, budget=None, shared=None):
        self.states = states
        self.plan = plan
        self.stop_event = stop_event
        self.register_proc = register_proc       # callable(proc, add: bool): lets the caller's cancel handler see the decoder
        per_device = _BUDGET_BYTES if budget is None else budget
        self.budget = per_device * max(1, len(states))    # the budget is per device; frames are dealt round-robin
        self.shared = shared if shared is not None else SharedBudget(self.budget)   # `shared`: the engine's, one for all sessions
        self.frames = {}                          # k -> (state, DeviceBuffer, H, W, dtype), k in [first, count)
        self.first = 0                            # frames below `first` have been retired
        self.count = 0                            # frames published so far
        self.bytes = 0
        self.peak_bytes = 0
        self.retired = 0
        self.finished = False
        self.error: Optional[str] = None
        self.cond = threading.Condition()
        self.cursors = {}                         # job token -> index of the frame the job is working on (all below are done)
        self._next_token = 0
        self.active_jobs = 0
        self.done_jobs = 0
        self.closing = False
        self.proc = None
        self.thread = threading.Thread(target=self._reader, name="gs360-video-decode", daemon=True)
        self.thread.start()

    # reader thread ---------------------------------------------------------------------------------------------
    def _make_room(self, nbytes):
        """Called by the reader with self.cond held: retire passed frames until `nbytes` more fit, waiting for the view jobs
        to advance when nothing can go.  A single frame larger than the whole budget is admitted alone."""
        while self.shared.used + nbytes > self.shared.total and self.bytes > 0:
            if not self._retire_one():
                if self.closing or (self.stop_event is not None and self.stop_event.is_set()):
                    raise PpmError("cancelled")
                self.cond.wait(timeout=0.25)

    def _retire_one(self) -> bool:
        """(self.cond held) free the oldest frame if every registered job has passed it"""
        low = min(self.cursors.values()) if self.cursors else self.first      # no job registered: nothing has been passed
        if self.first >= low or self.first not in self.frames:
            return False
        st, buf, h, w, dt = self.frames.pop(self.first)
        n = h * w * 3 * np.dtype(dt).itemsize
        self.bytes -= n
        self.shared.add(-n)
        self.first += 1
        self.retired += 1
        if st.ctx.handle:
            st.ctx.free(buf)
        return True

    def _alloc(self, st, nbytes):
        """device block for the next frame; when the device itself is out of memory (other users of the GPU, a budget set too
        high) the reader does what it does at the budget: retire a passed frame, or wait for the view jobs, and try again"""
        while True:
            try:
                return st.ctx.alloc(nbytes)
            except capi.Gs360Error as exc:
                if exc.code != -5:
                    raise
                with self.cond:
                    if self.bytes == 0:
                        raise                     # nothing of ours to give back: the frame does not fit at all
                    if not self._retire_one():
                        if self.closing or (self.stop_event is not None and self.stop_event.is_set()):
                            raise PpmError("cancelled") from exc
                        self.cond.wait(timeout=0.25)

    def _publish(self, st, buf, h, w, fdtype, nbytes):
        with self.cond:
            self.frames[self.count] = (st, buf, h, w, fdtype)
            self.count += 1
            self.bytes += nbytes
            self.shared.add(nbytes)
            self.peak_bytes = max(self.peak_bytes, self.bytes)
            self.cond.notify_all()

    def _reader(self):
        pinned = {}                               # (device, nbytes) -> [two pinned blocks]
        turn = {}                                 # device -> which of its two blocks the next frame takes
        pending = None                            # (state, event index, publish args) of the upload still in flight
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
                st = self.state_for(k)
                # two pinned staging blocks per device: the pipe read of frame k overlaps the H2D copy of frame k-1
                pair = pinned.get((id(st), nbytes))
                if pair is None:
                    pair = pinned[(id(st), nbytes)] = [st.ctx.pinned(nbytes), st.ctx.pinned(nbytes)]
                which = turn.get(id(st), 0)        # the device's two staging blocks in turn
                turn[id(st)] = which ^ 1
                stage = pair[which]
                host = np.frombuffer(stage.view, dtype=np.uint8, count=nbytes)      # the pinned block as an array
                read_exact_into(out, memoryview(host))
                if pending is not None:           # frame k-1 has had this whole pipe read to land
                    pst, pev, pargs = pending
                    pst.ctx.event_sync(pst.upload_slot, pev)
                    self._publish(*pargs)
                    pending = None
                with self.cond:
                    self._make_room(nbytes)
                buf = self._alloc(st, nbytes)
                ev = _EVENT_BASE + which
                try:
                    with st.upload_lock:          # the engine's still-image uploads share this stream (engine.resident_frame)
                        st.ctx.upload(buf, host, slot=st.upload_slot, sync=False)
                        if fdtype == np.uint16:           # PPM samples are big-endian: swapped on the device, behind the copy
                            st.ctx.bswap16(buf, nbytes // 2, slot=st.upload_slot)
                        st.ctx.event_record(st.upload_slot, ev)
                except Exception:
                    st.ctx.free(buf)              # not published, not pending: nobody else would release it
                    raise
                pending = (st, ev, (st, buf, h, w, fdtype, nbytes))
                k += 1
            if pending is not None:
                pst, pev, pargs = pending
                pst.ctx.event_sync(pst.upload_slot, pev)
                self._publish(*pargs)
                pending = None
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
            if pending is not None:               # the copy may still be running: let it finish before its blocks are freed
                try:
                    pending[0].ctx.event_sync(pending[0].upload_slot, pending[1])
                    if pending[0].ctx.handle:
                        pending[0].ctx.free(pending[2][1])
                except Exception:
                    pass
        finally:
            for pair in pinned.values():
                for stage in pair:
                    stage.free()
            if errlog is not None:
                errlog.close()
            if self.proc is not None and self.register_proc:
                self.register_proc(self.proc, False)
            with self.cond:
                self.finished = True
                self.cond.notify_all()

    # view jobs -------------------------------------------------------------------------------------------------
    def join(self):
        """Register a view job -> token, or None when the video's first frames have already been retired (a job can only
        walk the video from its beginning)."""
        with self.cond:
            if self.first > 0:
                return None
            tok = self._next_token
            self._next_token += 1
            self.cursors[tok] = 0
            self.active_jobs += 1
            return tok

    def leave(self, token):
        with self.cond:
            if self.cursors.pop(token, None) is not None:
                self.active_jobs -= 1
                self.done_jobs += 1
            self.cond.notify_all()

    def state_for(self, k: int):
        """the device frame k is (or will be) resident on: blocks of _WINDOW frames round-robin"""
        return self.states[(k // _WINDOW) % len(self.states)]

    def advance(self, token, k: int):
        """the job `token` declares frames < k done (without asking for one)"""
        with self.cond:
            if token in self.cursors and k > self.cursors[token]:
                self.cursors[token] = k
                self.cond.notify_all()

    def window(self, token, k0: int, limit: int = _WINDOW):
        """frames k0, k0 + 1, ... of ONE device for one batched launch: waits for frame k0 like frame(), then takes its successors -- at
        most `limit`, never past the end of k0's block -- as (state, [DeviceBuffer], H, W, dtype); None after the last frame.  For the
        rest of the window it waits only while the decoder is still delivering, and at most _WINDOW_WAIT_S in all: the reader may be
        blocked at its budget until this very job moves on (a bounded wait cannot deadlock; a decoder slower than that is the
        bottleneck whatever the window)."""
        first = self.frame(token, k0)
        if first is None:
            return None
        st, buf, h, w, dt = first
        end = min(k0 + max(1, limit), (k0 // _WINDOW + 1) * _WINDOW)
        bufs = [buf]
        deadline = time.monotonic() + _WINDOW_WAIT_S
        with self.cond:
            while self.count < end and not self.finished and not self.closing:
                left = deadline - time.monotonic()
                if left <= 0 or (self.stop_event is not None and self.stop_event.is_set()):
                    break
                self.cond.wait(timeout=left)
            for k in range(k0 + 1, min(end, self.count)):
                fr = self.frames.get(k)
                if fr is None or fr[0] is not st or fr[2:] != (h, w, dt):
                    break
                bufs.append(fr[1])
        return st, bufs, h, w, dt

    def frame(self, token, k: int):
        """k-th decoded frame as (state, DeviceBuffer, H, W, dtype) for the job `token`, which thereby declares frames < k
        done; None after the last one.  Raises on decoder failure."""
        with self.cond:
            if token in self.cursors and k > self.cursors[token]:
                self.cursors[token] = k
                self.cond.notify_all()            # the reader may be waiting for exactly this
            while k >= self.count and not self.finished:
                self.cond.wait(timeout=0.25)
                if self.stop_event is not None and self.stop_event.is_set():
                    return None
            if k < self.count:
                if k < self.first:
                    raise PpmError("frame {} was retired before this view job asked for it".format(k))
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
        with self.cond:
            self.closing = True
            self.cursors.clear()
            self.cond.notify_all()
        self.thread.join(timeout=5.0)
        with self.cond:
            for st, buf, _h, _w, _dt in self.frames.values():
                if st.ctx.handle:
                    st.ctx.free(buf)
            self.frames = {}
            self.shared.add(-self.bytes)
            self.bytes = 0
