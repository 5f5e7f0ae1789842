"""Process-wide execution engine behind run_one(): frames x views fanned out over the visible GPUs.

Unit of work = (frame, view), exactly the reference's job granularity (PC:830-836).  Sharding is by FRAME:
all views of one source image run on the same device, so the decoded frame is uploaded once and stays
resident in HBM (an LRU of device frames per GPU); there is no exchange step and therefore no collective.
Concurrency comes from the caller's thread pool (PC:1049 / gs360_GUI.py:19297): each call takes one of the
device's stream slots.
"""
import collections
import itertools
import os
import threading
import zlib

import numpy as np

from . import capi, imageio, video
from .jobspec import JobSpec

_FRAME_CACHE_BYTES = int(os.environ.get("GS360_FRAME_CACHE_MB", "4096")) << 20
_SLOTS_PER_DEVICE = 4


class _DeviceState:
    def __init__(self, device):
        # render slots 0.._SLOTS_PER_DEVICE-1 plus ONE upload stream that _render never uses: a frame upload's sync then
        # waits for that copy only, not for kernels and downloads queued by view jobs (and the reverse)
        self.ctx = capi.Context(device=device, n_slots=_SLOTS_PER_DEVICE + 1)
        self.upload_slot = _SLOTS_PER_DEVICE
        self.upload_lock = threading.Lock()
        self.frames = collections.OrderedDict()   # key -> [DeviceBuffer, H, W, C, users]
        self.frame_bytes = 0
        self.lock = threading.Lock()              # guards the LRU bookkeeping
        self.key_locks = {}                       # key -> lock: one decode+upload per frame
        self.slot_cycle = itertools.cycle(range(_SLOTS_PER_DEVICE))
        self.out_bufs = [None] * _SLOTS_PER_DEVICE   # grow-only per-slot output buffers (no hipMalloc/hipFree per job)

    def out_buffer(self, slot, nbytes):
        buf = self.out_bufs[slot]
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                self.ctx.free(buf)
            buf = self.out_bufs[slot] = self.ctx.alloc(max(nbytes, 1 << 20))
        return buf


class Engine:
    def __init__(self, devices=None):
        n = capi.device_count()
        if n <= 0:
            raise capi.Gs360Error(-3, "no MI355X visible: the gs360 engine has no CPU fallback "
                                      "(use --engine ffmpeg to run the reference's ffmpeg path)")
        want = os.environ.get("GS360_DEVICES")
        if devices is None and want:
            devices = [int(t) for t in want.split(",") if t.strip()]
        self.devices = list(devices) if devices is not None else list(range(n))
        self.states = [_DeviceState(d) for d in self.devices]
        self._warned_cubic = False
        self.videos = {}                          # DecodePlan.key -> video.VideoSession
        self.videos_lock = threading.Lock()

    def close(self):
        with self.videos_lock:
            sessions, self.videos = list(self.videos.values()), {}
        for sess in sessions:
            sess.close()
        for st in self.states:
            st.ctx.close()
        self.states = []

    # -- sharding ---------------------------------------------------------------------------------
    def device_for(self, src_path) -> int:
        """frame -> device index (stable hash of the source path: all views of a frame share a device)."""
        return zlib.crc32(os.fsencode(str(src_path))) % len(self.states)

    # -- frame residency --------------------------------------------------------------------------
    def _frame_key(self, path):
        st = os.stat(path)
        return (str(path), st.st_mtime_ns, st.st_size)

    def resident_frame(self, st: _DeviceState, path):
        key = self._frame_key(path)
        with st.lock:
            hit = st.frames.get(key)
            if hit is not None:
                st.frames.move_to_end(key)
                hit[4] += 1
                return hit
            klock = st.key_locks.setdefault(key, threading.Lock())
        with klock:
            with st.lock:
                hit = st.frames.get(key)
                if hit is not None:
                    hit[4] += 1
                    return hit
            img = imageio.read_image(path)
            H, W, C = img.shape
            if C not in (1, 3, 4):
                raise capi.Gs360Error(-1, f"{path}: unsupported channel count {C}")
            buf = st.ctx.alloc(img.nbytes)
            with st.upload_lock:
                st.ctx.upload(buf, img, slot=st.upload_slot, sync=True)
            entry = [buf, H, W, C, 1]             # last field: users currently holding the frame
            with st.lock:
                st.frames[key] = entry
                st.frame_bytes += img.nbytes
                for k in list(st.frames):         # evict least-recently-used frames nobody is reading
                    if st.frame_bytes <= _FRAME_CACHE_BYTES:
                        break
                    old = st.frames[k]
                    if k == key or old[4] > 0:
                        continue
                    del st.frames[k]
                    st.frame_bytes -= old[1] * old[2] * old[3]
                    st.ctx.free(old[0])
                st.key_locks.pop(key, None)
            return entry

    def release_frame(self, st: _DeviceState, entry):
        with st.lock:
            entry[4] -= 1

    # -- one job ----------------------------------------------------------------------------------
    def _interp_for(self, job: JobSpec) -> int:
        if job.interp in ("linear", "bilinear", "line"):
            interp = capi.INTERP_LINEAR
        elif job.interp in ("cubic", "bicubic"):
            interp = capi.INTERP_CUBIC
        else:
            interp = capi.INTERP_CUBIC
            if not self._warned_cubic:
                self._warned_cubic = True
                print(f"[INFO] gs360 engine: v360 interp={job.interp} is not implemented; sampling with cubic", flush=True)
        interp_env = os.environ.get("GS360_INTERP")           # additive override: linear | cubic
        if interp_env in ("linear", "cubic"):
            interp = capi.INTERP_LINEAR if interp_env == "linear" else capi.INTERP_CUBIC
        return interp

    def _view_for(self, job: JobSpec):
        """-> (View, flags).  rectilinear: v360's h_fov/v_fov; fisheye (the fisheyeXY preset, PC:351-414): v360 takes a
        diagonal field of view d_fov and spreads it over the image diagonal (equidistant)."""
        if job.input_projection != "equirect" or job.output_projection not in ("rectilinear", "fisheye"):
            raise capi.Gs360Error(-4, f"v360 {job.input_projection}->{job.output_projection} is not implemented by "
                                      "the HIP engine (equirect->rectilinear|fisheye only); use --engine ffmpeg")
        if abs(job.fnum("roll", 0.0)) > 1e-12:
            raise capi.Gs360Error(-4, "roll != 0 is not implemented by the HIP engine")
        if job.output_projection == "fisheye":
            w, h = job.width, job.height
            if "d_fov" in job.v360:
                diag = float(np.hypot(w, h))
                hfov, vfov = job.fnum("d_fov") * w / diag, job.fnum("d_fov") * h / diag
            else:
                hfov, vfov = job.fnum("h_fov"), job.fnum("v_fov")
            return capi.View.make(job.fnum("yaw"), job.fnum("pitch"), hfov, vfov, w, h), capi.EQ_FISHEYE_OUT
        return capi.View.make(job.fnum("yaw"), job.fnum("pitch"), job.fnum("h_fov"), job.fnum("v_fov"), job.width, job.height), 0

    def _render(self, st: _DeviceState, buf, H, W, C, view, interp, flags=0):
        with st.lock:
            slot = next(st.slot_cycle)
        out_bytes = view.height * view.width * C
        with st.ctx.slot_locks[slot]:
            dst = st.out_buffer(slot, out_bytes)
            st.ctx.equirect_views_dev([buf], W, H, C, [view], [dst], slot=slot, interp=interp, flags=flags)
            return st.ctx.download(dst, (view.height, view.width, C), slot=slot)

    def run_job(self, job: JobSpec):
        """Execute one (frame, view) job; returns the output array after writing job.dst."""
        view, flags = self._view_for(job)
        interp = self._interp_for(job)
        st = self.states[self.device_for(job.src)]
        entry = self.resident_frame(st, job.src)
        try:
            buf, H, W, C = entry[:4]
            out = self._render(st, buf, H, W, C, view, interp, flags)
        finally:
            self.release_frame(st, entry)
        imageio.write_image(job.dst, out, jpeg_q=job.jpeg_q)
        return out

    # -- video: one decode, frames resident in HBM, every view job walks them (gs360/video.py) -------------------
    def _video_session(self, plan, stop_event, register_proc):
        with self.videos_lock:
            sess = self.videos.get(plan.key)
            if sess is None:
                for key in [k for k, s in self.videos.items() if s.active_jobs == 0 and s.finished]:
                    self.videos.pop(key).close()          # idle sessions of other videos give their memory back
                sess = self.videos[plan.key] = video.VideoSession(self.states, plan, stop_event, register_proc)
            sess.active_jobs += 1
            return sess

    def run_video_job(self, job: JobSpec, plan, stop_event=None, register_proc=None, expected_jobs=None) -> int:
        """All frames of one view of a video; returns the number of frames written."""
        view, flags = self._view_for(job)
        interp = self._interp_for(job)
        sess = self._video_session(plan, stop_event, register_proc)
        written = 0
        try:
            while True:
                fr = sess.frame(written)
                if fr is None:
                    break
                st, buf, H, W = fr
                out = self._render(st, buf, H, W, 3, view, interp, flags)
                imageio.write_image(video.output_path(job, plan, written), out, jpeg_q=job.jpeg_q)
                written += 1
        finally:
            with self.videos_lock:
                sess.active_jobs -= 1
                sess.done_jobs += 1
                if expected_jobs and sess.done_jobs >= expected_jobs and sess.active_jobs == 0:
                    self.videos.pop(plan.key, None)
                    sess.close()
        return written


_engine = None
_engine_lock = threading.Lock()


def get_engine() -> Engine:
    global _engine
    with _engine_lock:
        if _engine is None:
            _engine = Engine()
        return _engine


def shutdown():
    global _engine
    with _engine_lock:
        if _engine is not None:
            _engine.close()
            _engine = None
