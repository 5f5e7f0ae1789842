"""Process-wide execution engine behind run_one(): frames x views fanned out over the visible GPUs.

Unit of work = (frame, view), exactly the reference's job granularity (PC:830-836).  Sharding is by FRAME:
all views of one source image run on the same device, so the decoded frame is uploaded once and stays
resident in HBM (an LRU of device frames per GPU); there is no exchange step and therefore no collective.
Concurrency comes from the caller's thread pool (PC:1049 / gs360_GUI.py:19297).

Launch shape.  The callers hand over ONE (frame, view) at a time, but a single 800^2 view is ~350 workgroups -- a
quarter of the chip's resident slots -- so a launch per view is all tail.  The jobs of one frame arrive together (the
planner emits them frame-major and the pool starts them concurrently), so they are COALESCED: the first job of a
frame becomes the batch leader, gets the frame resident, waits a short linger (GS360_BATCH_LINGER_MS, default 3 ms;
it leaves early once the announced number of views has arrived), then renders every view that joined in ONE batched
launch, queues all device->pinned-host copies on the same stream and synchronises once.  Every job then encodes its
own view in its own thread (the encoders are the end-to-end bound, scripts/bench_cli_e2e.py).
"""
import collections
import itertools
import os
import threading
import time

import numpy as np

from . import capi, hostmem, imageio, video
from .jobspec import JobSpec

_FRAME_CACHE_BYTES = int(os.environ.get("GS360_FRAME_CACHE_MB", "4096")) << 20
_SLOTS_PER_DEVICE = 4
_LINGER_S = float(os.environ.get("GS360_BATCH_LINGER_MS", "3")) * 1e-3
_PREFETCH_FRAMES = int(os.environ.get("GS360_PREFETCH_FRAMES", "8"))      # still images decoded ahead of their view jobs (0 = off)
_PREFETCH_THREADS = int(os.environ.get("GS360_PREFETCH_THREADS", "4"))    # decoder threads of the read-ahead
_STALE_RUN_S = float(os.environ.get("GS360_STALE_RUN_S", "30"))            # an earlier run's untouched sources are dropped by the next announce() after this long
_RECENT_SOURCES = 512                   # sources whose device is remembered after their record is gone (bounded)


class _Batch:
    """Views of ONE frame (same interpolation / projection flags) that go out as one launch."""
    __slots__ = ("views", "state", "done", "results", "error", "abandoned")

    def __init__(self):
        self.views = []                 # capi.View, in arrival order
        self.state = "open"             # open -> running -> done
        self.done = threading.Event()
        self.results = None             # list of (ndarray aliasing a pinned buffer, PinnedBuffer)
        self.error = None
        self.abandoned = set()          # indices of followers that left (cancel) before the results existed


class _Source:
    """What the engine remembers about one source while it is in flight."""
    __slots__ = ("dev", "expected", "remaining", "active", "touched", "ahead", "run", "stamp")

    def __init__(self, dev):
        self.dev = dev                  # index into Engine.states: all views of a source share a device
        self.expected = None            # view jobs that can arrive together (announce()), None = not announced
        self.remaining = None           # announced view jobs that have not finished yet, None = not announced
        self.active = 0                 # view jobs inside run_job right now
        self.touched = False            # a view job has asked for the frame (the read-ahead leaves it alone from then on)
        self.ahead = False              # decoded ahead and still holding a read-ahead permit
        self.run = 0                    # the announce() that listed it last (0 = never announced)
        self.stamp = time.monotonic()   # last time a job list or a job touched it


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
        self.batch_cond = threading.Condition()   # guards open_batches
        self.open_batches = {}                    # (frame key, interp, flags) -> _Batch still collecting views
        self.pool_lock = threading.Lock()
        self.dev_pool = collections.defaultdict(list)    # nbytes -> free DeviceBuffers (view outputs)
        self.pin_pool = collections.defaultdict(list)    # nbytes -> free PinnedBuffers
        self.stats = collections.Counter()        # launches, views, gpu_s (launch + copies, wall on the stream), ...

    def take(self, pool, nbytes, make):
        with self.pool_lock:
            free = pool[nbytes]
            if free:
                return free.pop()
        return make(nbytes)

    def give(self, pool, buf):
        with self.pool_lock:
            pool[buf.nbytes].append(buf)

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
        self.videos = {}                          # DecodePlan.key -> the video.VideoSession new view jobs join
        self._video_done = {}                     # DecodePlan.key -> view jobs finished so far (all sessions of that video)
        self.videos_lock = threading.Lock()
        self._video_budget = video.SharedBudget(video._BUDGET_BYTES * max(1, len(self.states)))   # all live sessions together
        self._init_bookkeeping()

    def _init_bookkeeping(self):
        # per-source bookkeeping (a long-lived host -- the GUI imports the module once and exports many times -- must not grow):
        # one record per source that is announced or has a job running; it goes when its announced jobs have all finished (or, for
        # sources nobody announced, when its last running job leaves), and announce() / retire() drop what a cancelled run left
        self._sources = {}                        # str(source path) -> _Source
        self._announce_lock = threading.Lock()    # guards _sources, _inflight, the prefetch queue
        self._inflight = [0] * len(self.states)   # sources currently held per device (device_for balances on THIS, not on history)
        self._recent = collections.OrderedDict()  # source -> device of its last record, the newest _RECENT_SOURCES of them
        self._last_dev = len(self.states) - 1     # round-robin tie-break starts at device 0
        self._prefetch_queue = collections.deque()   # (run, source): announced still-image sources not yet decoded ahead
        self._run_id = 0                          # announce() counter
        self._prefetch_threads = []
        self._prefetch_permits = threading.Semaphore(_PREFETCH_FRAMES)
        self._prefetch_stop = threading.Event()

    def close(self):
        self._prefetch_stop.set()
        for t in list(self._prefetch_threads):
            t.join(timeout=5.0)
        with self.videos_lock:
            sessions, self.videos = list(self.videos.values()), {}
        for sess in sessions:
            sess.close()
        for st in self.states:
            with st.pool_lock:
                pinned = [hb for free in st.pin_pool.values() for hb in free]
                st.pin_pool.clear()
                st.dev_pool.clear()               # device buffers are owned (and freed) by the context
            for hb in pinned:
                hb.free()
            st.ctx.close()
        self.states = []

    # -- sharding ---------------------------------------------------------------------------------
    def _source(self, key):
        """(announce lock held) the record of `key`.  A new source goes where an earlier record of it lived (a bounded memory of
        recent sources: the views of a frame that a serial caller hands over one by one stay on one device), else to the device
        with the fewest sources in flight, ties broken round-robin from the last choice."""
        rec = self._sources.get(key)
        if rec is None:
            dev = self._recent.get(key)
            if dev is None or dev >= len(self.states):
                n = len(self.states)
                dev = min(range(n), key=lambda d: (self._inflight[d], (d - self._last_dev - 1) % n))
                self._last_dev = dev
            self._recent[key] = dev
            self._recent.move_to_end(key)
            while len(self._recent) > _RECENT_SOURCES:
                self._recent.popitem(last=False)
            rec = self._sources[key] = _Source(dev)
            self._inflight[dev] += 1
        return rec

    def _drop(self, key):
        """(announce lock held) forget `key`: its device slot and, if it was decoded ahead and never used, its permit"""
        rec = self._sources.pop(key, None)
        if rec is None:
            return
        self._inflight[rec.dev] -= 1
        if rec.ahead:
            rec.ahead = False
            self._prefetch_permits.release()

    def device_for(self, src_path) -> int:
        """frame -> device index; all views of a frame share a device.  Sources are dealt to the device holding the fewest
        sources IN FLIGHT, in the order they become known -- the order of the announced job list when the caller announces one
        (the drop-in CLI does), else first come first served -- so a folder of n frames occupies min(n, devices) devices with at
        most one frame of spread (a hash of the path left devices idle by chance on the 6-8 file folders of the presets' typical
        use), and an engine that has already served other folders starts level again."""
        with self._announce_lock:
            return self._source(str(src_path)).dev

    def retire(self, run=None):
        """End of a run (the drop-in CLI's main() calls it): forget every source no job is working on -- a cancelled or failed run leaves
        announced jobs that never arrive -- and empty the read-ahead queue; permits of frames decoded ahead and never used go back.
        With `run` (the value announce() returned) only THAT run's sources and queue entries go: a long-lived host with overlapping
        exports (the GUI) ends one run without touching the other's queued sources."""
        with self._announce_lock:
            if run is None:
                self._prefetch_queue.clear()
            else:
                self._prefetch_queue = collections.deque(e for e in self._prefetch_queue if e[0] != run)
            for key in [k for k, r in self._sources.items() if r.active == 0 and (run is None or r.run == run)]:
                self._drop(key)

    def _retire_stale(self, current_run):
        """(announce lock held) what an earlier run left behind and nobody came back for: records of OTHER runs with no job inside and no
        activity for _STALE_RUN_S seconds.  A run that is still working keeps its queued sources, their `expected` counts and their
        read-ahead (round-4 ADVICE: a second announce() used to discard them)."""
        now = time.monotonic()
        for key in [k for k, r in self._sources.items() if r.active == 0 and r.run != current_run and now - r.stamp > _STALE_RUN_S]:
            self._drop(key)
        self._prefetch_queue = collections.deque(e for e in self._prefetch_queue if e[1] in self._sources)

    def bookkeeping(self):
        """sizes of the per-source tables (tests: a long-lived engine must come back to empty)"""
        with self._announce_lock:
            assert len(self._recent) <= _RECENT_SOURCES
            return {"sources": len(self._sources), "inflight": list(self._inflight), "queue": len(self._prefetch_queue)}

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
            entry = [buf, H, W, C, 1, img.dtype]  # [4]: users currently holding the frame; [5]: uint8 / uint16
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
                    st.frame_bytes -= old[1] * old[2] * old[3] * np.dtype(old[5]).itemsize
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

    def _launch_batch(self, st: _DeviceState, buf, H, W, C, views, interp, flags, dtype=np.uint8):
        """One batched launch for `views` of one resident frame -- or of a WINDOW of resident frames (`buf` a list: the frames of a video
        the view jobs walk together, PC:746-749, PC:1049-1078; ring families reach the source-major kernel from four frames per call).
        Returns [(array aliasing pinned memory, PinnedBuffer)] per view; for a window a list of those, one per frame."""
        with st.lock:
            slot = next(st.slot_cycle)
        ctx, L = st.ctx, st.ctx.L
        esz = np.dtype(dtype).itemsize
        window = isinstance(buf, list)
        bufs = buf if window else [buf]
        n_views = len(views)
        sizes = [v.height * v.width * C * esz for v in views] * len(bufs)           # frame-major, as gs360_equirect_views_u8 writes them
        d_out = [st.take(st.dev_pool, n, ctx.alloc) for n in sizes]
        h_out = [st.take(st.pin_pool, n, ctx.pinned) for n in sizes]
        t0 = time.perf_counter()
        try:
            with ctx.slot_locks[slot]:
                ctx.equirect_views_dev(bufs, W, H, C, views, d_out, slot=slot, interp=interp, flags=flags, dtype=dtype)
                for d, hb, n in zip(d_out, h_out, sizes):
                    capi._check(L.gs360_download(ctx.handle, hb.ptr, d.ptr, n, slot), L)
                ctx.sync(slot)
        except BaseException:
            for hb in h_out:                      # nobody will ever hold these results: the pinned blocks go back to the pool
                st.give(st.pin_pool, hb)
            raise
        finally:
            for d in d_out:
                st.give(st.dev_pool, d)
        with st.pool_lock:
            st.stats["launches"] += 1
            st.stats["views"] += len(sizes)
            st.stats["frames"] = st.stats.get("frames", 0) + len(bufs)
            # which kernel the launch ran (the context's read-only option; another slot's launch may have overwritten it: statistics only)
            kname = {0: "launches_gather", 1: "launches_staged", 2: "launches_srcmajor"}.get(ctx.get_option("last_eq_kernel"))
            if kname:
                st.stats[kname] = st.stats.get(kname, 0) + 1
            st.stats["gpu_s"] += time.perf_counter() - t0
        flat = [(np.frombuffer(hb.view, dtype=dtype, count=n // esz).reshape(v.height, v.width, C), hb)
                for hb, n, v in zip(h_out, sizes, views * len(bufs))]
        return [flat[f * n_views:(f + 1) * n_views] for f in range(len(bufs))] if window else flat

    def _render(self, st: _DeviceState, fkey, get_frame, view, interp, flags=0, expected=1, stop_event=None):
        """Render `view` of the frame identified by `fkey`, coalesced with the other views of that frame that arrive
        within the linger window.  get_frame() -> (DeviceBuffer, H, W, C, dtype) is called by the batch leader only.
        Returns (array, release): the array aliases pinned host memory until release() is called."""
        key = (fkey, interp, flags)
        with st.batch_cond:
            b = st.open_batches.get(key)
            leader = b is None
            if leader:
                b = st.open_batches[key] = _Batch()
            idx = len(b.views)
            b.views.append(view)
            st.batch_cond.notify_all()
        if leader:
            try:
                buf, H, W, C, dtype = get_frame()     # decode + upload happen here; followers keep arriving meanwhile
                deadline = time.monotonic() + (_LINGER_S if expected > 1 else 0.0)
                with st.batch_cond:
                    while len(b.views) < min(expected, capi.MAX_VIEWS):
                        left = deadline - time.monotonic()
                        if left <= 0 or (stop_event is not None and stop_event.is_set()):
                            break
                        st.batch_cond.wait(left)
                    b.state = "running"
                    del st.open_batches[key]
                    views = list(b.views)
                if buf is None:                   # (a video's window past its last frame)
                    res = [None] * len(views)
                else:
                    res = self._launch_batch(st, buf, H, W, C, views, interp, flags, dtype)
                    if isinstance(buf, list):     # a window: per member the list of its view's frames
                        res = [[res[f][i] for f in range(len(buf))] for i in range(len(views))]
                with st.batch_cond:
                    b.results = res
                    for i in b.abandoned:         # followers that were cancelled meanwhile will not come for their view
                        self._give_back(st, res[i])
            except BaseException as exc:  # noqa: BLE001  (handed to every member of the batch)
                with st.batch_cond:
                    if st.open_batches.get(key) is b:
                        del st.open_batches[key]
                b.error = exc
            finally:
                b.state = "done"
                b.done.set()
        else:
            while not b.done.wait(0.25):
                if stop_event is not None and stop_event.is_set():
                    with st.batch_cond:
                        if b.results is None:
                            b.abandoned.add(idx)  # the leader returns this view's pinned block when the launch completes
                        else:
                            self._give_back(st, b.results[idx])
                    raise capi.Gs360Error(-2, "cancelled")
        if b.error is not None:
            if leader:
                raise b.error
            raise capi.Gs360Error(-2, f"batched launch failed: {b.error}")
        if b.results is None:
            raise capi.Gs360Error(-2, "cancelled")
        mine = b.results[idx]
        if mine is None:
            return None, (lambda: None)
        if isinstance(mine, list):                # a window of frames: [array per frame]
            return [a for a, _hb in mine], (lambda: self._give_back(st, mine))
        arr, hb = mine
        return arr, (lambda: st.give(st.pin_pool, hb))

    @staticmethod
    def _give_back(st, res):
        """the pinned block(s) of one member's result go back to the pool"""
        if res is None:
            return
        for _a, hb in (res if isinstance(res, list) else [res]):
            st.give(st.pin_pool, hb)

    def announce(self, jobs, workers=None):
        """Optional hint from a caller that knows its whole job list (the drop-in CLI's main()): how many view jobs each
        source has, and how many of them can be in flight at once.  Lets a batch leader stop lingering as soon as every
        view that can arrive has arrived.  Without it every batch simply lingers for the full window."""
        counts = collections.Counter(str(j.src) for j in jobs)
        with self._announce_lock:
            self._run_id += 1
            run = self._run_id
            self._retire_stale(run)               # leftovers of earlier runs that were cancelled (nothing has touched them for a while)
            for k, n in counts.items():           # deal the sources to the devices in job-list order (Counter keeps it)
                rec = self._source(k)
                rec.expected = min(n, workers) if workers else n
                if rec.run != run and rec.run != 0 and rec.active == 0:
                    # listed again by a NEW run while no job of the earlier one is inside: what that run still had announced will never
                    # arrive (a cancelled export started again within the stale window) -- adding to it would leave a count that never
                    # reaches zero, the frame resident and, if it was decoded ahead, its read-ahead permit held for good
                    rec.remaining = n
                    if rec.ahead:
                        rec.ahead = False
                        self._prefetch_permits.release()
                else:
                    rec.remaining = (rec.remaining or 0) + n
                rec.run, rec.stamp = run, time.monotonic()
        self._start_prefetch([j.src for j in jobs if j.is_still_image], run)
        return run

    # -- decode-ahead -----------------------------------------------------------------------------
    def _start_prefetch(self, sources, run=0):
        """With the whole job list known, decode + upload the next still images in the background while the view jobs of the
        current ones render and encode.  Without it a frame's decode (70 ms for a 5.7K PNG) sits on the critical path of its 8
        view jobs: the first job decodes, the others wait, and with `-j 16` only two frames are ever in flight.  At most
        GS360_PREFETCH_FRAMES (8) frames run ahead of the jobs; the first job that touches a source hands its permit back."""
        order = []
        for src in sources:
            k = str(src)
            if k not in order:
                order.append(k)
        if len(order) < 2 or _PREFETCH_FRAMES <= 0:
            return
        with self._announce_lock:
            self._prefetch_queue.extend((run, k) for k in order)
            need = min(_PREFETCH_THREADS, len(order)) - len(self._prefetch_threads)
            for _ in range(max(0, need)):
                t = threading.Thread(target=self._prefetch_loop, name="gs360-decode-ahead", daemon=True)
                self._prefetch_threads.append(t)
                t.start()

    def _prefetch_loop(self):
        while not self._prefetch_stop.is_set():
            if not self._prefetch_permits.acquire(timeout=0.25):      # a permit first, then the NEXT source: strict job-list order
                with self._announce_lock:
                    if not self._prefetch_queue:
                        break
                continue
            src = None
            with self._announce_lock:
                while self._prefetch_queue:
                    _run, cand = self._prefetch_queue.popleft()
                    rec = self._sources.get(cand)
                    if rec is not None and not rec.touched:   # else its view jobs are already running (they decode it themselves)
                        src = cand
                        rec.ahead = True
                        dev = rec.dev                         # resolved HERE: a record dropped after the lock is released must not be re-created by a lookup
                        break
            if src is None:
                self._prefetch_permits.release()
                break
            try:
                st = self.states[dev]
                self.release_frame(st, self.resident_frame(st, src))
            except Exception:  # noqa: BLE001  (the view job reports the real error when it gets there)
                self._job_touches(src, entering=False)
        with self._announce_lock:
            self._prefetch_threads = [t for t in self._prefetch_threads if t is not threading.current_thread()]

    def _job_touches(self, src, entering=True):
        """a view job (or a failed prefetch) reached `src`: whatever ran ahead for it is consumed.  -> the source's record"""
        with self._announce_lock:
            rec = self._source(str(src))
            rec.touched = True
            rec.stamp = time.monotonic()
            if entering:
                rec.active += 1
            if rec.ahead:
                rec.ahead = False
                self._prefetch_permits.release()
            return rec

    def _job_leaves(self, src):
        """a view job is done with `src` (written, failed or cancelled): the record goes with the last announced job -- or, for a
        source nobody announced, with the last job working on it"""
        key = str(src)
        with self._announce_lock:
            rec = self._sources.get(key)
            if rec is None:
                return
            rec.active -= 1
            if rec.remaining is not None:
                rec.remaining -= 1
            if rec.active <= 0 and (rec.remaining is None or rec.remaining <= 0):
                self._drop(key)

    def run_job(self, job: JobSpec, stop_event=None, want_array: bool = False):
        """Execute one (frame, view) job: render (coalesced with the frame's other views) and write job.dst.  Returns a copy
        of the view when want_array is set (the result itself lives in pooled pinned memory), else None."""
        view, flags = self._view_for(job)
        interp = self._interp_for(job)
        rec = self._job_touches(job.src)
        try:
            st = self.states[rec.dev]
            held = []

            def get_frame():
                entry = self.resident_frame(st, job.src)
                held.append(entry)
                return entry[0], entry[1], entry[2], entry[3], entry[5]
            try:
                arr, release = self._render(st, self._frame_key(job.src), get_frame, view, interp, flags,
                                            expected=rec.expected or capi.MAX_VIEWS, stop_event=stop_event)
            finally:
                for entry in held:
                    self.release_frame(st, entry)
            try:
                imageio.write_image(job.dst, arr, jpeg_q=job.jpeg_q)
                return np.array(arr) if want_array else None
            finally:
                release()
        finally:
            self._job_leaves(job.src)

    def stats(self):
        """Counters since start: batched launches, views, seconds the launch+copy sections held a stream."""
        total = collections.Counter()
        for st in self.states:
            total.update(st.stats)
        return dict(total)

    # -- video: one decode, frames resident in HBM, every view job walks them (gs360/video.py) -------------------
    def _video_session(self, plan, stop_event, register_proc):
        """-> (session, token).  A job joins the video's current session; when that one has already retired its first frames
        (a long video under a small budget and a job that starts late) a fresh session -- a second decode -- takes its place
        for this job and every later one, and the old session is closed by its last job."""
        with self.videos_lock:
            sess = self.videos.get(plan.key)
            token = sess.join() if sess is not None else None
            if token is None:
                for key in [k for k, s in self.videos.items() if s.active_jobs == 0 and s.finished and k != plan.key]:
                    self.videos.pop(key).close()          # idle sessions of other videos give their memory back
                if sess is not None and sess.active_jobs == 0:
                    sess.close()
                    sess = None
                if sess is None:
                    self._video_done.pop(plan.key, None)  # no session of this video is alive: a count left by a cancelled or partly
                                                          # failed run must not make the new run's jobs look finished early
                self._video_budget.total = video._BUDGET_BYTES * max(1, len(self.states))
                sess = self.videos[plan.key] = video.VideoSession(self.states, plan, stop_event, register_proc, shared=self._video_budget)
                token = sess.join()
            return sess, token

    def run_video_job(self, job: JobSpec, plan, stop_event=None, register_proc=None, expected_jobs=None) -> int:
        """All frames of one view of a video; returns the number of frames written."""
        view, flags = self._view_for(job)
        interp = self._interp_for(job)
        sess, token = self._video_session(plan, stop_event, register_proc)
        written = 0
        try:
            while True:
                # The view jobs of a video walk its frames together, a WINDOW at a time: frames k .. k + n of every active view go out as
                # ONE launch (the frames are resident in HBM; n = what is decoded already, at most video._WINDOW).  The batch leader takes
                # the window from the session, the others adopt it.
                k0 = written
                st = sess.state_for(k0)
                sess.advance(token, k0)

                def get_window(k0=k0):
                    win = sess.window(token, k0)
                    if win is None:
                        return None, 0, 0, 3, np.uint8
                    _st, bufs, H, W, fdtype = win
                    return bufs, H, W, 3, fdtype
                outs, release = self._render(st, ("video", id(sess), k0), get_window, view, interp, flags,
                                             expected=min(sess.active_jobs, expected_jobs or sess.active_jobs), stop_event=stop_event)
                if outs is None:
                    break
                try:
                    for out in outs:
                        imageio.write_image(video.output_path(job, plan, written), out, jpeg_q=job.jpeg_q)
                        written += 1
                finally:
                    release()
        finally:
            with self.videos_lock:
                sess.leave(token)
                self._video_done[plan.key] = self._video_done.get(plan.key, 0) + 1
                current = self.videos.get(plan.key)
                all_done = bool(expected_jobs) and self._video_done[plan.key] >= expected_jobs
                if sess.active_jobs == 0 and (sess is not current or all_done):
                    sess.close()                          # a superseded session, or the video's last planned view job
                    if sess is current:
                        self.videos.pop(plan.key, None)
                if all_done:
                    self._video_done.pop(plan.key, None)
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
