"""Host-fed frame pipeline: pinned host frames -> H2D -> views kernel -> D2H, several frames in flight per GPU.

This is the end-to-end shape north_star describes ("pinned-host decoded frames fanned out on per-GPU HIP
streams").  Every frame in flight owns a buffer set (pinned input, device frame, device outputs, pinned outputs).
ALL uploads and launches go to one stream and ALL downloads to a second one, chained per frame by an event:
measured on MI355X, one upload stream + one download stream moves 97 GB/s over PCIe (both directions busy), while
both directions on one stream, or several streams each mixing both, stay at 57-67 GB/s.  Throughput is bounded by
PCIe (an 8K RGB frame is 88.5 MB in; its views come back out).

`batch` > 1 batches the KERNEL, not the copies: every frame is uploaded as it is committed, the views of `batch` consecutive frames are
rendered by ONE launch behind the last upload (ring families -- the `full360coverage` preset -- reach the source-major kernel from four
frames per call), their downloads follow.  The pipeline then needs more than `batch` slots to keep uploads flowing under the launch.
"""
import ctypes as C
from typing import List, Sequence

import numpy as np

from . import capi


class _Slot:
    def __init__(self, ctx, idx, frame_bytes, view_bytes):
        self.idx = idx
        self.h_in = ctx.pinned(frame_bytes)
        self.d_in = ctx.alloc(frame_bytes)
        self.d_out = [ctx.alloc(b) for b in view_bytes]
        self.h_out = [ctx.pinned(b) for b in view_bytes]
        self.busy = False
        self.tag = None


class FramePipeline:
    def __init__(self, ctx: capi.Context, W: int, H: int, Cn: int, views: Sequence[capi.View], n_slots: int = None,
                 copy_out: bool = True, batch: int = 1):
        """copy_out=False hands results out as arrays that ALIAS the slot's pinned output buffers (valid until that slot
        is handed out again by acquire()): a consumer that encodes or writes them right away needs no extra host copy."""
        self.ctx, self.W, self.H, self.C = ctx, W, H, Cn
        self.copy_out = copy_out
        self.views = list(views)
        n_slots = n_slots or max(2, ctx.n_slots, batch + 1 if batch > 1 else 0)
        if ctx.n_slots < 2 or n_slots > 8:
            raise ValueError("pipeline needs a context with >= 2 stream slots and keeps <= 8 frames in flight")
        if batch < 1 or batch > min(n_slots, capi.MAX_FRAMES):
            raise ValueError("batch must be between 1 and the number of slots")
        self.batch = batch
        self._pending = []                      # slots whose frame is uploaded but not launched yet (batch > 1)
        self.s_up, self.s_down = 0, 1          # stream slots: uploads + launches / downloads
        self.frame_bytes = W * H * Cn
        self.view_shapes = [(v.height, v.width, Cn) for v in self.views]
        vb = [h * w * c for h, w, c in self.view_shapes]
        self.slots = [_Slot(ctx, i, self.frame_bytes, vb) for i in range(n_slots)]
        self._next = 0

    def acquire(self):
        """Claim the next slot and hand out its PINNED input buffer as a writable uint8 array, so a producer (a
        decoder's readinto, see gs360/video.py) fills it in place and no staging copy is needed.  Returns
        (done, array): `done` = results of the frame that previously occupied the slot, or None.  Follow with commit()."""
        s = self.slots[self._next]
        if s in self._pending:
            self._launch_pending()               # (the ring came round to a frame that still waits for its batch)
        done = self._collect(s) if s.busy else None
        return done, np.frombuffer(s.h_in.view, dtype=np.uint8, count=self.frame_bytes)

    def commit(self, tag=None) -> None:
        """Enqueue upload -> all-views launch -> downloads for the slot handed out by acquire()."""
        s = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        L, h = self.ctx.L, self.ctx.handle
        # the buffer set is free: acquire() waited for its previous download event before handing it out
        capi._check(L.gs360_upload(h, s.d_in.ptr, s.h_in.ptr, self.frame_bytes, self.s_up), L)
        s.busy, s.tag = True, tag
        self._pending.append(s)
        if len(self._pending) >= self.batch:
            self._launch_pending()

    def _launch_pending(self) -> None:
        """one launch for the frames uploaded since the last one (behind their uploads on the same stream), then their downloads"""
        if not self._pending:
            return
        L, h = self.ctx.L, self.ctx.handle
        group, self._pending = self._pending, []
        self.ctx.equirect_views_dev([s.d_in for s in group], self.W, self.H, self.C, self.views, [d for s in group for d in s.d_out],
                                    slot=self.s_up)
        last = group[-1]
        self.ctx.event_record(self.s_up, last.idx)
        self.ctx.stream_wait_event(self.s_down, self.s_up, last.idx)
        for s in group:
            for d, hbuf in zip(s.d_out, s.h_out):
                capi._check(L.gs360_download(h, hbuf.ptr, d.ptr, hbuf.nbytes, self.s_down), L)
            self.ctx.event_record(self.s_down, s.idx)

    def submit(self, frame: np.ndarray, tag=None):
        """Enqueue one frame held in ordinary host memory (copied into the slot's pinned buffer first; blocks only if
        that slot is still busy).  Returns results of the frame that previously occupied the slot, or None."""
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        if a.nbytes != self.frame_bytes:
            raise ValueError("frame size mismatch")
        done, staging = self.acquire()
        staging[:] = a.reshape(-1)
        self.commit(tag)
        return done

    def _collect(self, s):
        if s in self._pending:
            self._launch_pending()
        self.ctx.event_sync(self.s_down, s.idx)
        outs = [np.frombuffer(hb.view, dtype=np.uint8).reshape(shape) for hb, shape in zip(s.h_out, self.view_shapes)]
        if self.copy_out:
            outs = [o.copy() for o in outs]
        s.busy = False
        return s.tag, outs

    def drain(self) -> List:
        """Finish every frame in flight, oldest first."""
        self._launch_pending()
        res = []
        for k in range(len(self.slots)):
            s = self.slots[(self._next + k) % len(self.slots)]
            if s.busy:
                res.append(self._collect(s))
        return res

    def close(self):
        for s in self.slots:
            self.ctx.free(s.d_in)
            for d in s.d_out:
                self.ctx.free(d)
            self.ctx.unpin(s.h_in)
            for hb in s.h_out:
                self.ctx.unpin(hb)
        self.slots = []
