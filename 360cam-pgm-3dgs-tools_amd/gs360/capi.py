"""ctypes binding of libgs360hip.so (C ABI declared in include/gs360.h).

This is the only door from Python into the engine.  There is NO CPU fallback: if the shared
library or a gfx950 device is missing every call raises Gs360Error.
"""
import ctypes as C
import os
import pathlib
import threading

import numpy as np

PKG_DIR = pathlib.Path(__file__).resolve().parent.parent
LIB_PATH = pathlib.Path(os.environ.get("GS360_LIB", PKG_DIR / "lib" / "libgs360hip.so"))  # override: profiling probes

INTERP_NEAREST = 0  # == cv2.INTER_NEAREST
INTERP_LINEAR = 1   # == cv2.INTER_LINEAR
INTERP_CUBIC = 2    # == cv2.INTER_CUBIC
INTERP_LANCZOS4 = 4  # == cv2.INTER_LANCZOS4 (table remap / fused fisheye only)
EQ_FISHEYE_OUT = 1   # flags: the views are equidistant-fisheye outputs (v360 output=fisheye), see include/gs360.h
MAX_VIEWS = 16
MAX_FRAMES = 16

EXPORTS = (
    "gs360_abi_version", "gs360_device_count", "gs360_last_error", "gs360_ctx_create", "gs360_ctx_destroy",
    "gs360_device_info", "gs360_device_pci_bus_id", "gs360_ctx_set_option", "gs360_ctx_get_option", "gs360_dev_alloc", "gs360_dev_free", "gs360_host_alloc", "gs360_host_free",
    "gs360_upload", "gs360_download", "gs360_dev_memset", "gs360_dev_bswap16", "gs360_sync", "gs360_event_record",
    "gs360_event_elapsed_ms", "gs360_equirect_views_u8", "gs360_equirect_views_masked_u8", "gs360_remap_table_u8",
    "gs360_fisheye_views_u8", "gs360_remap_tables_u8", "gs360_map_plan_create", "gs360_map_plan_destroy", "gs360_remap_plans_u8", "gs360_remap_plans_u16",
    "gs360_color_plan_create", "gs360_color_plan_destroy", "gs360_color_apply_u8",
    "gs360_equirect_views_u8_host", "gs360_remap_table_u8_host",
    "gs360_equirect_views_u16", "gs360_remap_table_u16", "gs360_remap_tables_u16", "gs360_equirect_views_u16_host", "gs360_remap_table_u16_host",
    "gs360_png_unfilter", "gs360_event_sync", "gs360_stream_wait_event",
    "gs360_color_plan16_create", "gs360_color_plan16_destroy", "gs360_color_apply_u16", "gs360_tiff_lzw_decode", "gs360_selftest_arith",
)


class Gs360Error(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"gs360 error {code}: {text}")
        self.code = code
        self.text = text


class View(C.Structure):
    """Numeric fields of the reference's ViewSpec (gs360_360PerspCut.py:32-45)."""
    _fields_ = [("yaw_deg", C.c_double), ("pitch_deg", C.c_double), ("hfov_deg", C.c_double),
                ("vfov_deg", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]

    @classmethod
    def make(cls, yaw, pitch, hfov, vfov, width, height):
        return cls(float(yaw), float(pitch), float(hfov), float(vfov), int(width), int(height))


class Calib(C.Structure):
    """Numeric fields of the reference's SensorCalibration (gs360_DualFisheyeDistortionCalibration.py:67-85)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32)] + [
        (n, C.c_double) for n in ("f", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2")]

    @classmethod
    def make(cls, width, height, f, cx=0.0, cy=0.0, k1=0.0, k2=0.0, k3=0.0, k4=0.0, p1=0.0, p2=0.0, b1=0.0, b2=0.0):
        return cls(int(width), int(height), *[float(v) for v in (f, cx, cy, k1, k2, k3, k4, p1, p2, b1, b2)])


class RemapJob(C.Structure):
    """gs360_remap_job: one cv2.remap call of a batched launch (device pointers)."""
    _fields_ = [("src", C.c_void_p), ("H", C.c_int32), ("W", C.c_int32), ("src_stride", C.c_size_t),
                ("map_x", C.c_void_p), ("map_y", C.c_void_p), ("valid", C.c_void_p), ("h", C.c_int32), ("w", C.c_int32),
                ("fill_value", C.c_int32), ("dst", C.c_void_p), ("dst_stride", C.c_size_t)]


ABI_VERSION = 2          # GS360_ABI_VERSION of include/gs360.h this binding was written against
_lib = None
_lib_lock = threading.Lock()


def load_library(path=None):
    """Load libgs360hip.so and declare prototypes.  Raises Gs360Error if it is not built."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = pathlib.Path(path) if path else LIB_PATH
        if not p.exists():
            raise Gs360Error(-3, f"{p} is missing -- build it with `python __graft_entry__.py` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = C.CDLL(str(p))
        # the version first: a stale library should say so, not fail on a symbol it does not have yet
        try:
            L.gs360_abi_version.argtypes = []
            have = int(L.gs360_abi_version())
        except AttributeError:
            have = -1
        if have != ABI_VERSION:
            raise Gs360Error(-4, f"{p} has C-ABI version {have}, this binding needs {ABI_VERSION} (include/gs360.h): rebuild it with "
                                 "`python __graft_entry__.py`")
        vp, i, sz, u32 = C.c_void_p, C.c_int, C.c_size_t, C.c_uint32
        pvp = C.POINTER(C.c_void_p)
        L.gs360_abi_version.argtypes = []
        L.gs360_device_count.argtypes = []
        L.gs360_last_error.argtypes = [C.c_char_p, sz]
        L.gs360_ctx_create.argtypes = [i, i, pvp]
        L.gs360_ctx_destroy.argtypes = [vp]
        L.gs360_device_info.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.gs360_device_pci_bus_id.argtypes = [vp, C.c_char_p, sz]
        L.gs360_ctx_set_option.argtypes = [vp, C.c_char_p, i]
        L.gs360_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
        L.gs360_dev_alloc.argtypes = [vp, sz, pvp]
        L.gs360_dev_free.argtypes = [vp, vp]
        L.gs360_host_alloc.argtypes = [vp, sz, pvp]
        L.gs360_host_free.argtypes = [vp, vp]
        L.gs360_upload.argtypes = [vp, vp, vp, sz, i]
        L.gs360_download.argtypes = [vp, vp, vp, sz, i]
        L.gs360_dev_memset.argtypes = [vp, vp, i, sz, i]
        L.gs360_dev_bswap16.argtypes = [vp, vp, sz, i]
        L.gs360_sync.argtypes = [vp, i]
        L.gs360_event_record.argtypes = [vp, i, i]
        L.gs360_event_elapsed_ms.argtypes = [vp, i, i, i, C.POINTER(C.c_float)]
        L.gs360_equirect_views_u8.argtypes = [vp, pvp, i, i, i, i, sz, C.POINTER(View), i, pvp, sz, i, u32, i]
        L.gs360_equirect_views_masked_u8.argtypes = [vp, pvp, pvp, i, i, i, i, sz, sz, C.POINTER(View), i, pvp, sz, i, u32, i]
        L.gs360_remap_table_u8.argtypes = [vp, vp, i, i, i, sz, vp, vp, vp, i, i, i, C.POINTER(C.c_double), i, vp, sz, i]
        L.gs360_fisheye_views_u8.argtypes = [vp, pvp, C.POINTER(Calib), i, sz, C.POINTER(View), i, C.c_double, i, i, i,
                                             pvp, sz, pvp, i]
        L.gs360_remap_tables_u8.argtypes = [vp, C.POINTER(RemapJob), i, i, i, C.POINTER(C.c_double), i]
        L.gs360_map_plan_create.argtypes = [vp, vp, vp, vp, i, i, i, i, pvp]
        L.gs360_map_plan_destroy.argtypes = [vp, vp]
        L.gs360_remap_plans_u8.argtypes = [vp, C.POINTER(RemapJob), pvp, i, i, i, C.POINTER(C.c_double), i]
        L.gs360_remap_plans_u16.argtypes = L.gs360_remap_plans_u8.argtypes
        L.gs360_color_plan_create.argtypes = [vp, vp, i, vp, vp, pvp]
        L.gs360_color_plan_destroy.argtypes = [vp, vp]
        L.gs360_color_apply_u8.argtypes = [vp, vp, vp, i, i, i, sz, i, vp, sz, i]
        L.gs360_equirect_views_u8_host.argtypes = [vp, vp, i, i, i, sz, C.POINTER(View), i, pvp, sz, i, u32, i]
        L.gs360_remap_table_u8_host.argtypes = [vp, vp, i, i, i, sz, vp, vp, vp, i, i, i, C.POINTER(C.c_double), i, vp,
                                                sz, i]
        L.gs360_equirect_views_u16.argtypes = L.gs360_equirect_views_u8.argtypes
        L.gs360_remap_table_u16.argtypes = L.gs360_remap_table_u8.argtypes
        L.gs360_remap_tables_u16.argtypes = L.gs360_remap_tables_u8.argtypes
        L.gs360_equirect_views_u16_host.argtypes = L.gs360_equirect_views_u8_host.argtypes
        L.gs360_remap_table_u16_host.argtypes = L.gs360_remap_table_u8_host.argtypes
        L.gs360_png_unfilter.argtypes = [vp, i, i, i]
        L.gs360_selftest_arith.argtypes = [vp, u32, i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gs360_tiff_lzw_decode.argtypes = [vp, sz, vp, sz, C.POINTER(C.c_size_t)]
        L.gs360_color_plan16_create.argtypes = [vp, vp, i, vp, vp, i, vp, vp, vp, vp, pvp]
        L.gs360_color_plan16_destroy.argtypes = [vp, vp]
        L.gs360_color_apply_u16.argtypes = [vp, vp, vp, i, i, i, sz, i, vp, sz, i]
        L.gs360_event_sync.argtypes = [vp, i, i]
        L.gs360_stream_wait_event.argtypes = [vp, i, i, i]
        for name in EXPORTS:
            getattr(L, name).restype = C.c_int
        if path is None:
            _lib = L
        return L


def last_error(L=None):
    L = L or load_library()
    buf = C.create_string_buffer(512)
    L.gs360_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def _check(rc, L):
    if rc != 0:
        raise Gs360Error(rc, last_error(L))


def device_count():
    return load_library().gs360_device_count()


class DeviceBuffer:
    """A device allocation owned by a Context (freed with the context or explicitly)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(ctx.L.gs360_dev_alloc(ctx.handle, self.nbytes, C.byref(p)), ctx.L)
        self.ptr = p.value

    def free(self):
        if self.ptr:
            self.ctx.L.gs360_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None


class PinnedBuffer:
    """Page-locked host memory (gs360_host_alloc): async copies to/from it overlap with kernels."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        _check(ctx.L.gs360_host_alloc(ctx.handle, self.nbytes, C.byref(p)), ctx.L)
        self.ptr = p.value
        self.view = (C.c_uint8 * self.nbytes).from_address(self.ptr)   # writable buffer for numpy.frombuffer

    def free(self):
        if self.ptr:
            self.view = None
            self.ctx.L.gs360_host_free(self.ctx.handle, self.ptr)
            self.ptr = None


class Context:
    """One engine context = one GPU + n_slots HIP streams.  Thread-safe per slot (a lock per slot)."""

    def __init__(self, device=0, n_slots=2):
        self.L = load_library()
        h = C.c_void_p()
        _check(self.L.gs360_ctx_create(int(device), int(n_slots), C.byref(h)), self.L)
        self.handle = h
        self.device = int(device)
        self.n_slots = int(n_slots)
        self.slot_locks = [threading.Lock() for _ in range(n_slots)]
        self._buffers = set()                 # live DeviceBuffers; touched by reader and render threads alike
        self._buffers_lock = threading.Lock()

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if self.handle:
            with self._buffers_lock:
                live, self._buffers = self._buffers, set()
            for b in live:
                b.free()
            self.L.gs360_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int32()
        mem = C.c_uint64()
        _check(self.L.gs360_device_info(self.handle, name, 256, C.byref(cu), C.byref(mem)), self.L)
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": mem.value}

    # -- memory -----------------------------------------------------------------------------
    def alloc(self, nbytes):
        b = DeviceBuffer(self, nbytes)
        with self._buffers_lock:
            self._buffers.add(b)
        return b

    def free(self, buf):
        buf.free()
        with self._buffers_lock:
            self._buffers.discard(buf)

    def pinned(self, nbytes):
        return PinnedBuffer(self, nbytes)

    def unpin(self, hbuf):
        hbuf.free()

    def upload(self, buf, array, slot=0, sync=True):
        a = np.ascontiguousarray(array)
        if a.nbytes > buf.nbytes:
            raise ValueError("upload larger than buffer")
        _check(self.L.gs360_upload(self.handle, buf.ptr, a.ctypes.data, a.nbytes, slot), self.L)
        if sync:
            self.sync(slot)
        return buf

    def to_device(self, array, slot=0):
        a = np.ascontiguousarray(array)
        return self.upload(self.alloc(a.nbytes), a, slot)

    def download(self, buf, shape, dtype=np.uint8, slot=0):
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > buf.nbytes:
            raise ValueError("download larger than buffer")
        _check(self.L.gs360_download(self.handle, out.ctypes.data, buf.ptr, out.nbytes, slot), self.L)
        self.sync(slot)
        return out

    def memset(self, buf, value, slot=0):
        _check(self.L.gs360_dev_memset(self.handle, buf.ptr, int(value), buf.nbytes, slot), self.L)

    def bswap16(self, buf, n_samples, slot=0):
        """in-place byte swap of 16-bit samples on the device (asynchronous on `slot`)"""
        _check(self.L.gs360_dev_bswap16(self.handle, buf.ptr, int(n_samples), slot), self.L)

    def sync(self, slot=-1):
        _check(self.L.gs360_sync(self.handle, slot), self.L)

    def pci_bus_id(self):
        """'domain:bus:device.function' of this context's GPU"""
        buf = C.create_string_buffer(64)
        _check(self.L.gs360_device_pci_bus_id(self.handle, buf, 64), self.L)
        return buf.value.decode()

    def set_option(self, key, value):
        """kernel-selection switch of this context (include/gs360.h: gs360_ctx_set_option); results never depend on it"""
        _check(self.L.gs360_ctx_set_option(self.handle, key.encode(), int(value)), self.L)

    def get_option(self, key):
        v = C.c_int(0)
        _check(self.L.gs360_ctx_get_option(self.handle, key.encode(), C.byref(v)), self.L)
        return int(v.value)

    def options(self, **kw):
        """context manager: set options, restore the previous values on exit (tests)"""
        ctx = self

        class _Scope:
            def __enter__(self_inner):
                self_inner.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self_inner, *exc):
                for k, v in self_inner.old.items():
                    ctx.set_option(k, v)
                return False
        return _Scope()

    def event_record(self, slot, idx):
        _check(self.L.gs360_event_record(self.handle, slot, idx), self.L)

    def selftest_arith(self, seed=1, n_millions=256):
        """-> (operand sets checked, mismatches) of the reduced division / square-root sequences against IEEE `/`, sqrt"""
        n, bad = C.c_uint64(0), C.c_uint64(0)
        _check(self.L.gs360_selftest_arith(self.handle, int(seed) & 0xFFFFFFFF, int(n_millions), C.byref(n), C.byref(bad)), self.L)
        return int(n.value), int(bad.value)

    def event_sync(self, slot, idx):
        _check(self.L.gs360_event_sync(self.handle, slot, idx), self.L)

    def stream_wait_event(self, waiting_slot, event_slot, idx):
        _check(self.L.gs360_stream_wait_event(self.handle, waiting_slot, event_slot, idx), self.L)

    def event_elapsed_ms(self, slot, i_from, i_to):
        ms = C.c_float()
        _check(self.L.gs360_event_elapsed_ms(self.handle, slot, i_from, i_to, C.byref(ms)), self.L)
        return float(ms.value)

    # -- hot path, device-resident ----------------------------------------------------------
    def equirect_views_dev(self, frames, W, H, Cn, views, dsts, slot=0, src_stride=0, dst_stride=0,
                           interp=INTERP_LINEAR, masks=None, flags=0, dtype=np.uint8):
        """frames: list of DeviceBuffer (H x W x C); dsts: list (len frames*views) of DeviceBuffer;
        masks: optional list of DeviceBuffer (H x W u8 keep-masks, one per frame) fused into the output.
        dtype: np.uint8 or np.uint16 samples (strides in bytes)."""
        nf, nv = len(frames), len(views)
        fp = (C.c_void_p * max(nf, 1))(*[b.ptr for b in frames])
        dp = (C.c_void_p * max(nf * nv, 1))(*[b.ptr for b in dsts])
        va = (View * max(nv, 1))(*views)
        if masks is not None:
            if len(masks) != nf:
                raise ValueError("one mask per frame")
            if np.dtype(dtype) != np.uint8:
                raise Gs360Error(-4, "the fused keep-mask exists for 8-bit images only")
            mp = (C.c_void_p * max(nf, 1))(*[b.ptr for b in masks])
            _check(self.L.gs360_equirect_views_masked_u8(self.handle, fp, mp, nf, W, H, Cn, src_stride, 0, va, nv, dp,
                                                         dst_stride, interp, int(flags), slot), self.L)
            return
        fn = self.L.gs360_equirect_views_u16 if np.dtype(dtype) == np.uint16 else self.L.gs360_equirect_views_u8
        _check(fn(self.handle, fp, nf, W, H, Cn, src_stride, va, nv, dp, dst_stride, interp, int(flags), slot), self.L)

    def make_equirect_call(self, frames, W, H, Cn, views, dsts, slot=0, interp=INTERP_LINEAR, src_stride=0):
        """Pre-marshal one batched launch; returns a zero-argument callable (used by bench loops)."""
        nf, nv = len(frames), len(views)
        fp = (C.c_void_p * nf)(*[b.ptr for b in frames])
        dp = (C.c_void_p * (nf * nv))(*[b.ptr for b in dsts])
        va = (View * nv)(*views)
        fn, h, L = self.L.gs360_equirect_views_u8, self.handle, self.L

        def call():
            rc = fn(h, fp, nf, W, H, Cn, int(src_stride), va, nv, dp, 0, int(interp), 0, slot)
            if rc:
                _check(rc, L)
        call.keepalive = (fp, dp, va)
        return call

    def remap_table_dev(self, src, H, W, Cn, map_x, map_y, valid, h, w, dst, interp=INTERP_LINEAR,
                        border_value=(0, 0, 0, 0), fill_value=0, slot=0, dtype=np.uint8):
        bv = (C.c_double * 4)(*[float(x) for x in border_value])
        fn = self.L.gs360_remap_table_u16 if np.dtype(dtype) == np.uint16 else self.L.gs360_remap_table_u8
        _check(fn(self.handle, src.ptr, H, W, Cn, 0, map_x.ptr, map_y.ptr, valid.ptr if valid is not None else None, h, w, interp, bv,
                  int(fill_value), dst.ptr, 0, slot), self.L)

    def remap_tables_dev(self, jobs, Cn, interp=INTERP_LINEAR, border_value=(0, 0, 0, 0), slot=0, dtype=np.uint8):
        """Several remaps in one launch.  jobs: iterable of (src, H, W, map_x, map_y, valid_or_None, h, w, fill_value, dst)
        with DeviceBuffer objects for src / maps / valid / dst."""
        arr = (RemapJob * len(jobs))()
        for k, (src, H, W, mx, my, valid, h, w, fill, dst) in enumerate(jobs):
            arr[k] = RemapJob(src.ptr, H, W, 0, mx.ptr, my.ptr, valid.ptr if valid is not None else None, h, w, int(fill), dst.ptr, 0)
        bv = (C.c_double * 4)(*[float(x) for x in border_value])
        fn = self.L.gs360_remap_tables_u16 if np.dtype(dtype) == np.uint16 else self.L.gs360_remap_tables_u8
        _check(fn(self.handle, arr, len(jobs), Cn, interp, bv, slot), self.L)

    # -- map plans: the float maps of a run packed once (include/gs360.h) -------------------------
    MAP_PLAN_MAX_DIM = 4079

    def map_plan(self, map_x, map_y, valid, h, w, nearest=False, slot=0):
        """DeviceBuffers holding h x w float32 maps (+ uint8 valid or None) -> plan handle; the buffers may be freed afterwards."""
        hnd = C.c_void_p()
        _check(self.L.gs360_map_plan_create(self.handle, map_x.ptr, map_y.ptr, valid.ptr if valid is not None else None, int(h), int(w),
                                            1 if nearest else 0, slot, C.byref(hnd)), self.L)
        return hnd

    def map_plan_free(self, plan):
        if plan:
            _check(self.L.gs360_map_plan_destroy(self.handle, plan), self.L)

    def remap_plans_dev(self, jobs, Cn, interp=INTERP_LINEAR, border_value=(0, 0, 0, 0), slot=0, dtype=np.uint8):
        """remap_tables_dev with plans: jobs = iterable of (src, H, W, plan, use_valid, h, w, fill_value, dst)."""
        n = len(jobs)
        arr = (RemapJob * n)()
        pl = (C.c_void_p * n)()
        for k, (src, H, W, plan, use_valid, h, w, fill, dst) in enumerate(jobs):
            # valid: any non-NULL value asks for the plan's valid bit; the pointer is not read
            arr[k] = RemapJob(src.ptr, H, W, 0, None, None, dst.ptr if use_valid else None, h, w, int(fill), dst.ptr, 0)
            pl[k] = plan
        bv = (C.c_double * 4)(*[float(x) for x in border_value])
        fn = self.L.gs360_remap_plans_u16 if np.dtype(dtype) == np.uint16 else self.L.gs360_remap_plans_u8
        _check(fn(self.handle, arr, pl, n, Cn, interp, bv, slot), self.L)

    def fisheye_views_dev(self, lens_bufs, calibs, Cn, views, lens_fov_deg, dsts, valid_outs=None,
                          interp=INTERP_LINEAR, mask_outside=True, mask_value=0, slot=0):
        nv = len(views)
        sp = (C.c_void_p * nv)(*[b.ptr for b in lens_bufs])
        dp = (C.c_void_p * nv)(*[b.ptr for b in dsts])
        vo = (C.c_void_p * nv)(*[(b.ptr if b is not None else None) for b in valid_outs]) if valid_outs else None
        ca = (Calib * nv)(*calibs)
        va = (View * nv)(*views)
        _check(self.L.gs360_fisheye_views_u8(self.handle, sp, ca, Cn, 0, va, nv, float(lens_fov_deg), interp,
                                             1 if mask_outside else 0, int(mask_value), dp, 0, vo, slot), self.L)

    # -- input colour stage ------------------------------------------------------------------
    def color_plan(self, lut_table, level_pos, out_thresholds):
        """lut_table: float32 [n][n][n][3] ([b][g][r], .cube order); level_pos: float32 [3][256]; out_thresholds:
        float32 [256] (entry 0 unused).  Returns an opaque plan handle (free with color_plan_free)."""
        lut = np.ascontiguousarray(lut_table, dtype=np.float32)
        if lut.ndim != 4 or lut.shape[3] != 3 or not (lut.shape[0] == lut.shape[1] == lut.shape[2]):
            raise ValueError("lut_table must be [n][n][n][3]")
        pos = np.ascontiguousarray(level_pos, dtype=np.float32)
        thr = np.ascontiguousarray(out_thresholds, dtype=np.float32)
        if pos.shape != (3, 256) or thr.shape != (256,):
            raise ValueError("level_pos must be [3][256] and out_thresholds [256]")
        h = C.c_void_p()
        _check(self.L.gs360_color_plan_create(self.handle, lut.ctypes.data, int(lut.shape[0]), pos.ctypes.data,
                                              thr.ctypes.data, C.byref(h)), self.L)
        return h

    def color_plan_free(self, plan):
        if plan:
            _check(self.L.gs360_color_plan_destroy(self.handle, plan), self.L)

    def color_plan16(self, lut_table, domain_min, domain_max, n_pieces, start, base, off, thresholds):
        """16-bit colour plan (include/gs360.h): LUT + domain + the piecewise output thresholds of gs360.color.output_pieces16."""
        lut = np.ascontiguousarray(lut_table, dtype=np.float32)
        if lut.ndim != 4 or lut.shape[3] != 3 or not (lut.shape[0] == lut.shape[1] == lut.shape[2]):
            raise ValueError("lut_table must be [n][n][n][3]")
        dmin = np.ascontiguousarray(domain_min, dtype=np.float32)
        dmax = np.ascontiguousarray(domain_max, dtype=np.float32)
        st = np.ascontiguousarray(start, dtype=np.float32)
        ba = np.ascontiguousarray(base, dtype=np.int32)
        of = np.ascontiguousarray(off, dtype=np.int32)
        th = np.ascontiguousarray(thresholds, dtype=np.float32)
        if dmin.shape != (3,) or dmax.shape != (3,) or st.size < 4 or ba.size < 4 or of.size < 5:
            raise ValueError("bad colour plan tables")
        h = C.c_void_p()
        _check(self.L.gs360_color_plan16_create(self.handle, lut.ctypes.data, int(lut.shape[0]), dmin.ctypes.data, dmax.ctypes.data,
                                                int(n_pieces), st.ctypes.data, ba.ctypes.data, of.ctypes.data,
                                                th.ctypes.data if th.size else None, C.byref(h)), self.L)
        return h

    def color_plan16_free(self, plan):
        if plan:
            _check(self.L.gs360_color_plan16_destroy(self.handle, plan), self.L)

    def color_apply16_dev(self, plan, src, H, W, Cn, dst=None, red_index=0, src_stride=0, dst_stride=0, slot=0):
        _check(self.L.gs360_color_apply_u16(self.handle, plan, src.ptr, H, W, Cn, src_stride, red_index,
                                            (dst or src).ptr, dst_stride, slot), self.L)

    def color_apply_dev(self, plan, src, H, W, Cn, dst=None, red_index=0, src_stride=0, dst_stride=0, slot=0):
        """Apply a colour plan to a device-resident H x W x C image (in place when dst is None)."""
        _check(self.L.gs360_color_apply_u8(self.handle, plan, src.ptr, H, W, Cn, src_stride, red_index,
                                           (dst or src).ptr, dst_stride, slot), self.L)

    # -- hot path, host buffers (synchronous) -----------------------------------------------
    def equirect_views(self, src, views, slot=0, interp=INTERP_LINEAR, flags=0):
        """src: H x W x C uint8 (or uint16) ndarray -> list of per-view ndarrays (height x width x C) of the same dtype."""
        dt = np.uint16 if np.asarray(src).dtype == np.uint16 else np.uint8
        src = np.ascontiguousarray(src, dtype=dt)
        if src.ndim == 2:
            src = src[:, :, None]
        H, W, Cn = src.shape
        outs = [np.empty((v.height, v.width, Cn), dt) for v in views]
        if not views:
            return outs
        va = (View * len(views))(*views)
        dp = (C.c_void_p * len(views))(*[o.ctypes.data for o in outs])
        with self.slot_locks[slot]:
            fn = self.L.gs360_equirect_views_u16_host if dt == np.uint16 else self.L.gs360_equirect_views_u8_host
            _check(fn(self.handle, src.ctypes.data, W, H, Cn, src.strides[0], va, len(views), dp, 0, interp, int(flags), slot), self.L)
        return outs

    def remap(self, src, map_x, map_y, interpolation=INTERP_LINEAR, border_value=0.0, valid=None, fill_value=0,
              slot=0):
        """cv2.remap(src, map_x, map_y, interpolation, borderMode=BORDER_CONSTANT, borderValue=...) drop-in
        (DF:2001-2008); `valid`/`fill_value` fuse the reference's `out[~valid] = mask_value` (DF:2009-2014).
        uint16 sources take cv2's CV_16U (float-weight) samplers and return uint16."""
        dt = np.uint16 if np.asarray(src).dtype == np.uint16 else np.uint8
        src = np.ascontiguousarray(src, dtype=dt)
        s3 = src if src.ndim == 3 else src[:, :, None]
        H, W, Cn = s3.shape
        mx = np.ascontiguousarray(map_x, dtype=np.float32)
        my = np.ascontiguousarray(map_y, dtype=np.float32)
        if mx.shape != my.shape or mx.ndim != 2:
            raise ValueError("map_x / map_y must be 2-D arrays of equal shape")
        h, w = mx.shape
        if np.isscalar(border_value):
            border_value = (float(border_value), 0.0, 0.0, 0.0)  # cv::Scalar(v)
        bv = (C.c_double * 4)(*[float(b) for b in (list(border_value) + [0, 0, 0, 0])[:4]])
        va = None
        if valid is not None:
            va = np.ascontiguousarray(valid, dtype=np.uint8)
            if va.shape != (h, w):
                raise ValueError("valid mask shape mismatch")
        dst = np.empty((h, w, Cn), dt)
        with self.slot_locks[slot]:
            fn = self.L.gs360_remap_table_u16_host if dt == np.uint16 else self.L.gs360_remap_table_u8_host
            _check(fn(self.handle, s3.ctypes.data, H, W, Cn, s3.strides[0], mx.ctypes.data, my.ctypes.data,
                      va.ctypes.data if va is not None else None, h, w, int(interpolation), bv, int(fill_value), dst.ctypes.data, 0, slot),
                   self.L)
        return dst if src.ndim == 3 else dst[:, :, 0]
