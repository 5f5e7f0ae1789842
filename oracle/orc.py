"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE -- see gs360_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None


class OrcView(C.Structure):
    _fields_ = [("yaw_deg", C.c_double), ("pitch_deg", C.c_double),
                ("hfov_deg", C.c_double), ("vfov_deg", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32)]


class OrcCalib(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32)] + [
        (n, C.c_double) for n in ("f", "cx", "cy", "k1", "k2", "k3", "k4", "p1", "p2", "b1", "b2")]


def build(force=False):
    so = _HERE / "libgs360oracle.so"
    src = _HERE / "gs360_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-s"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "libgs360oracle.so"
        if not so.exists():
            build()
        L = C.CDLL(str(so))
        u8p, f32p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_remap_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_long, C.c_int]
        L.orc_valid_fill.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fisheye_map.argtypes = [C.POINTER(OrcCalib)] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_double, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_undistort_map.argtypes = [C.POINTER(OrcCalib), C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int]
        L.orc_equirect_map.argtypes = [C.POINTER(OrcView), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_equirect_views_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.POINTER(OrcView),
                                            C.c_int, C.POINTER(C.c_void_p), C.c_long, C.c_int]
        L.orc_equirect_views_u8_interp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.POINTER(OrcView),
                                                   C.c_int, C.POINTER(C.c_void_p), C.c_long, C.c_int, C.c_int]
        L.orc_cubic_table.argtypes = [C.c_void_p]
        L.orc_equirect_map_proj.argtypes = [C.POINTER(OrcView), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_equirect_fisheye_views_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.POINTER(OrcView),
                                                    C.c_int, C.POINTER(C.c_void_p), C.c_long, C.c_int, C.c_int]
        L.orc_lanczos4_table.argtypes = [C.c_void_p]
        L.orc_equirect_views_masked_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long,
                                                   C.POINTER(OrcView), C.c_int, C.POINTER(C.c_void_p), C.c_long, C.c_int, C.c_int]
        L.orc_remap_u16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p, C.c_long, C.c_int]
        L.orc_valid_fill_u16.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_equirect_views_u16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.POINTER(OrcView),
                                             C.c_int, C.POINTER(C.c_void_p), C.c_long, C.c_int, C.c_int, C.c_int]
        L.orc_equirect_distinct_texels.argtypes = [C.POINTER(OrcView), C.c_int, C.c_int, C.c_void_p]
        L.orc_equirect_distinct_texels.restype = C.c_long
        L.orc_table_distinct_texels.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int]
        L.orc_table_distinct_texels.restype = C.c_long
        L.orc_fisheye_spec_map.argtypes = [C.POINTER(OrcCalib)] + [C.c_double] * 4 + [C.c_int, C.c_int, C.c_double,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
        del u8p, f32p, i32p
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_view(yaw, pitch, hfov, vfov, w, h):
    return OrcView(float(yaw), float(pitch), float(hfov), float(vfov), int(w), int(h))


def make_calib(width, height, f, cx=0.0, cy=0.0, k1=0.0, k2=0.0, k3=0.0, k4=0.0, p1=0.0, p2=0.0, b1=0.0, b2=0.0):
    return OrcCalib(int(width), int(height), *[float(v) for v in (f, cx, cy, k1, k2, k3, k4, p1, p2, b1, b2)])


def remap_u8(src, map_x, map_y, interp=1, border_value=(0, 0, 0, 0), threads=1):
    """cv2.remap(src, map_x, map_y, interp, BORDER_CONSTANT, borderValue) restatement (u8)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    s3 = src if src.ndim == 3 else src[:, :, None]
    H, W, Cn = s3.shape
    mx = np.ascontiguousarray(map_x, dtype=np.float32)
    my = np.ascontiguousarray(map_y, dtype=np.float32)
    h, w = mx.shape
    if np.isscalar(border_value):
        border_value = (float(border_value), 0.0, 0.0, 0.0)  # cv::Scalar(v) from a Python float
    bv = (C.c_double * 4)(*[float(b) for b in (list(border_value) + [0, 0, 0, 0])[:4]])
    dst = np.empty((h, w, Cn), dtype=np.uint8)
    rc = lib().orc_remap_u8(_ptr(s3), H, W, Cn, s3.strides[0], _ptr(mx), _ptr(my), h, w, int(interp), bv,
                            _ptr(dst), dst.strides[0], int(threads))
    if rc != 0:
        raise RuntimeError(f"orc_remap_u8 rc={rc}")
    return dst if src.ndim == 3 else dst[:, :, 0]


def valid_fill(img, valid, fill):
    a = img if img.ndim == 3 else img[:, :, None]
    v = np.ascontiguousarray(valid, dtype=np.uint8)
    fn = lib().orc_valid_fill_u16 if img.dtype == np.uint16 else lib().orc_valid_fill
    fn(_ptr(a), a.strides[0], a.shape[0], a.shape[1], a.shape[2], _ptr(v), int(fill))
    return img


def remap_u16(src, map_x, map_y, interp=1, border_value=(0, 0, 0, 0), threads=1):
    """cv2.remap on a CV_16U source (float-weight samplers), BORDER_CONSTANT."""
    src = np.ascontiguousarray(src, dtype=np.uint16)
    s3 = src if src.ndim == 3 else src[:, :, None]
    H, W, Cn = s3.shape
    mx = np.ascontiguousarray(map_x, dtype=np.float32)
    my = np.ascontiguousarray(map_y, dtype=np.float32)
    h, w = mx.shape
    if np.isscalar(border_value):
        border_value = (float(border_value), 0.0, 0.0, 0.0)
    bv = (C.c_double * 4)(*[float(b) for b in (list(border_value) + [0, 0, 0, 0])[:4]])
    dst = np.empty((h, w, Cn), dtype=np.uint16)
    rc = lib().orc_remap_u16(_ptr(s3), H, W, Cn, s3.strides[0], _ptr(mx), _ptr(my), h, w, int(interp), bv,
                             _ptr(dst), dst.strides[0], int(threads))
    if rc != 0:
        raise RuntimeError(f"orc_remap_u16 rc={rc}")
    return dst if src.ndim == 3 else dst[:, :, 0]


def equirect_views_u16(src, views, threads=1, interp=1, fisheye=False):
    src = np.ascontiguousarray(src, dtype=np.uint16)
    H, W, Cn = src.shape
    arr = (OrcView * len(views))(*views)
    outs = [np.empty((v.height, v.width, Cn), np.uint16) for v in views]
    ptrs = (C.c_void_p * len(views))(*[o.ctypes.data for o in outs])
    rc = lib().orc_equirect_views_u16(_ptr(src), W, H, Cn, src.strides[0], arr, len(views), ptrs, 0, int(interp), int(bool(fisheye)),
                                      int(threads))
    if rc != 0:
        raise RuntimeError(f"orc_equirect_views_u16 rc={rc}")
    return outs


def fisheye_map(calib, yaw, pitch, hfov, vfov, w, h, lens_fov, numpy2=True, threads=1):
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    va = np.empty((h, w), np.uint8)
    rc = lib().orc_fisheye_map(C.byref(calib), yaw, pitch, hfov, vfov, w, h, lens_fov, int(numpy2),
                               _ptr(mx), _ptr(my), _ptr(va), threads)
    assert rc == 0
    return mx, my, va.astype(bool)


def fisheye_spec_map(calib, yaw, pitch, hfov, vfov, w, h, lens_fov):
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    va = np.empty((h, w), np.uint8)
    rc = lib().orc_fisheye_spec_map(C.byref(calib), yaw, pitch, hfov, vfov, w, h, lens_fov,
                                    _ptr(mx), _ptr(my), _ptr(va))
    assert rc == 0
    return mx, my, va.astype(bool)


def undistort_map(calib, zoom, lens_fov, threads=1):
    h, w = calib.height, calib.width
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    va = np.empty((h, w), np.uint8)
    rc = lib().orc_undistort_map(C.byref(calib), float(zoom), float(lens_fov), _ptr(mx), _ptr(my), _ptr(va), threads)
    assert rc == 0
    return mx, my, va.astype(bool)


def equirect_map(view, W, H):
    sx = np.empty((view.height, view.width), np.int32)
    sy = np.empty((view.height, view.width), np.int32)
    rc = lib().orc_equirect_map(C.byref(view), W, H, _ptr(sx), _ptr(sy))
    assert rc == 0
    return sx, sy


def equirect_fisheye_map(view, W, H):
    """quantised (1/32 px) source coordinates of an equidistant-fisheye output view (hfov/vfov = its full field of view)"""
    sx = np.empty((view.height, view.width), np.int32)
    sy = np.empty((view.height, view.width), np.int32)
    rc = lib().orc_equirect_map_proj(C.byref(view), W, H, 1, _ptr(sx), _ptr(sy))
    assert rc == 0
    return sx, sy


def equirect_fisheye_views_u8(src, views, threads=1, interp=1):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, W, Cn = src.shape
    arr = (OrcView * len(views))(*views)
    outs = [np.empty((v.height, v.width, Cn), np.uint8) for v in views]
    ptrs = (C.c_void_p * len(views))(*[o.ctypes.data for o in outs])
    rc = lib().orc_equirect_fisheye_views_u8(_ptr(src), W, H, Cn, src.strides[0], arr, len(views), ptrs, 0, int(interp), int(threads))
    if rc != 0:
        raise RuntimeError(f"orc_equirect_fisheye_views_u8 rc={rc}")
    return outs


def cubic_table():
    """OpenCV's INTER_CUBIC fixed-point table as restated by the oracle: int16 [fy][fx][ky][kx]."""
    t = np.zeros(32 * 32 * 16, np.int16)
    lib().orc_cubic_table(_ptr(t))
    return t.reshape(32, 32, 4, 4)


def lanczos4_table():
    """OpenCV's INTER_LANCZOS4 fixed-point table as restated by the oracle: int16 [fy][fx][ky][kx] (8x8 taps)."""
    t = np.zeros(32 * 32 * 64, np.int16)
    lib().orc_lanczos4_table(_ptr(t))
    return t.reshape(32, 32, 8, 8)


def equirect_views_u8(src, views, threads=1, interp=1, mask=None):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    H, W, Cn = src.shape
    arr = (OrcView * len(views))(*views)
    outs = [np.empty((v.height, v.width, Cn), np.uint8) for v in views]
    ptrs = (C.c_void_p * len(views))(*[o.ctypes.data for o in outs])
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (H, W)
        rc = lib().orc_equirect_views_masked_u8(_ptr(src), _ptr(m), W, H, Cn, src.strides[0], m.strides[0], arr, len(views),
                                                ptrs, 0, int(interp), int(threads))
    else:
        rc = lib().orc_equirect_views_u8_interp(_ptr(src), W, H, Cn, src.strides[0], arr, len(views), ptrs, 0, int(interp),
                                                int(threads))
    if rc != 0:
        raise RuntimeError(f"orc_equirect_views_u8 rc={rc}")
    return outs


def equirect_distinct_texels(view, W, H, union=None):
    return int(lib().orc_equirect_distinct_texels(C.byref(view), W, H, _ptr(union) if union is not None else None))


def table_distinct_texels(map_x, map_y, W, H):
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    return int(lib().orc_table_distinct_texels(_ptr(mx), _ptr(my), mx.size, W, H))
