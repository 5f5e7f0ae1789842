"""gs360 -- host side of the MI355X reprojection engine (ctypes over libgs360hip.so)."""
from .capi import (Calib, Context, Gs360Error, View, EQ_FISHEYE_OUT, INTERP_CUBIC, INTERP_LANCZOS4, INTERP_LINEAR, INTERP_NEAREST, device_count,  # noqa: F401
                   load_library)
