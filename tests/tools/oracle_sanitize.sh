#!/bin/bash
# CPU-only sanitizer pass over the oracle (GPU ASan is unavailable on the pool): builds the ASan+UBSan variant and
# drives every exported function once through ctypes.
set -e
cd "$(dirname "$0")/../.."
make -C oracle asan >/dev/null
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
python - <<'PY'
import ctypes as C, pathlib, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import orc
so = pathlib.Path("oracle/libgs360oracle_asan.so").resolve()
orig = C.CDLL
C.CDLL = lambda p, *a, **k: orig(str(so), *a, **k) if "libgs360oracle.so" in str(p) else orig(p, *a, **k)
orc.lib(); C.CDLL = orig
from util import rand_image, ring_views, TEMPLATE_CALIB
src = rand_image(97, 131)
rng = np.random.default_rng(1)
mx = rng.uniform(-20, 150, (60, 70)).astype(np.float32); my = rng.uniform(-20, 120, (60, 70)).astype(np.float32)
mx[0, :3] = [np.nan, 1e30, -1e30]
for it in (0, 1, 2):
    orc.remap_u8(src, mx, my, interp=it, border_value=(1, 2, 3, 4), threads=2)
views = [orc.make_view(*s) for s in ring_views(5, 40, 100.0)] + [orc.make_view(0, 90, 120, 120, 33, 17), orc.make_view(180, -90, 60, 60, 5, 5)]
orc.equirect_views_u8(rand_image(64, 128), views, threads=3); orc.equirect_views_u8(rand_image(64, 128), views, threads=3, interp=2)
cal = orc.make_calib(**{**TEMPLATE_CALIB, "width": 240, "height": 240, "f": 65.0})
orc.fisheye_map(cal, 40, 10, 100, 100, 50, 40, 190.0, threads=2); orc.fisheye_spec_map(cal, 140, 0, 100, 100, 50, 40, 190.0)
orc.undistort_map(cal, 1.0, 190.0, threads=2); orc.equirect_distinct_texels(views[0], 128, 64); orc.table_distinct_texels(mx, my, 131, 97)
print("oracle ASan/UBSan pass: clean")
PY
