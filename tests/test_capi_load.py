"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/gs360.h declares, and fails
loudly (no CPU fallback) when there is no GPU."""
import ctypes
import pathlib
import re

import pytest

import gs360
from conftest import ROOT

HEADER = (ROOT / "include" / "gs360.h").read_text()


def declared_functions():
    return sorted(set(re.findall(r"^\s*int\s+(gs360_\w+)\s*\(", HEADER, flags=re.M)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(gs360.capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(gs360.capi.LIB_PATH))
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.gs360_abi_version() == int(re.search(r"#define GS360_ABI_VERSION (\d+)", HEADER).group(1))


def test_binding_struct_layouts_match_the_header():
    """ctypes mirrors of the POD structs: sizes and field offsets as a C compiler lays out include/gs360.h (LP64)"""
    import subprocess
    import tempfile
    from gs360 import capi
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "gs360.h"
int main(void) {
    printf("%zu %zu %zu\n", sizeof(gs360_view), sizeof(gs360_calib), sizeof(gs360_remap_job));
    printf("%zu %zu %zu %zu %zu\n", offsetof(gs360_remap_job, src_stride), offsetof(gs360_remap_job, valid),
           offsetof(gs360_remap_job, fill_value), offsetof(gs360_remap_job, dst), offsetof(gs360_remap_job, dst_stride));
    return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        src = pathlib.Path(td) / "t.c"
        src.write_text(prog)
        exe = pathlib.Path(td) / "t"
        subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", str(exe), str(src)], check=True)
        out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    sizes, offs = [int(v) for v in out[:3]], [int(v) for v in out[3:]]
    assert sizes == [ctypes.sizeof(capi.View), ctypes.sizeof(capi.Calib), ctypes.sizeof(capi.RemapJob)] == [40, 96, 80]
    J = capi.RemapJob
    assert offs == [J.src_stride.offset, J.valid.offset, J.fill_value.offset, J.dst.offset, J.dst_stride.offset]


def test_no_torch_or_oracle_in_product_signatures_or_imports():
    pkg = ROOT / "360cam-pgm-3dgs-tools_amd"
    for path in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        text = path.read_text()
        assert "import torch" not in text and "from torch" not in text, path
        assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), path
        assert "gs360_oracle" not in text and "libgs360oracle" not in text, path
    assert "torch" not in HEADER.lower().replace("torch.distributed", "")


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    assert gs360.device_count() == 0
    with pytest.raises(gs360.Gs360Error) as e:
        gs360.Context(device=0)
    assert e.value.code == -3 and "no CPU path" in e.value.text
    from gs360 import engine
    with pytest.raises(gs360.Gs360Error):
        engine.Engine()


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(gs360.Gs360Error):
        gs360.load_library(tmp_path / "libgs360hip.so")


def test_run_one_reports_engine_failure_as_rc_and_text(tmp_path):
    """without a GPU the drop-in's run_one must return (rc != 0, text), never raise (PC:569-590 contract)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    import gs360_360PerspCut as cut
    from gs360 import imageio
    import numpy as np
    src = tmp_path / "pano.png"
    imageio.write_image(src, np.zeros((8, 16, 3), np.uint8))
    argv = ["ffmpeg", "-hide_banner", "-loglevel", "error", "-y", "-i", str(src), "-vf",
            "v360=input=equirect:output=rectilinear:w=8:h=8:yaw=0.0:pitch=0.0:roll=0:h_fov=90.0:v_fov=90.0:interp=cubic",
            "-threads", "1", "-frames:v", "1", str(tmp_path / "pano_A.png")]
    rc, text = cut.run_one(argv)
    assert rc == 1 and "gs360" in text
    assert not (tmp_path / "pano_A.png").exists()
