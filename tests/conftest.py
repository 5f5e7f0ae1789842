import pathlib
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
PKG = ROOT / "360cam-pgm-3dgs-tools_amd"
for p in (str(ROOT), str(PKG), str(PKG / "cli_tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (lib/ is git-ignored): build it once with hipcc if it is there
    lib = PKG / "lib" / "libgs360hip.so"
    if not lib.exists():
        import shutil
        import subprocess
        if shutil.which("hipcc") or pathlib.Path("/opt/rocm/bin/hipcc").exists():
            subprocess.run(["make", "-C", str(PKG / "csrc")], check=False)


@pytest.fixture(scope="session")
def ctx():
    """One engine context for the whole GPU session (fails loudly if the HIP library/GPU is absent)."""
    import gs360
    c = gs360.Context(device=0, n_slots=2)
    yield c
    c.close()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as _orc
    _orc.build()
    return _orc
