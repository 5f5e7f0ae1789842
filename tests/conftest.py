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
