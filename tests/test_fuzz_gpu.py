"""A short run of the randomised parity campaign (tests/tools/fuzz_parity.py): random shapes, strides, channel counts,
views, maps (incl. NaN/inf), masks and interpolation modes for all three kernels, bit-exact against the oracle."""
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("lanemap", ["", "rows", "blocked"])
def test_fuzz_parity_short(lanemap):
    import os
    env = dict(os.environ)
    env.pop("GS360_LANEMAP", None)
    if lanemap:
        env["GS360_LANEMAP"] = lanemap
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "fuzz_parity.py"), "--seconds", "8", "--seed", "77"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "failures=0" in r.stdout
