"""A short run of the randomised parity campaign (tests/tools/fuzz_parity.py): random shapes, strides, channel counts,
views, maps (incl. NaN/inf), masks and interpolation modes for all three kernels, bit-exact against the oracle."""
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("lanemap,ring", [("", ""), ("rows", ""), ("blocked", ""), ("", "1"), ("blocked", "3"), ("staged", ""), ("staged", "2")])
def test_fuzz_parity_short(lanemap, ring):
    """lanemap forces one lane map of the equirect kernel, ring caps the members of a yaw ring (1 = no coordinate sharing)"""
    import os
    env = dict(os.environ)
    env.pop("GS360_LANEMAP", None)
    env.pop("GS360_RING", None)
    env.pop("GS360_STAGE", None)
    if lanemap == "staged":                            # the LDS-staged kernel forced on every call that can take it (auto: only calls dominated by pitched, >= 1.75-texel-step views)
        env["GS360_LANEMAP"], env["GS360_STAGE"] = "rows", "1"
    elif lanemap:
        env["GS360_LANEMAP"] = lanemap
    if ring:
        env["GS360_RING"] = ring
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "tools" / "fuzz_parity.py"), "--seconds", "8", "--seed", "77"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "failures=0" in r.stdout
