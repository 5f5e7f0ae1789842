"""A short run of the randomised parity campaign (tests/tools/fuzz_parity.py): random shapes, strides, channel counts,
views, maps (incl. NaN/inf), masks and interpolation modes for all three kernels, bit-exact against the oracle."""
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("lanemap,ring", [("", ""), ("rows", ""), ("blocked", ""), ("", "1"), ("blocked", "3"), ("staged", ""), ("staged", "2"),
                                          ("srcmajor", ""), ("srcmajor-only", ""), ("tablestage-only", ""), ("tablestage-forced", "")])
def test_fuzz_parity_short(lanemap, ring):
    """lanemap forces one lane map of the equirect kernel, ring caps the members of a yaw ring (1 = no coordinate sharing); "srcmajor" forces
    the source-major kernel onto every call whose geometry fits it, "srcmajor-only" spends the whole run on such rings (context options:
    include/gs360.h, gs360_ctx_set_option -- the library reads no environment variable after context creation)"""
    opts, extra = [], []
    if lanemap == "staged":                            # the LDS-staged kernel forced on every call that can take it (auto: only calls dominated by pitched, >= 1.75-texel-step views)
        opts += ["lanemap=0", "stage=1"]
    elif lanemap == "srcmajor":
        opts += ["srcmajor=1"]
    elif lanemap == "srcmajor-only":
        opts += ["srcmajor=1"]
        extra = ["--only", "srcmajor"]
    elif lanemap == "tablestage-only":                 # the LDS-staged table kernel's own family (smooth maps, several jobs per call, padded outputs)
        extra = ["--only", "tablestage"]
    elif lanemap == "tablestage-forced":               # every family with the staged table kernel forced onto each plan job it can take (random maps: all SLOW)
        opts += ["table_stage=1"]
    elif lanemap:
        opts += [f"lanemap={1 if lanemap == 'blocked' else 0}"]
    if ring:
        opts += [f"ring={ring}"]
    cmd = [sys.executable, str(ROOT / "tests" / "tools" / "fuzz_parity.py"), "--seconds", "8", "--seed", "77"] + extra
    for o in opts:
        cmd += ["--option", o]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "failures=0" in r.stdout
