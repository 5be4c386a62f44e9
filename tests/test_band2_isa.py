"""The hand-issued row-record fetches of the two-pairs-per-wavefront banded sweep (historian_amd/csrc/hx_band2.hip) rely on the
compiler leaving their landing registers alone between a fetch and the counted wait in front of its first use.  This test
compiles the file to assembly with the product's flags and checks exactly that (tools/check_landing_regs.py): every fetch
writes the one announced register pair, and nothing but the moves directly behind an `s_waitcnt vmcnt(N)` reads it."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_landing_registers_of_the_row_record_fetches(tmp_path):
    import check_landing_regs
    src = os.path.join(ROOT, "historian_amd", "csrc", "hx_band2.hip")
    out = str(tmp_path / "hx_band2.s")
    # the flags of historian_amd/csrc/Makefile
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function",
                    "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", out, src],
                   check=True, capture_output=True, timeout=600)
    kernels, problems = check_landing_regs.check(out)
    assert kernels == 12, kernels          # {trunc, linear} x {1, 2, 4 wavefronts} x {Forward, Backward}
    assert not problems, problems
