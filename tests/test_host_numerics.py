"""Numerics the device code relies on, checked on the CPU.

div_by_1em4 (historian_amd/csrc/hx_lse.h:17-23) replaces the reference's `x / 1e-4` (src/logsumexp.h:53-57) by a
multiply and two fused multiply-adds; bit-identity of the exact fill mode with the reference depends on that sequence
being the correctly rounded quotient.  The C program runs the same three operations (libm's fma is exact) against IEEE
division over every table bin boundary and its neighbours, random arguments of the look-up's range, random remainders,
and random bit patterns."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))


def test_division_by_1em4_sequence_is_ieee_division():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "div1em4_check")
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "csrc", "div1em4_check.c"), "-lm"], check=True)
        out = subprocess.run([exe, "20000000"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        checked, bad = (int(v) for v in out.stdout.split())
        assert checked > 4e7, checked
        assert bad == 0, out.stderr.decode()
