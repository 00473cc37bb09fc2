"""div_const<C>() (nano_amd/csrc/device_common.h) replaces the quantizers' divisions by 15, 63 and 127 with
multiply + two FMAs.  The claim -- equal to the IEEE quotient for every finite float except -0 -- is checked exhaustively
by tools/div_const_check.c (about a minute); this test runs the same program on every 251st bit pattern."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_three_operation_quotient_equals_the_ieee_division(tmp_path):
    exe = str(tmp_path / "chk")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "div_const_check.c"), "-lm"])
    out = subprocess.check_output([exe, "251"], text=True)
    rows = re.findall(r"c=(\d+) .* mismatches=(\d+) .* sample ([0-9a-f]{8})", out)
    assert [int(c) for c, _, _ in rows] == [15, 63, 127], out
    for _, bad, sample in rows:
        assert int(bad) == 0 or (int(bad) == 1 and sample == "80000000"), out      # -0 is the one exception (never a maximum)
