"""GPU: `popdel call` over forty segments -- the hand-over queue, the slot tables' rotation and the device-side walk state in
their steady state -- with the window loop on the device, with every hand-over carrying a copy of its entries, and walked on
the host: the same bytes (tools/long_run_check.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.gpu
def test_forty_segments_three_ways(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "long_run_check.py"), "128", "8"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "identical" in r.stdout and "40 segments in closed form on the device" in r.stdout, r.stdout[-1500:]
