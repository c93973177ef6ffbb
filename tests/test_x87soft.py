"""CPU: the integer model of the x87 multiply / divide / add / store-to-double (popdel_amd/csrc/hip/x87soft.h -- the code the
kernels run for reference-shift quotients on a knife edge) against the host FPU, bit for bit."""
import os
import subprocess

from conftest import ROOT


def test_x87_model_matches_the_fpu(tmp_path):
    exe = str(tmp_path / "x87soft_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "popdel_amd", "csrc", "hip"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "x87soft_check.cpp")])
    out = subprocess.run([exe, "400000"], capture_output=True, text=True)
    bad, total = map(int, out.stdout.split())
    assert out.returncode == 0 and bad == 0 and total > 1500000, (out.stdout, out.stderr[-2000:])
