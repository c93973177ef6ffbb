import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libpopdel_oracle.so")
ORACLE_CALL = os.path.join(ROOT, "oracle", "_build", "popdel_oracle_call")
HIP_LIB = os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_hip.so")
# test build of the HIP library (stage 2 compiled with PD_TEST_SNAP_SHIFTS), lives beside the oracle
HIP_SNAP_LIB = os.path.join(ROOT, "oracle", "_build", "libpopdel_hip_snap.so")
POPDEL_BIN = os.path.join(ROOT, "popdel_amd", "lib", "popdel")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test (enabled with POPDEL_SLOW=1)")


@pytest.fixture(scope="session")
def built():
    """Builds the oracle (always) and the product libraries (cross-compiled; needs hipcc) once per session."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j4"], stdout=subprocess.DEVNULL)
    if not (os.path.exists(HIP_LIB) and os.path.exists(POPDEL_BIN) and os.path.exists(HIP_SNAP_LIB)):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "popdel_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    return True


@pytest.fixture(scope="session")
def cohort_dir(tmp_path_factory):
    """Directory factory for synthetic cohorts, cached per (name) for the session."""
    cache = {}

    def make(name, n_samples, deletions, compressed=True):
        from popdel_amd import synth
        key = (name, n_samples, compressed)
        if key not in cache:
            d = tmp_path_factory.mktemp(name)
            synth.write_cohort(str(d), n_samples, deletions=deletions, compressed=compressed)
            cache[key] = str(d)
        return cache[key]

    return make


def strip_date(text):
    return "".join(l for l in text.splitlines(True) if not l.startswith("##fileDate"))


def run_call(binary, cohort, out, *args):
    cmd = [binary] + (["call"] if os.path.basename(binary) == "popdel" else []) + [os.path.join(cohort, "profiles.txt"), "-o", out,
                                                                                  "-r", "chr20:1000000-2000000"] + list(args)
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return strip_date(open(out).read())
