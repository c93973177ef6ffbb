"""CPU: the window loop in closed form (ChromProfile::planSegment + WalkReference, the host reference of the device
kernel) against the window loop itself. With POPDEL_WALK_CHECK=1 the oracle-backed `popdel call` driver plans every
segment in closed form before it walks it and compares shape, cursors, every pair instance (start, id, on, off) and the
number of pairs left active; any difference ends the run with an error. Cohorts: the golden 2-sample cohort on the region
that moves the slot borders off the index blocks (B1-B4 of SURVEY 8.1), heterogeneous libraries with multi-read-group and
low-coverage samples under every option set of tests/test_gpu_cli_options.py (several ROIs, -b, -A with read groups that
saturate), a candidate-dense cohort."""
import os
import re
import subprocess

import pytest

from conftest import ORACLE_CALL
from test_gpu_cli_options import OPTION_SETS, mixed_cohort  # noqa: F401  (fixture)


def checked(cohort_list, args, tmp_path):
    env = dict(os.environ, POPDEL_WALK_CHECK="1", POPDEL_THREADS="4")
    r = subprocess.run([ORACLE_CALL, cohort_list, "-o", str(tmp_path / "o.vcf")] + list(args), env=env, capture_output=True, text=True)
    assert r.returncode == 0, (r.stderr or r.stdout)[-1500:]
    m = re.search(r"\[popdel walk check\] (\d+) segments planned .* (\d+) not planned", r.stderr)
    assert m, r.stderr[-500:]
    return int(m.group(1)), int(m.group(2))


def test_golden_cohort_across_slot_borders(built, cohort_dir, tmp_path):
    from popdel_amd import synth
    d = cohort_dir("s2", 2, synth.single_deletion)
    lst = os.path.join(d, "profiles.txt")
    assert checked(lst, ["-r", "chr20:1000000-2000000"], tmp_path) == (5, 0)
    # the region of tests/golden/s2_active_r1003001.tsv.gz: slot borders off the 10-kb index blocks
    planned, skipped = checked(lst, ["-r", "chr20:1003001-2000000"], tmp_path)
    assert planned >= 5 and skipped == 0


@pytest.mark.parametrize("opts", OPTION_SETS, ids=lambda o: "_".join(x.replace(":", "").replace("-", "") for x in o)[:60])
def test_option_sets_on_the_heterogeneous_cohort(built, mixed_cohort, tmp_path, opts):
    planned, skipped = checked(os.path.join(mixed_cohort, "profiles.txt"), opts, tmp_path)
    assert planned >= 2 and skipped == 0


def test_saturating_read_groups_and_roi_files(built, mixed_cohort, tmp_path):
    opts = ["-A", os.path.join(mixed_cohort, "maxcov.txt"), "-R", os.path.join(mixed_cohort, "rois.txt"), "-n"]
    planned, _ = checked(os.path.join(mixed_cohort, "profiles.txt"), opts, tmp_path)
    assert planned >= 1


def test_candidate_dense_cohort(built, cohort_dir, tmp_path):
    from popdel_amd import synth
    d = cohort_dir("dd40", 40, synth.dense_deletions)
    assert checked(os.path.join(d, "profiles.txt"), ["-r", "chr20:1000000-2000000"], tmp_path) == (5, 0)
