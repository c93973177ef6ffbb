#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from popdel_amd import synth
d = "/tmp/asan_cohort"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, 600, region=(1000000, 2000000), deletions=synth.single_deletion, workers=16)
PY
export POPDEL_DEVICE_GUNZIP_WAIT=1 POPDEL_TIMING=1
for v in "POPDEL_TEST_LEAVE_WALK_AT=3" "POPDEL_TEST_LEAVE_WALK_AT=3" "POPDEL_TEST_LEAVE_WALK_AT=3 POPDEL_LOADER_DEPTH=1" "POPDEL_TEST_LEAVE_WALK_AT=3 POPDEL_LOADER_TRACE=1"; do
  echo "== $v"
  env $v "$ROOT/popdel_amd/lib/popdel" call /tmp/asan_cohort/profiles.txt -o /tmp/x.vcf -r chr20:1003001-2000000 -b 60000 2>&1 > /dev/null | grep -E "filing:|window loop:|popdel loader|rounds fetched" | cut -c1-420 | head -30
done
