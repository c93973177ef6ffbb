#!/bin/bash
# GPU box: the AddressSanitizer build of the product's host side (tools/asan_product.sh, built beforehand) on a 600-sample cohort,
# several option sets, against the plain binary: same bytes, no report.
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
rc=0
for args in "-r chr20:1000000-2000000" "-r chr20:1000000-1300000 -r chr20:1500000-2000000" "-r chr20:1003001-2000000 -b 60000 -n"; do
  "$ROOT/popdel_amd/lib/popdel" call /tmp/asan_cohort/profiles.txt -o /tmp/plain.vcf $args > /dev/null 2>&1
  ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0:use_sigaltstack=0 "$ROOT/popdel_amd/lib/popdel_asan" call /tmp/asan_cohort/profiles.txt -o /tmp/asan.vcf $args > /dev/null 2> /tmp/asan.err
  code=$?
  if [ $code -ne 0 ] || grep -q "AddressSanitizer" /tmp/asan.err || ! cmp -s <(grep -v fileDate /tmp/plain.vcf) <(grep -v fileDate /tmp/asan.vcf); then
    echo "FAILED: $args (exit $code)"; grep -A25 "AddressSanitizer" /tmp/asan.err | head -40; rc=1
  else
    echo "ok: $args ($(grep -c -v '^#' /tmp/asan.vcf) records)"
  fi
done
exit $rc
