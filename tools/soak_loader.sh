#!/bin/bash
# GPU box: `popdel call` over and over on a 600-sample cohort with small window buffers (many short rounds: the loaders' threads,
# the starter callbacks and the hand-over thread meet as often as they can) -- every run under a time limit, every VCF the same
# bytes. Usage: tools/soak_loader.sh [runs=30]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
RUNS=${1:-30}
python3 - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from popdel_amd import synth
d = "/tmp/asan_cohort"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, 600, region=(1000000, 2000000), deletions=synth.single_deletion, workers=16)
PY
export POPDEL_DEVICE_GUNZIP_WAIT=1
first=""
bad=0
for i in $(seq 1 $RUNS); do
  case $((i % 3)) in 0) args="-r chr20:1000000-2000000 -b 25000";; 1) args="-r chr20:1003001-2000000 -b 40000 -n";; 2) args="-r chr20:1000000-1400000 -r chr20:1500000-2000000 -b 30000";; esac
  timeout 120 "$ROOT/popdel_amd/lib/popdel" call /tmp/asan_cohort/profiles.txt -o /tmp/soak.vcf $args > /dev/null 2> /tmp/soak.err
  rc=$?
  sum=$(grep -v fileDate /tmp/soak.vcf | md5sum | cut -c1-12)
  key="k$((i % 3))"
  eval "ref=\${$key:-}"
  if [ -z "$ref" ]; then eval "$key=$sum"; ref=$sum; fi
  if [ $rc -ne 0 ] || [ "$sum" != "$ref" ]; then echo "run $i ($args): rc $rc sum $sum (first $ref)"; tail -3 /tmp/soak.err; bad=$((bad + 1)); fi
done
echo "$RUNS runs, $bad bad"
exit $bad
