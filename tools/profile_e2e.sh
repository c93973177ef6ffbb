#!/bin/bash
# GPU box: kernel trace of the product binary on the e2e workload (1,000 generated profiles x 1 Mb: 5 segments).
# Usage: tools/profile_e2e.sh <tag>   -> gpurun_out/prof_<tag>/ ; python tools/kstats.py <kernel_stats.csv> 5 summarises per segment
set -u
TAG=${1:-e2e}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from popdel_amd import synth
d = "/tmp/e2e_cohort_1000"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, 1000, deletions=synth.single_deletion, workers=16)
PY
cd /tmp
"$ROOT/popdel_amd/lib/popdel" call /tmp/e2e_cohort_1000/profiles.txt -o /tmp/e2e_warm.vcf -r chr20:1000000-2000000 > /dev/null 2>&1
POPDEL_KEEP_TEARDOWN=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$ROOT/popdel_amd/lib/popdel" call /tmp/e2e_cohort_1000/profiles.txt -o /tmp/e2e_prof.vcf -r chr20:1000000-2000000 > "$OUT/trace.out" 2> "$OUT/trace.err"
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
python3 "$ROOT/tools/kstats.py" "$STATS" 5 > "$OUT/kernel_stats_per_segment.txt"
cp "$STATS" "$OUT/kernel_stats.csv"
cat "$OUT/kernel_stats_per_segment.txt"
rm -rf "$OUT/trace"
