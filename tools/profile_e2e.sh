#!/bin/bash
# GPU box: kernel trace of the product binary on the cohort-scale e2e workload (4,000 generated profiles x 2 Mb: 10 segments).
# Usage: tools/profile_e2e.sh <tag> [samples] [mb]  -> gpurun_out/prof_<tag>/ ; python tools/kstats.py <kernel_stats.csv> <segments> summarises per segment
set -u
TAG=${1:-e2e}
N=${2:-4000}
MB=${3:-2}
SEGS=$((MB * 5))
END=$((1000000 + MB * 1000000))
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from popdel_amd import synth
d = "/tmp/e2e_cohort_$N"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, $N, region=(1000000, $END), deletions=synth.single_deletion, workers=16)
PY
cd /tmp
"$ROOT/popdel_amd/lib/popdel" call /tmp/e2e_cohort_$N/profiles.txt -o /tmp/e2e_warm.vcf -r chr20:1000000-$END > /dev/null 2>&1
POPDEL_KEEP_TEARDOWN=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$ROOT/popdel_amd/lib/popdel" call /tmp/e2e_cohort_$N/profiles.txt -o /tmp/e2e_prof.vcf -r chr20:1000000-$END > "$OUT/trace.out" 2> "$OUT/trace.err"
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
python3 "$ROOT/tools/kstats.py" "$STATS" $SEGS > "$OUT/kernel_stats_per_segment.txt"
cp "$STATS" "$OUT/kernel_stats.csv"
cat "$OUT/kernel_stats_per_segment.txt"
rm -rf "$OUT/trace"
