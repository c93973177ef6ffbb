#!/bin/bash
# GPU box: kernels AND memory copies of `popdel call` on the cohort-scale workload, one warmed round as a time line
# (what the loaders' streams do between two walk submissions). Usage: tools/loader_timeline.sh <tag> [samples] [mb]
set -u
TAG=${1:-loader}
N=${2:-4000}
MB=${3:-2}
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
POPDEL_KEEP_TEARDOWN=1 POPDEL_TIMING=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- "$ROOT/popdel_amd/lib/popdel" call /tmp/e2e_cohort_$N/profiles.txt -o /tmp/e2e_prof.vcf -r chr20:1000000-$END > "$OUT/trace.out" 2> "$OUT/trace.err"
K=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
M=$(find "$OUT/trace" -name "*memory_copy_trace.csv" | head -1)
python3 "$ROOT/tools/loader_timeline.py" "$K" "$M" > "$OUT/timeline.txt"
head -c 6000 "$OUT/timeline.txt"
rm -rf "$OUT/trace"
