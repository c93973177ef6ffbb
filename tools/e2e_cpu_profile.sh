#!/bin/bash
# GPU box: where the CPU time of `popdel call`'s warmed rounds goes (4,000 generated profiles x 4 Mb by default).
# Usage: tools/e2e_cpu_profile.sh <tag> [samples] [mb] -> gpurun_out/cpu_<tag>.txt (POPDEL_TIMING lines + the top functions)
set -u
TAG=${1:-a}
N=${2:-4000}
MB=${3:-4}
END=$((1000000 + MB * 1000000))
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
OUT=$ROOT/gpurun_out/cpu_$TAG.txt
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
POPDEL_TIMING=1 POPDEL_CPU_PROFILE=/tmp/cpu_prof.txt POPDEL_CPU_PROFILE_FROM_ROUND=4 "$ROOT/popdel_amd/lib/popdel" call /tmp/e2e_cohort_$N/profiles.txt -o /tmp/e2e_prof.vcf -r chr20:1000000-$END > /dev/null 2> "$OUT"
python3 "$ROOT/tools/cpu_profile_symbols.py" /tmp/cpu_prof.txt 60 >> "$OUT" 2>&1
tail -80 "$OUT"
