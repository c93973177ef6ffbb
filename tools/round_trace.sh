#!/bin/bash
# GPU box: the host's own time line of `popdel call` (POPDEL_TIMING=2: loader, batches, hand-over) on the cohort-scale workload,
# the lines of a few warmed rounds. Usage: tools/round_trace.sh [samples=4000] [mb=2] [binary=popdel]
N=${1:-4000}
MB=${2:-2}
BIN=${3:-popdel}
END=$((1000000 + MB * 1000000))
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from popdel_amd import synth
d = "/tmp/e2e_cohort_$N"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, $N, region=(1000000, $END), deletions=synth.single_deletion, workers=16)
PY
"$ROOT/popdel_amd/lib/$BIN" call /tmp/e2e_cohort_$N/profiles.txt -o /tmp/e2e_warm.vcf -r chr20:1000000-$END > /dev/null 2>&1
POPDEL_TIMING=2 "$ROOT/popdel_amd/lib/$BIN" call /tmp/e2e_cohort_$N/profiles.txt -o /tmp/e2e_t.vcf -r chr20:1000000-$END 2> /tmp/trace.err > /dev/null
mkdir -p "$ROOT/gpurun_out"
grep "popdel timeline" /tmp/trace.err > "$ROOT/gpurun_out/round_trace_$BIN.txt"
python3 - <<PY
lines = [l.split(" ", 3) for l in open("$ROOT/gpurun_out/round_trace_$BIN.txt")]
ev = []
for l in lines:
    try:
        ev.append((float(l[2]), l[3].strip()))
    except (ValueError, IndexError):
        pass
ev.sort()
# two warmed rounds (the batches' "switches done" lines left out)
begins = [i for i, e in enumerate(ev) if e[1].startswith("round begins")]
a, b = begins[-4], begins[-2]
t0 = ev[a][0]
for t, w in ev[a:b + 1]:
    if not w.startswith("batch: switches done"):
        print("%9.2f  %s" % ((t - t0) * 1e3, w[:110]))
PY
