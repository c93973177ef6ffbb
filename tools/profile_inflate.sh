#!/bin/bash
# GPU box: instruction mix and lane utilisation of inflateMembersKernel on the member population of tools/inflate_probe.py.
# Usage: tools/profile_inflate.sh <tag> [members]  -> gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-inf}
N=${2:-100000}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PROBE="python3 $ROOT/tools/inflate_probe.py $N 10000"
pass() {
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- $PROBE > "$OUT/pmc_$name.txt" 2> "$OUT/pmc_$name.err" || echo "pass $name failed" >> "$OUT/failed.txt"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass sq3 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 $ROOT/tools/pmcsum.py "$OUT" "$OUT/summary.json" > "$OUT/summary.txt" 2>&1
# (pmcsum keeps pd:: kernels only: the inflate kernel lives in an anonymous namespace)
python3 - <<PY > "$OUT/summary.txt"
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "inflateMembersKernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-28s launches %3d  mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
cat "$OUT/summary.txt"
