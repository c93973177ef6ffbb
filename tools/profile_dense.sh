#!/bin/bash
# GPU box: PMC passes of the candidate-dense segment (tools/dense_probe.py: 1,796 cells per pass), for the stage-2 kernels
# where stage 2 dominates. Usage: tools/profile_dense.sh <tag>   (writes gpurun_out/prof_<tag>/; summarise with tools/pmcsum.py)
set -u
TAG=${1:-dense}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PROBE="python3 $ROOT/tools/dense_probe.py 10000 2"
pass() {
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- $PROBE > "$OUT/pmc_$name.txt" 2> "$OUT/pmc_$name.err" || echo "pass $name failed" >> "$OUT/failed.txt"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls "$OUT"
