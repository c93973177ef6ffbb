#!/bin/bash
# GPU box: kernel trace and PMC passes of the bench workload. Usage: tools/profile_round.sh <tag>   (writes gpurun_out/prof_<tag>/)
# Counters go in their own runs (one hardware block's slots per pass), with --kernel-trace only.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --serial --lanes 1 --no-legs --no-cpu"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
# 1. kernel trace of one chr20-sized step, one ticket at a time (see bench.py --serial)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 1 --warmup 1 > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
# 2. PMC passes on four passes over the resident segment
pass() {
    name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- $BENCH --passes 4 --steps 1 --warmup 0 > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" || echo "pass $name failed" >> "$OUT/failed.txt"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
ls "$OUT"
# 3. the context as the bench drives it (three lanes, six tickets in flight): kernel trace of 60 passes for the time line of
#    the lanes' overlap (tools/overlap_timeline.py)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/lanes" -- python3 $ROOT/bench.py --no-legs --no-cpu --steps 1 --warmup 0 --passes 60 > "$OUT/lanes_bench.json" 2> "$OUT/lanes.err"
f=$(find "$OUT/lanes" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 $ROOT/tools/overlap_timeline.py "$f" 12 160 > "$OUT/lanes_timeline.txt"
ls "$OUT"
