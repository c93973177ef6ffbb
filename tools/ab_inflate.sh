#!/bin/bash
# A/B of two builds of the HIP library on the gunzip kernel alone (tools/inflate_probe.py) and in `popdel call`
# (tools/e2e_cohort_probe.py): popdel_amd/lib/ab_old.so against popdel_amd/lib/ab_new.so, alternating.
# usage: tools/ab_inflate.sh [members=200000] [e2e samples=0 (skip)]
cd "$(dirname "$0")/.."
members=${1:-200000}
samples=${2:-0}
mkdir -p gpurun_out
for round in 1 2; do
  for which in old new; do
    cp popdel_amd/lib/ab_$which.so popdel_amd/lib/libpopdel_hip.so
    python tools/inflate_probe.py $members 20000 > gpurun_out/ab_inflate_${which}_$round.json 2> gpurun_out/ab_inflate_${which}_$round.err
    python - <<PY
import json
r = json.load(open("gpurun_out/ab_inflate_${which}_$round.json"))
print("$which $round: kernel_ms", r["gpu"]["kernel_ms"], "GB/s", round(r["gpu_inflated_GBs"], 2))
PY
  done
done
if [ "$samples" != "0" ]; then
  for which in old new old new; do
    cp popdel_amd/lib/ab_$which.so popdel_amd/lib/libpopdel_hip.so
    echo "e2e $which"; python tools/e2e_cohort_probe.py $samples 4 - 2>&1 | tail -2
  done
fi
cp popdel_amd/lib/ab_new.so popdel_amd/lib/libpopdel_hip.so
