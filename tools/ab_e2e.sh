#!/bin/bash
# GPU box: two builds of the product binary on the same generated cohort (4,000 profiles x 4 Mb), alternating on ONE box (boxes
# differ by more than most changes are worth): copy the build to compare against to popdel_amd/lib/popdel_old first.
# Prints the POPDEL_TIMING lines of each run and whether the VCFs agree. usage: tools/ab_e2e.sh
python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from popdel_amd import synth
d = "/tmp/e2e_cohort_4000"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, 4000, region=(1000000, 5000000), deletions=synth.single_deletion, workers=16)
PY
for k in 1 2 3; do for B in popdel_old popdel; do POPDEL_TIMING=1 ./popdel_amd/lib/$B call /tmp/e2e_cohort_4000/profiles.txt -o /tmp/o_$B.vcf -r chr20:1000000-5000000 2>&1 >/dev/null | grep -E "popdel timing" | grep -v "parity\|window loop\|NUMA" | sed "s/^/$B: /" | cut -c1-420; done; done
cmp <(grep -v fileDate /tmp/o_popdel.vcf) <(grep -v fileDate /tmp/o_popdel_old.vcf) && echo same_vcf
