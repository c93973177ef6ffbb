#!/bin/bash
# GPU box: the product binary on one generated cohort (4,000 profiles x 4 Mb) under several settings of the environment,
# alternating on ONE box; prints wall / CPU of the steady and the warmed rounds. usage: tools/ab_env_e2e.sh "VAR=v VAR2=w" "-" ...
python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from popdel_amd import synth
d = "/tmp/e2e_cohort_4000x4"
if not os.path.exists(os.path.join(d, "profiles.txt")):
    synth.write_cohort(d, 4000, region=(1000000, 5000000), deletions=synth.single_deletion, workers=16)
PY
for k in 1 2 3; do for V in "$@"; do
  if [ "$V" = "-" ]; then E=""; else E="$V"; fi
  env $E POPDEL_TIMING=1 ./popdel_amd/lib/popdel call /tmp/e2e_cohort_4000x4/profiles.txt -o /tmp/o_ab.vcf -r chr20:1000000-5000000 2>&1 >/dev/null | python3 -c "
import re,sys
t=sys.stdin.read()
s=re.search(r'steady state: segments (\d+).*?wall ([0-9.]+) s', t); w=re.search(r'warmed rounds: segments (\d+).*?wall ([0-9.]+) s.*?CPU ([0-9.]+) s', t)
print('%-40s steady %.4f s/round  warmed %.4f s/round  warmed CPU %.3f s/round' % ('$V', float(s.group(2))/int(s.group(1)), float(w.group(2))/int(w.group(1)), float(w.group(3))/int(w.group(1))))"
done; done
