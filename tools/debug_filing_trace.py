#!/usr/bin/env python3
"""Debugging the filing from block summaries (CallRun::fileRun): `popdel call` on a small generated cohort with
POPDEL_FILING_TRACE=1 -- which blocks of the first samples' runs were taken, and what stopped the run of blocks.
usage: debug_filing_trace.py [samples=64] [mb=1]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = tempfile.mkdtemp()
region = (1000000, 1000000 + mb * 1000000)
lst = synth.write_cohort(d, n, region=region, deletions=synth.single_deletion, workers=8)
env = dict(os.environ, POPDEL_TIMING="1", POPDEL_DEVICE_GUNZIP_WAIT="1", POPDEL_FILING_TRACE="1")
r = subprocess.run([os.path.join(ROOT, "popdel_amd", "lib", "popdel"), "call", lst, "-o", d + "/o.vcf", "-r", "chr20:%d-%d" % region], env=env,
                   capture_output=True, text=True)
print("\n".join(l for l in r.stderr.splitlines() if "filing" in l))
