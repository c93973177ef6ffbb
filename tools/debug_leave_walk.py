#!/usr/bin/env python3
"""Debugging: `popdel call` on a generated cohort with the window loop handed back to the host at round k (POPDEL_TEST_LEAVE_WALK_AT)
under several settings of the loader, each held against the run that stays on the device. usage: debug_leave_walk.py [samples=300] [k=3]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
k = sys.argv[2] if len(sys.argv) > 2 else "3"
d = tempfile.mkdtemp()
region = (1000000, 2000000)
lst = synth.write_cohort(d, n, region=region, deletions=synth.single_deletion, workers=8)
binary = os.path.join(ROOT, "popdel_amd", "lib", os.environ.get("POPDEL_BINARY", "popdel"))


def run(env):
    e = dict(os.environ, POPDEL_TIMING="1", POPDEL_DEVICE_GUNZIP_WAIT="1")
    e.update(env)
    out = os.path.join(d, "o.vcf")
    r = subprocess.run([binary, "call", lst, "-o", out, "-r", "chr20:%d-%d" % region], env=e, capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr[-1500:]
    return "".join(l for l in open(out) if not l.startswith("##fileDate")), "\n".join(l for l in r.stderr.splitlines() if "filing:" in l or "window loop" in l)


base, note = run({})
print("base:", note)
for env in ({"POPDEL_HOST_FILING": "1"}, {"POPDEL_HOST_RECORDS": "1"}, {"POPDEL_HOST_RECORDS": "1", "POPDEL_FULL_PARSE": "1"}, {}):
    e = dict(env, POPDEL_TEST_LEAVE_WALK_AT=k)
    got, note = run(e)
    print(e, "->", "same" if got == base else ("DIFFERENT" if got is not None else "FAILED"), "|", note)
    if got is not None and got != base:
        a, b = base.splitlines(), got.splitlines()
        i = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), None)
        print("   lines", len(a), len(b), "first differing line", i)
        if i is not None:
            fa, fb = a[i].split("\t"), b[i].split("\t")
            diff = [j for j in range(min(len(fa), len(fb))) if fa[j] != fb[j]]
            print("   fields that differ:", len(diff), "of", len(fa), "first:", [(j, fa[j][:80], fb[j][:80]) for j in diff[:4]])
