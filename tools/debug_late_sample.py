#!/usr/bin/env python3
"""Debugging: a one-contig cohort in which one sample's data begins late (and ends early): the device's closed-form window loop
against the host's loop (POPDEL_HOST_WALK=1) and the CPU driver. usage: debug_late_sample.py [samples=12] [first=1150000] [last=1420000] [deletion at=1250000]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1150000
last = int(sys.argv[3]) if len(sys.argv) > 3 else 1420000
where = int(sys.argv[4]) if len(sys.argv) > 4 else 1250000       # the planted deletion
d = tempfile.mkdtemp()
paths = []
for k in range(n):
    lib = synth.Library(hist_seed=100 + k)
    pos, dev = synth.sample_pairs(9100 + k, (1000000, 1500000), lib, [synth.Deletion(where, 1500, [2, 0, 1][k % 3])])
    if k == 5:
        keep = (pos >= first) & (pos < last)
        pos, dev = pos[keep], dev[keep]
    p = os.path.join(d, "l%02d.profile" % k)
    synth.write_profile(p, "L%02d" % k, pos, dev, lib)
    paths.append(p)
lst = os.path.join(d, "profiles.txt")
open(lst, "w").write("\n".join(paths) + "\n")
args = ["-n"]
oracle = os.path.join(ROOT, "oracle", "_build", "popdel_oracle_call")
subprocess.check_call([oracle, lst, "-o", os.path.join(d, "cpu.vcf")] + args, stdout=subprocess.DEVNULL)
strip = lambda path: [l for l in open(path) if not l.startswith("##fileDate")]
cpu = strip(os.path.join(d, "cpu.vcf"))
for env in ({}, {"POPDEL_HOST_WALK": "1"}, {"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"}):
    e = dict(os.environ, POPDEL_TIMING="1", POPDEL_DEVICE_GUNZIP_WAIT="1")
    e.update(env)
    out = os.path.join(d, "o.vcf")
    r = subprocess.run([os.path.join(ROOT, "popdel_amd", "lib", "popdel"), "call", lst, "-o", out] + args, env=e, capture_output=True, text=True)
    if r.returncode != 0:
        print(env, "failed:", r.stderr[-300:])
        continue
    got = strip(out)
    bad = [i for i in range(min(len(cpu), len(got))) if cpu[i] != got[i]]
    print(env, "->", "same as the CPU driver" if got == cpu else "%d of %d lines differ (lengths %d / %d)" % (len(bad), len(cpu), len(cpu), len(got)))
    for i in bad[:3]:
        fa, fb = cpu[i].split("\t"), got[i].split("\t")
        diff = [j for j in range(min(len(fa), len(fb))) if fa[j] != fb[j]]
        print("    line", i, "pos", fa[1], fb[1], "fields", diff[:6], [(fa[j][:60], fb[j][:60]) for j in diff[:2]])
