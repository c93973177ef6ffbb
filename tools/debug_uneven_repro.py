#!/usr/bin/env python3
"""Debugging: the cohort of tests/test_gpu_loader.py::test_uneven_profiles_behave_as_on_the_host (index regions of `region_size`
bp, stretches without pairs, a profile that ends early, a sparse one) run through `popdel call` with the given arguments; a crash is
run again under gdb for a backtrace. usage: debug_uneven_repro.py [region_size=2000]"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from popdel_amd import synth  # noqa: E402

region_size = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
d = tempfile.mkdtemp()
region = (1000000, 1600000)
paths = []
for k in range(30):
    lib = synth.Library(hist_seed=300 + k, coverage=3.0 if k == 11 else 30.0)
    dels = [synth.Deletion(1200000, 1200, [1, 2, 0][k % 3]), synth.Deletion(1450000, 600, [2, 1, 1][k % 3])]
    pos, dev = synth.sample_pairs(7000 + k, region, lib, dels)
    if k % 4 == 1:
        keep = (pos < 1280000 + 3000 * k) | (pos >= 1420000)
        pos, dev = pos[keep], dev[keep]
    if k == 17:
        keep = pos < 1330000
        pos, dev = pos[keep], dev[keep]
    p = os.path.join(d, "u%02d.profile" % k)
    synth.write_profile(p, "U%02d" % k, pos, dev, lib, region_size=region_size)
    paths.append(p)
lst = os.path.join(d, "profiles.txt")
with open(lst, "w") as fh:
    fh.write("\n".join(paths) + "\n")
binary = os.path.join(ROOT, "popdel_amd", "lib", os.environ.get("POPDEL_BINARY", "popdel"))
env = dict(os.environ, POPDEL_TIMING="1", POPDEL_DEVICE_GUNZIP_WAIT="1")
for args in (["-r", "chr20:1000000-1600000"], ["-r", "chr20:1000000-1600000", "-b", "40000", "-n"], ["-r", "chr20:1030000-1590000", "-b", "70000"]):
    cmd = [binary, "call", lst, "-o", os.path.join(d, "o.vcf")] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    print(" ".join(args), "->", r.returncode, "|", "\n".join(l for l in r.stderr.splitlines() if "filing:" in l or "rror" in l)[-600:])
    if r.returncode != 0:
        print(r.stderr[-5000:])
    if r.returncode < 0 and shutil.which("gdb"):
        g = subprocess.run(["gdb", "-batch", "-ex", "run", "-ex", "bt 25", "-ex", "info threads", "--args"] + cmd, env=env, capture_output=True, text=True)
        print(g.stdout[-6000:])
shutil.rmtree(d, ignore_errors=True)
