#!/usr/bin/env python3
"""Debugging: the two-contig cohort of tests/test_gpu_loader.py::test_data_on_two_contigs through `popdel call` (device loader) and the
host-only configuration; prints what fails. usage: debug_two_contigs.py [samples=40] [popdel call options ...]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_loader import two_contig_cohort  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
args = sys.argv[2:]
d = tempfile.mkdtemp()
two_contig_cohort(d, n)
binary = os.path.join(ROOT, "popdel_amd", "lib", os.environ.get("POPDEL_BINARY", "popdel"))
outs = []
def fields_that_differ(a, b):
    a, b = a.splitlines(), b.splitlines()
    for i in range(min(len(a), len(b))):
        if a[i] != b[i]:
            fa, fb = a[i].split("\t"), b[i].split("\t")
            diff = [j for j in range(min(len(fa), len(fb))) if fa[j] != fb[j]]
            return "line %d: %d of %d fields differ, first: %s" % (i, len(diff), len(fa), [(j, fa[j][:90], fb[j][:90]) for j in diff[:3]])
    return "lengths %d %d" % (len(a), len(b))


oracle = os.path.join(ROOT, "oracle", "_build", "popdel_oracle_call")
subprocess.check_call([oracle, os.path.join(d, "profiles.txt"), "-o", os.path.join(d, "cpu.vcf")] + args, stdout=subprocess.DEVNULL)
cpu = "".join(l for l in open(os.path.join(d, "cpu.vcf")) if not l.startswith("##fileDate"))
for env in ({}, {"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"}, {"POPDEL_HOST_WALK": "1"}, {"POPDEL_HOST_FILING": "1"}):
    e = dict(os.environ, POPDEL_TIMING="1", POPDEL_DEVICE_GUNZIP_WAIT="1")
    e.update(env)
    out = os.path.join(d, "o.vcf")
    r = subprocess.run([binary, "call", os.path.join(d, "profiles.txt"), "-o", out] + args, env=e, capture_output=True, text=True)
    body = "".join(l for l in open(out) if not l.startswith("##fileDate")) if r.returncode == 0 else None
    outs.append(body)
    print(env, "->", r.returncode, "same as first" if body is not None and body == outs[0] else "", "| same as the CPU driver" if body == cpu else "| CPU driver: " + (fields_that_differ(cpu, body) if body else ""))
    print("   ", "\n    ".join(l[:300] for l in r.stderr.splitlines() if "window loop" in l or "region change" in l or "timing" not in l))
    if body and body != cpu:
        a, b = cpu.splitlines(), body.splitlines()
        bad = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
        print("    %d lines differ of %d / %d; positions %s" % (len(bad), len(a), len(b), [a[i].split("\t")[:2] for i in bad[:40:3]]))
        cols = set()
        for i in bad:
            fa, fb = a[i].split("\t"), b[i].split("\t")
            cols.update(j for j in range(min(len(fa), len(fb))) if fa[j] != fb[j])
        print("    columns that differ anywhere:", sorted(cols)[:20])
        i = bad[len(bad) // 2]
        print("    e.g.", a[i].split("\t")[1], [x.split("\t")[14][:70] for x in (a[i], b[i])])
