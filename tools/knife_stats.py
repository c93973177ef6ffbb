"""CPU only: run the oracle built with -DPD_KNIFE_STATS over fuzz cohorts and print its classification of the reference-shift
quotients that sit on an integer (see oracle/genotype_oracle.cpp, knifeStats). Usage: knife_stats.py <first> <last>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import random_case
from popdel_amd import capi
lib = capi.load_library(os.path.join(ROOT, "oracle", "_build", "libpopdel_oracle_stats.so"))
calls = cols = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    seg, rgs, offs, opts = random_case(seed)
    ctx = capi.Context(lib, rgs, offs, **opts)
    r = ctx.collect(ctx.submit(seg))
    calls += r["n_calls"]; cols += r["n_calls"] * (len(offs) - 1)
    ctx.close()
print("calls", calls, "columns", cols)
