"""Wider sweep (GPU box): python tools/debug_fuzz_sweep.py <first seed> <last seed>
Demands what tests/test_gpu_fuzz.py demands (every integer field bit-exact, LR/AF within 1e-6) on more seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_fuzz import random_case
from test_gpu_parity import assert_same, both
fails = calls = cols = diff_cols = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, **opts)
    (gc, _, _), (oc, _, _) = gpu, cpu
    n = gc["n_calls"]
    calls += n
    cols += n * gc["n_samples"]
    try:
        assert_same(gpu, cpu)
        print("seed", seed, "calls", n, "exact", flush=True)
    except AssertionError as e:
        fails += 1
        d = 0
        if n == oc["n_calls"] and n:
            c = np.zeros(gc["pl"].shape[:2], bool)
            for k in ("pl", "lad", "dad"):
                c |= (gc[k] != oc[k]).any(axis=2)
            d = int(c.sum())
        diff_cols += d
        print("seed", seed, "calls", n, "FAILED: %d sample columns differ;" % d, str(e)[:200].replace("\n", " "), flush=True)
print("total: %d seeds failed, %d calls, %d sample columns, %d differing columns" % (fails, calls, cols, diff_cols))
