"""CPU: python tools/oracle_order_probe.py <seed> [n_perms] -- runs the oracle on a fuzz case with the pairs of equal start
window handed over in several orders and lists the calls whose integer outputs change (order-unstable cells)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_LIB
from test_gpu_fuzz import random_case
from popdel_amd import capi
seed = int(sys.argv[1]); nperm = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seg, rgs, offs, opts = random_case(seed)
lib = capi.load_library(ORACLE_LIB)
ctx = capi.Context(lib, rgs, offs, **opts)
a = ctx.collect(ctx.submit(seg))
print("calls", a["n_calls"])
o = np.asarray(seg.rg_offsets, dtype=np.int64)
for p in range(nperm):
    rng = np.random.default_rng(100 + p)
    perm = np.arange(len(seg.dev))
    for r in range(len(o) - 1):
        lo, hi = o[r], o[r + 1]
        key = seg.on[lo:hi].astype(np.int64) * 1000003 + rng.integers(0, 1000003, hi - lo)
        perm[lo:hi] = lo + np.argsort(key, kind="stable")
    seg2 = capi.Segment(seg.first_window_pos, seg.n_windows, seg.rg_offsets, seg.dev[perm], seg.start[perm], seg.on[perm], seg.off[perm])
    b = ctx.collect(ctx.submit(seg2))
    if b["n_calls"] != a["n_calls"]:
        print("perm", p, "call count changes", a["n_calls"], b["n_calls"]); continue
    for k in ("iterations", "deletion_length", "position", "end_position"):
        idx = np.flatnonzero(np.asarray(a[k]) != np.asarray(b[k]))
        for c in idx:
            print("perm", p, k, "call", int(c), "win", int(a["window_position"][c]), int(a[k][c]), "->", int(b[k][c]))
    print("perm", p, "pl columns changed", int((a["pl"] != b["pl"]).any(axis=2).sum()), "LR max diff",
          float(np.abs(a["likelihood_ratio"] - b["likelihood_ratio"]).max()))
ctx.close()
