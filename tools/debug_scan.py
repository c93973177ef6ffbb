"""GPU box: Q3 of the scan against the oracle on a small cohort, listing the cells that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popdel_amd import capi, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import ORACLE_LIB
lib = synth.Library()
rg = synth.conditioned_readgroup(lib)
n, W = 8, 1900
seg = synth.cohort_segment(n, (1000000, 1060000), 1000020, W, lib)
out = []
for l in (capi.load_hip(), capi.load_library(ORACLE_LIB)):
    ctx = capi.Context(l, [rg] * n, list(range(n + 1)))
    ctx.collect(ctx.submit(seg))
    out.append(ctx.stage1())
    ctx.close()
(gq, gf), (oq, of) = out
bad = np.argwhere(gq != oq)
print("flags equal", np.array_equal(gf, of), "q3 mismatches", len(bad), "of", gq.size)
for w, s in bad[:40]:
    lo, hi = int(seg.rg_offsets[s]), int(seg.rg_offsets[s + 1])
    on, off = seg.on[lo:hi], seg.off[lo:hi]
    act = int(((on <= w) & (off > w)).sum())
    print("w", w, "(half %s)" % ("A" if w % 2 == 0 else "B"), "sample", s, "gpu", gq[w, s], "oracle", oq[w, s], "n", act)
