"""Debug helper (GPU box): python tools/debug_fuzz.py <seed> -- prints where HIP and oracle differ for a fuzz case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_LIB
from test_gpu_fuzz import random_case
from popdel_amd import capi
seed = int(sys.argv[1])
seg, rgs, offs, opts = random_case(seed)
print("opts", opts, "n", len(offs) - 1, "W", seg.n_windows, "rg0 median/sd/readlen", rgs[0].median, rgs[0].stddev, rgs[0].read_length, "maxload", rgs[0].max_load)
res = []
st = []
for lib in (capi.load_hip(), capi.load_library(ORACLE_LIB)):
    ctx = capi.Context(lib, rgs, offs, **opts)
    res.append(ctx.collect(ctx.submit(seg)))
    st.append(ctx.stage1())
g, o = res
(gq, gf), (oq, of) = st
print("q3 differs at", int((gq != oq).sum()), "flags differ at", int((gf != of).sum()))
if (gq != oq).any():
    idx = np.argwhere(gq != oq)
    for t in idx[:8]:
        print("   window", t[0], "sample", t[1], "gpu", gq[t[0], t[1]], "cpu", oq[t[0], t[1]], "flags", gf[t[0], t[1]], of[t[0], t[1]])
print("calls", g["n_calls"], o["n_calls"])
for k in g:
    if k in ("n_calls", "n_samples"): continue
    a, b = np.asarray(g[k]), np.asarray(o[k])
    if a.shape != b.shape:
        print(k, "shape", a.shape, b.shape)
        if k == "window_position":
            print("   gpu only", sorted(set(zip(g["window_position"].tolist(), g["initial_length"].tolist())) - set(zip(o["window_position"].tolist(), o["initial_length"].tolist())))[:10])
            print("   cpu only", sorted(set(zip(o["window_position"].tolist(), o["initial_length"].tolist())) - set(zip(g["window_position"].tolist(), g["initial_length"].tolist())))[:10])
        continue
    if a.dtype.kind == "f":
        d = np.abs(a - b)
        print(k, "max abs diff", d.max() if d.size else 0)
    else:
        idx = np.argwhere(a != b)
        print(k, "differs at", len(idx))
        for t in idx[:12]:
            c, s = int(t[0]), (int(t[1]) if len(t) > 1 else -1)
            print("   call", c, "sample", s, "gpu", a[c, s] if s >= 0 else a[c], "cpu", b[c, s] if s >= 0 else b[c],
                  "| win", g["window_position"][c], "len", g["deletion_length"][c], "it", g["iterations"][c],
                  "lad", g["lad"][c, s] if s >= 0 else "", "dad", g["dad"][c, s] if s >= 0 else "")
