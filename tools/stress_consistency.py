"""Self-consistency stress on the GPU box: random cohorts (size, read groups per sample, coverage, deletions), each run
(a) in one stage-2 batch, (b) in many batches, (c) again -- all three must agree bit for bit. No oracle involved, so the
cohorts can be larger than in the parity tests. usage: python tools/stress_consistency.py [first seed] [last seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from popdel_amd import capi, synth


from stress_cases import case


def run(seg, rgs, offs, opts, budget=None):
    if budget:
        os.environ["POPDEL_HIP_STAGE2_BYTES"] = str(budget)
    try:
        ctx = capi.Context(capi.load_hip(), rgs, offs, **opts)
        h = ctx.stage(seg)
        out = [ctx.collect(ctx.run(h)) for _ in range(2)]
        ctx.unstage(h); ctx.close()
    finally:
        os.environ.pop("POPDEL_HIP_STAGE2_BYTES", None)
    return out


bad = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    t = time.time()
    seg, rgs, offs, opts = case(seed)
    a1, a2 = run(seg, rgs, offs, opts)
    b1, _ = run(seg, rgs, offs, opts, budget=30_000_000)
    ok = True
    for k, v in a1.items():
        ok = ok and np.array_equal(np.asarray(a2[k]), np.asarray(v)) and np.array_equal(np.asarray(b1[k]), np.asarray(v))
    bad += not ok
    print("seed", seed, "samples", len(offs) - 1, "read groups", len(rgs), "windows", seg.n_windows, "calls", a1["n_calls"],
          "OK" if ok else "MISMATCH", "%.1f s" % (time.time() - t), flush=True)
print("mismatching cases:", bad)
