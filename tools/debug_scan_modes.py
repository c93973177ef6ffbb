"""Debugging aid (GPU box): python tools/debug_scan_modes.py [coverage] -- the deep-coverage cohort of tests/test_gpu_screen.py through
the full scan twice, the screen twice and the screen without the view; says which result arrays differ between which runs."""
import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from popdel_amd import capi, synth
cov = float(sys.argv[1]) if len(sys.argv) > 1 else 170.0
lib = synth.Library(coverage=cov)
rg = synth.conditioned_readgroup(lib, max_load=1000)
n = 70
seg = synth.cohort_segment(n, (1000000, 1040000), 1000020, 1250, lib,
                           deletions=lambda k, r: [synth.Deletion(1015000, 700, [1, 2, 0][k % 3]), synth.Deletion(1030000, 2600, [0, 1, 0, 0][k % 4])])
res = {}
for mode in ("full", "full2", "screen", "screen2", "screen_noview"):
    if mode.startswith("full"):
        os.environ["POPDEL_HIP_SCAN"] = "full"
    else:
        os.environ.pop("POPDEL_HIP_SCAN", None)
    ctx = capi.Context(capi.load_hip(), [rg] * n, list(range(n + 1)), stage1_view=(mode != "screen_noview"))
    res[mode] = ctx.collect(ctx.submit(seg))
    ctx.close()
keys = [k for k in res["full"] if isinstance(res["full"][k], np.ndarray)]
for a, b in (("full", "full2"), ("screen", "screen2"), ("full", "screen"), ("screen", "screen_noview")):
    diff = [k for k in keys if not np.array_equal(res[a][k], res[b][k])]
    print(a, b, res[a]["n_calls"], res[b]["n_calls"], "differ:", diff)
    for k in diff[:3]:
        x, y = res[a][k], res[b][k]
        idx = np.argwhere(x != y)[:5]
        print("   ", k, [(tuple(i), x[tuple(i)], y[tuple(i)]) for i in idx])
