"""Wider sweep (GPU box): python tools/debug_fuzz_sweep.py <first seed> <last seed> [snap]
"snap" takes the knife edge out on both sides (oracle knob + test build of the library) and demands exact equality (tests/test_gpu_fuzz.py explains the knob)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import random_case, assert_same_up_to_knife_edges
from test_gpu_parity import assert_same, both
from conftest import HIP_SNAP_LIB
snap = len(sys.argv) > 3 and sys.argv[3] == "snap"
if snap:
    os.environ["POPDEL_SNAP_SHIFTS"] = "1"
tot_bad = tot_cols = fails = tot_div = tot_touched = tot_calls = 0
worst_lr = worst_af = 0.0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, hip_library=HIP_SNAP_LIB if snap else None, **opts)
    try:
        if snap:
            assert_same(gpu, cpu)
            _calls, nb = 0, 0
        else:
            _calls, nb, dv, wlr, waf = assert_same_up_to_knife_edges(gpu, cpu, allowed_diverged=1)
            tot_div += dv
            tot_touched += _calls
            worst_lr, worst_af = max(worst_lr, wlr), max(worst_af, waf)
            if dv:
                print("seed", seed, "has a diverged call")
    except AssertionError as e:
        fails += 1
        print("seed", seed, "FAILED", str(e)[:300])
        continue
    cols = gpu[0]["pl"].shape[0] * gpu[0]["pl"].shape[1] if gpu[0]["n_calls"] else 0
    tot_bad += nb; tot_cols += cols; tot_calls += gpu[0]["n_calls"]
    print("seed", seed, "calls", gpu[0]["n_calls"], "columns", cols, "knife-edge", nb,
          ("touched %d worst dLR %.3g dAF %.3g" % (_calls, wlr, waf)) if not snap and _calls else "", flush=True)
print("total knife-edge", tot_bad, "of", tot_cols, "columns; diverged calls", tot_div, "; failures", fails)
print("touched calls", tot_touched, "of", tot_calls, "; worst |dLR| (relative to max(1, |LR|))", worst_lr, "; worst |dAF|", worst_af)
