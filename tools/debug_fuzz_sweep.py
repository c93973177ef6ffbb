"""Wider sweep (GPU box): python tools/debug_fuzz_sweep.py <first seed> <last seed> [snap]

Every cohort of tests/test_gpu_fuzz.py::random_case in [first, last) through the shipped HIP library and the oracle; one JSON
line per cohort (tests/residual.py::summary: calls, touched, diverged, differing columns, worst dLR / dAF, the largest share of
one call's columns) and the totals. No cohort is dropped for being bad: the invariants of residual_record() (stage 1 and call
list exact, genotypes of undiverged calls unchanged, every difference on a counted near-integer quotient) are the only way to
FAIL, and such a cohort is printed with its message.
"snap" takes the knife edge out on both sides (oracle knob + test build of the library) and demands exact equality."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import HIP_SNAP_LIB                      # noqa: E402
from residual import residual_record, summary          # noqa: E402
from test_gpu_fuzz import random_case                  # noqa: E402
from test_gpu_parity import assert_same, both          # noqa: E402

snap = len(sys.argv) > 3 and sys.argv[3] == "snap"
if snap:
    os.environ["POPDEL_SNAP_SHIFTS"] = "1"
tot = dict(cohorts=0, calls=0, columns=0, touched_calls=0, diverged_calls=0, differing_columns=0, lr_beyond_1e6=0, failures=0)
worst = dict(worst_dlr=0.0, worst_daf=0.0, worst_column_share=0.0)
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, hip_library=HIP_SNAP_LIB if snap else None, **opts)
    try:
        if snap:
            assert_same(gpu, cpu)
            print(json.dumps(dict(seed=seed, calls=int(gpu[0]["n_calls"]), exact=True)), flush=True)
            tot["cohorts"] += 1
            tot["calls"] += int(gpu[0]["n_calls"])
            continue
        rec = residual_record(seed, gpu, cpu)
    except AssertionError as e:
        tot["failures"] += 1
        print(json.dumps(dict(seed=seed, FAILED=str(e)[:300])), flush=True)
        continue
    s = summary(rec)
    s["lr_beyond_1e6"] = sum(1 for t in rec["touched"] if not t["diverged"] and t["dlr_rel"] > 1e-6)
    print(json.dumps(s), flush=True)
    tot["cohorts"] += 1
    for k in ("calls", "columns", "touched_calls", "diverged_calls", "differing_columns", "lr_beyond_1e6"):
        tot[k] += s[k]
    for k in worst:
        worst[k] = max(worst[k], s[k])
tot.update(worst)
print("TOTAL", json.dumps(tot))
