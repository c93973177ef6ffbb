"""GPU box: python tools/debug_fuzz.py <seed> -- HIP against the oracle on one fuzz cohort; lists what differs and, for the first
differing call, the EM traces of both sides (POPDEL_TRACE_POS: a debugging aid of library and oracle, results unchanged)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

def run(seed):
    from test_gpu_fuzz import random_case
    from test_gpu_parity import both
    seg, rgs, offs, opts = random_case(seed)
    return both(seg, rgs, offs, **opts), opts

if len(sys.argv) > 2 and sys.argv[2] == "trace":
    run(int(sys.argv[1]))
    sys.exit(0)
seed = int(sys.argv[1])
((gc, gq, gf), (oc, oq, of)), opts = run(seed)
print("seed", seed, "options", opts, "calls", gc["n_calls"], oc["n_calls"], "stage 1 equal", np.array_equal(gq, oq) and np.array_equal(gf, of))
if gc["n_calls"] != oc["n_calls"]:
    sys.exit("call counts differ")
first = None
for i in range(gc["n_calls"]):
    what = [k for k in ("initial_length", "iterations", "deletion_length", "window_position", "position", "end_position", "filter")
            if gc[k][i] != oc[k][i]]
    cols = [s for s in range(gc["n_samples"]) if any((gc[k][i, s] != oc[k][i, s]).any() for k in ("pl", "lad", "dad", "first_last"))]
    dlr = abs(gc["likelihood_ratio"][i] - oc["likelihood_ratio"][i])
    daf = abs(gc["frequency"][i] - oc["frequency"][i])
    if what or cols or dlr > 1e-6 * max(1, abs(oc["likelihood_ratio"][i])) or daf > 1e-6:
        print("call %d window_position %d init %d: fields %s, %d columns %s, |dLR| %.3g |dAF| %.3g, iterations %d/%d len %d/%d"
              % (i, oc["window_position"][i], oc["initial_length"][i], what, len(cols), cols[:8], dlr, daf, gc["iterations"][i],
                 oc["iterations"][i], gc["deletion_length"][i], oc["deletion_length"][i]))
        for s in cols[:3]:
            print("    sample %d: pl %s / %s lad %s / %s" % (s, gc["pl"][i, s], oc["pl"][i, s], gc["lad"][i, s], oc["lad"][i, s]))
        if first is None:
            first = int(oc["window_position"][i]) + 1
if first is not None:
    env = dict(os.environ, POPDEL_TRACE_POS=str(first))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(seed), "trace"], env=env, capture_output=True, text=True)
    print("---- traces of the window at", first)
    print("\n".join(l for l in out.stderr.splitlines() if "trace]" in l))
