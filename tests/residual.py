"""The parity residual of the SHIPPED HIP library against the oracle, as data (test infrastructure).

One dependence of the reference cannot be reproduced in FP64 (Numerics; reference
genotype_deletion_popdel_call.h:447-450): a reference shift is int(sum(p * dev) / sum(p)) with weights that come out of x87
intermediates, and where the exact quotient is an integer the truncation hangs off their last bit. `residual_record()` turns a
(HIP, oracle) pair of results into the list of calls that differ in any way, with everything that differs; the expectation
files tests/golden/residual_<seed>.json hold such records for the cohorts in which the builder's sweeps found the largest
footprints, and tests/test_gpu_fuzz.py::test_pinned_residual asserts them exactly -- so a kernel change that moves the
residual in EITHER direction shows. tools/make_residual_goldens.py writes the files (on a GPU box).

What no record may contain is asserted here, for every cohort: stage 1 (Q3, flags), the call list itself (number of calls,
window and initial length of each) are bit-exact; a call that did not diverge keeps every implied genotype and every zero
phred value; and every call that differs anywhere has a non-zero count of near-integer shift quotients in its own EM, on both
sides (pd_calls::near_integer_shifts) -- i.e. every difference coincides with a counted knife-edge event."""
import numpy as np

SCALARS = ("iterations", "deletion_length", "position", "end_position", "filter")


def residual_record(seed, gpu, cpu):
    (gc, gq, gf), (oc, oq, of) = gpu, cpu
    assert np.array_equal(gf, of), "stage-1 flags differ"
    assert np.array_equal(gq, oq), "stage-1 Q3 differs"
    assert gc["n_calls"] == oc["n_calls"], ("number of calls", gc["n_calls"], oc["n_calls"])
    n = int(gc["n_calls"])
    rec = dict(seed=int(seed), n_calls=n, n_samples=int(gc["n_samples"]), touched=[])
    if n == 0:
        return rec
    for k in ("initial_length", "window_position"):
        assert np.array_equal(gc[k], oc[k]), k
    diverged = np.abs(gc["iterations"].astype(int) - oc["iterations"].astype(int)) > 1
    for k in ("deletion_length", "position", "end_position", "filter"):
        diverged |= gc[k] != oc[k]
    diverged |= (gc["first_last"] != oc["first_last"]).reshape(n, -1).any(axis=1)
    cols = np.zeros(gc["pl"].shape[:2], bool)
    for k in ("pl", "lad", "dad"):
        cols |= (gc[k] != oc[k]).any(axis=2)
    fl_cols = (gc["first_last"] != oc["first_last"]).any(axis=2)
    lr_ok = np.isclose(gc["likelihood_ratio"], oc["likelihood_ratio"], rtol=1e-6, atol=1e-6)
    af_ok = np.isclose(gc["frequency"], oc["frequency"], rtol=0, atol=1e-6)
    touched = cols.any(axis=1) | ~lr_ok | ~af_ok | (gc["iterations"] != oc["iterations"]) | diverged
    keep = ~diverged
    # invariants of the class
    assert np.array_equal(np.argmin(gc["pl"], axis=2)[keep], np.argmin(oc["pl"], axis=2)[keep]), "an implied genotype changed"
    assert np.array_equal((gc["pl"] == 0)[keep], (oc["pl"] == 0)[keep]), "a zero phred value moved"
    assert (gc["near_integer_shifts"][touched] > 0).all(), "a call differs without a counted near-integer shift quotient (HIP count)"
    assert (oc["near_integer_shifts"][touched] > 0).all(), "a call differs without a counted near-integer shift quotient (oracle count)"
    d_lr = np.abs(gc["likelihood_ratio"] - oc["likelihood_ratio"]) / np.maximum(1.0, np.abs(oc["likelihood_ratio"]))
    d_af = np.abs(gc["frequency"] - oc["frequency"])
    for i in np.flatnonzero(touched):
        t = dict(call=int(i), window_position=int(gc["window_position"][i]), initial_length=int(gc["initial_length"][i]),
                 diverged=bool(diverged[i]), columns=int(cols[i].sum()), first_last_columns=int(fl_cols[i].sum()),
                 genotypes_changed=int((np.argmin(gc["pl"][i], axis=1) != np.argmin(oc["pl"][i], axis=1)).sum()),
                 dlr_rel=float(d_lr[i]), daf=float(d_af[i]))
        for k in SCALARS:
            t[k] = [int(gc[k][i]), int(oc[k][i])]             # [HIP, oracle]
        rec["touched"].append(t)
    return rec


def compare_records(got, want, float_rtol=1e-6):
    """List of differences between two records (empty: the residual is the pinned one). Integers exactly; the two float
    magnitudes to a relative 1e-6 of themselves (both sides are deterministic; the slack is for the JSON round trip and for
    libm builds whose last bit differs on the oracle's side)."""
    bad = []
    for k in ("n_calls", "n_samples"):
        if got[k] != want[k]:
            bad.append("%s: %s, pinned %s" % (k, got[k], want[k]))
    g = {t["call"]: t for t in got["touched"]}
    w = {t["call"]: t for t in want["touched"]}
    for c in sorted(set(g) - set(w)):
        bad.append("call %d differs from the oracle's and is not in the pinned residual: %s" % (c, g[c]))
    for c in sorted(set(w) - set(g)):
        bad.append("call %d of the pinned residual no longer differs (re-generate the expectation if that is intended): %s" % (c, w[c]))
    for c in sorted(set(g) & set(w)):
        for k, v in w[c].items():
            if isinstance(v, float):
                if abs(g[c][k] - v) > float_rtol * max(abs(v), 1e-12):
                    bad.append("call %d %s: %.6g, pinned %.6g" % (c, k, g[c][k], v))
            elif g[c][k] != v:
                bad.append("call %d %s: %s, pinned %s" % (c, k, g[c][k], v))
    return bad


def summary(rec):
    t = rec["touched"]
    return dict(seed=rec["seed"], calls=rec["n_calls"], columns=rec["n_calls"] * rec["n_samples"], touched_calls=sum(not x["diverged"] for x in t),
                diverged_calls=sum(x["diverged"] for x in t), differing_columns=sum(x["columns"] for x in t if not x["diverged"]),
                worst_dlr=max([x["dlr_rel"] for x in t if not x["diverged"]] or [0.0]),
                worst_daf=max([x["daf"] for x in t if not x["diverged"]] or [0.0]),
                worst_column_share=max([x["columns"] / rec["n_samples"] for x in t if not x["diverged"]] or [0.0]))
