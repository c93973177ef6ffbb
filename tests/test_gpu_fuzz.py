"""GPU: seeded random cohorts -- library shape, coverage, cohort size, deletions, genotypes and calling options all drawn
at random -- HIP against the oracle through the C ABI. Each case is small enough for the oracle to finish in a second or
two; together they sweep corners the hand-made cases do not (short/long libraries, 100-bp reads, deep and shallow
coverage side by side, overlapping deletions, few iterations, somatic and window-wise modes)."""
import numpy as np
import pytest

from conftest import HIP_SNAP_LIB, ORACLE_LIB
from test_gpu_parity import INT_KEYS, assert_same, both

pytestmark = pytest.mark.gpu


def random_case(seed):
    from popdel_amd import capi, synth
    rng = np.random.default_rng(900 + seed)
    lib = synth.Library(median=int(rng.integers(320, 620)), stddev=float(rng.uniform(18, 75)),
                        read_len=int(rng.choice([100, 125, 150])), hist_seed=int(rng.integers(1, 10 ** 6)))
    n = int(rng.choice([3, 17, 64, 130, 257, 300]))
    W = int(rng.integers(500, 2200))
    first = 1000020 + 30 * int(rng.integers(0, 1000))
    region = (first - 800, first + 30 * W)
    n_del = int(rng.integers(1, 5))
    dels = []
    for _ in range(n_del):
        length = int(np.exp(rng.uniform(np.log(250), np.log(6000))))
        start = int(rng.integers(region[0] + 3000, max(region[0] + 3001, region[1] - length - 3000)))
        af = float(rng.beta(0.7, 1.5))
        dels.append((start, length, rng.binomial(2, af, size=n)))
    base = synth.conditioned_readgroup(lib, max_load=int(rng.choice([100, 100, 40])))
    parts, offsets, rgs = [], [0], []
    for k in range(n):
        cov = float(rng.choice([30.0, 30.0, 12.0, 55.0, 4.0]))
        sub = synth.Library(median=lib.median, stddev=lib.stddev, read_len=lib.read_len, coverage=cov)
        pos, dev = synth.sample_pairs(50000 * seed + k, region, sub,
                                      [synth.Deletion(s, l, int(g[k])) for s, l, g in dels])
        d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
        parts.append((d, s, a, b))
        offsets.append(offsets[-1] + len(d))
        rgs.append(base)
    sample_offsets = list(range(n + 1))
    if seed % 3 == 2:
        # same pairs, but every sample's pairs dealt to 1-3 read groups of the same library (stable: each read group stays
        # ordered by start window): exercises the scan read groups and the lane-per-read-group stage-2 kernels
        split, sample_offsets, offsets, rgs = [], [0], [0], []
        for d, s_, a, b in parts:
            R = int(rng.integers(1, 4))
            which = rng.integers(0, R, size=len(d))
            for r in range(R):
                sel = which == r
                split.append((d[sel], s_[sel], a[sel], b[sel]))
                offsets.append(offsets[-1] + int(sel.sum()))
                rgs.append(base)
            sample_offsets.append(len(rgs))
        parts = split
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32),
                       np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    opts = dict(iterations=int(rng.choice([15, 15, 3, 7])), window_wise_output=bool(rng.integers(0, 2)),
                somatic=bool(rng.integers(0, 4) == 0), min_sample_fraction=float(rng.choice([0.1, 0.6])),
                min_len=int(rng.choice([200, 120, 400])))
    return seg, rgs, sample_offsets, opts


def assert_same_up_to_knife_edges(gpu, cpu, allowed_diverged=0):
    """Everything the hand-made cases demand, with one documented exception (Numerics): a reference shift is
    int(sum(p * dev) / sum(p)), zeroed if it exceeds the library's standard deviation. For a read group with a single
    active pair the exact quotient IS the pair's deviation, and for equally weighted pairs whose deviations add up to a
    multiple of their number it is an integer too: whether the rounded quotient lands on the integer or one ulp below it
    -- and so what the truncation, and at |shift| = sigma the zeroing, give -- is decided by the last bit of the weights
    p, which the reference computes in x87 extended precision. The oracle shows the same sensitivity on its own when pairs
    with equal start window are handed over in another order (tests/test_oracle_golden.py::
    test_reference_algorithm_is_order_sensitive), and with the quotient snapped (test below) nothing differs at all.
    Calls touched by such a flip may differ in a few sample columns (phred values, never the implied genotype) and in
    LR/AF beyond 1e-6; everything else must be exact. A flip inside an EM that oscillates up to the iteration cap can send
    it down another path altogether (seed 155: two shifts on -24 with sigma 23.1, lengths wandering between 5041 and 5105
    for 15 iterations); such a call counts as diverged.
    Per cohort the invariants are asserted here together with a loose FLOOR on the magnitudes (every call: LR within 1e-3,
    AF within 2e-4; at most max(2, n / 10) calls touched; a touched call differs in at most a third of its columns) -- so
    that a regression which moves LR by per cents without flipping a genotype fails in whatever subset of the sweep is run
    (-k, --lf, xdist); the tight figures -- HOW MANY calls and columns may be touched over the sweep -- are budgets over the
    whole session (test_session_budgets, conftest.py). Returns (touched calls, differing columns, diverged calls, worst
    |dLR| relative to max(1, |LR|) and worst |dAF| over the touched calls)."""
    (gc, gq, gf), (oc, oq, of) = gpu, cpu
    assert np.array_equal(gf, of)
    assert np.array_equal(gq, oq)
    assert gc["n_calls"] == oc["n_calls"]
    for k in ("initial_length", "window_position"):
        assert np.array_equal(gc[k], oc[k]), k
    n = gc["n_calls"]
    if n == 0:
        return 0, 0, 0, 0.0, 0.0
    diverged = np.abs(gc["iterations"].astype(int) - oc["iterations"].astype(int)) > 1
    for k in ("deletion_length", "position", "end_position", "filter"):
        diverged |= gc[k] != oc[k]
    diverged |= (gc["first_last"] != oc["first_last"]).reshape(n, -1).any(axis=1)
    assert diverged.sum() <= allowed_diverged, ("diverged calls", np.flatnonzero(diverged).tolist())
    keep = ~diverged
    cols = np.zeros(gc["pl"].shape[:2], bool)
    for k in ("pl", "lad", "dad"):
        cols |= (gc[k] != oc[k]).any(axis=2)
    cols[diverged] = False
    lr_ok = np.isclose(gc["likelihood_ratio"], oc["likelihood_ratio"], rtol=1e-6, atol=1e-6)
    af_ok = np.isclose(gc["frequency"], oc["frequency"], rtol=0, atol=1e-6)
    touched = (cols.any(axis=1) | ~lr_ok | ~af_ok | (gc["iterations"] != oc["iterations"])) & keep
    # the floor (per cohort, whatever else ran in this session)
    assert np.allclose(gc["likelihood_ratio"][keep], oc["likelihood_ratio"][keep], rtol=1e-3, atol=1e-3), "LR beyond 1e-3"
    assert np.allclose(gc["frequency"][keep], oc["frequency"][keep], rtol=0, atol=2e-4), "AF beyond 2e-4"
    assert touched.sum() <= max(2, n // 10), ("touched calls", int(touched.sum()), n)
    assert (cols.sum(axis=1) <= max(3, cols.shape[1] // 3)).all(), "a touched call differs in more than a third of its columns"
    worst_lr = worst_af = 0.0
    if touched.any():
        # never a changed implied genotype, never a zero phred value that moves
        assert np.array_equal(np.argmin(gc["pl"], axis=2)[keep], np.argmin(oc["pl"], axis=2)[keep])
        assert np.array_equal((gc["pl"] == 0)[keep], (oc["pl"] == 0)[keep])
        d_lr = np.abs(gc["likelihood_ratio"] - oc["likelihood_ratio"]) / np.maximum(1.0, np.abs(oc["likelihood_ratio"]))
        worst_lr = float(d_lr[touched].max())
        worst_af = float(np.abs(gc["frequency"] - oc["frequency"])[touched].max())
        LAST_TOUCHED_LR[:] = [float(x) for x in d_lr[touched]]
    else:
        LAST_TOUCHED_LR[:] = []
    return int(touched.sum()), int(cols.sum()), int(diverged.sum()), worst_lr, worst_af


def note_cohort(seed, figures):
    """One file per unsnapped cohort under the session's note directory (conftest.FUZZ_NOTES): whoever runs the cohorts --
    one process, xdist workers, a rerun -- the budgets are checked over the files (conftest.fuzz_session_figures)."""
    import json
    import os
    from conftest import FUZZ_NOTES
    os.makedirs(FUZZ_NOTES, exist_ok=True)
    with open(os.path.join(FUZZ_NOTES, "seed_%d.json" % seed), "w") as fh:
        json.dump(figures, fh)


# (side channel of assert_same_up_to_knife_edges: the relative LR differences of the touched calls of the cohort it looked at last)
LAST_TOUCHED_LR = []


# 64 cohorts as the reference computes them -- seeds 0-63, taken as they come: an UNBIASED sample of the knife-edge class
# (Numerics), on which the rates of conftest.py are checked. The cohorts that the builder's wider sweeps
# (1,050 cohorts, ~190,000 calls; profiles/r04_fuzz_sweeps.md) singled out for the largest footprints are NOT mixed into that
# sample (they would be a selection); they run below, against the shipped library, with their residual pinned call by call.
UNSNAPPED_SEEDS = list(range(64))


@pytest.mark.parametrize("seed", UNSNAPPED_SEEDS)
def test_random_cohort(built, seed):
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, **opts)
    calls, cols, div, worst_lr, worst_af = assert_same_up_to_knife_edges(gpu, cpu, allowed_diverged=0)
    n = gpu[0]["n_calls"]
    note_cohort(seed, dict(calls=n, columns=(n * gpu[0]["pl"].shape[1] if n else 0), differing_columns=cols, touched_calls=calls,
                           diverged_calls=div, touched_lr=list(LAST_TOUCHED_LR), worst_af=worst_af))
    if calls or div:
        print("seed %d: %d of %d calls touched by a knife-edge shift, %d sample column(s) differ, %d diverged; worst dLR %.3g, "
              "dAF %.3g" % (seed, calls, n, cols, div, worst_lr, worst_af))


# The worst cohorts of every sweep so far, against the SHIPPED library: 155 and 887 hold the two diverged calls of the first
# ~190,000 (length / position differ after an EM that oscillates up to the iteration cap); 533 and 1003 a call with more than a
# third of its sample columns touched (same length, position, genotypes); 632, 483, 620 the largest LR differences on calls
# that are otherwise the same (3.4e-4, 1.3e-4, 3.2e-5 relative); 693 and 515 the largest AF differences of rounds 2-4 (7.0e-5,
# 5.5e-5); 132 a long oscillating EM; from round 5's sweep of seeds 1050-1349 (profiles/r05_fuzz_sweeps.md): 1145 the largest AF
# difference so far (9.8e-5), 1112 / 1328 / 1070 calls with 70 / 59 / 56 % of their columns touched (1070 also LR by 2.6e-6);
# from the sweep of seeds 1350-1649: 1435 the third diverged call (length 5093 / 5097 after 15 / 13 iterations), 1424 twelve
# touched calls (LR by 8.9e-6); from the sweeps of seeds 1650-3149 (1,500 cohorts): 1997, 2741, 2875, 3025 one diverged call each and
# 2954 four (neighbouring candidate lengths of one deletion); 2093 the largest LR difference seen (3.0e-2: same length and
# position, the EM stops one iteration apart), 1916 the next of those sweeps (1.5e-5); 1829 the largest AF difference (1.03e-4);
# 1658, 1760, 2519, 2584, 2699, 2945 calls with 54-69 % of their columns touched; from round 6's sweep of seeds 3150-3749 on the
# final build (profiles/r06_fuzz_sweeps.md): 3392 a diverged call, 3665 a call with 74 % of its columns touched (the largest share so
# far), 3367 LR by 1.7e-5, 3666 AF by 7.5e-5. What the shipped library does on them is pinned, call by call, in tests/golden/residual_<seed>.json
# (tools/make_residual_goldens.py; tests/residual.py says what a record holds and which invariants hold regardless).
RESIDUAL_SEEDS = [132, 155, 483, 515, 533, 620, 632, 693, 887, 1003, 1070, 1112, 1145, 1328, 1424, 1435,
                  1658, 1760, 1829, 1916, 1997, 2093, 2519, 2584, 2699, 2741, 2875, 2945, 2954, 3025, 3367, 3392, 3665, 3666]


@pytest.mark.parametrize("seed", RESIDUAL_SEEDS)
def test_pinned_residual(built, seed):
    """The shipped library's known parity residual, asserted EXACTLY: which calls differ from the oracle's, whether they
    diverged, in how many sample columns, by how much in LR and AF. A change that makes the residual larger fails here; one
    that makes it smaller fails too and asks for the expectation to be re-generated and read. Every difference must coincide
    with a counted near-integer shift quotient in that call's own EM (asserted inside residual_record)."""
    import json
    import os
    from conftest import GOLDEN
    from residual import compare_records, residual_record, summary
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, **opts)
    got = residual_record(seed, gpu, cpu)
    with open(os.path.join(GOLDEN, "residual_%d.json" % seed)) as fh:
        want = json.load(fh)
    problems = compare_records(got, want)
    assert not problems, problems
    from conftest import FUZZ_NOTES
    os.makedirs(FUZZ_NOTES, exist_ok=True)
    with open(os.path.join(FUZZ_NOTES, "parity_notes.txt"), "a") as fh:
        fh.write("pinned residual, seed %(seed)d: %(calls)d calls, %(touched_calls)d touched, %(diverged_calls)d diverged, %(differing_columns)d "
                 "columns, worst dLR %(worst_dlr).3g, dAF %(worst_daf).3g, largest share of a call's columns %(worst_column_share).2f\n"
                 % summary(got))


SNAPPED_SEEDS = list(range(16, 64)) + RESIDUAL_SEEDS       # every pinned-residual cohort is EXACT once the knife edge is out


@pytest.mark.parametrize("seed", SNAPPED_SEEDS)
def test_random_cohort_exact_with_snapped_shifts(built, seed, monkeypatch):
    """The same sweep with the knife edge taken out on both sides: a reference-shift quotient within 1e-9 of an integer
    counts as that integer -- in the oracle through its POPDEL_SNAP_SHIFTS knob, on the GPU through a TEST build of the
    library (oracle/_build/libpopdel_hip_snap.so: stage2.hip compiled with PD_TEST_SNAP_SHIFTS; the shipped library has
    no such switch). What remains is every other step of the path -- scan, clustering, table lookups, the EM sums,
    convergence, support percentiles, phred rounding -- and there nothing may differ: all integer fields bit-exact, LR and
    AF within 1e-6. Seeds 132 and 155 are the cases in which an unsnapped flip cascades furthest (a long oscillating EM;
    two shifts sitting on -24 with sigma 23.1)."""
    monkeypatch.setenv("POPDEL_SNAP_SHIFTS", "1")
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, hip_library=HIP_SNAP_LIB, **opts)
    assert_same(gpu, cpu)


def test_the_sweep_is_not_vacuous(built):
    """The random cases together must produce a healthy number of calls (checked on the oracle side only)."""
    from popdel_amd import capi
    total = 0
    for seed in range(0, 16, 3):
        seg, rgs, offs, opts = random_case(seed)
        ctx = capi.Context(capi.load_library(ORACLE_LIB), rgs, offs, **opts)
        total += ctx.collect(ctx.submit(seg))["n_calls"]
        ctx.close()
    assert total > 150


# (budgets of the knife-edge class over the whole sweep: conftest.py)


def test_session_budgets(built):
    """The budgets over the cohorts THIS session has noted so far (one process: all 64, this test runs last in the file). Under
    xdist or a partial selection fewer may be there: then the check is left to conftest.pytest_sessionfinish, which sees the
    notes of all workers and fails the session the same way (and never skips once the whole sweep has run)."""
    from conftest import check_fuzz_budgets, fuzz_session_figures
    s = fuzz_session_figures()
    if s["cohorts"] < 60:
        pytest.skip("only %d unsnapped cohorts noted so far (the budgets are stated for the whole sweep; conftest checks them at the "
                    "end of the session)" % s["cohorts"])
    problems = check_fuzz_budgets(s)
    assert not problems, problems
