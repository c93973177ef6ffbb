"""GPU: stage 1 as a screen (stage1.hip, round 5). The product path counts, per cell, the active pairs whose deviation exceeds the
smallest threshold and takes exact Q3 values only in the windows that may carry a candidate; the round-1..4 scan (exact Q3 of
every cell) stays as POPDEL_HIP_SCAN=full and as the diagnostic view. Here: both ways give the same calls bit for bit, both
agree with the oracle, and the view's own cell-by-cell check (pd_stage1_last: flags everywhere, every window with a Q3 above the
threshold flagged, the exact pass's Q3 equal to the full scan's in every flagged window) holds -- on ordinary and deep coverage
(the 128-input networks and rank counting of the exact pass), deviations beyond 16 bits and a multi-read-group sample at its
maxLoad (blocks the screen hands to the fallback kernel), and hand-made cells whose Q3 sits exactly on the threshold."""
import numpy as np
import pytest

from conftest import ORACLE_LIB
from test_gpu_parity import INT_KEYS

pytestmark = pytest.mark.gpu


def three_ways(seg, rgs, offs, monkeypatch, **kw):
    """(screen, full scan, oracle): calls and the stage-1 view of each."""
    from popdel_amd import capi
    out = []
    for mode in ("screen", "full", "oracle"):
        if mode == "full":
            monkeypatch.setenv("POPDEL_HIP_SCAN", "full")
        else:
            monkeypatch.delenv("POPDEL_HIP_SCAN", raising=False)
        lib = capi.load_library(ORACLE_LIB) if mode == "oracle" else capi.load_hip()
        ctx = capi.Context(lib, rgs, offs, stage1_view=True, **kw)
        calls = ctx.collect(ctx.submit(seg))
        q3, fl = ctx.stage1()
        out.append((calls, q3, fl))
        ctx.close()
    return out


def assert_same(gpu, cpu):
    """Against the oracle: exact up to the knife-edge class of DESIGN.md 4 (tests/residual.py asserts the invariants: stage 1
    and the call list bit-exact, genotypes unchanged, every difference on a counted near-integer quotient), and few."""
    from residual import residual_record
    rec = residual_record(0, gpu, cpu)
    assert len(rec["touched"]) <= max(2, rec["n_calls"] // 10), rec["touched"]
    assert not any(t["diverged"] for t in rec["touched"]), rec["touched"]


def assert_identical(a, b):
    (ac, aq, af), (bc, bq, bf) = a, b
    assert np.array_equal(aq, bq) and np.array_equal(af, bf)
    assert ac["n_calls"] == bc["n_calls"]
    for k in INT_KEYS + ("likelihood_ratio", "frequency", "near_integer_shifts"):
        assert np.array_equal(ac[k], bc[k]), k          # (doubles too: the same kernels ran on the same Q3 values)


@pytest.mark.parametrize("coverage,min_calls", [(30.0, 20), (75.0, 20), (170.0, 20)])
def test_screen_full_scan_and_oracle_agree(built, monkeypatch, coverage, min_calls):
    """30x: up to 64 candidates per cell (the exact pass's first kernel); 75x: 65-128 (its second kernel's networks);
    170x: beyond 128 in many cells (rank counting)."""
    from popdel_amd import synth
    lib = synth.Library(coverage=coverage)
    rg = synth.conditioned_readgroup(lib, max_load=1000)
    n = 70
    seg = synth.cohort_segment(n, (1000000, 1040000), 1000020, 1250, lib,
                               deletions=lambda k, r: [synth.Deletion(1015000, 700, [1, 2, 0][k % 3]), synth.Deletion(1030000, 2600, [0, 1, 0, 0][k % 4])])
    screen, full, oracle = three_ways(seg, [rg] * n, list(range(n + 1)), monkeypatch)
    assert oracle[0]["n_calls"] >= min_calls
    assert_identical(screen, full)
    assert_same(screen, oracle)


def test_deviations_beyond_16_bits_go_to_the_fallback_kernel(built, monkeypatch):
    """A 40-kb deletion: its supporting pairs carry deviations above 32,767, the read group has no packed image and the screen
    hands every block of it to the 32-bit fallback kernel."""
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 12
    seg = synth.cohort_segment(n, (1000000, 1100000), 1000020, 3200, lib,
                               deletions=lambda k, r: [synth.Deletion(1020000, 40000, [1, 2, 0, 1][k % 4])])
    assert int(seg.dev.max()) > 32767
    screen, full, oracle = three_ways(seg, [rg] * n, list(range(n + 1)), monkeypatch)
    assert oracle[0]["n_calls"] > 5
    assert_identical(screen, full)
    assert_same(screen, oracle)


def test_multi_read_group_sample_with_a_read_group_at_its_max_load(built, monkeypatch):
    """Samples of three read groups of which one reaches its maxLoad in most windows (the reference then drops that read
    group's pairs): the merged scan stream cannot tell, the blocks are flagged for the fallback kernel."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    rg_tight = synth.conditioned_readgroup(lib, max_load=9)
    region, first, W = (1000000, 1050000), 1000020, 1600
    dels = lambda k: [synth.Deletion(1020000, 1500, [2, 1, 0][k % 3])]
    parts, offsets, rgs, sample_offsets = [], [0], [], [0]
    seed = 8800
    for k in range(9):
        n_rg = 3 if k % 2 == 0 else 1
        for j in range(n_rg):
            seed += 1
            sub = synth.Library(coverage=lib.coverage / n_rg)
            pos, dev = synth.sample_pairs(seed, region, sub, dels(k))
            d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
            parts.append((d, s, a, b))
            offsets.append(offsets[-1] + len(d))
            rgs.append(rg_tight if (n_rg == 3 and j == 1) else rg)
        sample_offsets.append(len(rgs))
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), *[np.concatenate([p[i] for p in parts]) for i in range(4)])
    screen, full, oracle = three_ways(seg, rgs, sample_offsets, monkeypatch)
    assert oracle[0]["n_calls"] > 5
    assert_identical(screen, full)
    assert_same(screen, oracle)


def test_cells_whose_q3_sits_on_the_threshold(built, monkeypatch):
    """Hand-made cells: 2..13 active pairs whose deviations are drawn from {T - 1, T, T + 1, T + 2, 40}, so that the two order
    statistics Q3 interpolates straddle the threshold T in every possible way (a <= T < b with every weight k / 4, a == T,
    b == T + 1, n < 4). The screen must flag every window whose exact Q3 exceeds T (checked cell by cell inside
    pd_stage1_last) and the calls must be the full scan's and the oracle's."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    T = rg.min_init_del_len
    rng = np.random.default_rng(4242)
    n, groups, span = 24, 90, 3
    W, first = groups * span, 1000020
    parts, offsets = [], [0]
    for k in range(n):
        d, s, a, b = [], [], [], []
        for g in range(groups):
            m = int(rng.integers(2, 14))
            vals = rng.choice([T - 1, T, T + 1, T + 2, 40], size=m, p=[0.2, 0.25, 0.25, 0.1, 0.2])
            d += [int(v) for v in vals]
            s += [first + 30 * span * g] * m
            a += [span * g] * m
            b += [span * g + span] * m
        parts.append((np.array(d, np.int32), np.array(s, np.uint32), np.array(a, np.uint32), np.array(b, np.uint32)))
        offsets.append(offsets[-1] + len(d))
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), *[np.concatenate([p[i] for p in parts]) for i in range(4)])
    screen, full, oracle = three_ways(seg, [rg] * n, list(range(n + 1)), monkeypatch, min_len=100)
    q3, fl = oracle[1], oracle[2]
    on_edge = ((q3 == T) | (q3 == T + 1)) & ((fl & 2) != 0)
    assert on_edge.sum() > 200 and (q3 > T).any() and (q3[(fl & 2) != 0] <= T).any()
    assert_identical(screen, full)
    assert_same(screen, oracle)


def test_the_view_is_refused_unless_asked_for(built):
    """pd_stage1_last() on a screening context without pd_ctx_stage1_view(): an error that says what to do, not stale values."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    seg = synth.cohort_segment(4, (1000000, 1010000), 1000020, 300, lib, deletions=None)
    ctx = capi.Context(capi.load_hip(), [rg] * 4, list(range(5)))
    ctx.collect(ctx.submit(seg))
    with pytest.raises(capi.BackendError, match="pd_ctx_stage1_view"):
        ctx.stage1()
    ctx.close()
