"""GPU: size-independent properties at the BASELINE-sized launch (10,000 samples x a full 200-kb segment), where the
oracle would take minutes: (1) stage-1 Q3 of randomly drawn cells against numpy; (2) cutting the segment in two and
carrying the active pairs over gives exactly the calls of the uncut segment; (3) every pass over resident data gives
the same answer (no state leaks between runs)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

S, W, FIRST = 10000, 6667, 1000020        # configs[2]: the bench workload itself


@pytest.fixture(scope="module")
def big():
    import torch
    from popdel_amd import capi, synth
    import bench
    first, w, offsets, arrays, planted = bench.device_tile(S, 77, torch.device("cuda", 0))
    assert (first, w) == (FIRST, W)
    rg = synth.conditioned_readgroup(synth.Library())
    ctx = capi.Context(capi.load_hip(), [rg] * S, list(range(S + 1)), stage1_view=True)
    seg = capi.Segment(first, w, offsets, arrays["dev"], arrays["start"], arrays["on"], arrays["off"])
    return ctx, seg, offsets, arrays


def q3_numpy(values):
    v = np.sort(values.astype(np.int64))
    n = len(v)
    if n < 4:
        return int(v[-1])
    pos = (3.0 * n + 2.0 + (n % 2)) / 4.0 - 1.0
    l = int(pos)
    r = pos - l
    return int(np.floor((1 - r) * v[l] + r * v[l + 1] + 0.5))


def test_q3_of_random_cells_matches_numpy(built, big):
    ctx, seg, offsets, arrays = big
    calls = ctx.collect(ctx.submit(seg))
    q3, flags = ctx.stage1()
    assert q3.shape == (W, S)
    rng = np.random.default_rng(5)
    checked = 0
    for s in rng.integers(0, S, 60):
        a, b = int(offsets[s]), int(offsets[s + 1])
        on, off, dev = arrays["on"][a:b], arrays["off"][a:b], arrays["dev"][a:b]
        for w in rng.integers(0, W, 40):
            act = dev[(on <= w) & (off > w)]
            if len(act) < 2:
                assert flags[w, s] & 1
            else:
                assert flags[w, s] == 2 and q3[w, s] == q3_numpy(act), (s, w)
                checked += 1
    assert checked > 2000
    assert calls["n_calls"] > 0
    big[0].calls_full = calls


def test_repeated_runs_on_resident_data_are_identical(built, big):
    ctx, seg, _, _ = big
    h = ctx.stage(seg)
    a = ctx.collect(ctx.run(h))
    b = ctx.collect(ctx.run(h))
    ctx.unstage(h)
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_cut_segment_with_carry_over_gives_the_same_calls(built, big):
    from popdel_amd import capi
    ctx, seg, offsets, arrays = big
    full = ctx.collect(ctx.submit(seg))
    cut = 3301
    parts = []
    for lo, hi in ((0, cut), (cut, W)):
        keep = (arrays["off"] > lo) & (arrays["on"] < hi)
        counts = np.add.reduceat(keep.astype(np.uint64), offsets[:-1].astype(np.int64))
        counts[np.diff(offsets.astype(np.int64)) == 0] = 0
        off2 = np.zeros(S + 1, dtype=np.uint64)
        np.cumsum(counts, out=off2[1:])
        on = np.maximum(arrays["on"][keep].astype(np.int64) - lo, 0).astype(np.uint32)
        off = (np.minimum(arrays["off"][keep].astype(np.int64), hi) - lo).astype(np.uint32)
        # carried-over pairs (on = 0) must come first within each sample: stable sort by `on` per sample
        order = np.lexsort((on, np.repeat(np.arange(S), counts.astype(np.int64))))
        part = capi.Segment(FIRST + 30 * lo, hi - lo, off2, arrays["dev"][keep][order], arrays["start"][keep][order], on[order],
                            off[order])
        parts.append(ctx.collect(ctx.submit(part)))
    for key in ("initial_length", "iterations", "deletion_length", "window_position", "position", "end_position", "filter", "pl",
                "lad", "dad", "first_last"):
        joined = np.concatenate([parts[0][key], parts[1][key]])
        assert np.array_equal(joined, full[key]), key
    assert np.allclose(np.concatenate([parts[0]["likelihood_ratio"], parts[1]["likelihood_ratio"]]), full["likelihood_ratio"],
                       rtol=1e-9, atol=1e-9)


def test_more_than_one_sweep_of_sample_tiles_against_the_oracle(built):
    """10,500 samples = 42 tiles of 256: the per-sample kernels of stage 2 pad their grid to 48 tiles (an XCD then always
    serves the same tiles) and walk them in two sweeps (40 + 8 tiles, all cells per sweep; pd::kTilesPerSweep). Calls and
    per-sample columns against the oracle."""
    from conftest import ORACLE_LIB
    from popdel_amd import capi, synth
    from test_gpu_parity import INT_KEYS
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n, region, first, W = 10500, (1000000, 1017000), 1000020, 520
    seg = synth.cohort_segment(n, region, first, W, lib, deletions=lambda k, r: [synth.Deletion(1007000, 900, [1, 0, 2, 0, 0][k % 5])])
    out = []
    for library in (capi.load_hip(), capi.load_library(ORACLE_LIB)):
        ctx = capi.Context(library, [rg] * n, list(range(n + 1)))
        out.append(ctx.collect(ctx.submit(seg)))
        ctx.close()
    got, want = out
    assert want["n_calls"] > 10 and got["n_calls"] == want["n_calls"]
    for k in INT_KEYS:
        assert np.array_equal(got[k], want[k]), k
    assert np.allclose(got["likelihood_ratio"], want["likelihood_ratio"], rtol=1e-6, atol=1e-6)
    assert np.allclose(got["frequency"], want["frequency"], rtol=0, atol=1e-6)


def test_multi_read_group_tiles_with_a_padded_grid_against_the_oracle(built):
    """3,400 samples with two read groups each = 14 multi-read-group tiles: the lane-per-read-group kernels run on a grid
    padded to 16 tiles. Against the oracle."""
    from conftest import ORACLE_LIB
    from popdel_amd import capi, synth
    from test_gpu_parity import INT_KEYS
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n, region, first, W = 3400, (1000000, 1017000), 1000020, 520
    rng = np.random.default_rng(77)
    parts, offsets = [], [0]
    for k in range(n):
        pos, dev = synth.sample_pairs(3000 + k, region, lib, [synth.Deletion(1007000, 900, [1, 0, 2, 0][k % 4])])
        d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
        which = rng.integers(0, 2, size=len(d))
        for r in range(2):                                  # (stable: each read group stays ordered by its first window)
            sel = which == r
            parts.append((d[sel], s[sel], a[sel], b[sel]))
            offsets.append(offsets[-1] + int(sel.sum()))
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), *[np.concatenate([p[i] for p in parts]) for i in range(4)])
    out = []
    for library in (capi.load_hip(), capi.load_library(ORACLE_LIB)):
        ctx = capi.Context(library, [rg] * (2 * n), list(range(0, 2 * n + 1, 2)))
        out.append(ctx.collect(ctx.submit(seg)))
        ctx.close()
    got, want = out
    assert want["n_calls"] > 10 and got["n_calls"] == want["n_calls"]
    for k in INT_KEYS:
        assert np.array_equal(got[k], want[k]), k
    assert np.allclose(got["likelihood_ratio"], want["likelihood_ratio"], rtol=1e-6, atol=1e-6)
    assert np.allclose(got["frequency"], want["frequency"], rtol=0, atol=1e-6)
