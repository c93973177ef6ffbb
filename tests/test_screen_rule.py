"""CPU: the arithmetic behind stage 1's screen (popdel_amd/csrc/hip/stage1.hip, screenMaybe), restated in Python and checked by
brute force against the reference's upperHalfMedian (genotype_deletion_popdel_call.h:15-27, as the oracle restates it): for
every multiset of deviations, "the screen flags the cell" must follow from "Q3 exceeds the threshold" -- the screen may flag
more (a MAYBE), never less -- and where it says no, Q3 is at or below the threshold. The count c of active deviations above
the threshold and the number n of active pairs are all the screen knows of a cell."""
import itertools

import numpy as np


def q3_reference(values):
    """upperHalfMedian as the scan computes it (stage1.hip: q3Ranks + q3FromRankPair; equals the oracle's on every input the GPU
    parity tests compare)."""
    v = sorted(values)
    n = len(v)
    if n < 4:
        return v[n - 1]
    pos4 = 3 * n + 2 + (n & 1) - 4
    l0, k = pos4 >> 2, pos4 & 3
    a, b = v[l0], v[min(l0 + 1, n - 1)]
    return ((4 - k) * a + k * b + 2) >> 2


def screen_maybe(n, c):
    """stage1.hip: screenMaybe (n >= 2)."""
    if n < 4:
        return c >= 1
    pos4 = 3 * n + 2 + (n & 1) - 4
    l0, k = pos4 >> 2, pos4 & 3
    need = n - l0
    return c >= need if k == 0 else c + 1 >= need


def test_the_screen_never_misses_a_q3_above_the_threshold_exhaustively():
    """All multisets over {T - 2 .. T + 3} of 2..9 elements."""
    T = 200
    alphabet = [T - 2, T - 1, T, T + 1, T + 2, T + 3]
    sure = maybe = 0
    for n in range(2, 10):
        for combo in itertools.combinations_with_replacement(alphabet, n):
            q = q3_reference(combo)
            c = sum(1 for x in combo if x > T)
            flagged = screen_maybe(n, c)
            if q > T:
                assert flagged, (combo, q, c)
                sure += 1
            elif flagged:
                maybe += 1              # a MAYBE: resolved by the exact pass
            else:
                assert q <= T
    assert sure > 1000 and maybe > 0


def test_the_screen_on_random_cells_of_every_size():
    rng = np.random.default_rng(11)
    missed = maybes = 0
    for n in range(2, 130):
        for _ in range(300):
            T = int(rng.integers(-50, 400))
            spread = int(rng.choice([1, 3, 40, 400]))
            vals = (T + rng.integers(-spread, spread + 1, size=n)).tolist()
            if rng.random() < 0.3:                      # a cluster of supporting pairs far above
                k = int(rng.integers(1, n + 1))
                vals[:k] = (T + 500 + rng.integers(0, 50, size=k)).tolist()
            q = q3_reference(vals)
            c = sum(1 for x in vals if x > T)
            flagged = screen_maybe(n, c)
            if q > T and not flagged:
                missed += 1
            if q <= T and flagged:
                maybes += 1
                # a maybe needs b > T: exactly n - l0 - 1 values above T and a weight on b
                pos4 = 3 * n + 2 + (n & 1) - 4
                assert n >= 4 and c == n - (pos4 >> 2) - 1 and (pos4 & 3) != 0
    assert missed == 0 and maybes > 0


def test_the_restated_q3_is_the_oracles(built):
    """q3_reference above against the oracle's upperHalfMedian through the C ABI: one window, every sample a random multiset."""
    from conftest import ORACLE_LIB
    from popdel_amd import capi, synth
    rng = np.random.default_rng(5)
    rg = synth.conditioned_readgroup(synth.Library(), max_load=1000)
    n_samples = 120
    devs, offsets, want = [], [0], []
    for s in range(n_samples):
        n = int(rng.integers(2, 100))
        v = rng.integers(-150, 900, size=n)
        devs.append(v)
        offsets.append(offsets[-1] + n)
        want.append(q3_reference(v.tolist()))
    dev = np.concatenate(devs).astype(np.int32)
    z = np.zeros(len(dev), np.uint32)
    seg = capi.Segment(1000020, 1, np.array(offsets, np.uint64), dev, z + 1000020, z, z + 1)
    ctx = capi.Context(capi.load_library(ORACLE_LIB), [rg] * n_samples, list(range(n_samples + 1)), stage1_view=True)
    ctx.collect(ctx.submit(seg))
    q3, fl = ctx.stage1()
    ctx.close()
    assert (fl[0] == 2).all() and q3[0].tolist() == want
