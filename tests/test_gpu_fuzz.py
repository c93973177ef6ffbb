"""GPU: seeded random cohorts -- library shape, coverage, cohort size, deletions, genotypes and calling options all drawn
at random -- HIP against the oracle through the C ABI. Each case is small enough for the oracle to finish in a second or
two; together they sweep corners the hand-made cases do not (short/long libraries, 100-bp reads, deep and shallow
coverage side by side, overlapping deletions, few iterations, somatic and window-wise modes)."""
import numpy as np
import pytest

from conftest import ORACLE_LIB
from test_gpu_parity import INT_KEYS, both

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
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32),
                       np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    opts = dict(iterations=int(rng.choice([15, 15, 3, 7])), window_wise_output=bool(rng.integers(0, 2)),
                somatic=bool(rng.integers(0, 4) == 0), min_sample_fraction=float(rng.choice([0.1, 0.6])),
                min_len=int(rng.choice([200, 120, 400])))
    return seg, rgs, list(range(n + 1)), opts


def assert_same_up_to_knife_edges(gpu, cpu):
    """Everything the hand-made cases demand, with one documented exception (Numerics): a reference shift is
    int(sum(p * dev) / sum(p)); when the pairs of a read group carry (almost) equal weights and their deviations add up
    to a multiple of their number, that quotient sits exactly on an integer and its truncation depends on the last bit
    -- x87 intermediates and the hash-set iteration order in the reference, FP64 and pair order here. The shift then
    differs by one and a phred value of that sample column may move by one. Such columns are tolerated if they are
    rare (<= 1 in 5,000), the genotype they imply is unchanged and nothing else differs."""
    (gc, gq, gf), (oc, oq, of) = gpu, cpu
    assert np.array_equal(gf, of)
    assert np.array_equal(gq, oq)
    assert gc["n_calls"] == oc["n_calls"]
    for k in INT_KEYS:
        if k not in ("pl", "lad", "dad"):
            assert np.array_equal(gc[k], oc[k]), k
    assert np.allclose(gc["likelihood_ratio"], oc["likelihood_ratio"], rtol=1e-6, atol=1e-6)
    assert np.allclose(gc["frequency"], oc["frequency"], rtol=0, atol=1e-6)
    if gc["n_calls"] == 0:
        return 0
    bad = np.zeros(gc["pl"].shape[:2], bool)
    for k in ("pl", "lad", "dad"):
        bad |= (gc[k] != oc[k]).any(axis=2)
    n_bad = int(bad.sum())
    assert n_bad <= max(1, bad.size // 5000), (n_bad, bad.size)
    if n_bad:
        d = np.abs(gc["pl"].astype(np.int64) - oc["pl"].astype(np.int64))
        assert d.max() <= 2
        assert np.array_equal(np.argmin(gc["pl"], axis=2), np.argmin(oc["pl"], axis=2))
        assert np.array_equal((gc["pl"] == 0), (oc["pl"] == 0))
    return n_bad


@pytest.mark.parametrize("seed", range(16))
def test_random_cohort(built, seed):
    seg, rgs, offs, opts = random_case(seed)
    gpu, cpu = both(seg, rgs, offs, **opts)
    n_bad = assert_same_up_to_knife_edges(gpu, cpu)
    if n_bad:
        print("seed %d: %d knife-edge sample column(s) of %d" % (seed, n_bad, gpu[0]["pl"].shape[0] * gpu[0]["pl"].shape[1]))


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
