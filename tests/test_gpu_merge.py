"""GPU: the per-sample half of call merging on the device (pd_calls_merge; reference utils_popdel.h:446-501 setGenotypes,
:339-359 setFreqFromGTs) against a numpy restatement of those lines, on the window calls of random cohorts and randomly
drawn groups; the columns left in HBM equal the ones an ordinary collect returns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def merged_reference(calls, groups, heads):
    S = calls["n_samples"]
    pl = np.zeros((len(groups), S, 3), np.uint32)
    lad = np.zeros((len(groups), S, 3), np.uint32)
    dad = np.zeros((len(groups), S, 5), np.uint32)
    alleles = np.zeros(len(groups), np.uint32)
    for g, members in enumerate(groups):
        count = len(members)
        sums = calls["pl"][members].astype(np.int64).sum(axis=0)                 # [S][3]
        mn = sums.min(axis=1, keepdims=True).astype(np.float64)
        mean = (sums - mn) / count
        r = np.floor(np.abs(mean) + 0.5).astype(np.int64)                         # std::round: half away from zero (values >= 0)
        ref, het, hom = mean[:, 0], mean[:, 1], mean[:, 2]
        tie12 = r[:, 1] == r[:, 2]
        r[:, 1] += tie12 & (het > hom)
        r[:, 2] += tie12 & ~(het > hom)
        tie01 = ~tie12 & (r[:, 0] == r[:, 1])
        r[:, 0] += tie01 & (ref > het)
        r[:, 1] += tie01 & ~(ref > het)
        pl[g] = r
        m = count // 2                                                            # upper median
        lad[g] = np.sort(calls["lad"][members], axis=0)[m]
        dad[g] = np.sort(calls["dad"][members], axis=0)[m]
        alleles[g] = int(np.where(r[:, 0] == 0, 0, np.where(r[:, 1] == 0, 1, 2)).sum())
    return dict(pl=pl, lad=lad, dad=dad, first_last=calls["first_last"][heads], alleles=alleles)


@pytest.mark.parametrize("seed", [3, 7, 14, 20])
def test_merge_on_device_against_numpy(built, seed):
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(seed)
    hip = capi.load_hip()
    plain = capi.Context(hip, rgs, offs, **opts)
    want = plain.collect(plain.submit(seg))
    plain.close()
    n = want["n_calls"]
    assert n > 10
    ctx = capi.Context(hip, rgs, offs, **opts)
    ctx.keep_results_on_device(True)
    p = ctx.collect_raw(ctx.submit(seg))
    assert p.contents.n_calls == n and not p.contents.pl            # call-level fields only
    rng = np.random.default_rng(seed)
    groups, heads = [], []
    at = 0
    while at < n:
        size = int(min(n - at, rng.integers(1, 40)))
        members = np.sort(rng.choice(np.arange(at, at + size), size=int(rng.integers(1, size + 1)), replace=False))
        groups.append(members.tolist())
        heads.append(int(rng.integers(at, at + size)))
        at += size
    got = ctx.merge(p, groups, heads)
    assert got is not None
    ref = merged_reference(want, groups, heads)
    for k in ("pl", "lad", "dad", "first_last", "alleles"):
        assert np.array_equal(got[k], ref[k]), k
    # the columns themselves, fetched after the fact, are what an ordinary collect returns
    ctx.fetch_columns(p)
    c = p.contents
    S = c.n_samples
    assert np.array_equal(np.ctypeslib.as_array(c.pl, shape=(n * S * 3,)).reshape(n, S, 3), want["pl"])
    assert np.array_equal(np.ctypeslib.as_array(c.dad, shape=(n * S * 5,)).reshape(n, S, 5), want["dad"])
    ctx.free_raw(p)
    ctx.close()
