"""GPU: a context with lanes (POPDEL_HIP_LANES = K: K child contexts that take the tickets in turn, each on a host thread of
its own, so that one segment's stage 2 fills the gaps of another's) must return exactly what a plain context returns, in
ticket order -- through pd_segment_run / collect with several tickets in flight, through pd_segment_submit, with the
columns merged on the device, sharded with one rank, and through the product binary."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, POPDEL_BIN, run_call

pytestmark = pytest.mark.gpu


def segments(n_samples, count):
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    segs = []
    for k in range(count):
        region = (1000000 + 100000 * k, 1060000 + 100000 * k)
        dels = (lambda kk, r, k=k: [synth.Deletion(r[0] + 30000, 600 + 250 * k, [1, 2, 0][(kk + k) % 3])]) if k != 2 else None
        segs.append(synth.cohort_segment(n_samples, region, region[0] + 20, 1900, lib, deletions=dels, seed_base=1000 + 50 * k))
    return rg, segs


def same(a, b):
    assert a["n_calls"] == b["n_calls"]
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k]), k


@pytest.mark.parametrize("lanes", [2, 3])
def test_lanes_return_what_a_plain_context_returns(built, monkeypatch, lanes):
    from popdel_amd import capi
    n = 300
    rg, segs = segments(n, 5)
    hip = capi.load_hip()
    monkeypatch.setenv("POPDEL_HIP_LANES", "1")
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)), stage1_view=True)
    want = [ctx.collect(ctx.submit(s)) for s in segs]
    want_q3 = ctx.stage1()
    ctx.close()
    assert sum(w["n_calls"] for w in want) > 30 and want[2]["n_calls"] == 0
    monkeypatch.setenv("POPDEL_HIP_LANES", str(lanes))
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)), stage1_view=True)
    # staged segments, as many tickets in flight as the context takes (2 per lane), collected in order
    handles = [ctx.stage(s) for s in segs]
    order = [i % len(segs) for i in range(4 * lanes + 3)]
    tickets, got = [], []
    for i in order:
        if len(tickets) == 2 * lanes:
            got.append(ctx.collect(tickets.pop(0)))
        tickets.append(ctx.run(handles[i]))
    with pytest.raises(Exception):
        for _ in range(2 * lanes + 1):
            tickets.append(ctx.run(handles[0]))          # one too many
    while tickets:
        got.append(ctx.collect(tickets.pop(0)))
    for i, g in zip(order + [0] * 2 * lanes, got):
        same(g, want[i])
    for h in handles:
        ctx.unstage(h)
    # submit + collect, and the stage-1 view of the last collected ticket
    for s, w in zip(segs, want):
        same(ctx.collect(ctx.submit(s)), w)
    q3 = ctx.stage1()
    assert np.array_equal(q3[0], want_q3[0]) and np.array_equal(q3[1], want_q3[1])
    assert ctx.timing()["n_cells"] == n * 1900
    ctx.close()


def test_lanes_with_one_rank_sharded_and_merged_columns(built, monkeypatch):
    from popdel_amd import capi, distributed
    n = 300
    rg, segs = segments(n, 3)
    hip = capi.load_hip()
    monkeypatch.setenv("POPDEL_HIP_LANES", "1")
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
    want = [ctx.collect(ctx.submit(s)) for s in segs]
    ctx.close()
    monkeypatch.setenv("POPDEL_HIP_LANES", "2")
    sh = distributed.SampleShard(hip, [rg] * n, list(range(n + 1)), 0, 1, native=True)
    for s, w in zip(segs, want):
        same(sh.ctx.collect(sh.ctx.submit(s)), w)
        assert sh.ctx.timing()["n_all_gathers"] > 0
    sh.close()
    # columns kept on the device and merged there (pd_calls_merge finds the lane that holds them)
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
    ctx.keep_results_on_device(True)
    for s, w in zip(segs, want):
        p = ctx.collect_raw(ctx.submit(s))
        k = p.contents.n_calls
        assert k == w["n_calls"]
        if k:
            assert not p.contents.pl
            m = ctx.merge(p, [list(range(k))], [0])
            assert m is not None
            ctx.fetch_columns(p)
            pl = np.ctypeslib.as_array(p.contents.pl, shape=(k, n, 3))
            assert np.array_equal(pl, w["pl"])
        ctx.free_raw(p)
    ctx.close()


def test_product_binary_with_lanes_writes_the_golden_vcf(built, cohort_dir, tmp_path, monkeypatch):
    from popdel_amd import synth
    monkeypatch.setenv("POPDEL_HIP_LANES", "2")
    d = cohort_dir("s2", 2, synth.single_deletion)
    assert run_call(POPDEL_BIN, d, str(tmp_path / "a.vcf")) == open(os.path.join(GOLDEN, "s2_default.vcf")).read()
    assert run_call(POPDEL_BIN, d, str(tmp_path / "b.vcf"), "-n") == open(os.path.join(GOLDEN, "s2_windowwise.vcf")).read()
    d = cohort_dir("s200", 200, synth.single_deletion)
    assert run_call(POPDEL_BIN, d, str(tmp_path / "c.vcf")) == open(os.path.join(GOLDEN, "s200_default.vcf")).read()
