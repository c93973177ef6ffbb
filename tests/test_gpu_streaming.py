"""GPU: segments streamed from host memory (BASELINE configs[4]'s mechanism: stage(N + 1) -- packing on the host threads, upload
and layout build on the copy stream -- while ticket N computes) give the calls of the same segments staged one at a time on an
idle context, bit for bit; at the bench's cohort size (10,000 samples x 6,667 windows) through two different segments taken in
turn, and with the staging's own figures (pd_last_stage_timing) in a sane relation to each other."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("initial_length", "iterations", "deletion_length", "window_position", "position", "end_position", "filter", "pl", "lad", "dad",
        "first_last", "likelihood_ratio", "frequency", "near_integer_shifts")


def same(a, b):
    assert a["n_calls"] == b["n_calls"]
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n_samples", [700, 10000])
def test_streamed_segments_equal_resident_ones(built, n_samples):
    import torch
    import bench
    from popdel_amd import capi, synth
    dev = torch.device("cuda", 0)
    rg = synth.conditioned_readgroup(synth.Library())
    segs = []
    for seed, n_del in ((31, 1), (32, 2)):
        first, W, offsets, a, _ = bench.device_tile(n_samples, seed, dev, n_deletions=n_del)
        segs.append(capi.Segment(first, W, offsets, a["dev"], a["start"], a["on"], a["off"]))
    hip = capi.load_hip()
    # one at a time on an idle context
    ctx = capi.Context(hip, [rg] * n_samples, list(range(n_samples + 1)))
    want = []
    for sg in segs:
        h = ctx.stage(sg)
        want.append(ctx.collect(ctx.run(h)))
        ctx.unstage(h)
    ctx.close()
    assert all(w["n_calls"] > 0 for w in want)
    # streamed: the next segment is staged while the open ticket computes; six segments, the two taken in turn
    ctx = capi.Context(hip, [rg] * n_samples, list(range(n_samples + 1)))
    order = [0, 1, 0, 1, 1, 0]
    h_cur = ctx.stage(segs[order[0]])
    t_cur = ctx.run(h_cur)
    got = []
    for i in range(len(order)):
        if i + 1 < len(order):
            h_nxt = ctx.stage(segs[order[i + 1]])
            st = ctx.stage_timing()
            assert st["n_instances"] == len(segs[order[i + 1]].dev) and st["flavour"] == 5
            assert st["bytes_uploaded"] >= 5 * st["n_instances"] and 0 < st["pack_seconds"] <= st["uploaded_seconds"] * 1.05
            t_nxt = ctx.run(h_nxt)
        got.append(ctx.collect(t_cur))
        ctx.unstage(h_cur)
        if i + 1 < len(order):
            h_cur, t_cur = h_nxt, t_nxt
    ctx.close()
    for i, g in zip(order, got):
        same(g, want[i])
    assert hip.pd_host_copy_gbs(1 << 28) > 0.1
