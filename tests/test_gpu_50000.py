"""GPU: the shape of BASELINE configs[4] on one GPU -- 50,000 samples x a full 200-kb segment (6,667 windows, 10^9 pair
instances) -- through size-independent properties: stage-1 Q3 of randomly drawn cells against numpy, identical results
from repeated runs on resident data, and stage 2 cut into several batches of cells (the per-(cell, sample) buffers of a
large cohort do not fit one batch in general) against the single batch."""
import numpy as np
import pytest

from test_gpu_fullsize import q3_numpy

pytestmark = pytest.mark.gpu

S, W = 50000, 6667


@pytest.fixture(scope="module")
def huge():
    import torch
    from popdel_amd import capi, synth
    import bench
    first, w, offsets, arrays, planted = bench.device_tile(S, 4242, torch.device("cuda", 0))
    assert w == W
    rg = synth.conditioned_readgroup(synth.Library())
    ctx = capi.Context(capi.load_hip(), [rg] * S, list(range(S + 1)), stage1_view=True)
    seg = capi.Segment(first, w, offsets, arrays["dev"], arrays["start"], arrays["on"], arrays["off"])
    handle = ctx.stage(seg)
    yield ctx, handle, offsets, arrays
    ctx.unstage(handle)
    ctx.close()


def test_q3_and_repeatability_at_50000_samples(built, huge):
    ctx, handle, offsets, arrays = huge
    a = ctx.collect(ctx.run(handle))
    q3, flags = ctx.stage1()
    assert q3.shape == (W, S) and a["n_calls"] > 0 and a["pl"].shape[1] == S
    rng = np.random.default_rng(9)
    checked = 0
    for s in rng.integers(0, S, 50):
        lo, hi = int(offsets[s]), int(offsets[s + 1])
        on, off, dev = arrays["on"][lo:hi], arrays["off"][lo:hi], arrays["dev"][lo:hi]
        for w in rng.integers(0, W, 30):
            act = dev[(on <= w) & (off > w)]
            if len(act) < 2:
                assert flags[w, s] & 1
            else:
                assert flags[w, s] == 2 and q3[w, s] == q3_numpy(act), (s, w)
                checked += 1
    assert checked > 1200
    b = ctx.collect(ctx.run(handle))
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    huge[0].single_batch = a


def test_stage2_in_batches_equals_one_batch_at_50000_samples(built, huge, monkeypatch):
    ctx, handle, _, _ = huge
    one = getattr(ctx, "single_batch", None) or ctx.collect(ctx.run(handle))
    # per cell and sample ~ 0.5 kB of stage-2 buffers: 300 MB hold about a dozen cells of this cohort
    monkeypatch.setenv("POPDEL_HIP_STAGE2_BYTES", "300000000")
    many = ctx.collect(ctx.run(handle))
    assert ctx.timing()["n_candidate_cells"] > 24
    for k in one:
        assert np.array_equal(np.asarray(one[k]), np.asarray(many[k])), k
