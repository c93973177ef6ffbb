"""GPU: the staged segment layout built on the device (build.hip; instances cross PCIe once, 5, 8 or 16 bytes each) against the
host-side builder (POPDEL_HIP_HOST_BUILD=1, kept for segments the device builder does not take): every result field
identical, floating point included -- chunks keep the order of the pair instances either way -- and both equal to the
oracle."""
import numpy as np
import pytest

from conftest import ORACLE_LIB
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def run(lib, seg, rgs, offs, opts, monkeypatch=None, host_build=False):
    from popdel_amd import capi
    if monkeypatch is not None:
        if host_build:
            monkeypatch.setenv("POPDEL_HIP_HOST_BUILD", "1")
        else:
            monkeypatch.delenv("POPDEL_HIP_HOST_BUILD", raising=False)
    ctx = capi.Context(lib, rgs, offs, stage1_view=True, **opts)
    calls = ctx.collect(ctx.submit(seg))
    q3, fl = ctx.stage1()
    tm = ctx.timing()
    ctx.close()
    return calls, q3, fl, tm


def identical(a, b):
    (ca, qa, fa, ta), (cb, qb, fb, tb) = a, b
    assert np.array_equal(qa, qb) and np.array_equal(fa, fb)
    assert ca["n_calls"] == cb["n_calls"]
    for k, v in ca.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, cb[k]), k                 # likelihood ratio and frequency bit for bit, too
    for k in ("n_pair_chunks", "n_active_pairs", "max_life", "n_candidate_cells"):
        assert ta[k] == tb[k], k


@pytest.mark.parametrize("seed", [1, 2, 5, 8, 11, 23])          # 2, 5, 8, 11, 23: samples with several read groups
def test_device_build_equals_host_build_and_oracle(built, seed, monkeypatch):
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(seed)
    hip = capi.load_hip()
    dev = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=False)
    host = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=True)
    identical(dev, host)
    assert dev[0]["n_calls"] > 0
    if seed in (1, 2):
        cpu = run(capi.load_library(ORACLE_LIB), seg, rgs, offs, opts)
        # (the knife-edge tolerance of the fuzz tests is not needed for these two seeds)
        assert_same(dev[:3], cpu[:3])


def test_wide_upload_when_fields_do_not_fit_16_bits(built, monkeypatch):
    """A deviation beyond int16 and a pair start far behind its first window: the segment is uploaded with 32-bit fields;
    the read group with the wide deviation leaves the packed scan (fallback kernel), the result is the host builder's."""
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(3)
    o = np.asarray(seg.rg_offsets, dtype=np.int64)
    seg.dev = seg.dev.copy()
    seg.start = seg.start.copy()
    seg.dev[o[1] + 5] = 70000                       # read group 1
    seg.start[o[2] + 7] = seg.start[o[2] + 7] - 100000 if seg.start[o[2] + 7] > 100000 else seg.start[o[2] + 7]
    hip = capi.load_hip()
    dev = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=False)
    host = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=True)
    identical(dev, host)


def test_unordered_instances_are_refused(built, monkeypatch):
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(4)
    o = np.asarray(seg.rg_offsets, dtype=np.int64)
    lo, hi = int(o[0]), int(o[1])
    perm = np.arange(lo, hi)[::-1]
    for name in ("dev", "start", "on", "off"):
        arr = getattr(seg, name).copy()
        arr[lo:hi] = arr[perm]
        setattr(seg, name, arr)
    monkeypatch.delenv("POPDEL_HIP_HOST_BUILD", raising=False)
    ctx = capi.Context(capi.load_hip(), rgs, offs, **opts)
    with pytest.raises(capi.BackendError, match="not ordered"):
        ctx.submit(seg)
    ctx.close()


@pytest.mark.parametrize("scan_life", ["5", "8", "13"])
def test_results_do_not_depend_on_the_chunk_life_of_the_stage1_image(built, scan_life, monkeypatch):
    """The stage-1 image is cut with its own chunk life (Q3 is an order statistic of integers): whatever that life, every
    field of the result is the same, floating point included."""
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(8)
    hip = capi.load_hip()
    monkeypatch.delenv("POPDEL_HIP_MAXLIFE_SCAN", raising=False)
    want = run(hip, seg, rgs, offs, opts)
    monkeypatch.setenv("POPDEL_HIP_MAXLIFE_SCAN", scan_life)
    got = run(hip, seg, rgs, offs, opts)
    identical(got, want)


def test_compact_upload_and_its_exceptions(built, monkeypatch):
    """The narrow upload is compact by default (5 bytes per instance): `on` as a one-byte difference to the previous instance
    of the read group, the life and the distance of the pair start behind its first window in one byte each, exceptions in
    a list (every pair that was active before the segment began is one: its start lies far behind window 0). A pair that
    lives 600 windows, a read group whose pairs pause for 300 windows, one that starts 400 windows into the segment: same
    result as the plain 8-byte upload and as the host builder."""
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(6)
    W, shift = seg.n_windows, 30
    o = np.asarray(seg.rg_offsets, dtype=np.int64)
    for name in ("dev", "start", "on", "off"):
        setattr(seg, name, getattr(seg, name).copy())
    # read group 0: one long-lived pair
    i = int(o[0]) + 40
    seg.off[i] = min(W, int(seg.on[i]) + 600)
    # read group 1: nothing new for 300 windows from its 50th instance on
    a, b = int(o[1]) + 50, int(o[2])
    seg.on[a:b] += 300
    seg.off[a:b] += 300
    seg.start[a:b] += 300 * shift
    # read group 2: starts 400 windows into the segment
    a, b = int(o[2]), int(o[3])
    seg.on[a:b] += 400
    seg.off[a:b] += 400
    seg.start[a:b] += 400 * shift
    keep = seg.on < W                                   # (instances pushed beyond the segment are dropped, read group by read group)
    counts = [int(keep[int(o[r]):int(o[r + 1])].sum()) for r in range(len(o) - 1)]
    for name in ("dev", "start", "on", "off"):
        setattr(seg, name, getattr(seg, name)[keep])
    seg.off = np.minimum(seg.off, W).astype(seg.off.dtype)
    seg.rg_offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    hip = capi.load_hip()
    monkeypatch.delenv("POPDEL_HIP_PLAIN_UPLOAD", raising=False)
    compact = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=False)
    monkeypatch.setenv("POPDEL_HIP_PLAIN_UPLOAD", "1")
    plain = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=False)
    monkeypatch.delenv("POPDEL_HIP_PLAIN_UPLOAD", raising=False)
    host = run(hip, seg, rgs, offs, opts, monkeypatch, host_build=True)
    identical(compact, plain)
    identical(compact, host)
    assert compact[3]["max_life"] >= 4 and compact[0]["n_calls"] > 0
