"""GPU: pd_walk_submit through the C ABI -- the window loop in closed form on the device -- against pd_segment_submit with
the instances the same pairs have under the interval rule (popdel_amd.synth.instances_by_interval_rule), and the state it
carries from one segment to the next."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def cohort(n, region, seed0=300):
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    pairs = [synth.sample_pairs(seed0 + k, region, lib, [synth.Deletion(region[0] + 20000, 900, [1, 2, 0][k % 3])]) for k in range(n)]
    return lib, rg, pairs


def test_first_segment_equals_the_interval_rule(built):
    """A first segment (no step ahead of window 0, nothing carried, every end record in the read slot): the instances the
    device derives from the start entries are those of the interval rule, so both ways of handing the segment over must
    give the same calls, bit for bit."""
    from popdel_amd import capi, synth
    n, region, first, W = 40, (1000000, 1090000), 1000020, 2900
    lib, rg, pairs = cohort(n, region)
    parts, offsets = [], [0]
    starts, devs = [], []
    for pos, dev in pairs:
        # (the loader files the records of the windows from the first window position on, in start order)
        keep = pos >= first
        pos, dev = pos[keep], dev[keep]
        d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
        parts.append((d, s, a, b))
        offsets.append(offsets[-1] + len(d))
        starts.append(pos)
        devs.append(dev)
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32), np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    hip = capi.load_hip()
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
    a = ctx.collect(ctx.submit(seg))
    ctx.close()
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
    assert hip.pd_walk_supported(ctx.handle) == 1 and hip.pd_walk_supported(None) == 1
    equal_run = [1 if len(s) and s[0] == first else 0 for s in starts]
    t = ctx.submit_walk(first, W, starts, devs, end_border=[0xffffffff] * n, switches=[2] * n, equal_run=equal_run)
    b = ctx.collect(t)
    ctx.close()
    assert a["n_calls"] > 10 and a["n_calls"] == b["n_calls"]
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k]), k


_ARENA = None


def arena():
    """16 MB of anonymous memory that lives as long as the process (what the library page-locks is never given back)."""
    global _ARENA
    if _ARENA is None:
        import mmap
        _ARENA = (mmap.mmap(-1, 16 << 20), )
        _ARENA = _ARENA + (np.frombuffer(_ARENA[0], np.uint32).reshape(-1, 2), )
    return _ARENA[1]


@pytest.mark.parametrize("share", ["all", "half"])
def test_entries_fetched_from_offered_host_regions(built, share):
    """pd_host_regions_offer(): start entries that lie in a range the host offered are fetched by the device from the tables
    themselves (a gather kernel over the link), the others go through the staging copy -- same calls, bit for bit, as with
    POPDEL_HIP_WALK_GATHER unset and nothing offered (the test above), and the counters say which way the entries went."""
    from popdel_amd import capi
    n, region, first, W = 40, (1000000, 1090000), 1000020, 2900
    lib, rg, pairs = cohort(n, region)
    starts, devs = [], []
    for pos, dev in pairs:
        keep = pos >= first
        starts.append(pos[keep])
        devs.append(dev[keep])
    equal_run = [1 if len(s) and s[0] == first else 0 for s in starts]
    hip = capi.load_hip()
    kw = dict(end_border=[0xffffffff] * n, switches=[2] * n, equal_run=equal_run)
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
    a = ctx.collect(ctx.submit_walk(first, W, starts, devs, **kw))
    assert ctx.walk_entry_counts()[0] == 0
    ctx.close()
    buf = arena()
    assert capi.offer_host_region(hip, buf) == 0 and capi.offer_host_region(hip, buf) == 0      # (offering again is harmless)
    inside = [share == "all" or k % 2 == 0 for k in range(n)]
    ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
    b = ctx.collect(ctx.submit_walk(first, W, starts, devs, table_arena=buf, in_arena=inside, **kw))
    gathered, staged, locked = ctx.walk_entry_counts()
    ctx.close()
    total = sum(len(s) for s in starts)
    want = sum(len(s) for s, i in zip(starts, inside) if i)
    assert gathered == want and staged == total - want and locked >= buf.nbytes
    assert a["n_calls"] > 10 and a["n_calls"] == b["n_calls"]
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            assert np.array_equal(v, b[k]), k


def test_state_from_segment_to_segment(built):
    """Two segments of one read group, handed over by hand: pairs filed one end slot ahead stay active into the second
    segment and leave at their time there; pairs whose end record sat in the first read slot are gone after the switch
    (quirk B1); start entries no window reached are dropped by the switch (B2) but kept without one."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    hip = capi.load_hip()
    ctx = capi.Context(hip, [rg], [0, 1])
    inner = rg.median - 2 * rg.read_length                      # end = start + max(0, dev + inner)
    F, n1 = 3000, 7                                             # windows at 3000, 3030, ..., 3180
    border = 3300
    #          start  dev            what
    entries = [(3000, 0),          # ends at 3000 + inner (3200 for 500/150): last window 3180 < border: read slot; active in all 7 windows
               (3010, 200),        # ends 3410: last window 3390 >= border: filed ahead, carried
               (3100, -inner),     # ends at its start: leaves at the first window beyond 3090 + 30
               (3290, 0)]          # no window reaches it: pending
    st = [np.array([e[0] for e in entries], np.uint32)]
    dv = [np.array([e[1] for e in entries], np.int32)]
    ctx.collect(ctx.submit_walk(F, n1, st, dv, end_border=[border], switches=[2], equal_run=[1]))
    _, a_start, _, a_id = ctx.walk_active()
    assert sorted(a_start.tolist()) == [3000, 3010] and sorted(a_id.tolist()) == [0, 1]
    # without a switch the pending entry is still a candidate and the read slot's pairs leave at their time
    F2 = 3210
    ctx.collect(ctx.submit_walk(F2, 8, [np.zeros(0, np.uint32)], [np.zeros(0, np.int32)], end_border=[border + 300], switches=[0],
                                equal_run=[0], step_pos=3210))
    _, a_start, _, _ = ctx.walk_active()
    assert sorted(a_start.tolist()) == [3010, 3290]             # 3000 left at window 1 (last window 3180), 3010 is filed ahead: stays; 3290 came in
    # a switch: the pair filed ahead moves into the read slot and leaves at its time; the other one's end record sat in the
    # drained slot -> gone
    ctx.collect(ctx.submit_walk(3540, 6, [np.zeros(0, np.uint32)], [np.zeros(0, np.int32)], end_border=[border + 600], switches=[1],
                                equal_run=[0], step_pos=3540))
    assert ctx.walk_active()[1].tolist() == []                  # last windows 3390 and 3480, both over before 3540: left at the step
    ctx.close()


def test_lists_grow_instead_of_failing(built, cohort_dir, tmp_path, monkeypatch):
    """A read group with more active (or not-yet-reached) pairs than the device-side window loop has room for -- a pile-up
    under an unlimited maxLoad (-A), the reference's parameter_calculation_popdel_call.h:129 -- makes the lists longer; it
    used to abort `popdel call` mid-run. Forced here by starting with room for 4 pairs per read group (POPDEL_HIP_WALK_CAP,
    a test lever): every segment of the golden cohort outgrows that several times, and the VCF must be the golden one."""
    import os
    from conftest import GOLDEN, POPDEL_BIN, run_call
    from popdel_amd import synth
    d = cohort_dir("s2", 2, synth.single_deletion)
    monkeypatch.setenv("POPDEL_HIP_WALK_CAP", "4")
    assert run_call(POPDEL_BIN, d, str(tmp_path / "a.vcf")) == open(os.path.join(GOLDEN, "s2_default.vcf")).read()
    assert run_call(POPDEL_BIN, d, str(tmp_path / "b.vcf"), "-n") == open(os.path.join(GOLDEN, "s2_windowwise.vcf")).read()
    d = cohort_dir("dd200", 200, synth.dense_deletions)
    import gzip
    assert run_call(POPDEL_BIN, d, str(tmp_path / "t.vcf")) == gzip.open(os.path.join(GOLDEN, "dd200_default.vcf.gz"), "rt").read()
