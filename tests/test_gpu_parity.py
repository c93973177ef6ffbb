"""GPU: the HIP library against the oracle through the C ABI, and the product binary against the golden fixtures."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, ORACLE_LIB, POPDEL_BIN, run_call

pytestmark = pytest.mark.gpu

INT_KEYS = ("initial_length", "iterations", "deletion_length", "window_position", "position", "end_position", "filter", "pl",
            "lad", "dad", "first_last")


def both(seg, rgs, offsets, hip_library=None, **kw):
    """hip_library: path of a TEST build of the HIP library (tests/test_gpu_fuzz.py); default: the product library."""
    from popdel_amd import capi
    out = []
    for lib in (capi.load_library(hip_library) if hip_library else capi.load_hip(), capi.load_library(ORACLE_LIB)):
        ctx = capi.Context(lib, rgs, offsets, stage1_view=True, **kw)
        calls = ctx.collect(ctx.submit(seg))
        q3, fl = ctx.stage1()        # (HIP: the full scan's values, checked cell by cell against the screen + exact pass that produced the calls)
        out.append((calls, q3, fl))
        ctx.close()
    return out


def assert_same(gpu, cpu):
    (gc, gq, gf), (oc, oq, of) = gpu, cpu
    assert np.array_equal(gf, of)
    assert np.array_equal(gq, oq)
    assert gc["n_calls"] == oc["n_calls"]
    for k in INT_KEYS:
        assert np.array_equal(gc[k], oc[k]), k
    # north_star tolerance on the likelihood-ratio score: 1e-6
    assert np.allclose(gc["likelihood_ratio"], oc["likelihood_ratio"], rtol=1e-6, atol=1e-6)
    # AF: a reference shift is int(weighted mean of deviations); for a read group with a single active pair that mean
    # is an exact integer up to the last bit of an x87 product, so FP64 can truncate to the neighbour.
    assert np.allclose(gc["frequency"], oc["frequency"], rtol=0, atol=1e-6)


def test_stage1_and_calls_single_deletion(built):
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 48
    seg = synth.cohort_segment(n, (1000000, 1150000), 1000020, 4900, lib,
                               deletions=lambda k, r: [synth.Deletion(1070000, 1000, [1, 2, 0][k % 3])])
    gpu, cpu = both(seg, [rg] * n, list(range(n + 1)))
    assert cpu[0]["n_calls"] > 20
    assert_same(gpu, cpu)


def test_dense_deletions_window_wise_and_somatic(built):
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 40
    seg = synth.cohort_segment(n, (1000000, 1200000), 1000020, 6600, lib, deletions=synth.dense_deletions)
    for kw in (dict(), dict(window_wise_output=True), dict(somatic=True), dict(iterations=3), dict(min_sample_fraction=1.0)):
        gpu, cpu = both(seg, [rg] * n, list(range(n + 1)), **kw)
        assert cpu[0]["n_calls"] > 100
        assert_same(gpu, cpu)


def test_ragged_inputs(built):
    """Empty samples, low coverage, a sample above max_load, multi-read-group samples, long-lived pairs."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    lowcov = synth.Library(coverage=1.0)
    rg = synth.conditioned_readgroup(lib)
    rg_tight = synth.conditioned_readgroup(lib, max_load=20)
    region, first, W = (1000000, 1090000), 1000020, 2900
    dels = lambda k, r: [synth.Deletion(1040000, 2500, [2, 1, 0][k % 3]), synth.Deletion(1060000, 400, [1, 2, 2][k % 3])]
    parts, offsets, rgs, sample_offsets = [], [0], [], [0]
    specs = [("normal", lib, rg)] * 5 + [("empty", None, rg), ("low", lowcov, rg), ("tight", lib, rg_tight),
                                          ("multi", lib, rg), ("multi", lib, rg)]
    seed = 4000
    for kind, l, r in specs:
        n_rg = 3 if kind == "multi" else 1
        for j in range(n_rg):
            seed += 1
            if l is None:
                d = s = a = b = np.zeros(0, dtype=np.int64)
            else:
                sub = synth.Library(coverage=l.coverage / n_rg)
                pos, dev = synth.sample_pairs(seed, region, sub, dels(seed, region))
                d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
            parts.append((d, s, a, b))
            offsets.append(offsets[-1] + len(d))
            rgs.append(r)
        sample_offsets.append(len(rgs))
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32),
                       np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    gpu, cpu = both(seg, rgs, sample_offsets)
    assert cpu[0]["n_calls"] > 10
    assert (cpu[2] & 1).any()            # some low-coverage cells exist
    assert_same(gpu, cpu)


def test_no_pairs_at_all(built):
    from popdel_amd import capi, synth
    rg = synth.conditioned_readgroup(synth.Library())
    z = np.zeros(0, dtype=np.uint32)
    seg = capi.Segment(1000020, 50, np.zeros(4, np.uint64), z.astype(np.int32), z, z, z)
    gpu, cpu = both(seg, [rg] * 3, [0, 1, 2, 3])
    assert gpu[0]["n_calls"] == 0
    assert_same(gpu, cpu)


def golden_text(name):
    path = os.path.join(GOLDEN, name)
    return gzip.open(path, "rt").read() if name.endswith(".gz") else open(path).read()


def test_product_binary_two_samples_golden(built, cohort_dir, tmp_path):
    from popdel_amd import synth
    d = cohort_dir("s2", 2, synth.single_deletion)
    assert run_call(POPDEL_BIN, d, str(tmp_path / "a.vcf")) == golden_text("s2_default.vcf")
    assert run_call(POPDEL_BIN, d, str(tmp_path / "b.vcf"), "-n") == golden_text("s2_windowwise.vcf")


def test_product_binary_dense_200_golden(built, cohort_dir, tmp_path):
    from popdel_amd import synth
    d = cohort_dir("dd200", 200, synth.dense_deletions)
    assert run_call(POPDEL_BIN, d, str(tmp_path / "t.vcf")) == golden_text("dd200_default.vcf.gz")
    text = run_call(POPDEL_BIN, d, str(tmp_path / "tn.vcf"), "-n")
    out = []
    for l in text.splitlines(True):
        if l.startswith("#"):
            out.append(l)
            continue
        cols = l.rstrip("\n").split("\t")
        out.append("\t".join(cols[:8]) + "\t" + hashlib.sha1("\t".join(cols[8:]).encode()).hexdigest() + "\n")
    assert "".join(out) == golden_text("dd200_windowwise.digest.tsv.gz")


def test_product_binary_200_samples_golden(built, cohort_dir, tmp_path):
    """The 200-sample default-mode capture of the reference, through the product binary (the oracle-backed driver
    reproduces it in tests/test_oracle_golden.py::test_200_samples_default), with the loop on the device and walked on the
    host; the run's own count of knife-edge truncations (pd_timing::n_near_integer_shifts, printed under POPDEL_TIMING)
    goes into the session's parity notes."""
    import re
    import subprocess
    from conftest import FUZZ_NOTES, strip_date
    from popdel_amd import synth
    d = cohort_dir("s200", 200, synth.single_deletion)
    assert run_call(POPDEL_BIN, d, str(tmp_path / "a.vcf")) == golden_text("s200_default.vcf")
    env = dict(os.environ, POPDEL_HOST_WALK="1", POPDEL_TIMING="1")
    r = subprocess.run([POPDEL_BIN, "call", os.path.join(d, "profiles.txt"), "-o", str(tmp_path / "b.vcf"), "-r", "chr20:1000000-2000000"],
                       env=env, capture_output=True, text=True, check=True)
    assert strip_date(open(tmp_path / "b.vcf").read()) == golden_text("s200_default.vcf")
    m = re.search(r"\[popdel timing\] parity: (\d+) reference-shift quotients .* in (\d+) candidate cells, (\d+) window calls", r.stderr)
    assert m, r.stderr[-400:]
    os.makedirs(FUZZ_NOTES, exist_ok=True)
    with open(os.path.join(FUZZ_NOTES, "parity_notes.txt"), "a") as fh:
        fh.write("popdel call, 200 samples x 1 Mb (golden s200_default.vcf, byte-identical): %s reference-shift quotients within 1e-9 of an "
                 "integer in %s candidate cells, %s window calls\n" % m.groups())


def test_product_binary_1000_samples_golden(built, cohort_dir, tmp_path):
    """BASELINE.md's largest survey run of the reference (1000 samples x 1 Mb, default mode): byte-identical VCF."""
    from popdel_amd import synth
    d = cohort_dir("s1000", 1000, synth.single_deletion)
    assert run_call(POPDEL_BIN, d, str(tmp_path / "a.vcf")) == golden_text("s1000_default.vcf")


def test_wide_q3_range_takes_the_sorting_path(built):
    """A 12-kb deletion spreads the per-sample Q3 values over more than 8192 integers: clustering by bitonic sort."""
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 30
    seg = synth.cohort_segment(n, (1000000, 1100000), 1000020, 3200, lib,
                               deletions=lambda k, r: [synth.Deletion(1040000, 12000, [1, 2, 0][k % 3])])
    gpu, cpu = both(seg, [rg] * n, list(range(n + 1)))
    assert cpu[1].max() - cpu[1].min() > 8192
    assert cpu[0]["n_calls"] > 20
    assert_same(gpu, cpu)


def test_two_tickets_pipelined_equal_sequential(built, monkeypatch):
    """A second ticket may be issued before the open one is collected (its scan then starts underneath the result copy
    of the open one); results must equal the one-at-a-time order, a third ticket and out-of-order collects are refused.
    (A plain context, one lane: tests/test_gpu_lanes.py has the same for contexts with lanes.)"""
    from popdel_amd import capi, synth
    monkeypatch.setenv("POPDEL_HIP_LANES", "1")
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 20
    segs = [synth.cohort_segment(n, (1000000 + k * 60000, 1000000 + (k + 1) * 60000), 1000020 + k * 60000, 1900, lib,
                                 deletions=lambda j, r, k=k: [synth.Deletion(r[0] + 20000, 600 + 300 * k, [1, 2, 0][(j + k) % 3])])
            for k in range(3)]
    ctx = capi.Context(capi.load_hip(), [rg] * n, list(range(n + 1)))
    handles = [ctx.stage(s) for s in segs]
    sequential = [ctx.collect(ctx.run(h)) for h in handles]
    assert all(r["n_calls"] > 5 for r in sequential)
    piped = []
    ticket = ctx.run(handles[0])
    for i in range(3):
        nxt = ctx.run(handles[i + 1]) if i + 1 < 3 else None
        if nxt is not None:
            with pytest.raises(capi.BackendError):
                ctx.run(handles[0])                      # a third outstanding ticket
            with pytest.raises(capi.BackendError):
                ctx.collect(nxt)                         # out of order
        piped.append(ctx.collect(ticket))
        ticket = nxt
    for a, b in zip(piped, sequential):
        for k, v in b.items():
            assert np.array_equal(np.asarray(a[k]), np.asarray(v)), k
    for h in handles:
        ctx.unstage(h)
    ctx.close()


def test_three_contexts_on_one_gpu_concurrently(built):
    """Several segments in flight on one GPU, one context and host thread each (what bench.py's concurrent_segments leg
    and INTEGRATION.md 3 recommend for throughput): every context must return exactly what a context alone returns."""
    import threading
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 300
    segs = [synth.cohort_segment(n, (1000000, 1100000), 1000020, 3300, lib, seed_base=1000 + 500 * k,
                                 deletions=lambda i, r, k=k: [synth.Deletion(1030000 + 7000 * k, 700 + 300 * k, [1, 2, 0][(i + k) % 3])])
            for k in range(3)]
    hip = capi.load_hip()
    alone = []
    for seg in segs:
        ctx = capi.Context(hip, [rg] * n, list(range(n + 1)))
        alone.append(ctx.collect(ctx.submit(seg)))
        ctx.close()
    assert all(a["n_calls"] > 10 for a in alone)
    ctxs = [capi.Context(hip, [rg] * n, list(range(n + 1))) for _ in segs]
    got = [[] for _ in segs]

    def work(k):
        for _ in range(6):
            got[k].append(ctxs[k].collect(ctxs[k].submit(segs[k])))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for c in ctxs:
        c.close()
    for k in range(3):
        for res in got[k]:
            for key, v in alone[k].items():
                assert np.array_equal(np.asarray(res[key]), np.asarray(v)), (k, key)


def test_dense_and_generic_tiles_mixed(built):
    """520 samples = 3 tiles of 256: the first is served by the dense lane-per-sample kernels and holds a read group above
    max_load as sample 0 (its triple drives quirk Q2), the second holds an 85x sample (more active pairs than the dense
    image keeps -> generic kernels for that tile), the third a 3-read-group sample (generic as well)."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    rg_tight = synth.conditioned_readgroup(lib, max_load=20)
    region, first, W = (1000000, 1046000), 1000020, 1450
    dels = lambda k, r: [synth.Deletion(1015000, 1200, [2, 1, 0, 0][k % 4]), synth.Deletion(1031000, 500, [0, 1, 2][k % 3])]
    parts, offsets, rgs, sample_offsets = [], [0], [], [0]
    for k in range(520):
        kind = "tight" if k == 0 else ("deep" if k == 300 else ("multi" if k == 515 else "normal"))
        n_rg = 3 if kind == "multi" else 1
        for j in range(n_rg):
            cov = 85.0 if kind == "deep" else 30.0 / n_rg
            pos, dev = synth.sample_pairs(7000 + 3 * k + j, region, synth.Library(coverage=cov), dels(k, region))
            d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
            parts.append((d, s, a, b))
            offsets.append(offsets[-1] + len(d))
            rgs.append(rg_tight if kind == "tight" else rg)
        sample_offsets.append(len(rgs))
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32),
                       np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    gpu, cpu = both(seg, rgs, sample_offsets)
    assert cpu[0]["n_calls"] > 30
    assert_same(gpu, cpu)


def test_stage2_in_batches_equals_one_batch(built):
    """A small byte budget cuts the cell list of a candidate-dense segment into several stage-2 batches (what a large
    cohort needs when a segment is rich in candidates); the calls must not change."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 40
    seg = synth.cohort_segment(n, (1000000, 1200000), 1000020, 6600, lib, deletions=synth.dense_deletions)
    ctx = capi.Context(capi.load_hip(), [rg] * n, list(range(n + 1)), window_wise_output=True)
    one = ctx.collect(ctx.submit(seg))
    ctx.close()
    assert one["n_calls"] > 200
    # the budget is read from the environment at collect time
    os.environ["POPDEL_HIP_STAGE2_BYTES"] = "6500000"                   # room for about 40 cells per batch -> 13 batches
    try:
        ctx = capi.Context(capi.load_hip(), [rg] * n, list(range(n + 1)), window_wise_output=True)
        many = ctx.collect(ctx.submit(seg))
        ctx.close()
    finally:
        del os.environ["POPDEL_HIP_STAGE2_BYTES"]
    for k, v in one.items():
        assert np.array_equal(np.asarray(many[k]), np.asarray(v)), k


def test_many_candidate_lengths_in_one_window(built):
    """Twelve deletions of different lengths over the same windows, each carried by a few samples: a dozen Q3 clusters,
    hence a dozen (window, candidate length) cells per window."""
    from popdel_amd import synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 48
    seg = synth.cohort_segment(n, (1000000, 1060000), 1000020, 1900, lib,
                               deletions=lambda k, r: [synth.Deletion(1020000, 400 + 330 * (k % 12), 2 if k % 24 < 12 else 1)])
    gpu, cpu = both(seg, [rg] * n, list(range(n + 1)), window_wise_output=True)
    per_window = np.unique(cpu[0]["window_position"], return_counts=True)[1]
    assert per_window.max() > 8
    assert_same(gpu, cpu)


def test_deep_coverage_takes_the_128_candidate_scan(built):
    """45x to 85x samples next to 30x ones: more than 64 candidate chunks per cell -> the second scan kernel with networks
    of up to 128 inputs (and, past 64 ACTIVE pairs, the generic stage-2 kernels); above ~100 active pairs the read group
    is over its max_load in places, which the reference skips."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    region, first, W = (1000000, 1050000), 1000020, 1600
    dels = lambda k, r: [synth.Deletion(1020000, 1500, [1, 2, 0][k % 3])]
    covs = [30.0, 45.0, 60.0, 75.0, 85.0, 30.0, 52.0, 68.0]
    parts, offsets = [], [0]
    for k, cov in enumerate(covs * 3):
        pos, dev = synth.sample_pairs(9100 + k, region, synth.Library(coverage=cov), dels(k, region))
        d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
        parts.append((d, s, a, b))
        offsets.append(offsets[-1] + len(d))
    n = len(parts)
    seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                       np.concatenate([p[1] for p in parts]).astype(np.uint32),
                       np.concatenate([p[2] for p in parts]).astype(np.uint32),
                       np.concatenate([p[3] for p in parts]).astype(np.uint32))
    gpu, cpu = both(seg, [rg] * n, list(range(n + 1)))
    assert cpu[0]["n_calls"] > 20
    assert_same(gpu, cpu)


def test_device_log_tables_against_the_x87_logarithms(built):
    """Row f4: the lookup tables the kernels read -- conditioned densities with their natural and decadic logarithms,
    computed once on the device -- against what the reference evaluates per pair (logl / log10l of the same densities,
    here through the oracle): the densities bit for bit, the logarithms within one unit in the last place of a double."""
    from popdel_amd import capi, synth
    libs = [synth.Library(), synth.Library(median=380, stddev=31.0, read_len=100, hist_seed=5),
            synth.Library(median=610, stddev=74.0, read_len=125, hist_seed=9)]
    rgs = [synth.conditioned_readgroup(l) for l in libs]
    out = []
    for lib in (capi.load_hip(), capi.load_library(ORACLE_LIB)):
        ctx = capi.Context(lib, rgs, list(range(len(rgs) + 1)))
        out.append([ctx.log_tables(i, len(r.hist_values)) for i, r in enumerate(rgs)])
        ctx.close()
    for (gv, gl, g10), (ov, ol, o10), r in zip(out[0], out[1], rgs):
        assert len(gv) == len(r.hist_values) and np.array_equal(gv, ov) and np.array_equal(gv, np.asarray(r.hist_values))
        assert np.all(np.abs(gl - ol) <= np.spacing(np.abs(ol))) and np.all(np.abs(g10 - o10) <= np.spacing(np.abs(o10)))
        assert (gl == ol).mean() > 0.5


def test_stage2_launched_ahead_of_the_pool_sizes(built):
    """Stage 2 does not wait for the host to size the support pools: they are sized from the segment's average with
    headroom and checked on the device; a batch whose pools do not hold (here: all pairs of the segment sit around the
    deletion, the rest of the windows is empty), or that needs the generic kernels for the first time, is run again the
    exact way. Either way the calls are the oracle's."""
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n, region, first, W = 128, (1000000, 1060000), 1000020, 1900
    ordinary = synth.cohort_segment(n, region, first, W, lib)

    def around_the_deletion(k):
        pos, dev = synth.sample_pairs(1000 + k, region, lib, synth.single_deletion(k, region))
        d = synth.single_deletion(k, region)[0]
        near = (pos > d.start - 1500) & (pos < d.start + d.length + 1500)
        return synth.instances_by_interval_rule(pos[near], dev[near], lib, first, W)
    parts = [around_the_deletion(k) for k in range(n)]
    offsets = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.uint64)
    concentrated = capi.Segment(first, W, offsets, *[np.concatenate([p[i] for p in parts]) for i in range(4)])

    oracle = capi.Context(capi.load_library(ORACLE_LIB), [rg] * n, list(range(n + 1)))
    for order, expect in (((ordinary, ordinary), [0, 0]), ((concentrated, ordinary), [1, 1])):
        ctx = capi.Context(capi.load_hip(), [rg] * n, list(range(n + 1)))     # (the pools of a context only grow)
        repeated = []
        for seg in order:
            got = ctx.collect(ctx.submit(seg))
            repeated.append(ctx.timing()["n_repeated_batches"])
            want = oracle.collect(oracle.submit(seg))
            assert want["n_calls"] > 0
            assert got["n_calls"] == want["n_calls"]
            for k in INT_KEYS:
                assert np.array_equal(got[k], want[k]), k
            assert np.allclose(got["likelihood_ratio"], want["likelihood_ratio"], rtol=1e-6, atol=1e-6)
        ctx.close()
        assert repeated == expect, repeated
    oracle.close()


@pytest.mark.parametrize("n_windows", [1900, 6600])
def test_flagged_windows_and_cells_listed_on_the_device(built, n_windows):
    """An unsharded context forms the lists of flagged windows and of (window, length) cells on the device and reads them
    in one round trip. With a threshold of 1 nearly every window is flagged: 1900 windows -> more cells than the first copy
    brings along (second fetch); 6600 windows -> more flagged windows than the device lists are made for (4096: the
    host-side lists take over). Both against the host-side lists and the oracle."""
    import dataclasses
    from popdel_amd import capi, synth
    lib = synth.Library()
    rg = dataclasses.replace(synth.conditioned_readgroup(lib), min_init_del_len=1)
    n = 5
    seg = synth.cohort_segment(n, (1000000, 1000000 + 30 * n_windows + 2000), 1000020, n_windows, lib)
    gpu, cpu = both(seg, [rg] * n, list(range(n + 1)), window_wise_output=True)
    assert cpu[0]["n_calls"] > 0
    assert_same(gpu, cpu)
    os.environ["POPDEL_HIP_HOST_LISTS"] = "1"
    try:
        ctx = capi.Context(capi.load_hip(), [rg] * n, list(range(n + 1)), window_wise_output=True)
        host_lists = ctx.collect(ctx.submit(seg))
        assert ctx.timing()["n_candidate_cells"] > (4096 if n_windows > 4096 else 1022)      # (at least one per flagged window)
        ctx.close()
    finally:
        del os.environ["POPDEL_HIP_HOST_LISTS"]
    for k, v in host_lists.items():
        assert np.array_equal(np.asarray(gpu[0][k]), np.asarray(v)), k
