"""GPU (one device): several contexts of ONE process sharing a cohort by samples, the library exchanging between them itself
(pd_ctx_set_shard_local) -- what `popdel call` does with POPDEL_DEVICES. Two carriers: the device-side one of round 5
(exchange.hip: one kernel per exchange acknowledges, pushes the context's chunk into the peers' buffers through peer-mapped
pointers and waits for their flags; no host in the data path; asked for by name, POPDEL_HIP_GROUP_CARRIER=flags) and the
host-driven one that is the default (peer copies bracketed by barriers of the host threads). On a one-GPU box all contexts sit
on device 0 (tests/conftest.py raises GPU_MAX_HW_QUEUES so that their streams do not share a queue); on a box with several GPUs
context r sits on GPU r and the same code writes over xGMI -- the run the device-side carrier has to pass before it may
become the default (it never ran on such a box: none was available to this project)."""
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import GOLDEN, POPDEL_BIN, run_call, strip_date

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("carrier", ["flags", "host"])
def test_two_contexts_in_one_process_equal_one_context(built, monkeypatch, capfd, carrier):
    import ctypes as C
    monkeypatch.setenv("POPDEL_HIP_GROUP_CARRIER", carrier)
    monkeypatch.setenv("POPDEL_HIP_GROUP_TRACE", "1")
    from popdel_amd import capi
    from test_gpu_fuzz import random_case
    seg, rgs, offs, opts = random_case(13)                  # 300 samples? the cohort must fill the contexts with whole tiles
    for seed in range(13, 60):
        seg, rgs, offs, opts = random_case(seed)
        if len(offs) - 1 >= 257 and seed % 3 != 2:
            break
    n = len(offs) - 1
    hip = capi.load_hip()
    one = capi.Context(hip, rgs, offs, **opts)
    want = one.collect(one.submit(seg))
    one.close()
    assert want["n_calls"] > 0
    world = 2
    import torch
    n_gpus = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    per = hip.pd_shard_samples(n, world)
    group = C.c_void_p()
    hip.pd_local_group_create.argtypes = [C.c_uint32, C.POINTER(C.c_void_p)]
    hip.pd_ctx_set_shard_local.argtypes = [C.c_void_p, C.POINTER(capi.PdShard), C.c_void_p]
    hip.pd_local_group_destroy.argtypes = [C.c_void_p]
    assert hip.pd_local_group_create(world, C.byref(group)) == 0
    o = np.asarray(seg.rg_offsets, dtype=np.int64)
    table = np.ascontiguousarray([r.min_init_del_len for r in rgs], dtype=np.uint32)
    ctxs, parts = [], []
    for r in range(world):
        s0, s1 = min(n, r * per), min(n, (r + 1) * per)
        ctx = capi.Context(hip, rgs[s0:s1], list(range(s1 - s0 + 1)), device=(r if n_gpus >= world else 0), **opts)
        sh = capi.PdShard()
        sh.rank, sh.world, sh.first_sample, sh.n_samples_global, sh.n_readgroups_global = r, world, s0, n, n
        sh.min_init_del_len_global = table.ctypes.data_as(C.POINTER(C.c_uint32))
        assert hip.pd_ctx_set_shard_local(ctx.handle, C.byref(sh), group) == 0, hip.pd_last_error(ctx.handle)
        a, b = int(o[s0]), int(o[s1])
        parts.append(capi.Segment(seg.first_window_pos, seg.n_windows, o[s0:s1 + 1] - a, seg.dev[a:b], seg.start[a:b], seg.on[a:b],
                                  seg.off[a:b]))
        ctxs.append(ctx)
    got = [None] * world
    errors = []

    def work(r):
        try:
            for _ in range(2):                               # twice: the exchange buffers are reused
                got[r] = ctxs[r].collect(ctxs[r].submit(parts[r]))
        except Exception as e:
            errors.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(c.timing()["n_all_gathers"] >= 9 for c in ctxs)
    for c in ctxs:
        c.close()
    said = capfd.readouterr().err
    assert ("device-side (flags) carrier" if carrier == "flags" else "host-driven carrier") in said
    hip.pd_local_group_destroy(group)           # (frees what the flags carrier parked while two contexts shared a GPU)
    for k in ("initial_length", "iterations", "deletion_length", "likelihood_ratio", "frequency", "window_position", "position",
              "end_position", "filter"):
        for r in range(world):
            assert np.array_equal(got[r][k], want[k]), k      # floating point included: sums are added in cohort tile order
    for k in ("pl", "lad", "dad", "first_last"):
        assert np.array_equal(np.concatenate([g[k] for g in got], axis=1), want[k]), k


def test_popdel_call_on_two_contexts_writes_the_golden_vcf(built, cohort_dir, tmp_path, monkeypatch):
    """The product binary with POPDEL_DEVICES=0,0: the 1,000-sample cohort sharded over two contexts (512 + 488 samples),
    byte-identical to the reference's VCF."""
    from popdel_amd import synth
    d = cohort_dir("s1000", 1000, synth.single_deletion)
    monkeypatch.setenv("POPDEL_DEVICES", "0,0")
    got = run_call(POPDEL_BIN, d, str(tmp_path / "two.vcf"))
    assert got == strip_date(open(os.path.join(GOLDEN, "s1000_default.vcf")).read())


def test_popdel_call_refuses_more_contexts_than_tiles(built, cohort_dir, tmp_path, monkeypatch):
    from popdel_amd import synth
    d = cohort_dir("dd200", 200, synth.dense_deletions)
    env = dict(os.environ, POPDEL_DEVICES="0,0")
    out = subprocess.run([POPDEL_BIN, "call", os.path.join(d, "profiles.txt"), "-o", str(tmp_path / "x.vcf"), "-r", "chr20:1000000-2000000"],
                         capture_output=True, text=True, env=env)
    assert out.returncode != 0 and "whole tiles of 256 samples" in (out.stderr + out.stdout)


def test_merging_on_the_device_equals_merging_on_the_host(built, cohort_dir, tmp_path, monkeypatch):
    """Default (merged) output: the per-sample half of unifyCalls runs on the device (pd_calls_merge) and only merged
    columns cross PCIe; POPDEL_HOST_MERGE=1 fetches every window call's columns and merges on the host threads as in
    round 1. Same bytes, on one context and on two; and both equal the reference's VCF (golden)."""
    from popdel_amd import synth
    d = cohort_dir("s1000", 1000, synth.single_deletion)
    golden = strip_date(open(os.path.join(GOLDEN, "s1000_default.vcf")).read())
    for devices in ("0", "0,0"):
        monkeypatch.setenv("POPDEL_DEVICES", devices)
        monkeypatch.delenv("POPDEL_HOST_MERGE", raising=False)
        on_device = run_call(POPDEL_BIN, d, str(tmp_path / "dev.vcf"))
        monkeypatch.setenv("POPDEL_HOST_MERGE", "1")
        on_host = run_call(POPDEL_BIN, d, str(tmp_path / "host.vcf"))
        assert on_device == on_host == golden
