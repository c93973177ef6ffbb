"""GPU: a cohort split by samples over two contexts (two processes sharing the box's one GPU, collectives over gloo)
must give bit-for-bit what one context gives for the whole cohort -- integers AND floating point, because every
cross-sample reduction adds all-gathered per-tile values in cohort order. On a multi-GPU node the same code runs with
one GPU per process and the library's own RCCL all-gathers (bench.py --gpus N: the sample split is its N > 1 headline)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ORACLE_LIB, ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "shard_worker.py")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(case, world, out, extra_env=None):
    port = str(free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, WORKER, ROOT, case, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    return pickle.load(open(out, "rb"))


@pytest.mark.parametrize("case,world", [("deletions600", 2), ("thresholds300", 2), ("deletions600", 3), ("mixed520", 2),
                                        ("stress3", 3), ("stress5", 2), ("tile3000", 2)])
def test_sample_shards_equal_one_context(built, tmp_path, case, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shard_worker import make_case
    from popdel_amd import capi
    sharded = run_world(case, world, tmp_path / "sharded.pkl")
    seg, rgs, offs, opts = make_case(case)
    ctx = capi.Context(capi.load_hip(), rgs, offs, **opts)
    single = ctx.collect(ctx.submit(seg))
    ctx.close()
    assert single["n_calls"] > 20
    assert sharded["exchanges"] > 10
    got = sharded["calls"]
    assert got["n_calls"] == single["n_calls"] and got["n_samples"] == single["n_samples"]
    for k, v in single.items():
        assert np.array_equal(np.asarray(got[k]), np.asarray(v)), k        # floating point included: bit-identical
    if case.startswith("stress") or case == "tile3000":
        return          # random cohorts: the oracle comparison is the business of tests/test_gpu_fuzz.py (knife edges)
    # and the unsharded run is the one the other GPU tests hold against the oracle; one direct check here too
    octx = capi.Context(capi.load_library(ORACLE_LIB), rgs, offs, **opts)
    ref = octx.collect(octx.submit(seg))
    for k in ("deletion_length", "position", "end_position", "pl", "lad", "dad", "first_last", "filter"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.allclose(got["likelihood_ratio"], ref["likelihood_ratio"], rtol=1e-6, atol=1e-6)


def test_sample_shards_with_stage2_batches(built, tmp_path):
    """Sharded AND batched: the exchanges run per stage-2 batch, all contexts cut the cell list at the same places."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shard_worker import make_case
    from popdel_amd import capi
    budget = {"POPDEL_HIP_STAGE2_BYTES": "4000000"}
    sharded = run_world("deletions600", 2, tmp_path / "sharded.pkl", budget)
    seg, rgs, offs, opts = make_case("deletions600")
    ctx = capi.Context(capi.load_hip(), rgs, offs, **opts)
    single = ctx.collect(ctx.submit(seg))
    ctx.close()
    assert sharded["exchanges"] > 60          # several batches, each with its own round of exchanges
    for k, v in single.items():
        assert np.array_equal(np.asarray(sharded["calls"][k]), np.asarray(v)), k


@pytest.mark.parametrize("case,world,lanes", [("deletions600", 2, 2), ("stress3", 3, 3)])
def test_sample_shards_keep_their_lanes(built, tmp_path, case, world, lanes):
    """A cohort over two (three) ranks whose contexts keep two (three) lanes (pd_ctx_set_shard_lanes: one pair of callbacks and one process
    group per lane): two tickets per lane in flight, lane k here exchanging with lane k there from the library's own threads --
    every ticket's calls are bit for bit those of one plain context."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shard_worker import make_case
    from popdel_amd import capi
    sharded = run_world(case, world, tmp_path / "sharded.pkl", {"POPDEL_TEST_LANES": str(lanes)})
    seg, rgs, offs, opts = make_case(case)
    os.environ["POPDEL_HIP_LANES"] = "1"
    try:
        ctx = capi.Context(capi.load_hip(), rgs, offs, **opts)
    finally:
        os.environ.pop("POPDEL_HIP_LANES", None)
    single = ctx.collect(ctx.submit(seg))
    ctx.close()
    assert single["n_calls"] > 20 and len(sharded["calls"]) == 2 * lanes
    assert len(sharded["lane_exchanges"]) == lanes and min(sharded["lane_exchanges"]) > 20   # every lane exchanged: two tickets each
    for got in sharded["calls"]:
        assert got["n_calls"] == single["n_calls"] and got["n_samples"] == single["n_samples"]
        for k, v in single.items():
            assert np.array_equal(np.asarray(got[k]), np.asarray(v)), k
