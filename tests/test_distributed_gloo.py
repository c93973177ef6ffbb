"""CPU: the N>1 path (segments dealt round-robin, call records gathered) with world_size 2 over gloo.
The compute backend in this CPU test is the oracle library; on GPUs bench.py uses the same sharding with the HIP library."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ORACLE_LIB, ROOT

WORKER = r'''
import os, sys, pickle
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch.distributed as dist
from popdel_amd import capi, synth, distributed
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = synth.Library()
rg = synth.conditioned_readgroup(lib)
n = 6
segs = []
for k in range(5):
    first = 1000020 + k * 30 * 700
    region = (first - 600, first + 30 * 700)
    segs.append(synth.cohort_segment(n, region, first, 700, lib,
                                     deletions=lambda j, r, k=k: [synth.Deletion(r[0] + 9000, 700 + 100 * k, [1, 2, 0][(j + k) % 3])]))
ctx = capi.Context(capi.load_library(sys.argv[2]), [rg] * n, list(range(n + 1)))
out = distributed.run_sharded(ctx, segs, rank, world)
assert len(distributed.shard_segments(len(segs), rank, world)) in (2, 3)
if rank == 0:
    pickle.dump(out, open(sys.argv[3], "wb"))
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_give_the_single_process_result(built, tmp_path):
    import pickle
    from popdel_amd import capi, synth, distributed
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = tmp_path / "out.pkl"
    port = str(free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(worker), ROOT, ORACLE_LIB, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    sharded = pickle.load(open(out, "rb"))

    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    n = 6
    segs = []
    for k in range(5):
        first = 1000020 + k * 30 * 700
        region = (first - 600, first + 30 * 700)
        segs.append(synth.cohort_segment(n, region, first, 700, lib,
                                         deletions=lambda j, r, k=k: [synth.Deletion(r[0] + 9000, 700 + 100 * k, [1, 2, 0][(j + k) % 3])]))
    ctx = capi.Context(capi.load_library(ORACLE_LIB), [rg] * n, list(range(n + 1)))
    single = distributed.run_sharded(ctx, segs)
    assert len(sharded) == len(single) == 5
    assert sum(s["n_calls"] for s in single) > 20
    for a, b in zip(sharded, single):
        assert a["n_calls"] == b["n_calls"]
        for key in b:
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key


EXCHANGE_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import torch.distributed as dist
from popdel_amd import capi, distributed
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
x = distributed.SampleShardExchange(rank, world, "cpu")
# the library's protocol: reserve, fill the own chunk of a region, all_gather(offset, bytes per rank)
for round_, (offset, per) in enumerate([(0, 64), (256, 1000), (4096, 8)]):
    base = x.reserve(offset + world * per + 17)
    assert base == x.arena.data_ptr()
    x.arena.zero_()
    mine = torch.arange(per, dtype=torch.int64).add(rank * 7 + round_).remainder(251).to(torch.uint8)
    x.arena[offset + rank * per: offset + (rank + 1) * per] = mine
    assert x.all_gather(offset, per, None) == 0
    for r in range(world):
        want = torch.arange(per, dtype=torch.int64).add(r * 7 + round_).remainder(251).to(torch.uint8)
        assert torch.equal(x.arena[offset + r * per: offset + (r + 1) * per], want), (round_, r)
    assert int(x.arena[:offset].sum()) == 0
assert x.calls == 3
# a buffer that is too small for the region is reported, not raised through the C caller
assert x.all_gather(x.arena.numel(), 64, None) == 1 and x.last_error is not None
dist.barrier()
dist.destroy_process_group()
'''


def test_sample_shard_exchange_over_gloo(built, tmp_path):
    """The host half of pd_shard (reserve / all_gather callbacks) with world_size 2 over gloo on CPU tensors; the device
    half runs in tests/test_gpu_sample_shard.py."""
    worker = tmp_path / "xworker.py"
    worker.write_text(EXCHANGE_WORKER)
    port = str(free_port())
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(worker), ROOT], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0


def test_sample_partition_rule(built):
    from popdel_amd import capi, distributed
    lib = capi.load_library(ORACLE_LIB)
    for n, world in [(1, 1), (255, 1), (256, 2), (257, 2), (600, 2), (10000, 8), (50000, 8), (65535, 7)]:
        ranges = distributed.sample_ranges(lib, n, world)
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c and a <= b
        assert all(a % 256 == 0 for a, b in ranges if a < b)
        sizes = [b - a for a, b in ranges if b > a]
        assert all(s == sizes[0] for s in sizes[:-1]) and sizes[-1] <= sizes[0]
    # the oracle is single-context: it accepts world 1 only
    from popdel_amd import synth
    rg = synth.conditioned_readgroup(synth.Library())
    ctx = capi.Context(lib, [rg] * 3, [0, 1, 2, 3])
    ctx.set_shard(0, 1, 0, 3, [0, 0, 0], lambda n: None, lambda o, p, s: 0)
    import pytest
    with pytest.raises(capi.BackendError):
        ctx.set_shard(0, 2, 0, 3, [0, 0, 0], lambda n: None, lambda o, p, s: 0)
