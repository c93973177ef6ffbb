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
