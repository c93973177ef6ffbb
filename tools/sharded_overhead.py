"""GPU box (one GPU): what the sample-sharded code path costs by itself. One rank, the bench's resident segment:
per-pass time of (a) an unsharded context, (b) a SampleShard of world 1 with the library's own RCCL exchange,
(c) the same through the host callbacks (torch.distributed, nccl). With one rank the collectives move nothing, so the
difference to (a) is the path's own overhead: host-side lists, exact pool sizes, ~20 exchange calls per pass.
usage: python tools/sharded_overhead.py [passes]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
import bench
from popdel_amd import capi, distributed, synth

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1280, 6400, 9984]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hip = capi.load_hip()
lib = synth.Library()
rg = synth.conditioned_readgroup(lib)
S = max(sizes)
first, W, offsets, arrays, planted = bench.device_tile(S, 20260, torch.device("cuda", 0), 30.0, deletion_seed=20260)


def measure(ctx, name):
    h = ctx.stage(seg)
    lane = bench.Lane(ctx, h)
    lane.passes(5, bench.new_acc())
    torch.cuda.synchronize()
    acc = bench.new_acc()
    t0 = time.perf_counter()
    lane.passes(passes, acc)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%6d samples  %-34s %.3f ms per pass (stage 1 %.3f, stage 2 %.3f by events; exchanges %.3f ms), %d exchanges per pass, %d calls per pass" % (
        S, name, el / passes * 1e3, acc[0] / passes, acc[1] / passes, acc[5] / passes, acc[4]["n_all_gathers"], acc[2] // passes), flush=True)
    ctx.unstage(h)


for S in sizes:
    o_, a_ = bench.take_samples(offsets, arrays, S)
    seg = capi.Segment(first, W, o_, a_["dev"], a_["start"], a_["on"], a_["off"])
    ctx = capi.Context(hip, [rg] * S, list(range(S + 1)), device=0)
    measure(ctx, "unsharded context")
    ctx.close()
    for native in (True, False):
        sh = distributed.SampleShard(hip, [rg] * S, list(range(S + 1)), 0, 1, device=0, native=native)
        measure(sh.ctx, "sharded, world 1, %s" % ("library RCCL" if native else "host callbacks"))
        sh.close()
dist.destroy_process_group()
