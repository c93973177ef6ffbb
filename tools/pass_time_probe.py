"""GPU box: per-pass time of the bench's resident segment for a cohort size and a process-group flavour:
python tools/pass_time_probe.py none|gloo|nccl|nccl_lazy|nccl_used [samples]
(how the dependence of stage 2 on the number of 256-sample tiles modulo 8 was found: DESIGN.md 4, XCD-stable tile order)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29542")
import torch
import bench
from popdel_amd import capi, synth
mode = sys.argv[1]
torch.cuda.set_device(0)
if mode != "none":
    import torch.distributed as dist
    if mode == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    elif mode == "nccl_lazy":
        dist.init_process_group("nccl", rank=0, world_size=1)
    elif mode == "nccl_used":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.zeros(1, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
hip = capi.load_hip()
lib = synth.Library()
rg = synth.conditioned_readgroup(lib)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 9984
first, W, offsets, arrays, planted = bench.device_tile(S, 20260, torch.device("cuda", 0), 30.0, deletion_seed=20260)
seg = capi.Segment(first, W, offsets, arrays["dev"], arrays["start"], arrays["on"], arrays["off"])
ctx = capi.Context(hip, [rg] * S, list(range(S + 1)), device=0)
h = ctx.stage(seg)
lane = bench.Lane(ctx, h)
lane.passes(5, bench.new_acc())
torch.cuda.synchronize()
for rep in range(2):
    acc = bench.new_acc()
    t0 = time.perf_counter()
    lane.passes(200, acc)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%-10s S=%d %.3f ms per pass (stage 1 %.3f, stage 2 %.3f) cells %d calls %d" % (mode, S, el / 200 * 1e3, acc[0] / 200, acc[1] / 200, acc[3] // 200, acc[2] // 200), flush=True)
