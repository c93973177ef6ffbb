"""GPU box: a few serial passes over the candidate-dense segment of bench.py (10 planted deletions per 200 kb), for kernel
traces of stage 2 where it dominates:  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/dense_probe.py [samples] [passes]
then  python tools/timeline.py <kernel_trace.csv> 2  lists one pass launch by launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from popdel_amd import capi, synth
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
hip = capi.load_hip()
rg = synth.conditioned_readgroup(synth.Library())
f2, W2, o2, a2, pl2 = bench.device_tile(S, 777, torch.device("cuda", 0), 30.0, n_deletions=10)
ctx = capi.Context(hip, [rg] * S, list(range(S + 1)), device=0)
lane = bench.Lane(ctx, ctx.stage(capi.Segment(f2, W2, o2, a2["dev"], a2["start"], a2["on"], a2["off"])), serial=True)
lane.passes(2, bench.new_acc())
torch.cuda.synchronize()
acc = bench.new_acc()
t0 = time.perf_counter()
lane.passes(passes, acc)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("dense segment, %d samples: %.3f ms per pass (stage 1 %.3f, stage 2 %.3f), %d cells, %d calls per pass; first data-likelihood launch %.3f ms"
      % (S, el / passes * 1e3, acc[0] / passes, acc[1] / passes, acc[3] // passes, acc[2] // passes, acc[4]["lik_first_ms"]), flush=True)
