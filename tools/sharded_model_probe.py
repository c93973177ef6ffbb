#!/usr/bin/env python3
"""What the sample-sharded path will cost on several GPUs, modelled on one: the sharded code path with a communicator of
ONE rank (library RCCL, every exchange issued) and, behind every exchange, a one-wave kernel that holds the stream for the
wire time an xGMI all-gather would add (POPDEL_HIP_XCH_DELAY_US) -- against the unsharded context on the same samples.
Rows: samples per rank x injected delay x lanes of the context (POPDEL_HIP_LANES: other segments' stage 2 fills the waits).
usage: sharded_model_probe.py [samples=1280,6400,10000] [delays=0,20,40] [lanes=1,2,3] [passes=160]
-> gpurun_out/sharded_model_probe.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from popdel_amd import capi, distributed, synth


def measure(make_ctx, seg, n_pass, depth):
    ctx, closer = make_ctx()
    lane = bench.Lane(ctx, ctx.stage(seg), depth=depth)
    lane.passes(8, bench.new_acc())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = bench.new_acc()
    lane.passes(n_pass, acc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.unstage(lane.handle)
    closer()
    return dt / n_pass * 1e3, acc


def main():
    samples = [int(x) if int(x) > 0 else 10000 for x in (sys.argv[1] if len(sys.argv) > 1 else "1280,6400,10000").split(",")]
    delays = [float(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,20,40").split(",")]
    lanes = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")]
    n_pass = int(sys.argv[4]) if len(sys.argv) > 4 else 160
    device = torch.device("cuda", 0)
    hip = capi.load_hip()
    rg = synth.conditioned_readgroup(synth.Library())
    S = max(samples)
    first, W, offsets, arrays, _ = bench.device_tile(S, 20260, device, 30.0)
    rows = []
    for n in samples:
        o, a = bench.take_samples(offsets, arrays, n)
        seg = capi.Segment(first, W, o, a["dev"], a["start"], a["on"], a["off"])
        os.environ["POPDEL_HIP_LANES"] = "1"
        os.environ.pop("POPDEL_HIP_XCH_DELAY_US", None)

        def plain():
            c = capi.Context(hip, [rg] * n, list(range(n + 1)), device=0)
            return c, c.close

        base_ms, _ = measure(plain, seg, n_pass, 2)
        for k in lanes:
            os.environ["POPDEL_HIP_LANES"] = str(k)
            if k > 1:
                lanes_ms, _ = measure(plain, seg, n_pass, 2 * k)
            else:
                lanes_ms = base_ms
            for d in delays:
                if d > 0:
                    os.environ["POPDEL_HIP_XCH_DELAY_US"] = str(d)
                else:
                    os.environ.pop("POPDEL_HIP_XCH_DELAY_US", None)

                def sharded():
                    sh = distributed.SampleShard(hip, [rg] * n, list(range(n + 1)), 0, 1, device=0, native=True)
                    return sh.ctx, sh.close

                ms, acc = measure(sharded, seg, n_pass, 2 * k if k > 1 else 2)
                row = dict(samples_per_rank=n, lanes=k, delay_us_per_exchange=d, unsharded_one_lane_ms_per_pass=round(base_ms, 4),
                           unsharded_same_lanes_ms_per_pass=round(lanes_ms, 4), modelled_ms_per_pass=round(ms, 4),
                           modelled_efficiency=round(base_ms / ms, 4), modelled_efficiency_vs_same_lanes=round(lanes_ms / ms, 4),
                           all_gathers_per_pass=acc[4]["n_all_gathers"], all_gather_bytes_per_pass=acc[4]["all_gather_bytes"],
                           exchange_ms_per_pass=round(acc[5] / n_pass, 4))
                rows.append(row)
                print(json.dumps(row), flush=True)
    os.environ.pop("POPDEL_HIP_XCH_DELAY_US", None)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "sharded_model_probe.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
