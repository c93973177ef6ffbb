#!/usr/bin/env python3
"""A/B of POPDEL_HIP_SCAN_OVERLAP (the waiting ticket's scan beside the open ticket's stage 2) on the bench workload.

For every mode: a fresh context, the same resident segments, pipelined passes (the bench's Lane), ms per pass, and the
calls of one pass per segment compared with the first mode's (must be identical: the scan's results do not depend on how
its launch shares the chip). usage: overlap_probe.py [samples] [passes] [modes,comma,separated] [n_segments]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from popdel_amd import capi, synth


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    n_pass = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    modes = (sys.argv[3] if len(sys.argv) > 3 else "0,s,l6,l5,l4,p6,p5,p4,p3").split(",")
    n_seg = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    device = torch.device("cuda", 0)
    hip = capi.load_hip()
    rg = synth.conditioned_readgroup(synth.Library())
    segs = []
    for k in range(n_seg):
        n_del = 0 if (k == n_seg - 1 and n_seg > 1) else 1
        f, W, o, a, p = bench.device_tile(S, 20260 + 1000 * k, device, 30.0, n_deletions=n_del)
        segs.append(capi.Segment(f, W, o, a["dev"], a["start"], a["on"], a["off"]))
    ref = None
    rows = []
    for mode in modes:
        # "L<k>": a context with k lanes (POPDEL_HIP_LANES); everything else: POPDEL_HIP_SCAN_OVERLAP
        lanes = int(mode[1:]) if mode.startswith("L") else 1
        os.environ["POPDEL_HIP_LANES"] = str(lanes)
        os.environ["POPDEL_HIP_SCAN_OVERLAP"] = "0" if mode.startswith("L") else mode
        ctx = capi.Context(hip, [rg] * S, list(range(S + 1)), device=0)
        hs = [ctx.stage(s) for s in segs]
        got = []
        cells = []
        for h in hs:
            got.append(ctx.collect(ctx.run(h)))
            cells.append(ctx.timing()["n_candidate_cells"])
        if ref is None:
            ref = got
        same = all(g["n_calls"] == r["n_calls"] and all(np.array_equal(g[k], r[k]) for k in g if isinstance(g[k], np.ndarray))
                   for g, r in zip(got, ref))
        row = dict(mode=mode, identical_to_first=bool(same), cells=cells)
        for label, handles in (("one_segment", hs[:1]), ("all_segments", hs)):
            lane = bench.Lane(ctx, handles, depth=(2 * lanes if lanes > 1 else 2))
            lane.passes(max(10, len(handles) * 2), bench.new_acc())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            acc = bench.new_acc()
            lane.passes(n_pass, acc)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            row[label] = dict(ms_per_pass=round(dt / n_pass * 1e3, 4), stage1_ms=round(acc[0] / n_pass, 4), stage2_ms=round(acc[1] / n_pass, 4))
        for h in hs:
            ctx.unstage(h)
        ctx.close()
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "overlap_probe.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
