#!/usr/bin/env python3
"""Gunzip of profile blocks: the device (pd_inflate_members, one lane per member) against the loader's own host path
(libdeflate / zlib on all usable cores) on the member population of one 200-kb segment of a cohort.
usage: inflate_probe.py [members=200000] [distinct=20000] -> gpurun_out/inflate_probe.json"""
import gzip
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np


def make(args):
    from popdel_amd import inflate
    seed, n = args
    return [gzip.compress(b, 6, mtime=0) for b in inflate.profile_blocks(n, seed=seed)]


def main():
    import bench
    from popdel_amd import capi, inflate
    n_members = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    cores = bench.usable_cores()
    t0 = time.perf_counter()
    per = 500
    with ProcessPoolExecutor(max_workers=min(cores, 16)) as ex:
        parts = list(ex.map(make, [(1000 + k, per) for k in range(distinct // per)]))
    blobs = [b for p in parts for b in p]
    t_make = time.perf_counter() - t0
    reps = -(-n_members // len(blobs))
    blobs = (blobs * reps)[:n_members]
    data, offs, out_offs = inflate.pack_members(blobs)
    hip = capi.load_hip()
    inflate.inflate_gpu(hip, data[: int(offs[64])], offs[:65], out_offs[:65])          # runtime start, code object
    out, status, lengths, timing = inflate.inflate_gpu(hip, data, offs, out_offs, repeat=5)
    assert (status == 0).all(), np.unique(status)
    out_c, secs, which, failures = inflate.inflate_cpu(data, offs, out_offs, cores)
    assert failures == 0 and np.array_equal(out, out_c)
    _, secs1, _, _ = inflate.inflate_cpu(data[: int(offs[n_members // 16])], offs[: n_members // 16 + 1], out_offs[: n_members // 16 + 1], 1)
    pairs = (int(out_offs[-1]) - 12 * 40 * n_members) / 5.0
    row = dict(members=n_members, distinct_members=distinct, compressed_bytes=int(offs[-1]), inflated_bytes=int(out_offs[-1]),
               gpu=timing, gpu_inflated_GBs=int(out_offs[-1]) / timing["kernel_ms"] / 1e6,
               cpu=dict(library=which, threads=cores, seconds=secs, inflated_GBs=int(out_offs[-1]) / secs / 1e9,
                        one_thread_seconds_per_sixteenth=secs1, one_thread_MBs=int(out_offs[n_members // 16]) / secs1 / 1e6),
               kernel_vs_cpu=secs * 1e3 / timing["kernel_ms"], device_path_ms=timing["h2d_ms"] + timing["kernel_ms"],
               identical_to_cpu=True, generated_in_seconds=t_make, approx_pairs=pairs)
    print(json.dumps(row))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "inflate_probe.json"), "w") as f:
        json.dump(row, f, indent=1)


if __name__ == "__main__":
    main()
