#!/usr/bin/env python3
"""inflateMembersKernel by members per wave (POPDEL_HIP_INFLATE_LANES = 32 | 16 | 8): kernel time on a product-sized chunk
(21,000 members: what one decoder thread hands over at 4,000 samples) and on a whole 10,000-sample segment (200,000 members),
each checked against zlib. One child process per shape (the shape is read once per process).
usage: inflate_lanes_sweep.py -> gpurun_out/inflate_lanes_sweep.json"""
import gzip
import json
import os
import subprocess
import sys
import zlib
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

SIZES = (21000, 200000)


def make(args):
    from popdel_amd import inflate
    seed, n = args
    return [gzip.compress(b, 6, mtime=0) for b in inflate.profile_blocks(n, seed=seed)]


def child(path):
    from popdel_amd import capi, inflate
    z = np.load(path)
    data, offs, out_offs, want = z["data"], z["offs"], z["out_offs"], z["want"]
    hip = capi.load_hip()
    inflate.inflate_gpu(hip, data[: int(offs[64])], offs[:65], out_offs[:65])
    rows = {}
    for n in SIZES:
        out, status, lengths, timing = inflate.inflate_gpu(hip, data[: int(offs[n])], offs[: n + 1], out_offs[: n + 1], repeat=5)
        if not os.environ.get("SWEEP_NOCHECK"):
            assert (status == 0).all()
            assert np.array_equal(out, want[: int(out_offs[n])])
        rows[str(n)] = dict(kernel_ms=timing["kernel_ms"], workgroups=timing["workgroups"], inflated_GBs=int(out_offs[n]) / timing["kernel_ms"] / 1e6)
    print(json.dumps(rows))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    from popdel_amd import inflate
    distinct, per = 20000, 500
    with ProcessPoolExecutor(max_workers=8) as ex:
        parts = list(ex.map(make, [(1000 + k, per) for k in range(distinct // per)]))
    blobs = [b for p in parts for b in p]
    blobs = (blobs * (-(-max(SIZES) // len(blobs))))[: max(SIZES)]
    data, offs, out_offs = inflate.pack_members(blobs)
    want = np.frombuffer(b"".join(zlib.decompress(b, 31) for b in blobs), np.uint8)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "_lanes_members.npz")
    np.savez(path, data=data, offs=offs, out_offs=out_offs, want=want)
    result = {}
    variants = [("", l) for l in (8, 4)]
    vdir = os.path.join(ROOT, "popdel_amd", "lib", "variants")          # experiment builds of the library, if any
    if os.path.isdir(vdir):
        variants += [(os.path.join(vdir, f), l) for f in sorted(os.listdir(vdir)) if f.endswith(".so") for l in (8, 4)]
    for lib, lanes in variants:
        env = dict(os.environ, POPDEL_HIP_INFLATE_LANES=str(lanes))
        if lib:
            env["POPDEL_HIP_LIBRARY"] = lib
            pass
        key = "%s%d" % (os.path.basename(lib)[len("libpopdel_hip_"):-3] + ":" if lib else "", lanes)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            result[key] = dict(error=r.stderr[-2000:])
        else:
            result[key] = json.loads(r.stdout.strip().splitlines()[-1])
        print(key, result[key], flush=True)
    os.remove(path)
    with open(os.path.join(ROOT, "gpurun_out", "inflate_lanes_sweep.json"), "w") as f:
        json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
