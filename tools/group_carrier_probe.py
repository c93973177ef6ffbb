"""Two (or more) contexts of ONE process on one GPU sharing a cohort by samples (pd_ctx_set_shard_local), per pass over a resident
segment: the device-side carrier (exchange.hip) against the host-driven one. One GPU stands in for several: the kernels of both
contexts share the chip, so the figures say what an exchange costs the HOST and the STREAM, not what a rank of a cohort over
several GPUs gains. usage: group_carrier_probe.py [samples per context=1280] [contexts=2] [passes=120] -> one JSON line per carrier"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(per, world, passes):
    import numpy as np
    import torch
    import bench
    from popdel_amd import capi, synth
    n = per * world
    first, W, offsets, a, _ = bench.device_tile(n, 20260, torch.device("cuda", 0))
    rg = synth.conditioned_readgroup(synth.Library())
    hip = capi.load_hip()
    group = C.c_void_p()
    hip.pd_local_group_create.argtypes = [C.c_uint32, C.POINTER(C.c_void_p)]
    hip.pd_ctx_set_shard_local.argtypes = [C.c_void_p, C.POINTER(capi.PdShard), C.c_void_p]
    hip.pd_local_group_destroy.argtypes = [C.c_void_p]
    assert hip.pd_local_group_create(world, C.byref(group)) == 0
    table = np.full(n, rg.min_init_del_len, dtype=np.uint32)
    o = np.asarray(offsets, dtype=np.int64)
    ctxs, parts = [], []
    for r in range(world):
        s0, s1 = r * per, (r + 1) * per
        ctx = capi.Context(hip, [rg] * per, list(range(per + 1)))
        sh = capi.PdShard()
        sh.rank, sh.world, sh.first_sample, sh.n_samples_global, sh.n_readgroups_global = r, world, s0, n, n
        sh.min_init_del_len_global = table.ctypes.data_as(C.POINTER(C.c_uint32))
        assert hip.pd_ctx_set_shard_local(ctx.handle, C.byref(sh), group) == 0
        x, y = int(o[s0]), int(o[s1])
        parts.append(capi.Segment(first, W, o[s0:s1 + 1] - x, a["dev"][x:y], a["start"][x:y], a["on"][x:y], a["off"][x:y]))
        ctxs.append(ctx)
    out = [None] * world
    errors = []

    def work(r):
        try:
            c = ctxs[r]
            h = c.stage(parts[r])
            for _ in range(5):
                c.collect(c.run(h), copy=False)
            t0 = time.perf_counter()
            xch = 0.0
            t_prev = c.run(h)
            for _ in range(passes - 1):
                t_next = c.run(h)
                c.collect(t_prev, copy=False)
                xch += c.timing()["exchange_ms"]
                t_prev = t_next
            c.collect(t_prev, copy=False)
            tm = c.timing()
            xch += tm["exchange_ms"]
            out[r] = dict(ms_per_pass=(time.perf_counter() - t0) / passes * 1e3, exchange_ms_per_pass=xch / passes,
                          all_gathers_per_pass=tm["n_all_gathers"], candidate_cells=tm["n_candidate_cells"])
            c.unstage(h)
        except Exception as e:
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        print(json.dumps(dict(carrier=os.environ.get("POPDEL_HIP_GROUP_CARRIER"), error=errors[0][:300])))
        return
    print(json.dumps(dict(carrier=os.environ.get("POPDEL_HIP_GROUP_CARRIER"), samples_per_context=per, contexts=world, passes=passes,
                          ms_per_pass=max(x["ms_per_pass"] for x in out), exchange_ms_per_pass=max(x["exchange_ms_per_pass"] for x in out),
                          all_gathers_per_pass=out[0]["all_gathers_per_pass"], candidate_cells=out[0]["candidate_cells"])))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        per = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
        world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
        passes = int(sys.argv[3]) if len(sys.argv) > 3 else 120
        for carrier in ("flags", "host"):
            env = dict(os.environ, POPDEL_HIP_GROUP_CARRIER=carrier, GPU_MAX_HW_QUEUES="16", POPDEL_HIP_LANES="1")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(per), str(world), str(passes)], env=env,
                               capture_output=True, text=True, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(lines[-1] if lines else json.dumps(dict(carrier=carrier, error=(r.stderr or r.stdout)[-300:])), flush=True)
