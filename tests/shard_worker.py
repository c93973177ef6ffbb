"""Worker of the sample-shard tests: rank RANK of WORLD_SIZE processes (gloo rendezvous on 127.0.0.1) runs its block of
samples of the cohort described by `make_case()` through the HIP library; rank 0 pickles the joined result.
usage: shard_worker.py <repo root> <case> <out.pkl> [device]"""
import os
import pickle
import sys

import numpy as np


def capi_segment(first, W, offsets, arrays):
    from popdel_amd import capi
    return capi.Segment(first, W, offsets, arrays["dev"], arrays["start"], arrays["on"], arrays["off"])


def make_case(name):
    """(segment of the whole cohort, read groups, sample offsets, context options)"""
    from popdel_amd import synth
    lib = synth.Library()
    base = synth.conditioned_readgroup(lib)
    if name == "deletions600":
        n = 600
        seg = synth.cohort_segment(n, (1000000, 1036000), 1000020, 1100, lib,
                                   deletions=lambda k, r: [synth.Deletion(1010000, 900, [1, 2, 0, 0][k % 4]),
                                                           synth.Deletion(1024000, 2500, [0, 0, 1, 2, 0][k % 5])])
        rgs = [base] * n
        opts = {}
    elif name == "thresholds300":
        # per-sample thresholds differ: the rank-indexed threshold lookup (quirk Q1) must see the cohort-wide table
        import dataclasses
        n = 300
        seg = synth.cohort_segment(n, (1000000, 1030000), 1000020, 900, lib,
                                   deletions=lambda k, r: [synth.Deletion(1012000, 700 + 40 * (k % 3), [1, 2, 0][k % 3])])
        rgs = [dataclasses.replace(base, min_init_del_len=[0, 350, 900, 120][(k // 7) % 4]) for k in range(n)]
        opts = dict(window_wise_output=True, min_sample_fraction=0.5)
    elif name == "mixed520":
        # dense and generic stage-2 tiles, a read group above max_load as cohort sample 0, a 3-read-group sample on the
        # second rank (same cohort as tests/test_gpu_parity.py::test_dense_and_generic_tiles_mixed)
        from popdel_amd import capi
        rg_tight = synth.conditioned_readgroup(lib, max_load=20)
        region, first, W = (1000000, 1046000), 1000020, 1450
        dels = lambda k, r: [synth.Deletion(1015000, 1200, [2, 1, 0, 0][k % 4]), synth.Deletion(1031000, 500, [0, 1, 2][k % 3])]
        parts, offsets, rgs, sample_offsets = [], [0], [], [0]
        for k in range(520):
            kind = "tight" if k == 0 else ("deep" if k == 300 else ("multi" if k == 515 else "normal"))
            n_rg = 3 if kind == "multi" else 1
            for j in range(n_rg):
                cov = 85.0 if kind == "deep" else 30.0 / n_rg
                pos, dev = synth.sample_pairs(7000 + 3 * k + j, region, synth.Library(coverage=cov), dels(k, region))
                d, s, a, b = synth.instances_by_interval_rule(pos, dev, lib, first, W)
                parts.append((d, s, a, b))
                offsets.append(offsets[-1] + len(d))
                rgs.append(rg_tight if kind == "tight" else base)
            sample_offsets.append(len(rgs))
        seg = capi.Segment(first, W, np.array(offsets, np.uint64), np.concatenate([p[0] for p in parts]).astype(np.int32),
                           np.concatenate([p[1] for p in parts]).astype(np.uint32),
                           np.concatenate([p[2] for p in parts]).astype(np.uint32),
                           np.concatenate([p[3] for p in parts]).astype(np.uint32))
        return seg, rgs, sample_offsets, {}
    elif name == "tile3000":
        # a BASELINE-shaped segment (6,667 windows, 3,000 samples) generated on the GPU exactly like bench.py does
        import importlib.util
        import torch
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("popdel_bench", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        n = 3000
        first, W, offsets, arrays, _ = bench.device_tile(n, 777, torch.device("cuda", 0))
        seg = capi_segment(first, W, offsets, arrays)
        return seg, [base] * n, list(range(n + 1)), {}
    elif name.startswith("stress"):
        # random cohort of tools/stress_consistency.py (several hundred samples, 1-3 read groups each)
        import importlib.util
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("stress_consistency_cases", os.path.join(root, "tools", "stress_cases.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.case(int(name[6:]))
    else:
        raise ValueError(name)
    return seg, rgs, list(range(n + 1)), opts


def main():
    root, case, out = sys.argv[1:4]
    device = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    sys.path.insert(0, root)
    import torch.distributed as dist
    from popdel_amd import capi, distributed
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(os.environ.get("POPDEL_TEST_BACKEND", "gloo"), rank=rank, world_size=world)
    seg, rgs, offs, opts = make_case(case)
    lanes = int(os.environ.get("POPDEL_TEST_LANES", "0"))
    if lanes:
        # the cohort keeps its lanes: one process group per lane (created in the same order on every rank), several tickets
        # in flight, every one of them the whole segment
        os.environ["POPDEL_HIP_LANES"] = str(lanes)
        groups = [dist.new_group(backend=os.environ.get("POPDEL_TEST_BACKEND", "gloo")) for _ in range(lanes)]
        shard = distributed.SampleShard(capi.load_hip(), rgs, offs, rank, world, device=device, lane_groups=groups, **opts)
        handle = shard.ctx.stage(shard.local_segment(seg))
        tickets = [shard.ctx.run(handle) for _ in range(2 * lanes)]
        results = [shard.gather_columns(shard.ctx.collect(t)) for t in tickets]
        shard.ctx.unstage(handle)
        for x in [shard.exchange] + shard.lane_exchanges:
            if x.last_error is not None:
                raise x.last_error
        if rank == 0:
            pickle.dump(dict(calls=results, exchanges=shard.exchange.calls, lane_exchanges=[x.calls for x in shard.lane_exchanges]), open(out, "wb"))
        dist.barrier()
        shard.close()
        dist.destroy_process_group()
        return
    shard = distributed.SampleShard(capi.load_hip(), rgs, offs, rank, world, device=device, stage1_view=True, **opts)
    res = shard.run(shard.local_segment(seg))
    assert res["n_samples"] == shard.last - shard.first
    full = shard.gather_columns(res)
    q3, fl = shard.ctx.stage1()
    if rank == 0:
        pickle.dump(dict(calls=full, exchanges=shard.exchange.calls, q3_shape=q3.shape), open(out, "wb"))
    dist.barrier()
    shard.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
