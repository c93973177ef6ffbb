#!/usr/bin/env python3
"""Throughput of the joint-calling hot path (window x sample likelihood evaluations per second) on MI355X.

Workload (BASELINE.json configs[2]): 10,000 synthetic profiles x chr20 (2,148,139 calling windows of 30 bp).
The pair stream of 10,000 samples x chr20 is ~770 GB, so the chromosome is processed the way the host driver does it:
one 200-kb segment (6,667 windows) of all samples at a time. The bench keeps ONE such segment of all 10,000 samples
resident in HBM (generated on the device, staged through the C ABI before the clock starts) and a "step" runs the
complete hot path -- stage-1 Q3 scan, candidate clustering, EM + final genotyping for every candidate cell, result
packing -- over it 323 times, i.e. over as many windows as chr20 has. Nothing is cached between passes.

With --gpus N every rank runs such a chromosome-sized step on its own GPU (segments are independent given the host-side
carry-over of active pairs, so whole-genome work is dealt out by segment/chromosome with no data-path collective):
per-GPU work is fixed (weak scaling), value = N x cells / max-over-ranks time.

With N > 1 the same JSON line also carries "sample_sharded": the other exact split (north_star) -- ONE cohort
of N x 9,984 samples, every rank owning the pairs of its own 9,984 samples, the library all-gathering over RCCL what
crosses samples (popdel_amd/distributed.py: SampleShard) -- timed on a few passes of the same segment. It is a secondary
figure: per-pass collectives cost latency that whole-segment sharding does not pay, so the headline stays on segments.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR20_WINDOWS = 2148139
SEG_WINDOWS = 6667            # windows walked per 200-kb segment (windowBuffer 200000 / 30)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def device_tile(n_samples, seed, device, coverage=30.0):
    """Pair instances of one 200-kb segment for n_samples single-read-group samples, generated with torch on the GPU:
    30x coverage, 2x150 bp, insert ~ round(N(500, 50)), one planted deletion (length log-uniform 300..5000,
    population AF ~ Beta(0.5, 2), genotype ~ Binomial(2, AF)); activity by the interval rule."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    first, shift, W = 1000020, 30, SEG_WINDOWS
    lo, hi = first - 600, first + W * shift
    n = int((hi - lo) * coverage / 300.0)
    median, read_len = 500, 150
    pos = torch.randint(lo, hi, (n_samples, n), generator=g, device=device, dtype=torch.int64)
    pos, _ = torch.sort(pos, dim=1)
    ins = torch.round(torch.randn((n_samples, n), generator=g, device=device) * 50.0 + median).to(torch.int64)
    hap = torch.randint(0, 2, (n_samples, n), generator=g, device=device, dtype=torch.int64)
    rs = np.random.default_rng(seed)
    del_len = int(np.exp(rs.uniform(np.log(300), np.log(5000))))
    del_start = first + (W // 2) * shift
    af = rs.beta(0.5, 2.0)
    gt = torch.from_numpy(rs.binomial(2, af, size=n_samples)).to(device).view(-1, 1)
    carrier = (gt == 2) | ((gt == 1) & (hap == 0))
    inner = torch.clamp(ins - 2 * read_len, min=1)
    inside = carrier & (pos >= del_start) & (pos < del_start + del_len)
    span = carrier & (pos < del_start) & (pos + inner > del_start)
    ins = torch.where(span, ins + del_len, ins)
    dev = ins - median
    end = pos + torch.clamp(dev + median - 2 * read_len, min=0)
    last = torch.div(end, shift, rounding_mode="floor") * shift + shift
    on = torch.clamp(-torch.div(first - pos, shift, rounding_mode="floor"), min=0)
    off = torch.clamp(torch.div(last - first, shift, rounding_mode="floor") + 1, max=W)
    keep = (~inside) & (off > on) & (on < W)
    counts = keep.sum(dim=1).cpu().numpy().astype(np.uint64)
    offsets = np.zeros(n_samples + 1, dtype=np.uint64)
    np.cumsum(counts, out=offsets[1:])
    flat = keep.view(-1)
    out = dict(dev=dev.view(-1)[flat].to(torch.int32).cpu().numpy(), start=pos.view(-1)[flat].to(torch.int32).cpu().numpy().astype(np.uint32),
               on=on.view(-1)[flat].to(torch.int32).cpu().numpy().astype(np.uint32),
               off=off.view(-1)[flat].to(torch.int32).cpu().numpy().astype(np.uint32))
    del pos, ins, hap, dev, end, last, on, off, keep, flat
    torch.cuda.empty_cache()
    return first, W, offsets, out, dict(del_len=del_len, af=float(af))


def take_samples(offsets, arrays, n):
    e = int(offsets[n])
    return offsets[: n + 1].copy(), {k: v[:e] for k, v in arrays.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--passes", type=int, default=0, help="segment passes per step (default: chr20 = 323)")
    ap.add_argument("--cpu-samples", type=int, default=600, help="samples of the same segment timed on one host core")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the all-cores CPU figure (0/1: skip)")
    ap.add_argument("--sharded-passes", type=int, default=40,
                    help="passes of the secondary sample-sharded leg (N > 1 only; 0 switches it off)")
    ap.add_argument("--sharded-timeout", type=float, default=240.0, help="give up on the secondary leg after this many seconds")
    ap.add_argument("--stream-segments", type=int, default=4,
                    help="segments of the streaming leg (host memory -> HBM -> calls, double buffered); 0 switches it off")
    ap.add_argument("--coverage", type=float, default=30.0, help="sequencing depth of the synthetic samples (headline: 30)")
    ap.add_argument("--rgs-per-sample", type=int, default=1,
                    help="split every sample's pairs over this many read groups (same library): measures the generic "
                         "kernels that multi-read-group samples take; the headline workload has 1")
    ap.add_argument("--serial", action="store_true",
                    help="one ticket at a time (no pipelining across segments). Used for the rocprofv3 kernel trace: under "
                         "the profiler the result copy runs as a blit KERNEL that competes with the next scan, which "
                         "inflates the scan's traced duration; un-profiled runs copy over SDMA")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--concurrent", type=int, default=3,
                    help="contexts of the secondary concurrent-segments leg (0/1: skip it)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="segments in flight per GPU (one context and host thread each); 1 keeps the kernel timings clean")
    args = ap.parse_args()

    import torch
    from popdel_amd import capi, distributed, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    is_dist = world > 1
    if is_dist:
        import torch.distributed as dist
        if args.share_gpu:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    device = torch.device("cuda", local_rank)
    red_device = device if args.backend == "nccl" else torch.device("cpu")       # where timing reductions live

    hip = capi.load_hip()                     # no fallback: raises if the HIP library is missing
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    S = args.samples
    first, W, offsets, arrays, planted = device_tile(S, 20260 + rank, device, args.coverage)
    R = max(1, args.rgs_per_sample)
    sample_offsets = list(range(0, R * S + 1, R))
    if R > 1:
        # deal the pairs of every sample to R read groups (stable, so each read group stays ordered by start window)
        rs = np.random.default_rng(7 + rank)
        which = rs.integers(0, R, size=len(arrays["dev"]))
        sample_of = np.repeat(np.arange(S), np.diff(offsets).astype(np.int64))
        key = sample_of * R + which
        order = np.argsort(key, kind="stable")
        arrays = {k: v[order] for k, v in arrays.items()}
        counts = np.bincount(key, minlength=R * S).astype(np.uint64)
        offsets = np.zeros(R * S + 1, dtype=np.uint64)
        np.cumsum(counts, out=offsets[1:])
    seg = capi.Segment(first, W, offsets, arrays["dev"], arrays["start"], arrays["on"], arrays["off"])
    ctx = capi.Context(hip, [rg] * (R * S), sample_offsets, device=local_rank)
    t_stage = time.perf_counter()
    handle = ctx.stage(seg)                   # host -> HBM, outside the timed region
    stage_seconds = time.perf_counter() - t_stage
    # optional extra contexts: independent segments in flight on the same GPU (segments do not depend on each other)
    extra = []
    for _ in range(max(0, args.contexts - 1)):
        c2 = capi.Context(hip, [rg] * (R * S), sample_offsets, device=local_rank)
        extra.append((c2, c2.stage(seg)))

    total_passes = args.passes or -(-CHR20_WINDOWS // W)
    # weak scaling: the global job is world x total_passes segments, dealt round-robin -> total_passes per rank
    my_passes = len(distributed.shard_segments(total_passes * world, rank, world))

    def passes_on(c, h, n, acc):
        # software-pipelined the way a streaming host drives the library: the next segment's ticket is issued before
        # the open one is collected, so its scan starts behind the open segment's last kernel, under the result D2H
        ticket = c.run(h) if n else None
        for i in range(n):
            nxt = c.run(h) if (i + 1 < n and not args.serial) else None
            res = c.collect(ticket, copy=False)
            ticket = nxt if not args.serial else (c.run(h) if i + 1 < n else None)
            t = c.timing()
            acc[0] += t["stage1_ms"]
            acc[1] += t["stage2_ms"]
            acc[2] += res["n_calls"]
            acc[3] += t["n_candidate_cells"]
            acc[4] = t

    def one_step(lanes=None):
        lanes = lanes or ([(ctx, handle)] + extra)
        if len(lanes) == 1:
            acc = [0.0, 0.0, 0, 0, None]
            passes_on(ctx, handle, my_passes, acc)
            return tuple(acc)
        import threading
        share = [len(range(k, my_passes, len(lanes))) for k in range(len(lanes))]
        accs = [[0.0, 0.0, 0, 0, None] for _ in lanes]
        threads = [threading.Thread(target=passes_on, args=(c, h, n, a)) for (c, h), n, a in zip(lanes, share, accs)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        return (sum(a[0] for a in accs), sum(a[1] for a in accs), sum(a[2] for a in accs), sum(a[3] for a in accs), accs[0][4])

    def sync():
        torch.cuda.synchronize()
        if is_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    s1_ms = s2_ms = 0.0
    n_calls = n_cand = 0
    for _ in range(args.steps):
        a, b, c, d, tinfo = one_step()
        s1_ms += a
        s2_ms += b
        n_calls += c
        n_cand += d
    sync()
    elapsed = time.perf_counter() - t0
    if is_dist:
        tt = torch.tensor([elapsed], device=red_device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # streaming leg (reported, never `value`): every segment comes from host memory -- chunk layout built by the host
    # threads, uploaded over the copy stream while the previous segment computes, results collected -- i.e. what a host
    # that has the pair instances in RAM can sustain through one GPU
    streaming = None
    if args.stream_segments > 0:
        sync()
        q0 = time.perf_counter()
        h_cur = ctx.stage(seg)
        t_cur = ctx.run(h_cur)
        for i in range(args.stream_segments):
            last = i + 1 == args.stream_segments
            if not last:
                h_nxt = ctx.stage(seg)
                t_nxt = ctx.run(h_nxt)
            ctx.collect(t_cur, copy=False)
            ctx.unstage(h_cur)
            if not last:
                h_cur, t_cur = h_nxt, t_nxt
        sync()
        dt = time.perf_counter() - q0
        streaming = dict(value=args.stream_segments * W * S / dt, unit="window*sample/s", segments=args.stream_segments,
                         seconds_per_segment=dt / args.stream_segments,
                         host_bytes_per_segment=int(16 * len(arrays["dev"]) + 8 * (S + 1)))

    cells_per_step = total_passes * W * S * world
    value = cells_per_step * args.steps / elapsed

    # concurrent-segments leg (reported, never `value`): the same step with several independent segments in flight on the
    # GPU, one context and host thread each. Stage 2 is a chain of short gather-bound kernels that leaves most of the
    # chip idle; another segment's scan fills it. Kept out of the headline because concurrent kernels stretch each
    # other's durations, which would make the scan's roofline figure meaningless.
    concurrent = None
    if args.concurrent > 1 and args.contexts == 1:
        try:
            more = []
            for _ in range(args.concurrent - 1):
                c2 = capi.Context(hip, [rg] * (R * S), sample_offsets, device=local_rank)
                more.append((c2, c2.stage(seg)))
            lanes = [(ctx, handle)] + more
            saved = my_passes
            my_passes = max(len(lanes), saved // 8)
            one_step(lanes)                                        # warm-up of the new contexts
            my_passes = saved
            torch.cuda.synchronize()
            q0 = time.perf_counter()
            one_step(lanes)
            torch.cuda.synchronize()
            dt = time.perf_counter() - q0
            # no collective in here (a rank that fails must not hang the others): with more than one rank this is rank
            # 0's own clock, the ranks doing identical work side by side
            concurrent = dict(value=cells_per_step / dt, unit="window*sample/s", contexts_per_gpu=len(lanes),
                              ms_per_pass=dt / my_passes * 1e3, steps=1)
            for c2, h2 in more:
                c2.unstage(h2)
                c2.close()
        except Exception as e:          # a secondary leg must never take the headline with it
            my_passes = len(distributed.shard_segments(total_passes * world, rank, world))
            concurrent = {"error": "%s: %s" % (type(e).__name__, e)}

    # roofline of the dominant kernel (stage-1 scan): algorithmic bytes per launch / mean launch duration
    launches = my_passes * args.steps
    s1_mean_s = s1_ms / launches / 1e3
    cells = tinfo["n_cells"]
    alg_bytes = 4 * tinfo["n_active_pairs"] + 4 * cells + 4 * cells          # gather + cell index + Q3 out
    # what this layout actually has to move: 8 B per chunk + chunk index + end counts (4 B each per window) + 5 B out
    compact_bytes = 8 * tinfo["n_pair_chunks"] + 4 * (W + tinfo["max_life"]) * S + 4 * (W + 1) * S + 5 * cells
    achieved = alg_bytes / s1_mean_s / 1e9
    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured with rocprofv3 on
    # exactly this launch shape; summary committed under profiles/. Only quoted when the shape matches.
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_stage1.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        if pmc["config"] == {"samples": S, "windows": W}:
            traffic = pmc["hbm_bytes_per_launch"]
    roofline = dict(bound="hbm", kernel="stage1Kernel", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source="profiles/r01_pmc_stage1.json" if traffic else None,
                    algorithmic_bytes_per_launch=alg_bytes,
                    compact_bytes_per_launch=compact_bytes, compact_GBs=compact_bytes / s1_mean_s / 1e9,
                    mean_launch_ms=s1_mean_s * 1e3, mean_active_pairs_per_cell=tinfo["n_active_pairs"] / cells)

    cpu_baseline = None
    if rank == 0 and not args.no_cpu:
        oracle_path = os.path.join(ROOT, "oracle", "_build", "libpopdel_oracle.so")
        if os.path.exists(oracle_path):
            n_cpu = min(args.cpu_samples, S)
            o_off, o_arr = take_samples(offsets, arrays, n_cpu)
            oseg = capi.Segment(first, W, o_off, o_arr["dev"], o_arr["start"], o_arr["on"], o_arr["off"])
            octx = capi.Context(capi.load_library(oracle_path), [rg] * n_cpu, list(range(n_cpu + 1)))
            c0 = time.perf_counter()
            octx.collect(octx.submit(oseg))
            cpu_s = time.perf_counter() - c0
            octx.close()
            cpu_baseline = dict(value=n_cpu * W / cpu_s, unit="window*sample/s", cores=1, kind="port",
                                sample="%d of the %d samples x the same %d-window segment, oracle/ (x87 long double), %.1f s"
                                       % (n_cpu, S, W, cpu_s))
            # the same on all host cores the way a CPU run would use them: independent cohorts of n_cpu samples each (the
            # oracle is single-threaded; ctypes releases the GIL), reported next to the one-core figure
            import threading
            n_thr = max(1, min(args.cpu_threads, os.cpu_count() or 1, S // n_cpu))
            if n_thr > 1:
                olib = capi.load_library(oracle_path)
                parts = []
                for t in range(n_thr):
                    lo, hi = int(offsets[t * n_cpu]), int(offsets[(t + 1) * n_cpu])
                    parts.append(capi.Segment(first, W, offsets[t * n_cpu:(t + 1) * n_cpu + 1] - offsets[t * n_cpu],
                                              arrays["dev"][lo:hi], arrays["start"][lo:hi], arrays["on"][lo:hi], arrays["off"][lo:hi]))
                ctxs = [capi.Context(olib, [rg] * n_cpu, list(range(n_cpu + 1))) for _ in range(n_thr)]
                work = [threading.Thread(target=lambda c=c, p=p: c.collect(c.submit(p), copy=False)) for c, p in zip(ctxs, parts)]
                c0 = time.perf_counter()
                for th in work:
                    th.start()
                for th in work:
                    th.join()
                cpu_all_s = time.perf_counter() - c0
                for c in ctxs:
                    c.close()
                cpu_baseline["all_cores"] = dict(value=n_thr * n_cpu * W / cpu_all_s, unit="window*sample/s", cores=n_thr,
                                                 sample="%d independent cohorts of %d samples, one oracle context per thread, %.1f s"
                                                        % (n_thr, n_cpu, cpu_all_s))

    # ---- secondary leg: the cohort sharded by SAMPLES over the ranks (collectives in the data path) --------------------
    result = {}

    def emit_and_finish(sample_sharded):
        if rank == 0:
            line = dict(result)
            line["sample_sharded"] = sample_sharded
            print(json.dumps(line), flush=True)

    def sample_sharded_leg():
        n = S // 256 * 256                                     # whole tiles per rank (partition rule of pd_shard_samples)
        o, a = take_samples(offsets, arrays, n)
        lseg = capi.Segment(first, W, o, a["dev"], a["start"], a["on"], a["off"])
        shard = distributed.SampleShard(hip, [rg] * (n * world), list(range(n * world + 1)), rank, world, device=local_rank)
        h = shard.ctx.stage(lseg)
        calls = 0
        for _ in range(2):
            shard.ctx.collect(shard.ctx.run(h), copy=False)
        x0 = shard.exchange.calls, shard.exchange.bytes
        sync()
        q0 = time.perf_counter()
        ticket = shard.ctx.run(h)
        for i in range(args.sharded_passes):
            nxt = shard.ctx.run(h) if i + 1 < args.sharded_passes else None
            calls += shard.ctx.collect(ticket, copy=False)["n_calls"]
            ticket = nxt
        sync()
        dt = time.perf_counter() - q0
        tt = torch.tensor([dt], device=red_device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tm = shard.ctx.timing()
        out = dict(value=args.sharded_passes * W * n * world / dt, unit="window*sample/s", samples_per_gpu=n,
                   cohort_samples=n * world, passes=args.sharded_passes, ms_per_pass=dt / args.sharded_passes * 1e3,
                   all_gathers_per_pass=(shard.exchange.calls - x0[0]) / args.sharded_passes,
                   all_gather_bytes_per_pass=(shard.exchange.bytes - x0[1]) / args.sharded_passes,
                   calls_per_pass=calls / args.sharded_passes, candidate_cells_per_pass=tm["n_candidate_cells"],
                   backend=dist.get_backend(), scaling="weak")
        shard.ctx.unstage(h)
        shard.close()
        return out

    if rank == 0:
        result = ({
            "metric": "window x sample likelihood evals/sec", "value": value, "unit": "window*sample/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 scan + f64 likelihoods",
            "data": "synthetic",
            "config": {"workload": "10,000 synthetic profiles x chr20 windows on MI355X (configs[2])" if S == 10000 and not args.passes
                       else "%d synthetic profiles x %d windows" % (S, total_passes * W),
                       "samples": S, "windows_per_step": total_passes * W * world, "segment_windows": W,
                       "segment_passes_per_step": total_passes, "resident_segment": "one 200-kb segment of all samples, re-scanned",
                       "planted_deletion": planted, "parallelism": "%d rank(s), each a chr20-sized set of segments (no collective in the data path)" % world},
            "stage1_ms_per_step": s1_ms / args.steps, "stage2_ms_per_step": s2_ms / args.steps,
            "calls_per_step": n_calls / args.steps, "candidate_cells_per_step": n_cand / args.steps,
            "max_life": tinfo["max_life"], "contexts_per_gpu": args.contexts, "pipelined_tickets": not args.serial,
            # host arrays -> chunked layout -> HBM for one segment of all samples (single host thread + PCIe); with it
            # inside every pass the rate would be cells_per_pass / (stage_seconds + pass time): reported, never `value`
            "staging_seconds_per_segment": stage_seconds,
            "pcie_inclusive_value": (W * S) / (stage_seconds + elapsed / args.steps / max(1, my_passes)),
            "streaming": streaming, "concurrent_segments": concurrent,
            "roofline": roofline, "cpu_baseline": cpu_baseline})
    ctx.unstage(handle)
    ctx.close()
    for c2, h2 in extra:
        c2.unstage(h2)
        c2.close()
    if not is_dist or args.sharded_passes <= 0:
        emit_and_finish(None)
    else:
        # A collective that never completes must not take the headline with it: a watchdog prints the line and ends the
        # process if the secondary leg overruns.
        import threading

        def give_up():
            emit_and_finish({"error": "sample-sharded leg did not finish within %.0f s" % args.sharded_timeout})
            os._exit(0)

        dog = threading.Timer(args.sharded_timeout, give_up)
        dog.daemon = True
        dog.start()
        try:
            leg = sample_sharded_leg()
        except Exception as e:          # reported, never fatal for the headline
            leg = {"error": "%s: %s" % (type(e).__name__, e)}
        dog.cancel()
        emit_and_finish(leg)
    if is_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
