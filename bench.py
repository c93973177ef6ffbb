#!/usr/bin/env python3
"""Throughput of the joint-calling hot path (window x sample likelihood evaluations per second) on MI355X.

Workload (BASELINE.json configs[2]): 10,000 synthetic profiles x chr20 (2,148,139 calling windows of 30 bp).
The pair stream of 10,000 samples x chr20 is ~770 GB, so the chromosome is processed the way the host driver does it:
one 200-kb segment (6,667 windows) of all samples at a time. The bench keeps ONE such segment of all samples resident in
HBM (generated on the device, staged through the C ABI before the clock starts) and a "step" runs the complete hot path
-- stage-1 Q3 scan, candidate clustering, EM + final genotyping for every candidate cell, result packing -- over it 323
times, i.e. over as many windows as chr20 has. Nothing is cached between passes.

`--gpus N` (N > 1): when no launcher has set WORLD_SIZE, this process -- before torch is imported or a GPU touched --
starts N ranks of itself through `python -m torch.distributed.run` (one per GPU, RCCL) and exits with their status.
The N-rank headline is the split north_star names: ONE cohort of N x 9,984 samples sharded BY SAMPLES, every
rank staging and scanning only its own samples, the library all-gathering what crosses samples between its kernels
(popdel_amd/distributed.py: SampleShard); per-GPU work is fixed (weak scaling), value = cohort cells / max-over-ranks
time. The same line carries `segment_sharded`: the other exact split (whole segments of a 9,984-sample cohort per rank,
no collective in the data path), measured before the headline so that an RCCL failure in the exchanges still leaves a
line: it is then printed with `headline_fallback` set and the process exits non-zero.

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR20_WINDOWS = 2148139
SEG_WINDOWS = 6667            # windows walked per 200-kb segment (windowBuffer 200000 / 30)
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
SCAN_SOURCES = ("popdel_amd/csrc/hip/stage1.hip", "popdel_amd/csrc/hip/sortnet_gen.h", "popdel_amd/csrc/Makefile")   # (the Makefile: the scan's own compiler flags)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--passes", type=int, default=0, help="segment passes per step (default: chr20 = 323)")
    ap.add_argument("--cpu-samples", type=int, default=600, help="samples of the same segment timed on one host core")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="threads of the all-cores CPU figure (-1: every core the job may use; 0/1: skip)")
    ap.add_argument("--sharded-timeout", type=float, default=420.0,
                    help="N > 1: give up on the sample-sharded headline after this many seconds (the line is then printed "
                         "with the segment split as value and the process exits 3)")
    ap.add_argument("--stream-segments", type=int, default=8,
                    help="segments of the streaming leg (host memory -> HBM -> calls, double buffered); 0 switches it off")
    ap.add_argument("--stream50k-samples", type=int, default=50000,
                    help="cohort of the streaming_50000 leg (configs[4]'s size through one GPU; 0: skip)")
    ap.add_argument("--stream50k-segments", type=int, default=3)
    ap.add_argument("--coverage", type=float, default=30.0, help="sequencing depth of the synthetic samples (headline: 30)")
    ap.add_argument("--rgs-per-sample", type=int, default=1,
                    help="split every sample's pairs over this many read groups (same library): measures the kernels "
                         "that multi-read-group samples take; the headline workload has 1")
    ap.add_argument("--serial", action="store_true",
                    help="one ticket at a time (no pipelining across segments). Used for the rocprofv3 kernel trace: under "
                         "the profiler the result copy runs as a blit KERNEL that competes with the next scan, which "
                         "inflates the scan's traced duration; un-profiled runs copy over SDMA")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--exchange", default="auto", choices=("auto", "native", "torch"),
                    help="N > 1: who runs the all-gathers of the sample-sharded cohort: native = the library itself over RCCL "
                         "on its own stream (default with --backend nccl), torch = host callbacks over torch.distributed "
                         "(default otherwise: gloo dry runs)")
    ap.add_argument("--concurrent", type=int, default=3,
                    help="contexts of the secondary concurrent-segments leg (0/1: skip it)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="segments in flight per GPU (one context and host thread each); 1 keeps the kernel timings clean")
    ap.add_argument("--dense-deletions", type=int, default=10,
                    help="planted deletions of the candidate-dense leg's segment (SURVEY 6: 49 per Mb ~ 10 per 200 kb); 0: skip")
    ap.add_argument("--small-cohort", type=int, default=1000,
                    help="samples of the configs[1] leg (1,000 profiles x chr20 windows); 0: skip")
    ap.add_argument("--no-legs", action="store_true", help="headline only (profiling runs)")
    ap.add_argument("--resident-segments", type=int, default=10,
                    help="N = 1: distinct 200-kb segments (own pairs, own planted deletion) kept resident and scanned in turn by the "
                         "headline's passes. Deletions as SURVEY 8(d) plants them -- one per 250 kb, length log-uniform 300..5,000 bp, "
                         "AF ~ Beta(0.5, 2) -- so of every five consecutive 200-kb segments one has none (default 10: eight with a "
                         "deletion, two without; 1: the single re-scanned segment of rounds 1-3)")
    ap.add_argument("--distinct-segments", type=int, default=0,
                    help="N = 1: segments of the `distinct_segments` leg of round 3 (superseded by the headline's own mix); 0: skip")
    ap.add_argument("--lanes", type=int, default=3,
                    help="lanes of every context the bench creates (POPDEL_HIP_LANES; the library's default is 3): child contexts that "
                         "take the tickets in turn so that one segment's stage 2 fills the gaps of another's; the passes keep "
                         "2 x lanes tickets in flight. 1: the plain context of rounds 1-3")
    ap.add_argument("--configs3-samples", type=int, default=10000,
                    help="N > 1: TOTAL samples of the configs[3]-shaped leg (ONE cohort sharded by samples, pd_shard_samples(total, N) "
                         "per rank, resident segment); 0: skip")
    ap.add_argument("--configs4-samples", type=int, default=50000,
                    help="N > 1: TOTAL samples of the configs[4]-shaped leg (the same, every segment streamed from host memory, "
                         "double buffered); 0: skip")
    ap.add_argument("--configs-passes", type=int, default=0,
                    help="passes of the configs3 leg (default: chr20 = 323) -- the configs4 leg streams --configs4-segments")
    ap.add_argument("--configs4-segments", type=int, default=6)
    ap.add_argument("--e2e-samples", type=int, default=1000,
                    help="N = 1: profiles of the end-to-end leg (`popdel call` on generated profile files, 1 Mb region); 0: skip")
    ap.add_argument("--e2e-cohort-samples", type=int, default=4000,
                    help="N = 1: profiles of the cohort-scale end-to-end leg (`popdel call` on generated profiles x 4 Mb = 20 segments; "
                         "steady state: start-up and teardown excluded); 0: skip")
    ap.add_argument("--e2e-cohort-mb", type=int, default=4, help="region of that leg in Mb")
    ap.add_argument("--e2e-cohort-budget", type=float, default=300.0,
                    help="wall-time budget of that leg in seconds (cohort writing included): the cohort is cut down to what the "
                         "writing rate measured on the first profiles allows")
    ap.add_argument("--inflate-members", type=int, default=100000,
                    help="N = 1: gzip members of the device-gunzip leg (pd_inflate_members against libdeflate on the host cores); 0: skip")
    ap.add_argument("--sharded-world1", default="1280,6400,0",
                    help="N = 1: samples of the legs that run the sample-sharded code path with ONE rank (library RCCL): its own "
                         "cost per pass next to the unsharded context; 0 = the headline cohort; empty: skip")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks of this script (one per GPU). Runs in a process that has not imported
    torch; the ranks are fresh child processes of the launcher module, nothing is re-executed in place."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def device_tile(n_samples, seed, device, coverage=30.0, n_deletions=1, deletion_seed=None):
    """Pair instances of one 200-kb segment for n_samples single-read-group samples, generated with torch on the GPU:
    30x coverage, 2x150 bp, insert ~ round(N(500, 50)), planted deletions (length log-uniform 300..5000, population
    AF ~ Beta(0.5, 2), genotype ~ Binomial(2, AF); one in the middle of the segment, or n_deletions evenly spaced);
    activity by the interval rule."""
    import numpy as np
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
    # the deletions (length, population frequency) belong to the cohort, the genotypes to this rank's samples
    rs = np.random.default_rng(seed if deletion_seed is None else deletion_seed)
    rs_gt = rs if deletion_seed in (None, seed) else np.random.default_rng(seed)
    inner = torch.clamp(ins - 2 * read_len, min=1)
    inside = torch.zeros_like(pos, dtype=torch.bool)
    planted = []
    for k in range(n_deletions):
        del_len = int(np.exp(rs.uniform(np.log(300), np.log(5000))))
        del_start = first + int((k + 0.5) * W / n_deletions) * shift
        af = rs.beta(0.5, 2.0)
        gt = torch.from_numpy(rs_gt.binomial(2, af, size=n_samples)).to(device).view(-1, 1)
        carrier = (gt == 2) | ((gt == 1) & (hap == 0))
        inside |= carrier & (pos >= del_start) & (pos < del_start + del_len)
        span = carrier & (pos < del_start) & (pos + inner > del_start)
        ins = torch.where(span, ins + del_len, ins)
        planted.append(dict(del_len=del_len, af=float(af)))
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
    del pos, ins, hap, dev, end, last, on, off, keep, flat, inside
    torch.cuda.empty_cache()
    return first, W, offsets, out, (planted[0] if n_deletions == 1 else planted)


def take_samples(offsets, arrays, n):
    e = int(offsets[n])
    return offsets[: n + 1].copy(), {k: v[:e] for k, v in arrays.items()}


def scan_source_stamp():
    """Identifies the build of the scan kernel a PMC summary under profiles/ was taken on."""
    h = hashlib.sha256()
    for rel in SCAN_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cfs_throttle():
    """(periods throttled, seconds throttled) of this job's CPU controller so far, or None: the GPU boxes give a job a CFS quota
    (16 CPUs' worth per period); a phase that runs more threads than that is stopped for the rest of the period."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"):
        try:
            kv = dict(l.split()[:2] for l in open(path) if len(l.split()) >= 2)
        except OSError:
            continue
        if "nr_throttled" in kv:
            t = float(kv["throttled_usec"]) / 1e6 if "throttled_usec" in kv else float(kv.get("throttled_time", 0)) / 1e9
            return int(kv["nr_throttled"]), t
    return None


def usable_cores():
    """CPUs this job may use: the affinity mask, cut down to the cgroup CPU-time quota if there is one (the GPU boxes show
    256 logical CPUs and grant a job 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def bind_to_gpu_node(device_index):
    """Binds this process to the CPUs of the NUMA node its GPU hangs off (before anything touches the GPU, and before the big
    host arrays are allocated: first touch then puts them next to the GPU's PCIe root). On the two-socket GPU boxes an
    unbound run lands on either socket from run to run -- the streaming leg was seen at 0.55 and at 1.8e9 window x sample/s
    on one box. Which node: the k-th GPU of the KFD topology this job may see -> its PCI address -> sysfs. Returns the node,
    or None if anything is not as expected (the process is then left as it is). BENCH_NO_NUMA_BIND=1 switches it off."""
    if os.environ.get("BENCH_NO_NUMA_BIND") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        gpus = []
        base = "/sys/class/kfd/kfd/topology/nodes"
        for name in sorted(os.listdir(base), key=int):
            try:
                props = dict(l.split()[:2] for l in open(os.path.join(base, name, "properties")) if len(l.split()) >= 2)
            except OSError:
                continue            # (other jobs' GPUs are hidden from this one)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        if device_index >= len(gpus):
            return None
        loc, dom = int(gpus[device_index]["location_id"]), int(gpus[device_index].get("domain", "0"))
        bdf = "%04x:%02x:%02x.%x" % (dom & 0xffff, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except (OSError, ValueError, KeyError):
        return None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def new_acc():
    """[stage-1 ms, stage-2 ms, calls, candidate cells, last pd_timing, exchange ms] summed over passes"""
    return [0.0, 0.0, 0, 0, None, 0.0]


LANES = 1       # lanes of the contexts the bench creates (set in main from --lanes)


class Lane:
    """One context with one resident segment; runs passes the way a streaming host drives the library: the next
    segment's ticket is issued before the open one is collected."""

    def __init__(self, ctx, handle, serial=False, depth=None):
        """handle: a staged segment, or a list of them (scanned in turn); depth: tickets kept in flight (2 for a plain context:
        the next segment's ticket is issued before the open one is collected; 2 per lane for a context with lanes)"""
        self.ctx, self.serial = ctx, serial
        self.depth = max(1, depth if depth is not None else (2 * LANES if LANES > 1 else 2))
        self.handles = list(handle) if isinstance(handle, (list, tuple)) else [handle]
        self.handle = self.handles[0]
        self.turn = 0

    def next_handle(self):
        h = self.handles[self.turn % len(self.handles)]
        self.turn += 1
        return h

    def passes(self, n, acc):
        c = self.ctx
        depth = 1 if self.serial else self.depth
        open_tickets, issued = [], 0
        for i in range(n):
            while issued < n and len(open_tickets) < depth:
                open_tickets.append(c.run(self.next_handle()))
                issued += 1
            res = c.collect(open_tickets.pop(0), copy=False)
            t = c.timing()
            acc[0] += t["stage1_ms"]
            acc[1] += t["stage2_ms"]
            acc[2] += res["n_calls"]
            acc[3] += t["n_candidate_cells"]
            acc[4] = t
            if len(acc) > 5:
                acc[5] += t.get("exchange_ms", 0.0)


def run_lanes(lanes, n_passes):
    if len(lanes) == 1:
        acc = new_acc()
        lanes[0].passes(n_passes, acc)
        return tuple(acc)
    share = [len(range(k, n_passes, len(lanes))) for k in range(len(lanes))]
    accs = [new_acc() for _ in lanes]
    threads = [threading.Thread(target=l.passes, args=(n, a)) for l, n, a in zip(lanes, share, accs)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    return (sum(a[0] for a in accs), sum(a[1] for a in accs), sum(a[2] for a in accs), sum(a[3] for a in accs), accs[0][4],
            sum(a[5] for a in accs))


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if int(env_world or "1") > 1:
        args.lanes = 1              # (a cohort over several ranks runs without lanes)
    os.environ["POPDEL_HIP_LANES"] = str(max(1, args.lanes))
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to print a line whose "
                 "n_gpus would not be what was asked for" % (args.gpus, world))

    # The one JSON line goes to the REAL stdout; everything else any library prints there (RCCL announces its version on
    # stdout when a communicator is created) is sent to stderr for the lifetime of the process.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    numa_node = bind_to_gpu_node(0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0")))

    import numpy as np
    import torch
    from popdel_amd import capi, distributed, synth

    global LANES
    LANES = max(1, args.lanes) if int(env_world or "1") == 1 else 1     # (a cohort over several ranks runs without lanes)
    t_bench0 = time.perf_counter()

    def progress(msg):
        print("[bench %7.1f s] %s" % (time.perf_counter() - t_bench0, msg), file=sys.stderr, flush=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    is_dist = world > 1
    dist = None
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

    def sync():
        torch.cuda.synchronize()
        if is_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if not is_dist:
            return seconds
        tt = torch.tensor([seconds], device=red_device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    hip = capi.load_hip()                     # no fallback: raises if the HIP library is missing
    lib = synth.Library()
    rg = synth.conditioned_readgroup(lib)
    S = args.samples if not is_dist else args.samples // 256 * 256     # whole 256-sample tiles per rank (pd_shard_samples)
    first, W, offsets, arrays, planted = device_tile(S, 20260 + rank, device, args.coverage, deletion_seed=20260)
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
    # further resident segments of the same cohort (N = 1, one read group per sample): other pairs, another deletion
    handles, planted_all = [handle], [planted]
    if not is_dist and R == 1 and args.contexts == 1:
        for k in range(1, max(1, args.resident_segments)):
            # one deletion per 250 kb: of five consecutive 200-kb segments the third has none
            n_del = 0 if k % 5 == 2 else 1
            f_k, W_k, o_k, a_k, p_k = device_tile(S, 20260 + 1000 * k, device, args.coverage, n_deletions=n_del)
            handles.append(ctx.stage(capi.Segment(f_k, W_k, o_k, a_k["dev"], a_k["start"], a_k["on"], a_k["off"])))
            planted_all.append(p_k if n_del else None)
            del o_k, a_k
    lanes = [Lane(ctx, handles, args.serial)]
    for _ in range(max(0, args.contexts - 1)):   # optional: independent segments in flight on the same GPU
        c2 = capi.Context(hip, [rg] * (R * S), sample_offsets, device=local_rank)
        lanes.append(Lane(c2, c2.stage(seg), args.serial))

    total_passes = args.passes or -(-CHR20_WINDOWS // W)
    # what one launch of the scan has to read, averaged over the resident segments (a pass each; also warms them up)
    seg_sizes = []
    for h in handles:
        ctx.collect(ctx.run(h), copy=False)
        t = ctx.timing()
        seg_sizes.append((t["n_active_pairs"], t["n_pair_chunks"], t["n_candidate_cells"]))
    # The scan's launch duration with the kernel ALONE on the chip (HIP events around it on its stream, one ticket at a time):
    # what the roofline of the dominant kernel is quoted on. In the timed region below a context with lanes runs other
    # segments' stage 2 -- and scans -- beside it, which stretches every launch while the job as a whole gets faster; that
    # figure is reported next to it (roofline.mean_launch_ms_in_timed_region).
    scan_alone = []
    for _ in range(3 if len(handles) > 1 else 12):
        for h in handles:
            ctx.collect(ctx.run(h), copy=False)
            scan_alone.append(ctx.timing()["stage1_ms"])
    scan_alone_ms = sum(scan_alone[len(handles):]) / max(1, len(scan_alone) - len(handles))

    def timed_steps(step_fn, warmup, steps):
        for _ in range(warmup):
            step_fn()
        sync()
        t0 = time.perf_counter()
        tot = new_acc()
        for _ in range(steps):
            a, b, c, d, tinfo, x = step_fn()
            tot[0] += a
            tot[1] += b
            tot[2] += c
            tot[3] += d
            tot[4] = tinfo
            tot[5] += x
        sync()
        return max_over_ranks(time.perf_counter() - t0), tot

    # ---- one context per rank over a chr20-sized set of segments: the N = 1 headline, the segment split for N > 1 -----
    # weak scaling: the global job is world x total_passes segments dealt round-robin -> total_passes per rank
    my_passes = len(distributed.shard_segments(total_passes * world, rank, world))
    if not is_dist:
        elapsed, tot = timed_steps(lambda: run_lanes(lanes, my_passes), args.warmup, args.steps)
        steps_done = args.steps
    else:
        elapsed, tot = timed_steps(lambda: run_lanes(lanes, my_passes), 1, 1)
        steps_done = 1
    seg_value = total_passes * W * S * world * steps_done / elapsed
    seg_split = dict(value=seg_value, unit="window*sample/s", ms_per_step=elapsed / steps_done * 1e3, steps=steps_done,
                     samples_per_gpu=S, all_gathers_per_pass=0,
                     parallelism="%d rank(s), each a chr20-sized set of segments of its own %d-sample cohort "
                                 "(no collective in the data path)" % (world, S))
    s1_ms, s2_ms, n_calls, n_cand, tinfo, _ = tot
    headline = dict(value=seg_value, elapsed=elapsed, steps=steps_done, s1_ms=s1_ms, s2_ms=s2_ms, n_calls=n_calls, n_cand=n_cand,
                    tinfo=tinfo, launches=my_passes * steps_done, exchange=dict(all_gathers_per_pass=0, all_gather_bytes_per_pass=0),
                    parallelism=seg_split["parallelism"], cohort=S, fallback=None)

    result_lock = threading.Lock()
    emitted = []

    def emit(line):
        with result_lock:                      # exactly one line, whoever gets here first (watchdog or main thread)
            if emitted:
                return False
            emitted.append(True)
            if rank == 0:
                print(json.dumps(line), file=real_stdout, flush=True)
            return True

    def assemble(legs):
        h = headline
        tinfo = dict(h["tinfo"])
        if not is_dist and len(seg_sizes) > 1:      # the passes took the resident segments in turn
            tinfo["n_active_pairs"] = sum(x[0] for x in seg_sizes) / len(seg_sizes)
            tinfo["n_pair_chunks"] = sum(x[1] for x in seg_sizes) / len(seg_sizes)
        s1_shared_ms = h["s1_ms"] / h["launches"]
        # (N > 1: the headline's context is another one -- sharded, no lanes -- and its scans run alone anyway)
        s1_mean_s = (scan_alone_ms if not is_dist else s1_shared_ms) / 1e3
        cells = tinfo["n_cells"]
        # gather of the [samples x windows x pairs] SoA + cell index + Q3 out -- the CONTRACT's bytes per cell, which
        # `value`, whole_path_frac and contract_frac are quoted on
        alg_bytes = 4 * tinfo["n_active_pairs"] + 4 * cells + 4 * cells
        # What the scan's own algorithm has to move (round 5: the screen looks at every pair CHUNK once instead of once per window
        # it is active in): 4 B per chunk + 1 B of active count in and 1 B of flags out per cell (the slot index is touched at the
        # block borders and by the rare chunks beyond the threshold only). This is what `achieved` / `frac` are quoted on -- a true fraction of the HBM peak, to be read next to the
        # counters' `traffic`; in the contract's accounting the kernel would stand ABOVE the peak (contract_frac), which only says
        # that the expanded gather the survey priced is not what is read.
        compact_bytes = 4 * tinfo["n_pair_chunks"] + 1 * cells + 1 * cells
        achieved = compact_bytes / s1_mean_s / 1e9
        # HBM bytes per launch and SQ counters from rocprofv3 --pmc passes on exactly this launch shape and THIS build of
        # the scan kernel (the summary carries a hash of the kernel's sources); dropped when either does not match.
        traffic = traffic_src = valu = None
        stamp = scan_source_stamp()
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            if not (name.endswith("_pmc_stage1.json")):
                continue
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            if pmc.get("config") == {"samples": h["cohort_local"], "windows": W} and pmc.get("scan_source_stamp") == stamp:
                traffic, traffic_src = pmc["hbm_bytes_per_launch"], "profiles/" + name
                valu = pmc.get("valu_issue_utilisation")
                break
        roofline = dict(bound="hbm", kernel="screenKernel", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=traffic_src,
                        accounting="achieved = the bytes the scan's algorithm must read and write per launch (4 B per pair chunk -- "
                                   "each chunk is read ONCE --, 1 B of active count and 1 B of flags per cell: "
                                   "algorithmic_bytes_per_launch) / mean launch time; `traffic` = the PMC counters' HBM bytes of the "
                                   "same launch. SURVEY 8(d)'s contract figure (4 B per active pair of the expanded [samples x windows "
                                   "x pairs] gather + 8 B per cell: contract_bytes_per_launch) prices every pair once per window it is "
                                   "active in; against it the kernel stands at contract_frac of the peak, i.e. above it",
                        limiter="hbm", valu_issue_utilisation=valu,
                        traffic_GBs=(traffic / s1_mean_s / 1e9) if traffic else None,
                        traffic_frac=(traffic / s1_mean_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                        scan_source_stamp=stamp,
                        algorithmic_bytes_per_launch=compact_bytes,
                        contract_bytes_per_launch=alg_bytes, contract_achieved=alg_bytes / s1_mean_s / 1e9,
                        contract_frac=alg_bytes / s1_mean_s / 1e9 / HBM_PEAK_GBS,
                        mean_launch_ms=s1_mean_s * 1e3, mean_launch_ms_in_timed_region=s1_shared_ms,
                        launch_timing="HIP events around the kernel on its stream inside bench.py: mean_launch_ms with one ticket "
                                      "at a time (the kernel alone on the chip; %d launches over the resident segments ahead of the timed "
                                      "region), mean_launch_ms_in_timed_region over the timed passes, where the context's %d lanes run other "
                                      "segments' kernels beside it" % (max(1, len(scan_alone) - len(handles)), LANES),
                        mean_active_pairs_per_cell=tinfo["n_active_pairs"] / cells,
                        mean_chunks_per_cell=tinfo["n_pair_chunks"] / cells)
        steps = h["steps"]
        line = {
            "metric": "window x sample likelihood evals/sec", "value": h["value"], "unit": "window*sample/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": h["elapsed"] / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32 scan + f64 likelihoods",
            "data": "synthetic",
            "config": {"workload": ("10,000 synthetic profiles x chr20 windows on MI355X (configs[2])" if h["cohort"] == 10000 and not args.passes
                                    else "%d synthetic profiles x %d windows" % (h["cohort"], total_passes * W)),
                       "samples": h["cohort"], "samples_per_gpu": h["cohort_local"], "windows_per_step": h["windows_per_step"],
                       "segment_windows": W, "segment_passes_per_step": total_passes,
                       "resident_segment": ("%d distinct 200-kb segments of all samples, resident and scanned in turn" % len(handles))
                                           if (not is_dist and len(handles) > 1) else "one 200-kb segment of all samples, re-scanned",
                       "planted_deletion": planted_all if (not is_dist and len(handles) > 1) else planted,
                       "candidate_cells_per_segment": [x[2] for x in seg_sizes] if not is_dist else None,
                       "parallelism": h["parallelism"],
                       "layout": "pair chunks [slot][chunk] of {dev16, off16} (4 B) + per-window slot index; "
                                 "flags [sample][window], Q3 in the flagged windows; stage 2 dense image [cell][64-sample wave tile][row][lane]"},
            "exchange": h["exchange"],
            "stage1_ms_per_step": h["s1_ms"] / steps, "stage2_ms_per_step": h["s2_ms"] / steps,
            "calls_per_step": h["n_calls"] / steps, "candidate_cells_per_step": h["n_cand"] / steps,
            "max_life": tinfo["max_life"], "contexts_per_gpu": args.contexts, "pipelined_tickets": not args.serial,
            "lanes_per_context": LANES, "tickets_in_flight": (1 if args.serial else (2 * LANES if LANES > 1 else 2)),
            "host_numa_node": numa_node,        # (rank 0's: the process runs on its GPU's node, bind_to_gpu_node())
            # host arrays -> chunked layout -> HBM for one segment of all samples; with it inside every pass the rate
            # would be cells_per_pass / (stage_seconds + pass time): reported, never `value`
            "staging_seconds_per_segment": stage_seconds,
            "pcie_inclusive_value": (W * S) / (stage_seconds + seg_split["ms_per_step"] / 1e3 / max(1, my_passes)),
            "roofline": roofline,
            # the WHOLE hot path against the roofline SURVEY 8(d) states for it: value x algorithmic bytes per unit / HBM peak
            # per GPU (the scan's own figure is roofline.frac)
            "whole_path_frac": h["value"] * (alg_bytes / cells) / (HBM_PEAK_GBS * 1e9 * world),
            "algorithmic_bytes_per_unit": alg_bytes / cells}
        if h["fallback"]:
            line["headline_fallback"] = h["fallback"]
        line.update(legs)
        return line

    headline["cohort_local"] = S
    headline["windows_per_step"] = total_passes * W * world
    progress("headline (one context per rank) measured")
    legs = {}

    # ---- N > 1 headline: ONE cohort of world x S samples sharded by samples (collectives in the data path) ------------
    if is_dist:
        legs["segment_sharded"] = seg_split

        wd_state = {"headline_done": False}

        def give_up():
            if wd_state["headline_done"]:
                # the headline stands; a configs3 / configs4 leg hangs (its collectives no longer match across the ranks):
                # the line is printed with the note, and the process says so with its exit code
                legs.setdefault("configs_legs_timeout", "a configs3/configs4 leg did not finish within %.0f s" % args.sharded_timeout)
                if emit(assemble(legs)):
                    sys.stdout.flush()
                    os._exit(4)
                return
            headline["fallback"] = ("value is the SEGMENT split: the sample-sharded headline did not finish within %.0f s"
                                    % args.sharded_timeout)
            if emit(assemble(legs)):
                sys.stdout.flush()
                os._exit(3)

        def watchdog():
            t = threading.Timer(args.sharded_timeout, give_up)
            t.daemon = True
            t.start()
            return t

        dog = watchdog()
        try:
            native = args.exchange == "native" or (args.exchange == "auto" and args.backend == "nccl")

            def open_shard(use_native):
                sh = distributed.SampleShard(hip, [rg] * (R * S * world), list(range(0, R * S * world + 1, R)), rank, world,
                                             device=local_rank, native=use_native)
                h = sh.ctx.stage(seg)
                ln = Lane(sh.ctx, h, args.serial)
                ln.passes(2, new_acc())                # first exchanges (communicator set-up) outside the warm-up
                return sh, h, ln

            # With --exchange auto a failure of the library's own RCCL exchange (set-up or first passes) on ANY rank sends
            # all ranks to the host-callback exchange over torch.distributed; the ranks agree on that with one all-reduce.
            opened, exchange_note = None, None
            try:
                opened = open_shard(native)
                failed = 0
            except Exception as e:
                failed, exchange_note = 1, "%s: %s" % (type(e).__name__, e)
            if native and args.exchange == "auto":
                flag = torch.tensor([failed], device=red_device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()):
                    if opened is not None:
                        opened[0].close()
                    native = False
                    opened = open_shard(False)
                    exchange_note = "library-native RCCL exchange failed on a rank (%s); host callbacks used" % exchange_note
            elif failed:
                raise RuntimeError(exchange_note)
            shard, h_sh, lane = opened

            def step():
                acc = new_acc()
                lane.passes(total_passes, acc)
                return tuple(acc)

            for _ in range(args.warmup):
                step()
            el, tot = timed_steps(step, 0, args.steps)
            passes_timed = total_passes * args.steps
            headline.update(value=total_passes * W * S * world * args.steps / el, elapsed=el, steps=args.steps, s1_ms=tot[0],
                            s2_ms=tot[1], n_calls=tot[2], n_cand=tot[3], tinfo=tot[4], launches=passes_timed, cohort=S * world,
                            windows_per_step=total_passes * W,
                            exchange=dict(all_gathers_per_pass=tot[4]["n_all_gathers"], all_gather_bytes_per_pass=tot[4]["all_gather_bytes"],
                                          exchange_ms_per_pass=tot[5] / passes_timed,
                                          backend="rccl (library-native, ncclAllGather on the context's stream)" if native else dist.get_backend(),
                                          note=exchange_note,
                                          what="window flags, Q3 of flagged windows, a5 counts, per-iteration tile sums + read "
                                               "group 0's record, LR sums, support-percentile digit histograms"),
                            parallelism="ONE cohort of %d samples sharded by samples over %d ranks (%d each); every rank scans "
                                        "its own samples, stage 2 all-gathers per-tile sums (bit-identical to one context)"
                                        % (S * world, world, S))
            shard.ctx.unstage(h_sh)
            shard.close()
            wd_state["headline_done"] = True
            dog.cancel()
            dog = watchdog()

            # ---- the shapes BASELINE names for several GPUs: a cohort of FIXED total size sharded by samples -------------
            def fixed_cohort(total, seed):
                """this rank's part (pd_shard_samples(total, world) samples, the last ranks fewer) of a `total`-sample cohort"""
                per = hip.pd_shard_samples(total, world)
                own = max(0, min(per, total - rank * per))
                if per * (world - 1) >= total:
                    raise RuntimeError("%d samples cannot be split over %d contexts in whole 256-sample tiles" % (total, world))
                f3, W3, o3, a3, _ = device_tile(own, seed + rank, device, args.coverage, deletion_seed=seed)
                sh3 = distributed.SampleShard(hip, [rg] * total, list(range(total + 1)), rank, world, device=local_rank, native=native)
                return sh3, capi.Segment(f3, W3, o3, a3["dev"], a3["start"], a3["on"], a3["off"]), own, per

            def configs3_leg():
                total = args.configs3_samples
                sh3, seg3, own, per = fixed_cohort(total, 31000)
                n_pass = args.configs_passes or total_passes
                ln = Lane(sh3.ctx, sh3.ctx.stage(seg3), args.serial)
                ln.passes(3, new_acc())
                sync()
                q0 = time.perf_counter()
                acc = new_acc()
                ln.passes(n_pass, acc)
                sync()
                dt = max_over_ranks(time.perf_counter() - q0)
                sh3.ctx.unstage(ln.handle)
                sh3.close()
                return dict(workload="%d synthetic profiles x chr20 windows sharded by samples over %d GPUs (configs[3] per "
                                     "chromosome; resident segment re-scanned)" % (total, world),
                            value=total * seg3.n_windows * n_pass / dt, unit="window*sample/s", samples=total,
                            samples_per_rank=per, samples_last_rank=total - per * (world - 1), passes=n_pass,
                            ms_per_pass=dt / n_pass * 1e3, stage1_ms_per_pass=acc[0] / n_pass, stage2_ms_per_pass=acc[1] / n_pass,
                            exchange_ms_per_pass=acc[5] / n_pass, all_gathers_per_pass=acc[4]["n_all_gathers"],
                            all_gather_bytes_per_pass=acc[4]["all_gather_bytes"], calls_per_pass=acc[2] / n_pass, scaling="strong")

            def configs4_leg():
                total = args.configs4_samples
                sh4, seg4, own, per = fixed_cohort(total, 41000)
                c4 = sh4.ctx
                n_seg = max(2, args.configs4_segments)
                # every segment comes from host memory: staged over the copy stream while the previous one computes
                h_cur = c4.stage(seg4)
                c4.collect(c4.run(h_cur), copy=False)               # first staging (page locking) and exchanges outside the clock
                sync()
                q0 = time.perf_counter()
                t_cur = c4.run(h_cur)
                xms = gathers = 0
                for i in range(n_seg):
                    last = i + 1 == n_seg
                    if not last:
                        h_nxt = c4.stage(seg4)
                        t_nxt = c4.run(h_nxt)
                    c4.collect(t_cur, copy=False)
                    t4 = c4.timing()
                    xms += t4["exchange_ms"]
                    gathers = t4["n_all_gathers"]
                    c4.unstage(h_cur)
                    if not last:
                        h_cur, t_cur = h_nxt, t_nxt
                sync()
                dt = max_over_ranks(time.perf_counter() - q0)
                sh4.close()
                return dict(workload="%d synthetic profiles sharded by samples over %d GPUs, every segment streamed host -> HBM "
                                     "double buffered (configs[4] shape)" % (total, world),
                            value=total * seg4.n_windows * n_seg / dt, unit="window*sample/s", samples=total, samples_per_rank=per,
                            segments=n_seg, seconds_per_segment=dt / n_seg, exchange_ms_per_pass=xms / n_seg,
                            all_gathers_per_pass=gathers, host_bytes_per_segment_per_rank=int(16 * len(seg4.dev)), scaling="strong")

            # (collective legs: every rank runs them; a failure on one rank fails them on all through the watchdog)
            for name, fn, on in (("configs3", configs3_leg, args.configs3_samples > 0), ("configs4", configs4_leg, args.configs4_samples > 0)):
                if not on or args.no_legs:
                    continue
                try:
                    legs[name] = fn()
                except Exception as e:
                    legs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                    break               # the ranks are no longer in step: no further collective legs
        except Exception as e:
            headline["fallback"] = "value is the SEGMENT split: the sample-sharded headline failed: %s: %s" % (type(e).__name__, e)
        dog.cancel()
        if headline["fallback"]:
            if emit(assemble(legs)):
                sys.stdout.flush()
                os._exit(3)
            return

    # ---- secondary legs (rank 0's own clock, no collectives; a failing leg is reported, never fatal) -------------------
    def leg(name, fn):
        progress("leg %s ..." % name)
        try:
            legs[name] = fn()
        except Exception as e:
            legs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        progress("leg %s done" % name)

    def stream_through(c, sg, n_segments, warm=False):
        """`n_segments` times the segment `sg` from host memory through context `c`, double buffered: stage(N + 1) -- packing on
        the host threads, upload and layout build on the copy stream -- while ticket N computes; collect(N). Returns the
        wall time and what pd_last_stage_timing() said about the stagings. warm: one untimed segment first (a fresh context
        allocates and page-locks its staging buffers on the first one)."""
        cold = None
        if warm:
            # two segments staged side by side, as the double-buffered loop below holds them, then given back: the context
            # keeps their device buffers for re-use. The FIRST stagings of a context allocate the layout's 3.3 GB per
            # 10,000-sample segment, which costs 0-140 ms per staging depending on the box (fresh VRAM) -- round 3/4's second
            # "mode" of this leg was these two cold stagings averaged over its four segments.
            a = time.perf_counter()
            h0 = c.stage(sg)
            h1 = c.stage(sg)
            cold = (time.perf_counter() - a) / 2
            c.collect(c.run(h0), copy=False)
            c.collect(c.run(h1), copy=False)
            c.unstage(h0)
            c.unstage(h1)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        t_stage = t_collect = 0.0
        st_acc = dict(pack_seconds=0.0, uploaded_seconds=0.0, counted_seconds=0.0, total_seconds=0.0, bytes_uploaded=0, n=0)

        def note():
            st = c.stage_timing()
            st_acc["pack_seconds"] += st["pack_seconds"]
            st_acc["uploaded_seconds"] += st["uploaded_seconds"]
            st_acc["counted_seconds"] += st["counted_seconds"]
            st_acc["total_seconds"] += st["total_seconds"]
            st_acc["bytes_uploaded"] += st["bytes_uploaded"]
            st_acc["flavour"] = st["flavour"]
            st_acc["host_threads"] = st["host_threads"]
            st_acc["n"] += 1

        h_cur = c.stage(sg)
        note()
        t_first = time.perf_counter() - q0
        t_cur = c.run(h_cur)
        for i in range(n_segments):
            last = i + 1 == n_segments
            if not last:
                a = time.perf_counter()
                h_nxt = c.stage(sg)
                note()
                t_nxt = c.run(h_nxt)
                t_stage += time.perf_counter() - a
            a = time.perf_counter()
            c.collect(t_cur, copy=False)
            c.unstage(h_cur)
            t_collect += time.perf_counter() - a
            if not last:
                h_cur, t_cur = h_nxt, t_nxt
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        k = max(1, st_acc["n"])
        return dt, dict(seconds_per_segment=dt / n_segments, cold_stage_seconds_per_segment=cold,
                        stage_seconds_per_segment=t_stage / max(1, n_segments - 1),
                        collect_seconds_per_segment=t_collect / n_segments, first_stage_seconds=t_first,
                        pack_seconds_per_segment=st_acc["pack_seconds"] / k, uploaded_seconds_per_segment=st_acc["uploaded_seconds"] / k,
                        # (the two waits for the device inside a stage call and its end, all from its start: the last uploaded byte
                        # there, the totals of the counting pass back, return)
                        counted_seconds_per_segment=st_acc["counted_seconds"] / k, stage_call_seconds_per_segment=st_acc["total_seconds"] / k,
                        h2d_bytes_per_segment=st_acc["bytes_uploaded"] // k,
                        # the link's rate over the window in which bytes were travelling (packing of later slabs included: the
                        # upload of a slab runs under the packing of the next)
                        h2d_GBs_while_staging=st_acc["bytes_uploaded"] / max(1e-9, st_acc["uploaded_seconds"]) / 1e9,
                        bytes_per_instance_on_the_link=st_acc.get("flavour"), host_threads=st_acc.get("host_threads"))

    def streaming_leg():
        # every segment comes from host memory -- layout built, uploaded over the copy stream while the previous segment
        # computes, results collected -- i.e. what a host that has the pair instances in RAM sustains through one GPU.
        # What bounds it is the packing pass on the host threads (16 B per instance read from the caller's pageable arrays,
        # 5 B written to page-locked staging): the probe beside it says what the host's memory system gives those threads in
        # this very run -- rounds 3/4 saw this leg in two modes (0.03 and 0.1 s per segment) on the same build, box to box.
        probe = [hip.pd_host_copy_gbs(1 << 30)]
        thr0 = cfs_throttle()
        dt, d = stream_through(ctx, seg, args.stream_segments, warm=True)
        thr1 = cfs_throttle()
        probe.append(hip.pd_host_copy_gbs(1 << 30))
        if thr0 and thr1:
            # the packing pass runs 2 threads per usable CPU: on a box that enforces the job's CPU quota per period the process is
            # stopped for the rest of the period once the quota is burnt -- the second "mode" of this leg
            d.update(cfs_periods_throttled=thr1[0] - thr0[0], cfs_seconds_throttled=thr1[1] - thr0[1])
        d.update(value=args.stream_segments * W * S / dt, unit="window*sample/s", segments=args.stream_segments,
                 host_bytes_per_segment=int(16 * len(arrays["dev"]) + 8 * (S + 1)),
                 host_pack_read_GBs=16 * len(arrays["dev"]) / max(1e-9, d["pack_seconds_per_segment"]) / 1e9,
                 host_copy_probe_GBs=probe,
                 host_copy_probe="pageable -> page-locked copy of 1 GiB on the library's host threads, before and after the leg")
        return d

    def streaming_50000_leg():
        # configs[4]'s mechanism at its cohort size on ONE GPU: 50,000 samples x one 200-kb segment (10^9 pair instances,
        # 16 GB in the caller's arrays, 5 GB on the link) streamed from host memory, double buffered
        n50 = args.stream50k_samples
        if n50 <= 0:
            return {"skipped": "--stream50k-samples 0"}
        f5, W5, o5, a5, _ = device_tile(n50, 777, device, args.coverage)
        c5 = capi.Context(hip, [rg] * n50, list(range(n50 + 1)), device=local_rank)
        try:
            sg5 = capi.Segment(f5, W5, o5, a5["dev"], a5["start"], a5["on"], a5["off"])
            dt, d = stream_through(c5, sg5, args.stream50k_segments, warm=True)
            t5 = c5.timing()
            d.update(value=args.stream50k_segments * W5 * n50 / dt, unit="window*sample/s", samples=n50, segments=args.stream50k_segments,
                     pair_instances_per_segment=int(len(a5["dev"])), host_bytes_per_segment=int(16 * len(a5["dev"]) + 8 * (n50 + 1)),
                     host_pack_read_GBs=16 * len(a5["dev"]) / max(1e-9, d["pack_seconds_per_segment"]) / 1e9,
                     scan_ms=t5["stage1_ms"], stage2_ms=t5["stage2_ms"], candidate_cells=t5["n_candidate_cells"])
        finally:
            c5.close()
            del a5
        return d

    def concurrent_leg():
        # the same step with several independent segments in flight on the GPU, one context and host thread each
        # (plain contexts, no lanes: what a context with three lanes is measured against)
        os.environ["POPDEL_HIP_LANES"] = "1"
        more = []
        try:
            for _ in range(args.concurrent):
                c2 = capi.Context(hip, [rg] * (R * S), sample_offsets, device=local_rank)
                more.append(Lane(c2, c2.stage(seg), args.serial, depth=2))
        finally:
            os.environ["POPDEL_HIP_LANES"] = str(max(1, args.lanes))
        all_lanes = more
        run_lanes(all_lanes, max(len(all_lanes), my_passes // 8))          # warm-up of the new contexts
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        run_lanes(all_lanes, my_passes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        for l in more:
            l.ctx.unstage(l.handle)
            l.ctx.close()
        return dict(value=total_passes * W * S / dt, unit="window*sample/s", contexts_per_gpu=len(all_lanes),
                    ms_per_pass=dt / my_passes * 1e3, steps=1)

    def distinct_segments_leg():
        # the headline's step, but the passes take several DIFFERENT resident segments in turn (other pairs, other deletions
        # -- length log-uniform 300..5,000 decides how many windows and candidate cells a deletion yields --, one segment
        # without any): what a chromosome's worth of segments costs, where the headline re-scans one of them
        c2 = capi.Context(hip, [rg] * S, list(range(S + 1)), device=local_rank)
        hs, info = [], []
        for k in range(args.distinct_segments):
            n_del = 0 if k == args.distinct_segments - 1 and args.distinct_segments > 1 else 1
            f_k, W_k, o_k, a_k, p_k = device_tile(S, 20260 + 1000 * (k + 1), device, args.coverage, n_deletions=n_del)
            hs.append(c2.stage(capi.Segment(f_k, W_k, o_k, a_k["dev"], a_k["start"], a_k["on"], a_k["off"])))
            c2.collect(c2.run(hs[-1]), copy=False)
            t = c2.timing()
            info.append(dict(del_len=(p_k["del_len"] if n_del else None), af=(round(p_k["af"], 4) if n_del else None),
                             candidate_cells=t["n_candidate_cells"]))
            del o_k, a_k
        l2 = Lane(c2, hs, args.serial)
        run_lanes([l2], max(len(hs), total_passes // 8))
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        acc = run_lanes([l2], total_passes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        for hh in hs:
            c2.unstage(hh)
        c2.close()
        return dict(workload="%d samples x chr20 windows as %d passes over %d distinct resident segments in turn" % (S, total_passes, len(hs)),
                    value=total_passes * W * S / dt, unit="window*sample/s", ms_per_pass=dt / total_passes * 1e3,
                    stage1_ms_per_pass=acc[0] / total_passes, stage2_ms_per_pass=acc[1] / total_passes,
                    candidate_cells_per_pass=acc[3] / total_passes, calls_per_pass=acc[2] / total_passes, segments=info)

    def one_segment_leg():
        # the headline of rounds 1-3: the step as passes over ONE resident segment (615-bp deletion, 74 candidate cells),
        # same context
        l1 = Lane(ctx, handles[0], args.serial)
        run_lanes([l1], max(4, total_passes // 8))
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        acc = run_lanes([l1], total_passes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        per_unit = 4.0 * seg_sizes[0][0] / (W * S) + 8
        return dict(workload="%d samples x chr20 windows as %d passes over one resident segment (the headline of rounds 1-3)" % (S, total_passes),
                    value=total_passes * W * S / dt, unit="window*sample/s", ms_per_pass=dt / total_passes * 1e3,
                    whole_path_frac=total_passes * W * S / dt * per_unit / (HBM_PEAK_GBS * 1e9),
                    stage1_ms_per_pass=acc[0] / total_passes, stage2_ms_per_pass=acc[1] / total_passes,
                    candidate_cells_per_pass=acc[3] / total_passes, calls_per_pass=acc[2] / total_passes, planted_deletion=planted)

    def plain_context_leg():
        # the same step through a context WITHOUT lanes (POPDEL_HIP_LANES=1: the context of rounds 1-3, two tickets in flight)
        os.environ["POPDEL_HIP_LANES"] = "1"
        try:
            c1 = capi.Context(hip, [rg] * S, list(range(S + 1)), device=local_rank)
        finally:
            os.environ["POPDEL_HIP_LANES"] = str(max(1, args.lanes))
        # (its own staging of the same segments would double the resident set: it takes the first four)
        hs = []
        for k in range(min(4, len(handles))):
            n_del = 0 if k % 5 == 2 else 1
            f_k, W_k, o_k, a_k, _ = device_tile(S, 20260 + 1000 * k, device, args.coverage, n_deletions=n_del, **({"deletion_seed": 20260} if k == 0 else {}))
            hs.append(c1.stage(capi.Segment(f_k, W_k, o_k, a_k["dev"], a_k["start"], a_k["on"], a_k["off"])))
            del o_k, a_k
        l1 = Lane(c1, hs, args.serial, depth=2)
        n_pass = max(8, total_passes // 2)
        run_lanes([l1], max(len(hs), n_pass // 8))
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        acc = run_lanes([l1], n_pass)
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        for hh in hs:
            c1.unstage(hh)
        c1.close()
        return dict(workload="%d passes over the first %d of the headline's segments, one context without lanes, two tickets in flight" % (n_pass, len(hs)),
                    value=n_pass * W * S / dt, unit="window*sample/s", ms_per_pass=dt / n_pass * 1e3,
                    stage1_ms_per_pass=acc[0] / n_pass, stage2_ms_per_pass=acc[1] / n_pass, candidate_cells_per_pass=acc[3] / n_pass)

    def small_cohort_leg():
        # BASELINE configs[1]: 1,000 synthetic profiles x chr20 windows on one GPU
        n = min(args.small_cohort, S)
        o, a = take_samples(offsets, arrays, n * R)
        c2 = capi.Context(hip, [rg] * (R * n), sample_offsets[: n + 1], device=local_rank)
        l2 = Lane(c2, c2.stage(capi.Segment(first, W, o, a["dev"], a["start"], a["on"], a["off"])), args.serial)
        run_lanes([l2], max(2, total_passes // 8))
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        acc = run_lanes([l2], total_passes)
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        c2.unstage(l2.handle)
        c2.close()
        return dict(workload="%d synthetic profiles x chr20 windows (configs[1])" % n, value=total_passes * W * n / dt,
                    unit="window*sample/s", ms_per_pass=dt / total_passes * 1e3, stage1_ms_per_pass=acc[0] / total_passes,
                    stage2_ms_per_pass=acc[1] / total_passes, calls_per_pass=acc[2] / total_passes)

    def dense_leg():
        # candidate-dense segment (one deletion per 20 kb): stage 2 dominates on the CPU (7x stage 1)
        f2, W2, o2, a2, pl2 = device_tile(S, 777 + rank, device, args.coverage, n_deletions=args.dense_deletions)
        c2 = capi.Context(hip, [rg] * S, list(range(S + 1)), device=local_rank)
        l2 = Lane(c2, c2.stage(capi.Segment(f2, W2, o2, a2["dev"], a2["start"], a2["on"], a2["off"])), args.serial)
        n_pass = 12
        run_lanes([l2], max(2, l2.depth + 1))       # (every ticket in flight has had its page-locked result block once)
        torch.cuda.synchronize()
        q0 = time.perf_counter()
        acc = run_lanes([l2], n_pass)
        torch.cuda.synchronize()
        dt = time.perf_counter() - q0
        # the same passes with the window calls' columns left in HBM and merged there (pd_calls_merge: what `popdel call`
        # does for merged output): the groups are runs of calls at most 60 windows apart, as the host's grouping would form
        # them around the planted deletions; only the merged columns cross PCIe
        merged = None
        try:
            c2.keep_results_on_device(True)
            torch.cuda.synchronize()
            q1 = time.perf_counter()
            n_groups = 0
            ticket = c2.run(l2.handle)
            for i in range(n_pass):
                nxt = c2.run(l2.handle) if i + 1 < n_pass else None
                p = c2.collect_raw(ticket)
                k = p.contents.n_calls
                wp = np.ctypeslib.as_array(p.contents.window_position, shape=(k,)) if k else np.zeros(0, np.uint32)
                cuts = np.flatnonzero(np.diff(wp.astype(np.int64)) > 60 * 30) + 1
                groups = [g.tolist() for g in np.split(np.arange(k), cuts)] if k else []
                m = c2.merge(p, groups, [g[0] for g in groups])
                n_groups = len(groups)
                c2.free_raw(p)
                ticket = nxt
            torch.cuda.synchronize()
            dm = time.perf_counter() - q1
            merged = dict(ms_per_pass=dm / n_pass * 1e3, groups_per_pass=n_groups, declined=m is None,
                          result_bytes_per_pass=int(n_groups * S * 52), full_result_bytes_per_pass=int(acc[2] / n_pass * S * 52))
        except Exception as e:
            merged = {"error": "%s: %s" % (type(e).__name__, e)}
        c2.unstage(l2.handle)
        c2.close()
        # Stage 2's dominant kernel against ITS roofline: the data-likelihood launch ahead of the EM loop takes every
        # candidate cell x every sample (the launches inside the loop take the cells still iterating: same kernel, fewer
        # units); algorithmic bytes per (cell, sample) unit in the accounting of SURVEY 8(d): 4 B per active pair (int32
        # deviation) + 4 B count in, the 24-byte likelihood triple out; lookup tables not counted (cache resident by design)
        t2 = acc[4]
        if t2 and t2.get("lik_first_units") and t2.get("lik_first_ms"):
            per_unit = 4.0 * t2["n_active_pairs"] / t2["n_cells"] + 4 + 24
            ach = t2["lik_first_units"] * per_unit / (t2["lik_first_ms"] / 1e3) / 1e9
            # HBM bytes of that launch from the PMC passes of tools/profile_dense.sh on this very shape (the largest launch of the
            # kernel in the probe = the one ahead of the loop; FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction)
            traffic2 = traffic2_src = None
            for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if not name.endswith("_dense_pmc.json"):
                    continue
                pm = json.load(open(os.path.join(ROOT, "profiles", name)))
                k = pm.get("kernels", {}).get("pd::dataLikZeroShiftDenseKernel")
                if pm.get("config") == {"samples": S, "cells": int(t2["n_candidate_cells"])} and k and "FETCH_SIZE" in k:
                    traffic2 = (2.0 * k["FETCH_SIZE"]["max"] + k.get("WRITE_SIZE", {"max": 0.0})["max"]) * 1024.0
                    traffic2_src = "profiles/" + name
                    break
            legs["roofline_stage2"] = dict(
                bound="hbm", kernel="dataLikZeroShiftDenseKernel", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                traffic=traffic2, traffic_source=traffic2_src, launch_ms=t2["lik_first_ms"], units_per_launch=t2["lik_first_units"],
                algorithmic_bytes_per_unit=per_unit, workload="candidate_dense leg (%d cells x %d samples)" % (t2["n_candidate_cells"], S),
                accounting="the data-likelihood launch ahead of the EM loop (zero reference shifts: all candidate cells x all samples, "
                           "one workgroup per window image and sample tile), HIP events on the context's stream; "
                           "bytes = 4 per active pair + 4 + 24 per (cell, sample)",
                limiter="per-lane gathers into per-sample lookup tables (two per active pair): L1 miss handling, see DESIGN.md")
        return dict(workload="%d samples x %d windows, %d planted deletions in the segment" % (S, W2, args.dense_deletions),
                    value=n_pass * W2 * S / dt, unit="window*sample/s", ms_per_pass=dt / n_pass * 1e3,
                    stage1_ms_per_pass=acc[0] / n_pass, stage2_ms_per_pass=acc[1] / n_pass,
                    candidate_cells_per_pass=acc[3] / n_pass, calls_per_pass=acc[2] / n_pass, merged_on_device=merged,
                    planted=[dict(del_len=p["del_len"], af=round(p["af"], 4)) for p in pl2])

    def cpu_leg():
        oracle_path = os.path.join(ROOT, "oracle", "_build", "libpopdel_oracle.so")
        if not os.path.exists(oracle_path):
            return None
        n_cpu = min(args.cpu_samples, S)
        o_off, o_arr = take_samples(offsets, arrays, n_cpu * R)
        oseg = capi.Segment(first, W, o_off, o_arr["dev"], o_arr["start"], o_arr["on"], o_arr["off"])
        olib = capi.load_library(oracle_path)
        octx = capi.Context(olib, [rg] * (n_cpu * R), sample_offsets[: n_cpu + 1])
        c0 = time.perf_counter()
        octx.collect(octx.submit(oseg))
        cpu_s = time.perf_counter() - c0
        octx.close()
        out = dict(value=n_cpu * W / cpu_s, unit="window*sample/s", cores=1, kind="port", cpu_model=cpu_model(),
                   host_cores=os.cpu_count(), usable_cores=usable_cores(),
                   sample="%d of the %d samples x the same %d-window segment, oracle/ (x87 long double), %.1f s"
                          % (n_cpu, S, W, cpu_s),
                   scope="hot path only: the CPU restatement of genotype_deletion_window() over the staged pair instances of the "
                         "segment (no profile reading, no window walk, no merging, no VCF). The like-for-like CPU figure of the "
                         "drop-in -- `popdel call` from profile files to VCF -- is e2e.cpu_baseline")
        # the same on all host cores the way a CPU run would use them: independent cohorts of n_cpu samples each (the
        # oracle is single-threaded; ctypes releases the GIL)
        # every core this job may use (cgroup quota / affinity; more threads than that are only throttled)
        n_thr = usable_cores() if args.cpu_threads < 0 else min(args.cpu_threads, os.cpu_count() or 1)
        if n_thr > 1 and R == 1:
            distinct = max(1, S // n_cpu)
            parts = []
            for t in range(min(distinct, n_thr)):
                lo, hi = int(offsets[t * n_cpu]), int(offsets[(t + 1) * n_cpu])
                parts.append(capi.Segment(first, W, offsets[t * n_cpu:(t + 1) * n_cpu + 1] - offsets[t * n_cpu],
                                          arrays["dev"][lo:hi], arrays["start"][lo:hi], arrays["on"][lo:hi], arrays["off"][lo:hi]))
            ctxs = [capi.Context(olib, [rg] * n_cpu, list(range(n_cpu + 1))) for _ in range(n_thr)]
            work = [threading.Thread(target=lambda c=c, p=parts[i % len(parts)]: c.collect(c.submit(p), copy=False))
                    for i, c in enumerate(ctxs)]
            c0 = time.perf_counter()
            for th in work:
                th.start()
            for th in work:
                th.join()
            cpu_all_s = time.perf_counter() - c0
            for c in ctxs:
                c.close()
            out["all_cores"] = dict(value=n_thr * n_cpu * W / cpu_all_s, unit="window*sample/s", cores=n_thr,
                                    sample="%d independent cohorts of %d samples (%d distinct), one oracle context per thread, %.1f s"
                                           % (n_thr, n_cpu, len(parts), cpu_all_s))
        return out

    def sharded_world1_leg():
        # The sample-sharded code path with ONE rank and the library's own RCCL exchange: every exchange is issued and
        # timed, nothing crosses a wire -- what the path costs by itself next to the unsharded context (same segment) --, and
        # the same with the wire time of an exchange over xGMI injected behind every one of them (a one-wave kernel that holds
        # the stream for d microseconds, POPDEL_HIP_XCH_DELAY_US): what a rank of a cohort sharded over several GPUs would see.
        # Measured in a process of its own (tools/sharded_model_probe.py), as a rank of such a cohort is one: beside this
        # process's contexts (ten resident segments, three lanes' streams) the lanes of the measured context share hardware
        # queues with them and a held stream holds its neighbours (1,280 samples, 40 us: 1.02 ms per pass in here, 0.65 alone).
        import subprocess
        sizes = ",".join(str(S if int(x) <= 0 else min(int(x), S)) for x in args.sharded_world1.split(",") if x.strip() != "")
        n_pass = max(40, total_passes // 2)
        env = dict(os.environ)
        env.pop("POPDEL_HIP_LANES", None)
        env.pop("POPDEL_HIP_XCH_DELAY_US", None)
        lanes_arg = "1" if LANES == 1 else "1,%d" % LANES
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sharded_model_probe.py"), sizes, "0,10,20,40", lanes_arg, str(n_pass)],
                           env=env, capture_output=True, text=True, timeout=args.sharded_timeout)
        probe = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not probe:
            raise RuntimeError("sharded_model_probe.py exited %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
        out = []
        for n in [int(x) for x in sizes.split(",")]:
            mine = {(p["lanes"], p["delay_us_per_exchange"]): p for p in probe if p["samples_per_rank"] == n}
            k = LANES if (LANES, 0.0) in mine else 1
            zero, d10, d20, d40 = mine[(k, 0.0)], mine[(k, 10.0)], mine[(k, 20.0)], mine[(k, 40.0)]
            row = dict(samples=n, passes=n_pass, lanes=k, unsharded_plain_ms_per_pass=zero["unsharded_one_lane_ms_per_pass"],
                       unsharded_ms_per_pass=zero["unsharded_same_lanes_ms_per_pass"], sharded_ms_per_pass=zero["modelled_ms_per_pass"],
                       exchange_ms_per_pass=zero["exchange_ms_per_pass"], all_gathers_per_pass=zero["all_gathers_per_pass"],
                       all_gather_bytes_per_pass=zero.get("all_gather_bytes_per_pass"),
                       modelled_20us_ms_per_pass=d20["modelled_ms_per_pass"], modelled_40us_ms_per_pass=d40["modelled_ms_per_pass"])
            row["modelled_10us_ms_per_pass"] = d10["modelled_ms_per_pass"]
            for kind, p in (("modelled_10us", d10), ("modelled_20us", d20), ("modelled_40us", d40)):
                row[kind + "_efficiency"] = p["modelled_efficiency_vs_same_lanes"]
                row[kind + "_efficiency_vs_plain_context"] = p["modelled_efficiency"]
            if (1, 40.0) in mine:
                row["modelled_40us_one_lane_ms_per_pass"] = mine[(1, 40.0)]["modelled_ms_per_pass"]
                row["modelled_40us_one_lane_efficiency"] = mine[(1, 40.0)]["modelled_efficiency"]
            row["overhead_ms_per_pass"] = row["sharded_ms_per_pass"] - row["unsharded_ms_per_pass"]
            row["modelled_ms_per_pass"] = row["modelled_40us_ms_per_pass"]
            row["modelled_efficiency"] = row["modelled_40us_efficiency"]
            row["modelled_efficiency_vs_plain_context"] = row["modelled_40us_efficiency_vs_plain_context"]
            out.append(row)
        return dict(what="per pass over the resident segment, in a process of its own (tools/sharded_model_probe.py): unsharded context vs. "
                         "the sample-sharded path with a communicator of ONE rank (library RCCL; exchanges issued and event-timed, no wire), and "
                         "the same with 10 / 20 / 40 us of wire time injected behind every exchange (modelled_*: the latency model of a cohort "
                         "sharded over several GPUs; modelled_efficiency = the 40-us row against the unsharded context WITH the bench's lanes, "
                         "..._vs_plain_context against an unsharded context without lanes, ..._one_lane_* the sharded context without lanes)", rows=out)

    def e2e_leg():
        # `popdel call` from profile FILES to a VCF (what a user of the reference runs): generated profiles of a 1-Mb
        # region, the product binary (HIP backend) against the oracle-backed driver (CPU; test infrastructure, here as the
        # timed baseline) on the same files -- wall time of the whole process, phase split of the product
        import re
        import shutil
        import subprocess
        import tempfile
        n = args.e2e_samples
        region, n_win = (1000000, 2000000), 33334
        d = tempfile.mkdtemp(prefix="popdel_e2e_")
        try:
            q0 = time.perf_counter()
            lst = synth.write_cohort(d, n, region=region, deletions=synth.single_deletion, workers=usable_cores())
            t_write = time.perf_counter() - q0
            product = os.path.join(ROOT, "popdel_amd", "lib", "popdel")
            oracle_bin = os.path.join(ROOT, "oracle", "_build", "popdel_oracle_call")

            def call(binary, out, threads):
                env = dict(os.environ, POPDEL_TIMING="1")
                env.pop("POPDEL_HIP_LANES", None)          # (the binary's own default: one lane)
                if threads:
                    env["POPDEL_THREADS"] = str(threads)
                q = time.perf_counter()
                # (the oracle-backed driver IS `popdel call`: it takes the arguments without the sub-command)
                sub = ["call"] if os.path.basename(binary) == "popdel" else []
                r = subprocess.run([binary] + sub + [lst, "-o", out, "-r", "chr20:%d-%d" % region], env=env, capture_output=True,
                                   text=True, timeout=600)
                wall = time.perf_counter() - q
                if r.returncode != 0:
                    raise RuntimeError("%s exited %d: %s" % (os.path.basename(binary), r.returncode, (r.stderr or r.stdout)[-300:]))
                phases = {}
                for line in r.stderr.splitlines():
                    if line.startswith("[popdel timing]"):
                        for name, val in re.findall(r"([a-z][a-z+ ]*?) ([0-9][0-9.e+-]*) s", line):
                            phases[name.strip().replace(" ", "_")] = float(val)
                        m = re.search(r"threads (\d+);", line)
                        if m:
                            phases["host_threads"] = int(m.group(1))
                        m = re.search(r"CPU time ([0-9.e+-]+) s user, ([0-9.e+-]+) s system", line)
                        if m:
                            phases["cpu_seconds"] = float(m.group(1)) + float(m.group(2))
                            phases.pop("time", None)            # (the phase pattern's reading of "CPU time x s user")
                return wall, phases

            call(product, os.path.join(d, "warm.vcf"), 0)          # page cache, GPU runtime start
            wall, phases = call(product, os.path.join(d, "gpu.vcf"), 0)
            out = dict(workload="popdel call on %d generated profiles, chr20:%d-%d (%d windows), merged VCF" % (n, region[0], region[1], n_win),
                       value=n * n_win / wall, unit="window*sample/s", wall_seconds=wall, phases_seconds=phases,
                       host_threads=phases.pop("host_threads", None), usable_cores=usable_cores(), profile_bytes=sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".profile")),
                       cohort_written_in_seconds=t_write)
            if os.path.exists(oracle_bin) and not args.no_cpu:
                cpu = {}
                for label, thr in (("all_cores", usable_cores()), ("one_thread", 1)):
                    w, _ = call(oracle_bin, os.path.join(d, "cpu_%s.vcf" % label), thr)
                    cpu[label] = dict(value=n * n_win / w, unit="window*sample/s", wall_seconds=w, cores=thr)
                strip = lambda t: "\n".join(l for l in t.splitlines() if not l.startswith("##fileDate"))
                cpu["identical_vcf"] = strip(open(os.path.join(d, "gpu.vcf")).read()) == strip(open(os.path.join(d, "cpu_all_cores.vcf")).read())
                cpu["kind"] = "port"
                out["cpu_baseline"] = cpu
            return out
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def e2e_cohort_leg():
        # `popdel call` at cohort scale, steady state: >= 4,000 generated profiles x 4 Mb (20 segments of 200 kb), the product
        # binary on all usable cores; what counts is the time per segment once the backend is up (POPDEL_TIMING's
        # "steady state" line: from the first round that began after the backend was created to the last record written),
        # the share of it the device works, and the host phases behind it
        import re
        import shutil
        import subprocess
        import tempfile
        n = args.e2e_cohort_samples
        region = (1000000, 1000000 + args.e2e_cohort_mb * 1000000)
        d = tempfile.mkdtemp(prefix="popdel_e2e_cohort_")
        try:
            # what does a profile cost to write here? (then the cohort that fits the budget: writing takes most of it)
            q0 = time.perf_counter()
            synth.write_cohort(d, min(n, 4 * usable_cores()), region=region, deletions=synth.single_deletion, workers=usable_cores())
            per = (time.perf_counter() - q0) / min(n, 4 * usable_cores())
            afford = int(0.6 * args.e2e_cohort_budget / per)
            if afford < n:
                n = max(256, afford // 256 * 256)
            q0 = time.perf_counter()
            lst = synth.write_cohort(d, n, region=region, deletions=synth.single_deletion, workers=usable_cores())
            t_write = time.perf_counter() - q0
            product = os.path.join(ROOT, "popdel_amd", "lib", "popdel")
            env = dict(os.environ, POPDEL_TIMING="1")
            env.pop("POPDEL_HIP_LANES", None)              # (the binary's own default: one lane)
            q = time.perf_counter()
            r = subprocess.run([product, "call", lst, "-o", os.path.join(d, "out.vcf"), "-r", "chr20:%d-%d" % region], env=env,
                               capture_output=True, text=True, timeout=600)
            wall = time.perf_counter() - q
            if r.returncode != 0:
                raise RuntimeError("popdel exited %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:]))
            m = re.search(r"steady state: segments (\d+) of (\d+); windows (\d+); samples (\d+); wall ([0-9.e+-]+) s; decode ([0-9.e+-]+) s; "
                          r"replay ([0-9.e+-]+) s; device kernels ([0-9.e+-]+) s", r.stderr)
            if not m:
                # (a cohort so small that every round was decoded before the GPU runtime had started)
                return dict(workload="popdel call on %d generated profiles, chr20:%d-%d" % (n, region[0], region[1]), value=None,
                            note="no round began after the backend was up: the run is over before the steady state starts",
                            whole_run_wall_seconds=wall, whole_run_value=n * (region[1] - region[0]) / 30.0 / wall)
            segs, segs_all, windows, samples = (int(x) for x in m.groups()[:4])
            t_wall, t_decode, t_replay, t_device = (float(x) for x in m.groups()[4:])
            cpu = re.search(r"CPU time ([0-9.e+-]+) s user, ([0-9.e+-]+) s system", r.stderr)
            par = re.search(r"parity: (\d+) reference-shift quotients .* in (\d+) candidate cells, (\d+) window calls", r.stderr)
            n_records = sum(1 for l in open(os.path.join(d, "out.vcf")) if not l.startswith("#"))
            # ... and the same without the ends: the first two of those rounds pay for the loaders' page-locked buffers and the
            # tables' first touch, the last ends with the drain (POPDEL_TIMING's "warmed rounds" line: begin of the third such
            # round to begin of the last) -- what a whole-genome run (15,000 rounds) sees
            mw = re.search(r"warmed rounds: segments (\d+); windows (\d+); samples (\d+); wall ([0-9.e+-]+) s; decode ([0-9.e+-]+) s; replay ([0-9.e+-]+) s",
                           r.stderr)
            warmed = {}
            if mw:
                ws, ww, wn = int(mw.group(1)), int(mw.group(2)), int(mw.group(3))
                wt = float(mw.group(4))
                warmed = dict(warmed_segments=ws, warmed_seconds_per_segment=wt / ws, warmed_value=ww * wn / wt,
                              warmed_decode_seconds_per_segment=float(mw.group(5)) / ws, warmed_replay_seconds_per_segment=float(mw.group(6)) / ws)
                mc = re.search(r"warmed rounds: .*; CPU ([0-9.e+-]+) s \(all threads\)", r.stderr)
                if mc:
                    # CPU seconds of all threads per warmed round and the CPUs that kept busy: at the job's quota the round is CPU-bound
                    warmed.update(warmed_cpu_seconds_per_segment=float(mc.group(1)) / ws, warmed_cpus_busy=float(mc.group(1)) / wt)
            # (round 6: how the round's records were filed -- from the device's block summaries, and how many start entries never
            # crossed the link -- and where the window loop took its start entries from)
            for key, mark in (("filing", "[popdel timing] filing: "), ("walk_entries", "[popdel timing] start entries of the window loop: ")):
                lines = [l for l in r.stderr.splitlines() if mark in l]
                if lines:
                    warmed[key] = lines[-1][len(mark):]
            loader = [l for l in r.stderr.splitlines() if "[popdel timing] loader:" in l]
            if loader:
                warmed["loader"] = loader[-1][len("[popdel timing] loader: "):]
                # gpu_busy_share counts the genotyping kernels of the context only; the loader's gunzip + parse kernels (all
                # rounds of the run) are the larger part of what the device does in the product
                mk = re.search(r"copies \+ kernel ([0-9.e+-]+) s \(kernel ([0-9.e+-]+) s\)", loader[-1])
                if mk:
                    warmed["loader_kernel_seconds_per_segment"] = float(mk.group(2)) / segs_all
                    warmed["gpu_busy_share_with_loader"] = t_device / t_wall + float(mk.group(2)) / segs_all / (t_wall / segs)
            return dict(warmed, workload="popdel call on %d generated profiles, chr20:%d-%d (%d segments of 200 kb), merged VCF; steady state = "
                                 "the segments whose round began after the backend was up, start-up and teardown excluded" % (n, region[0], region[1], segs_all),
                        value=windows * samples / t_wall, unit="window*sample/s", steady_segments=segs, segments=segs_all,
                        steady_seconds_per_segment=t_wall / segs, steady_wall_seconds=t_wall, steady_windows=windows, samples=samples,
                        decode_seconds_per_segment=t_decode / segs, replay_seconds_per_segment=t_replay / segs,
                        device_kernel_seconds_per_segment=t_device / segs, gpu_busy_share=t_device / t_wall,
                        whole_run_wall_seconds=wall, whole_run_value=n * (region[1] - region[0]) / 30.0 / wall,
                        cpu_seconds=(float(cpu.group(1)) + float(cpu.group(2))) if cpu else None, usable_cores=usable_cores(),
                        vcf_records=n_records, near_integer_shifts=int(par.group(1)) if par else None,
                        candidate_cells=int(par.group(2)) if par else None,
                        profile_bytes=sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.endswith(".profile")),
                        cohort_written_in_seconds=t_write)
        finally:
            shutil.rmtree(d, ignore_errors=True)

    def inflate_leg():
        # gunzip of profile blocks on the device (prototype: pd_inflate_members, one lane per gzip member) against the loader's
        # own host path (libdeflate, all usable cores) on profile-shaped members: compressed bytes over PCIe + kernel vs. host
        import gzip
        from concurrent.futures import ThreadPoolExecutor
        from popdel_amd import inflate
        distinct = min(args.inflate_members, 4000)
        blocks = inflate.profile_blocks(distinct, seed=77)
        with ThreadPoolExecutor(max_workers=usable_cores()) as ex:          # (zlib releases the GIL)
            blobs = list(ex.map(lambda b: gzip.compress(b, 6, mtime=0), blocks))
        blobs = (blobs * (-(-args.inflate_members // distinct)))[: args.inflate_members]
        data, offs, out_offs = inflate.pack_members(blobs)
        inflate.inflate_gpu(hip, data[: int(offs[64])], offs[:65], out_offs[:65], device=local_rank)
        out, status, lengths, timing = inflate.inflate_gpu(hip, data, offs, out_offs, device=local_rank, repeat=5)
        out_c, secs, which, failures = inflate.inflate_cpu(data, offs, out_offs, usable_cores())
        return dict(what="gzip members of profile index regions (level 6, ~1,000 pairs each): device = one lane per member, host = %s on %d "
                         "threads; identical output" % (which, usable_cores()),
                    members=len(blobs), distinct_members=distinct, compressed_bytes=int(offs[-1]), inflated_bytes=int(out_offs[-1]),
                    all_accepted=bool((status == 0).all()), identical_to_host=bool(failures == 0 and np.array_equal(out, out_c)),
                    kernel_ms=timing["kernel_ms"], h2d_ms=timing["h2d_ms"], d2h_ms=timing["d2h_ms"],
                    device_inflated_GBs=int(out_offs[-1]) / timing["kernel_ms"] / 1e6,
                    host_seconds=secs, host_inflated_GBs=int(out_offs[-1]) / secs / 1e9, host_library=which, host_threads=usable_cores(),
                    kernel_speedup_over_host=secs * 1e3 / timing["kernel_ms"],
                    h2d_plus_kernel_speedup_over_host=secs * 1e3 / (timing["kernel_ms"] + timing["h2d_ms"]))

    if rank == 0 and not args.no_legs and not is_dist:      # N > 1 runs carry the two splits only
        if args.stream_segments > 0:
            leg("streaming", streaming_leg)
            leg("streaming_50000", streaming_50000_leg)
        if args.sharded_world1.strip():
            leg("sharded_world1", sharded_world1_leg)
        if args.concurrent > 1 and args.contexts == 1:
            leg("concurrent_segments", concurrent_leg)
        if len(handles) > 1 and args.contexts == 1:
            leg("one_segment", one_segment_leg)
        if LANES > 1 and R == 1 and args.contexts == 1:
            leg("plain_context", plain_context_leg)
        if args.small_cohort > 0:
            leg("small_cohort", small_cohort_leg)
        if args.distinct_segments > 0 and R == 1:
            leg("distinct_segments", distinct_segments_leg)
        if args.dense_deletions > 0 and R == 1:
            leg("candidate_dense", dense_leg)
        # (last of the GPU legs: its writer processes and host threads eat into the job's CPU quota, which the streaming
        # leg's packing threads would miss)
        if args.inflate_members > 0:
            leg("device_gunzip", inflate_leg)
        if args.e2e_samples > 0:
            leg("e2e", e2e_leg)
        if args.e2e_cohort_samples > 0:
            leg("e2e_cohort", e2e_cohort_leg)
    if rank == 0:
        legs["cpu_baseline"] = None
        if not args.no_cpu and not is_dist:                 # timed on rank 0 at N = 1 only
            leg("cpu_baseline", cpu_leg)
    for l in lanes:
        for hh in l.handles:
            l.ctx.unstage(hh)
        l.ctx.close()
    emit(assemble(legs))
    if is_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
