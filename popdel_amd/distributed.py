"""Multi-GPU drivers for the hot path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Two exact ways to split the work:

* **by samples** (``SampleShard``): rank r owns a contiguous block of samples -- it parses, stages and scans only those
  -- and the library all-gathers what crosses samples between its kernels (sorted Q3 values of the flagged windows,
  integer counts, per-tile EM sums, support-pool digit histograms) through the two callbacks of ``pd_shard``. The
  callbacks are implemented here with ``torch.distributed.all_gather_into_tensor`` on the context's own HIP stream, so
  RCCL runs stream-ordered between the kernels (no host synchronisation on the nccl backend). Results are bit-identical
  to the unsharded run because every reduction adds the all-gathered per-tile values in cohort order.
* **by segments** (``run_sharded``): segments are independent once the host has resolved the carry-over of active
  pairs, so ranks can run complete segments for ALL samples with no data-path collective; only call records travel.
"""
from __future__ import annotations

import numpy as np


# ---- split by segments ------------------------------------------------------------------------------------------------
def shard_segments(n_segments, rank, world):
    """Indices of the segments rank `rank` of `world` processes."""
    return list(range(rank, n_segments, world))


def run_sharded(ctx, segments, rank=0, world=1, group=None):
    """Run `segments` (a list of capi.Segment) through `ctx` split over the ranks of `group`.

    Returns the list of per-segment call dictionaries in segment order (identical on every rank)."""
    mine = shard_segments(len(segments), rank, world)
    local = {}
    for i in mine:                       # one ticket in flight per context (see include/popdel_hip.h)
        local[i] = ctx.collect(ctx.submit(segments[i]))
    if world == 1:
        return [local[i] for i in range(len(segments))]
    import torch.distributed as dist
    gathered = [None] * world
    dist.all_gather_object(gathered, local, group=group)
    merged = {}
    for part in gathered:
        merged.update(part)
    return [merged[i] for i in range(len(segments))]


# ---- split by samples -------------------------------------------------------------------------------------------------
def sample_ranges(lib, n_samples, world):
    """[first, last) sample range of every rank under the library's partition rule (whole 256-sample tiles)."""
    per = lib.pd_shard_samples(n_samples, world)
    return [(min(n_samples, r * per), min(n_samples, (r + 1) * per)) for r in range(world)]


class SampleShardExchange:
    """Host side of ``pd_shard``: owns the exchange buffer and all-gathers regions of it over the process group.

    ``device`` is a torch device; with a CPU device (tests) the buffer is host memory and the HIP stream is ignored."""

    def __init__(self, rank, world, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.group = rank, world, group
        self.device = torch.device(device)
        self.arena = None
        self.on_gpu = self.device.type == "cuda"
        self.direct = self.on_gpu and dist.get_backend(group) == "nccl"
        self.calls = 0
        self.bytes = 0
        self.last_error = None
        self._streams = {}

    def reserve(self, nbytes):
        try:
            if self.arena is None or self.arena.numel() < nbytes:
                self.arena = None
                self.arena = self.torch.empty(int(nbytes) + int(nbytes) // 4 + 4096, dtype=self.torch.uint8, device=self.device)
            return self.arena.data_ptr()
        except Exception as e:                   # an exception must not unwind through the C caller
            self.last_error = e
            return None

    def all_gather(self, offset, per, stream):
        try:
            torch, dist = self.torch, self.dist
            region = self.arena[offset:offset + self.world * per]
            mine = region[self.rank * per:(self.rank + 1) * per]
            if self.direct:
                # RCCL, ordered on the library's stream: kernels before -> collective -> kernels after, no host wait
                ext = self._streams.get(stream)
                if ext is None:                  # one wrapper per library stream, not one per exchange (20 per segment)
                    ext = self._streams[stream] = torch.cuda.ExternalStream(stream, device=self.device)
                with torch.cuda.stream(ext):
                    dist.all_gather_into_tensor(region, mine.clone(), group=self.group)
            elif self.on_gpu:
                # a backend without device support (gloo): stage through host memory
                ext = torch.cuda.ExternalStream(stream, device=self.device)
                with torch.cuda.stream(ext):
                    host = mine.cpu()
                    out = torch.empty(self.world * per, dtype=torch.uint8)
                    dist.all_gather_into_tensor(out, host, group=self.group)
                    region.copy_(out)
                ext.synchronize()
            else:
                out = torch.empty(self.world * per, dtype=torch.uint8)
                dist.all_gather_into_tensor(out, mine.clone(), group=self.group)
                region.copy_(out)
            self.calls += 1
            self.bytes += self.world * per
            return 0
        except Exception as e:
            self.last_error = e
            return 1


class NativeExchangeStats:
    """The counters of SampleShardExchange for a context whose exchanges the library runs itself (RCCL): read from
    ``pd_timing`` after every collect (``SampleShard.run`` calls ``update``)."""

    def __init__(self):
        self.calls = 0
        self.bytes = 0
        self.exchange_ms = 0.0
        self.last_error = None

    def update(self, timing):
        self.calls += int(timing["n_all_gathers"])
        self.bytes += int(timing["all_gather_bytes"])
        self.exchange_ms += float(timing["exchange_ms"])


class SampleShard:
    """One rank's context of a cohort split by samples.

    `readgroups` / `sample_rg_offsets` describe the WHOLE cohort here (convenient for tests and the bench; a production
    loader would parse only its own profiles and exchange the per-read-group ``min_init_del_len`` values); the context is
    created for this rank's samples only."""

    def __init__(self, lib, readgroups, sample_rg_offsets, rank, world, group=None, device=0, native=False, lane_groups=None, **config):
        """native=True: the library runs the all-gathers itself with RCCL on its own stream (pd_ctx_set_shard_rccl; the
        communicator's id travels over `group` once); otherwise through the two host callbacks implemented by
        SampleShardExchange (any torch.distributed backend; what the CPU and one-GPU tests use). lane_groups: a list of
        process groups, one per lane of the context (POPDEL_HIP_LANES) -- the cohort then keeps its lanes, lane k exchanging
        over lane_groups[k] (pd_ctx_set_shard_lanes); every rank must pass groups created in the same order."""
        import torch
        from . import capi
        offs = np.asarray(sample_rg_offsets, dtype=np.int64)
        n_samples = len(offs) - 1
        self.rank, self.world, self.group = rank, world, group
        self.ranges = sample_ranges(lib, n_samples, world)
        self.first, self.last = self.ranges[rank]
        if self.first >= self.last:
            raise ValueError("rank %d owns no samples: %d samples cannot be split over %d contexts" % (rank, n_samples, world))
        self.rg_lo, self.rg_hi = int(offs[self.first]), int(offs[self.last])
        self.n_samples_global = n_samples
        self.ctx = capi.Context(lib, readgroups[self.rg_lo:self.rg_hi], offs[self.first:self.last + 1] - self.rg_lo,
                                device=device, **config)
        dev = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")
        self.native = bool(native)
        if self.native:
            import torch.distributed as dist
            box = [capi.rccl_unique_id(lib) if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            self.exchange = NativeExchangeStats()
            self.ctx.set_shard_rccl(rank, world, self.first, n_samples, [r.min_init_del_len for r in readgroups], box[0])
        elif lane_groups:
            self.exchange = SampleShardExchange(rank, world, dev, group)
            self.lane_exchanges = [SampleShardExchange(rank, world, dev, g) for g in lane_groups]
            self.ctx.set_shard_lanes(rank, world, self.first, n_samples, [r.min_init_del_len for r in readgroups],
                                     [(x.reserve, x.all_gather) for x in [self.exchange] + self.lane_exchanges])
        else:
            self.exchange = SampleShardExchange(rank, world, dev, group)
            self.ctx.set_shard(rank, world, self.first, n_samples, [r.min_init_del_len for r in readgroups],
                               self.exchange.reserve, self.exchange.all_gather)

    def local_segment(self, seg):
        """The read groups of this rank out of a whole-cohort capi.Segment."""
        from . import capi
        o = np.asarray(seg.rg_offsets, dtype=np.int64)
        a, b = int(o[self.rg_lo]), int(o[self.rg_hi])
        return capi.Segment(seg.first_window_pos, seg.n_windows, o[self.rg_lo:self.rg_hi + 1] - a, seg.dev[a:b], seg.start[a:b],
                            seg.on[a:b], seg.off[a:b], seg.window_shift)

    def run(self, local_seg):
        """Collective: every rank calls it with its part of the same segment. Returns the calls of the whole cohort with
        the per-sample columns of this rank's samples."""
        try:
            out = self.ctx.collect(self.ctx.submit(local_seg))
            if self.native:
                self.exchange.update(self.ctx.timing())
            return out
        except Exception:
            if self.exchange.last_error is not None:
                raise self.exchange.last_error
            raise

    def gather_columns(self, out):
        """All ranks' sample columns joined in cohort order (every rank gets the full result)."""
        if self.world == 1:
            return out
        parts = [None] * self.world
        self.dist_all_gather_object(parts, {k: out[k] for k in ("pl", "lad", "dad", "first_last")})
        full = dict(out)
        for k in ("pl", "lad", "dad", "first_last"):
            full[k] = np.concatenate([p[k] for p in parts], axis=1)
        full["n_samples"] = self.n_samples_global
        return full

    def dist_all_gather_object(self, parts, obj):
        import torch.distributed as dist
        dist.all_gather_object(parts, obj, group=self.group)

    def close(self):
        self.ctx.close()
