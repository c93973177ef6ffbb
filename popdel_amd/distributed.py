"""Multi-GPU driver for the hot path: one process per GPU, segments dealt round-robin to the ranks.

Segments (the windows walked by one processSegment() call) are independent once the host has resolved the carry-over
of active pairs between them, so every rank can run complete segments for ALL samples with no data-path collective;
only the (small) call records travel, gathered in segment order on every rank. This is the exact-result sharding of
SURVEY.md section 8(e) "window/segment sharding". The sample-sharded variant (per-window Q3 all-gather plus one
all-reduce of five doubles per candidate cell and EM iteration over RCCL) is not wired yet.
"""
from __future__ import annotations


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
