"""GPU: the parsed blocks summarised for the loader's tables on the device (pd_inflater_run_summarized: summarizeBlocksKernel,
inflate.hip) against the same computation on the host (pdh::summarizeRun, chrom_profile.cpp) -- per block how many records begin /
end in each 30-bp window, the extremes and the order flags: every word and every count. What ChromProfile::addBulk() makes of such
summaries is held against record-by-record filing on the CPU (tests/test_host_filing.py) and, through the product binary, in
tests/test_gpu_loader.py (POPDEL_HOST_FILING=1 against the default); the reference step they stand for is ChromosomeProfile::add
with getEndCount after every record (profile_structure_popdel_call.h:1084-1112, :870-924)."""
import gzip
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

IN_ORDER, ENDS_IN_ORDER, SATURATED, NO_ROOM = 1, 2, 4, 8


def block(begin, pos, dev):
    """one index-region block: 256-bp windows of {offset u8, deviation i32} records at the given positions"""
    blk = bytearray()
    win = (pos - begin) // 256
    for w in np.unique(win):
        sel = win == w
        rec = np.zeros(int(sel.sum()), dtype=np.dtype([("off", "u1"), ("dev", "<i4")]))
        rec["off"] = (pos[sel] - begin - w * 256).astype(np.uint8)
        rec["dev"] = dev[sel]
        blk += struct.pack("<III", 19, begin + int(w) * 256, int(sel.sum())) + rec.tobytes()
    return bytes(blk)


def compare(dev, host, dcounts, hcounts, where):
    assert dev["n"] == host["n"], where
    if host["n"] == 0:
        return
    for k in ("first_pos", "last_pos", "bin_base", "min_end_bin", "max_end_bin", "first_end_bin", "last_end_bin"):
        assert dev[k] == host[k], (where, k, dev, host)
    assert dev["flags"] & (IN_ORDER | ENDS_IN_ORDER | NO_ROOM) == host["flags"] & (IN_ORDER | ENDS_IN_ORDER | NO_ROOM), (where, dev, host)
    if host["flags"] & NO_ROOM:
        return
    assert dev["n_bins"] == host["n_bins"], where
    assert dev["flags"] == host["flags"], (where, dev, host)
    assert np.array_equal(dcounts, hcounts), where


def test_block_summaries_equal_the_host_computation(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    rng = np.random.default_rng(41)
    # blocks of everyday records (one pair per ~10 bp, deviations sd 50), then blocks with wide inserts, a pile-up on a few
    # positions (counts beyond 255 in a window), far pairs (end windows beyond the room), one-record and two-record blocks, records
    # that step back (the parser flags those blocks: nothing is summarised)
    blocks, begin = [], 3000000
    for m in range(120):
        n = int(rng.poisson(1000))
        if 100 <= m < 110:
            n = int(rng.integers(1, 4))
        pos = begin + np.sort(rng.integers(0, 10000, n))
        dev = np.round(rng.normal(0, 50 if m < 60 else 400, n)).astype(np.int32)
        if 70 <= m < 74:
            pos = begin + np.sort(rng.integers(5000, 5040, n))
        if 84 <= m < 90:
            dev[rng.integers(0, n, 3)] = 400000
        blocks.append(block(begin, pos, dev))
        begin += 10000
    blobs = [gzip.compress(b, 6, mtime=0) for b in blocks]
    starts_room, ends_room = 335, 463
    inner = rng.choice([200, 200, 150, -40, 0], len(blocks)).astype(np.int32)
    dev = inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=starts_room, ends_room=ends_room)
    assert dev["summarize_rc"] == 0 and (dev["flags"] == 0).all()
    rec = dev["records"]
    at = np.concatenate([[0], np.cumsum(dev["counts"][:, 2])]).astype(np.int64)
    seen = dict(plain=0, unordered=0, saturated=0, noroom=0, tiny=0)
    for m in range(len(blocks)):
        n = int(at[m + 1] - at[m])
        host, hcounts = inflate.summarize_host(rec, int(at[m]), n, int(inner[m]), 1 << 30, starts_room, ends_room)
        compare(dev["summaries"][m], host, dev["block_counts"][m], hcounts, m)
        assert host["n"] == n
        bad = host["flags"] & (SATURATED | NO_ROOM)
        seen["plain"] += not bad and n > 100
        seen["tiny"] += n <= 3
        seen["unordered"] += not host["flags"] & ENDS_IN_ORDER
        seen["saturated"] += bool(host["flags"] & SATURATED)
        seen["noroom"] += bool(host["flags"] & NO_ROOM)
        if not bad:
            assert int(hcounts[:starts_room].sum()) == n == int(hcounts[starts_room:].sum())
    assert min(seen.values()) > 0, seen
    # ... and nothing else changes: the parsed arrays are those of pd_inflater_run_parsed
    plain = inflate.parse_gpu(hip, blobs)
    for k in ("records", "call_windows", "call_window_records", "profile_windows", "counts", "flags"):
        assert np.array_equal(plain[k], dev[k]), k


def test_summaries_are_refused_without_room(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    blobs = [gzip.compress(b, 6, mtime=0) for b in inflate.profile_blocks(4, seed=2)]
    inner = np.full(4, 200, np.int32)
    assert inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=6000, ends_room=6001)["summarize_rc"] != 0     # more than LDS holds
    assert inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=0, ends_room=400)["summarize_rc"] != 0
    ok = inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=335, ends_room=463)
    assert ok["summarize_rc"] == 0 and sum(s["n"] for s in ok["summaries"]) == len(ok["records"])
