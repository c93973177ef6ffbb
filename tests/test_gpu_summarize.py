"""GPU: runs of parsed records summarised for the loader's tables on the device (pd_inflater_summarize: summarizeRunsKernel,
inflate.hip) against the same computation on the host (pdh::summarizeRun, chrom_profile.cpp) -- the prefix cut at the first pair
that ends at or beyond the border window, how many records begin / end in each 30-bp window, the extremes and the order flags:
every word and every count. What ChromProfile::addBulk() makes of such a summary is held against record-by-record filing on the
CPU (tests/test_host_filing.py); the reference step both stand for is ChromosomeProfile::add with getEndCount after every record
(profile_structure_popdel_call.h:1084-1112, :870-924)."""
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


def test_summaries_of_runs_equal_the_host_computation(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    rng = np.random.default_rng(41)
    # one sample: 60 index regions of everyday records (one pair per ~10 bp, deviations sd 50), then regions with wide inserts, a
    # pile-up on a few positions (counts beyond 255 in a window) and far pairs (end windows beyond any room)
    blocks, begin = [], 3000000
    for m in range(90):
        n = int(rng.poisson(1000))
        pos = begin + np.sort(rng.integers(0, 10000, n))
        dev = np.round(rng.normal(0, 50 if m < 60 else 400, n)).astype(np.int32)
        if 70 <= m < 74:
            pos = begin + np.sort(rng.integers(5000, 5040, n))
        if m >= 84:
            dev[rng.integers(0, n, 3)] = 400000
        blocks.append(block(begin, pos, dev))
        begin += 10000
    blobs = [gzip.compress(b, 6, mtime=0) for b in blocks]
    room = 7179
    parsed = inflate.parse_gpu(hip, blobs)
    rec = parsed["records"]
    total = len(rec)
    assert (parsed["flags"] == 0).all() and total > 80000
    runs = []
    for k in range(400):
        first = int(rng.integers(0, total - 10))
        n = int(rng.choice([1, 2, 17, 300, 5000, 20000, 60000]))
        n = min(n, total - first)
        inner = int(rng.choice([200, 200, 150, -40, 0]))
        last_pos = int(rec[first + n - 1, 0])
        border = int(rng.choice([last_pos // 30 + 40, (last_pos - 700) // 30, int(rec[first, 0]) // 30 + 3, 1 << 30, 0]))
        runs.append((first, n, inner, border))
    runs.append((0, total, 200, 1 << 30))                     # everything: windows beyond the room
    runs.append((0, 0, 200, 1 << 30))                         # nothing
    dev = inflate.parse_gpu(hip, blobs, summarize=runs, room=room)
    assert dev["summarize_rc"] == 0
    seen = dict(cut=0, whole=0, unordered=0, saturated=0, noroom=0, empty=0)
    for k, (first, n, inner, border) in enumerate(runs):
        host, hcounts = inflate.summarize_host(rec, first, n, inner, border, room)
        compare(dev["summaries"][k], host, dev["run_counts"][k], hcounts, (k, first, n, inner, border))
        seen["empty"] += host["n"] == 0
        seen["cut"] += 0 < host["n"] < n
        seen["whole"] += host["n"] == n and n > 0
        seen["unordered"] += host["n"] > 0 and not host["flags"] & ENDS_IN_ORDER
        seen["saturated"] += bool(host["flags"] & SATURATED)
        seen["noroom"] += bool(host["flags"] & NO_ROOM)
        if host["n"] and not host["flags"] & (NO_ROOM | SATURATED):
            assert int(hcounts[0].sum()) == host["n"] == int(hcounts[1].sum())
    assert min(seen.values()) > 0, seen


def test_summarize_refuses_what_it_cannot_hold(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    blobs = [gzip.compress(b, 6, mtime=0) for b in inflate.profile_blocks(4, seed=2)]
    parsed = inflate.parse_gpu(hip, blobs)
    n = len(parsed["records"])
    assert inflate.parse_gpu(hip, blobs, summarize=[(0, n, 200, 1 << 30)], room=8001)["summarize_rc"] != 0      # more room than LDS holds
    assert inflate.parse_gpu(hip, blobs, summarize=[(1, n, 200, 1 << 30)], room=400)["summarize_rc"] != 0       # a run beyond the records
    ok = inflate.parse_gpu(hip, blobs, summarize=[(0, n, 200, 1 << 30)], room=2000)
    assert ok["summarize_rc"] == 0 and ok["summaries"][0]["n"] == n
