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


def test_lean_arrays_are_the_full_ones_without_the_call_windows(built):
    """pd_inflater_run_summarized with lean != 0: per 256-bp window where its records lie and the positions of its first and last
    record instead of the 30-bp windows -- held against the full form of the same blocks (which tests/test_gpu_parse.py holds
    against the loader's reader): same windows, same records; first record = the first 30-bp window's, record count = the sum of
    the 30-bp windows' counts; and the 30-bp windows ARE the runs of equal position / 30 among the records."""
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    rng = np.random.default_rng(6)
    blocks = inflate.profile_blocks(120, seed=21) + inflate.profile_blocks(30, seed=22, pairs_per_member=40)
    for k in range(40):         # windows whose first record sits anywhere on the 30-grid, single-record windows, records on one position
        begin = 7000000 + int(rng.integers(0, 30)) + 2560 * k
        blk = b""
        for w in range(int(rng.integers(1, 6))):
            n = int(rng.choice([1, 1, 2, 7, 40, 255]))
            offs = np.sort(rng.integers(0, 256, n)) if k % 5 else np.full(n, int(rng.integers(0, 256)))
            pos = begin + 256 * w + offs
            blk += block(begin + 256 * w, pos, rng.integers(-2000, 40000, n).astype(np.int32))
        blocks.append(blk)
    blobs = [gzip.compress(b, 6, mtime=0) for b in blocks]
    inner = np.full(len(blocks), 200, np.int32)
    first = np.zeros(len(blocks), np.uint32)
    first[100:] = 100                                          # two samples: indices are relative to the sample's first block
    full = inflate.parse_gpu(hip, blobs, sample_first=first)
    lean = inflate.parse_gpu(hip, blobs, sample_first=first, inner_offsets=inner, starts_room=345, ends_room=473, lean=True)
    assert lean["summarize_rc"] == 0 and (lean["flags"] == 0).all() and (full["flags"] == 0).all()
    assert np.array_equal(lean["records"], full["records"]) and np.array_equal(lean["counts"][:, [0, 2]], full["counts"][:, [0, 2]])
    assert len(lean["call_windows"]) == 0
    pw, lw, cw, cr, rec = full["profile_windows"], lean["profile_windows"], full["call_windows"], full["call_window_records"], full["records"]
    assert len(pw) == len(lw) > 3000
    assert np.array_equal(pw[:, :2], lw[:, :2])
    # a sample's indices restart at its first block: rows of sample 0, then rows of sample 1
    raws_per_block, wins_per_block, recs_per_block = full["counts"][:, 0], full["counts"][:, 1], full["counts"][:, 2]
    for lo, hi in ((0, 100), (100, len(blocks))):
        r0, r1 = int(raws_per_block[:lo].sum()), int(raws_per_block[:hi].sum())
        w0, c0 = int(wins_per_block[:lo].sum()), int(recs_per_block[:lo].sum())
        for k in range(r0, r1):
            fw, nw = int(pw[k, 3]) + w0, int(pw[k, 4])
            assert int(lw[k, 2]) == int(cw[fw, 2]), k                                      # first record (sample-relative)
            assert int(lw[k, 3]) == int(cr[fw: fw + nw].sum()), k                          # records
            a, b = c0 + int(lw[k, 2]), c0 + int(lw[k, 2]) + int(lw[k, 3])
            assert int(lw[k, 4]) == int(rec[a, 0]) and int(lw[k, 5]) == int(rec[b - 1, 0]), k
            # the 30-bp windows: runs of equal position // 30
            bins = rec[a:b, 0] // 30
            starts = np.flatnonzero(np.concatenate([[True], bins[1:] != bins[:-1]]))
            assert len(starts) == nw and np.array_equal(bins[starts] * 30, cw[fw: fw + nw, 1]), k
            assert np.array_equal(np.diff(np.concatenate([starts, [b - a]])), cr[fw: fw + nw]), k


def test_records_come_back_for_the_blocks_at_the_ends_only(built):
    """keep_below / keep_from: the records of a block are copied back only if one of them begins below / at or beyond the block's
    limits; the kept blocks' records are packed in block order, record_index says where each begins, everything else -- lean
    windows, summaries, the count of ALL records -- is unchanged."""
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    blocks = inflate.profile_blocks(60, seed=33)              # regions of 10,000 bp from 1,000,000 on
    blobs = [gzip.compress(b, 6, mtime=0) for b in blocks]
    inner = np.full(60, 200, np.int32)
    first = np.zeros(60, np.uint32)
    first[30:] = 30                                            # two samples of 30 blocks
    below = np.where(np.arange(60) < 30, 1000000 + 25000, 1300000 + 5000).astype(np.uint32)
    beyond = np.where(np.arange(60) < 30, 1000000 + 281000, 1300000 + 300000).astype(np.uint32)
    whole = inflate.parse_gpu(hip, blobs, sample_first=first, inner_offsets=inner, starts_room=345, ends_room=473, lean=True)
    part = inflate.parse_gpu(hip, blobs, sample_first=first, inner_offsets=inner, starts_room=345, ends_room=473, lean=True, keep=(below, beyond))
    assert part["summarize_rc"] == 0
    for k in ("profile_windows", "counts", "flags"):
        assert np.array_equal(part[k], whole[k]), k
    assert part["summaries"] == whole["summaries"] and np.array_equal(part["block_counts"], whole["block_counts"])
    at = np.concatenate([[0], np.cumsum(whole["counts"][:, 2])]).astype(np.int64)
    idx = part["record_index"]
    kept = [m for m in range(60) if idx[m] != np.uint64(2**64 - 1)]
    spans = [(int(whole["records"][at[m], 0]), int(whole["records"][at[m + 1] - 1, 0])) for m in range(60)]
    expect = [m for m in range(60) if spans[m][0] < below[m] or spans[m][1] >= beyond[m]]
    assert kept == expect and 3 <= len(kept) <= 8, (kept, expect)
    packed = 0
    for m in kept:
        n = int(at[m + 1] - at[m])
        assert int(idx[m]) == packed
        assert np.array_equal(part["records"][packed: packed + n], whole["records"][at[m]: at[m + 1]]), m
        packed += n
    assert packed == len(part["records"]) < len(whole["records"]) // 4


def test_summaries_are_refused_without_room(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    blobs = [gzip.compress(b, 6, mtime=0) for b in inflate.profile_blocks(4, seed=2)]
    inner = np.full(4, 200, np.int32)
    assert inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=6000, ends_room=6001)["summarize_rc"] != 0     # more than LDS holds
    assert inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=0, ends_room=400)["summarize_rc"] != 0
    ok = inflate.parse_gpu(hip, blobs, inner_offsets=inner, starts_room=335, ends_room=463)
    assert ok["summarize_rc"] == 0 and sum(s["n"] for s in ok["summaries"]) == len(ok["records"])
