"""GPU: the device's parser of inflated profile blocks (pd_inflater_run_parsed: parseCount / parseScan / parseWriteKernel,
inflate.hip) against the loader's own reader (readWindowHead + readWindowBodyFlat, profile_io.cpp -- the reference's
readWindow, window_podel.h:210-251, and convertWindow, :314-418): the same 256-bp windows, 30-bp windows, record counts and
records for every block the device parses, and a flag -- the sample then goes to the host's reader -- on every block that
reader would stumble over (cut short, a window without records, records out of order)."""
import gzip
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def window(chrom, begin, offs, devs):
    rec = np.zeros(len(offs), dtype=np.dtype([("off", "u1"), ("dev", "<i4")]))
    rec["off"] = offs
    rec["dev"] = devs
    return struct.pack("<III", chrom, begin, len(offs)) + rec.tobytes()


def test_parsed_blocks_are_what_the_host_reader_makes_of_them(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    rng = np.random.default_rng(5)
    blocks = inflate.profile_blocks(300, seed=9) + inflate.profile_blocks(40, seed=10, pairs_per_member=30)
    # windows whose first record sits anywhere on the 30-grid, single-record windows, records on one position, the last window of a contig
    for k in range(60):
        blk = b""
        begin = 5000000 + int(rng.integers(0, 30)) + 256 * k
        for w in range(int(rng.integers(1, 6))):
            n = int(rng.choice([1, 1, 2, 7, 40, 255]))
            offs = np.sort(rng.integers(0, 256, n)).astype(np.uint8) if k % 5 else np.full(n, int(rng.integers(0, 256)), np.uint8)
            blk += window(int(rng.integers(0, 25)), begin + 256 * w, offs, rng.integers(-2000, 40000, n).astype(np.int32))
        blocks.append(blk)
    regular = len(blocks)
    # what the reader stumbles over: a block cut short (inside a header, inside the records), a window without records, records out of order
    whole = inflate.profile_blocks(3, seed=11)
    blocks.append(whole[0][:-3])
    blocks.append(whole[1][: 12 * 1 + 7])
    blocks.append(whole[2] + struct.pack("<II", 19, 77))
    blocks.append(window(19, 1000, [], []) + whole[0])
    blocks.append(whole[0] + window(19, 9000000, [], []))
    blocks.append(window(19, 2000, [5, 200, 100, 250], [1, 2, 3, 4]))
    blocks.append(window(19, 2000, [5, 100, 200], [1, 2, 3]) + window(19, 2256, [9, 8], [1, 2]) + whole[1])
    blocks.append(b"")
    blobs = [gzip.compress(b, 6, mtime=0) for b in blocks]
    dev = inflate.parse_gpu(hip, blobs)
    ref = inflate.parse_host(blocks)
    assert (dev["status"] == 0).all()
    n = len(blocks)
    # every block the host reads to the end in order is parsed; every other one is flagged
    assert (dev["flags"][:regular] == 0).all(), np.flatnonzero(dev["flags"][:regular])
    assert (ref["state"][:regular] == 0).all()
    for b in range(regular, n - 1):
        assert dev["flags"][b] != 0, b
    assert dev["flags"][n - 1] == 0 and ref["state"][n - 1] == 0 and (dev["counts"][n - 1] == 0).all()      # (an empty block: nothing to read, nothing wrong)
    assert dev["bytes_copied"] == 1
    # the rows of the parsed blocks, block by block (the device's indices are relative to the sample's first block: here block 0)
    d_at, r_at = np.zeros(3, np.int64), np.zeros(3, np.int64)
    compared = 0
    for b in range(n):
        dc = dev["counts"][b].astype(np.int64)
        rc = ref["counts"][b] if ref["state"][b] == 0 else np.zeros(3, np.int64)
        if dev["flags"][b] == 0:
            assert ref["state"][b] == 0 and (dc == rc).all(), (b, dc, rc)
            pw_d = dev["profile_windows"][d_at[0]: d_at[0] + dc[0]].astype(np.int64)
            pw_r = ref["profile_windows"][r_at[0]: r_at[0] + rc[0]].astype(np.int64)
            assert (pw_d[:, [0, 1, 2, 4]] == pw_r[:, [0, 1, 2, 4]]).all(), b
            assert (pw_d[:, 3] - d_at[1] == pw_r[:, 3]).all(), b                      # first 30-bp window: sample-relative vs block-relative
            cw_d = dev["call_windows"][d_at[1]: d_at[1] + dc[1]].astype(np.int64)
            cw_r = ref["call_windows"][r_at[1]: r_at[1] + rc[1]].astype(np.int64)
            assert (cw_d[:, :2] == cw_r[:, :2]).all() and (cw_d[:, 2] - d_at[2] == cw_r[:, 2]).all(), b
            assert (dev["call_window_records"][d_at[1]: d_at[1] + dc[1]] == ref["call_window_records"][r_at[1]: r_at[1] + rc[1]]).all(), b
            assert (dev["records"][d_at[2]: d_at[2] + dc[2]] == ref["records"][r_at[2]: r_at[2] + rc[2]]).all(), b
            compared += int(dc[2])
        else:
            assert (dc == 0).all(), b
        d_at += dc
        r_at += rc
    assert compared > 250000


def test_everything_parsed_means_no_bytes_come_back(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    blocks = inflate.profile_blocks(64, seed=3)
    dev = inflate.parse_gpu(hip, [gzip.compress(b, 6, mtime=0) for b in blocks])
    assert (dev["flags"] == 0).all() and dev["bytes_copied"] == 0
    ref = inflate.parse_host(blocks)
    assert (dev["counts"] == ref["counts"]).all() and (dev["records"] == ref["records"]).all()
