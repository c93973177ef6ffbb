"""ctypes access to the gunzip of profile blocks: pd_inflate_members (device, libpopdel_hip.so), and from libpopdel_host.so
the same decoder source run on the CPU (pdh_inflate_members_core: what the "not gpu" tests check against zlib) and the
loader's own libdeflate / zlib path on host threads (pdh_inflate_members_cpu: the CPU figure beside the kernel)."""
from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class PdInflateTiming(C.Structure):
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double), ("in_bytes", C.c_uint64),
                ("out_bytes", C.c_uint64), ("workgroups", C.c_uint32)]


def pack_members(blobs, room=None):
    """Concatenates gzip members; output room per member from its ISIZE trailer (or `room`, a list). Returns
    (data uint8[], offsets uint64[n+1], out_offsets uint64[n+1])."""
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8).copy() if blobs else np.zeros(0, np.uint8)
    offsets = np.zeros(len(blobs) + 1, np.uint64)
    np.cumsum([len(b) for b in blobs], out=offsets[1:])
    sizes = room if room is not None else [struct.unpack("<I", b[-4:])[0] if len(b) >= 4 else 0 for b in blobs]
    out_offsets = np.zeros(len(blobs) + 1, np.uint64)
    np.cumsum(sizes, out=out_offsets[1:])
    return data, offsets, out_offsets


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _host():
    return C.CDLL(os.path.join(_HERE, "lib", "libpopdel_host.so"))


def inflate_core_host(data, offsets, out_offsets):
    """inflate_core.h on the CPU. Returns (out uint8[], status uint32[n], lengths uint64[n])."""
    n = len(offsets) - 1
    out = np.zeros(max(1, int(out_offsets[-1])), np.uint8)
    status = np.zeros(n, np.uint32)
    lengths = np.zeros(n, np.uint64)
    lib = _host()
    lib.pdh_inflate_members_core.restype = C.c_int
    rc = lib.pdh_inflate_members_core(_ptr(data, C.c_uint8), _ptr(offsets, C.c_uint64), C.c_uint32(n), _ptr(out, C.c_uint8),
                                      _ptr(out_offsets, C.c_uint64), _ptr(status, C.c_uint32), _ptr(lengths, C.c_uint64))
    if rc != 0:
        raise RuntimeError("pdh_inflate_members_core failed")
    return out, status, lengths


def inflate_cpu(data, offsets, out_offsets, threads, zlib_only=False):
    """libdeflate (or zlib) on `threads` host threads. Returns (out, seconds, 'libdeflate' | 'zlib', failures)."""
    n = len(offsets) - 1
    out = np.zeros(max(1, int(out_offsets[-1])), np.uint8)
    seconds, used, failures = C.c_double(), C.c_int(), C.c_uint32()
    lib = _host()
    lib.pdh_inflate_members_cpu.restype = C.c_int
    rc = lib.pdh_inflate_members_cpu(_ptr(data, C.c_uint8), _ptr(offsets, C.c_uint64), C.c_uint32(n), _ptr(out, C.c_uint8),
                                     _ptr(out_offsets, C.c_uint64), C.c_int(1 if zlib_only else 0), C.c_uint(threads),
                                     C.byref(seconds), C.byref(used), C.byref(failures))
    if rc != 0:
        raise RuntimeError("pdh_inflate_members_cpu failed")
    return out, seconds.value, ("zlib" if used.value else "libdeflate"), failures.value


def inflate_gpu(hip_lib, data, offsets, out_offsets, device=0, repeat=1):
    """pd_inflate_members. Returns (out, status, lengths, timing dict)."""
    n = len(offsets) - 1
    out = np.zeros(max(1, int(out_offsets[-1])), np.uint8)
    status = np.zeros(n, np.uint32)
    lengths = np.zeros(n, np.uint64)
    t = PdInflateTiming()
    hip_lib.pd_inflate_members.restype = C.c_int
    hip_lib.pd_inflate_last_error.restype = C.c_char_p
    rc = hip_lib.pd_inflate_members(C.c_int(device), _ptr(data, C.c_uint8), _ptr(offsets, C.c_uint64), C.c_uint32(n),
                                    _ptr(out, C.c_uint8), _ptr(out_offsets, C.c_uint64), _ptr(status, C.c_uint32),
                                    _ptr(lengths, C.c_uint64), C.c_uint32(repeat), C.byref(t))
    if rc != 0:
        raise RuntimeError(hip_lib.pd_inflate_last_error().decode())
    return out, status, lengths, {k: getattr(t, k) for k, _ in PdInflateTiming._fields_}


def profile_blocks(n_members, seed=1, pairs_per_member=1000):
    """Uncompressed index-region blocks shaped like a 30x profile's (per 256-bp window chrom, beginPos and
    per read group a count and {u8 offset, i32 deviation} records; ~1,000 pairs per 10,000 bp)."""
    rng = np.random.default_rng(seed)
    blocks = []
    for m in range(n_members):
        pos = np.sort(rng.integers(0, 10000, size=int(rng.poisson(pairs_per_member))))
        dev = np.round(rng.normal(0, 50, size=len(pos))).astype(np.int32)
        blk = bytearray()
        win = pos // 256
        for w in np.unique(win):
            sel = win == w
            blk += struct.pack("<III", 19, 1000000 + m * 10000 + int(w) * 256, int(sel.sum()))
            rec = np.zeros(int(sel.sum()), dtype=np.dtype([("off", "u1"), ("dev", "<i4")]))
            rec["off"] = (pos[sel] - w * 256).astype(np.uint8)
            rec["dev"] = dev[sel]
            blk += rec.tobytes()
        blocks.append(bytes(blk))
    return blocks


class PdParsedBlocks(C.Structure):
    _fields_ = [("profile_windows", C.c_void_p), ("call_windows", C.c_void_p), ("call_window_records", C.POINTER(C.c_uint32)),
                ("records", C.c_void_p), ("profile_window_offsets", C.POINTER(C.c_uint64)),
                ("call_window_offsets", C.POINTER(C.c_uint64)), ("record_offsets", C.POINTER(C.c_uint64)),
                ("block_flags", C.POINTER(C.c_uint32)), ("bytes_copied", C.c_uint32)]


SUMMARY_FIELDS = ("n", "first_pos", "last_pos", "bin_base", "n_bins", "min_end_bin", "max_end_bin", "first_end_bin", "last_end_bin", "flags")


class PdBlockSummaries(C.Structure):
    _fields_ = [("block_summaries", C.POINTER(C.c_uint32)), ("block_counts", C.POINTER(C.c_uint8)), ("starts_room", C.c_uint32),
                ("ends_room", C.c_uint32), ("record_index", C.POINTER(C.c_uint64))]


class PdSummarizeOptions(C.Structure):
    _fields_ = [("inner_offset", C.POINTER(C.c_int32)), ("starts_room", C.c_uint32), ("ends_room", C.c_uint32), ("lean", C.c_uint32),
                ("keep_below", C.POINTER(C.c_uint32)), ("keep_from", C.POINTER(C.c_uint32))]


def summarize_host(records, first, n, inner_offset, border_window, starts_room, ends_room):
    """pdh::summarizeRun (chrom_profile.cpp) on records[first : first + n] ({pos, dev} rows). Returns (summary dict, counts
    uint8[starts_room + ends_room]: records that begin / end per window)."""
    rec = np.ascontiguousarray(records[first: first + n], dtype=np.uint32)
    if len(rec) == 0:
        rec = np.zeros((1, 2), np.uint32)
    summary = np.zeros(10, np.uint32)
    counts = np.zeros(starts_room + ends_room, np.uint8)
    lib = _host()
    lib.pdh_summarize_run.restype = C.c_int
    if lib.pdh_summarize_run(_ptr(rec, C.c_uint32), C.c_uint64(n), C.c_int(inner_offset), C.c_uint32(border_window), C.c_uint32(starts_room),
                             C.c_uint32(ends_room), _ptr(summary, C.c_uint32), _ptr(counts, C.c_uint8)) != 0:
        raise RuntimeError("pdh_summarize_run failed")
    return dict(zip(SUMMARY_FIELDS, (int(v) for v in summary))), counts


def parse_gpu(hip_lib, blobs, sample_first=None, device=0, inner_offsets=None, starts_room=0, ends_room=0, lean=False, keep=None):
    """pd_inflater_run_parsed on gzip members (one sample unless `sample_first` says otherwise). Returns a dict: status[n],
    flags[n], counts[n, 3] (256-bp windows, 30-bp windows, records per block), profile_windows[.., 5], call_windows[.., 3],
    call_window_records[..], records[.., 2] (all blocks' rows in block order), bytes_copied. `inner_offsets` (one per member):
    pd_inflater_run_summarized instead, with `starts_room` / `ends_room` windows per block -> 'summaries' (a dict per block) and
    'block_counts' uint8[n, starts_room + ends_room] as well ('summarize_rc' != 0 if the call was refused: nothing else then).
    `lean`: the lean form -- 'profile_windows' holds rows {chrom, beginPos, first record, records, first position, last position},
    'call_windows' / 'call_window_records' are empty. `keep` = (keep_below[n], keep_from[n]): records come back for the blocks with a
    record below / at or beyond only -> 'records' holds those blocks' records packed, 'record_index'[n] where each block's begin
    (2**64 - 1: not copied back)."""
    data, offsets, out_offsets = pack_members(blobs)
    n = len(blobs)
    first = np.zeros(n, np.uint32) if sample_first is None else np.asarray(sample_first, np.uint32)
    want = np.ones(n, np.uint8)
    hip_lib.pd_inflater_create.restype = C.c_void_p
    hip_lib.pd_inflater_input.restype = C.POINTER(C.c_uint8)
    hip_lib.pd_inflater_input.argtypes = [C.c_void_p, C.c_uint64]
    hip_lib.pd_inflater_destroy.argtypes = [C.c_void_p]
    hip_lib.pd_inflater_run_parsed.restype = C.c_int
    inf = hip_lib.pd_inflater_create(C.c_int(device))
    if not inf:
        raise RuntimeError("pd_inflater_create failed")
    try:
        buf = hip_lib.pd_inflater_input(inf, C.c_uint64(len(data)))
        C.memmove(buf, data.ctypes.data, len(data))
        out, status, lengths, pb, ms = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)(), PdParsedBlocks(), C.c_double()
        sums = PdBlockSummaries()
        if inner_offsets is not None:
            inner = np.ascontiguousarray(inner_offsets, dtype=np.int32)
            hip_lib.pd_inflater_run_summarized.restype = C.c_int
            opt = PdSummarizeOptions(_ptr(inner, C.c_int32), starts_room, ends_room, 1 if lean else 0, None, None)
            if keep is not None:
                below, beyond = np.ascontiguousarray(keep[0], np.uint32), np.ascontiguousarray(keep[1], np.uint32)
                opt.keep_below, opt.keep_from = _ptr(below, C.c_uint32), _ptr(beyond, C.c_uint32)
            rc = hip_lib.pd_inflater_run_summarized(C.c_void_p(inf), _ptr(offsets, C.c_uint64), C.c_uint32(n), _ptr(out_offsets, C.c_uint64),
                                                    _ptr(first, C.c_uint32), _ptr(want, C.c_uint8), C.byref(opt), C.byref(out), C.byref(status),
                                                    C.byref(lengths), C.byref(pb), C.byref(sums), C.byref(ms))
            if rc != 0:
                return dict(summarize_rc=rc)
        else:
            rc = hip_lib.pd_inflater_run_parsed(C.c_void_p(inf), _ptr(offsets, C.c_uint64), C.c_uint32(n), _ptr(out_offsets, C.c_uint64),
                                                _ptr(first, C.c_uint32), _ptr(want, C.c_uint8), C.byref(out), C.byref(status), C.byref(lengths),
                                                C.byref(pb), C.byref(ms))
            if rc != 0:
                raise RuntimeError("pd_inflater_run_parsed failed (%d)" % rc)
        st = np.ctypeslib.as_array(status, (n,)).copy()
        fl = np.ctypeslib.as_array(pb.block_flags, (n, 4)).copy()
        ro = np.ctypeslib.as_array(pb.profile_window_offsets, (n + 1,)).copy()
        wo = np.ctypeslib.as_array(pb.call_window_offsets, (n + 1,)).copy()
        co = np.ctypeslib.as_array(pb.record_offsets, (n + 1,)).copy()

        def rows(ptr, count, width):
            if count == 0:
                return np.zeros((0, width), np.uint32)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), (count * width,)).reshape(count, width).copy()

        n_cw = 0 if lean else int(wo[-1])
        n_back = int(co[-1])
        record_index = None
        if inner_offsets is not None and keep is not None:
            record_index = np.ctypeslib.as_array(sums.record_index, (n,)).copy()
            kept = record_index != np.uint64(2**64 - 1)
            n_back = int(np.diff(co)[kept].sum())
        res = dict(status=st, flags=fl[:, 3], counts=np.stack([np.diff(ro), np.diff(wo), np.diff(co)], axis=1),
                   profile_windows=rows(pb.profile_windows, int(ro[-1]), 6 if lean else 5), call_windows=rows(pb.call_windows, n_cw, 3),
                   call_window_records=rows(pb.call_window_records, n_cw, 1)[:, 0], records=rows(pb.records, n_back, 2),
                   bytes_copied=int(pb.bytes_copied))
        if record_index is not None:
            res["record_index"] = record_index
        if inner_offsets is not None:
            res["summarize_rc"] = 0
            words = np.ctypeslib.as_array(sums.block_summaries, (n * 10,)).reshape(n, 10).copy()
            res["summaries"] = [dict(zip(SUMMARY_FIELDS, (int(v) for v in row))) for row in words]
            res["block_counts"] = np.ctypeslib.as_array(sums.block_counts, (n * (starts_room + ends_room),)).reshape(n, starts_room + ends_room).copy()
        return res
    finally:
        hip_lib.pd_inflater_destroy(C.c_void_p(inf))


def parse_host(blocks):
    """The loader's own reader on inflated blocks (pdh_parse_blocks). Returns a dict like parse_gpu's (`state` instead of flags:
    0 read to the end, 1 the reader threw, 2 a window without records); rows of the blocks with state 0 only, indices relative to the block."""
    data = np.frombuffer(b"".join(blocks), dtype=np.uint8).copy() if blocks else np.zeros(1, np.uint8)
    offsets = np.zeros(len(blocks) + 1, np.uint64)
    np.cumsum([len(b) for b in blocks], out=offsets[1:])
    n = len(blocks)
    counts = np.zeros((n, 3), np.uint64)
    state = np.zeros(n, np.uint32)
    lib = _host()
    lib.pdh_parse_blocks.restype = C.c_int
    null = C.POINTER(C.c_uint32)()
    if lib.pdh_parse_blocks(_ptr(data, C.c_uint8), _ptr(offsets, C.c_uint64), C.c_uint32(n), _ptr(counts, C.c_uint64), _ptr(state, C.c_uint32),
                            null, null, null, null) != 0:
        raise RuntimeError("pdh_parse_blocks failed")
    ok = state == 0
    tot = counts[ok].sum(axis=0).astype(np.int64) if ok.any() else np.zeros(3, np.int64)
    pw = np.zeros((max(1, int(tot[0])), 5), np.uint32)
    cw = np.zeros((max(1, int(tot[1])), 3), np.uint32)
    cr = np.zeros(max(1, int(tot[1])), np.uint32)
    rc = np.zeros((max(1, int(tot[2])), 2), np.uint32)
    if lib.pdh_parse_blocks(_ptr(data, C.c_uint8), _ptr(offsets, C.c_uint64), C.c_uint32(n), _ptr(counts, C.c_uint64), _ptr(state, C.c_uint32),
                            _ptr(pw, C.c_uint32), _ptr(cw, C.c_uint32), _ptr(cr, C.c_uint32), _ptr(rc, C.c_uint32)) != 0:
        raise RuntimeError("pdh_parse_blocks failed")
    return dict(state=state, counts=counts.astype(np.int64), profile_windows=pw[: int(tot[0])], call_windows=cw[: int(tot[1])],
                call_window_records=cr[: int(tot[1])], records=rc[: int(tot[2])])
