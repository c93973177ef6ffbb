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
