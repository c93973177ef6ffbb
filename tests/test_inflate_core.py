"""CPU: the device's gunzip (popdel_amd/csrc/hip/inflate_core.h -- the source inflate.hip compiles for the GPU) run on the
host against zlib: bit-identical output on profile-shaped blocks at every compression level, on stored and fixed-code
blocks, long matches and incompressible data; and the same accept / refuse decision as zlib on truncated and damaged
members. The library exports pd_inflate_members (checked here without a GPU: the symbol only)."""
import ctypes
import gzip
import zlib

import numpy as np

from conftest import HIP_LIB


def zlib_inflates(blob):
    try:
        d = zlib.decompressobj(31)
        out = d.decompress(blob)
        return out if d.eof else None
    except zlib.error:
        return None


def run_core(blobs, room=None):
    from popdel_amd import inflate
    data, offs, out_offs = inflate.pack_members(blobs, room)
    out, status, lengths = inflate.inflate_core_host(data, offs, out_offs)
    return [bytes(out[int(out_offs[i]):int(out_offs[i]) + int(lengths[i])]) for i in range(len(blobs))], status


def test_profile_blocks_at_every_level(built):
    from popdel_amd import inflate
    blocks = inflate.profile_blocks(40, seed=3)
    for level in (1, 6, 9):
        blobs = [gzip.compress(b, level, mtime=0) for b in blocks]
        got, status = run_core(blobs)
        assert (status == 0).all(), status
        assert got == blocks


def test_block_types_and_corner_cases(built):
    rng = np.random.default_rng(5)
    cases = [b"", b"a", b"abc" * 5, bytes(70000), bytes(rng.integers(0, 256, 40000, dtype=np.uint8)),
             b"".join(bytes([i % 251]) * (i % 300) for i in range(400)), bytes(rng.integers(0, 4, 100000, dtype=np.uint8))]
    blobs = []
    for c in cases:
        blobs.append(gzip.compress(c, 0, mtime=0))                          # stored blocks (64-KB pieces)
        blobs.append(gzip.compress(c, 6, mtime=0))
        co = zlib.compressobj(9, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)        # fixed Huffman codes
        blobs.append(co.compress(c) + co.flush())
        co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_HUFFMAN_ONLY)
        blobs.append(co.compress(c) + co.flush())
        co = zlib.compressobj(6, zlib.DEFLATED, 31)                         # several blocks: a sync flush in the middle
        blobs.append(co.compress(c[: len(c) // 2]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(c[len(c) // 2:]) + co.flush())
    want = [zlib_inflates(b) for b in blobs]
    assert all(w is not None for w in want)
    got, status = run_core(blobs)
    assert (status == 0).all(), status
    assert got == want
    # a header with optional fields (FEXTRA, FNAME, FCOMMENT, FHCRC) in front of the same stream
    plain = blobs[1]
    hdr = bytes([0x1f, 0x8b, 8, 2 | 4 | 8 | 16]) + plain[4:10] + b"\x03\x00abc" + b"name\0" + b"comment\0"
    crc16 = (zlib.crc32(hdr) & 0xffff).to_bytes(2, "little")
    fancy = hdr + crc16 + plain[10:]
    assert zlib_inflates(fancy) == cases[0]
    got, status = run_core([fancy])
    assert status[0] == 0 and got[0] == cases[0]


def test_truncated_and_damaged_members_are_refused_like_zlib(built):
    from popdel_amd import inflate
    blocks = inflate.profile_blocks(6, seed=9)
    rng = np.random.default_rng(11)
    blobs, expect = [], []
    for b in blocks:
        z = gzip.compress(b, 6, mtime=0)
        for cut in sorted(set([0, 1, 5, 10, 11, 17, len(z) // 3, len(z) // 2, len(z) - 9, len(z) - 8, len(z) - 1])):
            blobs.append(z[:cut])
            expect.append(None)
        for _ in range(60):
            k = int(rng.integers(0, len(z)))
            dmg = bytearray(z)
            dmg[k] ^= 1 << int(rng.integers(0, 8))
            blobs.append(bytes(dmg))
            expect.append(zlib_inflates(bytes(dmg)))
    # room: what the undamaged member needs (a damaged ISIZE word must not size the buffer)
    got, status = run_core(blobs, room=[70000] * len(blobs))
    for i, (g, e) in enumerate(zip(got, expect)):
        if e is None:
            assert status[i] != 0, (i, "zlib refuses this member")
        else:
            # zlib ignores the header's MTIME / XFL / OS bytes, and so does the core: the member is accepted with the same output
            assert status[i] == 0 and g == e, i
    assert sum(e is None for e in expect) > 300
    # too little room: refused, nothing written beyond it
    z = gzip.compress(blocks[0], 6, mtime=0)
    got, status = run_core([z], room=[len(blocks[0]) - 1])
    assert status[0] == 6 and len(got[0]) <= len(blocks[0]) - 1


def test_library_exports_the_entry_point(built):
    lib = ctypes.CDLL(HIP_LIB)
    assert hasattr(lib, "pd_inflate_members") and hasattr(lib, "pd_inflate_last_error")


def test_symbol_table_against_the_rfc(built):
    """pdinf::symbolKind() -- what a literal / length or distance symbol stands for, from one formula each -- against the tables of
    RFC 1951, 3.2.5, written out (length 3..258 with 0..5 extra bits, distance 1..24,577 with 0..13; 286, 287, 30 and 31: no symbols)."""
    import os
    from conftest import ROOT
    host = ctypes.CDLL(os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so"))
    assert host.pdh_selftest_symbol_kinds() == 0
