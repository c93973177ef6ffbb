"""GPU: pd_inflate_members (one lane per gzip member) against zlib -- profile-shaped blocks at three compression levels,
stored / fixed-code / multi-block members, and damaged or truncated members, which every lane must refuse on its own
without disturbing its neighbours in the wave."""
import gzip
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def zlib_inflates(blob):
    try:
        d = zlib.decompressobj(31)
        out = d.decompress(blob)
        return out if d.eof else None
    except zlib.error:
        return None


def test_members_against_zlib(built):
    from popdel_amd import capi, inflate
    hip = capi.load_hip()
    blocks = inflate.profile_blocks(2100, seed=21)
    rng = np.random.default_rng(2)
    blobs, want = [], []
    for i, b in enumerate(blocks):
        z = gzip.compress(b, (1, 6, 9)[i % 3], mtime=0)
        if i % 7 == 3:                          # damaged: one bit flipped somewhere
            dmg = bytearray(z)
            dmg[int(rng.integers(0, len(z)))] ^= 1 << int(rng.integers(0, 8))
            z = bytes(dmg)
        elif i % 11 == 5:                       # truncated (every fourth of them within the last bytes of the deflate stream)
            z = z[: int(rng.integers(0, len(z)))] if i % 4 else z[: len(z) - 8 - int(rng.integers(0, 6))]
        blobs.append(z)
        want.append(zlib_inflates(z))
    for c in (b"", b"x", bytes(70000), bytes(rng.integers(0, 256, 30000, dtype=np.uint8))):
        for z in (gzip.compress(c, 0, mtime=0), gzip.compress(c, 9, mtime=0)):
            blobs.append(z)
            want.append(c)
        co = zlib.compressobj(9, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
        blobs.append(co.compress(c) + co.flush())
        want.append(c)
        co = zlib.compressobj(6, zlib.DEFLATED, 31)                         # several blocks: a full flush in the middle
        blobs.append(co.compress(c[: len(c) // 2]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(c[len(c) // 2:]) + co.flush())
        want.append(c)
    # matches at every short distance, longer than their distance (runs), and a member that ends right at its room
    for d in (1, 2, 3, 4, 5, 7, 8, 9, 13):
        c = bytes(rng.integers(0, 256, d, dtype=np.uint8)) * (3000 // d) + b"tail" + bytes(range(d))
        blobs.append(gzip.compress(c, 9, mtime=0))
        want.append(c)
    room = [max(70000, len(w) if w else 0) for w in want]
    room[-9:] = [len(w) for w in want[-9:]]                # (no slack behind these: the last bytes go out one by one)
    data, offs, out_offs = inflate.pack_members(blobs, room=room)
    out, status, lengths, timing = inflate.inflate_gpu(hip, data, offs, out_offs)
    n_bad = 0
    for i, w in enumerate(want):
        if w is None:
            assert status[i] != 0, i
            n_bad += 1
        else:
            assert status[i] == 0, (i, int(status[i]))
            assert bytes(out[int(out_offs[i]):int(out_offs[i]) + int(lengths[i])]) == w, i
    assert n_bad > 150 and timing["kernel_ms"] > 0
    # and the same bytes as the decoder's source gives on the CPU
    out_c, status_c, lengths_c = inflate.inflate_core_host(data, offs, out_offs)
    assert np.array_equal(status, status_c) and np.array_equal(lengths, lengths_c)
    ok = np.flatnonzero(status == 0)
    for i in ok[:200]:
        a, n = int(out_offs[i]), int(lengths[i])
        assert np.array_equal(out[a:a + n], out_c[a:a + n])
