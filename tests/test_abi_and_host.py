"""CPU: the C-ABI libraries export what include/popdel_hip.h declares; host-side units."""
import ctypes
import os
import re
import subprocess

import numpy as np

from conftest import HIP_LIB, ORACLE_LIB, POPDEL_BIN, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "popdel_hip.h")).read()
    return sorted(set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_by_both_libraries(built):
    names = declared_symbols()
    assert "pd_segment_submit" in names and "pd_ctx_create" in names
    for path in (HIP_LIB, ORACLE_LIB):
        lib = ctypes.CDLL(path)          # loading needs no GPU
        for n in names:
            assert hasattr(lib, n), (path, n)
    from popdel_amd import capi
    assert sorted(capi.ABI_SYMBOLS) == names
    hip = capi.load_hip()
    assert hip.pd_backend_name() == b"hip-gfx950" and hip.pd_abi_version() == 14


def test_product_binary_links_only_the_hip_library(built):
    out = subprocess.check_output(["ldd", POPDEL_BIN]).decode()
    assert "libpopdel_hip.so" in out and "oracle" not in out
    needed = subprocess.check_output(["readelf", "-d", HIP_LIB]).decode()
    assert "oracle" not in needed


def test_product_sources_do_not_reference_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "popdel_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(base, f)).read()
                assert "genotype_oracle" not in text and "libpopdel_oracle" not in text and "oracle/_build" not in text, f


def test_histogram_conditioning_matches_start_up_log(built):
    """min_prob / quantile distances of the survey cohort's library; 4 sigma threshold = 200."""
    from popdel_amd import synth
    rg = synth.conditioned_readgroup(synth.Library())
    assert rg.min_init_del_len == 200
    assert len(rg.hist_values) == 301 and rg.hist_offset == 350
    assert rg.lower_quantile_dist == 114 and rg.upper_quantile_dist == 112
    assert abs(rg.min_prob - max(rg.hist_values) / 500) < 1e-18
    assert rg.hist_values.min() >= rg.min_prob


def test_interval_rule_matches_state_machine_inside_a_segment(built):
    """instances_by_interval_rule() is what bench/parity inputs use; away from segment borders it must agree with the
    host driver's event log (checked through the oracle: same calls from both descriptions)."""
    from popdel_amd import synth
    lib = synth.Library()
    pos, dev = synth.sample_pairs(1000, (1000000, 1050000), lib)
    d, s, on, off = synth.instances_by_interval_rule(pos, dev, lib, 1000020, 1000)
    assert (off > on).all() and (np.diff(on.astype(np.int64)) >= 0).all()
    # a pair starting exactly on a grid position is active from that window on
    k = int(np.flatnonzero((s - 1000020) % 30 == 0)[0])
    assert on[k] == (int(s[k]) - 1000020) // 30
    life = (off - on).astype(int)
    assert 6 <= life.mean() <= 10


def test_bulk_decoder_equals_the_window_reader(built):
    """readWindowHead() + readWindowBodyFlat() (what `popdel call` decodes with) against readWindow() + splitWindowFlat()
    on random window records: one to three read groups, empty ones, records out of position order."""
    host = ctypes.CDLL(os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so"))
    for seed in (1, 2, 3):
        assert host.pdh_selftest_window_decode(ctypes.c_uint(seed), ctypes.c_uint(20000)) == 0


def test_inflate_through_libdeflate_equals_zlib(built, tmp_path):
    """What `popdel call` reads compressed profiles with (libdeflate member by member where the system has it, zlib for
    whatever libdeflate does not take) delivers the bytes zlib delivers -- on whole, truncated and damaged streams, stored
    blocks, empty members, a member beyond the decoder's 64-MB limit, and bytes that are no gzip member."""
    import gzip
    import random
    host = ctypes.CDLL(os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so"))
    host.pdh_selftest_inflate.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_uint, ctypes.POINTER(ctypes.c_ulonglong)]
    rng = random.Random(1)
    noise = bytes(rng.getrandbits(8) for _ in range(300000))
    member = lambda data, level=6: gzip.compress(data, level, mtime=0)
    normal = b"".join(member(noise[i:i + 5000]) for i in range(0, 100000, 5000))
    damaged = bytearray(normal)
    damaged[len(damaged) // 2] ^= 0x55
    streams = {
        "normal": (normal, 100000),
        "large members": (member(noise) + member(b"abcdefgh" * 40000), 620000),
        "stored block": (member(noise[:70000], 0) + member(b"tail"), 70004),
        "truncated": (normal[:-37], None),
        "damaged": (bytes(damaged), None),
        "garbage behind": (member(b"x" * 1000) + b"not a gzip member at all", 1000),
        "empty members": (member(b"") + member(b"abc") + member(b"") + member(b"def"), 6),
        "beyond the limit": (member(b"\0" * (80 << 20), 1) + member(b"after"), (80 << 20) + 5),
    }
    n = ctypes.c_ulonglong()
    for name, (blob, want) in streams.items():
        path = tmp_path / "stream.bin"
        path.write_bytes(b"HEADER" + blob)
        for request in ((125, 1, 70000) if want is None or want < (1 << 20) else (4096,)):
            assert host.pdh_selftest_inflate(str(path).encode(), 6, request, ctypes.byref(n)) == 0, (name, request)
            assert want is None or n.value == want, (name, request, n.value)


def test_loader_chunks_and_the_replays_batch_limit(built):
    """The device loader cuts a round's READERS into chunks (three equal ones up to 52,224 readers, 17,408 each beyond) and
    recycles an inflater's buffers once the replay has passed the chunk three before it; the replay releases a batch of SAMPLES as
    a whole. Whatever the numbers: a batch must never touch four consecutive chunks (it would wait for a chunk whose buffers it
    holds itself) -- the limit on its samples is two chunks' worth, so it touches at most three."""
    host = ctypes.CDLL(os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so"))
    host.pdh_loader_chunks.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    out = (ctypes.c_uint64 * 3)()
    for samples in (1, 40, 255, 256, 1000, 4000, 4033, 10000, 12288, 12289, 20000, 50000, 52224, 52225, 100000, 200000):
        for readers in sorted({1, samples // 7 + 1, samples // 2 + 1, max(1, samples - 1), samples}):
            assert host.pdh_loader_chunks(readers, samples, out) == 0
            first, chunk, batch = out[0], out[1], out[2]
            assert 256 <= chunk <= 17408 and first == chunk
            n_chunks = 1 + max(0, -(-(readers - first) // chunk)) if readers > first else 1
            if readers <= 52224:
                assert n_chunks <= 3, (samples, readers, n_chunks)          # one per inflater: nothing is recycled, and the records
                #                                                             a round's segment takes from the device are still there
            # a batch holds at most `batch` readers: they lie in at most three consecutive chunks
            assert batch <= 2 * 17408 and (n_chunks <= 3 or batch <= first + chunk), (samples, readers)


def test_parallel_passes_under_fire(built):
    """parallel.h: the parked workers serve one caller at a time, everybody else (other threads, nested passes) starts threads --
    concurrent callers, nested passes and passes that throw all add up."""
    host = ctypes.CDLL(os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so"))
    host.pdh_selftest_parallel.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
    assert host.pdh_selftest_parallel(4, 60, 200) == 0
    assert host.pdh_selftest_parallel(1, 40, 3) == 0
