"""Seeded synthetic PopDel profiles (file format: SURVEY.md Appendix B) and in-memory pair streams.

Two consumers:
  * tests: write real ``.profile`` files so the host driver can be run end to end; with the default arguments the
    records are identical to the ones the survey stage fed to the reference when it captured the golden outputs
    under ``tests/golden/`` (same RNG call order: per sample ``default_rng(1000 + k)``, per haplotype first the
    read positions, then the insert sizes);
  * bench.py: ``sample_pairs`` gives the (position, deviation) stream of one sample without touching the disk.
"""
from __future__ import annotations

import gzip
import os
import struct
from dataclasses import dataclass, field

import numpy as np

CHR20_LEN = 64444167
_REC = np.dtype([("off", "u1"), ("dev", "<i4")])


@dataclass
class Library:
    median: int = 500
    stddev: float = 50.0
    read_len: int = 150
    coverage: float = 30.0
    hist_seed: int = 12345
    hist_draws: int = 200000

    def histogram(self):
        """Counts trimmed to [max(1, floor(median - 3 sd)), ceil(median + 3 sd) + 1)."""
        rng = np.random.default_rng(self.hist_seed)
        samp = np.rint(rng.normal(self.median, self.stddev, self.hist_draws)).astype(int)
        samp = np.clip(samp, 1, None)          # wide libraries: no negative insert sizes
        full = np.bincount(samp, minlength=2000).astype(float)
        start = max(1, int(np.floor(self.median - 3 * self.stddev)))
        end = int(np.ceil(self.median + 3 * self.stddev)) + 1
        return start, full[start:end]


@dataclass
class Deletion:
    start: int
    length: int
    genotype: int  # 0, 1 or 2 deleted haplotypes


def sample_pairs(seed: int, region, lib: Library, deletions=()):
    """Return (pos, dev) of one sample, sorted by (pos, dev). ``pos`` is the end of the forward read."""
    lo, hi = region
    rng = np.random.default_rng(seed)
    n_pairs = int((hi - lo) * lib.coverage / (2.0 * lib.read_len))
    pos_all, dev_all = [], []
    for hap in range(2):
        n = n_pairs // 2
        s = rng.integers(lo, hi, size=n)
        ins = np.rint(rng.normal(lib.median, lib.stddev, size=n)).astype(np.int64)
        inner = np.maximum(ins - 2 * lib.read_len, 1)
        keep = np.ones(n, bool)
        for d in deletions:
            carrier = d.genotype == 2 or (d.genotype == 1 and hap == 0)
            if not carrier:
                continue
            keep &= ~((s >= d.start) & (s < d.start + d.length))
            span = (s < d.start) & (s + inner > d.start)
            ins = np.where(span, ins + d.length, ins)
        pos_all.append(s[keep])
        dev_all.append(ins[keep] - lib.median)
    pos = np.concatenate(pos_all)
    dev = np.concatenate(dev_all)
    order = np.lexsort((dev, pos))
    return pos[order].astype(np.int64), dev[order].astype(np.int32)


def write_profile(path, rg_name, pos, dev, lib: Library, contigs=(("chr20", CHR20_LEN),), chrom=0, region_size=10000,
                  compressed=True):
    """Write one single-read-group profile. ``pos``/``dev`` must be sorted by (pos, dev). Data on SEVERAL contigs: ``chrom`` a list
    of contig indices in increasing order, ``pos`` and ``dev`` lists of the same length (one pair of arrays per contig)."""
    segments = list(zip(chrom, pos, dev)) if isinstance(chrom, (list, tuple)) else [(chrom, pos, dev)]
    hist_start, hist = lib.histogram()
    num_regions = sum(l // region_size + 1 for _, l in contigs)
    hdr = bytearray(b"POPDEL\x01")
    hdr += struct.pack("<II", region_size, num_regions)
    idx_off = len(hdr)
    hdr += b"\0" * (8 * num_regions)
    hdr += struct.pack("<I", 1)
    hdr += struct.pack("<I", len(rg_name) + 1) + rg_name.encode() + b"\0"
    hdr += struct.pack("<IdIII", lib.median, lib.stddev, lib.read_len, hist_start, hist_start + len(hist))
    hdr += np.asarray(hist, dtype="<f8").tobytes()
    hdr += struct.pack("<I", len(contigs))
    for name, length in contigs:
        hdr += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", length)

    offsets = [0] * num_regions
    body = bytearray()
    base = len(hdr)
    for chrom, pos, dev in segments:
        first_slot = sum(l // region_size + 1 for _, l in contigs[:chrom])
        win = pos // 256
        # boundaries of the 256-bp windows
        cuts = np.flatnonzero(np.diff(win)) + 1
        starts = np.concatenate(([0], cuts)) if len(pos) else np.array([], dtype=np.int64)
        ends = np.concatenate((cuts, [len(pos)])) if len(pos) else np.array([], dtype=np.int64)
        rec = np.empty(len(pos), dtype=_REC)
        rec["off"] = (pos - win * 256).astype(np.uint8)
        rec["dev"] = dev
        raw = rec.tobytes()
        cur_region, blk = None, bytearray()

        def flush():
            nonlocal blk
            if cur_region is None:
                return
            offsets[first_slot + cur_region] = base + len(body)
            body.extend(gzip.compress(bytes(blk), 6, mtime=0) if compressed else blk)
            blk = bytearray()

        for a, b in zip(starts.tolist(), ends.tolist()):
            bp = int(win[a]) * 256
            r = bp // region_size
            if r != cur_region:
                flush()
                cur_region = r
            blk += struct.pack("<III", chrom, bp, b - a)
            blk += raw[a * 5:b * 5]
        flush()
    prev = base + len(body)
    for j in range(num_regions - 1, -1, -1):
        if offsets[j] == 0:
            offsets[j] = prev
        else:
            prev = offsets[j]
    hdr[idx_off:idx_off + 8 * num_regions] = struct.pack("<%dQ" % num_regions, *offsets)
    with open(path, "wb") as fh:
        fh.write(hdr)
        fh.write(body)


def write_profile_multi(path, read_groups, contigs=(("chr20", CHR20_LEN),), chrom=0, region_size=10000, compressed=True):
    """Write a profile with several read groups. ``read_groups`` = [(name, pos, dev, Library), ...] (each sorted)."""
    num_regions = sum(l // region_size + 1 for _, l in contigs)
    hdr = bytearray(b"POPDEL\x01")
    hdr += struct.pack("<II", region_size, num_regions)
    idx_off = len(hdr)
    hdr += b"\0" * (8 * num_regions)
    hdr += struct.pack("<I", len(read_groups))
    for name, _, _, lib in read_groups:
        hist_start, hist = lib.histogram()
        hdr += struct.pack("<I", len(name) + 1) + name.encode() + b"\0"
        hdr += struct.pack("<IdIII", lib.median, lib.stddev, lib.read_len, hist_start, hist_start + len(hist))
        hdr += np.asarray(hist, dtype="<f8").tobytes()
    hdr += struct.pack("<I", len(contigs))
    for name, length in contigs:
        hdr += struct.pack("<I", len(name) + 1) + name.encode() + b"\0" + struct.pack("<I", length)
    first_slot = sum(l // region_size + 1 for _, l in contigs[:chrom])
    offsets = [0] * num_regions
    base = len(hdr)
    body = bytearray()
    # per read group: window id -> slice
    per_rg = []
    all_windows = set()
    for _, pos, dev, _ in read_groups:
        win = pos // 256
        rec = np.empty(len(pos), dtype=_REC)
        rec["off"] = (pos - win * 256).astype(np.uint8)
        rec["dev"] = dev
        raw = rec.tobytes()
        cuts = np.flatnonzero(np.diff(win)) + 1
        starts = np.concatenate(([0], cuts)) if len(pos) else np.array([], dtype=np.int64)
        ends = np.concatenate((cuts, [len(pos)])) if len(pos) else np.array([], dtype=np.int64)
        table = {int(win[a]): (a, b) for a, b in zip(starts.tolist(), ends.tolist())}
        per_rg.append((table, raw))
        all_windows.update(table)
    cur_region, blk = None, bytearray()

    def flush():
        nonlocal blk
        if cur_region is None:
            return
        offsets[first_slot + cur_region] = base + len(body)
        body.extend(gzip.compress(bytes(blk), 6, mtime=0) if compressed else blk)
        blk = bytearray()

    for wdx in sorted(all_windows):
        bp = wdx * 256
        r = bp // region_size
        if r != cur_region:
            flush()
            cur_region = r
        blk += struct.pack("<II", chrom, bp)
        for table, raw in per_rg:
            a, b = table.get(wdx, (0, 0))
            blk += struct.pack("<I", b - a)
            blk += raw[a * 5:b * 5]
    flush()
    prev = base + len(body)
    for j in range(num_regions - 1, -1, -1):
        if offsets[j] == 0:
            offsets[j] = prev
        else:
            prev = offsets[j]
    hdr[idx_off:idx_off + 8 * num_regions] = struct.pack("<%dQ" % num_regions, *offsets)
    with open(path, "wb") as fh:
        fh.write(hdr)
        fh.write(body)


def single_deletion(k, region):
    """Survey set 'd': one 1000-bp deletion at the region midpoint, genotype het/hom/ref for k mod 3 = 0/1/2."""
    lo, hi = region
    return [Deletion(lo + (hi - lo) // 2, 1000, [1, 2, 0][k % 3])]


def dense_deletions(k, region):
    """Survey set 'dd': 49 deletions of 300..4956 bp, one per 20 kb, genotype pattern [1, 2, 0, 0][(k + j) % 4]."""
    lo, _ = region
    return [Deletion(lo + 10000 + 20000 * j, 300 + 97 * j, [1, 2, 0, 0][(k + j) % 4]) for j in range(49)]


def _write_cohort_part(args):
    outdir, ks, region, deletions, lib, compressed, seed_base = args
    for k in ks:
        pos, dev = sample_pairs(seed_base + k, region, lib, deletions(k, region) if deletions else ())
        write_profile(os.path.join(outdir, "s%05d.profile" % k), "RG%05d" % k, pos, dev, lib, compressed=compressed)
    return len(ks)


def write_cohort(outdir, n_samples, region=(1000000, 2000000), deletions=single_deletion, lib: Library | None = None,
                 compressed=True, seed_base=1000, workers=1):
    """Write ``n_samples`` profiles plus ``profiles.txt``; returns the list file path. ``workers`` > 1 writes the
    profiles from that many child processes -- fresh interpreters running this module, not forks: the caller may hold a
    GPU runtime and gigabytes of page-locked memory (``deletions`` must then be a function of this module, ``lib`` the
    default library)."""
    os.makedirs(outdir, exist_ok=True)
    paths = [os.path.join(outdir, "s%05d.profile" % k) for k in range(n_samples)]
    workers = max(1, min(int(workers), n_samples))
    if workers == 1:
        _write_cohort_part((outdir, list(range(n_samples)), region, deletions, lib or Library(), compressed, seed_base))
    else:
        import subprocess
        import sys
        if lib is not None or (deletions is not None and globals().get(deletions.__name__) is not deletions):
            raise ValueError("write_cohort(workers > 1) takes the default library and a deletion function of popdel_amd.synth")
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        procs = [subprocess.Popen([sys.executable, "-m", "popdel_amd.synth", outdir, str(n_samples), str(w), str(workers),
                                   str(region[0]), str(region[1]), deletions.__name__ if deletions else "-",
                                   "1" if compressed else "0", str(seed_base)], cwd=root) for w in range(workers)]
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError("a profile writer process failed")
    lst = os.path.join(outdir, "profiles.txt")
    with open(lst, "w") as fh:
        fh.write("\n".join(paths) + "\n")
    return lst


# ---------------------------------------------------------------------------------------------------------------------
# In-memory cohorts for the C ABI (no files): conditioned read groups and pair instances by the interval rule.
# ---------------------------------------------------------------------------------------------------------------------
def conditioned_readgroup(lib: Library, max_load=100, smoothing=True, pseudo_count_fraction=500):
    """ReadGroup record as `popdel call` would hold it after loading a profile header written from ``lib``."""
    import ctypes as C

    from . import capi

    host = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libpopdel_host.so"))
    start, hist = lib.histogram()
    values = np.ascontiguousarray(hist, dtype=np.float64).copy()
    min_prob = C.c_double()
    lo, hi = C.c_uint(), C.c_uint()
    rc = host.pdh_condition_histogram(values.ctypes.data_as(C.POINTER(C.c_double)), C.c_uint(len(values)), C.c_int(start),
                                      C.c_uint(lib.median), C.c_double(lib.stddev), C.c_uint(lib.read_len),
                                      C.c_int(int(smoothing)), C.c_uint(pseudo_count_fraction), C.byref(min_prob),
                                      C.byref(lo), C.byref(hi))
    if rc != 0:
        raise RuntimeError("pdh_condition_histogram failed")
    return capi.ReadGroup(median=lib.median, read_length=lib.read_len, stddev=lib.stddev, min_prob=min_prob.value,
                          hist_offset=start, hist_values=values, lower_quantile_dist=lo.value,
                          upper_quantile_dist=hi.value, max_load=max_load,
                          min_init_del_len=int(np.floor(4 * lib.stddev + 0.5)))


def instances_by_interval_rule(pos, dev, lib: Library, first_window_pos, n_windows, shift=30):
    """Window indices [on, off) of every pair under the reference's ideal activity rule:
    a pair with start s and end e = s + max(0, dev + median - 2 readLen) is active for grid positions c with
    s <= c <= floor(e / 30) * 30 + 30. Pairs outside the window range are dropped. Returns (dev, start, on, off)."""
    pos = np.asarray(pos, dtype=np.int64)
    dev = np.asarray(dev, dtype=np.int64)
    end = pos + np.maximum(0, dev + lib.median - 2 * lib.read_len)
    last = (end // shift) * shift + shift
    on = np.maximum(0, -((first_window_pos - pos) // shift))        # ceil((pos - first) / shift)
    off = (last - first_window_pos) // shift + 1
    off = np.minimum(off, n_windows)
    keep = (off > on) & (on < n_windows)
    order = np.argsort(on[keep], kind="stable")
    return (dev[keep][order].astype(np.int32), pos[keep][order].astype(np.uint32), on[keep][order].astype(np.uint32),
            off[keep][order].astype(np.uint32))


def cohort_segment(n_samples, region, first_window_pos, n_windows, lib: Library | None = None, deletions=single_deletion,
                   seed_base=1000):
    """One pd_segment for ``n_samples`` single-read-group samples (test sizes; everything goes through numpy)."""
    from . import capi

    lib = lib or Library()
    devs, starts, ons, offs, offsets = [], [], [], [], [0]
    for k in range(n_samples):
        pos, dev = sample_pairs(seed_base + k, region, lib, deletions(k, region) if deletions else ())
        d, s, a, b = instances_by_interval_rule(pos, dev, lib, first_window_pos, n_windows)
        devs.append(d)
        starts.append(s)
        ons.append(a)
        offs.append(b)
        offsets.append(offsets[-1] + len(d))
    return capi.Segment(first_window_pos=first_window_pos, n_windows=n_windows, rg_offsets=np.array(offsets, dtype=np.uint64),
                        dev=np.concatenate(devs), start=np.concatenate(starts), on=np.concatenate(ons),
                        off=np.concatenate(offs))


if __name__ == "__main__":          # worker of write_cohort(workers > 1): the samples w, w + workers, ... of the cohort
    import sys
    _out, _n, _w, _workers, _r0, _r1, _dels, _comp, _seed = sys.argv[1:10]
    _write_cohort_part((_out, list(range(int(_w), int(_n), int(_workers))), (int(_r0), int(_r1)),
                        globals()[_dels] if _dels != "-" else None, Library(), _comp == "1", int(_seed)))
