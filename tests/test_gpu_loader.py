"""GPU: the loader changes of round 4 in the product binary -- profile blocks inflated on the device (GunzipPrefetch,
pd_inflater_*) and light end tables (counts per window while the device runs the window loop) -- against the host-only
configuration of round 3: same bytes on a cohort large enough for several prefetch chunks, with several regions, and with a
profile whose data is damaged (the device refuses the member, the host's zlib path takes the sample and decides what the
run does)."""
import os
import shutil
import subprocess

import pytest

from conftest import POPDEL_BIN, strip_date

pytestmark = pytest.mark.gpu


def call(cohort, out, *args, env=None):
    e = dict(os.environ, POPDEL_TIMING="1", POPDEL_DEVICE_GUNZIP_WAIT="1")     # (wait for the backend: every round takes the device path)
    e.pop("POPDEL_HIP_LANES", None)
    e.update(env or {})
    r = subprocess.run([POPDEL_BIN, "call", os.path.join(cohort, "profiles.txt"), "-o", out] + list(args), env=e,
                       capture_output=True, text=True)
    text = strip_date(open(out).read()) if r.returncode == 0 and os.path.exists(out) else None
    return r, text


def test_device_gunzip_and_light_tables_write_the_same_vcf(built, cohort_dir, tmp_path):
    from popdel_amd import synth
    d = cohort_dir("s600", 600, synth.single_deletion)
    host = {"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"}
    for args in (["-r", "chr20:1000000-2000000"], ["-r", "chr20:1000000-1300000", "-r", "chr20:1500000-2000000"], ["-r", "chr20:1003001-2000000", "-n"],
                 ["-r", "chr20:1000000-2000000", "-b", "50000"]):
        r1, new = call(d, str(tmp_path / "new.vcf"), *args)
        r0, old = call(d, str(tmp_path / "old.vcf"), *args, env=host)
        assert r0.returncode == 0 and r1.returncode == 0, (r0.stderr[-300:], r1.stderr[-300:])
        assert new == old and len(new.splitlines()) > 20, args
        # (the device did take the blocks: every round that began once the backend was up)
        line = [l for l in r1.stderr.splitlines() if "loader: profile blocks inflated on the device" in l]
        assert line, r1.stderr[-400:]
        assert int(line[0].split("for ")[1].split()[0]) >= 600 and "(0 left to the host)" in line[0], line[0]
        assert "loader:" not in r0.stderr


def test_backend_coming_up_in_the_middle_of_the_run(built, cohort_dir, tmp_path):
    """Without POPDEL_DEVICE_GUNZIP_WAIT the first rounds are inflated on the host while the backend starts, and the device's loader
    takes over at whatever round that is: same bytes as the host-only configuration, whenever that moment falls. (Round 6 tried
    guessing the round to come before the backend is up so that the first round on the device finds its blocks there: it does, and
    the second loader's cold start -- descriptors, page-locked and device buffers -- then shows in the round after it instead:
    no gain over the run, taken out again.)"""
    from popdel_amd import synth
    d = cohort_dir("s600", 600, synth.single_deletion)
    for args in (["-r", "chr20:1000000-2000000", "-b", "40000"], ["-r", "chr20:1003001-2000000", "-b", "25000", "-n"]):
        r0, old = call(d, str(tmp_path / "old.vcf"), *args, env={"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"})
        assert r0.returncode == 0, r0.stderr[-300:]
        for attempt in range(3):
            e = dict(os.environ, POPDEL_TIMING="1")
            e.pop("POPDEL_HIP_LANES", None)
            e.pop("POPDEL_DEVICE_GUNZIP_WAIT", None)
            out = str(tmp_path / "new.vcf")
            r1 = subprocess.run([POPDEL_BIN, "call", os.path.join(d, "profiles.txt"), "-o", out] + args, env=e, capture_output=True, text=True)
            assert r1.returncode == 0, r1.stderr[-400:]
            assert strip_date(open(out).read()) == old, (args, attempt)


def test_a_damaged_profile_behaves_as_on_the_host(built, cohort_dir, tmp_path):
    from popdel_amd import synth
    src = cohort_dir("s40", 40, synth.single_deletion)
    d = str(tmp_path / "cohort")
    shutil.copytree(src, d)
    with open(os.path.join(d, "profiles.txt"), "w") as fh:
        fh.write("\n".join(os.path.join(d, "s%05d.profile" % k) for k in range(40)) + "\n")
    # flip a byte in the middle of the data of one profile (inside some gzip member's deflate stream)
    victim = os.path.join(d, "s00007.profile")
    size = os.path.getsize(victim)
    with open(victim, "r+b") as fh:
        fh.seek(size * 2 // 3)
        b = fh.read(1)
        fh.seek(size * 2 // 3)
        fh.write(bytes([b[0] ^ 0x5a]))
    args = ["-r", "chr20:1000000-2000000"]
    r1, new = call(d, str(tmp_path / "new.vcf"), *args)
    r0, old = call(d, str(tmp_path / "old.vcf"), *args, env={"POPDEL_HOST_GUNZIP": "1"})
    assert r1.returncode == r0.returncode
    assert new == old
    if r0.returncode != 0:
        assert r1.stderr.splitlines()[-1] == r0.stderr.splitlines()[-1]


def test_loader_variants_write_the_same_vcf(built, cohort_dir, tmp_path):
    """The loader's switches -- blocks parsed on the host (POPDEL_HOST_PARSE=1), every round fetched when it begins
    (POPDEL_LOADER_AHEAD=0), page-locked memory from hipHostMalloc with spinning waits and no NUMA binding -- against the default
    (parsed on the device, the next round's blocks fetched ahead on a guess): same bytes; and the default did fetch rounds ahead."""
    from popdel_amd import synth
    d = cohort_dir("s600", 600, synth.single_deletion)
    args = ["-r", "chr20:1000000-2000000"]
    r, base = call(d, str(tmp_path / "base.vcf"), *args)
    assert r.returncode == 0, r.stderr[-300:]
    line = [l for l in r.stderr.splitlines() if "loader: profile blocks" in l][0]
    ahead = int(line.split("thread-seconds for it; ")[1].split()[0])
    parsed = int(line.split("parsed there for ")[1].split(";")[0])
    assert ahead >= 1 and parsed >= 600, line
    # (round 6: most records are filed from the device's block summaries, none with POPDEL_HOST_FILING=1 -- same bytes)
    filed = [l for l in r.stderr.splitlines() if "[popdel timing] filing:" in l][0]
    total, bulk = int(filed.split("filing: ")[1].split()[0]), int(filed.split("files filed in runs, ")[1].split()[0])
    assert total > 5000000 and bulk > 0.7 * total and " 0 refused by the tables" in filed, filed
    for env in ({"POPDEL_HOST_FILING": "1"}, {"POPDEL_HOST_PARSE": "1"}, {"POPDEL_LOADER_AHEAD": "0"}, {"POPDEL_HOST_PARSE": "1", "POPDEL_LOADER_AHEAD": "0"},
                {"POPDEL_HIP_PINNED": "malloc", "POPDEL_HIP_BLOCKING_SYNC": "0", "POPDEL_NO_NUMA_BIND": "1"}):
        r1, other = call(d, str(tmp_path / "other.vcf"), *args, env=env)
        assert r1.returncode == 0, r1.stderr[-300:]
        assert other == base, env
        line = [l for l in r1.stderr.splitlines() if "loader: profile blocks" in l][0]
        if "POPDEL_LOADER_AHEAD" in env:
            assert " 0 rounds fetched ahead" in line, line
        if "POPDEL_HOST_PARSE" in env:
            assert "parsed there for 0;" in line, line
        if "POPDEL_HOST_FILING" in env or "POPDEL_HOST_PARSE" in env:
            assert "files filed in runs, 0 of them" in [l for l in r1.stderr.splitlines() if "[popdel timing] filing:" in l][0]


def test_entries_the_device_kept_come_back_when_the_tables_need_them(built, cohort_dir, tmp_path):
    """Round 6: the records in the middle of a round's stretch stay on the device -- the start slots hold them as virtual runs, the
    end slots as deferred references, the window loop takes them from the loader's records there. When the tables do need the
    entries (the host's window loop taking over; a slot recycled under references) they are read again from the profiles:
    forced here -- every third sample's runs read back right after filing; the window loop handed back to the host at the
    second, third and a later round (with pairs active on the device and a slot switch pending: what the switch does to the active
    set is done to the pairs that come back, pd_walk_active_ex), also with small buffers so that slots are recycled while
    references to them are alive -- against the default, and against the host-only configuration."""
    from popdel_amd import synth
    d = cohort_dir("s600", 600, synth.single_deletion)
    for args, late in ((["-r", "chr20:1000000-2000000"], "4"), (["-r", "chr20:1003001-2000000", "-b", "60000"], "9")):
        r, base = call(d, str(tmp_path / "base.vcf"), *args)
        assert r.returncode == 0, r.stderr[-300:]
        filed = [l for l in r.stderr.splitlines() if "[popdel timing] filing:" in l][0]
        kept = int(filed.split("usable blocks); ")[1].split()[0])
        assert kept > 3000000, filed                                   # (most entries never left the device)
        r0, host = call(d, str(tmp_path / "host.vcf"), *args, env={"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"})
        assert r0.returncode == 0 and host == base
        for env in ({"POPDEL_TEST_MATERIALIZE": "3"}, {"POPDEL_TEST_LEAVE_WALK_AT": "3"}, {"POPDEL_TEST_LEAVE_WALK_AT": late},
                    {"POPDEL_TEST_LEAVE_WALK_AT": "2", "POPDEL_TEST_MATERIALIZE": "2"}, {"POPDEL_HOST_RECORDS": "1"}):
            r1, other = call(d, str(tmp_path / "other.vcf"), *args, env=env)
            assert r1.returncode == 0, (env, r1.stderr[-400:])
            assert other == base, (args, env)
            filed = [l for l in r1.stderr.splitlines() if "[popdel timing] filing:" in l][0]
            back = int(filed.split("never left the device, ")[1].split()[0])
            if "POPDEL_HOST_RECORDS" in env:
                assert " 0 start entries never left the device" in filed, filed
            else:
                assert back > 100000, (env, filed)                     # (entries were read again from the profiles)
            if "POPDEL_TEST_LEAVE_WALK_AT" in env:
                assert "walked on the host" in r1.stderr and " 0 walked on the host" not in r1.stderr, r1.stderr[-600:]


def two_contig_cohort(d, n):
    """profiles with data on two contigs (chr19: 500,000-1,050,000 with a 900-bp deletion, chr20: 1,000,000-1,500,000 with one of
    1,500 bp): what no other cohort of the suite has -- the loader crosses a contig border in the middle of a run. (The data of
    the first contig ends in the middle of a slot for every sample: no pair is left active at the border. With pairs left --
    samples that finish a round before the others are walked again by the reference's initializeActiveReads,
    profile_structure_popdel_call.h:1120-1135, and never deactivated -- the reference carries ids into recycled tables, undefined
    behaviour there, and the device's window loop hands over to the host's)"""
    from popdel_amd import synth
    contigs = (("chr19", 3000000), ("chr20", synth.CHR20_LEN))
    paths = []
    for k in range(n):
        lib = synth.Library(hist_seed=100 + k)
        p19, d19 = synth.sample_pairs(9000 + k, (500000, 1050000), lib, [synth.Deletion(800000, 900, [1, 2, 0][k % 3])])
        p20, d20 = synth.sample_pairs(9100 + k, (1000000, 1500000), lib, [synth.Deletion(1250000, 1500, [2, 0, 1][k % 3])])
        if k == 5:                          # (one sample's data on the second contig begins late and ends early)
            keep = (p20 >= 1150000) & (p20 < 1420000)
            p20, d20 = p20[keep], d20[keep]
        path = os.path.join(d, "m%02d.profile" % k)
        synth.write_profile(path, "M%02d" % k, [p19, p20], [d19, d20], lib, contigs=contigs, chrom=[0, 1])
        paths.append(path)
    with open(os.path.join(d, "profiles.txt"), "w") as fh:
        fh.write("\n".join(paths) + "\n")


def test_data_on_two_contigs(built, tmp_path):
    """The loader across a contig border (the tables are reset, the virtual runs and deferred references of the contig before are
    dropped, the loader that went ahead guessed the wrong contig): device loader with records staying on the device against the
    host-only configuration and against the CPU driver (oracle kernels), whole genome, window-wise, regions on both contigs,
    small buffers. Same bytes, and both planted deletions are called on their contigs."""
    from conftest import ORACLE_CALL
    d = str(tmp_path / "two")
    os.makedirs(d)
    two_contig_cohort(d, 40)
    for args in ([], ["-n"], ["-r", "chr19:600000-1000000", "-r", "chr20:1100000-1400000", "-n"], ["-b", "50000"]):
        r1, new = call(d, str(tmp_path / "new.vcf"), *args)
        r0, old = call(d, str(tmp_path / "old.vcf"), *args, env={"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"})
        assert r0.returncode == 0 and r1.returncode == 0, (args, r0.stderr[-300:], [l for l in r1.stderr.splitlines() if "timing" not in l][-6:])
        assert new == old, args
        subprocess.check_call([ORACLE_CALL, os.path.join(d, "profiles.txt"), "-o", str(tmp_path / "cpu.vcf")] + args, stdout=subprocess.DEVNULL)
        assert strip_date(open(str(tmp_path / "cpu.vcf")).read()) == new, args
        where = [(l.split("\t")[0], int(l.split("\t")[1])) for l in new.splitlines() if not l.startswith("#")]
        assert any(c == "chr19" and abs(p - 800000) < 400 for c, p in where) and any(c == "chr20" and abs(p - 1250000) < 400 for c, p in where), where[:5]
        filed = [l for l in r1.stderr.splitlines() if "[popdel timing] filing:" in l][0]
        assert int(filed.split("usable blocks); ")[1].split()[0]) > 500000, filed        # (entries stayed on the device on both contigs)


@pytest.mark.parametrize("region_size", [2000, 50000])
def test_uneven_profiles_behave_as_on_the_host(built, tmp_path, region_size):
    """Profiles that differ where the prefetch has to be careful: index regions of 2,000 or 50,000 bp instead of 10,000 (the
    guess for the round to come lands several regions away from where the round begins, or in the same one for rounds on
    end), stretches without any pair (index entries that repeat), a profile whose data ends early, a sparse one -- device
    loader against the host-only configuration, with the default and small window buffers (more rounds), same bytes."""
    from popdel_amd import synth
    d = str(tmp_path / "uneven")
    os.makedirs(d)
    region = (1000000, 1600000)
    paths = []
    for k in range(30):
        lib = synth.Library(hist_seed=300 + k, coverage=3.0 if k == 11 else 30.0)
        dels = [synth.Deletion(1200000, 1200, [1, 2, 0][k % 3]), synth.Deletion(1450000, 600, [2, 1, 1][k % 3])]
        pos, dev = synth.sample_pairs(7000 + k, region, lib, dels)
        if k % 4 == 1:                      # a stretch without pairs (several empty index regions)
            keep = (pos < 1280000 + 3000 * k) | (pos >= 1420000)
            pos, dev = pos[keep], dev[keep]
        if k == 17:                         # the data ends early
            keep = pos < 1330000
            pos, dev = pos[keep], dev[keep]
        p = os.path.join(d, "u%02d.profile" % k)
        synth.write_profile(p, "U%02d" % k, pos, dev, lib, region_size=region_size)
        paths.append(p)
    with open(os.path.join(d, "profiles.txt"), "w") as fh:
        fh.write("\n".join(paths) + "\n")
    for args in (["-r", "chr20:1000000-1600000"], ["-r", "chr20:1000000-1600000", "-b", "40000", "-n"], ["-r", "chr20:1030000-1590000", "-b", "70000"]):
        r1, new = call(d, str(tmp_path / "new.vcf"), *args)
        r0, old = call(d, str(tmp_path / "old.vcf"), *args, env={"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"})
        assert r0.returncode == 0 and r1.returncode == 0, (r0.stderr[-300:], r1.stderr[-300:])
        assert new == old and len(new.splitlines()) > 20, args
        line = [l for l in r1.stderr.splitlines() if "loader: profile blocks inflated on the device" in l]
        assert line, r1.stderr[-400:]
        print("region size %d, %s: %s" % (region_size, " ".join(args), line[0].split("thread-seconds for it; ")[1]))
        print([l for l in r1.stderr.splitlines() if "[popdel timing] filing:" in l][0])


def test_the_load_cap_bites_in_blocks_filed_from_summaries(built, tmp_path):
    """Where ChromosomeProfile::add's maxLoad rule drops records (profile_structure_popdel_call.h:1084-1112) a block cannot be filed
    from its counts: addBulk refuses it, the block's records are read again from the profile and filed one by one -- the same
    records dropped as on the host. Samples with stretches of 150-400x coverage (the default cap of 100 active pairs bites
    there), and caps of 12 / 25 / none from an -A file on samples without such stretches: device loader with records staying on
    the device against host filing, the host-only configuration and the CPU driver; same bytes, and blocks were refused."""
    import numpy as np
    from conftest import ORACLE_CALL
    from popdel_amd import synth
    d = str(tmp_path / "bursts")
    os.makedirs(d)
    region = (1000000, 1500000)
    paths = []
    for k in range(40):
        lib = synth.Library(hist_seed=500 + k)
        dels = [synth.Deletion(1200000, 1100, [1, 2, 0][k % 3]), synth.Deletion(1380000, 700, [2, 1, 1][k % 3])]
        pos, dev = synth.sample_pairs(8000 + k, region, lib, dels)
        if k % 5 == 2:                      # pile-ups: 3-6 kb at 150-400x, one of them across an index-region border
            for j, (at, length, cov) in enumerate(((1098500 + 40 * k, 3000, 150.0), (1301000, 6000, 400.0), (1449000 - 70 * k, 2500, 250.0))):
                heavy = synth.Library(hist_seed=500 + k, coverage=cov)
                p2, d2 = synth.sample_pairs(8500 + 3 * k + j, (at, at + length), heavy)
                pos, dev = np.concatenate([pos, p2]), np.concatenate([dev, d2])
            order = np.lexsort((dev, pos))
            pos, dev = pos[order], dev[order]
        p = os.path.join(d, "b%02d.profile" % k)
        synth.write_profile(p, "B%02d" % k, pos, dev, lib)
        paths.append(p)
    with open(os.path.join(d, "profiles.txt"), "w") as fh:
        fh.write("\n".join(paths) + "\n")
    caps = os.path.join(d, "caps.txt")
    with open(caps, "w") as fh:
        fh.write("B00 12\nB01 25\nB03 0\nB07 0\nB10 18\n")
    for args in (["-r", "chr20:1000000-1500000"], ["-r", "chr20:1000000-1500000", "-A", caps, "-n"], ["-r", "chr20:1003001-1500000", "-A", caps, "-b", "60000"]):
        r1, new = call(d, str(tmp_path / "new.vcf"), *args)
        assert r1.returncode == 0, r1.stderr[-400:]
        filed = [l for l in r1.stderr.splitlines() if "[popdel timing] filing:" in l][0]
        print(filed)
        refused = int(filed.split(" refused by the tables")[0].split()[-1])
        assert refused > 0 and int(filed.split("usable blocks); ")[1].split()[0]) > 200000, filed
        for env in ({"POPDEL_HOST_FILING": "1"}, {"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"}, {"POPDEL_TEST_LEAVE_WALK_AT": "3"}):
            r0, old = call(d, str(tmp_path / "old.vcf"), *args, env=env)
            assert r0.returncode == 0, (env, r0.stderr[-300:])
            assert new == old and len(new.splitlines()) > 20, (args, env)
        subprocess.check_call([ORACLE_CALL, os.path.join(d, "profiles.txt"), "-o", str(tmp_path / "cpu.vcf")] + args, stdout=subprocess.DEVNULL)
        assert strip_date(open(str(tmp_path / "cpu.vcf")).read()) == new, args
