"""GPU: the product binary (HIP backend) against the oracle-backed driver for command-line variants that have no
reference capture; both read the same profiles, outputs must be byte-identical."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ORACLE_CALL, POPDEL_BIN, strip_date

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mixed_cohort(tmp_path_factory):
    """12 samples: heterogeneous libraries (median 350..550, sd 40..100), two 3-read-group samples, one low coverage."""
    from popdel_amd import synth
    d = tmp_path_factory.mktemp("mixed")
    region = (1000000, 1400000)
    rs = np.random.default_rng(7)
    dels = lambda k: [synth.Deletion(1100000, 800, [1, 2, 0][k % 3]), synth.Deletion(1250000, 2500, [2, 0, 1][k % 3]),
                      synth.Deletion(1330000, 350, [1, 1, 2][k % 3])]
    paths = []
    for k in range(12):
        lib = synth.Library(median=int(rs.integers(350, 551)), stddev=float(rs.integers(40, 101)), hist_seed=100 + k,
                            coverage=2.0 if k == 5 else 30.0)
        p = os.path.join(str(d), "m%02d.profile" % k)
        if k in (3, 8):
            rgs = []
            for j in range(3):
                sub = synth.Library(median=lib.median + 10 * j, stddev=lib.stddev, hist_seed=200 + 3 * k + j, coverage=10.0)
                pos, dev = synth.sample_pairs(5000 + 10 * k + j, region, sub, dels(k))
                rgs.append(("M%02d_RG%d" % (k, j), pos, dev, sub))
            synth.write_profile_multi(p, rgs)
        else:
            pos, dev = synth.sample_pairs(5000 + 10 * k, region, lib, dels(k))
            synth.write_profile(p, "M%02d" % k, pos, dev, lib)
        paths.append(p)
    lst = os.path.join(str(d), "profiles.txt")
    open(lst, "w").write("\n".join(paths) + "\n")
    cov = os.path.join(str(d), "maxcov.txt")
    open(cov, "w").write("M00 12\nM01 0\nM03_RG1 8\n")
    rois = os.path.join(str(d), "rois.txt")
    open(rois, "w").write("chr20:1050001-1150000\nchr20:1240001-1290000\n")
    return str(d)


def run(binary, cohort, out, args):
    cmd = [binary] + (["call"] if os.path.basename(binary) == "popdel" else []) + [os.path.join(cohort, "profiles.txt"), "-o", out] + args
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return strip_date(open(out).read())


OPTION_SETS = [
    ["-r", "chr20:1000000-1400000"],
    ["-r", "chr20:1000000-1400000", "-n", "-F"],
    ["-r", "chr20:1000000-1400000", "-a", "15", "-n"],
    ["-r", "chr20:1000000-1400000", "-u", "-n"],
    ["-r", "chr20:1000000-1400000", "-C", "-n"],
    ["-r", "chr20:1000000-1400000", "-l", "250", "-m", "220", "-t", "4", "-n"],
    ["-r", "chr20:1000000-1400000", "-s", "1.0", "-F"],
    ["-r", "chr20:1000000-1400000", "-f", "100", "-p", "0.01", "-c", "0.2"],
    ["-r", "chr20:1050001-1150000", "-r", "chr20:1240001-1290000", "-n"],
    ["-b", "100000", "-r", "chr20:1000000-1400000", "-n"],
]


@pytest.mark.parametrize("opts", OPTION_SETS, ids=lambda o: "_".join(x.replace(":", "").replace("-", "") for x in o)[:60])
def test_product_matches_oracle_driver(built, mixed_cohort, tmp_path, opts):
    a = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "gpu.vcf"), list(opts))
    b = run(ORACLE_CALL, mixed_cohort, str(tmp_path / "cpu.vcf"), list(opts))
    assert a.count("\n") > 25            # header + at least one record
    assert a == b


def test_files_for_coverage_and_rois(built, mixed_cohort, tmp_path):
    opts = ["-A", os.path.join(mixed_cohort, "maxcov.txt"), "-R", os.path.join(mixed_cohort, "rois.txt"), "-n"]
    a = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "gpu.vcf"), opts)
    b = run(ORACLE_CALL, mixed_cohort, str(tmp_path / "cpu.vcf"), opts)
    assert a == b and a.count("\n") > 25


def test_more_than_65535_windows_per_segment(built, tmp_path):
    """-b 3000000 makes one segment of ~73,000 windows: window numbers no longer fit the packed 16-bit image, every tile
    takes the 32-bit fallback kernel; output must not change."""
    from popdel_amd import synth
    d = str(tmp_path / "long")
    synth.write_cohort(d, 6, region=(1000000, 3200000),
                       deletions=lambda k, r: [synth.Deletion(1500000, 900, [1, 2, 0][k % 3]), synth.Deletion(2900000, 1500, [2, 0, 1][k % 3])])
    opts = ["-r", "chr20:1000000-3200000", "-b", "3000000", "-n"]
    a = run(POPDEL_BIN, d, str(tmp_path / "gpu.vcf"), opts)
    b = run(ORACLE_CALL, d, str(tmp_path / "cpu.vcf"), opts)
    assert a == b and a.count("\n") > 60


@pytest.mark.parametrize("opts", OPTION_SETS + [["-R", "rois.txt", "-A", "maxcov.txt", "-n"]],
                         ids=lambda o: "_".join(x.replace(":", "").replace("-", "") for x in o)[:60])
def test_window_loop_on_the_device_equals_the_walk_on_the_host(built, mixed_cohort, tmp_path, opts):
    """`popdel call` hands the segments over as the loader's tables hold them and the library runs the window loop in closed
    form on the device (pd_walk_submit); POPDEL_HOST_WALK=1 walks the windows on the host as the reference does and hands
    over pair instances (pd_segment_submit). Same VCF, byte for byte -- and the device path must really have run."""
    import re
    opts = [os.path.join(mixed_cohort, o) if o.endswith(".txt") else o for o in opts]
    outs = []
    for host_walk in (False, True):
        env = dict(os.environ, POPDEL_TIMING="1")
        if host_walk:
            env["POPDEL_HOST_WALK"] = "1"
        out = str(tmp_path / ("h.vcf" if host_walk else "d.vcf"))
        r = subprocess.run([POPDEL_BIN, "call", os.path.join(mixed_cohort, "profiles.txt"), "-o", out] + opts, env=env,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        m = re.search(r"window loop: (\d+) segments in closed form on the device, (\d+) walked on the host", r.stderr)
        assert m, r.stderr[-800:]
        on_device, on_host = int(m.group(1)), int(m.group(2))
        assert (on_device == 0 and on_host >= 2) if host_walk else (on_device >= 2 and on_host == 0), (on_device, on_host)
        outs.append(strip_date(open(out).read()))
    assert outs[0] == outs[1] and outs[0].count("\n") > 25


@pytest.mark.parametrize("opts", [["-r", "chr20:1000000-1400000"], ["-r", "chr20:1000000-1400000", "-A", "MAXCOV", "-n"],
                                  ["-r", "chr20:1050001-1150000", "-r", "chr20:1240001-1290000", "-n"],
                                  ["-b", "100000", "-r", "chr20:1000000-1400000", "-n"]],
                         ids=["default", "coverage_file", "two_regions", "small_buffer"])
def test_device_loader_on_the_mixed_cohort(built, mixed_cohort, tmp_path, opts):
    """The round-4 loader (blocks inflated and parsed on the device, fetched a round ahead, light end tables, records filed in
    blocks) against the host-only configuration of round 3 on the cohort with three-read-group samples (their blocks are
    inflated on the device and parsed by the host), heterogeneous libraries, a low-coverage sample and read groups whose
    active-coverage cap binds (records dropped by the rule: filed one by one). Same bytes."""
    opts = [os.path.join(mixed_cohort, "maxcov.txt") if o == "MAXCOV" else o for o in opts]

    def call(out, env):
        e = dict(os.environ, POPDEL_TIMING="1", **env)
        e.pop("POPDEL_HIP_LANES", None)
        r = subprocess.run([POPDEL_BIN, "call", os.path.join(mixed_cohort, "profiles.txt"), "-o", out] + opts, env=e, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-300:]
        return strip_date(open(out).read()), r.stderr

    new, log = call(str(tmp_path / "new.vcf"), {"POPDEL_DEVICE_GUNZIP_WAIT": "1"})
    old, _ = call(str(tmp_path / "old.vcf"), {"POPDEL_HOST_GUNZIP": "1", "POPDEL_HEAVY_TABLES": "1"})
    assert new == old and new.count("\n") > 25
    line = [l for l in log.splitlines() if "loader: profile blocks inflated on the device" in l]
    assert line and int(line[0].split("for ")[1].split()[0]) >= 12, log[-400:]


def test_option_files_mean_what_their_options_mean(built, mixed_cohort, tmp_path):
    """Two checks of the option plumbing that need no second implementation: regions from a file (-R) against the same regions
    given with -r, and a coverage file (-A) that gives every read group the value against -a with that value -- same bytes."""
    from popdel_amd import synth
    rois = os.path.join(mixed_cohort, "rois.txt")
    a = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "file.vcf"), ["-R", rois, "-n"])
    b = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "opts.vcf"), ["-r", "chr20:1050001-1150000", "-r", "chr20:1240001-1290000", "-n"])
    assert a == b and a.count("\n") > 25
    # every read group of the cohort at 15: the three-read-group samples by their read groups' names
    names = []
    for k in range(12):
        names += ["M%02d_RG%d" % (k, j) for j in range(3)] if k in (3, 8) else ["M%02d" % k]
    cov = str(tmp_path / "all15.txt")
    with open(cov, "w") as fh:
        fh.write("".join("%s 15\n" % n for n in names))
    a = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "covfile.vcf"), ["-r", "chr20:1000000-1400000", "-A", cov, "-n"])
    b = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "covopt.vcf"), ["-r", "chr20:1000000-1400000", "-a", "15", "-n"])
    assert a == b and a.count("\n") > 25
    c = run(POPDEL_BIN, mixed_cohort, str(tmp_path / "covdefault.vcf"), ["-r", "chr20:1000000-1400000", "-n"])
    assert c != a                                   # (the value does something on this cohort)
