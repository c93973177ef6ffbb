"""CPU: the loader's table filing (ChromProfile::addRun, chrom_profile.cpp) with light end tables -- counts per window, records
filed in blocks with the end table asked once per block -- against the heavy tables (the reference's ordered inserts and
getEndCount(), profile_structure_popdel_call.h:760-790, :1003-1062, record by record): the same loads, cursors, slot
contents and plans (pdh_bench_filing_ex mixes them into a checksum round by round and entry by entry at the end), for
everyday streams, bursts that reach the load cap, records out of order, wide inserts and gaps. Round 6: the same for runs
filed from their SUMMARY (ChromProfile::addBulk: starts and ends counted per window, nothing else known about the records;
the end entries deferred to a reference into the start slot and written out when light mode is left) -- the summary comes from
summarizeRun(), the host twin of the device's kernel (pd_inflater_summarize, tests/test_gpu_summarize.py)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "popdel_amd", "lib", "libpopdel_host.so")


@pytest.fixture(scope="module")
def filing():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(LIB)
    f = lib.pdh_bench_filing_ex
    f.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                  ctypes.POINTER(ctypes.c_uint64)]
    f.restype = ctypes.c_int

    def run(rounds, light, cap, seed, flavour):
        s, c = ctypes.c_double(), ctypes.c_uint64()
        assert f(rounds, light, cap, seed, flavour, ctypes.byref(s), ctypes.byref(c)) == 0
        return c.value
    return run


@pytest.mark.parametrize("flavour", [0, 1, 2, 3, 4])
def test_light_tables_file_what_heavy_tables_file(filing, flavour):
    for cap in (12, 25, 40, 100, 100000):
        for seed in (1, 2, 3, 4):
            assert filing(30, 1, cap, seed, flavour) == filing(30, 0, cap, seed, flavour), (flavour, cap, seed)


@pytest.mark.parametrize("flavour", [0, 1, 2, 3, 4])
def test_runs_filed_from_their_summaries_file_what_heavy_tables_file(filing, flavour):
    lib = ctypes.CDLL(LIB)
    counts = lib.pdh_bench_filing_bulk_counts
    counts.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 2
    runs0, runs1 = ctypes.c_uint64(), ctypes.c_uint64()
    counts(ctypes.byref(runs0), None)
    for cap in (12, 25, 40, 100, 100000):
        for seed in (1, 2, 3, 4, 5, 6):
            assert filing(30, 2, cap, seed, flavour) == filing(30, 0, cap, seed, flavour), (flavour, cap, seed)
    counts(ctypes.byref(runs1), None)
    # (not vacuous -- 900 rounds each: on the everyday stream the rounds whose cap leaves room go the new way, 537 of them; with
    # bursts or wide inserts fewer do; a stream with records out of order is refused and filed record by record)
    if flavour in (0, 1, 3):
        assert runs1.value - runs0.value > {0: 400, 1: 100, 3: 250}[flavour]
    if flavour == 2:
        assert runs1.value == runs0.value


@pytest.mark.parametrize("flavour", [0, 1, 3, 4])
def test_virtual_start_runs_file_what_heavy_tables_file(filing, flavour):
    """... and with the summarised runs' start entries not even copied (StartSlots::VirtualRun: the device keeps them, the table
    knows their number and extremes): the same checksums round by round -- plans, cursors, loads -- and, after the entries have
    been read back (ChromProfile::materializeStarts through the filer's refill), entry by entry."""
    for cap in (25, 40, 100, 100000):
        for seed in (1, 2, 3, 4, 5, 6):
            assert filing(30, 4, cap, seed, flavour) == filing(30, 0, cap, seed, flavour), (flavour, cap, seed)


def test_the_checksum_sees_the_cap(filing):
    # (the check is not blind: a different cap files different records in a burst)
    assert filing(30, 1, 25, 1, 1) != filing(30, 1, 100, 1, 1)
