// Small C exports of host-side start-up code for Python callers (bench.py / tests build synthetic cohorts in memory
// and need exactly the histogram conditioning that `popdel call` applies when it loads a profile header).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <zlib.h>

#include "call_params.h"
#include "chrom_profile.h"
#include "histogram.h"
#include <atomic>
#include <stdexcept>
#include "inflate_core.h"
#include "loader_chunks.h"
#include "parallel.h"
#include "parallel.h"
#include "profile_io.h"
#include "window_call.h"

extern "C" {

// In: raw counts `values[n]` with table offset / median / stddev / read length. Out: conditioned densities written
// back into `values`, plus min_prob and the two quantile distances.
int pdh_condition_histogram(double * values, unsigned n, int offset, unsigned median, double stddev, unsigned readLength,
                            int smoothing, unsigned pseudoCountFraction, double * minProbOut, unsigned * lowerQOut,
                            unsigned * upperQOut)
{
    if (!values || n < 3 || !minProbOut || !lowerQOut || !upperQOut)
        return 1;
    pdh::InsertHistogram h;
    h.values.assign(values, values + n);
    h.offset = offset;
    h.median = median;
    h.stddev = stddev;
    h.readLength = readLength;
    pdh::conditionHistogram(h, 256, smoothing != 0, pseudoCountFraction);
    std::memcpy(values, h.values.data(), n * sizeof(double));
    *minProbOut = h.minProb;
    *lowerQOut = h.lowerQuantileDist;
    *upperQOut = h.upperQuantileDist;
    return 0;
}

// Micro-benchmark of the two host passes behind the kernels (tools/bench_host_post.py): `windows` synthetic window calls
// of one merge group with `samples` sample columns each are merged (unifyCalls) and the surviving record is formatted
// `records` times. Returns seconds in out[0] (merge) and out[1] (format), bytes of one formatted record in out[2].
int pdh_bench_merge_and_format(unsigned samples, unsigned windows, unsigned records, double * out)
{
    if (!out || samples == 0 || windows < 2)
        return 1;
    std::mt19937 gen(12345);
    auto rng = [&gen]() { return static_cast<uint32_t>(gen()); };
    std::vector<pdh::WindowCall> calls(windows);
    for (unsigned w = 0; w < windows; ++w)
    {
        pdh::WindowCall & c = calls[w];
        c.initialLength = 1000;
        c.iterations = 3;
        c.deletionLength = 1000 + (w % 3);
        c.likelihoodRatio = 500.0 + w;
        c.frequency = 0.2;
        c.windowPosition = 1000000 + 30 * w;
        c.position = 1000010;
        c.endPosition = 1001020;
        c.filter = 0;
        c.pl.resize(samples);
        c.lad.resize(samples);
        c.dad.resize(samples);
        c.firstLast.resize(samples);
        for (unsigned s = 0; s < samples; ++s)
        {
            c.pl[s] = {rng() % 300u, rng() % 50u, rng() % 900u};
            c.pl[s][rng() % 3u] = 0;
            c.lad[s] = {rng() % 30u, rng() % 5u, rng() % 20u};
            c.dad[s] = {rng() % 30u, rng() % 5u, rng() % 3u, rng() % 20u, rng() % 2u};
            c.firstLast[s] = {999000u + rng() % 500u, 1000500u + rng() % 500u};
        }
    }
    using clock = std::chrono::steady_clock;
    const auto t0 = clock::now();
    pdh::unifyCalls(calls, 50.0, 0.5, false);
    const auto t1 = clock::now();
    out[0] = std::chrono::duration<double>(t1 - t0).count();
    out[1] = 0;
    out[2] = 0;
    if (calls.empty())
        return 0;
    const pdh::QuantileTable q(0.0001);
    size_t bytes = 0;
    const auto t2 = clock::now();
    for (unsigned r = 0; r < records; ++r)
        bytes = pdh::formatRecord(q, calls[0], "chr20").size();
    out[1] = std::chrono::duration<double>(clock::now() - t2).count();
    out[2] = static_cast<double>(bytes);
    return 0;
}

// Self-test of the bulk decoder (tests/test_abi_and_host.py): `rounds` random 256-bp window records -- one to three read
// groups, empty read groups, records out of position order now and then -- are decoded by readWindow() + splitWindowFlat()
// and by readWindowHead() + readWindowBodyFlat() (what `popdel call` uses); windows, counts and records must be the same,
// and so must the verdict "not position-sorted". Returns 0 if they are, else 1 + the number of the first round that differs.
int pdh_selftest_window_decode(unsigned seed, unsigned rounds)
{
    struct MemorySource : pdh::ByteSource
    {
        const std::vector<unsigned char> & bytes;
        size_t at = 0;
        explicit MemorySource(const std::vector<unsigned char> & b) : bytes(b) {}
        size_t read(void * dst, size_t n) override
        {
            const size_t take = std::min(n, bytes.size() - at);
            std::memcpy(dst, bytes.data() + at, take);
            at += take;
            return take;
        }
    };
    std::mt19937 gen(seed);
    for (unsigned round = 0; round < rounds; ++round)
    {
        const unsigned nRg = 1 + gen() % 3;
        const uint32_t chrom = gen() % 3, begin = 256u * (gen() % 100000u);
        std::vector<unsigned char> bytes;
        auto put32 = [&bytes](uint32_t v) {
            for (int i = 0; i < 4; ++i)
                bytes.push_back(static_cast<unsigned char>(v >> (8 * i)));
        };
        put32(chrom);
        put32(begin);
        const bool shuffle = gen() % 8 == 0;
        for (unsigned rg = 0; rg < nRg; ++rg)
        {
            const uint32_t n = gen() % 4 == 0 ? 0u : gen() % 60;
            std::vector<unsigned> offs(n);
            for (unsigned & o : offs)
                o = gen() % 256;
            std::sort(offs.begin(), offs.end());
            if (shuffle && n > 1)
                std::swap(offs[gen() % n], offs[gen() % n]);
            put32(n);
            for (uint32_t i = 0; i < n; ++i)
            {
                bytes.push_back(static_cast<unsigned char>(offs[i]));
                put32(static_cast<uint32_t>(static_cast<int32_t>(gen() % 2000) - 1000));
            }
        }
        pdh::FlatWindows winsA, winsB;
        pdh::FlatCounts countsA, countsB;
        pdh::FlatRecords recsA, recsB;
        // (something in front, as in a stream that already holds windows)
        winsA.push_back(pdh::FlatWindow{7, 30, 0});
        countsA.assign(nRg, 1);
        recsA.assign(nRg, pdh::PairRecord{31, 5});
        winsB = winsA;
        countsB = countsA;
        recsB = recsA;
        int a = -1, b = -1;
        try
        {
            MemorySource src(bytes);
            pdh::ProfileWindow win;
            a = pdh::readWindow(src, win, nRg) && pdh::splitWindowFlat(win, winsA, countsA, recsA, 256, 30) ? 1 : 0;
        }
        catch (const pdh::ProfileError &)
        {
            a = 2;
        }
        try
        {
            MemorySource src(bytes);
            pdh::WindowHead head;
            b = pdh::readWindowHead(src, head) && pdh::readWindowBodyFlat(src, head, nRg, winsB, countsB, recsB, 256, 30) ? 1 : 0;
        }
        catch (const pdh::ProfileError &)
        {
            b = 2;
        }
        bool same = a == b;
        if (same && a != 2)
        {
            same = winsA.size() == winsB.size() && countsA.size() == countsB.size() && recsA.size() == recsB.size();
            for (size_t i = 0; same && i < winsA.size(); ++i)
                same = winsA[i].chrom == winsB[i].chrom && winsA[i].beginPos == winsB[i].beginPos && winsA[i].firstRec == winsB[i].firstRec;
            for (size_t i = 0; same && i < countsA.size(); ++i)
                same = countsA[i] == countsB[i];
            for (size_t i = 0; same && i < recsA.size(); ++i)
                same = recsA[i].pos == recsB[i].pos && recsA[i].dev == recsB[i].dev;
        }
        if (!same)
            return 1 + static_cast<int>(round);
    }
    return 0;
}

// Self-test of the two inflaters (tests/test_abi_and_host.py): the compressed stream of `path` from byte `offset` on, read
// through zlib and through the source `popdel call` uses (libdeflate where the system has it), in requests of `request`
// bytes. Returns 0 if both deliver the same bytes (their number in *bytes), 1 if not, 2 if the file cannot be opened.
int pdh_selftest_inflate(const char * path, unsigned long long offset, unsigned request, unsigned long long * bytes)
{
    if (!path || request == 0 || !bytes)
        return 2;
    try
    {
        pdh::ProfileFile a(path), b(path);
        if (fseeko(a.handle(), static_cast<off_t>(offset), SEEK_SET) != 0 || fseeko(b.handle(), static_cast<off_t>(offset), SEEK_SET) != 0)
            return 2;
        auto za = a.openSource(false, true);
        auto zb = b.openSource(false, false);
        std::vector<unsigned char> ba(request), bb(request);
        unsigned long long total = 0;
        for (;;)
        {
            const size_t na = za->read(ba.data(), request), nb = zb->read(bb.data(), request);
            if (na != nb || std::memcmp(ba.data(), bb.data(), na) != 0)
                return 1;
            total += na;
            if (na < request)
                break;
        }
        *bytes = total;
        return 0;
    }
    catch (const std::exception &)
    {
        return 2;
    }
}


// The device loader's chunking of a round's readers and the replay's batch limit (loader_chunks.h), for the test that holds
// the relation between them: out[0] = first chunk, out[1] = the other chunks (for `readers`), out[2] = batch limit (for `samples`).
int pdh_loader_chunks(uint64_t readers, uint64_t samples, uint64_t * out)
{
    if (!out)
        return 1;
    out[0] = pdh::loaderFirstChunkFor(readers);
    out[1] = pdh::loaderChunkFor(readers);
    out[2] = pdh::loaderBatchLimitFor(samples);
    return 0;
}

// parallelFor() under fire (parallel.h: parked workers when they are free, threads per pass otherwise): `callers` threads
// run `rounds` passes each over `n` items at the same time, every fourth item starts a nested pass, one pass in sixteen
// throws from an item. Returns 0 if every pass added up (or re-threw), else the number of passes that did not.
int pdh_selftest_parallel(unsigned callers, unsigned rounds, unsigned n)
{
    std::atomic<unsigned> bad(0);
    std::vector<std::thread> pool;
    for (unsigned c = 0; c < callers; ++c)
        pool.emplace_back([&, c] {
            for (unsigned r = 0; r < rounds; ++r)
            {
                std::atomic<uint64_t> sum(0);
                const bool throws = (r + c) % 16u == 5u;
                bool threw = false;
                try
                {
                    pdh::parallelFor(n, [&](size_t i) {
                        if (throws && i == n / 2)
                            throw std::runtime_error("item");
                        uint64_t add = i + 1;
                        if (i % 4u == 0)
                        {
                            std::atomic<uint64_t> inner(0);
                            pdh::parallelFor(8, [&](size_t j) { inner += j; });
                            add += inner.load() - 28u;          // (0 + ... + 7)
                        }
                        sum += add;
                    }, 1 + r % 3u);
                }
                catch (const std::runtime_error &)
                {
                    threw = true;
                }
                if (throws ? !threw : (threw || sum.load() != static_cast<uint64_t>(n) * (n + 1) / 2))
                    ++bad;
            }
        });
    for (std::thread & t : pool)
        t.join();
    return static_cast<int>(bad.load());
}

// What the gunzip core's symbol table says (pdinf::symbolKind: one formula for the lengths and one for the distances) against
// RFC 1951's own tables, section 3.2.5, written out. Returns 0, or 1 + the first entry that differs.
int pdh_selftest_symbol_kinds()
{
    static const uint16_t lengthBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lengthExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t distBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                          8193, 12289, 16385, 24577};
    static const uint8_t distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (unsigned i = 0; i < pdinf::kSymbolKinds; ++i)
    {
        const uint32_t k = pdinf::symbolKind(i);
        const uint32_t extra = k & 15u, m = (k >> 4) & 0xffu, flags = k & (pdinf::kKindLiteral | pdinf::kKindEnd | pdinf::kKindNone);
        bool fine;
        if (i < 256u)
            fine = flags == pdinf::kKindLiteral;
        else if (i == 256u)
            fine = flags == pdinf::kKindEnd;
        else if (i < 286u)
            fine = flags == 0u && extra == lengthExtra[i - 257u] && 3u + (m << extra) == lengthBase[i - 257u];
        else if (i < 288u)
            fine = flags == pdinf::kKindNone;
        else if (i < 318u)
            fine = flags == 0u && extra == distExtra[i - 288u] && 1u + (m << extra) == distBase[i - 288u];
        else
            fine = flags == pdinf::kKindNone;
        if (!fine)
            return 1 + static_cast<int>(i);
    }
    return 0;
}

// The device's gunzip (popdel_amd/csrc/hip/inflate_core.h, the source inflate.hip compiles for the GPU) run on the CPU, one
// member after the other: tests compare it with zlib bit for bit, including damaged and truncated members (-m "not gpu").
int pdh_inflate_members_core(const uint8_t * data, const uint64_t * offsets, uint32_t n, uint8_t * out, const uint64_t * outOffsets,
                             uint32_t * status, uint64_t * outLengths)
{
    if (!data || !offsets || !out || !outOffsets || !status || !outLengths)
        return 1;
    uint32_t crcTable[256];
    for (unsigned i = 0; i < 256u; ++i)
        crcTable[i] = pdinf::crcTableEntry(i);
    pdinf::FlatTables tab;
    for (uint32_t m = 0; m < n; ++m)
    {
        uint64_t got = 0, used = 0;
        status[m] = pdinf::inflateGzipMember(data + offsets[m], offsets[m + 1] - offsets[m], out + outOffsets[m],
                                             outOffsets[m + 1] - outOffsets[m], &got, &used, tab, crcTable);
        outLengths[m] = got;
    }
    return 0;
}

// The host's own way through the same members, as the loader takes it (profile_io.cpp: libdeflate where the system has it,
// else zlib; one member per call), on `threads` threads: the CPU figure beside pd_inflate_members(). which: 0 libdeflate
// (falls back to zlib if it is not there; *used says which ran: 0 / 1), 1 zlib. Returns seconds through *seconds.
int pdh_inflate_members_cpu(const uint8_t * data, const uint64_t * offsets, uint32_t n, uint8_t * out, const uint64_t * outOffsets,
                            int which, unsigned threads, double * seconds, int * used, uint32_t * failures)
{
    if (!data || !offsets || !out || !outOffsets || !seconds)
        return 1;
    struct Lib
    {
        void * (*alloc)() = nullptr;
        void (*release)(void *) = nullptr;
        int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    } lib;
    if (which == 0)
        if (void * h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL))
        {
            lib.alloc = reinterpret_cast<void * (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
            lib.release = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
            lib.gunzip = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *)>(
                dlsym(h, "libdeflate_gzip_decompress_ex"));
        }
    const bool fast = lib.alloc && lib.release && lib.gunzip;
    if (used)
        *used = fast ? 0 : 1;
    threads = std::max(1u, threads);
    std::vector<uint32_t> bad(threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            void * dec = fast ? lib.alloc() : nullptr;
            const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(n) * t / threads), hi = static_cast<uint32_t>(static_cast<uint64_t>(n) * (t + 1) / threads);
            for (uint32_t m = lo; m < hi; ++m)
            {
                const size_t inLen = offsets[m + 1] - offsets[m], room = outOffsets[m + 1] - outOffsets[m];
                if (fast)
                {
                    size_t usedIn = 0, got = 0;
                    if (lib.gunzip(dec, data + offsets[m], inLen, out + outOffsets[m], room, &usedIn, &got) != 0)
                        ++bad[t];
                }
                else
                {
                    z_stream zs;
                    std::memset(&zs, 0, sizeof(zs));
                    if (inflateInit2(&zs, 15 + 16) != Z_OK)
                    {
                        ++bad[t];
                        continue;
                    }
                    zs.next_in = const_cast<Bytef *>(data + offsets[m]);
                    zs.avail_in = static_cast<uInt>(inLen);
                    zs.next_out = out + outOffsets[m];
                    zs.avail_out = static_cast<uInt>(room);
                    if (inflate(&zs, Z_FINISH) != Z_STREAM_END)
                        ++bad[t];
                    inflateEnd(&zs);
                }
            }
            if (dec)
                lib.release(dec);
        });
    for (std::thread & th : pool)
        th.join();
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (failures)
    {
        *failures = 0;
        for (uint32_t b : bad)
            *failures += b;
    }
    return 0;
}

// Micro-benchmark and self-check of the loader's table filing (ChromProfile::addRun): `rounds` slot-lengths of synthetic
// records (30x coverage: one pair per 10 bp, insert 500 +- 50) of ONE read group are filed the way the driver files them --
// a run per round, slots switched in between -- with heavy or light end tables. Returns seconds through *seconds and a
// checksum of what the loader reads off the tables (load counter, cursors, slot sizes, last windows) through *checksum:
// light and heavy must give the same.
// `flavour` shapes the records: 0 the everyday stream; 1 bursts of pairs a base or two apart (the load reaches the cap and
// records are dropped); 2 one record in twenty steps back by up to 40 bp (blocks out of order); 3 wide inserts (sd 400: end
// records beyond the write slot's border, negative inner distances); 4 all of these and gaps of several slot lengths.
int pdh_bench_filing_ex(unsigned rounds, int light, unsigned maxLoad, unsigned seed, int flavour, double * seconds, uint64_t * checksum);
// (light == 2: runs filed from their summaries where the tables allow it; how many runs / records went that way since the
// process began)
static uint64_t g_bulkRuns = 0, g_bulkRecords = 0;
void pdh_bench_filing_bulk_counts(uint64_t * runs, uint64_t * records)
{
    if (runs)
        *runs = g_bulkRuns;
    if (records)
        *records = g_bulkRecords;
}
int pdh_bench_filing(unsigned rounds, int light, unsigned maxLoad, double * seconds, uint64_t * checksum)
{
    return pdh_bench_filing_ex(rounds, light, maxLoad, 4242u, 0, seconds, checksum);
}
int pdh_bench_filing_ex(unsigned rounds, int light, unsigned maxLoad, unsigned seed, int flavour, double * seconds, uint64_t * checksum)
{
    if (!seconds || !checksum || rounds == 0)
        return 1;
    const unsigned buffer = 200000;
    pdh::ChromProfile prof(1, std::vector<unsigned>(1, maxLoad), buffer);
    prof.currentPos = 1000000;
    prof.resetTo(1000000);
    if (light)
        prof.setLightEnds(true);
    std::mt19937 gen(seed), headGen(seed + 17u);       // (headGen: choices of the filing mode that must not change the records)
    // light == 4: as 2, and the summarised runs' start entries are not copied either (StartSlots::VirtualRun); they are read back --
    // here: from the rounds' records, which are kept -- when the tables are asked for them (at the end, and where a start slot is
    // recycled under end entries that refer to it)
    std::vector<std::vector<uint32_t>> kept;
    if (light == 4)
        prof.refill = [&kept](unsigned, const pdh::StartSlots::VirtualRun & run, pdh::StartEntry * dst) {
            const std::vector<uint32_t> & r = kept[run.tag];
            std::memcpy(dst, r.data() + 2 * static_cast<size_t>(run.skip), static_cast<size_t>(run.n) * sizeof(pdh::StartEntry));
        };
    std::normal_distribution<double> ins(0.0, flavour >= 3 ? 400.0 : 50.0);
    unsigned burst = 0;
    std::vector<uint32_t> rec;
    uint64_t sum = 1469598103934665603ull;
    auto mix = [&sum](uint64_t v) { sum = (sum ^ v) * 1099511628211ull; };
    double total = 0;
    unsigned pos = 1000000;
    const std::vector<unsigned> rgs(1, 0);
    for (unsigned r = 0; r < rounds; ++r)
    {
        rec.clear();
        const unsigned right = prof.starts[0].slots[prof.starts[0].writeSlot].right;
        while (pos < right)
        {
            unsigned at = pos;
            if ((flavour == 2 || flavour == 4) && gen() % 20 == 0 && at > 1000040)
                at -= gen() % 41;
            rec.push_back(at);
            rec.push_back(static_cast<uint32_t>(static_cast<int32_t>(std::lround(ins(gen)))));
            if ((flavour == 1 || flavour == 4) && burst == 0 && gen() % 400 == 0)
                burst = 50 + gen() % 400;
            if (burst)
            {
                --burst;
                pos += gen() % 3;
            }
            else
                pos += 1 + gen() % 19;
        }
        if (flavour == 4 && gen() % 7 == 0)
            pos += buffer * (1 + gen() % 3) + gen() % 5000;      // (a gap: the next round's records lie slots ahead)
        prof.reserveRun(0, rec.size() / 2);
        const auto t0 = std::chrono::steady_clock::now();
        if (light >= 2)
        {
            // the way the driver files a device-parsed sample since round 6: a head of records one by one (the end table's
            // counting cursor comes home to the write slot), the run up to the first pair that ends beyond the border from its
            // summary (ChromProfile::addBulk; light == 3: the summary is taken but the records are filed by addRun() all the
            // same -- the yardstick of the timing), the rest record by record again
            const size_t n = rec.size() / 2;
            const size_t head = std::min<size_t>(n, light == 2 || light == 4 ? headGen() % 200 : 64);
            if (light == 4)
                kept.push_back(rec);
            prof.addRun(0, rec.data(), head, 500 - 300);
            size_t done = head;
            if (done < n)
            {
                const pdh::EndSlots & en = prof.ends[0];
                const uint32_t borderBin = (en.slots[en.writeSlot].right + 29u) / 30u;
                const uint32_t room = buffer / 30u + 512u;
                static std::vector<uint8_t> counts;
                static std::vector<uint16_t> wide;
                counts.resize(2 * static_cast<size_t>(room));
                pdh::RunSummary sum;
                const auto s0 = std::chrono::steady_clock::now();
                pdh::summarizeRun(rec.data() + 2 * done, n - done, 500 - 300, borderBin, room, room, sum, counts.data(), counts.data() + room);
                wide.assign(counts.begin(), counts.end());
                total -= std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();     // (the device's work in the product)
                if (light == 3)
                    sum.n = 0;
                pdh::VirtualNote note;
                note.firstWindow = rec[0] / 30u * 30u;
                note.skip = static_cast<uint32_t>(done);
                note.tag = static_cast<uint32_t>(kept.empty() ? 0 : kept.size() - 1);
                if (sum.n >= 16 && prof.addBulk(0, light == 4 ? nullptr : rec.data() + 2 * done, sum, wide.data(), wide.data() + room, 500 - 300, note))
                {
                    done += sum.n;
                    ++g_bulkRuns;
                    g_bulkRecords += sum.n;
                }
            }
            prof.addRun(0, rec.data() + 2 * done, n - done, 500 - 300);
        }
        else
            prof.addRun(0, rec.data(), rec.size() / 2, 500 - 300);
        total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mix(prof.activeLoad[0]);
        mix(prof.starts[0].insertionCount);
        unsigned border = 0;
        prof.performPartialSwitches(rgs, &border);
        if (prof.ends[0].readSlot == prof.ends[0].writeSlot)
        {
            prof.ends[0].switchWrite();
            prof.ends[0].fixConsecutiveSwitch(prof.activeLoad[0]);
        }
        std::vector<uint8_t> run, sw;
        if (prof.canPlanSegment(30))
        {
            const pdh::WalkShape sh = prof.planSegment(30, run, sw);
            mix(sh.firstPos);
            mix(sh.nWindows);
            mix(sh.stepPos);
        }
        mix(prof.ends[0].readIdx);
        mix(prof.ends[0].dNext);
        mix(prof.ends[0].dNextOffset);
        mix(prof.ends[0].endPosSlot);
        for (int k = 0; k < 3; ++k)
            mix(prof.ends[0].slots[k].total);
    }
    prof.setLightEnds(false);               // (back to ordered tables: what the host's window loop would find)
    prof.materializeAllStarts();
    for (int k = 0; k < 3; ++k)
        for (const pdh::EndEntry & e : prof.ends[0].slots[k].entries)
        {
            mix(e.id);
            mix(e.lastWindow);
        }
    for (int k = 0; k < 3; ++k)
    {
        mix(prof.starts[0].slots[k].sorted ? 1u : 0u);
        for (const pdh::StartEntry & e : prof.starts[0].slots[k].entries)
        {
            mix(e.startPos);
            mix(static_cast<uint32_t>(e.deviation));
        }
    }
    *seconds = total;
    *checksum = sum;
    return 0;
}
}  // extern "C"

extern "C" {
// pdh::summarizeRun() for the tests: what the device's summarizeBlocksKernel (pd_inflater_run_summarized) must reproduce per block.
// `summary`: the ten words of pd_block_summary / pdh::RunSummary; `counts`: roomStarts + roomEnds bytes (records that begin / end per window).
int pdh_summarize_run(const uint32_t * posDev, uint64_t nRecords, int innerOffset, uint32_t borderBin, uint32_t roomStarts, uint32_t roomEnds,
                      uint32_t * summary, uint8_t * counts)
{
    if (!posDev || !summary || !counts || roomStarts == 0 || roomEnds == 0)
        return 1;
    static_assert(sizeof(pdh::RunSummary) == 10 * sizeof(uint32_t), "RunSummary is pd_run_summary");
    pdh::RunSummary s;
    pdh::summarizeRun(posDev, static_cast<size_t>(nRecords), innerOffset, borderBin, roomStarts, roomEnds, s, counts, counts + roomStarts);
    std::memcpy(summary, &s, sizeof(s));
    return 0;
}
}  // extern "C"

namespace {
// inflated bytes of one block, as the loader's reader sees them
class BlockSource : public pdh::ByteSource
{
public:
    BlockSource(const unsigned char * p, size_t n) : p_(p), left_(n) {}
    size_t read(void * dst, size_t n) override
    {
        const size_t k = std::min(n, left_);
        std::memcpy(dst, p_, k);
        p_ += k;
        left_ -= k;
        return k;
    }
    const unsigned char * peek(size_t & avail) override
    {
        avail = left_;
        return p_;
    }
    void consume(size_t n) override
    {
        p_ += n;
        left_ -= n;
    }

private:
    const unsigned char * p_;
    size_t left_;
};
}  // namespace

extern "C" {
// What the host's reader makes of inflated profile blocks of a one-read-group file (readWindowHead + readWindowBodyFlat,
// profile_io.cpp: the reference's readWindow + convertWindow) -- the yardstick for the device's parser (pd_inflater_run_parsed,
// tests/test_gpu_parse.py). Block b = data[offsets[b], offsets[b + 1]). Two calls: with the arrays null it fills counts[3 * b
// .. ] = {256-bp windows, 30-bp windows, records} and state[b] (0: read to the end; 1: the reader threw -- a window cut short;
// 2: a window without records, where the loader stops); with the arrays (sized by the counts of the blocks with state 0, in
// block order) it fills them: profile windows as {chrom, beginPos, splitFailed, firstWin, nWin} with indices relative to the
// block, call windows {chrom, beginPos, firstRec}, their record counts, records {pos, dev}.
int pdh_parse_blocks(const uint8_t * data, const uint64_t * offsets, uint32_t n, uint64_t * counts, uint32_t * state, uint32_t * profileWindows,
                     uint32_t * callWindows, uint32_t * callWindowRecords, uint32_t * records)
{
    if (!data || !offsets || !counts || !state)
        return 1;
    uint64_t atRaw = 0, atWin = 0, atRec = 0;
    for (uint32_t b = 0; b < n; ++b)
    {
        BlockSource src(data + offsets[b], static_cast<size_t>(offsets[b + 1] - offsets[b]));
        pdh::FlatWindows wins;
        pdh::FlatCounts cnt;
        pdh::FlatRecords recs;
        std::vector<uint32_t> raws;
        uint32_t st = 0;
        try
        {
            pdh::WindowHead head;
            while (pdh::readWindowHead(src, head))
            {
                const uint32_t firstWin = static_cast<uint32_t>(wins.size());
                if (!pdh::readWindowBodyFlat(src, head, 1, wins, cnt, recs, 256, 30))
                {
                    st = 2;
                    break;
                }
                const uint32_t row[5] = {static_cast<uint32_t>(head.chrom), static_cast<uint32_t>(head.beginPos), 0u, firstWin,
                                         static_cast<uint32_t>(wins.size()) - firstWin};
                raws.insert(raws.end(), row, row + 5);
            }
        }
        catch (const std::exception &)
        {
            st = 1;
        }
        state[b] = st;
        counts[3 * b] = raws.size() / 5;
        counts[3 * b + 1] = wins.size();
        counts[3 * b + 2] = recs.size();
        if (st != 0 || !profileWindows)
            continue;
        std::memcpy(profileWindows + 5 * atRaw, raws.data(), raws.size() * sizeof(uint32_t));
        for (size_t w = 0; w < wins.size(); ++w)
        {
            callWindows[3 * (atWin + w)] = static_cast<uint32_t>(wins[w].chrom);
            callWindows[3 * (atWin + w) + 1] = static_cast<uint32_t>(wins[w].beginPos);
            callWindows[3 * (atWin + w) + 2] = wins[w].firstRec;
            callWindowRecords[atWin + w] = cnt[w];
        }
        for (size_t r = 0; r < recs.size(); ++r)
        {
            records[2 * (atRec + r)] = recs[r].pos;
            records[2 * (atRec + r) + 1] = recs[r].dev;
        }
        atRaw += raws.size() / 5;
        atWin += wins.size();
        atRec += recs.size();
    }
    return 0;
}
}  // extern "C"
