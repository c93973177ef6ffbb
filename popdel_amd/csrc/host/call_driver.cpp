#include <unistd.h>

#include "call_driver.h"
#include "cpu_sampler.h"

#include <chrono>
#include <cstring>
#include <fstream>
#include <pthread.h>
#include <sched.h>
#include <iostream>
#include <exception>
#include <condition_variable>
#include <deque>
#include <memory>
#include <fcntl.h>
#include <malloc.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <thread>
#include <unordered_map>

#include "loader_chunks.h"
#include "parallel.h"
#include "popdel_hip.h"
#include "profile_io.h"

namespace pdh {
namespace {

constexpr unsigned kNoPos = std::numeric_limits<unsigned>::max();

// Wall-clock split of the host phases, printed at exit when POPDEL_TIMING is set.
struct PhaseClock
{
    unsigned orderedBatches = 0, parallelBatches = 0;
    double params = 0, backend = 0, switches = 0, decode = 0, replay = 0, walk = 0, flatten = 0, submit = 0, collect = 0, merge = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};

// POPDEL_TIMING=2: a time line of the run (seconds since the program's first instruction in main), printed at exit.
struct EventTrace
{
    bool on = false;
    double t0 = 0, epoch0 = 0;
    std::mutex m;
    std::vector<std::pair<double, std::string>> events;
    void mark(const char * what)
    {
        if (!on)
            return;
        std::lock_guard<std::mutex> g(m);
        events.emplace_back(PhaseClock::now() - t0, what);
    }
    void mark(const std::string & what)
    {
        if (on)
            mark(what.c_str());
    }
    void print()
    {
        std::lock_guard<std::mutex> g(m);
        for (const auto & e : events)
            std::cerr << "[popdel timeline] " << e.first << " " << e.second << "\n";
        events.clear();
    }
};
EventTrace gTrace;

struct Coord
{
    std::string contig;
    unsigned pos = 0;
};

struct WindowCursor
{
    Coord current;
    Coord next;
    Coord nextChrom{"", kNoPos};
};

bool allTrue(const std::vector<char> & v)
{
    for (char b : v)
        if (!b)
            return false;
    return true;
}

bool advanceRoi(CallParams & p)
{
    ++p.nextRoi;
    return p.nextRoi != p.allRois.size();
}

unsigned contigRank(const std::string & name, const CallParams & p)
{
    unsigned i = 0;
    for (size_t r = p.nextRoi; r < p.allRois.size(); ++r, ++i)
        if (name == p.allRois[r].seqName)
            return i;
    return kNoPos;
}

// true if `a` does not come after `b` in ROI order.
bool lowerCoord(const Coord & a, const Coord & b, const CallParams & p)
{
    unsigned ra = contigRank(a.contig, p), rb = contigRank(b.contig, p);
    if (ra != rb)
        return ra < rb;
    return a.pos <= b.pos;
}

// Align the ROI with the next candidate window or vice versa (load_profile_popdel_call.h:220-234).
bool adaptRegions(Coord & cand, Region & roi, unsigned w)
{
    if (cand.contig != roi.seqName)
        return false;
    if (static_cast<int>(cand.pos) >= roi.endPos)
        return false;
    if (static_cast<int>(cand.pos + w - 1) == roi.beginPos)
        return true;
    if (static_cast<int>(cand.pos + w - 1) > roi.beginPos)
        roi.beginPos = static_cast<int32_t>(cand.pos);
    else
        cand.pos += ((roi.beginPos - cand.pos) / w) * w;
    return true;
}

// What a decoder thread produced for one sample and one round: the 256-bp windows from the index jump up to (and
// including) the window at which the loader will stop, each already split into 30-bp windows. Decoding (inflate, parse,
// split) touches no shared state; inserting the records into the window state machine is replayed sequentially.
struct DecodedRaw                          // (the device writes the same 20 bytes: ParsedWindow, inflate.hip)
{
    int32_t  chrom = 0;
    int32_t  beginPos = 0;
    uint32_t splitFailed = 0;
    uint32_t firstWin = 0, nWin = 0;       // its 30-bp windows in DecodedStream::wins
};
// The device's lean form of a 256-bp window (round 6; LeanWindow, inflate.hip): where its records lie in the sample's stream and the
// positions of the first and the last. Its 30-bp windows are the runs of equal position / 30 among its records -- read off the
// records where the loader's decisions need them, at the two ends of what a round files; everywhere else the two positions do.
struct LeanRaw
{
    int32_t  chrom = 0, beginPos = 0;
    uint32_t firstRec = 0, nRecs = 0, firstPos = 0, lastPos = 0;
};
static_assert(sizeof(DecodedRaw) == 20 && sizeof(FlatWindow) == 12 && sizeof(PairRecord) == 8 && sizeof(LeanRaw) == 24, "layouts shared with the device's parser");

// Records [first, first + count) of a sample's stream (count >= 1) where the host has them, else nullptr. `recs`: the sample's
// records when all were copied back (recIndex == nullptr); else the records are those of the blocks recIndex does not mark ~0,
// packed (pd_summarize_options::keep_below / keep_from): block b = records [blockRecs[b], blockRecs[b + 1]) of the chunk, kept
// blocks that follow each other lie one behind the other at packed + recIndex[b].
inline const PairRecord * recordsOf(const PairRecord * recs, const PairRecord * packed, const uint64_t * recIndex, const uint64_t * blockRecs,
                                    size_t nBlocks, uint32_t first, uint32_t count)
{
    if (!recIndex)
        return recs + first;
    const uint64_t a = blockRecs[0] + first, b = a + count - 1;
    size_t j = static_cast<size_t>(std::upper_bound(blockRecs, blockRecs + nBlocks + 1, a) - blockRecs);
    if (j == 0 || j > nBlocks)
        return nullptr;
    --j;                                    // (the block that holds record a)
    if (recIndex[j] == ~static_cast<uint64_t>(0))
        return nullptr;
    const PairRecord * p = packed + recIndex[j] + (a - blockRecs[j]);
    for (; blockRecs[j + 1] <= b; ++j)      // (... and every block up to the one that holds record b)
    {
        if (j + 1 >= nBlocks)
            return nullptr;
        if (blockRecs[j + 2] != blockRecs[j + 1] && recIndex[j + 1] == ~static_cast<uint64_t>(0))
            return nullptr;
    }
    return p;
}

struct alignas(64) DecodedStream      // (neighbours in the pool are filled by different threads: no shared cache lines)
{
    std::vector<DecodedRaw> raws;
    FlatWindows wins;                      // all 30-bp windows, flat
    FlatCounts counts;                     // per window: records per read group of the file
    FlatRecords recs;
    bool eof = false;
    bool sawFarWindow = false;     // a window beyond the next slot: the replay may consult ALL tables (allEmpty/resetTo)
    // What the replay reads: the vectors above (the host parsed the sample) or arrays the device's parser left in an
    // inflater's page-locked buffers (valid until the batch has been replayed).
    const DecodedRaw * rawsV = nullptr;
    size_t nRawsV = 0;
    const FlatWindow * winsV = nullptr;
    const uint32_t * countsV = nullptr;
    const PairRecord * recsV = nullptr;
    size_t nRecsV = 0;
    // (device-parsed, lean form: these instead of rawsV / winsV / countsV)
    const LeanRaw * leanV = nullptr;
    size_t nLeanV = 0;
    // (device-parsed arrays: the sample's blocks summarised for the tables -- block b holds records [blockRecs[b] - blockRecs[0],
    // blockRecs[b + 1] - blockRecs[0]) of recsV; pd_inflater_run_summarized)
    const uint64_t * blockRecs = nullptr;
    size_t nBlocks = 0;
    const pd_block_summary * blockSums = nullptr;
    const uint8_t * blockCounts = nullptr;
    uint32_t startsRoom = 0, endsRoom = 0;
    pd_inflater * inflater = nullptr;          // the inflater whose device buffers hold the same records, recsV[0] at index recBase
    uint64_t recBase = 0, runId = 0;
    // (records copied back for the blocks at the two ends only: recsV is null, see recordsOf())
    const uint64_t * recIndex = nullptr;
    const PairRecord * recsPacked = nullptr;
    const PairRecord * records(uint32_t first, uint32_t count) const { return recordsOf(recsV, recsPacked, recIndex, blockRecs, nBlocks, first, count); }
    void viewOwn()
    {
        rawsV = raws.data();
        nRawsV = raws.size();
        winsV = wins.data();
        countsV = counts.data();
        recsV = recs.data();
        nRecsV = recs.size();
    }
    // Streams are pooled per batch position and reused round after round: their vectors keep their capacity, so the
    // steady state allocates nothing (hundreds of kilobytes per sample and round otherwise -- mmap/munmap from 64 threads).
    void reuse()
    {
        raws.clear();
        wins.clear();
        counts.clear();
        recs.clear();
        eof = false;
        sawFarWindow = false;
        rawsV = nullptr;
        nRawsV = 0;
        winsV = nullptr;
        countsV = nullptr;
        recsV = nullptr;
        nRecsV = 0;
        leanV = nullptr;
        nLeanV = 0;
        blockRecs = nullptr;
        nBlocks = 0;
        blockSums = nullptr;
        blockCounts = nullptr;
        startsRoom = endsRoom = 0;
        inflater = nullptr;
        recBase = 0;
        recIndex = nullptr;
        recsPacked = nullptr;
    }
};

// The sequential half of readSegment() as a loop that can be left and entered again: it runs until the stream is through or the
// loader stops (finished, code), or -- a file with one read group -- until it has found the next RUN of records that are filed
// one after the other with nothing in between (haveRun: records [runFirst, runFirst + runCount) of the stream), which the caller
// files before it lets the loop go on (CallRun::fileRun: from the blocks' summaries where the device took them).
struct ReplayState
{
    size_t   rawIdx = 0;
    uint32_t wi = 0;
    bool     inRaw = false, onRoiContig = false;
    bool     finished = false;
    unsigned code = 0;
    bool     haveRun = false;
    uint32_t runFirst = 0, runCount = 0;
    uint32_t runFirstWindow = 0;        // the 30-bp window (position) of the run's first record
};

// The loader's blocks on the device (round 4). Which samples read in a round, from where and up to where is known when the
// round starts (the readers are the samples whose slots are not switched; they read from the index block of the region's begin
// to the window behind their write slot's border), and the index brackets every gzip member of that stretch. A thread of its
// own takes the readers in chunks (256, then 1,024 samples): their members' bytes go from the files into an inflater's
// page-locked buffer (pd_inflater_input); a thread per inflater (three) has the device inflate them one lane per member and
// parse the blocks (pd_inflater_run_parsed, inflate.hip) and publishes, per sample, the arrays the replay reads -- or, for a
// sample with a block the device's parser left alone, the inflated bytes, which the decoder threads parse from memory with
// the file behind them if the stretch turns out too short (ProfileFile::openPrefetched). An inflater's buffers go round when
// the chunk that used them last has been replayed. A sample whose index, members or inflated sizes look wrong in any way is
// left to the host's own path (zlib's behaviour on damaged files is the specification there). CallRun keeps two of these:
// one serves the running round, the other works ahead on a guess of the next (startGunzipPrefetch(), adopt()).
// Loaders of `popdel call` (GunzipPrefetch): one runs the round that is being decoded, the others the rounds to come -- up to two of
// them (CallRun::aheadDepth_): a loader takes a round's length and more from its first gather to its last chunk (5 + 15-18 ms at
// 4,000 samples against 16 ms of decoding and replaying), so the loader of round N + 2 starts while round N runs. A loader's
// records on the device are what the segment of its round points into (pd_walk_source): a set is free again when that segment
// has been handed over -- hence three sets for two rounds ahead.
constexpr size_t kLoaderSets = 3;

class GunzipPrefetch
{
public:
    struct Item
    {
        unsigned sample = 0;
        int32_t beginPos = 0, endPos = 0;
        bool parse = false;             // one read group: the device parses the blocks as well
        uint32_t firstRegion = 0;       // beginPos / the file's index region size: what the index jump depends on
        int32_t innerOffset = 0;        // parse: median - 2 * read length of that read group (a pair ends at position + max(0, deviation + this))
        uint32_t keepBelow = 0xffffffffu, keepFrom = 0;     // records come back for blocks with a record below / at or beyond (edgesOnly)
    };
    struct Ready
    {
        const unsigned char * data = nullptr;       // the inflated bytes (bytesValid) ...
        size_t size = 0;
        uint64_t resume = 0;
        bool usable = false, bytesValid = false;
        bool toFileEnd = false;                     // the blocks are all that is left of the file
        // ... and / or the blocks parsed (the sample's 256-bp windows, 30-bp windows with their record counts, records)
        bool parsed = false;
        const DecodedRaw * raws = nullptr;
        const LeanRaw * lean = nullptr;             // (lean form: instead of raws / wins / counts; nRaws of them)
        size_t nRaws = 0;
        const FlatWindow * wins = nullptr;
        const uint32_t * counts = nullptr;
        const PairRecord * recs = nullptr;
        size_t nRecs = 0;
        // ... and the blocks summarised for the tables (block b of the sample: records [blockRecs[b], blockRecs[b + 1]) of the
        // chunk's records array, recs[0] being record blockRecs[0])
        const uint64_t * blockRecs = nullptr;
        size_t nBlocks = 0;
        const pd_block_summary * blockSums = nullptr;
        const uint8_t * blockCounts = nullptr;
        uint32_t startsRoom = 0, endsRoom = 0;
        pd_inflater * inflater = nullptr;           // whose device buffers hold the records as well (recs[0] = record recBase there)
        uint64_t recBase = 0, runId = 0;
        const uint64_t * recIndex = nullptr;        // records copied back for some blocks only (recs is null then): recordsOf()
        const PairRecord * recsPacked = nullptr;
        const PairRecord * records(uint32_t first, uint32_t count) const { return recordsOf(recs, recsPacked, recIndex, blockRecs, nBlocks, first, count); }
        // (adopt(): the reader's jump lands `skip` index regions behind the guess -> its stream begins at a later block)
        const uint64_t * rawsByBlock = nullptr;      // first 256-bp window of each of the sample's blocks (and one past)
        const uint64_t * bytesByBlock = nullptr;     // first inflated byte of each
        const std::vector<uint64_t> * cuts = nullptr;        // the blocks' file offsets
        const std::vector<uint64_t> * regionOffsets = nullptr;
    };
    // items per chunk and the batch limit of the replay: loader_chunks.h
    static constexpr size_t kInflaters = kLoaderInflaters;
    static size_t chunkFor(size_t readers) { return loaderChunkFor(readers); }
    static size_t firstChunkFor(size_t readers) { return loaderFirstChunkFor(readers); }
    size_t firstChunk_ = 256, chunk_ = 1024;            // of the round the loader is on (start())
    size_t chunkOf(size_t item) const { return item < firstChunk_ ? 0 : 1 + (item - firstChunk_) / chunk_; }
    size_t chunkBegin(size_t c) const { return c == 0 ? 0 : firstChunk_ + (c - 1) * chunk_; }
    static size_t batchLimitFor(size_t samples) { return loaderBatchLimitFor(samples); }

    // Descriptors of the profiles, kept open from round to round as far as the process's limit allows (every sample's file is
    // asked twice per round: its index, then its bytes); a sample beyond the limit is opened and closed per use as before.
    // One thread works on a sample at a time within a pass (the passes are parallel over samples).
    int descriptorOf(unsigned sample)
    {
        if (sample < fds_.size() && fds_[sample] >= 0)
            return fds_[sample];
        const int fd = ::open(p_.inputFiles[sample].c_str(), O_RDONLY | O_CLOEXEC);
        if (fd >= 0 && sample < fds_.size())
            fds_[sample] = fd;
        return fd;
    }
    void giveBack(unsigned sample, int fd)
    {
        if (fd >= 0 && !(sample < fds_.size() && fds_[sample] == fd))
            ::close(fd);
    }

    // `devices`: the GPUs of the run (POPDEL_DEVICES); the loader's inflaters are dealt over them, beginning at devices[first],
    // so that the gunzip + parse of a round is not one GPU's work while the others wait for their samples
    GunzipPrefetch(const CallParams & p, const std::vector<int> & devices, size_t first) : p_(p)
    {
        {
            // (each of the loaders keeps its own descriptors: a share of the limit each, one share is the process's own)
            struct rlimit lim;
            size_t keep = 256;
            if (getrlimit(RLIMIT_NOFILE, &lim) == 0)           // (raised as far as the cohort needs when the run began: runPopdelCall)
                keep = lim.rlim_cur == RLIM_INFINITY ? static_cast<size_t>(1) << 20
                                                     : (static_cast<size_t>(lim.rlim_cur) > 1024 ? (static_cast<size_t>(lim.rlim_cur) - 1024) / (kLoaderSets + 1) : 0);
            fds_.assign(std::min<size_t>(p.inputFiles.size(), keep), -1);
        }
        for (size_t k = 0; k < kInflaters; ++k)
            inflater_[k] = pd_inflater_create(devices[(first + k) % devices.size()]);
        ok_ = true;
        for (pd_inflater * f : inflater_)
            ok_ = ok_ && f;
        parseOnDevice_ = !getenv("POPDEL_HOST_PARSE");
    }
    ~GunzipPrefetch()
    {
        finish();
        for (int fd : fds_)
            if (fd >= 0)
                ::close(fd);
        for (pd_inflater * f : inflater_)
            pd_inflater_destroy(f);
    }
    bool ok() const { return ok_; }

    // the round's readers, in sample order; `contig`: the region's contig (the index jump's)
    void start(std::vector<Item> items, const std::string & contig)
    {
        finish();
        items_ = std::move(items);
        firstChunk_ = firstChunkFor(items_.size());
        chunk_ = chunkFor(items_.size());
        contig_ = contig;
        ready_.assign(items_.size(), Ready());
        valid_.assign(items_.size(), 1);
        skip_.assign(items_.size(), 0);
        kept_.clear();
        slotOf_.clear();
        for (size_t k = 0; k < items_.size(); ++k)
            slotOf_[items_[k].sample] = k;
        done_.assign(items_.empty() ? 0 : chunkOf(items_.size() - 1) + 1, 0);
        // (a round of more chunks than inflaters runs an inflater twice: its records of the first time are gone when the round's
        // segment is handed over -- such a round copies every record back and the window loop reads the host's tables)
        roundFitsInflaters_ = done_.size() <= kInflaters;
        gathered_ = false;
        releasedBelow_ = 0;
        stop_ = false;
        failed_ = false;
        if (!items_.empty())
            worker_ = std::thread([this] { run(); });
    }
    // A round started AHEAD of time (on a guess of who will read from where) meets the round as it is: a sample's blocks are
    // good if the sample reads and its index jump lands where the guess said or up to two index regions further on (same contig;
    // the bytes from the block at which the jump lands are what its reader would inflate, however far the guess went -- wait()
    // finds that block by its file offset). Returns the number of `actual` covered.
    size_t adopt(const std::vector<Item> & actual, const std::string & contig)
    {
        std::fill(valid_.begin(), valid_.end(), 0);
        std::fill(skip_.begin(), skip_.end(), 0);
        if (contig != contig_)
            return 0;
        size_t covered = 0;
        for (const Item & a : actual)
        {
            auto it = slotOf_.find(a.sample);
            if (it == slotOf_.end() || a.firstRegion < items_[it->second].firstRegion || a.firstRegion - items_[it->second].firstRegion > 2)
                continue;
            valid_[it->second] = 1;
            skip_[it->second] = static_cast<unsigned char>(a.firstRegion - items_[it->second].firstRegion);
            ++covered;
        }
        return covered;
    }
    // every chunk of the round is through the device (or given up)
    bool allDone()
    {
        std::lock_guard<std::mutex> lock(m_);
        for (char d : done_)
            if (!d)
                return false;
        return true;
    }
    // ... or at least gathered and handed to the threads that drive the device (the gathering thread is through)
    bool allGathered()
    {
        std::lock_guard<std::mutex> lock(m_);
        return gathered_ || failed_ || items_.empty();
    }
    // chunks of the round that are not through the device yet
    size_t outstanding()
    {
        std::lock_guard<std::mutex> lock(m_);
        size_t n = 0;
        for (char d : done_)
            n += d ? 0 : 1;
        return n;
    }
    // the sample's inflated bytes (waits for its chunk), or nullptr: inflate it yourself
    const Ready * wait(unsigned sample)
    {
        auto it = slotOf_.find(sample);
        if (it == slotOf_.end() || !valid_[it->second])
            return nullptr;
        const size_t chunk = chunkOf(it->second);
        const double t0 = PhaseClock::now();
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return done_[chunk] || failed_; });
        waitSeconds += PhaseClock::now() - t0;
        Ready & r = ready_[it->second];
        if (failed_ || !r.usable)
            return nullptr;
        if (skip_[it->second])
        {
            // the stream begins at the block whose file offset the index holds for the reader's region
            const unsigned d = skip_[it->second];
            skip_[it->second] = 0;
            r.usable = false;
            if (!r.cuts || !r.regionOffsets || d >= r.regionOffsets->size())
                return nullptr;
            const uint64_t at = (*r.regionOffsets)[d];
            const auto pos = std::lower_bound(r.cuts->begin(), r.cuts->end(), at);
            if (pos == r.cuts->end() || *pos != at || pos + 1 == r.cuts->end())
                return nullptr;             // (nothing of the stretch lies behind the jump: the reader goes to the file itself)
            const size_t b = static_cast<size_t>(pos - r.cuts->begin());
            const uint64_t bytes = r.bytesByBlock[b] - r.bytesByBlock[0];
            if (r.data)
                r.data += bytes;
            r.size -= static_cast<size_t>(bytes);
            if (r.parsed)
            {
                const uint64_t n = r.rawsByBlock[b] - r.rawsByBlock[0];
                if (r.lean)
                    r.lean += n;
                else
                    r.raws += n;
                r.nRaws -= static_cast<size_t>(n);
            }
            r.usable = true;
        }
        return &r;
    }
    // every sample below `sample` has been decoded: their chunks' buffers may be used again
    void releaseBelow(unsigned sample)
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            releasedBelow_ = std::max(releasedBelow_, sample);
        }
        cv_.notify_all();
    }
    void finish()
    {
        if (worker_.joinable())
        {
            {
                std::lock_guard<std::mutex> lock(m_);
                stop_ = true;
                releasedBelow_ = 0xffffffffu;
            }
            cv_.notify_all();
            worker_.join();
        }
    }
    double kernelMs = 0, gatherSeconds = 0, runSeconds = 0, waitSeconds = 0;      // (waitSeconds: summed over the decoder threads)
    uint64_t members = 0, compressedBytes = 0, inflatedBytes = 0, samplesServed = 0, samplesLeft = 0, samplesParsed = 0;
    bool parseOnDevice_ = true;         // POPDEL_HOST_PARSE=1: the device inflates, the host parses
    // the parsed blocks summarised for the tables as well (CallRun: runs filed from summaries), with this much room per block
    std::function<void()> onProgress;   // called (from a loader thread) when every chunk of the round has been gathered, and when a chunk is through the device
    bool summariesWanted = false;
    uint32_t startsRoom = 0, endsRoom = 0;
    bool leanWanted = false;            // ... and the 30-bp windows not written out (LeanRaw; POPDEL_FULL_PARSE=1 keeps them)
    bool edgesOnly = false;             // ... and records copied back for the blocks at the two ends of a sample's stretch only (Item::keepBelow / keepFrom)
    std::vector<int> fds_;              // descriptorOf()

private:
    // what the gathering thread hands to the one that drives the device
    struct Prepared
    {
        size_t lo = 0, hi = 0;
        std::vector<std::vector<uint64_t>> cuts, regionOffsets;
        std::vector<uint64_t> fileSize;
        std::vector<char> good;
        std::vector<uint64_t> inOff, outOff;
        std::vector<uint32_t> firstMember, sampleFirst;
        std::vector<uint8_t> parseMember;
        std::vector<int32_t> innerOffset;       // per member: median - 2 * read length of its (one) read group (block summaries)
        std::vector<uint32_t> keepBelow, keepFrom;
        uint64_t bytes = 0;
        double seconds = 0;
    };

    // A thread per inflater and round: it gathers its chunks (index offsets, the members' bytes into the inflater's page-locked
    // buffer), runs them on the device and publishes them -- the three gather at the same time (round 6; chunk after chunk from one
    // thread until then: 3.7 + 3.7 + 5.3 ms of a 25-ms round in front of the last chunk's copies), the copies of one chunk
    // overlap the kernels of another. The buffers of an inflater go round when the chunk before last has been decoded.
    void run()
    {
        const size_t nChunks = items_.empty() ? 0 : chunkOf(items_.size() - 1) + 1;
        size_t gatheredChunks = 0;
        auto deviceLoop = [&](size_t which) {
            for (size_t c = which; c < nChunks; c += kInflaters)
            {
                {
                    std::lock_guard<std::mutex> lock(m_);
                    if (stop_ || failed_)
                        break;
                }
                std::unique_ptr<Prepared> job(new Prepared());
                job->lo = chunkBegin(c);
                job->hi = std::min(items_.size(), chunkBegin(c + 1));
                bool prepared = false;
                try
                {
                    if (gTrace.on)
                        gTrace.mark("loader " + std::to_string(reinterpret_cast<uintptr_t>(this) % 997) + ": gathering chunk " + std::to_string(c));
                    prepare(c, *job);
                    if (gTrace.on)
                        gTrace.mark("loader " + std::to_string(reinterpret_cast<uintptr_t>(this) % 997) + ": chunk " + std::to_string(c) + " gathered");
                    prepared = true;
                }
                catch (...)
                {
                    std::lock_guard<std::mutex> lock(m_);
                    failed_ = true;
                    ok_ = false;
                }
                bool all = false;
                {
                    std::lock_guard<std::mutex> lock(m_);
                    if (++gatheredChunks == nChunks || failed_)
                        all = gathered_ = true;
                }
                cv_.notify_all();
                if (all && onProgress)
                    onProgress();
                if (!prepared)
                    break;
                {
                    std::unique_lock<std::mutex> lock(m_);
                    if (c >= kInflaters)
                    {
                        // the inflater's output still holds the chunk it ran last: wait until that chunk's last sample has been decoded
                        // (a batch of the decoder is at most 512 samples wide: the one that holds this sample needs nothing beyond the chunk after it)
                        const unsigned lastOfOld = items_[chunkBegin(c - kInflaters + 1) - 1].sample;
                        cv_.wait(lock, [&] { return releasedBelow_ > lastOfOld || stop_; });
                    }
                    if (stop_ || failed_)
                        break;
                }
                try
                {
                    if (gTrace.on)
                        gTrace.mark("loader " + std::to_string(reinterpret_cast<uintptr_t>(this) % 997) + ": chunk " + std::to_string(c) + " goes to the device");
                    execute(c, *job);
                    if (gTrace.on)
                        gTrace.mark("loader " + std::to_string(reinterpret_cast<uintptr_t>(this) % 997) + ": chunk " + std::to_string(c) + " is there");
                    std::lock_guard<std::mutex> lock(m_);
                    kept_.push_back(std::move(job));
                }
                catch (...)
                {
                    // (whatever it was: the remaining samples, and the rounds to come, are inflated on the host)
                    std::lock_guard<std::mutex> lock(m_);
                    failed_ = true;
                    ok_ = false;
                }
                {
                    std::lock_guard<std::mutex> lock(m_);
                    done_[c] = 1;
                }
                cv_.notify_all();
                if (onProgress)
                    onProgress();               // (a chunk is through: the round to come may be waiting for just that)
            }
            {
                // (whatever this thread leaves undone counts as done: the decoders then inflate those samples themselves)
                std::lock_guard<std::mutex> lock(m_);
                for (size_t c = which; c < nChunks; c += kInflaters)
                    done_[c] = 1;
                if (stop_ || failed_)
                    gathered_ = true;
            }
            cv_.notify_all();
        };
        std::vector<std::thread> device;
        for (size_t k = 1; k < kInflaters; ++k)
            device.emplace_back(deviceLoop, k);
        deviceLoop(0);
        for (std::thread & t : device)
            t.join();
    }
    void prepare(size_t c, Prepared & j)
    {
        pd_inflater * inf = inflater_[c % kInflaters];
        const size_t lo = j.lo, n = j.hi - j.lo;
        const double t0 = PhaseClock::now();
        // 1. the members' file offsets from the index
        std::vector<std::vector<uint64_t>> & cuts = j.cuts;
        cuts.assign(n, std::vector<uint64_t>());
        j.regionOffsets.assign(n, std::vector<uint64_t>());
        j.fileSize.assign(n, 0);
        parallelFor(n, [&](size_t k) {
            const Item & it = items_[lo + k];
            const unsigned contigIdx = p_.representativeContigs ? 0 : it.sample;
            try
            {
                const int fd = descriptorOf(it.sample);
                if (fd < 0 ||
                    !ProfileFile::blockOffsetsOf(fd, p_.contigNames[contigIdx], p_.contigLengths[contigIdx], p_.indexRegionSizes[contigIdx], contig_,
                                                 it.beginPos, it.endPos, cuts[k], &j.regionOffsets[k], &j.fileSize[k]) ||
                    cuts[k].size() < 2 || cuts[k].back() - cuts[k].front() > (static_cast<uint64_t>(256) << 20))
                    cuts[k].clear();
                giveBack(it.sample, fd);
            }
            catch (...)
            {
                cuts[k].clear();
            }
        }, 16, 8);
        std::vector<uint64_t> base(n + 1, 0);
        for (size_t k = 0; k < n; ++k)
            base[k + 1] = base[k] + (cuts[k].empty() ? 0 : cuts[k].back() - cuts[k].front());
        j.good.assign(n, 0);
        j.outOff.assign(1, 0);
        j.firstMember.assign(n + 1, 0);
        j.bytes = base[n];
        if (base[n] == 0)
            return;
        unsigned char * buf = pd_inflater_input(inf, base[n]);
        if (!buf)
            throw ProfileError("pd_inflater_input failed");
        // 2. their bytes
        std::vector<char> & good = j.good;
        parallelFor(n, [&](size_t k) {
            if (cuts[k].empty())
                return;
            const Item & it = items_[lo + k];
            const int fd = descriptorOf(it.sample);
            if (fd < 0)
                return;
            const size_t want = static_cast<size_t>(base[k + 1] - base[k]);
            size_t got = 0;
            while (got < want)
            {
                const ssize_t r = pread(fd, buf + base[k] + got, want - got, static_cast<off_t>(cuts[k].front() + got));
                if (r <= 0)
                    break;
                got += static_cast<size_t>(r);
            }
            giveBack(it.sample, fd);
            good[k] = got == want ? 1 : 0;
        }, 8, 8);
        // 3. member table: every member must begin like a gzip member with deflate data and end with a plausible size
        std::vector<uint64_t> &inOff = j.inOff, &outOff = j.outOff;
        std::vector<uint32_t> & firstMember = j.firstMember;
        uint64_t lastEnd = 0;
        for (size_t k = 0; k < n; ++k)
        {
            firstMember[k] = static_cast<uint32_t>(inOff.size());
            if (!good[k])
                continue;
            const std::vector<uint64_t> & cut = cuts[k];
            bool fine = true;
            for (size_t m = 0; m + 1 < cut.size() && fine; ++m)
            {
                const uint64_t a = base[k] + (cut[m] - cut.front()), b = base[k] + (cut[m + 1] - cut.front());
                uint32_t isize = 0;
                fine = b - a >= 18 && buf[a] == 0x1f && buf[a + 1] == 0x8b && buf[a + 2] == 8;
                if (fine)
                {
                    std::memcpy(&isize, buf + b - 4, 4);
                    fine = isize <= (64u << 20);
                }
                if (fine)
                {
                    inOff.push_back(a);
                    outOff.push_back(outOff.back() + isize);
                    j.sampleFirst.push_back(firstMember[k]);
                    j.parseMember.push_back(items_[lo + k].parse && parseOnDevice_ ? 1 : 0);
                    j.innerOffset.push_back(items_[lo + k].innerOffset);
                    j.keepBelow.push_back(items_[lo + k].keepBelow);
                    j.keepFrom.push_back(items_[lo + k].keepFrom);
                }
            }
            if (!fine)
            {
                // (not what the index promised: drop the sample's members again)
                outOff.resize(firstMember[k] + 1);
                inOff.resize(firstMember[k]);
                j.sampleFirst.resize(firstMember[k]);
                j.parseMember.resize(firstMember[k]);
                j.innerOffset.resize(firstMember[k]);
                j.keepBelow.resize(firstMember[k]);
                j.keepFrom.resize(firstMember[k]);
                good[k] = 0;
            }
            else
                lastEnd = base[k + 1];
        }
        firstMember[n] = static_cast<uint32_t>(inOff.size());
        // pd_inflater_run() takes one offset array (member m = [off[m], off[m + 1])); the bytes of a dropped sample between two
        // kept ones simply trail the member before them -- a gzip member ends where its trailer ends, what follows is not read
        if (!inOff.empty())
            inOff.push_back(lastEnd);
        j.seconds = PhaseClock::now() - t0;
    }

    void execute(size_t c, Prepared & j)
    {
        const size_t lo = j.lo, n = j.hi - j.lo;
        if (j.inOff.empty())
        {
            std::lock_guard<std::mutex> lock(m_);
            samplesLeft += n;
            return;
        }
        pd_inflater * inf = inflater_[c % kInflaters];
        const double t1 = PhaseClock::now();
        const uint8_t * out = nullptr;
        const uint32_t * status = nullptr;
        const uint64_t * lengths = nullptr;
        double ms = 0;
        pd_parsed_blocks pb;
        std::memset(&pb, 0, sizeof(pb));
        pd_block_summaries sums;
        std::memset(&sums, 0, sizeof(sums));
        const bool haveSums = summariesWanted && parseOnDevice_;
        const bool lean = haveSums && leanWanted;
        const bool partial = lean && edgesOnly && roundFitsInflaters_;
        if (haveSums)
        {
            pd_summarize_options opt;
            std::memset(&opt, 0, sizeof(opt));
            opt.inner_offset = j.innerOffset.data();
            opt.starts_room = startsRoom;
            opt.ends_room = endsRoom;
            opt.lean = lean ? 1u : 0u;
            opt.keep_below = partial ? j.keepBelow.data() : nullptr;
            opt.keep_from = partial ? j.keepFrom.data() : nullptr;
            if (pd_inflater_run_summarized(inf, j.inOff.data(), static_cast<uint32_t>(j.inOff.size() - 1), j.outOff.data(), j.sampleFirst.data(),
                                           j.parseMember.data(), &opt, &out, &status, &lengths, &pb, &sums, &ms) != 0)
                throw ProfileError("pd_inflater_run_summarized failed");
        }
        else if (pd_inflater_run_parsed(inf, j.inOff.data(), static_cast<uint32_t>(j.inOff.size() - 1), j.outOff.data(), j.sampleFirst.data(),
                                        j.parseMember.data(), &out, &status, &lengths, &pb, &ms) != 0)
            throw ProfileError("pd_inflater_run_parsed failed");
        const uint64_t runId = pd_inflater_run_id(inf);
        const double t2 = PhaseClock::now();
        std::lock_guard<std::mutex> lock(m_);           // (the counters below; ready_[] of this chunk is read once done_ says so)
        for (size_t k = 0; k < n; ++k)
        {
            Ready & r = ready_[lo + k];
            if (!j.good[k])
            {
                ++samplesLeft;
                continue;
            }
            bool fine = true;
            for (uint32_t m = j.firstMember[k]; m < j.firstMember[k + 1] && fine; ++m)
                fine = status[m] == 0 && lengths[m] == j.outOff[m + 1] - j.outOff[m];
            if (!fine)
            {
                ++samplesLeft;
                continue;
            }
            r.bytesValid = pb.bytes_copied != 0 && out;
            r.data = r.bytesValid ? out + j.outOff[j.firstMember[k]] : nullptr;
            r.size = static_cast<size_t>(j.outOff[j.firstMember[k + 1]] - j.outOff[j.firstMember[k]]);
            r.resume = j.cuts[k].back();
            r.toFileEnd = r.resume == j.fileSize[k];
            r.cuts = &j.cuts[k];
            r.regionOffsets = &j.regionOffsets[k];
            r.bytesByBlock = j.outOff.data() + j.firstMember[k];
            r.rawsByBlock = pb.profile_window_offsets + j.firstMember[k];
            bool parsedAll = j.firstMember[k + 1] > j.firstMember[k];
            for (uint32_t m = j.firstMember[k]; m < j.firstMember[k + 1] && parsedAll; ++m)
                parsedAll = pb.block_flags[4 * static_cast<size_t>(m) + 3] == 0;
            if (parsedAll)
            {
                const uint32_t m0 = j.firstMember[k], m1 = j.firstMember[k + 1];
                r.parsed = true;
                r.nRaws = static_cast<size_t>(pb.profile_window_offsets[m1] - pb.profile_window_offsets[m0]);
                if (lean)
                    r.lean = static_cast<const LeanRaw *>(pb.profile_windows) + pb.profile_window_offsets[m0];
                else
                {
                    r.raws = static_cast<const DecodedRaw *>(pb.profile_windows) + pb.profile_window_offsets[m0];
                    r.wins = static_cast<const FlatWindow *>(pb.call_windows) + pb.call_window_offsets[m0];
                    r.counts = pb.call_window_records + pb.call_window_offsets[m0];
                }
                if (haveSums && sums.record_index)
                {
                    r.recs = nullptr;
                    r.recIndex = sums.record_index + m0;
                    r.recsPacked = static_cast<const PairRecord *>(pb.records);
                }
                else
                    r.recs = static_cast<const PairRecord *>(pb.records) + pb.record_offsets[m0];
                r.nRecs = static_cast<size_t>(pb.record_offsets[m1] - pb.record_offsets[m0]);
                // (the window loop may take the entries from these records only while the inflater's buffers hold this chunk: a round
                // of more chunks than inflaters re-uses them before its segment is handed over)
                r.inflater = roundFitsInflaters_ ? inf : nullptr;
                r.recBase = pb.record_offsets[m0];
                r.runId = runId;
                if (haveSums)
                {
                    r.blockRecs = pb.record_offsets + m0;
                    r.nBlocks = m1 - m0;
                    r.blockSums = sums.block_summaries + m0;
                    r.blockCounts = sums.block_counts + static_cast<size_t>(m0) * (sums.starts_room + sums.ends_room);
                    r.startsRoom = sums.starts_room;
                    r.endsRoom = sums.ends_room;
                }
                ++samplesParsed;
            }
            r.usable = r.parsed || r.bytesValid;
            if (!r.usable)
            {
                ++samplesLeft;
                continue;
            }
            ++samplesServed;
        }
        kernelMs += ms;
        gatherSeconds += j.seconds;
        runSeconds += t2 - t1;
        members += j.inOff.size() - 1;
        compressedBytes += j.bytes;
        inflatedBytes += j.outOff.back();
    }

    const CallParams & p_;
    pd_inflater * inflater_[kInflaters] = {};
    bool ok_ = false;
    std::vector<Item> items_;
    std::string contig_;
    std::vector<Ready> ready_;
    std::vector<char> valid_;           // per item: its blocks are the ones its reader wants (adopt())
    std::vector<unsigned char> skip_;   // ... from the block of the skip_-th index region behind the guess on
    std::vector<std::unique_ptr<Prepared>> kept_;      // the chunks' tables (Ready points into them)
    std::unordered_map<unsigned, size_t> slotOf_;
    std::thread worker_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<char> done_;            // per chunk
    bool gathered_ = false;             // every chunk has been gathered (or the round given up)
    bool roundFitsInflaters_ = true;    // the round has at most one chunk per inflater
    unsigned releasedBelow_ = 0;
    bool stop_ = false, failed_ = false;
};

class CallRun
{
public:
    CallRun(CallParams & p, SegmentObserver * obs) : p_(p), observer_(obs) {}
    ~CallRun()
    {
        {
            std::lock_guard<std::recursive_mutex> aheadGuard(aheadLock_);
            aheadClosed_ = true;
        }
        if (handOverThread_.joinable())
        {
            {
                std::lock_guard<std::mutex> guard(handOverLock_);
                handOverStop_ = true;
            }
            handOverWake_.notify_all();
            handOverThread_.join();
        }
        if (backendThread_.joinable())
            backendThread_.join();
        const double t0 = PhaseClock::now();
        for (pd_ctx * c : ctxs_)
            pd_ctx_destroy(c);
        if (group_)
            pd_local_group_destroy(group_);
        const double t1 = PhaseClock::now();
        profile_.reset();
        const double t2 = PhaseClock::now();
        std::vector<DecodedStream>().swap(decodedPool_);
        if (getenv("POPDEL_TIMING"))
            std::cerr << "[popdel timing] teardown: backend " << t1 - t0 << " s, window tables " << t2 - t1 << " s, decoded streams "
                      << PhaseClock::now() - t2 << " s" << std::endl;
        printTiming();
        if (walkCheck_ || walkUnchecked_)
            std::cerr << "[popdel walk check] " << walkChecked_ << " segments planned in closed form and found equal to the walk, "
                      << walkUnchecked_ << " not planned" << std::endl;
    }
    uint64_t profile_materialized_ = 0;
    void printTiming()
    {
        if (profile_)
            profile_materialized_ = profile_->materialized();
        if (getenv("POPDEL_TIMING") && !timingPrinted_)
            std::cerr << "[popdel timing] window loop: " << deviceSegments_ << " segments in closed form on the device, " << hostSegments_
                      << " walked on the host" << std::endl;
        if (getenv("POPDEL_TIMING") && !timingPrinted_)
            std::cerr << "[popdel timing] threads " << hostThreads() << "; parameters " << clock_.params << " s; backend start " << clock_.backend << " s; switches " << clock_.switches << " s; decode " << clock_.decode << " s; replay " << clock_.replay << " s (" << clock_.parallelBatches << " batches in parallel, " << clock_.orderedBatches << " in order)"
                      << "; window walk " << clock_.walk << " s; flatten " << clock_.flatten << " s; stage+submit " << clock_.submit
                      << " s; collect+merge+write " << clock_.collect << " s; on the hand-over thread: hand-over " << workerSubmitSeconds_
                      << " s, collect+merge+write " << workerCollectSeconds_ << " s" << std::endl;
        if (getenv("POPDEL_TIMING") && !timingPrinted_)
        {
            std::lock_guard<std::mutex> guard(deviceNotesLock_);
            if (gunzipSet_[0])
            {
                struct { uint64_t samplesServed = 0, samplesLeft = 0, samplesParsed = 0, members = 0, compressedBytes = 0, inflatedBytes = 0;
                         double gatherSeconds = 0, runSeconds = 0, kernelMs = 0, waitSeconds = 0; } a, b;
                for (auto & g : gunzipSet_)
                {
                    a.samplesServed += g->samplesServed; a.samplesLeft += g->samplesLeft; a.samplesParsed += g->samplesParsed;
                    a.members += g->members; a.compressedBytes += g->compressedBytes; a.inflatedBytes += g->inflatedBytes;
                    a.gatherSeconds += g->gatherSeconds; a.runSeconds += g->runSeconds; a.kernelMs += g->kernelMs; a.waitSeconds += g->waitSeconds;
                }
                std::cerr << "[popdel timing] loader: profile blocks inflated on the device for " << a.samplesServed + b.samplesServed << " sample rounds (" << a.samplesLeft + b.samplesLeft
                          << " left to the host), parsed there for " << a.samplesParsed + b.samplesParsed << "; " << a.members + b.members << " members, " << a.compressedBytes + b.compressedBytes << " -> " << a.inflatedBytes + b.inflatedBytes
                          << " bytes; gather " << a.gatherSeconds + b.gatherSeconds << " s, copies + kernel " << a.runSeconds + b.runSeconds << " s (kernel "
                          << (a.kernelMs + b.kernelMs) / 1e3 << " s); decoder threads waited " << a.waitSeconds + b.waitSeconds << " thread-seconds for it; " << roundsAhead_
                          << " rounds fetched ahead, " << roundsOnDemand_ << " when they began (" << roundsGuessedWrong_ << " of them after a wrong guess); read by the decoder threads all the same: "
                          << hostDecodes_[0].load() << " sample rounds without usable blocks, " << hostDecodes_[1].load() << " whose parsed blocks ended early, "
                          << hostDecodes_[2].load() << " whose records at an end of the round had stayed on the device" << std::endl;
            }
            if (!ctxs_.empty() && ctxs_[0])
            {
                uint64_t gathered = 0, staged = 0, locked = 0, g = 0, st = 0, fromDevice = 0, fd = 0;
                for (pd_ctx * c : ctxs_)
                    if (c && pd_walk_entry_counts(c, &g, &st, &locked) == 0)
                    {
                        gathered += g;
                        staged += st;
                        if (pd_walk_source_counts(c, &fd) == 0)
                            fromDevice += fd;
                    }
                std::cerr << "[popdel timing] start entries of the window loop: " << fromDevice << " taken from the loader's records on the device, " << gathered << " fetched by the device from the tables themselves ("
                          << locked << " bytes of the table pool page-locked), " << staged << " copied by the host threads first" << std::endl;
            }
            std::cerr << "[popdel timing] filing: " << filedRecords_.load() << " records of one-read-group files filed in runs, " << bulkRecords_.load()
                      << " of them from the device's block summaries (" << bulkRuns_.load() << " runs; " << bulkRefused_.load()
                      << " refused by the tables, " << bulkNone_.load() << " without enough usable blocks); " << virtualRecords_.load()
                      << " start entries never left the device, " << profile_materialized_ << " of them were read again from the profiles for the tables, "
                      << runsReadAgain_.load() << " runs for records that had stayed there" << std::endl;
            std::cerr << "[popdel timing] parity: " << nearIntegerShifts_ << " reference-shift quotients within 1e-9 of an integer (their truncation "
                         "may differ from the x87 reference's) in " << candidateCells_ << " candidate cells, " << windowCalls_ << " window calls" << std::endl;
            // steady state: the rounds that began after the backend was up, to the end of the loop (last results written)
            size_t k0 = 0;
            while (k0 < rounds_.size() && rounds_[k0].start < backendReadyAt_)
                ++k0;
            if (k0 < rounds_.size() && loopEnd_ > rounds_[k0].start)
            {
                double decode = 0, replay = 0, device = 0;
                uint64_t windows = 0;
                for (size_t k = k0; k < rounds_.size(); ++k)
                {
                    decode += rounds_[k].decode;
                    replay += rounds_[k].replay;
                    windows += rounds_[k].windows;
                    if (k < deviceMs_.size())
                        device += deviceMs_[k] / 1e3;
                }
                const double wall = loopEnd_ - rounds_[k0].start;
                std::cerr << "[popdel timing] steady state: segments " << rounds_.size() - k0 << " of " << rounds_.size() << "; windows " << windows
                          << "; samples " << p_.sampleNum << "; wall " << wall << " s; decode " << decode << " s; replay " << replay
                          << " s; device kernels " << device << " s" << std::endl;
                // ... and without the ends: the first two of those rounds (the loaders' buffers are allocated and page-locked, the
                // tables' pages touched for the first time) and the last (the drain) -- begin of the third to begin of the last
                if (rounds_.size() >= k0 + 6)
                {
                    uint64_t w = 0;
                    double dec = 0, rep = 0;
                    for (size_t k = k0 + 2; k + 1 < rounds_.size(); ++k)
                    {
                        w += rounds_[k].windows;
                        dec += rounds_[k].decode;
                        rep += rounds_[k].replay;
                    }
                    std::cerr << "[popdel timing] warmed rounds: segments " << rounds_.size() - k0 - 3 << "; windows " << w << "; samples " << p_.sampleNum
                              << "; wall " << rounds_.back().start - rounds_[k0 + 2].start << " s; decode " << dec << " s; replay " << rep << " s; CPU "
                              << rounds_.back().cpu - rounds_[k0 + 2].cpu << " s (all threads)" << std::endl;
                }
            }
        }
        if (!timingPrinted_)
            gTrace.print();
        timingPrinted_ = true;
    }
    bool tableRegionsOffered() const { return regionsOffered_ != 0; }
    void closeOutput() { vcf_.reset(); }
    int run();

private:
    void createBackend();
    void jumpToRoi(ProfileFile & f, unsigned fileNum);
    bool goNextRoi(ProfileFile & f, unsigned fileNum);
    bool goNextRegion();
    Coord firstWindowCoordinate();
    bool firstWindowOnNextRoi(Coord & minCoord);
    bool checkAndSwitch(const std::vector<unsigned> & rg, unsigned beginPos, unsigned * border);
    void decodeSample(unsigned i, const std::string & coordContig, unsigned border, unsigned numWindows, DecodedStream & out) const;
    unsigned replaySegment(unsigned sample, const DecodedStream & in, const std::vector<std::string> & contigNames, Coord & coordinate,
                           const Region & roi, unsigned * border);
    bool replayAdvance(ReplayState & st, const DecodedStream & in, const std::string & fileName, const std::vector<unsigned> & rg,
                       const std::vector<std::string> & contigNames, Coord & coordinate, const Region & roi, unsigned * border);
    // runs filed from the summaries the device takes of the parsed blocks (round 6)
    void fileRun(unsigned sample, const DecodedStream & in, const ReplayState & st);
    void readRunFromProfile(unsigned sample, const std::string & contig, uint32_t jumpPos, uint32_t firstWindow, uint64_t skip, uint64_t n,
                            StartEntry * dst) const;
    // (virtual runs of start entries: which contig a round files -- the note ChromProfile hands back to refill)
    std::vector<std::string> contigTable_;
    std::string roundContig_;
    uint32_t roundContigTag_ = 0;
    std::vector<unsigned> sampleOfRg_;
    std::atomic<uint64_t> virtualRecords_{0}, runsReadAgain_{0};
    bool edgesOnly_ = false;                        // the loaders copy records back for the blocks at a stretch's two ends only
    // Which blocks' records a loader copies back when the middle of a stretch stays on the device: those with a record below the
    // begin + two index regions (the run's head: the blocks it begins in the middle of, and the first whole one) and those with a
    // record within an index region and the longest plausible pair of the end (the block in which the first pair reaches the
    // write slot's border, and what follows). A run that needs more reads it again from the profile (fileRun).
    void setKeepLimits(GunzipPrefetch::Item & it, unsigned sample) const
    {
        const int64_t region = p_.indexRegionSizes[p_.representativeContigs ? 0 : sample];
        it.keepBelow = static_cast<uint32_t>(std::min<int64_t>(static_cast<int64_t>(it.beginPos) + 2 * region + 512, 0xffffffffll));
        it.keepFrom = static_cast<uint32_t>(std::max<int64_t>(static_cast<int64_t>(it.endPos) - 256 - region - 2048, 0));
    }
    int32_t innerOffsetOf(unsigned sample) const
    {
        const InsertHistogram & h = p_.histograms[p_.rgs[sample][0]];
        return static_cast<int32_t>(h.median) - 2 * static_cast<int32_t>(h.readLength);
    }
    static constexpr uint32_t kBulkHead = 64;       // records filed one by one ahead of the summarised blocks: the end table's counting cursor comes home
    static constexpr uint32_t kBulkLeast = 256;     // fewer records are not worth it
    bool bulkFiling_ = false;                       // light tables, the device parses the blocks, POPDEL_HOST_FILING not set
    uint32_t bulkRoom_ = 0;                         // windows a run's counts may span (a slot's worth and some)
    std::atomic<uint64_t> bulkRuns_{0}, bulkRecords_{0}, bulkRefused_{0}, bulkNone_{0}, filedRecords_{0};
    // Where the content of a read group's start slot lies on the device as well: the one run of device-parsed records that was
    // filed into it, none dropped (fileRun). Handed to the window loop with the table (pd_walk_source): the device copies the
    // entries from the loader's records there instead of reading the table over the link. A loader must not start on new blocks
    // before the segment that points into its records has been handed over (loaderJob_, startLoaderAhead()).
    std::vector<pd_walk_source> slotSource_;
    std::vector<uint32_t> slotSourceCount_;
    uint64_t loaderJob_[kLoaderSets] = {};      // per loader: the newest hand-over job that points into its inflaters' records (under handOverLock_)
    bool handedOver(uint64_t seq)
    {
        std::lock_guard<std::mutex> lock(handOverLock_);
        return handOverFinished_ >= seq || handOverError_;
    }
    bool loaderHandedOver(int set)
    {
        std::lock_guard<std::mutex> lock(handOverLock_);
        return handOverFinished_ >= loaderJob_[set] || handOverError_;
    }
    // Who starts the loader of the round to come: the main thread where it passes by (round begin, batch decoded, batch replayed)
    // and -- round 6 -- the threads whose progress it waits for, the moment they are through: the running round's loader when it
    // has gathered its chunks, the hand-over thread when a segment is on the device (startLoaderAheadFrom; they only TRY the
    // lock: the main thread may be waiting for them while it holds it, and looks itself when it lets go).
    std::recursive_mutex aheadLock_;
    bool aheadClosed_ = false;                  // the run is being taken down: nobody starts a loader any more
    void startLoaderAheadFromOutside(const GunzipPrefetch * caller = nullptr)
    {
        std::unique_lock<std::recursive_mutex> lock(aheadLock_, std::try_to_lock);
        if (lock.owns_lock() && !aheadClosed_)
            startLoaderAhead(caller);
    }
    bool processSegmentCode(unsigned i, unsigned code);
    void processSegment();
    void flushPending();
    bool finalizeRoi();
    void markRoiComplete(unsigned i);

    CallParams & p_;
    SegmentObserver * observer_;
    // One context, or (POPDEL_DEVICES=0,1,...) one per listed GPU with the cohort sharded by samples over them: the
    // library exchanges between the contexts of this process itself (pd_ctx_set_shard_local); collective calls are issued
    // from one thread per context.
    std::vector<pd_ctx *> ctxs_;
    pd_local_group * group_ = nullptr;
    std::vector<uint32_t> shardRg_;         // first read group of every context's samples, plus the end
    std::vector<uint32_t> shardSample_;     // first sample of every context, plus the end
    std::vector<uint64_t> tickets_;
    template <typename F>
    void onEveryContext(F fn);
    void chooseMergeSite();
    bool mergeOnDevice_ = false;
    bool mergeCallsOnDevice(std::vector<WindowCall> & calls, const std::vector<pd_calls *> & parts);
    std::unique_ptr<ChromProfile> profile_;
    std::unique_ptr<VcfOutput> vcf_;
    WindowCursor windows_;
    std::vector<char> finishedFiles_, finishedRois_;
    std::vector<Coord> nextCandidate_;
    unsigned nextReadPos_ = kNoPos;
    // one segment in flight on the device while the next one is loaded
    PhaseClock clock_;
    std::vector<uint64_t> flatOffsets_;
    std::vector<int32_t> flatDev_;
    std::vector<uint32_t> flatStart_, flatOn_, flatOff_;
    std::vector<DecodedStream> decodedPool_;
    // POPDEL_WALK_CHECK=1: every segment is planned and evaluated in closed form as well (ChromProfile::planSegment,
    // WalkReference) and compared with what the walk logged -- shape, cursors, instances
    std::unique_ptr<WalkReference> walkCheck_;
    unsigned walkChecked_ = 0, walkUnchecked_ = 0;
    void checkWalkAgainstPlan(const WalkShape & shape, const std::vector<uint8_t> & equalRun, const std::vector<uint8_t> & switches,
                              const std::vector<std::pair<size_t, size_t>> & cursors, unsigned firstPos, unsigned nWindows,
                              const std::vector<std::vector<PairInstance>> & log);
    // The window loop in closed form on the device (pd_walk_submit) instead of walked here: whenever the backend offers it,
    // nobody wants to see the instances (observers, POPDEL_WALK_CHECK) and POPDEL_HOST_WALK is not set. deviceWalk_ says
    // whose turn it is: the device keeps the active pairs while it is on; leaving it (a segment that cannot be planned)
    // fetches them back into the tables' active sets, and a ROI change with no pair left active turns it on again.
    // the backend comes up on its own thread while the first round of profiles is read (run(), ensureBackend())
    std::thread backendThread_;
    std::exception_ptr backendError_;
    std::mutex backendLock_;            // (the main thread and the thread that hands a segment over may both come asking)
    void ensureBackend()
    {
        std::lock_guard<std::mutex> guard(backendLock_);
        if (backendThread_.joinable())
        {
            const double t0 = PhaseClock::now();
            backendThread_.join();
            clock_.backend += PhaseClock::now() - t0;           // (what the caller still had to wait for)
        }
        if (backendError_)                      // (to every caller: the hand-over thread may be the first to ask)
            std::rethrow_exception(backendError_);
    }
    bool deviceWalkOffered_ = false, deviceWalk_ = false, lightTables_ = false;
    // the round's profile blocks inflated on the device (GunzipPrefetch) once the backend is up; POPDEL_HOST_GUNZIP=1: never
    // two loaders: one serves the round that runs, the other works AHEAD on the round to come (startGunzipPrefetch())
    std::unique_ptr<GunzipPrefetch> gunzipSet_[kLoaderSets];
    GunzipPrefetch * gunzip_ = nullptr;         // the running round's, if any
    bool loaderAhead_ = true;                   // POPDEL_LOADER_AHEAD=0: every round's blocks are fetched when the round starts
    size_t aheadDepth_ = 2;                     // rounds the loaders go ahead of the running one (POPDEL_LOADER_DEPTH=1: one, as until round 6)
    // the rounds to come, in their order: loaders that have been started on a guess, and guesses that wait for a set and for the
    // loader before them to have gathered (startLoaderAhead)
    struct AheadLoader
    {
        int     set;
        int64_t begin;
    };
    struct Guess
    {
        std::vector<GunzipPrefetch::Item> items;
        std::string contig;
        int64_t begin = 0;
    };
    std::deque<AheadLoader> aheadLoaders_;
    std::deque<Guess> pendingGuesses_;
    int lastSet_ = -1;                          // the set of the loader that was started last (the running round's or one ahead)
    bool thirdSetUsed_ = false;                 // two rounds of lead have been taken once: every set is warm
    int setsInTurn() const { return thirdSetUsed_ ? static_cast<int>(kLoaderSets) : 2; }    // (the third set stays cold until it is wanted)
    void dropAhead()                            // (a guess did not hold: what was extrapolated from it goes, too)
    {
        for (const AheadLoader & a : aheadLoaders_)
            gunzipSet_[a.set]->finish();
        aheadLoaders_.clear();
        pendingGuesses_.clear();
    }
    uint64_t roundsAhead_ = 0, roundsOnDemand_ = 0, roundsGuessedWrong_ = 0;
    std::vector<int> gunzipDevices_ = {0};
    size_t regionsOffered_ = 0;             // regions of the table pool the backend has been told of (hand-over thread)
    // sample rounds the decoder threads read themselves although a loader ran: [0] it had nothing usable for the sample, [1] its
    // parsed blocks end before the point at which the round's reader stops
    mutable std::atomic<uint64_t> hostDecodes_[3] = {};     // ([2]: a window at an end of the round whose records had not been copied back)
    void startGunzipPrefetch();
    void startLoaderAhead(const GunzipPrefetch * caller = nullptr);    // caller: the loader whose thread asks (it cannot be started over from there)
    void createLoaders();
    unsigned deviceSegments_ = 0, hostSegments_ = 0;
    void leaveDeviceWalk(bool tablesAboutToGo = false);
    void submitPlannedSegment(const WalkShape & shape, const std::vector<uint8_t> & equalRun, const std::vector<uint8_t> & switches);
    // Planned segments go to the backend through a queue with a thread of its own behind it (collect + merge + write of the
    // previous segment, then the hand-over of the next one: the order of the sequential loop), beside the next rounds of
    // decoding. A job normally points INTO the tables' read slots, which the next round leaves alone (a round fills the write
    // slot and recycles the slot after it): the main thread waits for such a job's hand-over before it plans the next
    // segment. While the backend is still coming up (the first 0.2-0.3 s of a run) jobs carry a COPY of their entries instead
    // and nobody waits: the main thread decodes round after round until the device is there. Everything that needs the
    // device's state, the output in order or the contexts themselves drains the queue first (drainHandOver()).
    struct HandOverJob
    {
        std::vector<uint32_t> counts, borders, idOffsets;
        std::vector<const pd_walk_entry *> tables;
        std::vector<pd_walk_source> sources;        // (empty: none) where the entries lie on the device as well
        HugeVector<pd_walk_entry> copy;             // (empty: the tables point into the slot tables)
        std::vector<uint8_t> equalRun, switches;
        WalkShape shape;
        std::string contig;
        uint64_t seq = 0;
    };
    std::thread handOverThread_;
    std::mutex handOverLock_;
    std::condition_variable handOverWake_, handOverDone_;
    std::deque<std::shared_ptr<HandOverJob>> handOverQueue_;
    uint64_t handOverEnqueued_ = 0, handOverFinished_ = 0;      // jobs queued / handed over (or failed)
    uint64_t lastDirectJob_ = 0;                                // newest job that points into the tables
    bool handOverStop_ = false, handOverIdle_ = true;
    std::exception_ptr handOverError_;
    std::atomic<bool> backendReady_{false};
    // (state of the hand-over thread alone: the segment whose results it has yet to collect)
    bool workerPending_ = false;
    std::vector<uint64_t> workerTickets_;
    std::string workerContig_;
    double workerSubmitSeconds_ = 0, workerCollectSeconds_ = 0;
    void handOverLoop();
    void handOver(HandOverJob & job);
    void waitForHandOver(uint64_t seq);         // until job `seq` has been handed over
    void drainHandOver();                       // ... all of them, and the last one's results written
    void collectAndWrite(const std::vector<uint64_t> & tickets, const std::string & contig);
    bool timingPrinted_ = false;
    const bool timingWanted_ = getenv("POPDEL_TIMING") != nullptr;
    // POPDEL_TIMING: the rounds of the segment loop one by one (start, windows of the segment, decode and replay seconds) and
    // what the backend reports per collected segment -- for the steady-state figures of a long run (start-up excluded) and
    // for the count of reference-shift truncations that sit on the knife edge (pd_timing::n_near_integer_shifts)
    struct RoundNote
    {
        double start = 0, decode = 0, replay = 0;
        double cpu = 0;                 // the process's CPU time (user + system) when the round began (POPDEL_TIMING)
        unsigned windows = 0;
    };
    std::vector<RoundNote> rounds_;
    std::mutex deviceNotesLock_;            // (collects run on the hand-over thread)
    std::vector<double> deviceMs_;          // kernel time per collected segment (context 0)
    uint64_t nearIntegerShifts_ = 0, candidateCells_ = 0, windowCalls_ = 0;
    double backendReadyAt_ = 0, loopEnd_ = 0;
    bool pending_ = false;
    std::string pendingContig_;
};

void CallRun::createBackend()
{
    // (POPDEL_HIP_LANES: set by runPopdelCall() before any thread exists)
    std::vector<pd_readgroup> rgs(p_.histograms.size());
    for (size_t i = 0; i < rgs.size(); ++i)
    {
        const InsertHistogram & h = p_.histograms[i];
        pd_readgroup & r = rgs[i];
        r.median = h.median;
        r.read_length = h.readLength;
        r.stddev = h.stddev;
        r.min_prob = h.minProb;
        r.hist_offset = h.offset;
        r.hist_len = static_cast<uint32_t>(h.values.size());
        r.hist_values = h.values.data();
        r.lower_quantile_dist = h.lowerQuantileDist;
        r.upper_quantile_dist = h.upperQuantileDist;
        r.max_load = p_.maxLoad[i];
        r.min_init_del_len = p_.minInitDelLengths[i];
    }
    std::vector<uint32_t> offsets(p_.rgs.size() + 1, 0);
    for (size_t s = 0; s < p_.rgs.size(); ++s)
    {
        // read groups of one file are numbered consecutively
        for (size_t k = 0; k < p_.rgs[s].size(); ++k)
            if (p_.rgs[s][k] != offsets[s] + k)
                throw ProfileError("[PopDel] Internal error: read groups of a sample are not contiguous.");
        offsets[s + 1] = offsets[s] + static_cast<uint32_t>(p_.rgs[s].size());
    }
    pd_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.iterations = p_.iterations;
    cfg.min_len = p_.minLen;
    cfg.min_likelihood_ratio = p_.minimumLikelihoodRatio;
    cfg.min_sample_fraction = p_.minSampleFraction;
    cfg.somatic = p_.somatic ? 1 : 0;
    cfg.window_wise_output = p_.windowWiseOutput ? 1 : 0;
    std::vector<int> devices;
    if (const char * d = getenv("POPDEL_DEVICES"))
    {
        for (const char * q = d; *q;)
        {
            devices.push_back(atoi(q));
            while (*q && *q != ',')
                ++q;
            if (*q == ',')
                ++q;
        }
    }
    if (devices.empty())
    {
        const char * d = getenv("POPDEL_DEVICE");
        devices.push_back(d ? atoi(d) : 0);
    }
    gunzipDevices_ = devices;
    const uint32_t S = static_cast<uint32_t>(p_.rgs.size()), world = static_cast<uint32_t>(devices.size());
    ctxs_.assign(world, nullptr);
    tickets_.assign(world, 0);
    if (world == 1)
    {
        shardSample_ = {0u, S};
        shardRg_ = {0u, static_cast<uint32_t>(rgs.size())};
        if (pd_ctx_create(&cfg, rgs.data(), static_cast<uint32_t>(rgs.size()), offsets.data(), S, devices[0], &ctxs_[0]) != 0)
            throw ProfileError(std::string("[PopDel] Could not initialise the genotyping backend: ") + pd_last_error(nullptr));
        return;
    }
    // the cohort sharded by samples: whole 256-sample tiles per context (pd_shard_samples)
    const uint32_t per = pd_shard_samples(S, world);
    if (static_cast<uint64_t>(per) * (world - 1) >= S)
        throw ProfileError("[PopDel] POPDEL_DEVICES lists " + std::to_string(world) + " contexts, but " + std::to_string(S) +
                           " samples fill only " + std::to_string((S + per - 1) / per) + " (every context takes whole tiles of 256 samples).");
    if (pd_local_group_create(world, &group_) != 0)
        throw ProfileError(std::string("[PopDel] Could not initialise the genotyping backend: ") + pd_last_error(nullptr));
    std::vector<uint32_t> minInit(rgs.size());
    for (size_t i = 0; i < rgs.size(); ++i)
        minInit[i] = rgs[i].min_init_del_len;
    for (uint32_t r = 0; r <= world; ++r)
    {
        shardSample_.push_back(std::min<uint64_t>(S, static_cast<uint64_t>(r) * per));
        shardRg_.push_back(offsets[shardSample_.back()]);
    }
    for (uint32_t r = 0; r < world; ++r)
    {
        const uint32_t s0 = shardSample_[r], s1 = shardSample_[r + 1], g0 = shardRg_[r];
        std::vector<uint32_t> local(offsets.begin() + s0, offsets.begin() + s1 + 1);
        for (uint32_t & o : local)
            o -= g0;
        if (pd_ctx_create(&cfg, rgs.data() + g0, shardRg_[r + 1] - g0, local.data(), s1 - s0, devices[r], &ctxs_[r]) != 0)
            throw ProfileError(std::string("[PopDel] Could not initialise the genotyping backend: ") + pd_last_error(nullptr));
        pd_shard sh;
        std::memset(&sh, 0, sizeof(sh));
        sh.rank = r;
        sh.world = world;
        sh.first_sample = s0;
        sh.n_samples_global = S;
        sh.n_readgroups_global = static_cast<uint32_t>(rgs.size());
        sh.min_init_del_len_global = minInit.data();
        if (pd_ctx_set_shard_local(ctxs_[r], &sh, group_) != 0)
            throw ProfileError(std::string("[PopDel] Could not initialise the genotyping backend: ") + pd_last_error(ctxs_[r]));
    }
}

// Merged output (no -n) and no observer of the raw window calls: the per-sample columns of the window calls stay on the
// device, the backend merges them there (pd_calls_merge) and only the merged columns come back. A backend that does not
// offer this (the oracle) says so and everything is merged on the host as before.
void CallRun::chooseMergeSite()
{
    mergeOnDevice_ = false;
    if (p_.windowWiseOutput || (observer_ && observer_->onWindowCalls) || getenv("POPDEL_HOST_MERGE"))
        return;
    bool all = true;
    for (pd_ctx * c : ctxs_)
        all = all && pd_ctx_keep_results_on_device(c, 1) == 0;
    if (!all)
        for (pd_ctx * c : ctxs_)
            (void)pd_ctx_keep_results_on_device(c, 0);
    mergeOnDevice_ = all;
}

// fn(r) for every context; from one thread per context when there are several (their calls are collective)
template <typename F>
void CallRun::onEveryContext(F fn)
{
    if (ctxs_.size() == 1)
    {
        fn(0u);
        return;
    }
    std::vector<std::thread> threads;
    std::vector<std::exception_ptr> errors(ctxs_.size());
    for (unsigned r = 0; r < ctxs_.size(); ++r)
        threads.emplace_back([&, r] {
            try
            {
                fn(r);
            }
            catch (...)
            {
                errors[r] = std::current_exception();
            }
        });
    for (std::thread & t : threads)
        t.join();
    for (std::exception_ptr & e : errors)
        if (e)
            std::rethrow_exception(e);
}

void CallRun::jumpToRoi(ProfileFile & f, unsigned fileNum)
{
    unsigned idx = p_.representativeContigs ? 0 : fileNum;
    const Region & roi = p_.allRois[p_.nextRoi];
    f.jumpToRegion(p_.contigNames[idx], p_.contigLengths[idx], p_.indexRegionSizes[idx], roi.seqName, roi.beginPos);
}

bool CallRun::goNextRoi(ProfileFile & f, unsigned fileNum)
{
    if (allTrue(finishedRois_))
    {
        if (!advanceRoi(p_))
            return false;
        for (char & b : finishedRois_)
            b = 0;
    }
    jumpToRoi(f, fileNum);
    return true;
}

bool CallRun::goNextRegion()
{
    while (!adaptRegions(windows_.current, p_.allRois[p_.nextRoi], p_.windowSize))
    {
        while (advanceRoi(p_))
        {
            const Region & r = p_.allRois[p_.nextRoi];
            if (r.seqName == windows_.current.contig && r.beginPos >= static_cast<int>(windows_.current.pos))
                break;
        }
        if (p_.nextRoi == p_.allRois.size())
        {
            printStatus("Output written to '" + p_.outfile + "'.");
            return false;
        }
    }
    return true;
}

// Smallest first 30-bp window over all profiles for the current ROI (load_profile_popdel_call.h:291-347).
Coord CallRun::firstWindowCoordinate()
{
    Coord minCoord{"", kNoPos};
    bool windowSeen = false;
    while (true)
    {
        // every profile's first window of the region in parallel (open, index jump, one block inflated); the minimum is
        // taken in file order, as the loop over the files takes it
        if (allTrue(finishedRois_))         // (what goNextRoi() does for whichever file comes first)
        {
            if (!advanceRoi(p_))
                break;
            for (char & b : finishedRois_)
                b = 0;
        }
        struct First
        {
            Coord at{"", kNoPos};
            int32_t chrom = 0, beginPos = 0;
            char state = 0;         // 0: no window (end of the data), 1: a window with records, 2: a window that does not convert
        };
        std::vector<First> firsts(p_.fileCount);
        parallelFor(p_.fileCount, [&](size_t i) {
            const unsigned contigIdx = p_.representativeContigs ? 0 : static_cast<unsigned>(i);
            ProfileFile in(p_.inputFiles[i]);
            jumpToRoi(in, static_cast<unsigned>(i));
            auto src = in.openSource(p_.uncompressedIn);
            ProfileWindow window;
            std::vector<ProfileWindow> converted;
            if (!readWindow(*src, window, static_cast<unsigned>(p_.rgs[i].size())))
                return;
            First & f = firsts[i];
            f.chrom = window.chrom;
            f.beginPos = window.beginPos;
            f.state = 2;
            if (!splitWindow(window, converted, 256, 30))
                return;
            f.at = Coord{p_.contigNames[contigIdx][converted[0].chrom], static_cast<unsigned>(converted[0].beginPos)};
            f.state = 1;
        }, 8);
        for (unsigned i = 0; i < p_.fileCount; ++i)
        {
            // (the reference reads every file into the same window object, which keeps its read groups once a file had
            // some: from then on a file without data in the region is not skipped but fails to convert)
            const unsigned contigIdx = p_.representativeContigs ? 0 : i;
            First & f = firsts[i];
            if (f.state == 0 && !windowSeen)
                continue;
            if (f.state != 1)
                throw ProfileError("[PopDel] Could not convert window " + p_.contigNames[contigIdx][f.chrom] + ":" +
                                   std::to_string(f.beginPos) + ". The profile \"" + p_.inputFiles[i] +
                                   "\" might be corrupted.");
            windowSeen = true;
            if (minCoord.contig.empty() || !lowerCoord(minCoord, f.at, p_))
                minCoord = f.at;
        }
        if (minCoord.contig == p_.allRois[p_.nextRoi].seqName)
            break;
        advanceRoi(p_);
        if (p_.nextRoi == p_.allRois.size())
            break;
    }
    if (minCoord.contig.empty() || p_.nextRoi == p_.allRois.size())
        throw ProfileError("[PopDel] Could not find any entries for any of the specified ROIs or reference sequence of the "
                           "first sample. Terminating.");
    std::ostringstream msg;
    msg << "The first window of all profiles starts at '" << minCoord.contig << ":" << minCoord.pos << "'.";
    printStatus(msg);
    return minCoord;
}

bool CallRun::firstWindowOnNextRoi(Coord & minCoord)
{
    minCoord = Coord{"", kNoPos};
    ProfileWindow window;
    std::vector<ProfileWindow> converted;
    while (true)
    {
        for (unsigned i = 0; i < p_.fileCount; ++i)
        {
            unsigned contigIdx = p_.representativeContigs ? 0 : i;
            ProfileFile in(p_.inputFiles[i]);
            goNextRoi(in, i);
            auto src = in.openSource(p_.uncompressedIn);
            readWindow(*src, window, static_cast<unsigned>(p_.rgs[i].size()));
            if (window.empty())
                continue;
            if (!splitWindow(window, converted, 256, 30))
                throw ProfileError("[PopDel] Could not convert window at " + p_.contigNames[contigIdx][window.chrom] + ":" +
                                   std::to_string(window.beginPos) + ". The profile \"" + p_.inputFiles[i] +
                                   "\" might be corrupted.");
            Coord cur{p_.contigNames[contigIdx][converted[0].chrom], static_cast<unsigned>(converted[0].beginPos)};
            if (minCoord.contig.empty() || !lowerCoord(minCoord, cur, p_))
                minCoord = cur;
        }
        if (minCoord.contig == p_.allRois[p_.nextRoi].seqName)
            break;
        if (p_.nextRoi + 1 != p_.allRois.size())
            advanceRoi(p_);
        else
            return false;
    }
    return true;
}

// Who reads in the round that begins, from where and up to where (see GunzipPrefetch): the samples that are not through
// with the file or the region and whose slots will not be switched (checkAndSwitch() below: a switch ends a sample's turn).
void CallRun::startGunzipPrefetch()
{
    std::lock_guard<std::recursive_mutex> aheadGuard(aheadLock_);
    gunzip_ = nullptr;
    if (!backendReady_.load() && getenv("POPDEL_DEVICE_GUNZIP_WAIT") && deviceWalkOffered_)
        ensureBackend();            // (tests: short runs are over before the backend is up and would never take this path)
    if (!backendReady_.load() || !gunzipSet_[0])      // (the loaders are created by the backend's thread, before it says ready)
        return;
    bool loadersFine = true;
    for (auto & g : gunzipSet_)
        loadersFine = loadersFine && g->ok();
    if (!loadersFine)
    {
        // (a loader gave up: this and the rounds to come are inflated on the host)
        for (auto & g : gunzipSet_)
            g->finish();
        aheadLoaders_.clear();
        pendingGuesses_.clear();
        return;
    }
    const Region & roi = p_.allRois[p_.nextRoi];
    std::vector<GunzipPrefetch::Item> items;
    int64_t farthest = -1, nearestBorder = -1;
    for (unsigned i = 0; i < p_.sampleNum; ++i)
    {
        if (finishedFiles_[i] || finishedRois_[i])
            continue;
        const StartSlots & st = profile_->starts[p_.rgs[i][0]];
        const unsigned pos = nextCandidate_[i].pos;
        if (st.tooBigForNext(pos) || st.needsSwitch(pos))
            continue;
        const int64_t right = st.slots[st.writeSlot].right;
        nearestBorder = nearestBorder < 0 ? right : std::min(nearestBorder, right);
        GunzipPrefetch::Item it;
        it.sample = i;
        it.parse = p_.rgs[i].size() == 1;
        it.innerOffset = innerOffsetOf(i);
        it.beginPos = roi.beginPos;
        it.firstRegion = static_cast<uint32_t>(roi.beginPos) / p_.indexRegionSizes[p_.representativeContigs ? 0 : i];
        // (the decoder stops at the first window at or beyond the border; 256-bp windows may begin up to 255 bp before it)
        const int64_t end = std::min<int64_t>(static_cast<int64_t>(st.slots[st.writeSlot].right) + 256, static_cast<int64_t>(roi.endPos) + 256);
        it.endPos = static_cast<int32_t>(std::min<int64_t>(end, 0x7fffffff));
        setKeepLimits(it, i);
        if (it.endPos >= it.beginPos)
        {
            items.push_back(it);
            farthest = std::max<int64_t>(farthest, it.endPos);
        }
    }
    // 1. this round: the loader that went ahead if its guess holds for (nearly) all readers, else a fresh one
    int current = -1;
    static const bool trace = getenv("POPDEL_LOADER_TRACE") != nullptr;
    if (trace)
        std::cerr << "[popdel loader] round begins at " << roi.seqName << ":" << roi.beginPos << " (region ends at " << roi.endPos << "), " << items.size()
                  << " readers, farthest end " << farthest
                  << (!aheadLoaders_.empty() ? "; guessed begin " + std::to_string(aheadLoaders_.front().begin) : std::string()) << std::endl;
    if (!aheadLoaders_.empty())
    {
        const AheadLoader ahead = aheadLoaders_.front();
        aheadLoaders_.pop_front();
        const size_t covered = items.empty() ? 0 : gunzipSet_[ahead.set]->adopt(items, roi.seqName);
        if (trace)
            std::cerr << "[popdel loader] the guess covers " << covered << " of them" << std::endl;
        if (!items.empty() && covered * 10 >= items.size() * 9)
        {
            current = ahead.set;
            ++roundsAhead_;
        }
        else
        {
            gunzipSet_[ahead.set]->finish();
            dropAhead();
            ++roundsGuessedWrong_;
        }
    }
    else
        pendingGuesses_.clear();                    // (a guess for THIS round that never got a loader)
    if (current < 0 && !items.empty())
    {
        dropAhead();
        current = (lastSet_ + 1) % setsInTurn();
        uint64_t lastJob = 0;
        {
            std::lock_guard<std::mutex> lock(handOverLock_);
            lastJob = loaderJob_[current];
        }
        waitForHandOver(lastJob);                   // (no segment may still point into that loader's records on the device)
        gunzipSet_[current]->edgesOnly = edgesOnly_ && deviceWalk_;     // (with the window loop back on the host every record is wanted here)
        gunzipSet_[current]->start(items, roi.seqName);
        lastSet_ = current;
        ++roundsOnDemand_;
    }
    if (current >= 0)
        gunzip_ = gunzipSet_[current].get();
    // 2. the rounds to come, guessed: everybody who still has data reads, from the border at which this round's readers stop
    //    (the next round begins at the first window at or beyond it: the same index region unless the border lies within a
    //    window's length of the region's end) and one slot further -- and the round after that one more slot on. Their chunks are
    //    inflated and parsed while this round is decoded and replayed. A wrong guess costs device time, never a result (adopt()).
    if (!loaderAhead_ || current < 0 || farthest < 0 || nearestBorder <= roi.beginPos)
    {
        dropAhead();
        return;
    }
    // (a round is a slot long: the borders lie a window buffer apart -- this round may be shorter, a region begins where it begins)
    const int64_t slotLen = profile_->starts[p_.rgs[items[0].sample][0]].numWindows;
    // (the second round of lead takes the third loader set, whose first use costs what every set's first use costs -- descriptors,
    // page-locked and device buffers: about a round at 50,000 samples. On a region with only a few rounds to go that is not earned
    // back -- 50,000 profiles x 1 Mb, five rounds: 8.2 s with one round of lead, 11.1 with two --, so: only with eight rounds ahead)
    const size_t depth = thirdSetUsed_ || (static_cast<int64_t>(roi.endPos) - nearestBorder) / std::max<int64_t>(slotLen, 1) >= 8 ? aheadDepth_ : 1;
    for (size_t k = aheadLoaders_.size() + pendingGuesses_.size(); k < depth; ++k)
    {
        // (round k + 1 from here)
        const int64_t nextBegin = nearestBorder + static_cast<int64_t>(k) * slotLen;
        if (nextBegin >= roi.endPos)
            break;
        thirdSetUsed_ = thirdSetUsed_ || k >= 1;
        const int64_t nextEnd = std::min<int64_t>(std::min<int64_t>(nextBegin + slotLen, static_cast<int64_t>(roi.endPos)) + 256, 0x7fffffff);
        Guess g;
        for (unsigned i = 0; i < p_.sampleNum; ++i)
        {
            if (finishedFiles_[i])
                continue;
            GunzipPrefetch::Item it;
            it.sample = i;
            it.parse = p_.rgs[i].size() == 1;
            it.innerOffset = innerOffsetOf(i);
            it.beginPos = static_cast<int32_t>(nextBegin);
            it.firstRegion = static_cast<uint32_t>(nextBegin) / p_.indexRegionSizes[p_.representativeContigs ? 0 : i];
            it.endPos = static_cast<int32_t>(nextEnd);
            setKeepLimits(it, i);
            if (it.endPos >= it.beginPos)
                g.items.push_back(it);
        }
        if (g.items.empty())
            break;
        g.contig = roi.seqName;
        g.begin = nextBegin;
        pendingGuesses_.push_back(std::move(g));
    }
    // (started once the loader before it has gathered its chunks, and its set is free: startLoaderAhead(), called batch by batch
    // and by whoever makes one of the two come true)
    startLoaderAhead();
}

void CallRun::createLoaders()
{
    if (!deviceWalkOffered_ || p_.uncompressedIn || getenv("POPDEL_HOST_GUNZIP"))
        return;
    size_t first = 0;
    for (auto & g : gunzipSet_)
    {
        g.reset(new GunzipPrefetch(p_, gunzipDevices_, first));
        g->onProgress = [this, self = g.get()] { startLoaderAheadFromOutside(self); };
        first += GunzipPrefetch::kInflaters;
        // (the parsed blocks summarised for the tables: whenever the tables can make use of it -- light end tables, which go with
        // the device's window loop; a block spans one index region, its pairs end a little further on)
        uint32_t region = 0;
        for (uint32_t r : p_.indexRegionSizes)
            region = std::max(region, r);
        g->startsRoom = (region + 256u) / 30u + 3u;         // (a block's last 256-bp window may reach a window's length beyond its region)
        g->endsRoom = g->startsRoom + 128u;
        g->summariesWanted = deviceWalkOffered_ && !getenv("POPDEL_HEAVY_TABLES") && !getenv("POPDEL_HOST_FILING") && region != 0 &&
                             g->startsRoom + g->endsRoom <= 12000u;
        g->leanWanted = !getenv("POPDEL_FULL_PARSE");
        // (... and the records of the blocks in the middle of a stretch stay on the device: every context of the run on the loaders'
        // GPU -- one context, or several on one GPU --, whose window loops take their entries from there; with the contexts on
        // several GPUs a sample's blocks may be parsed on another GPU than its context's, and every record is copied back;
        // POPDEL_HOST_RECORDS=1 copies every record back as before)
        const char * src = getenv("POPDEL_HIP_WALK_SOURCES");
        bool oneGpu = true;
        for (int dev : gunzipDevices_)
            oneGpu = oneGpu && dev == gunzipDevices_[0];
        g->edgesOnly = g->summariesWanted && g->leanWanted && oneGpu && !getenv("POPDEL_HOST_RECORDS") && !getenv("POPDEL_COPY_HAND_OVER") &&
                       !(src && src[0] == '0');
        edgesOnly_ = g->edgesOnly;
    }
    const char * ahead = getenv("POPDEL_LOADER_AHEAD");
    loaderAhead_ = !(ahead && ahead[0] == '0');
    if (const char * depth = getenv("POPDEL_LOADER_DEPTH"))
        aheadDepth_ = std::min<size_t>(std::max(1, atoi(depth)), kLoaderSets - 1);
}

void CallRun::startLoaderAhead(const GunzipPrefetch * caller)
{
    // (not before the running round's loader is through: started at once, the two loaders share the device and the round
    // that is waited for gets its blocks later -- 0.064 instead of 0.061 s per warmed round)
    // (round 6, with the records staying on the device: the running round's loader need only have GATHERED its chunks -- its last
    // gunzip launches and the next round's first then share the chip, and the kernel, a chain of dependent instructions per member,
    // gets more out of it the more members are in flight; POPDEL_LOADER_OVERLAP=0 waits for the chunks to be through as before)
    static const bool overlap = [] { const char * e = getenv("POPDEL_LOADER_OVERLAP"); return !(e && e[0] == '0'); }();
    std::lock_guard<std::recursive_mutex> aheadGuard(aheadLock_);
    // (POPDEL_LOADER_OUTSTANDING=n: ... and with at most n of its chunks still on the device. A round whose loader still has two or
    // three chunks there when the next one starts can take 45 ms instead of 16-27 -- the next round's copies and kernels get in front
    // on the queues they share --, but the rounds after it are the shorter for it: 0.0255 / 0.0249 s per warmed round with n = 1,
    // 0.0272 / 0.0267 with 0, 0.0248 / 0.0241 without a limit, which stays the default)
    static const size_t outstandingAllowed = [] { const char * e = getenv("POPDEL_LOADER_OUTSTANDING"); return e ? static_cast<size_t>(atoi(e)) : ~static_cast<size_t>(0); }();
    while (!pendingGuesses_.empty() && gunzip_)
    {
        // the loader before it: the one started last
        GunzipPrefetch * before = lastSet_ >= 0 ? gunzipSet_[lastSet_].get() : nullptr;
        if (before && !(overlap ? before->allGathered() && before->outstanding() <= outstandingAllowed : before->allDone()))
            return;
        // its set: the next in turn -- not the running round's, not one that is ahead, and no segment may point into its records any more
        const int set = (lastSet_ + 1) % setsInTurn();
        if (gunzipSet_[set].get() == gunzip_ || gunzipSet_[set].get() == caller)
            return;                                 // (caller: a thread of that very loader, late with its last report -- start() would join it)
        for (const AheadLoader & a : aheadLoaders_)
            if (a.set == set)
                return;
        if (!loaderHandedOver(set))
            return;
        Guess g = std::move(pendingGuesses_.front());
        pendingGuesses_.pop_front();
        gunzipSet_[set]->edgesOnly = edgesOnly_ && deviceWalk_;
        gunzipSet_[set]->start(std::move(g.items), g.contig);
        aheadLoaders_.push_back(AheadLoader{set, g.begin});
        lastSet_ = set;
    }
}

bool CallRun::checkAndSwitch(const std::vector<unsigned> & rg, unsigned beginPos, unsigned * border)
{
    ChromProfile & prof = *profile_;
    if (prof.starts[rg[0]].tooBigForNext(beginPos) || prof.starts[rg[0]].needsSwitch(beginPos))
    {
        prof.performSwitches(rg, border);
        return false;
    }
    if (prof.ends[rg[0]].readSlot == prof.ends[rg[0]].writeSlot)
    {
        for (unsigned r : rg)
        {
            prof.ends[r].switchWrite();
            prof.ends[r].fixConsecutiveSwitch(prof.activeLoad[r]);
        }
    }
    return true;
}

// Decoder thread body: everything readSegment() does that needs no profile state. The slot border is tracked
// optimistically (a window far beyond the next slot is assumed to trigger the "tables are empty -> reset" branch); the
// sequential replay takes the real decision and simply stops earlier if it differs.
void CallRun::decodeSample(unsigned i, const std::string & coordContig, unsigned border, unsigned numWindows,
                           DecodedStream & out) const
{
    const unsigned contigIdx = p_.representativeContigs ? 0 : i;
    const std::vector<std::string> & contigNames = p_.contigNames[contigIdx];
    const Region & roi = p_.allRois[p_.nextRoi];
    const unsigned borderIn = border;
    const GunzipPrefetch::Ready * pre = gunzip_ ? gunzip_->wait(i) : nullptr;
    if (gunzip_ && !pre)
        ++hostDecodes_[0];
    if (pre && pre->parsed && pre->lean)
    {
        // The device's lean arrays: the same scan, a 256-bp window taken by the positions of its first and last record where
        // that decides it, window by window over its records where not.
        size_t r0 = 0, r = 0;
        bool stopped = false, missed = false;
        int32_t coordChrom = -1, roiChrom = -1, otherChrom = -1;
        for (; r < pre->nRaws && !stopped; ++r)
        {
            const LeanRaw & raw = pre->lean[r];
            if (raw.chrom != coordChrom)
            {
                if (contigNames[raw.chrom] != coordContig)
                {
                    stopped = true;             // (the window of the other contig is the last one the replay sees)
                    break;
                }
                coordChrom = raw.chrom;
            }
            if (raw.beginPos + 255 < roi.beginPos)
            {
                if (r == r0)
                    ++r0;                   // (windows ahead of the region: a prefix, not handed on)
                continue;
            }
            if (raw.nRecs == 0)
                continue;
            if (raw.chrom != roiChrom && raw.chrom != otherChrom)
            {
                if (contigNames[raw.chrom] == roi.seqName)
                    roiChrom = raw.chrom;
                else
                    otherChrom = raw.chrom;
            }
            const uint32_t head = raw.firstPos / 30u * 30u, tail = raw.lastPos / 30u * 30u;
            if (raw.chrom == roiChrom && static_cast<int64_t>(head) >= roi.beginPos && static_cast<int64_t>(tail) < roi.endPos && tail < border && head <= tail)
                continue;
            uint32_t last = 0xffffffffu;
            const PairRecord * rr = pre->records(raw.firstRec, raw.nRecs);
            if (!rr)
            {
                missed = true;                  // (its records were not copied back: the guess of where the round's ends lie was off)
                break;
            }
            for (uint32_t k = 0; k < raw.nRecs; ++k)
            {
                const uint32_t wb = rr[k].pos / 30u * 30u;
                if (wb == last)
                    continue;
                last = wb;
                if (raw.chrom != roiChrom || static_cast<int64_t>(wb) >= roi.endPos)
                {
                    stopped = true;
                    break;
                }
                if (static_cast<int64_t>(wb) < roi.beginPos)
                    continue;
                if (wb >= border + numWindows)
                {
                    border = wb + numWindows;
                    out.sawFarWindow = true;
                }
                else if (wb >= border)
                {
                    stopped = true;
                    break;
                }
            }
            if (stopped)
                break;
        }
        const bool atEnd = !stopped && !missed && pre->toFileEnd;
        if (missed)
            ++hostDecodes_[2];
        if (!missed && (stopped || atEnd))
        {
            out.eof = atEnd;
            out.leanV = pre->lean + r0;
            out.nLeanV = (atEnd ? pre->nRaws : r + 1) - r0;
            out.recsV = pre->recs;
            out.nRecsV = pre->nRecs;
            out.blockRecs = pre->blockRecs;
            out.nBlocks = pre->nBlocks;
            out.blockSums = pre->blockSums;
            out.blockCounts = pre->blockCounts;
            out.startsRoom = pre->startsRoom;
            out.endsRoom = pre->endsRoom;
            out.inflater = pre->inflater;
            out.recBase = pre->recBase;
            out.runId = pre->runId;
            out.recIndex = pre->recIndex;
            out.recsPacked = pre->recsPacked;
            return;
        }
        if (!missed)
            ++hostDecodes_[1];
        out.sawFarWindow = false;           // (decoded again below)
        border = borderIn;
        if (!pre->bytesValid)
            pre = nullptr;
    }
    else if (pre && pre->parsed)
    {
        // The device has inflated AND parsed the blocks from the jump position on: the loop below, on its arrays. If the point at
        // which the loader stops lies within them (the rule) the replay reads them where they are; if not, the sample is decoded
        // the ordinary way.
        size_t r0 = 0;
        bool stopped = false;
        size_t r = 0;
        // (contig names are compared once per contig, not per window; a 256-bp window whose 30-bp windows all lie inside the
        // region and below the border -- they are in position order and share the window's contig -- changes nothing below and
        // is passed in one step: all but the last of a round's 800 per sample)
        int32_t coordChrom = -1, roiChrom = -1;
        for (; r < pre->nRaws && !stopped; ++r)
        {
            const DecodedRaw & raw = pre->raws[r];
            if (raw.chrom != coordChrom)
            {
                if (contigNames[raw.chrom] != coordContig)
                {
                    stopped = true;             // (the window of the other contig is the last one the replay sees)
                    break;
                }
                coordChrom = raw.chrom;
            }
            if (raw.beginPos + 255 < roi.beginPos)
            {
                if (r == r0)
                    ++r0;                   // (windows ahead of the region: a prefix, not handed on)
                continue;
            }
            if (raw.nWin != 0)
            {
                if (raw.chrom != roiChrom && contigNames[raw.chrom] == roi.seqName)
                    roiChrom = raw.chrom;
                const FlatWindow & head = pre->wins[raw.firstWin], & tail = pre->wins[raw.firstWin + raw.nWin - 1];
                if (raw.chrom == roiChrom && head.chrom == raw.chrom && tail.chrom == raw.chrom && head.beginPos >= roi.beginPos &&
                    tail.beginPos < roi.endPos && static_cast<unsigned>(tail.beginPos) < border && head.beginPos <= tail.beginPos)
                    continue;
            }
            for (uint32_t wi = raw.firstWin; wi < raw.firstWin + raw.nWin; ++wi)
            {
                const FlatWindow & w = pre->wins[wi];
                if (contigNames[w.chrom] != roi.seqName || w.beginPos >= roi.endPos)
                {
                    stopped = true;
                    break;
                }
                if (w.beginPos < roi.beginPos)
                    continue;
                const unsigned pos = static_cast<unsigned>(w.beginPos);
                if (pos >= border + numWindows)
                {
                    border = pos + numWindows;
                    out.sawFarWindow = true;
                }
                else if (pos >= border)
                {
                    stopped = true;
                    break;
                }
            }
            if (stopped)
                break;
        }
        // (the file ends before the loader would stop: the reader runs into the end of the file behind the last window)
        const bool atEnd = !stopped && pre->toFileEnd;
        if (stopped || atEnd)
        {
            out.eof = atEnd;
            out.rawsV = pre->raws + r0;
            out.nRawsV = (atEnd ? pre->nRaws : r + 1) - r0;
            out.winsV = pre->wins;
            out.countsV = pre->counts;
            out.recsV = pre->recs;
            out.nRecsV = pre->nRecs;
            out.blockRecs = pre->blockRecs;
            out.nBlocks = pre->nBlocks;
            out.blockSums = pre->blockSums;
            out.blockCounts = pre->blockCounts;
            out.startsRoom = pre->startsRoom;
            out.endsRoom = pre->endsRoom;
            out.inflater = pre->inflater;
            out.recBase = pre->recBase;
            out.runId = pre->runId;
            return;
        }
        ++hostDecodes_[1];
        out.sawFarWindow = false;           // (decoded again below)
        border = borderIn;
        if (!pre->bytesValid)
            pre = nullptr;
    }
    ProfileFile in(p_.inputFiles[i]);
    std::unique_ptr<ByteSource> src;
    if (pre && pre->bytesValid)
        src = in.openPrefetched(pre->data, pre->size, pre->resume);     // (the device has inflated the blocks from the jump position on)
    else
    {
        in.jumpToRegion(contigNames, p_.contigLengths[contigIdx], p_.indexRegionSizes[contigIdx], roi.seqName, roi.beginPos);
        src = in.openSource(p_.uncompressedIn);
    }
    const unsigned nRg = static_cast<unsigned>(p_.rgs[i].size());
    WindowHead window;
    for (;;)
    {
        if (!readWindowHead(*src, window))
        {
            out.eof = true;
            return;
        }
        DecodedRaw raw;
        raw.chrom = window.chrom;
        raw.beginPos = window.beginPos;
        if (contigNames[window.chrom] != coordContig)
        {
            skipWindowBody(*src, nRg);
            out.raws.push_back(std::move(raw));
            return;
        }
        if (window.beginPos + 255 < roi.beginPos)
        {
            skipWindowBody(*src, nRg);
            continue;
        }
        raw.firstWin = static_cast<uint32_t>(out.wins.size());
        if (!readWindowBodyFlat(*src, window, nRg, out.wins, out.counts, out.recs, 256, 30))
        {
            raw.splitFailed = true;
            out.raws.push_back(std::move(raw));
            return;
        }
        raw.nWin = static_cast<uint32_t>(out.wins.size()) - raw.firstWin;
        bool stop = false;
        for (uint32_t wi = raw.firstWin; wi < raw.firstWin + raw.nWin; ++wi)
        {
            const FlatWindow & w = out.wins[wi];
            if (contigNames[w.chrom] != roi.seqName || w.beginPos >= roi.endPos)
            {
                stop = true;
                break;
            }
            if (w.beginPos < roi.beginPos)
                continue;
            const unsigned pos = static_cast<unsigned>(w.beginPos);
            if (pos >= border + numWindows)
            {
                border = pos + numWindows;
                out.sawFarWindow = true;
            }
            else if (pos >= border)
            {
                stop = true;
                break;
            }
        }
        out.raws.push_back(std::move(raw));
        if (stop)
            return;
    }
}

// Sequential half of readSegment(): same decisions, same order, fed from a decoded stream. Returns true with st.haveRun set when a
// run of records of a one-read-group file is to be filed (fileRun()), false when the loop is over (st.code says how).
bool CallRun::replayAdvance(ReplayState & st, const DecodedStream & in, const std::string & fileName, const std::vector<unsigned> & rg,
                            const std::vector<std::string> & contigNames, Coord & coordinate, const Region & roi, unsigned * border)
{
    ChromProfile & prof = *profile_;
    const bool oneRg = rg.size() == 1;
    st.haveRun = false;
    auto stop = [&](unsigned code) {
        st.finished = true;
        st.code = code;
        return false;
    };
    if (in.leanV)
    {
        // ---- the device's lean arrays (one read group): the same loop with the 30-bp windows read off the records. st.wi is the
        // index of the next record to look at (in the sample's stream). Records are read only in 256-bp windows that do not pass
        // as a whole -- the ones at the two ends of what the round files, whose records decodeSample() found copied back.
        const StartSlots & sst = prof.starts[rg[0]];
        auto recordsOfRaw = [&](const LeanRaw & raw) {
            const PairRecord * rr = in.records(raw.firstRec, raw.nRecs);
            if (!rr)
                throw ProfileError("[PopDel] Internal error: the records of a window at an end of the round (\"" + fileName + "\") were not copied back from the device.");
            return rr - raw.firstRec;           // (indexed by the record's number in the stream)
        };
        while (!st.finished)
        {
            if (!st.inRaw)
            {
                if (st.rawIdx >= in.nLeanV)
                {
                    if (!in.eof)
                        throw ProfileError("[PopDel] Internal error: decoded stream of \"" + fileName + "\" ended before the loader stopped.");
                    prof.performPartialSwitches(rg, border);
                    coordinate.contig = "";
                    coordinate.pos = kNoPos;
                    return stop(3);
                }
                const LeanRaw & raw = in.leanV[st.rawIdx];
                if (contigNames[raw.chrom] != coordinate.contig)
                {
                    prof.performPartialSwitches(rg, border);
                    coordinate.contig = contigNames[raw.chrom];
                    coordinate.pos = static_cast<unsigned>(raw.beginPos);
                    return stop(2);
                }
                st.onRoiContig = contigNames[raw.chrom] == roi.seqName;
                st.wi = raw.firstRec;
                st.inRaw = true;
            }
            const LeanRaw & raw = in.leanV[st.rawIdx];
            const uint32_t endOfThis = raw.firstRec + raw.nRecs;
            if (st.wi >= endOfThis)
            {
                st.inRaw = false;
                ++st.rawIdx;
                continue;
            }
            unsigned right = sst.slots[sst.writeSlot].right;
            auto passes = [&](uint32_t pos) {
                const uint32_t b = pos / 30u * 30u;
                return static_cast<int64_t>(b) >= roi.beginPos && static_cast<int64_t>(b) < roi.endPos && b < right;
            };
            auto passesWhole = [&](const LeanRaw & x) { return passes(x.firstPos) && passes(x.lastPos) && x.firstPos <= x.lastPos; };
            uint32_t j;                         // the run will be records [st.wi, j)
            const PairRecord * rr = nullptr;    // the records of the 256-bp window the loop stands in, if they are looked at
            if (st.wi == raw.firstRec && st.onRoiContig && passesWhole(raw))
            {
                // (all its 30-bp windows begin inside the region and below the write slot's border: nothing stops the loader, the
                // run begins with its first record)
                j = endOfThis;
                st.runFirstWindow = raw.firstPos / 30u * 30u;
            }
            else
            {
                rr = recordsOfRaw(raw);
                // the 30-bp window of record st.wi: its records are [st.wi, we)
                const uint32_t wb = rr[st.wi].pos / 30u * 30u;
                uint32_t we = st.wi + 1;
                while (we < endOfThis && rr[we].pos / 30u * 30u == wb)
                    ++we;
                if (!st.onRoiContig || static_cast<int64_t>(wb) >= roi.endPos)
                {
                    prof.performPartialSwitches(rg, border);
                    coordinate.contig = contigNames[raw.chrom];
                    coordinate.pos = wb;
                    return stop(2);
                }
                if (static_cast<int64_t>(wb) < roi.beginPos)
                {
                    st.wi = we;
                    continue;
                }
                if (sst.tooBigForNext(wb))
                {
                    if (prof.allEmpty())
                    {
                        prof.resetTo(wb);
                        right = sst.slots[sst.writeSlot].right;
                    }
                    else
                    {
                        prof.performPartialSwitches(rg, border);
                        coordinate.contig = contigNames[raw.chrom];
                        coordinate.pos = wb;
                        return stop(0);
                    }
                }
                else if (sst.needsSwitch(wb))
                {
                    prof.performPartialSwitches(rg, border);
                    coordinate.contig = contigNames[raw.chrom];
                    coordinate.pos = wb;
                    return stop(0);
                }
                j = we;
                st.runFirstWindow = wb;
            }
            // the run: these records and those of the windows behind them that pass as well (see the full form below)
            size_t rj = st.rawIdx;
            uint32_t endOfRaw = endOfThis;
            int32_t knownChrom = raw.chrom;         // (its name equals coordinate.contig and roi.seqName)
            for (;;)
            {
                while (j < endOfRaw && passes(rr[j].pos))      // (j < endOfRaw only inside a window whose records are at hand)
                    ++j;
                if (j < endOfRaw)
                    break;
                bool more = false;
                while (rj + 1 < in.nLeanV)
                {
                    const LeanRaw & nx = in.leanV[rj + 1];
                    if (nx.nRecs == 0 || nx.firstRec != endOfRaw)
                        break;
                    if (nx.chrom != knownChrom)
                    {
                        if (contigNames[nx.chrom] != coordinate.contig || contigNames[nx.chrom] != roi.seqName)
                            break;
                        knownChrom = nx.chrom;
                    }
                    ++rj;
                    endOfRaw = nx.firstRec + nx.nRecs;
                    if (passesWhole(nx))
                    {
                        j = endOfRaw;
                        continue;
                    }
                    j = nx.firstRec;
                    rr = recordsOfRaw(nx);
                    more = true;                    // (record by record through this one)
                    break;
                }
                if (!more)
                    break;
            }
            st.haveRun = true;
            st.runFirst = st.wi;
            st.runCount = j - st.wi;
            st.rawIdx = rj;
            st.wi = j;
            st.inRaw = true;
            return true;
        }
        return false;
    }
    while (!st.finished)
    {
        if (!st.inRaw)
        {
            if (st.rawIdx >= in.nRawsV)
            {
                if (!in.eof)
                    throw ProfileError("[PopDel] Internal error: decoded stream of \"" + fileName + "\" ended before the loader stopped.");
                prof.performPartialSwitches(rg, border);
                coordinate.contig = "";
                coordinate.pos = kNoPos;
                return stop(3);
            }
            const DecodedRaw & raw = in.rawsV[st.rawIdx];
            if (contigNames[raw.chrom] != coordinate.contig)
            {
                prof.performPartialSwitches(rg, border);
                coordinate.contig = contigNames[raw.chrom];
                coordinate.pos = static_cast<unsigned>(raw.beginPos);
                return stop(2);
            }
            if (raw.splitFailed)
                throw ProfileError("[PopDel] Could not convert window at " + contigNames[raw.chrom] + ":" +
                                   std::to_string(raw.beginPos) + ". The profile \"" + fileName + "\" might be corrupted.");
            st.onRoiContig = contigNames[raw.chrom] == roi.seqName;         // (the 30-bp windows of a raw window share its contig)
            st.wi = raw.firstWin;
            st.inRaw = true;
        }
        const DecodedRaw & raw = in.rawsV[st.rawIdx];
        const uint32_t wEnd = raw.firstWin + raw.nWin;
        if (st.wi >= wEnd)
        {
            st.inRaw = false;
            ++st.rawIdx;
            continue;
        }
        const uint32_t wi = st.wi;
        const FlatWindow & w = in.winsV[wi];
        if (!st.onRoiContig || w.beginPos >= roi.endPos)
        {
            prof.performPartialSwitches(rg, border);
            coordinate.contig = contigNames[w.chrom];
            coordinate.pos = static_cast<unsigned>(w.beginPos);
            return stop(2);
        }
        if (w.beginPos < roi.beginPos)
        {
            ++st.wi;
            continue;
        }
        if (prof.starts[rg[0]].tooBigForNext(static_cast<unsigned>(w.beginPos)))
        {
            if (prof.allEmpty())
            {
                prof.resetTo(static_cast<unsigned>(w.beginPos));
            }
            else
            {
                prof.performPartialSwitches(rg, border);
                coordinate.contig = contigNames[w.chrom];
                coordinate.pos = static_cast<unsigned>(w.beginPos);
                return stop(0);
            }
        }
        else if (prof.starts[rg[0]].needsSwitch(static_cast<unsigned>(w.beginPos)))
        {
            prof.performPartialSwitches(rg, border);
            coordinate.contig = contigNames[w.chrom];
            coordinate.pos = static_cast<unsigned>(w.beginPos);
            return stop(0);
        }
        // add all records of the window, read group by read group (load_profile_popdel_call.h:463-480)
        static_assert(sizeof(PairRecord) == 2 * sizeof(uint32_t), "records are handed over as {pos, dev} words");
        if (oneRg)
        {
            // ... and with it those of the windows behind it that the checks above let pass as well -- they begin inside the
            // region and before the write slot's border, on the same contig, in this 256-bp window or the ones that follow: their
            // records follow in the stream, and filing them is the same sequence of operations. (A 256-bp window whose first and
            // last 30-bp windows pass is taken whole: its windows are in position order.)
            const StartSlots & sst = prof.starts[rg[0]];
            const unsigned right = sst.slots[sst.writeSlot].right;
            auto passes = [&](const FlatWindow & x) {
                return x.beginPos >= roi.beginPos && x.beginPos < roi.endPos && static_cast<unsigned>(x.beginPos) < right;
            };
            uint32_t wj = wi + 1;
            size_t rj = st.rawIdx;
            uint32_t endOfRaw = wEnd;
            int32_t knownChrom = raw.chrom;         // (its name equals coordinate.contig and roi.seqName)
            for (;;)
            {
                while (wj < endOfRaw && passes(in.winsV[wj]))
                    ++wj;
                if (wj < endOfRaw)
                    break;
                // the next 256-bp windows, whole ones first
                bool more = false;
                while (rj + 1 < in.nRawsV)
                {
                    const DecodedRaw & nx = in.rawsV[rj + 1];
                    if (nx.splitFailed || nx.nWin == 0 || nx.firstWin != endOfRaw)
                        break;
                    if (nx.chrom != knownChrom)
                    {
                        if (contigNames[nx.chrom] != coordinate.contig || contigNames[nx.chrom] != roi.seqName)
                            break;
                        knownChrom = nx.chrom;
                    }
                    const FlatWindow & head = in.winsV[nx.firstWin], & tail = in.winsV[nx.firstWin + nx.nWin - 1];
                    ++rj;
                    endOfRaw = nx.firstWin + nx.nWin;
                    if (passes(head) && passes(tail) && head.beginPos <= tail.beginPos)
                    {
                        wj = endOfRaw;
                        continue;
                    }
                    wj = nx.firstWin;
                    more = true;                    // (window by window through this one)
                    break;
                }
                if (!more)
                    break;
            }
            // the run's records: from this window's first to the last of the last window taken (windows wi .. wj - 1 follow each
            // other in the stream, and so do their records)
            const uint32_t firstRec = w.firstRec;
            const uint32_t endRec = in.winsV[wj - 1].firstRec + in.countsV[wj - 1];
            st.haveRun = true;
            st.runFirst = firstRec;
            st.runCount = endRec - firstRec;
            st.runFirstWindow = static_cast<uint32_t>(w.beginPos);
            // (the loop goes on at window wj of 256-bp window rj)
            st.rawIdx = rj;
            st.wi = wj;
            st.inRaw = true;
            return true;
        }
        uint32_t r0 = w.firstRec;
        for (size_t r = 0; r < rg.size(); ++r)
        {
            const uint32_t n = in.countsV[static_cast<size_t>(wi) * rg.size() + r];
            const InsertHistogram & h = p_.histograms[rg[r]];
            prof.addRun(rg[r], reinterpret_cast<const uint32_t *>(in.recsV + r0), n,
                        static_cast<int>(h.median) - 2 * static_cast<int>(h.readLength));
            r0 += n;
        }
        ++st.wi;
    }
    return false;
}

// The whole loop for one sample, every run filed as it is found (batches that are replayed in order: the reference's sequence).
unsigned CallRun::replaySegment(unsigned sample, const DecodedStream & in, const std::vector<std::string> & contigNames, Coord & coordinate,
                                const Region & roi, unsigned * border)
{
    ReplayState st;
    while (replayAdvance(st, in, p_.inputFiles[sample], p_.rgs[sample], contigNames, coordinate, roi, border))
        fileRun(sample, in, st);
    return st.code;
}

// The records of a stretch that was filed, read again from the profile by the host's own reader: those of the 30-bp windows from
// `firstWindow` on (contig `contig`), `skip` of them passed over, the next `n` to dst. What a start slot's virtual runs are
// materialised from (ChromProfile::refill), and what fileRun() falls back to when records it needs had stayed on the device.
void CallRun::readRunFromProfile(unsigned sample, const std::string & contig, uint32_t jumpPos, uint32_t firstWindow, uint64_t skip, uint64_t n,
                                 StartEntry * dst) const
{
    const unsigned contigIdx = p_.representativeContigs ? 0 : sample;
    const std::vector<std::string> & contigNames = p_.contigNames[contigIdx];
    ProfileFile in(p_.inputFiles[sample]);
    // (from where the round's reader had jumped to -- the region's begin then --: the stream is the one it saw)
    in.jumpToRegion(contigNames, p_.contigLengths[contigIdx], p_.indexRegionSizes[contigIdx], contig, static_cast<int>(jumpPos));
    std::unique_ptr<ByteSource> src = in.openSource(p_.uncompressedIn);
    const unsigned nRg = static_cast<unsigned>(p_.rgs[sample].size());
    WindowHead window;
    FlatWindows wins;
    FlatCounts counts;
    FlatRecords recs;
    uint64_t done = 0;
    while (done < n)
    {
        if (!readWindowHead(*src, window))
            break;
        if (contigNames[window.chrom] != contig)
        {
            if (done + skip != 0 || !wins.empty())
                break;                          // (beyond the contig)
            skipWindowBody(*src, nRg);
            continue;
        }
        if (static_cast<int64_t>(window.beginPos) + 255 < static_cast<int64_t>(firstWindow))
        {
            skipWindowBody(*src, nRg);
            continue;
        }
        wins.clear();
        counts.clear();
        recs.clear();
        if (!readWindowBodyFlat(*src, window, nRg, wins, counts, recs, 256, 30))
            break;
        for (size_t w = 0; w < wins.size() && done < n; ++w)
        {
            if (static_cast<int64_t>(wins[w].beginPos) < static_cast<int64_t>(firstWindow))
                continue;
            const uint32_t c = counts[w * nRg];
            for (uint32_t k = 0; k < c && done < n; ++k)
            {
                if (skip != 0)
                {
                    --skip;
                    continue;
                }
                const PairRecord & r = recs[wins[w].firstRec + k];
                dst[done++] = StartEntry{r.pos, static_cast<int32_t>(r.dev)};
            }
        }
    }
    if (done != n)
        throw ProfileError("[PopDel] Internal error: records of \"" + p_.inputFiles[sample] + "\" that had been filed were not found again in the profile.");
}

// One run of a one-read-group file into the tables. With the sample's blocks summarised by the device: a head of records one by
// one (the blocks the run begins in the middle of, and at least kBulkHead records: the end table's counting cursor comes home to the
// write slot), then every whole block whose pairs all end below the write slot's border at once -- their counts added up window
// by window, ChromProfile::addBulk, which checks the rest and never reads a record: where the device kept those blocks' records to
// itself (DecodedStream::recIndex) the start slot holds them as a virtual run --, then what is left record by record.
void CallRun::fileRun(unsigned sample, const DecodedStream & in, const ReplayState & st)
{
    ChromProfile & prof = *profile_;
    const unsigned rg = p_.rgs[sample][0];
    const int inner = innerOffsetOf(sample);
    const size_t n = st.runCount;
    filedRecords_.fetch_add(n, std::memory_order_relaxed);
    const EndSlots & en = prof.ends[rg];
    const StartSlots::Slot & target = prof.starts[rg].slots[prof.starts[rg].writeSlot];
    const size_t heldBefore = target.total;
    // ---- the plan: which whole blocks can be taken from their summaries
    RunSummary sum;
    size_t firstBlock = 0, endBlock = 0;            // blocks [firstBlock, endBlock) (empty ones among them)
    uint32_t lastPos = 0, lastEnd = 0;
    bool endsInOrder = true;
    if (bulkFiling_ && in.blockSums && en.light && n >= kBulkHead + kBulkLeast)
    {
        // the blocks that lie inside [runFirst + kBulkHead, runFirst + n), in record indices of the sample's stream
        const uint64_t base = in.blockRecs[0];
        const uint64_t lo = st.runFirst + kBulkHead, hi = static_cast<uint64_t>(st.runFirst) + n;
        const size_t j0 = static_cast<size_t>(std::lower_bound(in.blockRecs, in.blockRecs + in.nBlocks + 1, base + lo) - in.blockRecs);
        const uint32_t borderBin = (en.slots[en.writeSlot].right + 29u) / 30u;
        constexpr uint32_t kBad = RunSummary::kSaturated | RunSummary::kNoRoom;
        size_t j1 = j0;
        for (; j1 < in.nBlocks && in.blockRecs[j1 + 1] - base <= hi; ++j1)
        {
            const pd_block_summary & b = in.blockSums[j1];
            if (b.n == 0)
            {
                if (in.blockRecs[j1 + 1] != in.blockRecs[j1])
                    break;                      // (records the device did not summarise)
                continue;
            }
            if ((b.flags & kBad) != 0 || !(b.flags & RunSummary::kStartsInOrder) || b.max_end_bin >= borderBin || b.n != in.blockRecs[j1 + 1] - in.blockRecs[j1])
                break;
            if (sum.n == 0)
            {
                sum.firstPos = b.first_pos;
                sum.binBase = b.bin_base;
                sum.firstEndBin = b.first_end_bin;
                sum.minEndBin = b.min_end_bin;
                sum.maxEndBin = b.max_end_bin;
            }
            else
            {
                if (b.first_pos < lastPos || b.bin_base < sum.binBase)
                    break;                      // (positions step back between two blocks)
                endsInOrder = endsInOrder && b.first_end_bin >= lastEnd;
            }
            if (std::max(b.max_end_bin, sum.maxEndBin) - sum.binBase >= bulkRoom_)
                break;
            sum.n += b.n;
            sum.minEndBin = std::min(sum.minEndBin, b.min_end_bin);
            sum.maxEndBin = std::max(sum.maxEndBin, b.max_end_bin);
            endsInOrder = endsInOrder && (b.flags & RunSummary::kEndsInOrder) != 0;
            lastPos = b.last_pos;
            lastEnd = b.last_end_bin;
        }
        static const bool traceFiling = getenv("POPDEL_FILING_TRACE") != nullptr;
        if (traceFiling && sample < 2)
        {
            std::ostringstream msg;
            msg << "[popdel filing] sample " << sample << ": run of " << n << " records from " << st.runFirst << ", blocks " << in.nBlocks << ", first whole block " << j0
                << ", taken up to " << j1 << " (" << sum.n << " records), border window " << borderBin;
            if (j1 < in.nBlocks)
            {
                const pd_block_summary & b = in.blockSums[j1];
                msg << "; block " << j1 << ": n " << b.n << " records " << in.blockRecs[j1] - base << ".." << in.blockRecs[j1 + 1] - base << " flags " << b.flags
                    << " first " << b.first_pos << " last " << b.last_pos << " bin base " << b.bin_base << " end windows " << b.min_end_bin << ".." << b.max_end_bin;
            }
            std::cerr << msg.str() << std::endl;
        }
        if (sum.n >= kBulkLeast)
        {
            firstBlock = j0;
            while (in.blockSums[firstBlock].n == 0)
                ++firstBlock;
            endBlock = j1;
            while (in.blockSums[endBlock - 1].n == 0)
                --endBlock;
        }
        else
        {
            sum = RunSummary();
            bulkNone_.fetch_add(1, std::memory_order_relaxed);
        }
    }
    // ---- the records: [0, headEnd) and [bulkEnd, n) of the run are filed one by one; those in between only if the tables refuse
    const size_t headEnd = sum.n ? static_cast<size_t>(in.blockRecs[firstBlock] - in.blockRecs[0]) - st.runFirst : n;
    const size_t bulkEnd = headEnd + sum.n;
    // (where the device kept the middle blocks' records to itself the start slot notes a virtual run: the window loop takes the
    // entries from the device's copy, pd_walk_source)
    const bool keptOnDevice = in.recIndex != nullptr;
    const uint32_t * head = headEnd ? reinterpret_cast<const uint32_t *>(in.records(st.runFirst, static_cast<uint32_t>(headEnd))) : nullptr;
    const uint32_t * tail = bulkEnd < n ? reinterpret_cast<const uint32_t *>(in.records(st.runFirst + static_cast<uint32_t>(bulkEnd), static_cast<uint32_t>(n - bulkEnd))) : nullptr;
    const uint32_t * middle = sum.n && !keptOnDevice ? reinterpret_cast<const uint32_t *>(in.records(st.runFirst + static_cast<uint32_t>(headEnd), sum.n)) : nullptr;
    thread_local HugeVector<StartEntry> again;
    // (the run's first 30-bp window, for whoever reads it again from the profile)
    const uint32_t firstWindow = st.runFirstWindow;
    auto readAgain = [&]() {
        // (records this needs had stayed on the device: the whole run, from the profile)
        again.resize(n);
        readRunFromProfile(sample, nextCandidate_[sample].contig, static_cast<uint32_t>(std::max(0, p_.allRois[p_.nextRoi].beginPos)), firstWindow, 0, n,
                           again.data());
        runsReadAgain_.fetch_add(1, std::memory_order_relaxed);
        const uint32_t * all = reinterpret_cast<const uint32_t *>(again.data());
        head = all;
        tail = all + 2 * bulkEnd;
        middle = all + 2 * headEnd;
    };
    if ((headEnd && !head) || (bulkEnd < n && !tail))
        readAgain();
    size_t done = 0;
    if (sum.n)
    {
        prof.reserveStarts(rg, keptOnDevice && !middle ? n - sum.n : n);
        prof.addRun(rg, head, headEnd, inner);
        done = headEnd;
        sum.lastPos = lastPos;
        sum.lastEndBin = lastEnd;
        sum.nBins = sum.maxEndBin - sum.binBase + 1u;
        sum.flags = RunSummary::kStartsInOrder | (endsInOrder ? RunSummary::kEndsInOrder : 0u);
        // the blocks' counts added up into one row of starts and one of ends (per thread: kept from run to run)
        thread_local std::vector<uint16_t> rows;
        rows.assign(2 * static_cast<size_t>(sum.nBins), 0);
        uint16_t * starts = rows.data(), * ends = rows.data() + sum.nBins;
        const size_t stride = static_cast<size_t>(in.startsRoom) + in.endsRoom;
        for (size_t j = firstBlock; j < endBlock; ++j)
        {
            const pd_block_summary & b = in.blockSums[j];
            if (b.n == 0)
                continue;
            const uint8_t * cs = in.blockCounts + j * stride, * ce = cs + in.startsRoom;
            const uint32_t off = b.bin_base - sum.binBase;
            const uint32_t nStarts = b.last_pos / 30u - b.bin_base + 1u;
            for (uint32_t w = 0; w < nStarts; ++w)
                starts[off + w] = static_cast<uint16_t>(starts[off + w] + cs[w]);
            for (uint32_t w = b.min_end_bin - b.bin_base; w <= b.max_end_bin - b.bin_base; ++w)
                ends[off + w] = static_cast<uint16_t>(ends[off + w] + ce[w]);
        }
        VirtualNote note;
        note.firstWindow = firstWindow;
        note.skip = static_cast<uint32_t>(headEnd);
        note.jumpPos = static_cast<uint32_t>(std::max(0, p_.allRois[p_.nextRoi].beginPos));
        note.tag = roundContigTag_;
        const bool virtualRun = !middle && nextCandidate_[sample].contig == roundContig_;
        if (!middle && !virtualRun)
            readAgain();
        if (prof.addBulk(rg, virtualRun ? nullptr : middle, sum, starts, ends, inner, note))
        {
            done += sum.n;
            bulkRuns_.fetch_add(1, std::memory_order_relaxed);
            bulkRecords_.fetch_add(sum.n, std::memory_order_relaxed);
            if (virtualRun)
                virtualRecords_.fetch_add(sum.n, std::memory_order_relaxed);
        }
        else
        {
            bulkRefused_.fetch_add(1, std::memory_order_relaxed);
            if (!middle)
                readAgain();
            prof.addRun(rg, middle, sum.n, inner);
            done += sum.n;
        }
        prof.addRun(rg, tail, n - done, inner);
    }
    else
    {
        prof.reserveRun(rg, n);
        prof.addRun(rg, head, n, inner);
    }
    // (tests: POPDEL_TEST_MATERIALIZE=k reads every k-th sample's virtual runs back from the profile right away)
    static const unsigned materializeEvery = [] { const char * e = getenv("POPDEL_TEST_MATERIALIZE"); return e ? static_cast<unsigned>(atoi(e)) : 0u; }();
    if (materializeEvery && sample % materializeEvery == 0 && target.isVirtual())
        prof.materializeStarts(rg, prof.starts[rg].writeSlot);
    // (the slot holds exactly this run: the device has the same entries in the loader's records)
    if (in.inflater && heldBefore == 0 && target.total == n)
    {
        slotSource_[rg] = pd_walk_source{in.inflater, in.recBase + st.runFirst, in.runId};
        slotSourceCount_[rg] = static_cast<uint32_t>(n);
    }
    else
    {
        slotSource_[rg].inflater = nullptr;
        if (target.isVirtual())             // (nothing the window loop could take from the device: the entries, then)
            prof.materializeStarts(rg, prof.starts[rg].writeSlot);
    }
}

bool CallRun::processSegmentCode(unsigned i, unsigned code)
{
    if (code == 0u)
    {
        if (nextCandidate_[i].pos < nextReadPos_)
            nextReadPos_ = nextCandidate_[i].pos;
        return false;
    }
    if (code < 3u)
    {
        finishedRois_[i] = 1;
        if (windows_.nextChrom.contig.empty() || lowerCoord(nextCandidate_[i], windows_.nextChrom, p_))
            windows_.nextChrom = nextCandidate_[i];
        return true;
    }
    finishedRois_[i] = 1;
    finishedFiles_[i] = 1;
    --p_.fileCount;
    return true;
}

void CallRun::markRoiComplete(unsigned i)
{
    finishedRois_[i] = 1;
    if (!finishedFiles_[i])
    {
        finishedFiles_[i] = 1;
        --p_.fileCount;
    }
}

// unifyCalls() with its per-sample half on the device: the scalar grouping runs here on call-level fields, every context
// merges the columns of its own samples (pd_calls_merge) and the merged columns are joined. Falls back to the host merge
// for a segment the device declines (a LAD / DAD value beyond its median bins). Returns what unifyCalls() returns.
bool CallRun::mergeCallsOnDevice(std::vector<WindowCall> & calls, const std::vector<pd_calls *> & parts)
{
    const uint32_t S = shardSample_.back();
    std::vector<WindowCall> backup;
    std::vector<MergeGroup> groups;
    size_t firstKept = 0;
    backup = calls;                                             // call-level fields only: cheap; needed if the device declines
    if (!unifyCallsScalar(calls, p_.meanStddev, p_.minRelWinCover, p_.outputFailed, groups, firstKept))
        return false;
    std::vector<uint32_t> offsets(1, 0), members, heads;
    for (const MergeGroup & g : groups)
    {
        heads.push_back(calls[g.head].sourceIndex);
        for (size_t m : g.members)
            members.push_back(calls[m].sourceIndex);
        offsets.push_back(static_cast<uint32_t>(members.size()));
    }
    // A call can survive the grouping without having been merged (the head of a group whose position list ran empty,
    // utils_popdel.h:621-640): it keeps its own window's columns, which are not here -- such a segment goes to the host.
    bool declined = false;
    {
        std::vector<char> isHead(calls.size(), 0);
        for (const MergeGroup & g : groups)
            isHead[g.head] = 1;
        for (size_t i = firstKept; i < calls.size(); ++i)
            declined = declined || (calls[i].filter != kFilterInvalid && !isHead[i]);
    }
    std::vector<pd_merged *> out(ctxs_.size(), nullptr);
    std::vector<int> codes(ctxs_.size(), 0);
    // (not collective: every context works on its own columns; one after the other is as good as threads here)
    for (size_t r = 0; r < ctxs_.size() && !declined; ++r)
        codes[r] = pd_calls_merge(ctxs_[r], parts[r], static_cast<uint32_t>(groups.size()), offsets.data(), members.data(), heads.data(), &out[r]);
    for (size_t r = 0; r < ctxs_.size(); ++r)
    {
        if (codes[r] == 2)
            declined = true;
        else if (codes[r] != 0)
            throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
    }
    if (declined)
    {
        for (pd_merged * m : out)
            pd_merged_free(m);
        // the whole segment on the host: fetch the columns, then the ordinary merge
        calls = backup;
        for (size_t r = 0; r < parts.size(); ++r)
            if (pd_calls_fetch_columns(ctxs_[r], parts[r]) != 0)
                throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
        parallelFor(calls.size(), [&](size_t c) {
            WindowCall & w = calls[c];
            w.pl.resize(S);
            w.lad.resize(S);
            w.dad.resize(S);
            w.firstLast.resize(S);
            for (size_t r = 0; r < parts.size(); ++r)
            {
                const pd_calls * q = parts[r];
                const uint32_t s0 = shardSample_[r], n = q->n_samples;
                const size_t base = static_cast<size_t>(c) * n;
                for (uint32_t j = 0; j < n; ++j)
                {
                    for (int k = 0; k < 3; ++k)
                    {
                        w.pl[s0 + j][k] = q->pl[(base + j) * 3 + k];
                        w.lad[s0 + j][k] = q->lad[(base + j) * 3 + k];
                    }
                    for (int k = 0; k < 5; ++k)
                        w.dad[s0 + j][k] = q->dad[(base + j) * 5 + k];
                    w.firstLast[s0 + j][0] = q->first_last[(base + j) * 2];
                    w.firstLast[s0 + j][1] = q->first_last[(base + j) * 2 + 1];
                }
            }
        });
        return unifyCalls(calls, p_.meanStddev, p_.minRelWinCover, p_.outputFailed);
    }
    // join the merged columns of the contexts; allele frequency from the allele counts (reference utils_popdel.h:339-359)
    parallelFor(groups.size(), [&](size_t g) {
        WindowCall & head = calls[groups[g].head];
        head.pl.resize(S);
        head.lad.resize(S);
        head.dad.resize(S);
        head.firstLast.resize(S);
        unsigned alleles = 0;
        for (size_t r = 0; r < out.size(); ++r)
        {
            const pd_merged * m = out[r];
            const uint32_t s0 = shardSample_[r], n = m->n_samples;
            const size_t base = g * static_cast<size_t>(n);
            for (uint32_t j = 0; j < n; ++j)
            {
                for (int k = 0; k < 3; ++k)
                {
                    head.pl[s0 + j][k] = m->pl[(base + j) * 3 + k];
                    head.lad[s0 + j][k] = m->lad[(base + j) * 3 + k];
                }
                for (int k = 0; k < 5; ++k)
                    head.dad[s0 + j][k] = m->dad[(base + j) * 5 + k];
                head.firstLast[s0 + j][0] = m->first_last[(base + j) * 2];
                head.firstLast[s0 + j][1] = m->first_last[(base + j) * 2 + 1];
            }
            alleles += m->alleles[g];
        }
        head.frequency = static_cast<double>(alleles) / (static_cast<size_t>(S) * 2);
    });
    for (pd_merged * m : out)
        pd_merged_free(m);
    keepValidCalls(calls, firstKept);
    return true;
}

// Collect the calls of the segment that is in flight, merge them and write the records.
void CallRun::flushPending()
{
    drainHandOver();
    if (!pending_)
        return;
    pending_ = false;
    collectAndWrite(tickets_, pendingContig_);
}

// Results of one segment: collected from every context, merged, written.
void CallRun::collectAndWrite(const std::vector<uint64_t> & tickets, const std::string & contig)
{
    std::vector<pd_calls *> parts(ctxs_.size(), nullptr);
    onEveryContext([&](unsigned r) {
        if (pd_segment_collect(ctxs_[r], tickets[r], &parts[r]) != 0)
            throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
    });
    {
        pd_timing t;
        if (pd_last_timing(ctxs_[0], &t) == 0)
        {
            std::lock_guard<std::mutex> guard(deviceNotesLock_);
            deviceMs_.push_back(t.stage1_ms + t.stage2_ms);
            nearIntegerShifts_ += t.n_near_integer_shifts;
            candidateCells_ += t.n_candidate_cells;
            windowCalls_ += parts[0]->n_calls;
        }
    }
    // every context returns the calls of the whole cohort with the columns of its own samples
    pd_calls * pc = parts[0];
    for (pd_calls * q : parts)
        if (q->n_calls != pc->n_calls)
            throw ProfileError("[PopDel] Internal error: the contexts disagree on the calls of a segment.");
    std::vector<WindowCall> calls(pc->n_calls);
    const uint32_t S = shardSample_.back();
    // columns left on the device (every context decides the same way: the rule is the number of stage-2 batches)
    bool columnsOnDevice = mergeOnDevice_ && pc->n_calls != 0;
    for (pd_calls * q : parts)
        columnsOnDevice = columnsOnDevice && q->pl == nullptr;
    if (!columnsOnDevice)
        for (size_t r = 0; r < parts.size(); ++r)
            if (parts[r]->n_calls != 0 && parts[r]->pl == nullptr && pd_calls_fetch_columns(ctxs_[r], parts[r]) != 0)
                throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
    parallelFor(pc->n_calls, [&](size_t c) {
        WindowCall & w = calls[c];
        w.sourceIndex = static_cast<uint32_t>(c);
        w.initialLength = pc->initial_length[c];
        w.iterations = pc->iterations[c];
        w.deletionLength = pc->deletion_length[c];
        w.likelihoodRatio = pc->likelihood_ratio[c];
        w.frequency = pc->frequency[c];
        w.windowPosition = pc->window_position[c];
        w.position = pc->position[c];
        w.endPosition = pc->end_position[c];
        w.filter = pc->filter[c];
        if (columnsOnDevice)
            return;
        w.pl.resize(S);
        w.lad.resize(S);
        w.dad.resize(S);
        w.firstLast.resize(S);
        for (size_t r = 0; r < parts.size(); ++r)
        {
            const pd_calls * q = parts[r];
            const uint32_t s0 = shardSample_[r], n = q->n_samples;
            const size_t base = static_cast<size_t>(c) * n;
            for (uint32_t j = 0; j < n; ++j)
            {
                const uint32_t s = s0 + j;
                for (int k = 0; k < 3; ++k)
                {
                    w.pl[s][k] = q->pl[(base + j) * 3 + k];
                    w.lad[s][k] = q->lad[(base + j) * 3 + k];
                }
                for (int k = 0; k < 5; ++k)
                    w.dad[s][k] = q->dad[(base + j) * 5 + k];
                w.firstLast[s][0] = q->first_last[(base + j) * 2];
                w.firstLast[s][1] = q->first_last[(base + j) * 2 + 1];
            }
        }
    });
    bool merged = false, mergedAny = false;
    if (columnsOnDevice)
    {
        merged = true;
        mergedAny = mergeCallsOnDevice(calls, parts);
    }
    for (pd_calls * q : parts)
        pd_calls_free(q);
    if (observer_ && observer_->onWindowCalls)
        observer_->onWindowCalls(calls);
    if (!p_.windowWiseOutput)
    {
        if (merged ? mergedAny : unifyCalls(calls, p_.meanStddev, p_.minRelWinCover, p_.outputFailed))
        {
            vcf_->appendContigName(contig);
            vcf_->writeCalls(calls, p_);
            vcf_->flush();
        }
    }
    else
    {
        if (!calls.empty())
            vcf_->appendContigName(contig);
        vcf_->writeCalls(calls, p_);
        vcf_->flush();
    }
}

// Walk the windows of the loaded slot, record which pairs are active in which window, and submit that to the
// backend. The previous segment is collected and written first so that records stay in order.
void CallRun::processSegment()
{
    ChromProfile & prof = *profile_;
    if (deviceWalk_)
    {
        // (tests: POPDEL_TEST_LEAVE_WALK_AT=k hands the window loop back to the host at the k-th round -- the pairs the device
        // holds as active, the entries it kept to itself and the end tables' deferred runs all come back, and the rest of the run
        // is walked here: same output)
        static const int leaveAt = [] { const char * e = getenv("POPDEL_TEST_LEAVE_WALK_AT"); return e ? atoi(e) : 0; }();
        if (prof.canPlanSegment(p_.windowShift) && !(leaveAt > 0 && rounds_.size() == static_cast<size_t>(leaveAt)))
        {
            double t0 = PhaseClock::now();
            std::vector<uint8_t> equalRun, switches;
            const WalkShape shape = prof.planSegment(p_.windowShift, equalRun, switches);
            windows_.current.pos += shape.nWindows * p_.windowShift;
            if (!rounds_.empty())
                rounds_.back().windows = shape.nWindows;
            clock_.walk += PhaseClock::now() - t0;
            gTrace.mark("segment: planned");
            t0 = PhaseClock::now();
            if (pending_)                           // (a segment of the host's window loop is still out)
                flushPending();
            waitForHandOver(lastDirectJob_);        // jobs that point into the tables: handed over before the next one is cut
            gTrace.mark("segment: the previous one has been handed over");
            submitPlannedSegment(shape, equalRun, switches);
            gTrace.mark("segment: queued");
            clock_.submit += PhaseClock::now() - t0;
            ++deviceSegments_;
            return;
        }
        leaveDeviceWalk();
    }
    ensureBackend();
    ++hostSegments_;
    // (check mode: plan the segment on the tables as they are now, note what the plan says, put the state back, then walk)
    WalkShape planned;
    std::vector<uint8_t> plannedRun, plannedSwitches;
    std::vector<std::pair<size_t, size_t>> plannedCursors;
    bool havePlan = false;
    if (walkCheck_)
    {
        if (prof.canPlanSegment(p_.windowShift))
        {
            ChromProfile::PlanState saved = prof.savePlanState();
            planned = prof.planSegment(p_.windowShift, plannedRun, plannedSwitches);
            for (size_t rg = 0; rg < prof.starts.size(); ++rg)
                plannedCursors.emplace_back(prof.starts[rg].readIdx, prof.ends[rg].readIdx);
            prof.restorePlanState(saved);
            havePlan = true;
        }
        else
        {
            walkCheck_.reset();             // the reference's own state would fall out of step from here on
            ++walkUnchecked_;
        }
    }
    else if (walkUnchecked_)
        ++walkUnchecked_;
    prof.beginSegmentLog();
    if (prof.profilesAtEnd)
        prof.profilesAtEnd = !prof.nextWindow(p_.windowShift);
    prof.initializeActive();
    const unsigned firstPos = prof.currentPos;
    // the window loop: window (firstPos + k * shift) is genotyped on the device, read groups are walked in parallel
    double t0 = PhaseClock::now();
    const unsigned nWindows = prof.walkSegment(p_.windowShift);
    windows_.current.pos += nWindows * p_.windowShift;
    if (!rounds_.empty())
        rounds_.back().windows = nWindows;
    clock_.walk += PhaseClock::now() - t0;
    std::vector<std::vector<PairInstance>> & log = prof.endSegmentLog();
    prof.noteSegmentWalked();
    {
        static const bool traceActive = getenv("POPDEL_ACTIVE_TRACE") != nullptr;
        if (traceActive)
        {
            size_t left = 0, inst = 0;
            for (const auto & a : prof.active)
                left += a.size();
            size_t carried = 0, openEnd = 0;
            uint32_t minStart = 0xffffffffu, maxStart = 0;
            for (const auto & v : log)
            {
                inst += v.size();
                for (const PairInstance & pi : v)
                {
                    carried += pi.on == 0;
                    openEnd += pi.off >= nWindows;
                    minStart = std::min(minStart, pi.start);
                    maxStart = std::max(maxStart, pi.start);
                }
            }
            std::cerr << "[popdel active]   instances from window 0: " << carried << ", to the last window: " << openEnd << ", pair starts " << minStart << ".." << maxStart
                      << "; position after the walk " << prof.currentPos << ", right border " << prof.currentRightBorder << "; start slots read "
                      << prof.starts[0].readSlot << " write " << prof.starts[0].writeSlot << " sizes " << prof.starts[0].slots[0].total << " " << prof.starts[0].slots[1].total
                      << " " << prof.starts[0].slots[2].total << " id offsets " << prof.starts[0].slots[0].idOffset << " " << prof.starts[0].slots[1].idOffset << " "
                      << prof.starts[0].slots[2].idOffset << std::endl;
            std::cerr << "[popdel active] segment walked on the host: " << windows_.current.contig << ":" << firstPos << ", " << nWindows << " windows, " << inst
                      << " pair instances, " << left << " pairs active at its end; read groups' cursors: starts " << prof.starts[0].readIdx << "/"
                      << prof.starts[0].slots[prof.starts[0].readSlot].total << " ends " << prof.ends[0].readIdx << "/"
                      << prof.ends[0].slots[prof.ends[0].readSlot].total << " (end slots read " << prof.ends[0].readSlot << " write " << prof.ends[0].writeSlot
                      << ", sizes " << prof.ends[0].slots[0].total << " " << prof.ends[0].slots[1].total << " " << prof.ends[0].slots[2].total << ")" << std::endl;
        }
    }
    if (havePlan)
        checkWalkAgainstPlan(planned, plannedRun, plannedSwitches, plannedCursors, firstPos, nWindows, log);
    if (observer_ && observer_->onSegment)
        observer_->onSegment(firstPos, nWindows, log);
    if (observer_ && observer_->onActiveOps && nWindows != 0)
    {
        // (the mutations of a segment without windows stay in the log: they precede window 0 of the next one)
        observer_->onActiveOps(prof.activeOps(), log);
        prof.clearActiveOps();
    }

    t0 = PhaseClock::now();
    flushPending();
    clock_.collect += PhaseClock::now() - t0;
    if (nWindows == 0)
        return;
    t0 = PhaseClock::now();
    // Flatten to the SoA layout of pd_segment.
    // (buffers are members: their pages stay mapped from segment to segment)
    std::vector<uint64_t> & offsets = flatOffsets_;
    offsets.assign(log.size() + 1, 0);
    for (size_t rg = 0; rg < log.size(); ++rg)
        offsets[rg + 1] = offsets[rg] + log[rg].size();
    const size_t total = offsets.back();
    std::vector<int32_t> & dev = flatDev_;
    std::vector<uint32_t> &start = flatStart_, &on = flatOn_, &off = flatOff_;
    if (dev.size() < total)
    {
        dev.resize(total);
        start.resize(total);
        on.resize(total);
        off.resize(total);
    }
    parallelFor(log.size(), [&](size_t rg) {
        size_t o = offsets[rg];
        for (const PairInstance & pi : log[rg])
        {
            dev[o] = pi.dev;
            start[o] = pi.start;
            on[o] = pi.on;
            off[o] = pi.off;
            ++o;
        }
    }, 16);
    clock_.flatten += PhaseClock::now() - t0;
    t0 = PhaseClock::now();
    onEveryContext([&](unsigned r) {
        // the read groups of this context's samples: a slice of the arrays (they are read-group major)
        const uint32_t g0 = shardRg_[r], g1 = shardRg_[r + 1];
        std::vector<uint64_t> local(offsets.begin() + g0, offsets.begin() + g1 + 1);
        const uint64_t base = local.front();
        for (uint64_t & o : local)
            o -= base;
        pd_segment seg;
        seg.first_window_pos = firstPos;
        seg.n_windows = nWindows;
        seg.window_shift = p_.windowShift;
        seg.n_readgroups = g1 - g0;
        seg.rg_offsets = local.data();
        seg.pair_dev = dev.data() + base;
        seg.pair_start = start.data() + base;
        seg.pair_on = on.data() + base;
        seg.pair_off = off.data() + base;
        if (pd_segment_submit(ctxs_[r], &seg, &tickets_[r]) != 0)
            throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
    });
    pending_ = true;
    pendingContig_ = windows_.current.contig;
    clock_.submit += PhaseClock::now() - t0;
}

// The segment as the tables hold it, to every context: the entries of the read slots of the read groups whose slots were
// switched (the tables themselves -- no copy here), the borders their end records were filed under, the plan.
void CallRun::submitPlannedSegment(const WalkShape & shape, const std::vector<uint8_t> & equalRun, const std::vector<uint8_t> & switches)
{
    ChromProfile & prof = *profile_;
    const size_t nRg = prof.starts.size();
    std::vector<uint32_t> counts(nRg), borders(nRg), idOffsets(nRg);
    std::vector<const pd_walk_entry *> tables(nRg);
    std::vector<pd_walk_source> sources(nRg, pd_walk_source{nullptr, 0, 0});
    bool anySource = false;
    static_assert(sizeof(StartEntry) == sizeof(pd_walk_entry), "the start tables are handed over as they are");
    const bool direct = backendReady_.load() && !getenv("POPDEL_COPY_HAND_OVER");
    std::vector<unsigned> toMaterialize;
    for (size_t rg = 0; rg < nRg; ++rg)
    {
        const StartSlots & st = prof.starts[rg];
        const StartSlots::Slot & slot = st.slots[st.readSlot];
        counts[rg] = static_cast<uint32_t>(slot.total);
        if (direct && switches[rg] != 0 && slotSource_[rg].inflater && slotSourceCount_[rg] == counts[rg] && counts[rg] != 0 && gunzip_)
        {
            sources[rg] = slotSource_[rg];
            anySource = true;
        }
        else if (slot.isVirtual() && (switches[rg] != 0 || !direct))
            toMaterialize.push_back(static_cast<unsigned>(rg));      // (the window loop will want the entries themselves)
        slotSource_[rg].inflater = nullptr;         // (the slot it described is the read slot now)
    }
    if (!toMaterialize.empty())
        parallelFor(toMaterialize.size(), [&](size_t k) { prof.materializeStarts(toMaterialize[k], prof.starts[toMaterialize[k]].readSlot); }, 4);
    for (size_t rg = 0; rg < nRg; ++rg)
    {
        const StartSlots & st = prof.starts[rg];
        const StartSlots::Slot & slot = st.slots[st.readSlot];
        // (a slot with virtual runs has no table: its entries come from the loader's records on the device, sources[rg])
        tables[rg] = slot.isVirtual() ? nullptr : reinterpret_cast<const pd_walk_entry *>(slot.entries.data());
        borders[rg] = slot.endBorder;
        idOffsets[rg] = slot.idOffset;
    }
    auto job = std::make_shared<HandOverJob>();
    job->equalRun = equalRun;
    job->switches = switches;
    job->shape = shape;
    job->contig = windows_.current.contig;
    if (!direct)
    {
        // the device is not there yet: the job takes the entries along, and the tables are free again
        std::vector<size_t> offsets(nRg + 1, 0);
        for (size_t rg = 0; rg < nRg; ++rg)
            offsets[rg + 1] = offsets[rg] + (switches[rg] != 0 ? counts[rg] : 0u);
        job->copy.resize(offsets[nRg]);
        pd_walk_entry * base = job->copy.data();
        parallelFor(nRg, [&](size_t rg) {
            const size_t n = offsets[rg + 1] - offsets[rg];
            if (n)
                std::memcpy(base + offsets[rg], tables[rg], n * sizeof(pd_walk_entry));
            tables[rg] = base + offsets[rg];
        }, 16);
    }
    job->counts = std::move(counts);
    job->borders = std::move(borders);
    job->idOffsets = std::move(idOffsets);
    job->tables = std::move(tables);
    if (anySource && job->copy.empty())
        job->sources = std::move(sources);
    {
        std::unique_lock<std::mutex> lock(handOverLock_);
        if (handOverError_)
        {
            std::exception_ptr e = handOverError_;
            handOverError_ = nullptr;
            std::rethrow_exception(e);
        }
        // (at most four segments' worth of copied entries wait for the device)
        handOverDone_.wait(lock, [this] { return handOverEnqueued_ - handOverFinished_ < 4 || handOverError_; });
        if (handOverError_)                 // (the worker failed while this thread waited: nothing more is queued behind it)
        {
            std::exception_ptr e = handOverError_;
            handOverError_ = nullptr;
            std::rethrow_exception(e);
        }
        job->seq = ++handOverEnqueued_;
        if (job->copy.empty())
            lastDirectJob_ = job->seq;
        if (!job->sources.empty() && gunzip_)
            for (size_t k = 0; k < kLoaderSets; ++k)
                if (gunzip_ == gunzipSet_[k].get())
                    loaderJob_[k] = job->seq;
        handOverQueue_.push_back(job);
        handOverIdle_ = false;
        if (!handOverThread_.joinable())
            handOverThread_ = std::thread([this] { handOverLoop(); });
    }
    handOverWake_.notify_all();
}

void CallRun::handOverLoop()
{
    for (;;)
    {
        std::shared_ptr<HandOverJob> job;
        {
            std::unique_lock<std::mutex> lock(handOverLock_);
            handOverWake_.wait(lock, [this] { return handOverStop_ || !handOverQueue_.empty(); });
            if (handOverQueue_.empty())
                return;                             // (stop: nothing left to do)
            job = handOverQueue_.front();
            handOverQueue_.pop_front();
            if (handOverError_)
            {
                // an earlier job failed and the main thread has not picked the error up yet: the device's walk state missed a
                // segment, nothing behind it may run (or be written)
                handOverFinished_ = std::max(handOverFinished_, job->seq);
                if (handOverQueue_.empty())
                    handOverIdle_ = true;
                handOverDone_.notify_all();
                continue;
            }
        }
        bool failed = false;
        try
        {
            handOver(*job);
        }
        catch (...)
        {
            std::lock_guard<std::mutex> guard(handOverLock_);
            if (!handOverError_)
                handOverError_ = std::current_exception();
            failed = true;
        }
        bool last;
        {
            std::lock_guard<std::mutex> guard(handOverLock_);
            handOverFinished_ = job->seq;
            if (failed)
            {
                // nothing behind a failed job is handed over: the main thread finds the error at its next call
                handOverFinished_ = handOverEnqueued_;
                handOverQueue_.clear();
                workerPending_ = false;
                handOverIdle_ = true;
            }
            last = handOverQueue_.empty();
        }
        handOverDone_.notify_all();
        if (!failed)
            startLoaderAheadFromOutside();          // (the loader whose records this segment pointed into may go on to the round to come)
        if (last && !failed)
        {
            // nothing queued behind it: its results are collected and written right away (the end of the run, a change of
            // mode or a main thread that waits for the output need them; with more jobs queued the next hand-over does it)
            std::unique_lock<std::mutex> lock(handOverLock_);
            if (handOverQueue_.empty())
            {
                lock.unlock();
                try
                {
                    if (workerPending_)
                    {
                        const double t0 = PhaseClock::now();
                        workerPending_ = false;
                        collectAndWrite(workerTickets_, workerContig_);
                        workerCollectSeconds_ += PhaseClock::now() - t0;
                        gTrace.mark("segment: collected and written");
                    }
                }
                catch (...)
                {
                    std::lock_guard<std::mutex> guard(handOverLock_);
                    if (!handOverError_)
                        handOverError_ = std::current_exception();
                }
                lock.lock();
                if (handOverQueue_.empty())
                    handOverIdle_ = true;
            }
        }
        handOverDone_.notify_all();
    }
}

// One job on the hand-over thread: the previous segment's results first (records stay in order), then this segment.
void CallRun::handOver(HandOverJob & job)
{
    ensureBackend();
    if (workerPending_)
    {
        const double t0 = PhaseClock::now();
        workerPending_ = false;
        collectAndWrite(workerTickets_, workerContig_);
        workerCollectSeconds_ += PhaseClock::now() - t0;
        gTrace.mark("segment: previous collected and written");
    }
    const double t0 = PhaseClock::now();
    workerTickets_.assign(ctxs_.size(), 0);
    {
        // the table pool's regions (they live as long as the process): the backend may page-lock them and fetch the entries
        // from the tables themselves instead of having the host threads copy them into a page-locked array first
        const auto regions = HugePool::instance().regions();
        if (regions.size() != regionsOffered_)
        {
            std::vector<const void *> begin;
            std::vector<uint64_t> bytes;
            for (size_t k = regionsOffered_; k < regions.size(); ++k)
            {
                begin.push_back(regions[k].first);
                bytes.push_back(regions[k].second);
            }
            (void)pd_host_regions_offer(begin.data(), bytes.data(), static_cast<uint32_t>(begin.size()));
            regionsOffered_ = regions.size();
        }
    }
    onEveryContext([&](unsigned r) {
        const uint32_t g0 = shardRg_[r], g1 = shardRg_[r + 1];
        pd_walk wk;
        std::memset(&wk, 0, sizeof(wk));
        wk.first_window_pos = job.shape.firstPos;
        wk.step_pos = job.shape.stepPos;
        wk.has_step = job.shape.hasStep ? 1u : 0u;
        wk.n_windows = job.shape.nWindows;
        wk.window_shift = p_.windowShift;
        wk.n_readgroups = g1 - g0;
        wk.rg_counts = job.counts.data() + g0;
        wk.rg_entries = job.tables.data() + g0;
        wk.end_border = job.borders.data() + g0;
        wk.id_offset = job.idOffsets.data() + g0;
        wk.switches = job.switches.data() + g0;
        wk.equal_run = job.equalRun.data() + g0;
        wk.rg_sources = job.sources.empty() ? nullptr : job.sources.data() + g0;
        if (pd_walk_submit(ctxs_[r], &wk, &workerTickets_[r]) != 0)
            throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
    });
    workerPending_ = true;
    workerContig_ = job.contig;
    workerSubmitSeconds_ += PhaseClock::now() - t0;
    gTrace.mark("segment: on the device");
}

void CallRun::waitForHandOver(uint64_t seq)
{
    std::unique_lock<std::mutex> lock(handOverLock_);
    handOverDone_.wait(lock, [this, seq] { return handOverFinished_ >= seq || handOverError_; });
    if (handOverError_)
    {
        std::exception_ptr e = handOverError_;
        handOverError_ = nullptr;
        std::rethrow_exception(e);
    }
}

void CallRun::drainHandOver()
{
    std::unique_lock<std::mutex> lock(handOverLock_);
    handOverDone_.wait(lock, [this] { return handOverIdle_ || handOverError_; });
    if (handOverError_)
    {
        std::exception_ptr e = handOverError_;
        handOverError_ = nullptr;
        std::rethrow_exception(e);
    }
}

// From the device's window loop back to the one here: the pairs the device holds as active become the tables' active
// sets (the cursors have been kept where the walk would have left them all along).
void CallRun::leaveDeviceWalk(bool tablesAboutToGo)
{
    ensureBackend();
    drainHandOver();
    ChromProfile & prof = *profile_;
    if (!tablesAboutToGo)
    {
        prof.materializeAllStarts();    // (entries the device kept to itself: read again from the profiles)
        prof.setLightEnds(false);       // the window loop comes back to the host: it walks ordered end tables
    }                                   // (else: the caller resets the tables and turns light mode off on the empty ones)
    for (size_t r = 0; r < ctxs_.size(); ++r)
    {
        uint64_t n = 0;
        if (pd_walk_active(ctxs_[r], &n, nullptr, nullptr, nullptr, nullptr, 0) != 0)
            throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
        if (n == 0)
            continue;
        std::vector<uint32_t> rg(n), start(n), id(n), last(n);
        std::vector<int32_t> dev(n);
        if (pd_walk_active_ex(ctxs_[r], &n, rg.data(), start.data(), dev.data(), id.data(), last.data(), n) != 0)
            throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(ctxs_[r]));
        // The device holds the pairs as they stood after the last segment it was given. A round that was read since then has
        // switched the slots (its segment is the one the host walks now), and a switch drains the end read slot -- every pair
        // whose end record sits there leaves the active set (quirk B1; ChromProfile::drainEndReadSlot found nobody: the pairs
        // were here) --: after one switch the pairs filed one slot ahead remain, after two or more none.
        const std::vector<uint8_t> & pendingSwitches = prof.switchCounts();
        for (uint64_t i = 0; i < n; ++i)
        {
            const unsigned g = shardRg_[r] + rg[i];
            const unsigned sw = pendingSwitches[g];
            if (sw == 0 || (sw == 1 && (last[i] & 0x80000000u) != 0))
                prof.materializeActive(g, id[i], start[i], dev[i]);
        }
    }
    deviceWalk_ = false;
}

void CallRun::checkWalkAgainstPlan(const WalkShape & shape, const std::vector<uint8_t> & equalRun, const std::vector<uint8_t> & switches,
                                   const std::vector<std::pair<size_t, size_t>> & cursors, unsigned firstPos, unsigned nWindows,
                                   const std::vector<std::vector<PairInstance>> & log)
{
    ChromProfile & prof = *profile_;
    auto fail = [&](const std::string & what) {
        throw ProfileError("[PopDel] POPDEL_WALK_CHECK: the closed-form walk differs from the window loop: " + what);
    };
    if (shape.firstPos != firstPos || shape.nWindows != nWindows)
        fail("segment shape: planned first window " + std::to_string(shape.firstPos) + " x " + std::to_string(shape.nWindows) +
             ", walked " + std::to_string(firstPos) + " x " + std::to_string(nWindows));
    std::vector<PairInstance> mine;
    for (size_t rg = 0; rg < prof.starts.size(); ++rg)
    {
        const StartSlots & st = prof.starts[rg];
        if (cursors[rg].first != st.readIdx || cursors[rg].second != prof.ends[rg].readIdx)
            fail("cursors of read group " + std::to_string(rg));
        mine.clear();
        const InsertHistogram & h = p_.histograms[rg];
        walkCheck_->segment(static_cast<unsigned>(rg), shape, p_.windowShift, switches[rg], equalRun[rg] != 0,
                            switches[rg] ? &st.slots[st.readSlot] : nullptr, h.median, h.readLength, mine);
        const std::vector<PairInstance> & theirs = log[rg];
        bool same = mine.size() == theirs.size();
        for (size_t i = 0; same && i < mine.size(); ++i)
            same = mine[i].dev == theirs[i].dev && mine[i].start == theirs[i].start && mine[i].on == theirs[i].on &&
                   mine[i].off == theirs[i].off && mine[i].id == theirs[i].id;
        if (!same)
        {
            std::string what = "instances of read group " + std::to_string(rg) + " in the segment at " + std::to_string(firstPos) + " (" +
                               std::to_string(mine.size()) + " planned, " + std::to_string(theirs.size()) + " walked)";
            for (size_t i = 0; i < std::min(mine.size(), theirs.size()); ++i)
                if (mine[i].start != theirs[i].start || mine[i].on != theirs[i].on || mine[i].off != theirs[i].off || mine[i].id != theirs[i].id)
                {
                    what += "; first difference at " + std::to_string(i) + ": planned start " + std::to_string(mine[i].start) + " id " +
                            std::to_string(mine[i].id) + " [" + std::to_string(mine[i].on) + ", " + std::to_string(mine[i].off) + "), walked start " +
                            std::to_string(theirs[i].start) + " id " + std::to_string(theirs[i].id) + " [" + std::to_string(theirs[i].on) + ", " +
                            std::to_string(theirs[i].off) + ")";
                    break;
                }
            fail(what);
        }
        if (walkCheck_->active(static_cast<unsigned>(rg)).size() != prof.active[rg].size())
            fail("active pairs after the segment, read group " + std::to_string(rg));
    }
    ++walkChecked_;
}

bool CallRun::finalizeRoi()
{
    ensureBackend();
    if (!advanceRoi(p_))
    {
        flushPending();
        printStatus("Output written to '" + p_.outfile + "'.");
        return false;
    }
    flushPending();                             // records of the finished ROI carry the old contig id
    // (the tables start afresh but pairs that are still active stay so -- the reference keeps them across the ROI change --: only
    // with none left can the next ROI's first segment be planned in closed form. The pairs the device holds come back BEFORE the
    // tables are reset -- as they stand after the last segment, which has been handed over by then -- and the entries it kept to
    // itself are not read back: the tables are about to go.)
    uint64_t left = 0;
    if (deviceWalkOffered_ && deviceWalk_)
    {
        drainHandOver();
        for (pd_ctx * c : ctxs_)
        {
            uint64_t n = 0;
            if (pd_walk_active(c, &n, nullptr, nullptr, nullptr, nullptr, 0) != 0)
                throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(c));
            left += n;
        }
        if (left != 0)
            leaveDeviceWalk(true);
    }
    profile_->fullReset();
    if (deviceWalkOffered_ && !deviceWalk_ && profile_->lightEnds())
        profile_->setLightEnds(false);
    if (walkCheck_ && walkCheck_->activeTotal() == 0)
        profile_->setActiveKnownEmpty(true);
    {
        uint64_t kept = 0;
        for (const auto & a : profile_->active)
            kept += a.size();
        if (timingWanted_)
            std::cerr << "[popdel timing] region change: " << kept << " pairs stay active in the tables (" << left << " came back from the device)" << std::endl;
    }
    if (deviceWalkOffered_)
    {
        if (!deviceWalk_)
        {
            left = 0;
            for (const auto & a : profile_->active)
                left += a.size();
        }
        if (left == 0)
        {
            profile_->setActiveKnownEmpty(true);
            deviceWalk_ = true;
            if (lightTables_)
                profile_->setLightEnds(true);
        }
    }
    nextReadPos_ = kNoPos;
    windows_.current.contig = p_.allRois[p_.nextRoi].seqName;
    windows_.current.pos = 0;
    windows_.nextChrom = Coord{"", kNoPos};
    vcf_->unlockContigName();
    for (char & b : finishedRois_)
        b = 0;
    return true;
}

// The host side of a device lives on the device's NUMA node: the process is bound to that node's CPUs before any thread is
// started (page-locked buffers, tables and staging copies then sit next to the GPU's PCIe root; on the two-socket GPU boxes an
// unbound run lands on either socket from run to run). Which node: HIP device k -> the GPU of the KFD topology it is (the k-th
// the job may see, re-dealt by ROCR_VISIBLE_DEVICES and HIP_/CUDA_VISIBLE_DEVICES when they are lists of ordinals; a list of
// UUIDs cannot be resolved without starting the runtime, and then nothing is bound) -> its PCI address -> sysfs. The CPU mask
// is narrowed only if the node's share of it still holds as many CPUs as the job may use (affinity and CFS quota,
// usableCpus()): a cpuset that spans both sockets with few CPUs on each keeps them all and only PREFERS the node's memory.
// Anything unexpected leaves the process as it is. POPDEL_NO_NUMA_BIND=1 switches it off.
#ifdef __linux__
static bool ordinalList(const char * text, std::vector<int> & out)
{
    out.clear();
    for (const char * p = text; *p;)
    {
        char * end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || (*end != ',' && *end != 0))
            return false;                    // ("GPU-<uuid>" and anything else that is not a plain ordinal)
        out.push_back(static_cast<int>(v));
        p = *end ? end + 1 : end;
    }
    return true;
}

// The ordinal among the KFD GPUs this job can see that the runtime will call HIP device `device`; -1: cannot be told.
static int kfdOrdinalOfHipDevice(int device)
{
    std::vector<int> rocr, hip;
    bool haveRocr = false, haveHip = false;
    if (const char * e = getenv("ROCR_VISIBLE_DEVICES"))
    {
        if (!ordinalList(e, rocr))
            return -1;
        haveRocr = true;
    }
    const char * h = getenv("HIP_VISIBLE_DEVICES");
    if (!h)
        h = getenv("CUDA_VISIBLE_DEVICES");
    if (h)
    {
        if (!ordinalList(h, hip))
            return -1;
        haveHip = true;
    }
    int k = device;
    if (haveHip)
    {
        if (k >= static_cast<int>(hip.size()))
            return -1;
        k = hip[static_cast<size_t>(k)];
    }
    if (haveRocr)
    {
        if (k >= static_cast<int>(rocr.size()))
            return -1;
        k = rocr[static_cast<size_t>(k)];
    }
    return k;
}
#endif

static int bindToDeviceNode(int device)
{
#ifdef __linux__
    if (getenv("POPDEL_NO_NUMA_BIND") || device < 0)
        return -1;
    device = kfdOrdinalOfHipDevice(device);
    if (device < 0)
        return -1;
    const unsigned mayUse = usableCpus();
    int seen = 0;
    for (int node = 0; node < 64; ++node)
    {
        std::ifstream props("/sys/class/kfd/kfd/topology/nodes/" + std::to_string(node) + "/properties");
        if (!props)
            continue;           // (other jobs' GPUs are hidden from this one; gaps in the numbering are normal)
        std::string key;
        unsigned long long value = 0, simd = 0, location = 0, domain = 0;
        while (props >> key >> value)
        {
            if (key == "simd_count")
                simd = value;
            else if (key == "location_id")
                location = value;
            else if (key == "domain")
                domain = value;
        }
        if (simd == 0)
            continue;
        if (seen++ != device)
            continue;
        char bdf[32];
        snprintf(bdf, sizeof(bdf), "%04llx:%02llx:%02llx.%llx", domain & 0xffffull, (location >> 8) & 0xffull, (location >> 3) & 0x1full, location & 7ull);
        std::ifstream nodeFile(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
        int numa = -1;
        if (!(nodeFile >> numa) || numa < 0)
            return -1;
        std::ifstream list("/sys/devices/system/node/node" + std::to_string(numa) + "/cpulist");
        std::string ranges;
        if (!(list >> ranges))
            return -1;
        cpu_set_t now, want;
        CPU_ZERO(&want);
        if (sched_getaffinity(0, sizeof(now), &now) != 0)
            return -1;
        for (size_t i = 0; i < ranges.size();)
        {
            size_t j = ranges.find(',', i);
            if (j == std::string::npos)
                j = ranges.size();
            const std::string part = ranges.substr(i, j - i);
            const size_t dash = part.find('-');
            const int a = atoi(part.c_str()), b = dash == std::string::npos ? a : atoi(part.c_str() + dash + 1);
            for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
                if (CPU_ISSET(c, &now))
                    CPU_SET(c, &want);
            i = j + 1;
        }
        if (CPU_COUNT(&want) == 0)
            return -1;
        if (static_cast<unsigned>(CPU_COUNT(&want)) < mayUse && CPU_COUNT(&want) < CPU_COUNT(&now))
        {
            // narrowing would cost CPUs the job has been granted: keep them, prefer the node's memory
            unsigned long mask[16] = {0};
            if (numa < static_cast<int>(sizeof(mask) * 8))
            {
                mask[numa / (8 * sizeof(unsigned long))] |= 1ul << (numa % (8 * sizeof(unsigned long)));
                (void)syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, sizeof(mask) * 8);
            }
            return numa;
        }
        if (sched_setaffinity(0, sizeof(want), &want) != 0)
            return -1;
        return numa;
    }
#else
    (void)device;
#endif
    return -1;
}

int CallRun::run()
{
    double tStart = PhaseClock::now();
    // The GPU runtime starts up (~0.2 s) while the profile headers are read, and the backend is created (tables uploaded,
    // logarithms taken) while the first window is looked for and the first round of profiles is decoded and filed: the main
    // thread needs it only when the first segment is submitted (ensureBackend()).
    std::thread warmup([] {
        const char * d = getenv("POPDEL_DEVICES");
        if (!d)
            d = getenv("POPDEL_DEVICE");
        (void)pd_device_warmup(d ? atoi(d) : 0);
        gTrace.mark("device warm");
    });
    struct Joiner
    {
        std::thread & t;
        ~Joiner() { if (t.joinable()) t.join(); }
    } joinWarmup{warmup};
    loadAndCalculateParameters(p_);
    clock_.params = PhaseClock::now() - tStart;
    gTrace.mark("parameters read");
    printStatus("Finished parameter calculation.\n");
    {
        std::ostringstream msg;
        msg << "Initialized insert size profiles from " << p_.sampleNum << " input files.";
        printStatus(msg);
    }
    finishedFiles_.assign(p_.sampleNum, 0);
    finishedRois_.assign(p_.sampleNum, 0);
    windows_.next = windows_.current = firstWindowCoordinate();
    nextCandidate_.assign(p_.sampleNum, windows_.current);

    vcf_.reset(new VcfOutput(p_.outfile, p_.inputFiles, p_.contigNames[0], p_.contigLengths[0]));
    vcf_->appendContigName(windows_.current.contig);

    // (whose window loop: decided here, so that the first segment can be planned and handed over -- by a thread that waits
    // for the backend -- while the main thread already decodes the next round)
    deviceWalkOffered_ = !getenv("POPDEL_HOST_WALK") && !getenv("POPDEL_WALK_CHECK") && !observer_ && pd_walk_supported(nullptr) == 1;
    deviceWalk_ = deviceWalkOffered_;
    backendThread_ = std::thread([this, &warmup] {
        try
        {
            warmup.join();
            createBackend();
            chooseMergeSite();
            createLoaders();            // (streams, events: 0.15 s that the main thread spends on the first round anyway)
            backendReadyAt_ = PhaseClock::now();
            backendReady_.store(true);
            gTrace.mark("backend created");
        }
        catch (...)
        {
            backendError_ = std::current_exception();
        }
    });
    struct BackendJoiner
    {
        CallRun & r;
        ~BackendJoiner() { if (r.backendThread_.joinable()) r.backendThread_.join(); }
    } joinBackend{*this};
    profile_.reset(new ChromProfile(static_cast<unsigned>(p_.readGroups.size()), p_.maxLoad, p_.windowBuffer));
    if (observer_ && observer_->onActiveOps)
        profile_->enableActiveOpLog();
    profile_->currentPos = windows_.current.pos;
    profile_->resetTo(profile_->currentPos);
    // (while the device runs the window loop nobody walks the end tables: they are kept as counts per window -- EndSlots,
    // chrom_profile.h; POPDEL_HEAVY_TABLES=1 keeps the ordered arrays, A/B runs)
    lightTables_ = !getenv("POPDEL_HEAVY_TABLES");
    if (deviceWalk_ && lightTables_)
        profile_->setLightEnds(true);
    // (runs of device-parsed records are filed from summaries the device takes: ChromProfile::addBulk; POPDEL_HOST_FILING=1
    // files every record on the host as rounds 4-5 did -- A/B runs, tests)
    sampleOfRg_.assign(p_.readGroups.size(), 0);
    for (unsigned i = 0; i < p_.sampleNum; ++i)
        for (unsigned r : p_.rgs[i])
            sampleOfRg_[r] = i;
    profile_->refill = [this](unsigned rg, const StartSlots::VirtualRun & run, StartEntry * dst) {
        readRunFromProfile(sampleOfRg_[rg], contigTable_[run.tag], run.jumpPos, run.firstWindow, run.skip, run.n, dst);
    };
    slotSource_.assign(p_.readGroups.size(), pd_walk_source{nullptr, 0, 0});
    slotSourceCount_.assign(p_.readGroups.size(), 0);
    bulkRoom_ = p_.windowBuffer / 30u + 512u;
    bulkFiling_ = deviceWalk_ && lightTables_ && !getenv("POPDEL_HOST_FILING");
    if (getenv("POPDEL_WALK_CHECK"))
        walkCheck_.reset(new WalkReference(static_cast<unsigned>(p_.readGroups.size())));
    profile_->externalActive = [this]() -> size_t {
        ensureBackend();            // (only asked from the main thread: the replay of a batch with a far window runs in order)
        if (!deviceWalk_)
            return 0;
        drainHandOver();
        uint64_t left = 0;
        for (pd_ctx * c : ctxs_)
        {
            uint64_t n = 0;
            if (pd_walk_active(c, &n, nullptr, nullptr, nullptr, nullptr, 0) != 0)
                throw ProfileError(std::string("[PopDel] Genotyping backend failed: ") + pd_last_error(c));
            left += n;
        }
        return static_cast<size_t>(left);
    };

    while (p_.fileCount != 0)
    {
        if (!goNextRegion())
        {
            flushPending();
            loopEnd_ = PhaseClock::now();
            return 0;
        }
        rounds_.emplace_back();
        rounds_.back().start = PhaseClock::now();
        if (timingWanted_)
        {
            struct rusage ru;
            getrusage(RUSAGE_SELF, &ru);
            rounds_.back().cpu = ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
        }
        if (const char * from = getenv("POPDEL_CPU_PROFILE_FROM_ROUND"))       // (diagnosis: the warmed rounds only)
            if (rounds_.size() == static_cast<size_t>(atoi(from)))
                CpuSampler::start();
        if (gTrace.on)
        {
            struct rusage ru;
            std::memset(&ru, 0, sizeof(ru));
            getrusage(RUSAGE_SELF, &ru);
            gTrace.mark("round begins; process CPU so far " + std::to_string(ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec) + " s user, " +
                        std::to_string(ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec) + " s system, " + std::to_string(ru.ru_minflt) + " page faults");
        }
        if (windows_.current.contig != roundContig_ || contigTable_.empty())
        {
            // (what a virtual run of start entries notes of its contig: an index into this table, which only grows)
            roundContig_ = windows_.current.contig;
            size_t t = 0;
            while (t < contigTable_.size() && contigTable_[t] != roundContig_)
                ++t;
            if (t == contigTable_.size())
                contigTable_.push_back(roundContig_);
            roundContigTag_ = static_cast<uint32_t>(t);
        }
        startGunzipPrefetch();
        // (large batches: 16 threads replayed 1000 samples in 0.57 s with batches of 128 and in 0.22 s with 512)
        // (wider batches, 1024 and 2048 samples, were measured no faster with the blocks parsed on the device: 0.085-0.098 against 0.088-0.091 s per round)
        // (never wider than the first two chunks of the device gunzip: an inflater's buffers are recycled once the replay has
        // passed the chunk three before it, and the replay releases a batch as a whole -- a batch that spanned four chunks would
        // wait for a chunk whose buffers it holds itself)
        // (the loader chunks its READERS -- at most the samples; with the limit taken for the smallest number of readers that
        // still needs more than three chunks, a batch of samples never spans more than two chunks of any round)
        const unsigned batch = std::min<unsigned>(std::max(512u, 8u * hostThreads()),
                                                  static_cast<unsigned>(GunzipPrefetch::batchLimitFor(p_.sampleNum)));
        for (unsigned b0 = 0; b0 < p_.sampleNum; b0 += batch)
        {
            const unsigned b1 = std::min(p_.sampleNum, b0 + batch);
            // (1) who takes part in this round (sequential bookkeeping), then their slot switches in parallel: a switch
            //     touches only the sample's own tables plus the shared right border, which is published afterwards in
            //     sample order
            double t0 = PhaseClock::now();
            std::vector<unsigned> members;
            for (unsigned i = b0; i < b1; ++i)
            {
                if (finishedFiles_[i] || finishedRois_[i])
                    continue;
                if (allTrue(finishedRois_))                     // cannot happen while sample i itself is unfinished
                {
                    if (!advanceRoi(p_))
                    {
                        markRoiComplete(i);
                        continue;
                    }
                    for (char & f : finishedRois_)
                        f = 0;
                }
                members.push_back(i);
            }
            constexpr unsigned kNoBorder = 0xffffffffu;
            std::vector<unsigned> border(members.size(), kNoBorder);
            std::vector<char> reads(members.size(), 0);
            parallelFor(members.size(), [&](size_t k) {
                const unsigned i = members[k];
                reads[k] = checkAndSwitch(p_.rgs[i], nextCandidate_[i].pos, &border[k]) ? 1 : 0;
            }, 4);
            std::vector<unsigned> readers;
            for (size_t k = 0; k < members.size(); ++k)
            {
                if (border[k] != kNoBorder)
                    profile_->currentRightBorder = border[k];
                if (reads[k])
                    readers.push_back(members[k]);
            }
            clock_.switches += PhaseClock::now() - t0;
            gTrace.mark("batch: switches done");
            // (2) parallel: inflate, parse and split the profile windows of this round
            t0 = PhaseClock::now();
            if (decodedPool_.size() < readers.size())
                decodedPool_.resize(readers.size());
            std::vector<DecodedStream> & decoded = decodedPool_;
            for (size_t k = 0; k < readers.size(); ++k)
                decoded[k].reuse();
            parallelFor(readers.size(), [&](size_t k) {
                const unsigned i = readers[k];
                const StartSlots & st = profile_->starts[p_.rgs[i][0]];
                decodeSample(i, nextCandidate_[i].contig, st.slots[st.writeSlot].right, st.numWindows, decoded[k]);
                if (!decoded[k].rawsV && !decoded[k].leanV)     // (parsed on the host: the replay reads the stream's own arrays)
                    decoded[k].viewOwn();
            });
            clock_.decode += PhaseClock::now() - t0;
            rounds_.back().decode += PhaseClock::now() - t0;
            startLoaderAhead();
            gTrace.mark("batch: decoded");
            t0 = PhaseClock::now();
            // (3) insert the records and switch slots: per sample independent unless a stream contains a window beyond the
            //     next slot (then the loader looks at ALL tables and order matters -> this batch is replayed in order)
            bool ordered = false;
            for (size_t k = 0; k < readers.size(); ++k)
                ordered |= decoded[k].sawFarWindow;
            std::vector<unsigned> codes(readers.size(), 0);
            std::fill(border.begin(), border.end(), kNoBorder);
            border.resize(readers.size(), kNoBorder);
            ++(ordered ? clock_.orderedBatches : clock_.parallelBatches);
            if (ordered)
            {
                for (size_t k = 0; k < readers.size(); ++k)
                {
                    const unsigned i = readers[k];
                    const unsigned contigIdx = p_.representativeContigs ? 0 : i;
                    codes[k] = replaySegment(i, decoded[k], p_.contigNames[contigIdx], nextCandidate_[i], p_.allRois[p_.nextRoi], nullptr);
                }
            }
            else
            {
                // (round 6: a one-read-group sample's records are filed run by run -- a run is what the loop files with nothing in
                // between, as a rule everything a round reads -- from the summaries the device took of its blocks where the tables
                // allow it: replayAdvance / fileRun)
                parallelFor(readers.size(), [&](size_t k) {     // blocks of neighbours: the per-read-group tables of adjacent samples share cache lines
                    const unsigned i = readers[k];
                    const unsigned contigIdx = p_.representativeContigs ? 0 : i;
                    codes[k] = replaySegment(i, decoded[k], p_.contigNames[contigIdx], nextCandidate_[i], p_.allRois[p_.nextRoi], &border[k]);
                }, 8);
                for (size_t k = 0; k < readers.size(); ++k)
                    if (border[k] != kNoBorder)
                        profile_->currentRightBorder = border[k];
            }
            for (size_t k = 0; k < readers.size(); ++k)
                processSegmentCode(readers[k], codes[k]);
            clock_.replay += PhaseClock::now() - t0;
            rounds_.back().replay += PhaseClock::now() - t0;
            if (gunzip_)
            {
                gunzip_->releaseBelow(b1);      // (the replay has read the device-parsed arrays where they lay)
                startLoaderAhead();
            }
            gTrace.mark("batch: replayed");
        }
        processSegment();
        gTrace.mark("segment handed over");
        if (allTrue(finishedRois_))
        {
            if (!finalizeRoi())
            {
                loopEnd_ = PhaseClock::now();
                return 0;
            }
            firstWindowOnNextRoi(windows_.current);
            profile_->currentPos = windows_.current.pos;
            profile_->resetTo(profile_->currentPos);
            nextReadPos_ = kNoPos;
            for (Coord & c : nextCandidate_)
                c = windows_.current;
        }
        else
        {
            windows_.current.pos = nextReadPos_;
            nextReadPos_ = kNoPos;
        }
    }
    flushPending();
    loopEnd_ = PhaseClock::now();
    printStatus("Output written to '" + p_.outfile + "'.");
    return 0;
}

}  // namespace

int runPopdelCall(int argc, const char ** argv, SegmentObserver * observer, bool quickExit)
{
    CallParams params;
    try
    {
        const double t0 = PhaseClock::now();
        // (the loaders of up to three rounds work on the device at once, nine streams beside the contexts': with the runtime's four
        // hardware queues their copies and kernels queue behind each other -- 0.0231 -> 0.0208 s per warmed round of 4,000 samples with
        // sixteen; set before the runtime starts, and only if the caller has not decided)
        setenv("GPU_MAX_HW_QUEUES", "16", 0);
#ifdef __GLIBC__
        // (the rounds allocate and free the same large arrays over and over: kept in the heap instead of mapped, faulted in and
        // unmapped every time -- 10,000 profiles x 2 Mb: 7.6 -> 6.0 of 17.9 CPU-seconds in the kernel, 0.082 -> 0.075 s per round
        // over the run; the MALLOC_..._ environment variables still decide if they are set)
        if (!getenv("MALLOC_MMAP_THRESHOLD_") && !getenv("MALLOC_TRIM_THRESHOLD_") && !getenv("MALLOC_TOP_PAD_"))
        {
            mallopt(M_MMAP_THRESHOLD, 1 << 30);
            mallopt(M_TRIM_THRESHOLD, 0x7fffffff);
            mallopt(M_TOP_PAD, 256 << 20);
        }
#endif
        if (const char * t = getenv("POPDEL_TIMING"))
        {
            gTrace.on = atoi(t) >= 2;
            gTrace.t0 = t0;
            gTrace.epoch0 = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        }
        int res = parseCallCommandLine(params, argc, argv);
        if (res >= 0)
            return res;
        {
            // (before the first thread is started: they inherit the mask) the device's memory node if it can be told, else the
            // node the process happens to be on; several devices may sit on several nodes: left to the scheduler
            const char * d = getenv("POPDEL_DEVICES");
            const bool several = d && strchr(d, ',');
            if (!d)
                d = getenv("POPDEL_DEVICE");
            const int node = several ? -1 : bindToDeviceNode(d ? atoi(d) : 0);
            if (node < 0)
                stayOnOneMemoryNode();
            if (getenv("POPDEL_TIMING"))
                std::cerr << "[popdel timing] host bound to NUMA node " << node << " (the device's; -1: the node the process started on)" << std::endl;
        }
        // (threads that wait for the device sleep instead of spinning: the rounds are bound by the job's CPU time)
        setenv("POPDEL_HIP_BLOCKING_SYNC", "1", 0);
        // (one segment is in flight here at a time -- the device works a per cent of a host round -- so the context's lanes would
        // only cost their set-up: one lane unless the user asks. Here, not in createBackend(): that runs on a thread of its own
        // while the main thread reads the environment every round)
        setenv("POPDEL_HIP_LANES", "1", 0);
        {
            // The loaders keep the profiles' descriptors open from round to round (GunzipPrefetch::descriptorOf): room for them in
            // the limit, and the descriptor TABLE grown to its final size now, while the process has one thread -- once it has
            // more, every doubling of the table waits for an RCU grace period (0.6 s in all on the way to 8,000 descriptors,
            // seen as one round of 0.7 s).
            struct rlimit lim;
            if (getrlimit(RLIMIT_NOFILE, &lim) == 0)
            {
                const rlim_t want = static_cast<rlim_t>(3 * params.inputFiles.size() + 1024);
                if (lim.rlim_cur != RLIM_INFINITY && lim.rlim_cur < want && lim.rlim_cur < lim.rlim_max)
                {
                    lim.rlim_cur = lim.rlim_max == RLIM_INFINITY ? want : std::min(lim.rlim_max, want);
                    (void)setrlimit(RLIMIT_NOFILE, &lim);
                    (void)getrlimit(RLIMIT_NOFILE, &lim);
                }
                const rlim_t top = lim.rlim_cur == RLIM_INFINITY ? want : std::min(want, lim.rlim_cur);
                const int null = ::open("/dev/null", O_RDONLY | O_CLOEXEC);
                if (null >= 0 && top > 16)
                {
                    const int high = fcntl(null, F_DUPFD_CLOEXEC, static_cast<int>(top - 1));
                    if (high >= 0)
                        ::close(high);
                }
                if (null >= 0)
                    ::close(null);
            }
        }
        if (!getenv("POPDEL_CPU_PROFILE_FROM_ROUND"))
            CpuSampler::start();
        gTrace.mark("command line parsed");
        int rc;
        double tRun;
        {
            CallRun run(params, observer);
            rc = run.run();
            tRun = PhaseClock::now();
            gTrace.mark("run over");
            if (quickExit)
            {
                // A command-line run ends here: the records are written; tearing down a thousand window tables, HBM and
                // gigabytes of page-locked memory costs a few hundred milliseconds that the process exit does for free.
                run.closeOutput();
                run.printTiming();
                CpuSampler::stopAndWrite();
                // (not once the backend has been offered the regions: page-locked memory is the driver's to release)
                if (!getenv("POPDEL_NO_PARALLEL_DISCARD") && !run.tableRegionsOffered())
                {
                    // ... and the big arrays' pages go back from all threads at once (the exit path frees page by page on
                    // one CPU: 1.2 GB of them for 1,000 samples)
                    const auto regions = HugePool::instance().regions();
                    parallelFor(regions.size(), [&](size_t i) { HugePool::discardRegion(regions[i].first, regions[i].second); });
                }
                if (getenv("POPDEL_TIMING"))
                {
                    struct rusage ru;
                    std::memset(&ru, 0, sizeof(ru));
                    getrusage(RUSAGE_SELF, &ru);
                    std::cerr << "[popdel timing] run " << tRun - t0 << " s; teardown skipped; CPU time " << ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec
                              << " s user, " << ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec << " s system, " << ru.ru_minflt << " page faults" << std::endl;
                }
                if (gTrace.on)      // (wall-clock stamps: what the caller's clock sees before main and after the exit call)
                    std::cerr << "[popdel timeline] epoch seconds at main " << std::fixed << gTrace.epoch0 << ", at exit "
                              << std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() << std::endl;
                std::cout.flush();
                std::cerr.flush();
                fflush(nullptr);
                _exit(rc);
            }
        }
        if (getenv("POPDEL_TIMING"))
            std::cerr << "[popdel timing] run " << tRun - t0 << " s; teardown " << PhaseClock::now() - tRun << " s" << std::endl;
        return rc;
    }
    catch (const std::exception & e)
    {
        std::cerr << e.what() << std::endl;
        return 1;
    }
}

}  // namespace pdh
