// C ABI of libpopdel_hip.so (include/popdel_hip.h): context, HBM staging of segments, launch orchestration and
// result packing. No CPU fallback lives here: every entry point either runs the HIP kernels or fails.
#include <dlfcn.h>
#include <functional>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <thread>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "parallel.h"
#include "pd_build.h"
#include "pd_device.h"
#include "pd_exchange.h"
#include "pd_walk.h"
#include "pinned_host.h"
#include "popdel_hip.h"

namespace pd {

void launchStage1(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                  uint8_t * tileTodo, unsigned * todoList, unsigned * bigList, unsigned maxBlockChunks, const ScanRoom & room,
                  hipStream_t stream);
void launchStage1Fallback(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                          uint8_t * tileTodo, unsigned * todoList, const unsigned * bigList, hipStream_t stream);
void launchActive8(const uint32_t * winStartPk, const uint32_t * endCountPk, unsigned nEntities, unsigned W, unsigned K, uint8_t * out,
                   hipStream_t stream);
void launchScreen(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag, uint8_t * tileTodo,
                  unsigned * todoList, hipStream_t stream);
void launchScreenFallback(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                          const unsigned * todoList, hipStream_t stream);
void launchExactFlagged(const SegmentView & seg, const CohortView & co, const int * flagged, unsigned nFlagged,
                        const unsigned * nFlaggedDev, const int * ownWinFlag, const uint8_t * tileTodo, int32_t * q3, uint8_t * flags,
                        int * winFlag, unsigned * rest, hipStream_t stream);
void launchPackFlagged(const int32_t * q3, const uint8_t * flags, const int * flaggedWindows, unsigned W, unsigned nLocal,
                       const ClusterInput & in, unsigned rank, hipStream_t st);
void launchZero(const ZeroRanges & r, hipStream_t st);
void launchCompactFlagged(const int * winFlag, unsigned W, unsigned world, size_t rowStride, unsigned cap, int * flagged,
                          unsigned * hdr, hipStream_t st);
void launchListCells(const int * flagged, const int * candLen, const int * candCount, unsigned cap, unsigned listCap,
                     unsigned * hdr, hipStream_t st);
void launchCluster(const ClusterInput & in, int uniformThreshold, int * scratch, unsigned rowLen, int * candLen, int * candCount,
                   hipStream_t st);
void launchCopyBytes(const void * src, void * dst, size_t n, hipStream_t st);
void launchSpin(unsigned long long ticks, hipStream_t st);
void runStage2Init(const SegmentView & seg, const CohortView & co, Stage2Buffers & b, const uint8_t * flags, hipStream_t st);
void runStage2Main(const SegmentView & seg, const CohortView & co, Stage2Buffers & b, const uint8_t * flags, Exchange & x,
                   bool anyGenericTile, unsigned * pinnedWord, hipEvent_t likBegin, hipEvent_t likEnd, hipStream_t st);
void buildLogTables(const double * hist, double4 * tab, double2 * tab2, size_t n, RgConst * rg, unsigned nRg, hipStream_t st);
void launchGatherOut(const Stage2Buffers & b, unsigned * emitCells, unsigned nSamples, uint32_t * out, hipStream_t st);

}  // namespace pd

namespace {

std::string g_createError;

// RCCL is bound at run time, when a context is first sharded with pd_ctx_set_shard_rccl(): the library has no load-time
// dependency on it, and the copy that is opened is the system one, built against the HIP runtime this library links
// (a Python process may carry a second RCCL inside its torch wheel).
struct RcclApi
{
    void * handle = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
    std::string error;
};

RcclApi & rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char * names[] = {getenv("POPDEL_RCCL_LIBRARY"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char * n : names)
        {
            if (!n || !*n)
                continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle)
                break;
        }
        if (!api.handle)
        {
            const char * why = dlerror();                  // (a second call would return NULL: the first one clears it)
            api.error = std::string("cannot load RCCL: ") + (why ? why : "librccl.so.1 not found");
            return;
        }
        api.getUniqueId = reinterpret_cast<decltype(api.getUniqueId)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.commInitRank = reinterpret_cast<decltype(api.commInitRank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.allGather = reinterpret_cast<decltype(api.allGather)>(dlsym(api.handle, "ncclAllGather"));
        api.commDestroy = reinterpret_cast<decltype(api.commDestroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.getErrorString = reinterpret_cast<decltype(api.getErrorString)>(dlsym(api.handle, "ncclGetErrorString"));
        if (!api.getUniqueId || !api.commInitRank || !api.allGather || !api.commDestroy || !api.getErrorString)
            api.error = "the RCCL library lacks an expected entry point";
    });
    return api;
}

struct HipError
{
    std::string what;
};

#define PD_HIP(expr)                                                                                          \
    do                                                                                                        \
    {                                                                                                         \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            throw HipError{std::string(#expr) + ": " + hipGetErrorString(e_)};                                \
    } while (0)

// What the process holds in HBM and in page-locked host memory (POPDEL_HIP_MEMSTAT=1 prints it at every collect).
std::atomic<uint64_t> g_devBytes{0}, g_pinnedBytes{0}, g_devAllocs{0}, g_pinnedAllocs{0};
// (parking lasts as long as a local group that asked for it: pd_local_group_destroy frees what was parked)
std::atomic<bool> g_parkFrees{false};
std::mutex g_parkLock;
std::vector<void *> g_parked;
unsigned g_parkingGroups = 0;

// Device allocation that frees itself.
template <typename T>
struct DevBuf
{
    T *    p = nullptr;
    size_t n = 0;
    bool   fineGrained = false;     // allocate with hipDeviceMallocFinegrained (memory that peers write and this device reads)
    DevBuf() {}
    DevBuf(const DevBuf &) = delete;
    DevBuf & operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p)
        {
            // hipFree() waits for the whole device. Where contexts of one process share a GPU AND exchange through kernels that
            // wait for each other (exchange.hip on one device: tests), a context that frees while its peer's exchange kernel
            // waits for its next push would wait for that kernel in turn: such processes park replaced buffers until they end.
            if (g_parkFrees.load())
            {
                std::lock_guard<std::mutex> lock(g_parkLock);
                g_parked.push_back(p);
            }
            else
                (void)hipFree(p);
            g_devBytes -= n * sizeof(T);
        }
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count)
    {
        release();
        if (count)
        {
            if (fineGrained)
                PD_HIP(hipExtMallocWithFlags(reinterpret_cast<void **>(&p), count * sizeof(T), hipDeviceMallocFinegrained));
            else
                PD_HIP(hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
            g_devBytes += count * sizeof(T);
            ++g_devAllocs;
        }
        n = count;
    }
    void ensure(size_t count)
    {
        if (count > n)
            alloc(count + count / 8);
    }
    void upload(const std::vector<T> & v, hipStream_t st)
    {
        ensure(v.size());
        if (!v.empty())
            PD_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    }
    template <typename P>
    void uploadPinned(const P & v, hipStream_t st)
    {
        ensure(v.size());
        if (v.size())
            PD_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    }
};

// Page-locked host array (grow-only), so that segment staging copies run at PCIe rate.
template <typename T>
struct PinnedBuf
{
    T *    p = nullptr;
    size_t cap = 0, n = 0;
    PinnedBuf() {}
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf & operator=(const PinnedBuf &) = delete;
    ~PinnedBuf()
    {
        if (p)
        {
            pd::pinnedFree(p);
            g_pinnedBytes -= cap * sizeof(T);
        }
    }
    void resize(size_t count)
    {
        if (count > cap)
        {
            if (p)
            {
                pd::pinnedFree(p);
                g_pinnedBytes -= cap * sizeof(T);
            }
            p = nullptr;
            cap = 0;
            const size_t want = count + count / 8 + 1024;
            p = static_cast<T *>(pd::pinnedAlloc(want * sizeof(T)));
            if (!p)
                throw HipError{"page-locked host memory: " + std::to_string(want * sizeof(T)) + " bytes not to be had"};
            cap = want;
            g_pinnedBytes += cap * sizeof(T);
            ++g_pinnedAllocs;
        }
        n = count;
    }
    void assignZero(size_t count)
    {
        resize(count);
        std::memset(p, 0, count * sizeof(T));
    }
    T * data() { return p; }
    const T * data() const { return p; }
    size_t size() const { return n; }
    T & operator[](size_t i) { return p[i]; }
    const T & operator[](size_t i) const { return p[i]; }
};

struct StagedSegment
{
    uint32_t firstPos = 0, nWindows = 0, shift = 0, maxLife = 0;
    uint32_t maxBlockChunks = 0;    // chunks of the densest 512-window block any scan stream has (picks the kernel's LDS size)
    uint64_t nChunks = 0, nPairs = 0, nActivePairs = 0;
    DevBuf<pd::PairDO> pairs;
    DevBuf<uint32_t> start;
    DevBuf<uint64_t> rgBase;
    DevBuf<uint32_t> winStart;
    DevBuf<uint32_t> pairsPk;
    DevBuf<uint32_t> endCount;
    DevBuf<uint8_t> rgPacked;
    // index of the stage-1 image when it is cut with its own chunk life (SegmentView::maxLifePk); else the arrays above
    DevBuf<uint64_t> rgBasePk;
    DevBuf<uint32_t> winStartPk, endCountPk;
    DevBuf<uint8_t> active8;                // n(w) per (stream, window), capped at 255 (SegmentView::active8)
    uint32_t maxLifePk = 0;
    bool ownPkIndex = false;
    hipEvent_t built = nullptr;             // device-side build: recorded behind its last kernel (copy stream); runs wait for it
    bool waitBuilt = false;
    ~StagedSegment()
    {
        if (built)
            (void)hipEventDestroy(built);
    }
    DevBuf<uint8_t> scanGeneric;
    const uint32_t * scanRg = nullptr;      // the context's table (see SegmentView::scanRg)
    uint32_t nRealRg = 0;
    bool hasScanGeneric = false;
    bool ownedBySubmit = false;
};

struct Run
{
    uint64_t handle = 0;
    hipEvent_t s1Begin = nullptr, s1End = nullptr;      // around the scan kernel (timing)
    hipEvent_t scanDone = nullptr;                      // behind the scan's last kernel when it runs on the scan stream
    int set = 0;                                        // which of the two stage-1 output sets the scan writes
    bool launched = false;          // false: accepted while the previous ticket was still open, scan not enqueued yet
};

// Stage-1 outputs of one run. Two sets, so that the scan of the waiting ticket can be enqueued while the open one's
// values are still wanted (pd_stage1_last). (Round 2 measured the scan on its OWN stream, started when the ticket is
// issued, concurrently with the open segment's stage 2: the scan's one big launch fills every CU for its 1.2 ms and the
// ~60 short stage-2 kernels crawl until it is through -- 2.94 ms per pass against 2.59 for the order used here, with or
// without stream priorities.)
struct Stage1Out
{
    DevBuf<int32_t> q3;
    DevBuf<uint8_t> flags;
    DevBuf<int> winFlag;
    DevBuf<uint8_t> tileTodo;
    DevBuf<unsigned> todoList;      // [0] = number of entries, then the 256-window tiles left to the fallback kernel
    DevBuf<unsigned> bigList;       // [0] = number of entries, then the 512-window blocks left to the 128-candidate kernel
    DevBuf<unsigned> exactRest;     // screen mode: [0] = number of entries, then the (flagged window, 64-sample wave tile) units left
                                    // to the 128-candidate networks of the exact pass
    bool screened = false;          // q3 holds exact values in the flagged windows only (screen + exact pass), else everywhere
    // pd_ctx_stage1_view(): the full scan's outputs beside the screen's, for pd_stage1_last()
    DevBuf<int32_t> fullQ3;
    DevBuf<uint8_t> fullFlags;
    DevBuf<int> fullWinFlag;
    bool fullValid = false;
    uint32_t windows = 0;
};

}  // namespace

// Lanes of a context (POPDEL_HIP_LANES = K > 1; round 4). Stage 2 of a sparse segment is ~45 short dependent launches and
// three host round trips during which most of the chip idles, and a sharded cohort adds the wire time of ~15 exchanges:
// what fills those gaps is ANOTHER segment's stage 2. A context with lanes owns K child contexts -- streams, stage-1
// outputs and stage-2 scratch of their own, the cohort's tables uploaded once more -- that take the tickets in turn
// (ticket t -> lane t mod K), each on a host thread of its own running the ordinary run + collect sequence; the parent
// keeps what is per cohort and per segment (staged segments, the window loop's state, the upload path) and hands the
// finished calls out in ticket order. Results are those of a plain context, bit for bit (the lanes run the same code on the
// same staged segments). Up to 2 K tickets may be outstanding.
struct LaneJob
{
    uint64_t ticket = 0, handle = 0;
};
struct LaneResult
{
    bool done = false;
    uint64_t handle = 0;
    int rc = 0;
    pd_calls * calls = nullptr;
    std::string error;
    pd_timing timing;
    unsigned lane = 0;
};
struct LaneWorker
{
    pd_ctx * ctx = nullptr;
    std::thread thread;
    std::deque<LaneJob> jobs;           // (guarded by LanesState::m)
};
struct LanesState
{
    std::vector<std::unique_ptr<LaneWorker>> workers;
    unsigned active = 0;                // lanes that take tickets (a sharded cohort of several ranks uses one)
    std::mutex m;
    std::condition_variable wake, done;
    std::map<uint64_t, LaneResult> results;     // by ticket: everything outstanding
    bool stop = false;
    unsigned lastLane = 0;              // lane of the most recently collected ticket (pd_stage1_last)
    std::vector<uint64_t> repeatedSeen; // per lane: its cumulative count of repeated stage-2 batches as last reported
};

// Contexts of one process that exchange among themselves (pd_ctx_set_shard_local).
struct pd_local_group
{
    uint32_t world = 0;
    std::mutex m;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    bool failed = false;
    bool parksFrees = false;        // this group turned the parking of replaced device buffers on (groupDecide)
    std::vector<int> devices;       // of its members, as they registered (pd_local_group_destroy drains them)
    std::vector<pd_ctx *> members;
};

struct pd_ctx
{
    pd_config cfg;
    int device = 0;
    uint32_t nRg = 0, nSamples = 0;
    std::vector<uint32_t> sampleRg;
    std::string error;
    hipStream_t stream = nullptr;       // all kernels
    hipStream_t copyStream = nullptr;   // segment uploads and the result D2H, so that the next segment's scan can run underneath it
    // The scan of the WAITING ticket beside the open ticket's stage 2 (round 4): a stream of its own, lower priority, and a
    // launch that leaves room on every CU (pd::ScanRoom) -- stage 2 of a sparse segment is ~45 short dependent launches
    // that use a fraction of the chip, the scan is one VALU-bound launch. POPDEL_HIP_SCAN_OVERLAP: "0" = the scan on the
    // kernel stream behind the open ticket's last kernel (rounds 1-3), "p<N>" persistent grid of N workgroups per CU,
    // "l<N>" plain grid capped at N per CU through LDS, "s" plain grid, stream only.
    hipStream_t scanStream = nullptr;
    pd::ScanRoom scanRoom{pd::ScanRoom::kFull, 7, 256, nullptr};
    DevBuf<unsigned> dScanCounter;      // two words, one per stage-1 output set
    // POPDEL_HIP_XCH_DELAY_US: every exchange of a sharded context is followed by a one-wave kernel that waits this long on
    // the stream (wall-clock ticks): the wire time an exchange over xGMI would add, for measurements with one rank
    unsigned long long xchDelayTicks = 0;
    hipEvent_t resultsPacked = nullptr;
    hipEvent_t cellsFetched = nullptr;
    hipEvent_t likBegin = nullptr, likEnd = nullptr;    // around the data-likelihood launch ahead of the EM loop (pd_timing)
    DevBuf<pd::RgConst> dRg;
    DevBuf<double> dHist;
    DevBuf<double4> dTab;
    DevBuf<double2> dTab2;
    DevBuf<unsigned> dIterFlags;
    DevBuf<uint32_t> dSampleRg;
    DevBuf<uint32_t> dScanRg;
    // tiles with multi-read-group samples (pd::MultiTiles): static tables
    DevBuf<uint32_t> dMtTile, dMtRgLo, dMtRgHi, dMtWaveBase, dMtWaveTile, dRgSample;
    pd::MultiTiles multiTiles;
    std::vector<uint32_t> scanSample;       // appended scan read group j -> its sample (samples with several read groups)
    std::vector<uint32_t> maxLoad;          // per real read group
    pd::CohortView cohort;
    int uniformThreshold = -1;      // minInitDelLen if it is the same for all read groups of the cohort, else -1
    // The cohort may be sharded by samples over several contexts (pd_ctx_set_shard); defaults describe "all samples here".
    uint32_t rank = 0, world = 1, nSamplesGlobal = 0, samplesPerRank = 0;
    bool sharded = false;           // pd_ctx_set_shard*() was called (world may be 1): the exchanged code path runs
    uint32_t flagCapGuess = 64;     // sharded: flagged windows the next segment's clustering input is laid out for
    pd_shard shard;
    ncclComm_t comm = nullptr;      // exchange by the library itself (pd_ctx_set_shard_rccl) instead of the host's callbacks
    pd_local_group * group = nullptr;   // ... or among the contexts of this process (pd_ctx_set_shard_local)
    DevBuf<unsigned> dMinInitGlobal;
    DevBuf<uint8_t> ownArena;       // exchange areas when world == 1 (otherwise the host owns the buffer)
    uint8_t * arena = nullptr;
    // contexts of one process (pd_ctx_set_shard_local): the device-side carrier (exchange.hip) unless POPDEL_HIP_GROUP_CARRIER=host
    bool flagCarrier = false;       // (decided at the group's first collective step: groupDecide)
    bool flagsWanted = false, groupDecided = false;
    unsigned * xchFlags = nullptr;  // this context's flag block (fine-grained device memory)
    unsigned * xchErr = nullptr;    // page-locked host word raised by a wait that gave up
    unsigned xchSeq = 0;            // running number of the context's exchanges
    bool peersEnabled = false;

    std::map<uint64_t, std::unique_ptr<StagedSegment>> staged;
    std::vector<std::unique_ptr<StagedSegment>> spare;      // released segments: their HBM buffers are reused (no hipMalloc per segment)
    // lanes (see LanesState): a lane runs the segments its `owner` has staged; `stagedLock` guards the owner's two containers
    // above against its lanes' threads
    pd_ctx * owner = nullptr;
    std::mutex stagedLock;
    // Lanes share the chip well while stage 2 is a string of short launches; a candidate-rich segment's stage 2 fills the chip
    // by itself (its kernels are bound by the rate of table gathers), and two of them side by side only evict each other's
    // tables (measured: 45 ms per candidate-dense pass with three lanes against 27 with one). A lane whose segment has more
    // than kHeavyUnits (cell, sample) units therefore takes this lock for the length of its stage 2.
    std::mutex heavyStage2;
    std::unique_ptr<LanesState> lanes;
    // host-side scratch of pd_segment_stage(), kept between calls
    PinnedBuf<pd::PairDO> hPairs;
    PinnedBuf<uint32_t> hStart, hWinStart, hEndCount;
    std::vector<uint64_t> hRgBase;
    PinnedBuf<uint32_t> hPairsPk;
    std::vector<uint8_t> hRgPacked, hScanGeneric;
    // page-locked staging for the small control transfers of a collect (window flags, candidate lists, cell records,
    // counters): pageable memory costs every one of these round trips an extra bounce copy
    PinnedBuf<int> hWinFlag, hFlagged, hCandLen, hCandCount;
    PinnedBuf<pd::CellState> hCells;
    PinnedBuf<unsigned> hCounters, hIterFlag, hCandHdr;
    DevBuf<unsigned> dCandHdr;            // flagged-window count, cell count, cell list (listCellsKernel)
    bool deviceLists = true;              // POPDEL_HIP_HOST_LISTS=1: the host forms the flagged-window and cell lists
    // Stage 1 as a SCREEN (stage1.hip): count the pairs beyond the smallest threshold per cell, take exact Q3 values in the
    // flagged windows only. POPDEL_HIP_SCAN=full: the round-1..4 scan (exact Q3 of every cell).
    bool screenScan = true;
    pd_stage_timing stageTiming{};        // pd_last_stage_timing()
    std::chrono::steady_clock::time_point stageBegan;
    bool stage1View = false;              // pd_ctx_stage1_view(): keep the full scan's Q3 of every cell for pd_stage1_last()
    // stage 2 launched without waiting for the pool sizes (see pd_segment_collect); POPDEL_HIP_EXACT_POOLS=1 turns it off
    bool optimisticPools = true;
    size_t stage2Budget = static_cast<size_t>(24) << 30;    // bytes of per-(cell, sample) buffers per stage-2 batch (POPDEL_HIP_STAGE2_BYTES)
    bool shardSettingsChecked = false;    // sharded over several ranks: the first collect compares the ranks' settings
    bool genericSeen = false;             // some segment needed the generic stage-2 kernels
    uint64_t nRepeatedBatches = 0;
    // device-side layout build (build.hip): the uploaded instances and the small tables around them
    PinnedBuf<uint16_t> hIn16[4];           // dev, on, off, start distance (narrow upload)
    PinnedBuf<uint8_t> hIn8[3];             // on difference, life, start distance (compact upload)
    DevBuf<uint8_t> dIn8[3];
    PinnedBuf<uint32_t> hExc[5];            // its exceptions: instance, on difference, life, start distance; first one per read group
    DevBuf<uint32_t> dExc[5];
    bool compactUpload = true;              // POPDEL_HIP_PLAIN_UPLOAD=1: 8 bytes per instance (A/B runs)
    PinnedBuf<uint32_t> hIn32[4];           // dev, on, off, start (wide upload)
    PinnedBuf<uint64_t> hRgOffsets;
    DevBuf<uint16_t> dIn16[4];
    DevBuf<uint32_t> dIn32[4];
    DevBuf<uint64_t> dRgOffsets, dTotals;
    DevBuf<uint32_t> dScanSample, dEntChunks, dEntMax;
    DevBuf<unsigned long long> dLifeHist;
    DevBuf<unsigned> dProblems;
    PinnedBuf<unsigned long long> hLifeHist, hLifeXch;
    PinnedBuf<uint64_t> hTotals, hShardCheck;
    PinnedBuf<uint32_t> hEntMax;
    // the window loop in closed form (pd_walk_submit, walk.hip): the state that goes from segment to segment, in two sets
    // (a segment's kernel reads one and writes the other), and the staging of the loader's tables
    struct WalkBuffers
    {
        DevBuf<uint32_t> aStart, aId, aLast, nActive, pStart, pId, pLast, nPending;
        DevBuf<int32_t> aDev, pDev;
    } walkState[2];
    int walkCur = 0;                        // the set that holds the state after the last submitted walk
    bool walkReady = false;
    uint32_t walkCapActive = 0, walkCapPending = 1024;
    PinnedBuf<uint32_t> hWalkBorder, hWalkIdOffset;
    PinnedBuf<uint2> hWalkEntries;
    PinnedBuf<uint64_t> hWalkSrc;           // per read group: device-visible address of its entries in the host's own tables (0: staged)
    DevBuf<uint64_t> dWalkSrc;
    PinnedBuf<uint64_t> hWalkDevSrc;        // ... or their address in an inflater's records on this GPU (0: none)
    DevBuf<uint64_t> dWalkDevSrc;
    PinnedBuf<uint8_t> hWalkFlags;          // switches, then equal_run
    DevBuf<uint32_t> dWalkBorder, dWalkIdOffset, dWalkCount;
    DevBuf<uint2> dWalkEntries;
    DevBuf<uint8_t> dWalkFlags;
    DevBuf<uint64_t> dWalkEntryOffsets;
    DevBuf<unsigned long long> dWalkTotals; // [0] instances of the segment, [1] pairs left active, [2] problems
    PinnedBuf<unsigned long long> hWalkTotals;
    uint64_t walkActive = 0;                // pairs left active after the last submitted walk
    uint64_t walkGathered = 0, walkStaged = 0;      // entries the device fetched from the host's tables / the host threads staged
    uint64_t walkFromDevice = 0;                    // ... / taken from the records an inflater left on the device (pd_walk_source)
    uint64_t nextHandle = 1, nextTicket = 1;
    std::map<uint64_t, Run> runs;

    Stage1Out s1[2];
    int lastSet = 0;                // set of the most recently collected run (pd_stage1_last)
    std::vector<int32_t> hQ3View;
    std::vector<uint8_t> hFlagsView;
    // stage-2 scratch
    DevBuf<int> dFlagged, dClusterScratch, dCandLen, dCandCount;
    DevBuf<pd::CellState> dCells;
    DevBuf<double> dDl;
    DevBuf<int> dShifts;
    DevBuf<int16_t> dCellDev, dCellStart;
    DevBuf<uint16_t> dCellN;
    DevBuf<uint8_t> dWaveRows;
    DevBuf<unsigned> dTileMode;
    DevBuf<int16_t> dMDev, dMStart;
    DevBuf<uint16_t> dMN;
    DevBuf<uint8_t> dMWaveRows;
    DevBuf<double> dRgOutD;
    DevBuf<unsigned> dRgOutU;
    DevBuf<unsigned> dSuppFirst, dSuppLast, dEmit;
    DevBuf<uint32_t> dPl, dLad, dDad, dFl, dOutPl;
    pd_timing timing;
    // per-sample result arrays kept in HBM for pd_calls_merge (pd_ctx_keep_results_on_device)
    bool keepOnDevice = false;
    uint64_t keptId = 0, nextKeptId = 1;    // identifies the collect whose columns sit in dOutPl (0: none); see CallsBox
    DevBuf<uint32_t> dMergeIdx, dMergeOut;
    DevBuf<unsigned> dMergeCounters;
    uint32_t nAllGathers = 0;       // of the collect in progress
    uint64_t allGatherBytes = 0;
    // device time of the exchanges of the collect in progress: an event pair around each (recycled), summed at its end;
    // the in-process group exchanges under host synchronisation and is timed on the host clock instead
    std::vector<hipEvent_t> xchEvents;
    size_t xchEventsUsed = 0;
    double xchHostMs = 0;

    ~pd_ctx()
    {
        staged.clear();
        spare.clear();
        if (comm)
            (void)rccl().commDestroy(comm);
        for (hipEvent_t e : xchEvents)
            (void)hipEventDestroy(e);
        if (resultsPacked)
            (void)hipEventDestroy(resultsPacked);
        if (cellsFetched)
            (void)hipEventDestroy(cellsFetched);
        if (likBegin)
            (void)hipEventDestroy(likBegin);
        if (likEnd)
            (void)hipEventDestroy(likEnd);
        if (scanStream)
            (void)hipStreamDestroy(scanStream);
        if (copyStream)
            (void)hipStreamDestroy(copyStream);
        if (stream)
            (void)hipStreamDestroy(stream);
    }
};

namespace {

// Page-locked host blocks for the per-sample result arrays (D2H at PCIe rate, no staging copy). Blocks are recycled:
// pd_calls_free() hands them back.
struct PinnedBlock
{
    void * p = nullptr;
    size_t bytes = 0;
};
std::vector<PinnedBlock> g_pinnedPool;
std::mutex g_pinnedLock;            // contexts may live on different host threads

PinnedBlock takePinned(size_t bytes)
{
    std::unique_lock<std::mutex> guard(g_pinnedLock);
    size_t best = g_pinnedPool.size();
    for (size_t i = 0; i < g_pinnedPool.size(); ++i)
        if (g_pinnedPool[i].bytes >= bytes && (best == g_pinnedPool.size() || g_pinnedPool[i].bytes < g_pinnedPool[best].bytes))
            best = i;
    if (best != g_pinnedPool.size())
    {
        PinnedBlock b = g_pinnedPool[best];
        g_pinnedPool.erase(g_pinnedPool.begin() + best);
        return b;
    }
    guard.unlock();
    PinnedBlock b;
    b.bytes = bytes + bytes / 4 + 4096;
    b.p = pd::pinnedAlloc(b.bytes);
    if (!b.p)
        throw HipError{"page-locked host memory: " + std::to_string(b.bytes) + " bytes not to be had"};
    g_pinnedBytes += b.bytes;
    ++g_pinnedAllocs;
    return b;
}

void givePinned(PinnedBlock b)
{
    if (!b.p)
        return;
    PinnedBlock drop;
    {
        // keep the four LARGEST blocks seen: a full pool of small blocks (earlier, smaller segments) must not send every
        // large result block of a later segment through hipHostMalloc / hipHostFree (~70 ms for 0.5 GB)
        std::lock_guard<std::mutex> guard(g_pinnedLock);
        if (g_pinnedPool.size() < 8)           // (a context with three lanes has up to six results in flight)
        {
            g_pinnedPool.push_back(b);
            return;
        }
        size_t smallest = 0;
        for (size_t i = 1; i < g_pinnedPool.size(); ++i)
            if (g_pinnedPool[i].bytes < g_pinnedPool[smallest].bytes)
                smallest = i;
        if (g_pinnedPool[smallest].bytes < b.bytes)
        {
            drop = g_pinnedPool[smallest];
            g_pinnedPool[smallest] = b;
        }
        else
            drop = b;
    }
    pd::pinnedFree(drop.p);
    g_pinnedBytes -= drop.bytes;
}

// pd_calls plus what pd_calls_free() needs to know.
struct CallsBox
{
    pd_calls pub;
    PinnedBlock pinned;
    pd_ctx * owner = nullptr;       // the context whose HBM holds the per-sample columns (pd_ctx_keep_results_on_device) ...
    uint64_t keptId = 0;            // ... as long as this is still the context's keptId
};

// The exchange buffer: owned by the host if it runs the exchanges (pd_ctx_set_shard with callbacks), else by the library.
void groupBarrier(pd_ctx & c);

// Which carrier the contexts of a local group use, decided once, by all of them alike, at their first collective step (all
// members are registered then): the host-driven one unless the device-side one (exchange.hip) is asked for by name
// (POPDEL_HIP_GROUP_CARRIER=flags; it has never run with its contexts on different GPUs, pd_ctx_set_shard_local). Where two
// contexts sit on the SAME GPU (tests, debugging) a waiting exchange kernel can end up in front of what it waits for (a shared
// hardware queue; a call of the peer's host thread that waits for the whole device): the device-side carrier then also needs
// hardware queues to spare (GPU_MAX_HW_QUEUES >= 8; tests/conftest.py sets 16), and replaced device buffers are parked until
// the group is destroyed.
void groupDecide(pd_ctx & c)
{
    groupBarrier(c);
    bool flags = c.flagsWanted;
    bool shared = false;
    for (uint32_t p = 0; p < c.world; ++p)
        for (uint32_t q = p + 1; q < c.world; ++q)
            if (c.group->members[p] && c.group->members[q] && c.group->members[p]->device == c.group->members[q]->device)
                shared = true;
    for (uint32_t p = 0; p < c.world; ++p)
        flags = flags && c.group->members[p] && c.group->members[p]->xchFlags;
    if (shared)
    {
        // (one GPU standing in for several: only when asked for by name, with hardware queues to spare; frees are parked)
        const char * q = getenv("GPU_MAX_HW_QUEUES");
        const char * how = getenv("POPDEL_HIP_GROUP_CARRIER");
        flags = flags && how && std::strcmp(how, "flags") == 0 && q && atoi(q) >= 8;
        if (flags)
        {
            std::lock_guard<std::mutex> lock(g_parkLock);
            if (!c.group->parksFrees)
            {
                c.group->parksFrees = true;
                ++g_parkingGroups;
                g_parkFrees.store(true);
            }
        }
    }
    c.flagCarrier = flags;
    c.groupDecided = true;
    if (getenv("POPDEL_HIP_GROUP_TRACE"))
        fprintf(stderr, "[popdel_hip] local group of %u contexts: %s carrier\n", c.world, flags ? "device-side (flags)" : "host-driven");
    groupBarrier(c);
}

bool hostOwnsArena(const pd_ctx & c) { return c.sharded && !c.comm && !c.group && c.shard.reserve && c.shard.all_gather; }

uint8_t * reserveArena(pd_ctx & c, size_t bytes)
{
    if (!hostOwnsArena(c))
    {
        if (c.group && !c.groupDecided)
            groupDecide(c);
        if (bytes > c.ownArena.n)
        {
            PD_HIP(hipStreamSynchronize(c.stream));        // nothing in flight may still use the buffer that is replaced
            if (c.flagCarrier && c.world > 1)
            {
                // The peers write into this buffer through its address: growth is a collective step (every context asks for the
                // same sizes in the same order, so all arrive here together) -- everybody drains, the buffers are replaced, and
                // nobody goes on before all new addresses are published. After the first segments the sizes settle.
                PD_HIP(hipStreamSynchronize(c.copyStream));
                groupBarrier(c);
                // (peers store into this buffer through peer-mapped pointers and this device's kernels read it right after a
                // flag wait: fine-grained memory, like the flag block -- a coarse-grained allocation may keep stale lines of
                // the previous exchange in this device's L2 across kernels when the writer sits on another GPU)
                c.ownArena.fineGrained = true;
                c.ownArena.ensure(bytes);
                c.arena = c.ownArena.p;
                groupBarrier(c);
                return c.arena;
            }
        }
        c.ownArena.ensure(bytes);
        c.arena = c.ownArena.p;
    }
    else
    {
        PD_HIP(hipStreamSynchronize(c.stream));            // nothing in flight may still use the previous buffer
        c.arena = static_cast<uint8_t *>(c.shard.reserve(c.shard.user, bytes));
        if (!c.arena)
            throw HipError{"the host's reserve() callback returned no buffer"};
    }
    return c.arena;
}

// Barrier of the host threads of a local group; throws once any member has failed.
void groupBarrier(pd_ctx & c)
{
    pd_local_group & g = *c.group;
    std::unique_lock<std::mutex> lock(g.m);
    if (g.failed)
        throw HipError{"another context of the local group failed"};
    const uint64_t gen = g.generation;
    if (++g.arrived == g.world)
    {
        g.arrived = 0;
        ++g.generation;
        g.cv.notify_all();
        return;
    }
    g.cv.wait(lock, [&] { return g.generation != gen || g.failed; });
    if (g.generation == gen)
        throw HipError{"another context of the local group failed"};
}

void groupFail(pd_ctx * c)
{
    if (!c || !c->group)
        return;
    std::lock_guard<std::mutex> lock(c->group->m);
    c->group->failed = true;
    c->group->cv.notify_all();
}

// How long a kernel of the device-side carrier waits for a peer before it gives up. The peer's host thread may lag by
// seconds for good reasons (page-locking a table region for the first time, code objects loaded lazily, device memory
// growing, a CPU quota, a profiler): generous by default, POPDEL_HIP_XCH_TIMEOUT_S (seconds, fractions allowed) for tests.
unsigned long long exchangeTimeoutTicks()
{
    static const unsigned long long ticks = [] {
        double seconds = 60.0;
        if (const char * t = getenv("POPDEL_HIP_XCH_TIMEOUT_S"))
        {
            const double v = atof(t);
            if (v > 0)
                seconds = v;
        }
        return static_cast<unsigned long long>(seconds * 1e8);
    }();
    return ticks;
}

struct CtxExchange : pd::Exchange
{
    pd_ctx & c;
    hipStream_t st;             // the stream the exchanges are ordered on (the context's compute stream by default)
    explicit CtxExchange(pd_ctx & ctx, hipStream_t stream = nullptr) : c(ctx), st(stream ? stream : ctx.stream) {}
    hipEvent_t nextEvent()
    {
        if (c.xchEventsUsed == c.xchEvents.size())
        {
            hipEvent_t e = nullptr;
            PD_HIP(hipEventCreate(&e));
            c.xchEvents.push_back(e);
        }
        return c.xchEvents[c.xchEventsUsed++];
    }
    void allGather(void * area, size_t bytesPerRank) override
    {
        if (!c.sharded)
            return;
        ++c.nAllGathers;
        c.allGatherBytes += static_cast<uint64_t>(bytesPerRank) * c.world;
        if (c.group && !c.groupDecided)
            groupDecide(c);
        if (c.group && c.flagCarrier)
        {
            // exchange.hip: ack / push / wait in ONE kernel on the stream, the host stays out of it
            if (c.world > pd::kXchMaxWorld)
                throw HipError{"the device-side carrier holds 16 contexts; set POPDEL_HIP_GROUP_CARRIER=host"};
            if (bytesPerRank % 16u != 0 || (static_cast<uint8_t *>(area) - c.arena) % 16 != 0)
                throw HipError{"exchange area not 16-byte aligned"};
            if (!c.peersEnabled)
            {
                for (uint32_t p = 0; p < c.world; ++p)
                {
                    const pd_ctx * peer = c.group->members[p];
                    if (!peer)
                        throw HipError{"a context of the local group is missing"};
                    if (peer->device != c.device)
                    {
                        const hipError_t e = hipDeviceEnablePeerAccess(peer->device, 0);
                        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                            throw HipError{std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)};
                        (void)hipGetLastError();
                    }
                }
                c.peersEnabled = true;
            }
            PD_HIP(hipEventRecord(nextEvent(), st));
            pd::FlagExchange x;
            std::memset(&x, 0, sizeof(x));
            x.myArena = c.arena;
            x.myFlags = c.xchFlags;
            for (uint32_t p = 0; p < c.world; ++p)
            {
                const pd_ctx * peer = c.group->members[p];
                x.peerArena[p] = peer->arena;          // (stable: the buffers are replaced in a collective step, reserveArena)
                x.peerFlags[p] = peer->xchFlags;
            }
            x.offset = static_cast<size_t>(static_cast<uint8_t *>(area) - c.arena);
            x.bytesPerRank = bytesPerRank;
            x.rank = c.rank;
            x.world = c.world;
            x.seq = ++c.xchSeq;
            x.timeoutTicks = exchangeTimeoutTicks();    // of the 100-MHz wall clock (60 s unless POPDEL_HIP_XCH_TIMEOUT_S says otherwise)
            x.err = c.xchErr;
            pd::launchFlagExchange(x, st);
            PD_HIP(hipGetLastError());
            if (c.xchDelayTicks)
                pd::launchSpin(c.xchDelayTicks, st);
            PD_HIP(hipEventRecord(nextEvent(), st));
            return;
        }
        if (c.group)
        {
            // every context pulls the other contexts' chunks out of their exchange buffers (same layout everywhere)
            const auto t0 = std::chrono::steady_clock::now();
            const size_t off = static_cast<size_t>(static_cast<uint8_t *>(area) - c.arena);
            PD_HIP(hipStreamSynchronize(st));                  // my chunk is written
            groupBarrier(c);
            for (uint32_t p = 0; p < c.world; ++p)
            {
                if (p == c.rank)
                    continue;
                const pd_ctx * peer = c.group->members[p];
                PD_HIP(hipMemcpyPeerAsync(c.arena + off + static_cast<size_t>(p) * bytesPerRank, c.device,
                                          peer->arena + off + static_cast<size_t>(p) * bytesPerRank, peer->device, bytesPerRank, st));
            }
            PD_HIP(hipStreamSynchronize(st));
            groupBarrier(c);                                   // nobody rewrites its chunk before everybody has copied it
            c.xchHostMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return;
        }
        PD_HIP(hipEventRecord(nextEvent(), st));
        if (c.comm)
        {
            // in place: chunk `rank` of [area, area + world * bytesPerRank) is filled; stream-ordered between the kernels
            uint8_t * base = static_cast<uint8_t *>(area);
            const ncclResult_t r = rccl().allGather(base + static_cast<size_t>(c.rank) * bytesPerRank, base, bytesPerRank, ncclUint8,
                                                    c.comm, st);
            if (r != ncclSuccess)
                throw HipError{std::string("ncclAllGather: ") + rccl().getErrorString(r)};
        }
        else if (hostOwnsArena(c))
        {
            const uint64_t off = static_cast<uint64_t>(static_cast<uint8_t *>(area) - c.arena);
            if (c.shard.all_gather(c.shard.user, off, bytesPerRank, st) != 0)
                throw HipError{"the host's all_gather() callback failed"};
        }
        // (world == 1 without a carrier: nothing to move)
        if (c.xchDelayTicks)
            pd::launchSpin(c.xchDelayTicks, st);
        PD_HIP(hipEventRecord(nextEvent(), st));
    }
};

// The device-side carrier reports a wait that gave up through a page-locked word; call after a synchronisation point.
void checkExchange(pd_ctx & c)
{
    if (c.xchErr && *c.xchErr != 0)
    {
        const unsigned code = *c.xchErr;
        *c.xchErr = 0;              // (reported once; the group is failed below: its exchange sequence is out of step from here on,
                                    // every member's next collective step says so instead of repeating this message)
        groupFail(&c);
        throw HipError{std::string("an exchange between the contexts of the local group timed out (waiting for a peer's ") +
                       (code == 1 ? "acknowledgement" : "chunk") + "): a peer failed, or -- several contexts on ONE GPU -- their streams share a hardware "
                       "queue (raise GPU_MAX_HW_QUEUES or POPDEL_HIP_XCH_TIMEOUT_S, or leave POPDEL_HIP_GROUP_CARRIER unset for the host-driven carrier)"};
    }
}

inline size_t alignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T>
T * mallocCopy(const std::vector<T> & v)
{
    T * p = static_cast<T *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!v.empty())
        std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

// Stage 2 adds the pairs of a cell up in chunk order, and the chunk order depends on the chunk life K: the contexts of a
// sharded cohort must all cut with the same K or their results differ in the last bits from the unsharded run (and with
// the number of shards). K is therefore taken from the lifetime histogram of the WHOLE cohort: the contexts all-gather
// their 66 counters when a segment is staged (on the copy stream; pd_segment_stage is collective for a sharded cohort).
void cohortLifeHist(pd_ctx & c, unsigned long long * hist /* 66 counters, in: own samples, out: cohort */)
{
    if (!c.sharded || c.world == 1)
        return;
    hipStream_t cs = c.copyStream;
    const size_t per = alignUp(66 * sizeof(unsigned long long), 256), words = per / sizeof(unsigned long long);
    uint8_t * area = reserveArena(c, per * c.world);
    c.hLifeXch.resize(words * c.world);
    std::memcpy(c.hLifeXch.data() + static_cast<size_t>(c.rank) * words, hist, 66 * sizeof(unsigned long long));
    PD_HIP(hipMemcpyAsync(area + static_cast<size_t>(c.rank) * per, c.hLifeXch.data() + static_cast<size_t>(c.rank) * words, per,
                          hipMemcpyHostToDevice, cs));
    CtxExchange xch(c, cs);
    xch.allGather(area, per);
    PD_HIP(hipMemcpyAsync(c.hLifeXch.data(), area, per * c.world, hipMemcpyDeviceToHost, cs));
    PD_HIP(hipStreamSynchronize(cs));
    checkExchange(c);
    for (int l = 0; l < 66; ++l)
    {
        unsigned long long sum = 0;
        for (uint32_t r = 0; r < c.world; ++r)
            sum += c.hLifeXch[static_cast<size_t>(r) * words + l];
        hist[l] = sum;
    }
}

// Split every pair instance into chunks that live at most K windows, order the chunks of each read group by their
// first window and index them by window (winStart). K is chosen so that nearly all pairs need a single chunk.
// `sized()` is called once the chunk totals are known (winStart is final then), `slabDone(lo, hi)` whenever the chunks
// of the read groups [lo, hi) are in place: the caller uploads slab by slab while the next slab is being built.
template <typename Sized, typename SlabDone>
void buildChunks(pd_ctx & c, const pd_segment & seg, StagedSegment & out, PinnedBuf<pd::PairDO> & pairs,
                 PinnedBuf<uint32_t> & start, std::vector<uint64_t> & rgBase, PinnedBuf<uint32_t> & winStart,
                 PinnedBuf<uint32_t> & pairsPk, PinnedBuf<uint32_t> & endCount, std::vector<uint8_t> & rgPacked,
                 std::vector<uint8_t> & scanGeneric, Sized sized, SlabDone slabDone)
{
    const uint32_t W = seg.n_windows;
    // lifetime histogram -> K
    std::vector<uint64_t> lifeHist(66, 0);
    uint64_t active = 0;
    {
        // per read group in parallel, merged afterwards
        std::vector<uint64_t> part(static_cast<size_t>(seg.n_readgroups) * 67, 0);
        pdh::parallelFor(seg.n_readgroups, [&](size_t rg) {
            uint64_t * h = part.data() + rg * 67;
            for (uint64_t i = seg.rg_offsets[rg]; i < seg.rg_offsets[rg + 1]; ++i)
            {
                uint32_t on = seg.pair_on[i], off = std::min(seg.pair_off[i], W);
                if (off <= on)
                    continue;
                uint32_t life = off - on;
                h[66] += life;
                ++h[std::min<uint32_t>(life, 65)];
            }
        }, 16);
        for (uint32_t rg = 0; rg < seg.n_readgroups; ++rg)
        {
            for (int k = 0; k < 66; ++k)
                lifeHist[k] += part[static_cast<size_t>(rg) * 67 + k];
            active += part[static_cast<size_t>(rg) * 67 + 66];
        }
    }
    uint64_t nInst = 0;
    for (uint64_t h : lifeHist)
        nInst += h;
    uint32_t K = 4;
    {
        // K from the lives of the whole cohort (the same for every context of a sharded one)
        unsigned long long cohort[66];
        for (int l = 0; l < 66; ++l)
            cohort[l] = lifeHist[l];
        cohortLifeHist(c, cohort);
        uint64_t all = 0;
        for (int l = 0; l < 66; ++l)
        {
            lifeHist[l] = cohort[l];
            all += cohort[l];
        }
        std::swap(all, nInst);                  // (nInst: cohort-wide inside this block; restored below)
        uint64_t acc = 0;
        // ~98.4 % of the pairs in one chunk. (A smaller share shortens the scan -- bench segment, K = 9 / 10 / 11 / 12: 1.127 /
        // 1.120 / 1.161 / 1.225 ms per pass; 95.1 % of the lives are <= 10 windows, 98.8 % <= 11 -- but where the
        // distribution is steep the contexts of a sharded cohort pick different K from their own samples, the chunk order
        // of stage 2 differs and with it the last bits of LR: tests/test_gpu_local_group.py caught exactly that with 15/16.)
        const uint64_t want = nInst - nInst / 64;
        for (uint32_t l = 0; l <= 65; ++l)
        {
            acc += lifeHist[l];
            if (acc >= want)
            {
                K = std::max<uint32_t>(4, l);
                break;
            }
        }
        K = std::min<uint32_t>(K, 48);
        nInst = all;
    }
    out.maxLife = K;
    out.nPairs = nInst;
    out.nActivePairs = active;

    // "Entities" below are the real read groups followed by the scan read groups of the samples that have several read
    // groups (SegmentView::scanRg): such an entity covers the instances of ALL read groups of its sample -- they are
    // contiguous in the caller's arrays -- and only gets the stage-1 image plus its winStart / endCount rows.
    const uint32_t nRg = seg.n_readgroups, nV = static_cast<uint32_t>(c.scanSample.size()), nEnt = nRg + nV;
    auto entLo = [&](size_t e) { return e < nRg ? seg.rg_offsets[e] : seg.rg_offsets[c.sampleRg[c.scanSample[e - nRg]]]; };
    auto entHi = [&](size_t e) { return e < nRg ? seg.rg_offsets[e + 1] : seg.rg_offsets[c.sampleRg[c.scanSample[e - nRg] + 1]]; };
    // pass 1 (parallel over entities): chunks per window -> winStart, chunk totals -> rgBase
    rgBase.assign(nEnt, 0);
    const size_t wsRow = static_cast<size_t>(W) + K;           // slots 0 .. W + K - 2, plus the end entry
    auto slotOf = [K](uint32_t a, uint32_t chunkOff) { return a == 0 ? std::min(chunkOff, K) - 1 : a + K - 1; };
    winStart.assignZero(static_cast<size_t>(nEnt) * wsRow);
    std::vector<uint64_t> rgChunks(nEnt, 0);
    pdh::parallelFor(nEnt, [&](size_t e) {
        const uint64_t lo = entLo(e), hi = entHi(e);
        uint32_t * ws = winStart.data() + e * wsRow;
        for (uint64_t i = lo; i < hi; ++i)
        {
            uint32_t on = seg.pair_on[i], off = std::min(seg.pair_off[i], W);
            for (uint32_t a = on; a < off; a += K)
                ++ws[slotOf(a, std::min(off, a + K)) + 1];   // counts, shifted by one for the prefix sum
        }
        uint32_t acc = 0;
        for (size_t x = 0; x < wsRow; ++x)
        {
            acc += ws[x];
            ws[x] = acc;
        }
        rgChunks[e] = acc;
    }, 8);
    uint64_t totalChunks = 0;
    for (uint32_t e = 0; e < nEnt; ++e)
    {
        if (e == nRg)
            out.nChunks = totalChunks;                         // real chunks: pairs[] / start[] end here
        rgBase[e] = totalChunks;
        totalChunks += rgChunks[e];
    }
    if (nV == 0)
        out.nChunks = totalChunks;
    {
        // densest 512-window block over all streams the scan kernel reads (it stages ws[w0] .. ws[min(W, w0 + 512) + K - 1])
        std::vector<uint32_t> entMax(nEnt, 0);
        pdh::parallelFor(nEnt, [&](size_t e) {
            const uint32_t * ws = winStart.data() + e * wsRow;
            uint32_t m = 0;
            for (uint32_t w0 = 0; w0 < W; w0 += 2 * pd::kStage1Block)
            {
                const uint32_t whi = std::min<uint32_t>(W, w0 + 2 * pd::kStage1Block);
                m = std::max(m, ws[whi + K - 1] - ws[w0]);
            }
            entMax[e] = m;
        }, 64);
        // ... of 99 % of the streams: a few unusually deep samples should not cost everybody else occupancy (their
        // blocks go to the fallback kernel instead)
        std::sort(entMax.begin(), entMax.end());
        out.maxBlockChunks = entMax.empty() ? 0 : entMax[std::min(entMax.size() - 1, entMax.size() * 99 / 100)];
    }
    pairs.resize(out.nChunks);
    start.resize(out.nChunks);
    pairsPk.resize(totalChunks);
    endCount.assignZero(static_cast<size_t>(nEnt) * (W + 1));
    rgPacked.assign(nEnt, 0);
    sized();
    // pass 2 (parallel): scatter the chunks to their slots, stable within a window; in slabs of entities
    const size_t nSlabs = std::min<size_t>(8, std::max<size_t>(1, nRg / 64));
    for (size_t slab = 0; slab <= nSlabs; ++slab)
    {
        // slabs 0 .. nSlabs-1: the real read groups; the last one: the scan read groups
        const size_t slabLo = slab < nSlabs ? nRg * slab / nSlabs : nRg;
        const size_t slabHi = slab < nSlabs ? nRg * (slab + 1) / nSlabs : nEnt;
        if (slabHi == slabLo)
            continue;
        pdh::parallelFor(slabHi - slabLo, [&](size_t inSlab) {
            const size_t e = slabLo + inSlab;
            const bool real = e < nRg;
            const uint64_t lo = entLo(e), hi = entHi(e);
            const uint32_t * ws = winStart.data() + e * wsRow;
            std::vector<uint32_t> cursor(ws, ws + wsRow - 1);
            const size_t base = rgBase[e];
            uint32_t * ec = endCount.data() + e * (W + 1);
            bool fits = W <= 0xffffu;
            for (uint64_t i = lo; i < hi; ++i)
            {
                uint32_t on = seg.pair_on[i], off = std::min(seg.pair_off[i], W);
                const int32_t dv = seg.pair_dev[i];
                fits = fits && dv >= -32767 && dv <= 32766;
                const uint32_t d16 = (static_cast<uint32_t>(dv) & 0xffffu) ^ 0x8000u;     // biased: see stage1.hip
                for (uint32_t a = on; a < off; a += K)
                {
                    const uint32_t chunkOff = std::min(off, a + K);
                    const uint32_t slot = cursor[slotOf(a, chunkOff)]++;
                    if (real)
                    {
                        pairs[base + slot] = pd::PairDO{dv, chunkOff};
                        start[base + slot] = seg.pair_start[i];
                    }
                    pairsPk[base + slot] = d16 | ((chunkOff & 0xffffu) << 16);
                    ++ec[chunkOff];
                }
            }
            uint32_t acc = 0;
            for (uint32_t w = 0; w <= W; ++w)
            {
                acc += ec[w];
                ec[w] = acc;
            }
            rgPacked[e] = fits ? 1 : 0;
        }, 8);
        slabDone(slabLo, slabHi);
    }
    // where a real read group of a multi-read-group sample reaches its maxLoad, the reference drops that read group's
    // pairs: those 512-window blocks of the sample go to the fallback kernel (it knows the read groups)
    const size_t blocksPerSample = (static_cast<size_t>(W) + 2 * pd::kStage1Block - 1) / (2 * pd::kStage1Block);
    scanGeneric.assign(static_cast<size_t>(nV) * blocksPerSample, 0);
    pdh::parallelFor(nV, [&](size_t j) {
        const uint32_t sIdx = c.scanSample[j];
        for (uint32_t rg = c.sampleRg[sIdx]; rg < c.sampleRg[sIdx + 1]; ++rg)
        {
            const uint32_t * ws = winStart.data() + static_cast<size_t>(rg) * wsRow;
            const uint32_t * ec = endCount.data() + static_cast<size_t>(rg) * (W + 1);
            for (uint32_t w = 0; w < W; ++w)
                if (ws[w + K] - ec[w] >= c.maxLoad[rg])
                    scanGeneric[j * blocksPerSample + w / (2 * pd::kStage1Block)] = 1;
        }
    }, 4);
}

// The staged layout from pair instances that are in device memory already (uploaded, or derived there: walk.hip), on the
// copy stream. `onUploaded` (optional) runs after the first synchronisation point: everything enqueued before the call
// has completed then.
bool buildLayoutOnDevice(pd_ctx & c, const pd::InstanceArrays & in, uint64_t nInst, uint32_t W, StagedSegment & out,
                         const std::function<void()> & onUploaded)
{
    const uint32_t nRg = c.nRg, nV = static_cast<uint32_t>(c.scanSample.size()), nEnt = nRg + nV;
    hipStream_t cs = c.copyStream;
    pd::BuildEntities en;
    en.rgOffsets = c.dRgOffsets.p;
    en.sampleRg = c.dSampleRg.p;
    en.scanSample = c.dScanSample.p;
    en.nRg = nRg;
    en.nScan = nV;
    // ---- K from the lifetime histogram (the rule of the host builder)
    c.dLifeHist.ensure(67);
    c.hLifeHist.resize(67);
    PD_HIP(hipMemsetAsync(c.dLifeHist.p, 0, 67 * sizeof(unsigned long long), cs));
    pd::launchLifeStats(in, nInst, W, c.dLifeHist.p, cs);
    PD_HIP(hipGetLastError());
    PD_HIP(hipMemcpyAsync(c.hLifeHist.data(), c.dLifeHist.p, 67 * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
    PD_HIP(hipStreamSynchronize(cs));
    if (onUploaded)
        onUploaded();
    uint64_t living = 0;
    for (int l = 0; l <= 65; ++l)
        living += c.hLifeHist[l];
    // Chunk lives. Stage 2 (K): ~98.4 % of the pairs in one chunk, the rule of the host builder -- its chunk order is the
    // order stage 2 adds pairs up in. Stage-1 image (K1 <= K): 15/16 in one chunk -- fewer slots per cell of the scan
    // (bench segment, scan per pass with 9 / 10 / 11 / 12: 1.127 / 1.120 / 1.161 / 1.225 ms; 95.1 % of the lives are
    // <= 10 windows, 98.8 % <= 11); Q3 does not depend on how the pairs are cut, so every context may pick its own.
    auto lifeAt = [&](uint64_t tail) {
        uint64_t acc = 0;
        const uint64_t want = living - living / tail;
        for (uint32_t l = 0; l <= 65; ++l)
        {
            acc += c.hLifeHist[l];
            if (acc >= want)
                return std::min<uint32_t>(std::max<uint32_t>(4, l), 48);
        }
        return 48u;
    };
    // (K1 from the own samples' lives, then K from the whole cohort's: see cohortLifeHist)
    const uint32_t K1own = lifeAt(16);
    const unsigned long long ownActive = c.hLifeHist[66];
    cohortLifeHist(c, c.hLifeHist.data());
    uint64_t ownLiving = living;
    living = 0;
    for (int l = 0; l <= 65; ++l)
        living += c.hLifeHist[l];
    uint32_t K = lifeAt(64), K1 = std::min(K1own, K);
    living = ownLiving;
    if (const char * e = getenv("POPDEL_HIP_MAXLIFE_SCAN"))
        K1 = std::min<uint32_t>(K, std::max(1, atoi(e)));
    if (pd::buildLdsBytes(W, K) > 160u * 1024u)
        return false;
    const bool twoLives = K1 != K;
    out.maxLife = K;
    out.maxLifePk = K1;
    out.ownPkIndex = twoLives;
    out.nPairs = living;
    out.nActivePairs = ownActive;
    // ---- counts -> index rows, chunk totals -> bases (per image if the lives differ)
    out.winStart.ensure(static_cast<size_t>(twoLives ? nRg : nEnt) * (static_cast<size_t>(W) + K));
    out.endCount.ensure(static_cast<size_t>(twoLives ? nRg : nEnt) * (W + 1));
    out.rgBase.ensure(nEnt);
    out.rgPacked.ensure(nEnt);
    c.dEntChunks.ensure(2 * static_cast<size_t>(nEnt));
    c.dEntMax.ensure(2 * static_cast<size_t>(nEnt));
    c.dTotals.ensure(4);
    c.dProblems.ensure(1);
    c.hTotals.resize(5);
    PD_HIP(hipMemsetAsync(c.dProblems.p, 0, sizeof(unsigned), cs));
    pd::BuildEntities enReal = en;
    enReal.nScan = 0;
    if (twoLives)
    {
        out.winStartPk.ensure(static_cast<size_t>(nEnt) * (static_cast<size_t>(W) + K1));
        out.endCountPk.ensure(static_cast<size_t>(nEnt) * (W + 1));
        out.rgBasePk.ensure(nEnt);
    }
    {
        // stage-2 arrays (real read groups; all entities if there is one life only)
        pd::CountOut co;
        co.winStart = out.winStart.p;
        co.endCount = out.endCount.p;
        co.entChunks = c.dEntChunks.p;
        co.entMax = c.dEntMax.p;
        co.rgPacked = out.rgPacked.p;
        co.problems = c.dProblems.p;
        if (!pd::launchCount(in, twoLives ? enReal : en, W, K, co, cs))
            throw HipError{"the layout build kernel cannot get its LDS"};
        PD_HIP(hipGetLastError());
        pd::launchBaseScan(c.dEntChunks.p, twoLives ? nRg : nEnt, nRg, out.rgBase.p, c.dTotals.p, cs);
        PD_HIP(hipGetLastError());
    }
    if (twoLives)
    {
        pd::CountOut co;
        co.winStart = out.winStartPk.p;
        co.endCount = out.endCountPk.p;
        co.entChunks = c.dEntChunks.p + nEnt;
        co.entMax = c.dEntMax.p + nEnt;
        co.rgPacked = out.rgPacked.p;
        co.problems = c.dProblems.p;
        if (!pd::launchCount(in, en, W, K1, co, cs))
            throw HipError{"the layout build kernel cannot get its LDS"};
        PD_HIP(hipGetLastError());
        pd::launchBaseScan(c.dEntChunks.p + nEnt, nEnt, nRg, out.rgBasePk.p, c.dTotals.p + 2, cs);
        PD_HIP(hipGetLastError());
    }
    c.hEntMax.resize(nEnt);
    PD_HIP(hipMemcpyAsync(c.hTotals.data(), c.dTotals.p, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, cs));
    PD_HIP(hipMemcpyAsync(c.hTotals.data() + 4, c.dProblems.p, sizeof(unsigned), hipMemcpyDeviceToHost, cs));
    PD_HIP(hipMemcpyAsync(c.hEntMax.data(), c.dEntMax.p + (twoLives ? nEnt : 0), nEnt * sizeof(uint32_t), hipMemcpyDeviceToHost, cs));
    PD_HIP(hipStreamSynchronize(cs));
    c.stageTiming.counted_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - c.stageBegan).count();
    if (static_cast<unsigned>(c.hTotals[4] & 0xffffffffu) != 0)
        throw HipError{"pair instances of a read group are not ordered by their first window (pd_segment: non-decreasing `on`)"};
    const uint64_t totalReal = c.hTotals[0], totalPk = twoLives ? c.hTotals[3] : c.hTotals[1];
    out.nChunks = totalReal;
    {
        // ... of 99 % of the streams: a few unusually deep samples should not cost everybody else occupancy
        std::vector<uint32_t> entMax(c.hEntMax.data(), c.hEntMax.data() + nEnt);
        std::sort(entMax.begin(), entMax.end());
        out.maxBlockChunks = entMax.empty() ? 0 : entMax[std::min(entMax.size() - 1, entMax.size() * 99 / 100)];
    }
    // ---- chunks to their places
    const bool trace = getenv("POPDEL_HIP_STAGE_TRACE") != nullptr;
    auto mark = [&](const char * what) {
        if (trace)
            fprintf(stderr, "[popdel_hip stage] %7.2f ms  %s\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c.stageBegan).count(), what);
    };
    mark("totals read");
    out.pairs.ensure(totalReal);
    out.start.ensure(totalReal);
    out.pairsPk.ensure(totalPk + 256);             // the scan's LDS-DMA copies whole 256-chunk pieces (stage1.hip)
    mark("chunk arrays ensured");
    {
        pd::ScatterOut so;
        so.pairs = out.pairs.p;
        so.start = out.start.p;
        so.pairsPk = twoLives ? nullptr : out.pairsPk.p;
        so.rgBase = out.rgBase.p;
        so.winStart = out.winStart.p;
        if (!pd::launchScatter(in, twoLives ? enReal : en, W, K, so, cs))
            throw HipError{"the layout scatter kernel cannot get its LDS"};
        PD_HIP(hipGetLastError());
    }
    if (twoLives)
    {
        pd::ScatterOut so;
        so.pairs = nullptr;
        so.start = nullptr;
        so.pairsPk = out.pairsPk.p;
        so.rgBase = out.rgBasePk.p;
        so.winStart = out.winStartPk.p;
        if (!pd::launchScatter(in, en, W, K1, so, cs))
            throw HipError{"the layout scatter kernel cannot get its LDS"};
        PD_HIP(hipGetLastError());
    }
    const size_t blocksPerSample = (static_cast<size_t>(W) + 2 * pd::kStage1Block - 1) / (2 * pd::kStage1Block);
    out.hasScanGeneric = nV != 0;
    if (nV != 0)
    {
        out.scanGeneric.ensure(static_cast<size_t>(nV) * blocksPerSample);
        PD_HIP(hipMemsetAsync(out.scanGeneric.p, 0, static_cast<size_t>(nV) * blocksPerSample, cs));
        pd::launchScanGeneric(en, c.dRg.p, out.winStart.p, out.endCount.p, W, K, out.scanGeneric.p, static_cast<unsigned>(blocksPerSample), cs);
        PD_HIP(hipGetLastError());
    }
    mark("scatter launched");
    // the screen's one byte per cell (after the index of the stage-1 image is final)
    out.active8.ensure(static_cast<size_t>(nEnt) * W);
    pd::launchActive8(out.ownPkIndex ? out.winStartPk.p : out.winStart.p, out.ownPkIndex ? out.endCountPk.p : out.endCount.p, nEnt, W,
                      out.ownPkIndex ? out.maxLifePk : out.maxLife, out.active8.p, cs);
    PD_HIP(hipGetLastError());
    mark("active8 launched");
    return true;
}

// The same layout built on the device (build.hip) from instances that cross PCIe once: 8 bytes per instance when every
// field fits 16 bits (deviation, window indices, distance of the pair start behind its first window), else 16.
// Returns false if this segment has to take the host builder (window count beyond what the build kernels keep in LDS).
bool buildChunksOnDevice(pd_ctx & c, const pd_segment & seg, StagedSegment & out)
{
    const uint32_t W = seg.n_windows, nRg = seg.n_readgroups;
    const uint64_t nInst = seg.rg_offsets[nRg];
    if (pd::buildLdsBytes(W, 48) > 160u * 1024u || nInst == 0)
        return false;
    hipStream_t cs = c.copyStream;
    // ---- narrow or wide? decided while packing; a segment that does not fit is packed again, wide
    const bool narrowPossible = W <= 0xffffu;
    std::atomic<bool> narrow(narrowPossible);
    const size_t nSlabs = std::min<size_t>(8, std::max<size_t>(1, nRg / 64));
    c.hRgOffsets.resize(nRg + 1);
    std::memcpy(c.hRgOffsets.data(), seg.rg_offsets, (nRg + 1) * sizeof(uint64_t));
    c.dRgOffsets.uploadPinned(c.hRgOffsets, cs);
    // Narrow upload, two flavours. Compact (5 bytes per instance): the deviation in 16 bits; `on` as the difference to the
    // read group's previous instance, the life off - on and the start distance in one byte each (255: see the exception
    // list) -- expanded to the narrow arrays on the device (pd::launchExpand). Plain (8 bytes): four 16-bit fields.
    // The compact flavour needs the instances of a read group in `on` order and few exceptions; else the plain one is
    // packed (which reports instances out of order through the build kernels), else the wide one.
    bool compact = narrowPossible && nInst < 0xffffffffull && c.compactUpload;
    const bool debug = getenv("POPDEL_HIP_DEBUG") != nullptr;
    const auto tBuild0 = std::chrono::steady_clock::now();
    double packSeconds = 0;
    std::vector<std::array<uint32_t, 4>> exceptions;             // (instance, on difference, life, start distance)
    auto packNarrow = [&](bool asCompact) -> bool {
        std::atomic<bool> ok(true);
        std::mutex excLock;
        exceptions.clear();
        for (int k = 0; k < 4; ++k)
        {
            if (!asCompact || k == 0)
                c.hIn16[k].resize(nInst);
            c.dIn16[k].ensure(nInst);
        }
        if (asCompact)
            for (int k = 0; k < 3; ++k)
            {
                c.hIn8[k].resize(nInst);
                c.dIn8[k].ensure(nInst);
            }
        for (size_t slab = 0; slab < nSlabs && ok.load(); ++slab)
        {
            const size_t rgLo = nRg * slab / nSlabs, rgHi = nRg * (slab + 1) / nSlabs;
            const auto tPack0 = std::chrono::steady_clock::now();
            pdh::parallelFor(rgHi - rgLo, [&](size_t k) {
                const size_t rg = rgLo + k;
                const uint64_t lo = seg.rg_offsets[rg], n = seg.rg_offsets[rg + 1] - lo;
                // plain loops over restrict pointers, no early exit: the compiler vectorises them (AVX2 on the host)
                const int32_t * __restrict sd = seg.pair_dev + lo;
                const uint32_t * __restrict so = seg.pair_on + lo;
                const uint32_t * __restrict sf = seg.pair_off + lo;
                const uint32_t * __restrict ss = seg.pair_start + lo;
                int16_t * __restrict dv = reinterpret_cast<int16_t *>(c.hIn16[0].data()) + lo;
                const uint32_t first = seg.first_window_pos, shift = seg.window_shift;
                uint32_t bad = 0;
                for (uint64_t i = 0; i < n; ++i)
                {
                    const int32_t d = sd[i];
                    bad |= static_cast<uint32_t>(d + 32768) >> 16;
                    dv[i] = static_cast<int16_t>(d);
                }
                if (asCompact)
                {
                    uint8_t * __restrict dOn = c.hIn8[0].data() + lo;
                    uint8_t * __restrict life = c.hIn8[1].data() + lo;
                    uint8_t * __restrict back8 = c.hIn8[2].data() + lo;
                    uint32_t special = 0;
                    for (uint64_t i = 0; i < n; ++i)
                    {
                        // first + on * shift - start in [0, 65535]; 64-bit only where the 32-bit value could wrap
                        const uint32_t dist = first + so[i] * shift - ss[i];
                        bad |= dist >> 16;
                        special |= dist >= 255u;
                        back8[i] = static_cast<uint8_t>(dist < 255u ? dist : 255u);
                    }
                    if (n != 0)
                    {
                        bad |= so[0] >> 16;
                        special |= so[0] >= 255u;
                        dOn[0] = static_cast<uint8_t>(so[0] < 255u ? so[0] : 255u);
                    }
                    for (uint64_t i = 1; i < n; ++i)
                    {
                        const uint32_t a = so[i], d = a - so[i - 1];             // wraps if out of order: caught as "bad"
                        bad |= (a >> 16) | (d >> 16);
                        special |= d >= 255u;
                        dOn[i] = static_cast<uint8_t>(d < 255u ? d : 255u);
                    }
                    for (uint64_t i = 0; i < n; ++i)
                    {
                        const uint32_t e = sf[i] < W ? sf[i] : W, l = e > so[i] ? e - so[i] : 0u;
                        special |= l >= 255u;
                        life[i] = static_cast<uint8_t>(l < 255u ? l : 255u);
                    }
                    if (special && !bad)
                    {
                        std::vector<std::array<uint32_t, 4>> mine;
                        auto look = [&](uint64_t i) {
                            if (dOn[i] == 255u || life[i] == 255u || back8[i] == 255u)
                            {
                                const uint32_t e = sf[i] < W ? sf[i] : W;
                                mine.push_back({static_cast<uint32_t>(lo + i), i ? so[i] - so[i - 1] : so[0], e > so[i] ? e - so[i] : 0u,
                                                first + so[i] * shift - ss[i]});
                            }
                        };
                        // eight instances per step: does any of the 24 bytes equal 255? (most read groups have a handful)
                        auto anyFF = [](const uint8_t * p) {
                            uint64_t v;
                            std::memcpy(&v, p, 8);
                            v = ~v;
                            return ((v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull) != 0;
                        };
                        uint64_t i = 0;
                        for (; i + 8 <= n; i += 8)
                            if (anyFF(dOn + i) || anyFF(life + i) || anyFF(back8 + i))
                                for (uint64_t j = i; j < i + 8; ++j)
                                    look(j);
                        for (; i < n; ++i)
                            look(i);
                        std::lock_guard<std::mutex> hold(excLock);
                        exceptions.insert(exceptions.end(), mine.begin(), mine.end());
                    }
                }
                else
                {
                    uint16_t * __restrict on = c.hIn16[1].data() + lo;
                    uint16_t * __restrict off = c.hIn16[2].data() + lo;
                    uint16_t * __restrict back = c.hIn16[3].data() + lo;
                    for (uint64_t i = 0; i < n; ++i)
                    {
                        const uint32_t dist = first + so[i] * shift - ss[i];
                        bad |= dist >> 16;
                        back[i] = static_cast<uint16_t>(dist);
                    }
                    for (uint64_t i = 0; i < n; ++i)
                    {
                        const uint32_t a = so[i];
                        bad |= a >> 16;
                        on[i] = static_cast<uint16_t>(a);
                    }
                    for (uint64_t i = 0; i < n; ++i)
                        off[i] = static_cast<uint16_t>(sf[i] < W ? sf[i] : W);
                }
                // (window indices are < 2^16 here and positions < 2^32: the product and sum stay below 2^32 for any real
                // segment; a wrap would need first + on * shift >= 2^32, which the 64-bit check below excludes)
                if (n != 0 && static_cast<uint64_t>(first) + static_cast<uint64_t>(so[n - 1] & 0xffffu) * shift > 0xffffffffull)
                    bad = 1;
                if (bad)
                    ok.store(false);
            }, 8);
            packSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tPack0).count();
            if (!ok.load() || (asCompact && exceptions.size() > nInst / 32 + 1024))
                return false;
            const uint64_t a = seg.rg_offsets[rgLo], b = seg.rg_offsets[rgHi];
            if (b > a)
            {
                PD_HIP(hipMemcpyAsync(c.dIn16[0].p + a, c.hIn16[0].data() + a, (b - a) * sizeof(uint16_t), hipMemcpyHostToDevice, cs));
                if (asCompact)
                    for (int k = 0; k < 3; ++k)
                        PD_HIP(hipMemcpyAsync(c.dIn8[k].p + a, c.hIn8[k].data() + a, b - a, hipMemcpyHostToDevice, cs));
                else
                    for (int k = 1; k < 4; ++k)
                        PD_HIP(hipMemcpyAsync(c.dIn16[k].p + a, c.hIn16[k].data() + a, (b - a) * sizeof(uint16_t), hipMemcpyHostToDevice, cs));
            }
        }
        return true;
    };
    if (narrowPossible)
    {
        if (compact && !packNarrow(true))
            compact = false;
        if (!compact && !packNarrow(false))
            narrow.store(false);
        if (compact)
        {
            // the exception list (ascending instance index), then the expansion into the narrow arrays
            std::sort(exceptions.begin(), exceptions.end());
            const size_t nExc = exceptions.size();
            for (int k = 0; k < 4; ++k)
            {
                c.hExc[k].resize(nExc + 1);
                for (size_t i = 0; i < nExc; ++i)
                    c.hExc[k][i] = exceptions[i][k];
                c.hExc[k][nExc] = 0;
                c.dExc[k].uploadPinned(c.hExc[k], cs);
            }
            c.hExc[4].resize(nRg + 1);                          // first exception of every read group
            for (size_t rg = 0, e = 0; rg <= nRg; ++rg)
            {
                while (e < nExc && exceptions[e][0] < seg.rg_offsets[rg])
                    ++e;
                c.hExc[4][rg] = static_cast<uint32_t>(e);
            }
            c.dExc[4].uploadPinned(c.hExc[4], cs);
            pd::CompactInstances ci;
            ci.onDelta = c.dIn8[0].p;
            ci.life = c.dIn8[1].p;
            ci.back = c.dIn8[2].p;
            ci.excRgStart = c.dExc[4].p;
            ci.excIndex = c.dExc[0].p;
            ci.excOnDelta = c.dExc[1].p;
            ci.excLife = c.dExc[2].p;
            ci.excBack = c.dExc[3].p;
            pd::launchExpand(ci, c.dRgOffsets.p, nRg, W, c.dIn16[1].p, c.dIn16[2].p, c.dIn16[3].p, cs);
            PD_HIP(hipGetLastError());
        }
    }
    else
        narrow.store(false);
    if (!narrow.load())
    {
        const void * src[4] = {seg.pair_dev, seg.pair_on, seg.pair_off, seg.pair_start};
        for (int k = 0; k < 4; ++k)
        {
            c.hIn32[k].resize(nInst);
            c.dIn32[k].ensure(nInst);
        }
        for (size_t slab = 0; slab < nSlabs; ++slab)
        {
            const size_t rgLo = nRg * slab / nSlabs, rgHi = nRg * (slab + 1) / nSlabs;
            const uint64_t a = seg.rg_offsets[rgLo], b = seg.rg_offsets[rgHi];
            if (b == a)
                continue;
            // page-locked staging copy, split over the host threads
            const size_t pieces = 64;
            pdh::parallelFor(pieces * 4, [&](size_t j) {
                const int k = static_cast<int>(j / pieces);
                const uint64_t lo = a + (b - a) * (j % pieces) / pieces, hi = a + (b - a) * (j % pieces + 1) / pieces;
                std::memcpy(c.hIn32[k].data() + lo, static_cast<const uint32_t *>(src[k]) + lo, (hi - lo) * sizeof(uint32_t));
            });
            for (int k = 0; k < 4; ++k)
                PD_HIP(hipMemcpyAsync(c.dIn32[k].p + a, c.hIn32[k].data() + a, (b - a) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        }
    }
    pd::InstanceArrays in;
    std::memset(&in, 0, sizeof(in));
    in.wide = narrow.load() ? 0 : 1;
    in.dev16 = reinterpret_cast<const int16_t *>(c.dIn16[0].p);
    in.on16 = c.dIn16[1].p;
    in.off16 = c.dIn16[2].p;
    in.back16 = c.dIn16[3].p;
    in.dev32 = reinterpret_cast<const int32_t *>(c.dIn32[0].p);
    in.on32 = c.dIn32[1].p;
    in.off32 = c.dIn32[2].p;
    in.start32 = c.dIn32[3].p;
    in.firstPos = seg.first_window_pos;
    in.shift = seg.window_shift;
    const bool isNarrow = narrow.load();
    return buildLayoutOnDevice(c, in, nInst, W, out, [&]() {
        c.stageTiming.n_instances = nInst;
        c.stageTiming.flavour = isNarrow ? (compact ? 5u : 8u) : 16u;
        c.stageTiming.bytes_uploaded = nInst * c.stageTiming.flavour + (nRg + 1) * sizeof(uint64_t) + exceptions.size() * 16u;
        c.stageTiming.pack_seconds = packSeconds;
        c.stageTiming.uploaded_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild0).count();
        c.stageTiming.host_threads = pdh::hostThreads();
        if (debug)
            fprintf(stderr, "[popdel_hip] device build: %llu instances uploaded %s, %zu exceptions; packing %.1f ms on the host threads, "
                            "%.1f ms until the last byte was on the device\n", (unsigned long long)nInst,
                    isNarrow ? (compact ? "compact (5 B)" : "narrow (8 B)") : "wide (16 B)", exceptions.size(), packSeconds * 1e3,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild0).count() * 1e3);
    });
}

pd::SegmentView viewOf(const StagedSegment & s)
{
    pd::SegmentView v;
    v.pairs = s.pairs.p;
    v.start = s.start.p;
    v.rgBase = s.rgBase.p;
    v.winStart = s.winStart.p;
    v.pairsPk = s.pairsPk.p;
    v.endCount = s.endCount.p;
    v.rgBasePk = s.ownPkIndex ? s.rgBasePk.p : s.rgBase.p;
    v.winStartPk = s.ownPkIndex ? s.winStartPk.p : s.winStart.p;
    v.endCountPk = s.ownPkIndex ? s.endCountPk.p : s.endCount.p;
    v.maxLifePk = s.ownPkIndex ? s.maxLifePk : s.maxLife;
    v.active8 = s.active8.p;
    v.rgPacked = s.rgPacked.p;
    v.scanRg = s.scanRg;
    v.scanGeneric = s.hasScanGeneric ? s.scanGeneric.p : nullptr;
    v.nRealRg = s.nRealRg;
    v.nWindows = s.nWindows;
    v.maxLife = s.maxLife;
    v.firstPos = s.firstPos;
    v.shift = s.shift;
    return v;
}

int fail(pd_ctx * c, const std::string & msg)
{
    if (c)
        c->error = msg;
    else
        g_createError = msg;
    return 1;
}

}  // namespace

extern "C" {

static int createContext(const pd_config * cfg, const pd_readgroup * rgs, uint32_t n_readgroups, const uint32_t * sample_rg_offsets,
                         uint32_t n_samples, int device, pd_ctx ** out);
static void laneLoop(pd_ctx * parent, unsigned k);

int pd_ctx_create(const pd_config * cfg, const pd_readgroup * rgs, uint32_t n_readgroups, const uint32_t * sample_rg_offsets,
                  uint32_t n_samples, int device, pd_ctx ** out)
{
    if (int rc = createContext(cfg, rgs, n_readgroups, sample_rg_offsets, n_samples, device, out))
        return rc;
    // K lanes behind this context (LanesState): three by default -- measured on 10,000 samples x 6,667 windows: 2.28 / 2.08 /
    // 1.92 / 2.00 ms per pass with 1 / 2 / 3 / 4 lanes (profiles/r04_b_lanes_probe.json); POPDEL_HIP_LANES = 1 gives the plain
    // context of rounds 1-3
    unsigned k = 3;
    if (const char * e = getenv("POPDEL_HIP_LANES"))
        k = static_cast<unsigned>(std::max(1, std::min(8, atoi(e))));
    if (k < 2)
        return 0;
    pd_ctx * c = *out;
    c->lanes.reset(new LanesState());
    for (unsigned i = 0; i < k; ++i)
    {
        pd_ctx * lane = nullptr;
        if (int rc = createContext(cfg, rgs, n_readgroups, sample_rg_offsets, n_samples, device, &lane))
        {
            pd_ctx_destroy(c);
            *out = nullptr;
            return rc;
        }
        lane->owner = c;
        c->lanes->workers.emplace_back(new LaneWorker());
        c->lanes->workers.back()->ctx = lane;
    }
    c->lanes->active = k;
    // (measured and dropped: the lanes' scans one after the other on ONE shared low-priority stream instead of each on its
    // lane's stream -- 2.32 ms per pass against 1.92 with three lanes, profiles/r04_c_lanes_probe.json)
    for (unsigned i = 0; i < k; ++i)
        c->lanes->workers[i]->thread = std::thread(laneLoop, c, i);
    return 0;
}

static int createContext(const pd_config * cfg, const pd_readgroup * rgs, uint32_t n_readgroups, const uint32_t * sample_rg_offsets,
                         uint32_t n_samples, int device, pd_ctx ** out)
{
    if (!cfg || !rgs || !sample_rg_offsets || !out || n_readgroups == 0 || n_samples == 0)
        return fail(nullptr, "pd_ctx_create: invalid argument");
    if (sample_rg_offsets[0] != 0 || sample_rg_offsets[n_samples] != n_readgroups)
        return fail(nullptr, "pd_ctx_create: sample_rg_offsets do not cover the read groups");
    for (uint32_t i = 0; i < n_samples; ++i)
        if (sample_rg_offsets[i + 1] <= sample_rg_offsets[i])
            return fail(nullptr, "pd_ctx_create: every sample needs at least one read group");
    if (cfg->iterations > (uint32_t)pd::kMaxIterationsCap)
        return fail(nullptr, "pd_ctx_create: iterations exceeds the device limit of 64");
    if (n_samples > 65535u)
        return fail(nullptr, "pd_ctx_create: more than 65535 samples per context (shard the cohort across contexts)");
    std::unique_ptr<pd_ctx> c(new pd_ctx());
    try
    {
        int nDev = 0;
        PD_HIP(hipGetDeviceCount(&nDev));
        if (nDev <= 0 || device < 0 || device >= nDev)
            return fail(nullptr, "pd_ctx_create: no such HIP device");
        PD_HIP(hipSetDevice(device));
        c->device = device;
        c->cfg = *cfg;
        c->nRg = n_readgroups;
        c->nSamples = n_samples;
        c->sampleRg.assign(sample_rg_offsets, sample_rg_offsets + n_samples + 1);
        {
            const char * e = getenv("POPDEL_HIP_SCAN_OVERLAP");
            std::string how = e ? e : "0";
            int lowest = 0, highest = 0;
            PD_HIP(hipDeviceGetStreamPriorityRange(&lowest, &highest));
            if (how != "0" && !how.empty())
            {
                PD_HIP(hipStreamCreateWithPriority(&c->stream, hipStreamDefault, highest));
                PD_HIP(hipStreamCreateWithPriority(&c->scanStream, hipStreamDefault, lowest));
                int nCu = 0;
                PD_HIP(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, device));
                c->scanRoom.nCu = nCu > 0 ? (unsigned)nCu : 256u;
                const unsigned n = how.size() > 1 ? (unsigned)atoi(how.c_str() + 1) : 7u;
                c->scanRoom.wgPerCu = std::max(1u, std::min(7u, n ? n : 7u));
                c->scanRoom.mode = how[0] == 'p' ? pd::ScanRoom::kPersistent : (how[0] == 'l' ? pd::ScanRoom::kLdsCap : pd::ScanRoom::kFull);
                c->dScanCounter.alloc(2);
            }
            else
                PD_HIP(hipStreamCreate(&c->stream));
        }
        if (const char * e = getenv("POPDEL_HIP_XCH_DELAY_US"))
        {
            int khz = 0;
            PD_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
            c->xchDelayTicks = static_cast<unsigned long long>(std::max(0.0, atof(e)) * (khz > 0 ? khz : 100000) / 1000.0);
        }
        PD_HIP(hipStreamCreate(&c->copyStream));
        PD_HIP(hipEventCreateWithFlags(&c->resultsPacked, hipEventDisableTiming));
        PD_HIP(hipEventCreateWithFlags(&c->cellsFetched, hipEventDisableTiming));
        std::vector<pd::RgConst> rc(n_readgroups);
        std::vector<double> pool;
        int minThreshold = 0x7fffffff;
        for (uint32_t i = 0; i < n_readgroups; ++i)
        {
            rc[i].median = (int)rgs[i].median;
            rc[i].readLength = (int)rgs[i].read_length;
            rc[i].histOffset = rgs[i].hist_offset;
            rc[i].histLen = (int)rgs[i].hist_len;
            rc[i].lowerQ = (int)rgs[i].lower_quantile_dist;
            rc[i].upperQ = (int)rgs[i].upper_quantile_dist;
            rc[i].maxLoad = rgs[i].max_load;
            rc[i].minInitDelLen = rgs[i].min_init_del_len;
            rc[i].stddev = rgs[i].stddev;
            rc[i].minProb = rgs[i].min_prob;
            rc[i].histBase = pool.size();
            pool.insert(pool.end(), rgs[i].hist_values, rgs[i].hist_values + rgs[i].hist_len);
            minThreshold = std::min(minThreshold, (int)rgs[i].min_init_del_len);
        }
        // scan read groups of the samples that have several read groups (see SegmentView::scanRg)
        std::vector<uint32_t> scanRg(n_samples);
        c->maxLoad.resize(n_readgroups);
        for (uint32_t i = 0; i < n_readgroups; ++i)
            c->maxLoad[i] = rgs[i].max_load;
        for (uint32_t sIdx = 0; sIdx < n_samples; ++sIdx)
        {
            if (sample_rg_offsets[sIdx + 1] - sample_rg_offsets[sIdx] == 1u)
            {
                scanRg[sIdx] = sample_rg_offsets[sIdx];
                continue;
            }
            scanRg[sIdx] = n_readgroups + static_cast<uint32_t>(c->scanSample.size());
            c->scanSample.push_back(sIdx);
            pd::RgConst v = rc[sample_rg_offsets[sIdx]];
            v.maxLoad = 0xffffffffu;            // the per-read-group limit is enforced through scanGeneric
            rc.push_back(v);
        }
        c->dScanRg.upload(scanRg, c->stream);
        c->dScanSample.upload(c->scanSample, c->stream);
        {
            // multi tiles: 256-sample tiles with at least one multi-read-group sample
            std::vector<uint32_t> mtTile, mtRgLo, mtRgHi, mtWaveBase, mtWaveTile, rgSample(n_readgroups);
            for (uint32_t sIdx = 0; sIdx < n_samples; ++sIdx)
                for (uint32_t rg = sample_rg_offsets[sIdx]; rg < sample_rg_offsets[sIdx + 1]; ++rg)
                    rgSample[rg] = sIdx;
            const uint32_t tiles = (n_samples + pd::kSamplesPerBlock - 1) / pd::kSamplesPerBlock;
            uint32_t waves = 0;
            for (uint32_t t = 0; t < tiles; ++t)
            {
                const uint32_t s0 = t * pd::kSamplesPerBlock, s1 = std::min<uint32_t>(n_samples, s0 + pd::kSamplesPerBlock);
                const uint32_t lo = sample_rg_offsets[s0], hi = sample_rg_offsets[s1];
                if (hi - lo == s1 - s0)
                    continue;                                  // one read group per sample
                mtTile.push_back(t);
                mtRgLo.push_back(lo);
                mtRgHi.push_back(hi);
                mtWaveBase.push_back(waves);
                for (uint32_t k = 0; k < (hi - lo + 63) / 64; ++k)
                    mtWaveTile.push_back(static_cast<uint32_t>(mtTile.size() - 1));
                waves += (hi - lo + 63) / 64;
            }
            c->dRgSample.upload(rgSample, c->stream);
            std::memset(&c->multiTiles, 0, sizeof(c->multiTiles));
            c->multiTiles.n = static_cast<uint32_t>(mtTile.size());
            c->multiTiles.nWaves = waves;
            if (!mtTile.empty())
            {
                c->dMtTile.upload(mtTile, c->stream);
                c->dMtRgLo.upload(mtRgLo, c->stream);
                c->dMtRgHi.upload(mtRgHi, c->stream);
                c->dMtWaveBase.upload(mtWaveBase, c->stream);
                c->dMtWaveTile.upload(mtWaveTile, c->stream);
                c->multiTiles.tile = c->dMtTile.p;
                c->multiTiles.rgLo = c->dMtRgLo.p;
                c->multiTiles.rgHi = c->dMtRgHi.p;
                c->multiTiles.waveBase = c->dMtWaveBase.p;
                c->multiTiles.waveTile = c->dMtWaveTile.p;
            }
            c->multiTiles.rgSample = c->dRgSample.p;
        }
        c->dRg.upload(rc, c->stream);
        c->dHist.upload(pool, c->stream);
        c->dSampleRg.upload(c->sampleRg, c->stream);
        c->dTab.alloc(pool.size());
        c->dTab2.alloc(pool.size());
        c->dIterFlags.alloc(pd::kMaxIterationsCap + 2);          // the last word is the guard of an optimistic launch
        if (const char * e = getenv("POPDEL_HIP_EXACT_POOLS"))
            c->optimisticPools = atoi(e) == 0;
        if (const char * e = getenv("POPDEL_HIP_HOST_LISTS"))
            c->deviceLists = atoi(e) == 0;
        if (const char * e = getenv("POPDEL_HIP_SCAN"))
            c->screenScan = std::strcmp(e, "full") != 0;
        if (const char * e = getenv("POPDEL_HIP_PLAIN_UPLOAD"))
            c->compactUpload = atoi(e) == 0;
        if (const char * e = getenv("POPDEL_HIP_STAGE2_BYTES"))
            c->stage2Budget = static_cast<size_t>(std::max(1.0, atof(e)));
        pd::buildLogTables(c->dHist.p, c->dTab.p, c->dTab2.p, pool.size(), c->dRg.p, n_readgroups, c->stream);
        PD_HIP(hipGetLastError());
        PD_HIP(hipStreamSynchronize(c->stream));
        c->cohort.rg = c->dRg.p;
        c->cohort.hist = c->dHist.p;
        c->cohort.tab = c->dTab.p;
        c->cohort.tab2 = c->dTab2.p;
        c->cohort.sampleRg = c->dSampleRg.p;
        c->cohort.nSamples = n_samples;
        c->cohort.nRg = n_readgroups;
        c->cohort.minThreshold = minThreshold;
        c->uniformThreshold = (int)rgs[0].min_init_del_len;
        for (uint32_t i = 1; i < n_readgroups; ++i)
            if (rgs[i].min_init_del_len != rgs[0].min_init_del_len)
                c->uniformThreshold = -1;
        std::memset(&c->timing, 0, sizeof(c->timing));
        std::memset(&c->shard, 0, sizeof(c->shard));
        c->nSamplesGlobal = n_samples;
        c->samplesPerRank = pd_shard_samples(n_samples, 1);
        std::vector<unsigned> minInit(n_readgroups);
        for (uint32_t i = 0; i < n_readgroups; ++i)
            minInit[i] = rgs[i].min_init_del_len;
        c->dMinInitGlobal.upload(minInit, c->stream);
        PD_HIP(hipStreamSynchronize(c->stream));
    }
    catch (const HipError & e)
    {
        return fail(nullptr, "pd_ctx_create: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(nullptr, std::string("pd_ctx_create: ") + e.what());
    }
    catch (...)
    {
        return fail(nullptr, "pd_ctx_create: unknown exception");
    }
    *out = c.release();
    return 0;
}

uint32_t pd_shard_samples(uint32_t n_samples_global, uint32_t world)
{
    if (world == 0)
        return 0;
    const uint32_t tiles = (n_samples_global + pd::kSamplesPerBlock - 1) / pd::kSamplesPerBlock;
    return (tiles + world - 1) / world * pd::kSamplesPerBlock;
}

static void laneTeardown(pd_ctx * c);

static int applyShard(pd_ctx * c, const pd_shard * sh, const uint8_t * rcclId, pd_local_group * group, const char * who,
                      const pd_shard * laneShards = nullptr, uint32_t nLaneShards = 0)
{
    const std::string name = who;
    if (!c || !sh)
        return fail(c, name + ": invalid argument");
    if (!c->runs.empty() || !c->staged.empty())
        return fail(c, name + ": must be called before segments are staged");
    if (sh->world == 0 || sh->rank >= sh->world || !sh->min_init_del_len_global)
        return fail(c, name + ": invalid rank/world or missing threshold table");
    if (group && (group->world != sh->world || group->members.size() != sh->world))
        return fail(c, name + ": the group was created for another number of contexts");
    if (c->lanes)
    {
        // Lanes exchange on channels of their own (their exchanges interleave in an order no other rank could reproduce on
        // ONE channel). With one rank every lane makes itself an RCCL communicator; a caller that exchanges through host
        // callbacks may hand over one pair of callbacks per lane (pd_ctx_set_shard_lanes: the tickets go to as many lanes as
        // it gave channels for); any other cohort over several ranks -- or between the contexts of a process -- runs
        // without lanes.
        if (sh->world == 1 && rcclId && !group)
        {
            RcclApi & api = rccl();
            if (!api.error.empty())
                return fail(c, name + ": " + api.error);
            for (auto & w : c->lanes->workers)
            {
                ncclUniqueId own;
                if (api.getUniqueId(&own) != ncclSuccess)
                    return fail(c, name + ": ncclGetUniqueId failed");
                if (int rc = applyShard(w->ctx, sh, reinterpret_cast<const uint8_t *>(&own), nullptr, who))
                {
                    c->error = w->ctx->error;
                    return rc;
                }
            }
        }
        else if (laneShards && nLaneShards > 0 && !rcclId && !group)
        {
            LanesState & L = *c->lanes;
            const unsigned use = std::min<unsigned>(L.active, nLaneShards);
            for (unsigned k = 0; k < use; ++k)
                if (int rc = applyShard(L.workers[k]->ctx, &laneShards[k], nullptr, nullptr, who))
                {
                    c->error = L.workers[k]->ctx->error;
                    return rc;
                }
            L.active = use;
        }
        else
            laneTeardown(c);
    }
    if (!rcclId && !group && sh->world > 1 && (!sh->reserve || !sh->all_gather))
        return fail(c, name + ": a sharded cohort needs the reserve() and all_gather() callbacks");
    const uint32_t per = pd_shard_samples(sh->n_samples_global, sh->world);
    const uint64_t first = static_cast<uint64_t>(sh->rank) * per;
    if (first >= sh->n_samples_global)
        return fail(c, name + ": this rank owns no samples (cohort too small for that many contexts)");
    const uint32_t own = static_cast<uint32_t>(std::min<uint64_t>(per, sh->n_samples_global - first));
    if (sh->first_sample != first || c->nSamples != own)
        return fail(c, name + ": the context's samples are not this rank's part of the pd_shard_samples() partition");
    if (sh->n_readgroups_global < sh->n_samples_global)
        return fail(c, name + ": fewer read groups than samples");
    try
    {
        PD_HIP(hipSetDevice(c->device));
        if (rcclId)
        {
            RcclApi & api = rccl();
            if (!api.error.empty())
                return fail(c, name + ": " + api.error);
            ncclUniqueId id;
            static_assert(sizeof(id) == PD_RCCL_ID_BYTES, "ncclUniqueId size");
            std::memcpy(&id, rcclId, sizeof(id));
            if (c->comm)
            {
                (void)api.commDestroy(c->comm);
                c->comm = nullptr;
            }
            const ncclResult_t r = api.commInitRank(&c->comm, static_cast<int>(sh->world), id, static_cast<int>(sh->rank));
            if (r != ncclSuccess)
            {
                c->comm = nullptr;
                return fail(c, name + ": ncclCommInitRank: " + api.getErrorString(r));
            }
        }
        if (group)
        {
            // The device-side carrier (exchange.hip) has only ever run with its contexts on ONE GPU -- no box with two was
            // available to this project -- so the host-driven carrier is what a local group uses unless the device-side one is
            // asked for by name (POPDEL_HIP_GROUP_CARRIER=flags); tests/test_gpu_local_group.py holds the two against each
            // other, on distinct GPUs where the box has them.
            const char * how = getenv("POPDEL_HIP_GROUP_CARRIER");
            c->flagsWanted = how && std::strcmp(how, "flags") == 0 && sh->world <= pd::kXchMaxWorld;
            c->groupDecided = false;
            if (c->flagsWanted)
            {
                void * f = nullptr;
                PD_HIP(hipExtMallocWithFlags(&f, 2 * pd::kXchMaxWorld * sizeof(unsigned), hipDeviceMallocFinegrained));
                PD_HIP(hipMemset(f, 0, 2 * pd::kXchMaxWorld * sizeof(unsigned)));
                c->xchFlags = static_cast<unsigned *>(f);
                void * e = nullptr;
                PD_HIP(hipHostMalloc(&e, 64, hipHostMallocMapped));
                c->xchErr = static_cast<unsigned *>(e);
                *c->xchErr = 0;
                c->xchSeq = 0;
                PD_HIP(hipDeviceSynchronize());
            }
            std::lock_guard<std::mutex> lock(group->m);
            group->members[sh->rank] = c;
            if (std::find(group->devices.begin(), group->devices.end(), c->device) == group->devices.end())
                group->devices.push_back(c->device);
            c->group = group;
        }
        c->shard = *sh;
        c->sharded = true;
        c->rank = sh->rank;
        c->world = sh->world;
        c->nSamplesGlobal = sh->n_samples_global;
        c->samplesPerRank = per;
        std::vector<unsigned> minInit(sh->min_init_del_len_global, sh->min_init_del_len_global + sh->n_readgroups_global);
        c->dMinInitGlobal.upload(minInit, c->stream);
        PD_HIP(hipStreamSynchronize(c->stream));
        c->shard.min_init_del_len_global = nullptr;        // copied; the caller's table need not outlive this call
        // thresholds are looked up across the whole cohort
        unsigned mn = minInit[0];
        c->uniformThreshold = static_cast<int>(minInit[0]);
        for (unsigned v : minInit)
        {
            mn = std::min(mn, v);
            if (v != minInit[0])
                c->uniformThreshold = -1;
        }
        c->cohort.minThreshold = static_cast<int>(mn);
    }
    catch (const HipError & e)
    {
        return fail(c, name + ": " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, name + ": " + e.what());
    }
    catch (...)
    {
        return fail(c, name + ": unknown exception");
    }
    return 0;
}

int pd_ctx_set_shard(pd_ctx * c, const pd_shard * sh) { return applyShard(c, sh, nullptr, nullptr, "pd_ctx_set_shard"); }

int pd_ctx_set_shard_lanes(pd_ctx * c, const pd_shard * sh, const pd_shard * lane_shards, uint32_t n_lane_shards)
{
    if (!lane_shards || n_lane_shards == 0)
        return fail(c, "pd_ctx_set_shard_lanes: no lane channels given");
    for (uint32_t k = 0; k < n_lane_shards; ++k)
        if (!sh || lane_shards[k].rank != sh->rank || lane_shards[k].world != sh->world || lane_shards[k].first_sample != sh->first_sample ||
            lane_shards[k].n_samples_global != sh->n_samples_global || !lane_shards[k].reserve || !lane_shards[k].all_gather)
            return fail(c, "pd_ctx_set_shard_lanes: a lane's pd_shard must describe the same rank and cohort and carry its own callbacks");
    return applyShard(c, sh, nullptr, nullptr, "pd_ctx_set_shard_lanes", lane_shards, n_lane_shards);
}

int pd_local_group_create(uint32_t world, pd_local_group ** out)
{
    if (!out || world == 0)
        return fail(nullptr, "pd_local_group_create: invalid argument");
    pd_local_group * g = new (std::nothrow) pd_local_group();
    if (!g)
        return fail(nullptr, "pd_local_group_create: out of memory");
    g->world = world;
    g->members.assign(world, nullptr);
    *out = g;
    return 0;
}

void pd_local_group_destroy(pd_local_group * group)
{
    if (!group)
        return;
    if (group->parksFrees)
    {
        // the last group that parked replaced device buffers (contexts sharing a GPU under the device-side carrier) gives them
        // back: nothing of the group is in flight any more once its devices are idle
        std::vector<void *> parked;
        {
            std::lock_guard<std::mutex> lock(g_parkLock);
            if (--g_parkingGroups == 0)
            {
                g_parkFrees.store(false);
                parked.swap(g_parked);
            }
        }
        if (!parked.empty())
        {
            int before = 0;
            (void)hipGetDevice(&before);
            for (int d : group->devices)
            {
                (void)hipSetDevice(d);
                (void)hipDeviceSynchronize();
            }
            for (void * p : parked)
                (void)hipFree(p);
            (void)hipSetDevice(before);
        }
    }
    delete group;
}

int pd_ctx_set_shard_local(pd_ctx * c, const pd_shard * sh, pd_local_group * group)
{
    if (!group)
        return fail(c, "pd_ctx_set_shard_local: invalid argument");
    return applyShard(c, sh, nullptr, group, "pd_ctx_set_shard_local");
}

int pd_ctx_set_shard_rccl(pd_ctx * c, const pd_shard * sh, const uint8_t * id)
{
    if (!id)
        return fail(c, "pd_ctx_set_shard_rccl: invalid argument");
    return applyShard(c, sh, id, nullptr, "pd_ctx_set_shard_rccl");
}

int pd_rccl_unique_id(uint8_t * id)
{
    if (!id)
        return fail(nullptr, "pd_rccl_unique_id: invalid argument");
    RcclApi & api = rccl();
    if (!api.error.empty())
        return fail(nullptr, "pd_rccl_unique_id: " + api.error);
    ncclUniqueId u;
    const ncclResult_t r = api.getUniqueId(&u);
    if (r != ncclSuccess)
        return fail(nullptr, std::string("pd_rccl_unique_id: ") + api.getErrorString(r));
    std::memcpy(id, &u, sizeof(u));
    return 0;
}

static void laneTeardown(pd_ctx * c);

void pd_ctx_destroy(pd_ctx * ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    laneTeardown(ctx);
    if (ctx->xchFlags)
        (void)hipFree(ctx->xchFlags);
    if (ctx->xchErr)
        (void)hipHostFree(ctx->xchErr);
    for (auto & r : ctx->runs)
    {
        if (r.second.s1Begin)
            (void)hipEventDestroy(r.second.s1Begin);
        if (r.second.s1End)
            (void)hipEventDestroy(r.second.s1End);
        if (r.second.scanDone)
            (void)hipEventDestroy(r.second.scanDone);
    }
    delete ctx;
}

static std::unique_ptr<StagedSegment> takeSpare(pd_ctx * c)
{
    std::lock_guard<std::mutex> guard(c->stagedLock);
    if (c->spare.empty())
        return std::unique_ptr<StagedSegment>(new StagedSegment());
    std::unique_ptr<StagedSegment> s = std::move(c->spare.back());
    c->spare.pop_back();
    return s;
}

static void putSpare(pd_ctx * c, std::unique_ptr<StagedSegment> s)
{
    std::lock_guard<std::mutex> guard(c->stagedLock);
    c->spare.push_back(std::move(s));
}

static void putStaged(pd_ctx * c, uint64_t handle, std::unique_ptr<StagedSegment> s)
{
    std::lock_guard<std::mutex> guard(c->stagedLock);
    c->staged[handle] = std::move(s);
}

static StagedSegment * findStaged(pd_ctx * c, uint64_t handle);

// ---- lanes ---------------------------------------------------------------------------------------------------------------
// A lane's thread: the ordinary sequence on the lane's own context, pipelined the way a streaming host drives a plain
// context -- with work queued, the next job's ticket is issued before the open one is collected, so that its scan goes onto
// the lane's stream right behind the open segment's last kernel.
static void laneLoop(pd_ctx * parent, unsigned k)
{
    LanesState & L = *parent->lanes;
    LaneWorker & w = *L.workers[k];
    (void)hipSetDevice(w.ctx->device);
    struct Open
    {
        uint64_t ticket, child;
        int rc;
    };
    std::deque<Open> open;
    for (;;)
    {
        // top up to two open tickets; wait only if there is nothing to collect
        for (;;)
        {
            LaneJob job;
            {
                std::unique_lock<std::mutex> lock(L.m);
                if (open.empty())
                    L.wake.wait(lock, [&] { return L.stop || !w.jobs.empty(); });
                if (w.jobs.empty() || open.size() >= 2)
                {
                    if (open.empty())
                        return;                 // (stop)
                    break;
                }
                job = w.jobs.front();
                w.jobs.pop_front();
            }
            Open o{job.ticket, 0, 0};
            o.rc = pd_segment_run(w.ctx, job.handle, &o.child);
            open.push_back(o);
        }
        const Open o = open.front();
        open.pop_front();
        LaneResult r;
        r.lane = k;
        r.rc = o.rc;
        if (r.rc == 0)
            r.rc = pd_segment_collect(w.ctx, o.child, &r.calls);
        if (r.rc != 0)
            r.error = w.ctx->error;
        r.timing = w.ctx->timing;
        r.done = true;
        {
            std::lock_guard<std::mutex> lock(L.m);
            r.handle = L.results[o.ticket].handle;
            L.results[o.ticket] = std::move(r);
        }
        L.done.notify_all();
    }
}

static int laneRun(pd_ctx * c, uint64_t handle, uint64_t * ticket)
{
    LanesState & L = *c->lanes;
    if (!findStaged(c, handle))
        return fail(c, "pd_segment_run: unknown segment handle");
    std::unique_lock<std::mutex> lock(L.m);
    // (columns kept on the device live until the lane's next collect: then no lane gets a second ticket before its first is out)
    if (L.results.size() >= (c->keepOnDevice ? 1u : 2u) * L.active)
        return fail(c, "pd_segment_run: too many tickets outstanding (collect the oldest one first)");
    *ticket = c->nextTicket++;
    L.results[*ticket] = LaneResult();
    L.results[*ticket].handle = handle;
    LaneJob job;
    job.ticket = *ticket;
    job.handle = handle;
    L.workers[*ticket % L.active]->jobs.push_back(job);
    lock.unlock();
    L.wake.notify_all();
    return 0;
}

static int laneCollect(pd_ctx * c, uint64_t ticket, pd_calls ** out)
{
    LanesState & L = *c->lanes;
    std::unique_lock<std::mutex> lock(L.m);
    auto it = L.results.find(ticket);
    if (it == L.results.end())
        return fail(c, "pd_segment_collect: unknown ticket");
    if (it != L.results.begin())
        return fail(c, "pd_segment_collect: tickets are collected in the order they were issued");
    L.done.wait(lock, [&] { return it->second.done; });
    LaneResult r = std::move(it->second);
    L.results.erase(it);
    L.lastLane = r.lane;
    L.repeatedSeen.resize(L.workers.size(), 0);
    L.repeatedSeen[r.lane] = r.timing.n_repeated_batches;
    uint64_t repeated = 0;
    for (uint64_t v : L.repeatedSeen)
        repeated += v;
    lock.unlock();
    c->timing = r.timing;
    c->timing.n_repeated_batches = repeated;            // (cumulative over the context = over its lanes)
    if (r.rc != 0)
    {
        c->error = r.error;
        return r.rc;
    }
    *out = r.calls;
    return 0;
}

// until no lane has a ticket for segment `handle` left to run (results stay where they are: collect takes them)
static void laneDrain(pd_ctx * c, uint64_t handle)
{
    LanesState & L = *c->lanes;
    std::unique_lock<std::mutex> lock(L.m);
    L.done.wait(lock, [&] {
        for (const auto & r : L.results)
            if (!r.second.done && r.second.handle == handle)
                return false;
        return true;
    });
}

static void laneTeardown(pd_ctx * c)
{
    if (!c->lanes)
        return;
    LanesState & L = *c->lanes;
    {
        std::lock_guard<std::mutex> lock(L.m);
        L.stop = true;
        for (auto & w : L.workers)
            w->jobs.clear();
    }
    L.wake.notify_all();
    for (auto & w : L.workers)
        if (w->thread.joinable())
            w->thread.join();
    for (auto & r : L.results)
        if (r.second.calls)
            pd_calls_free(r.second.calls);
    for (auto & w : L.workers)
        pd_ctx_destroy(w->ctx);
    c->lanes.reset();
}

int pd_segment_stage(pd_ctx * c, const pd_segment * seg, uint64_t * handle)
{
    if (!c || !seg || !handle)
        return fail(c, "pd_segment_stage: invalid argument");
    if (seg->n_readgroups != c->nRg)
        return fail(c, "pd_segment_stage: read-group count mismatch");
    if (seg->n_windows == 0)
        return fail(c, "pd_segment_stage: empty segment");
    try
    {
        PD_HIP(hipSetDevice(c->device));
        c->stageBegan = std::chrono::steady_clock::now();
        std::memset(&c->stageTiming, 0, sizeof(c->stageTiming));
        std::unique_ptr<StagedSegment> s;
        s = takeSpare(c);
        s->ownedBySubmit = false;
        s->waitBuilt = false;
        s->firstPos = seg->first_window_pos;
        s->nWindows = seg->n_windows;
        s->shift = seg->window_shift;
        // the layout is built on the device from instances that cross PCIe once; POPDEL_HIP_HOST_BUILD=1 (A/B runs; the
        // results are identical) or a window count beyond the build kernels' LDS selects the host builder below
        const bool onDevice = !getenv("POPDEL_HIP_HOST_BUILD") && buildChunksOnDevice(*c, *seg, *s);
        if (onDevice)
        {
            s->scanRg = c->dScanRg.p;
            s->nRealRg = c->nRg;
            // No wait for the last build kernels: the uploads have completed (the build's own syncs come after them), so
            // the host scratch is free; the kernels keep running on the copy stream while the caller prepares the next
            // segment, and the first run of this segment waits for the event on the device.
            if (!s->built)
                PD_HIP(hipEventCreateWithFlags(&s->built, hipEventDisableTiming));
            PD_HIP(hipEventRecord(s->built, c->copyStream));
            s->waitBuilt = true;
            *handle = c->nextHandle++;
            putStaged(c, *handle, std::move(s));
            c->stageTiming.total_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - c->stageBegan).count();
            return 0;
        }
        PinnedBuf<pd::PairDO> & pairs = c->hPairs;
        PinnedBuf<uint32_t> &start = c->hStart, &winStart = c->hWinStart, &endCount = c->hEndCount;
        std::vector<uint64_t> & rgBase = c->hRgBase;
        PinnedBuf<uint32_t> & pairsPk = c->hPairsPk;
        std::vector<uint8_t> & rgPacked = c->hRgPacked;
        // Uploads go over the copy stream: a ticket that is running on the compute stream is not held up by the staging
        // of the next segment (double-buffered streaming: run(N), stage(N+1), collect(N), run(N+1), ...), and they go
        // slab by slab, so that PCIe works while the host threads still build the next slab.
        const uint32_t W = seg->n_windows;
        hipStream_t cs = c->copyStream;
        auto sized = [&]() {
            s->pairs.ensure(pairs.size());
            s->start.ensure(start.size());
            s->pairsPk.ensure(pairsPk.size() + 256);     // the scan's LDS-DMA copies whole 256-chunk pieces (stage1.hip)
            s->endCount.ensure(endCount.size());
            s->rgBase.upload(rgBase, cs);
            s->winStart.uploadPinned(winStart, cs);
        };
        auto slabDone = [&](size_t lo, size_t hi) {
            const uint64_t a = rgBase[lo], b = hi < rgBase.size() ? rgBase[hi] : pairsPk.size();
            if (b > a)
            {
                if (lo < c->nRg)            // real read groups: all three chunk arrays; scan read groups: the stage-1 image only
                {
                    PD_HIP(hipMemcpyAsync(s->pairs.p + a, pairs.data() + a, (b - a) * sizeof(pd::PairDO), hipMemcpyHostToDevice, cs));
                    PD_HIP(hipMemcpyAsync(s->start.p + a, start.data() + a, (b - a) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
                }
                PD_HIP(hipMemcpyAsync(s->pairsPk.p + a, pairsPk.data() + a, (b - a) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
            }
            const size_t e0 = lo * (static_cast<size_t>(W) + 1), e1 = hi * (static_cast<size_t>(W) + 1);
            PD_HIP(hipMemcpyAsync(s->endCount.p + e0, endCount.data() + e0, (e1 - e0) * sizeof(uint32_t), hipMemcpyHostToDevice, cs));
        };
        s->ownPkIndex = false;                      // the host builder cuts both images with one chunk life
        buildChunks(*c, *seg, *s, pairs, start, rgBase, winStart, pairsPk, endCount, rgPacked, c->hScanGeneric, sized, slabDone);
        s->rgPacked.upload(rgPacked, cs);
        s->hasScanGeneric = !c->hScanGeneric.empty();
        if (s->hasScanGeneric)
            s->scanGeneric.upload(c->hScanGeneric, cs);
        s->scanRg = c->dScanRg.p;
        s->nRealRg = c->nRg;
        {
            const uint32_t nEnt = static_cast<uint32_t>(winStart.size() / (static_cast<size_t>(W) + s->maxLife));
            s->active8.ensure(static_cast<size_t>(nEnt) * W);
            pd::launchActive8(s->winStart.p, s->endCount.p, nEnt, W, s->maxLife, s->active8.p, cs);
            PD_HIP(hipGetLastError());
        }
        PD_HIP(hipStreamSynchronize(cs));              // the host scratch is reused by the next call
        *handle = c->nextHandle++;
        putStaged(c, *handle, std::move(s));
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_segment_stage: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, std::string("pd_segment_stage: ") + e.what());
    }
    catch (...)
    {
        return fail(c, "pd_segment_stage: unknown exception");
    }
    return 0;
}

// Enqueues the stage-1 scan of a run on the context's stream.
// The staged segment behind a handle: a lane looks in its owner's table (the owner does not unstage a segment while one of its
// lanes has a ticket for it: pd_segment_unstage drains them first).
static StagedSegment * findStaged(pd_ctx * c, uint64_t handle)
{
    pd_ctx * o = c->owner ? c->owner : c;
    std::lock_guard<std::mutex> guard(o->stagedLock);
    auto it = o->staged.find(handle);
    return it == o->staged.end() ? nullptr : it->second.get();
}

static void launchScan(pd_ctx * c, Run & run)
{
    StagedSegment * found = findStaged(c, run.handle);
    if (!found)
        throw HipError{"the segment of a ticket was unstaged"};
    StagedSegment & s = *found;
    Stage1Out & o = c->s1[run.set];
    hipStream_t st = c->scanStream ? c->scanStream : c->stream;
    if (s.waitBuilt)
    {
        PD_HIP(hipStreamWaitEvent(st, s.built, 0));
        if (!c->owner)              // (lanes: every lane's stream waits, whichever runs the segment first)
            s.waitBuilt = false;
    }
    const size_t cells = static_cast<size_t>(c->nSamples) * s.nWindows;
    o.q3.ensure(cells);
    o.flags.ensure(cells);
    o.winFlag.ensure(s.nWindows);
    const size_t tiles = static_cast<size_t>((s.nWindows + pd::kStage1Block - 1) / pd::kStage1Block) * c->nSamples;
    o.tileTodo.ensure(tiles);
    o.todoList.ensure(tiles + 2);
    o.bigList.ensure(tiles / 2 + c->nSamples + 2);
    pd::ZeroRanges zero{};
    zero.add(o.winFlag.p, s.nWindows * sizeof(int));
    zero.add(o.todoList.p, sizeof(unsigned));
    zero.add(o.bigList.p, sizeof(unsigned));
    pd::ScanRoom room = c->scanRoom;
    if (room.mode == pd::ScanRoom::kPersistent)
    {
        room.counter = c->dScanCounter.p + run.set;
        zero.add(room.counter, sizeof(unsigned));
    }
    pd::launchZero(zero, st);
    o.screened = c->screenScan;
    o.fullValid = false;
    if (c->screenScan)
    {
        if (c->stage1View)
        {
            // the diagnostic view: the full scan's Q3 / flags / window flags of every cell beside the screen's outputs (the todo and
            // big lists are shared: the full scan runs first and leaves them as it found them for the screen -- re-zeroed below)
            o.fullQ3.ensure(cells);
            o.fullFlags.ensure(cells);
            o.fullWinFlag.ensure(s.nWindows);
            PD_HIP(hipMemsetAsync(o.fullWinFlag.p, 0, s.nWindows * sizeof(int), st));
            pd::launchStage1(viewOf(s), c->cohort, o.fullQ3.p, o.fullFlags.p, o.fullWinFlag.p, o.tileTodo.p, o.todoList.p, o.bigList.p, s.maxBlockChunks,
                             room, st);
            pd::launchStage1Fallback(viewOf(s), c->cohort, o.fullQ3.p, o.fullFlags.p, o.fullWinFlag.p, o.tileTodo.p, o.todoList.p, o.bigList.p, st);
            PD_HIP(hipGetLastError());
            pd::ZeroRanges again{};
            again.add(o.todoList.p, sizeof(unsigned));
            again.add(o.bigList.p, sizeof(unsigned));
            pd::launchZero(again, st);
            o.fullValid = true;
        }
        PD_HIP(hipEventRecord(run.s1Begin, st));
        pd::launchScreen(viewOf(s), c->cohort, o.q3.p, o.flags.p, o.winFlag.p, o.tileTodo.p, o.todoList.p, st);
        PD_HIP(hipGetLastError());
        PD_HIP(hipEventRecord(run.s1End, st));
        pd::launchScreenFallback(viewOf(s), c->cohort, o.q3.p, o.flags.p, o.winFlag.p, o.todoList.p, st);
        PD_HIP(hipGetLastError());
    }
    else
    {
        PD_HIP(hipEventRecord(run.s1Begin, st));
        pd::launchStage1(viewOf(s), c->cohort, o.q3.p, o.flags.p, o.winFlag.p, o.tileTodo.p, o.todoList.p, o.bigList.p, s.maxBlockChunks, room, st);
        PD_HIP(hipGetLastError());
        PD_HIP(hipEventRecord(run.s1End, st));
        pd::launchStage1Fallback(viewOf(s), c->cohort, o.q3.p, o.flags.p, o.winFlag.p, o.tileTodo.p, o.todoList.p, o.bigList.p, st);
        PD_HIP(hipGetLastError());
    }
    if (getenv("POPDEL_HIP_DEBUG"))
    {
        std::vector<uint8_t> todo(tiles);
        PD_HIP(hipMemcpyAsync(todo.data(), o.tileTodo.p, tiles, hipMemcpyDeviceToHost, st));
        PD_HIP(hipStreamSynchronize(st));
        size_t nTodo = 0;
        for (uint8_t t : todo)
            nTodo += t != 0;
        unsigned nBig = 0;
        PD_HIP(hipMemcpy(&nBig, o.bigList.p, sizeof(unsigned), hipMemcpyDeviceToHost));
        fprintf(stderr, "[popdel_hip] stage 1: %u 512-window blocks through the 128-candidate kernel, %zu of %zu 256-window tiles through the "
                        "32-bit fallback kernel (K=%u, densest block %u chunks)\n", nBig, nTodo, tiles, s.maxLife, s.maxBlockChunks);
    }
    if (st != c->stream)
    {
        if (!run.scanDone)
            PD_HIP(hipEventCreateWithFlags(&run.scanDone, hipEventDisableTiming));
        PD_HIP(hipEventRecord(run.scanDone, st));
    }
    o.windows = s.nWindows;
    run.launched = true;
}

int pd_segment_run(pd_ctx * c, uint64_t handle, uint64_t * ticket)
{
    if (!c || !ticket)
        return fail(c, "pd_segment_run: invalid argument");
    if (c->lanes)
        return laneRun(c, handle, ticket);
    if (!findStaged(c, handle))
        return fail(c, "pd_segment_run: unknown segment handle");
    // One ticket may be open plus one waiting: the waiting one's scan is enqueued by pd_segment_collect() of the open one,
    // right behind its last kernel, so that it runs while the open ticket's results travel to the host.
    if (c->runs.size() >= 2)
        return fail(c, "pd_segment_run: two tickets are already outstanding (collect the older one first)");
    try
    {
        PD_HIP(hipSetDevice(c->device));
        Run run;
        run.handle = handle;
        PD_HIP(hipEventCreate(&run.s1Begin));
        PD_HIP(hipEventCreate(&run.s1End));
        *ticket = c->nextTicket++;
        run.set = static_cast<int>(*ticket & 1u);          // tickets alternate between the two output sets
        auto & slot = c->runs[*ticket];
        slot = run;
        // (with a scan stream the scan starts at once, beside the open ticket's stage 2; its output set was last read by
        // the ticket before the open one, which has been collected)
        if (c->runs.size() == 1 || c->scanStream)
            launchScan(c, slot);
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_segment_run: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, std::string("pd_segment_run: ") + e.what());
    }
    catch (...)
    {
        return fail(c, "pd_segment_run: unknown exception");
    }
    return 0;
}

int pd_segment_unstage(pd_ctx * c, uint64_t handle)
{
    if (!c)
        return 1;
    (void)hipSetDevice(c->device);
    if (c->lanes)
        laneDrain(c, handle);       // no lane may still be running the segment
    (void)hipStreamSynchronize(c->stream);
    if (c->scanStream)
        (void)hipStreamSynchronize(c->scanStream);
    std::lock_guard<std::mutex> guard(c->stagedLock);
    auto it = c->staged.find(handle);
    if (it == c->staged.end())
        return fail(c, "pd_segment_unstage: unknown segment handle");
    if (c->spare.size() < 2)
        c->spare.push_back(std::move(it->second));
    c->staged.erase(it);
    return 0;
}

int pd_segment_submit(pd_ctx * c, const pd_segment * seg, uint64_t * ticket)
{
    uint64_t h = 0;
    if (int rc = pd_segment_stage(c, seg, &h))
        return rc;
    {
        std::lock_guard<std::mutex> guard(c->stagedLock);
        c->staged[h]->ownedBySubmit = true;
    }
    if (int rc = pd_segment_run(c, h, ticket))
    {
        std::lock_guard<std::mutex> guard(c->stagedLock);
        c->staged.erase(h);
        return rc;
    }
    return 0;
}

namespace {

pd::WalkState walkStateOf(pd_ctx & c, int set)
{
    pd_ctx::WalkBuffers & b = c.walkState[set];
    pd::WalkState st;
    st.aStart = b.aStart.p;
    st.aDev = b.aDev.p;
    st.aId = b.aId.p;
    st.aLast = b.aLast.p;
    st.nActive = b.nActive.p;
    st.pStart = b.pStart.p;
    st.pDev = b.pDev.p;
    st.pId = b.pId.p;
    st.pLast = b.pLast.p;
    st.nPending = b.nPending.p;
    st.capActive = c.walkCapActive;
    st.capPending = c.walkCapPending;
    return st;
}

// First use: room for the lists of every read group (active pairs are bounded by the read groups' maxLoad -- the loader
// stops filing pairs beyond it -- plus what is carried; the pending list holds the sliver of a slot no window reached).
void prepareWalkState(pd_ctx & c)
{
    if (c.walkReady)
        return;
    uint32_t most = 0;
    for (uint32_t m : c.maxLoad)
        most = std::max(most, std::min<uint32_t>(m, 4096u));
    c.walkCapActive = std::max<uint32_t>(256u, 2u * most + 64u);
    if (const char * e = getenv("POPDEL_HIP_WALK_CAP"))         // (tests: start small so that the lists have to grow)
        c.walkCapActive = c.walkCapPending = std::max(1, atoi(e));
    for (int k = 0; k < 2; ++k)
    {
        pd_ctx::WalkBuffers & b = c.walkState[k];
        const size_t na = static_cast<size_t>(c.nRg) * c.walkCapActive, np = static_cast<size_t>(c.nRg) * c.walkCapPending;
        b.aStart.alloc(na);
        b.aDev.alloc(na);
        b.aId.alloc(na);
        b.aLast.alloc(na);
        b.pStart.alloc(np);
        b.pDev.alloc(np);
        b.pId.alloc(np);
        b.pLast.alloc(np);
        b.nActive.alloc(c.nRg);
        b.nPending.alloc(c.nRg);
        PD_HIP(hipMemsetAsync(b.nActive.p, 0, c.nRg * sizeof(uint32_t), c.copyStream));
        PD_HIP(hipMemsetAsync(b.nPending.p, 0, c.nRg * sizeof(uint32_t), c.copyStream));
    }
    c.dWalkTotals.alloc(5);             // instances, pairs left active, problems, longest active / pending list needed
    c.hWalkTotals.resize(5);
    c.walkCur = 0;
    c.walkReady = true;
}

// A read group needs longer lists than the state holds (a pile-up under an unlimited maxLoad, -A; the sliver of a slot no
// window reached): both sets get the new row length, the current set's rows move over (the other set is written afresh by
// the emit pass). Runs on the copy stream behind the count pass.
void growWalkState(pd_ctx & c, uint32_t needActive, uint32_t needPending)
{
    const uint32_t capA = needActive > c.walkCapActive ? needActive + needActive / 2 + 64u : c.walkCapActive;
    const uint32_t capP = needPending > c.walkCapPending ? needPending + needPending / 2 + 64u : c.walkCapPending;
    hipStream_t cs = c.copyStream;
    auto regrow = [&](auto & buf, uint32_t oldCap, uint32_t newCap, bool keep) {
        if (newCap == oldCap)
            return;
        std::remove_reference_t<decltype(buf)> bigger;
        bigger.alloc(static_cast<size_t>(c.nRg) * newCap);
        if (keep)
            PD_HIP(hipMemcpy2DAsync(bigger.p, static_cast<size_t>(newCap) * 4, buf.p, static_cast<size_t>(oldCap) * 4,
                                    static_cast<size_t>(oldCap) * 4, c.nRg, hipMemcpyDeviceToDevice, cs));
        PD_HIP(hipStreamSynchronize(cs));          // (the old rows are freed below)
        std::swap(buf.p, bigger.p);
        std::swap(buf.n, bigger.n);
    };
    for (int k = 0; k < 2; ++k)
    {
        pd_ctx::WalkBuffers & b = c.walkState[k];
        const bool keep = k == c.walkCur;
        regrow(b.aStart, c.walkCapActive, capA, keep);
        regrow(b.aDev, c.walkCapActive, capA, keep);
        regrow(b.aId, c.walkCapActive, capA, keep);
        regrow(b.aLast, c.walkCapActive, capA, keep);
        regrow(b.pStart, c.walkCapPending, capP, keep);
        regrow(b.pDev, c.walkCapPending, capP, keep);
        regrow(b.pId, c.walkCapPending, capP, keep);
        regrow(b.pLast, c.walkCapPending, capP, keep);
    }
    c.walkCapActive = capA;
    c.walkCapPending = capP;
}

}  // namespace

namespace {
// Host ranges offered by pd_host_regions_offer(): page-locked (and mapped for the device) the first time a walk's entries lie
// inside one. POPDEL_HIP_WALK_GATHER=0 keeps the staging copy for everything.
struct HostRange
{
    uintptr_t begin = 0, end = 0;
    uintptr_t deviceBase = 0;       // device-visible address of `begin` once registered
    int state = 0;                  // 0 offered, 1 registered, 2 the runtime refused (left to the staging copy)
};
std::mutex g_hostRangeLock;
std::vector<HostRange> g_hostRanges;        // sorted by begin, disjoint
std::atomic<uint64_t> g_hostBytesRegistered{0};

bool walkGatherWanted()
{
    static const bool on = [] {
        const char * e = std::getenv("POPDEL_HIP_WALK_GATHER");
        return !(e && e[0] == '0');
    }();
    return on;
}

// device-visible address of [p, p + bytes) if it lies inside an offered range (registering that range on first use), else 0
uintptr_t deviceViewOfHost(const void * p, size_t bytes)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> guard(g_hostRangeLock);
    auto it = std::upper_bound(g_hostRanges.begin(), g_hostRanges.end(), a, [](uintptr_t v, const HostRange & r) { return v < r.begin; });
    if (it == g_hostRanges.begin())
        return 0;
    HostRange & r = *(it - 1);
    if (a < r.begin || a + bytes > r.end)
        return 0;
    if (r.state == 0)
    {
        void * dev = nullptr;
        if (hipHostRegister(reinterpret_cast<void *>(r.begin), r.end - r.begin, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess &&
            hipHostGetDevicePointer(&dev, reinterpret_cast<void *>(r.begin), 0) == hipSuccess && dev)
        {
            r.deviceBase = reinterpret_cast<uintptr_t>(dev);
            r.state = 1;
            g_hostBytesRegistered += r.end - r.begin;
        }
        else
        {
            (void)hipGetLastError();
            r.state = 2;
        }
    }
    return r.state == 1 ? r.deviceBase + (a - r.begin) : 0;
}
}  // namespace

int pd_host_regions_offer(const void * const * begin, const uint64_t * bytes, uint32_t n)
{
    if (n != 0 && (!begin || !bytes))
        return 1;
    std::lock_guard<std::mutex> guard(g_hostRangeLock);
    for (uint32_t k = 0; k < n; ++k)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(begin[k]), b = a + bytes[k];
        if (a == 0 || bytes[k] == 0)
            continue;
        auto it = std::lower_bound(g_hostRanges.begin(), g_hostRanges.end(), a, [](const HostRange & r, uintptr_t v) { return r.begin < v; });
        // (offered before, or overlapping something offered before: ignored)
        if ((it != g_hostRanges.end() && it->begin < b) || (it != g_hostRanges.begin() && (it - 1)->end > a))
            continue;
        HostRange r;
        r.begin = a;
        r.end = b;
        g_hostRanges.insert(it, r);
    }
    return 0;
}

int pd_walk_entry_counts(pd_ctx * c, uint64_t * gathered, uint64_t * staged, uint64_t * locked_bytes)
{
    if (!c)
        return 1;
    if (gathered)
        *gathered = c->walkGathered;
    if (staged)
        *staged = c->walkStaged;
    if (locked_bytes)
        *locked_bytes = g_hostBytesRegistered.load();
    return 0;
}

int pd_walk_source_counts(pd_ctx * c, uint64_t * from_device)
{
    if (!c)
        return 1;
    if (from_device)
        *from_device = c->walkFromDevice;
    return 0;
}

int pd_walk_supported(const pd_ctx *) { return 1; }

int pd_walk_submit(pd_ctx * c, const pd_walk * wk, uint64_t * ticket)
{
    if (!c || !wk || !ticket || !wk->rg_counts || !wk->rg_entries || !wk->end_border || !wk->id_offset || !wk->switches || !wk->equal_run)
        return fail(c, "pd_walk_submit: invalid argument");
    if (wk->n_readgroups != c->nRg)
        return fail(c, "pd_walk_submit: read-group count mismatch");
    if (wk->n_windows == 0 || wk->window_shift == 0)
        return fail(c, "pd_walk_submit: a segment needs at least one window");
    const uint32_t nRg = c->nRg, W = wk->n_windows;
    uint64_t handle = 0;
    try
    {
        PD_HIP(hipSetDevice(c->device));
        hipStream_t cs = c->copyStream;
        prepareWalkState(*c);
        std::unique_ptr<StagedSegment> s;
        s = takeSpare(c);
        s->ownedBySubmit = true;
        s->waitBuilt = false;
        s->firstPos = wk->first_window_pos;
        s->nWindows = W;
        s->shift = wk->window_shift;
        // ---- the loader's tables -> HBM (page-locked staging copy by the host threads, slab by slab under the upload)
        c->hRgOffsets.resize(nRg + 1);
        c->hRgOffsets[0] = 0;
        for (uint32_t r = 0; r < nRg; ++r)
        {
            const uint32_t n = wk->switches[r] != 0 ? wk->rg_counts[r] : 0u;
            if (n != 0 && !wk->rg_entries[r] && !(wk->rg_sources && wk->rg_sources[r].inflater))
                throw HipError{"a read group has entries but no table"};
            c->hRgOffsets[r + 1] = c->hRgOffsets[r] + n;
        }
        const uint64_t nEntries = c->hRgOffsets[nRg];
        DevBuf<uint64_t> & dEntryOffsets = c->dWalkEntryOffsets;
        dEntryOffsets.uploadPinned(c->hRgOffsets, cs);
        c->hWalkBorder.resize(nRg);
        c->hWalkIdOffset.resize(nRg);
        c->hWalkFlags.resize(2 * static_cast<size_t>(nRg));
        std::memcpy(c->hWalkBorder.data(), wk->end_border, nRg * sizeof(uint32_t));
        std::memcpy(c->hWalkIdOffset.data(), wk->id_offset, nRg * sizeof(uint32_t));
        std::memcpy(c->hWalkFlags.data(), wk->switches, nRg);
        std::memcpy(c->hWalkFlags.data() + nRg, wk->equal_run, nRg);
        c->dWalkBorder.uploadPinned(c->hWalkBorder, cs);
        c->dWalkIdOffset.uploadPinned(c->hWalkIdOffset, cs);
        c->dWalkFlags.uploadPinned(c->hWalkFlags, cs);
        c->hWalkEntries.resize(nEntries);
        c->dWalkEntries.ensure(std::max<uint64_t>(1, nEntries));
        {
            // Where the entries lie in host memory the device can read (ranges the host offered, page-locked on first use) it
            // fetches them itself: one gather kernel over the link instead of 8 bytes per pair copied by the host threads.
            // Everything else: the read groups' tables -> one page-locked array (host threads), slab by slab under the upload.
            static_assert(sizeof(pd_walk_entry) == sizeof(uint2), "entry layout");
            c->hWalkSrc.resize(nRg);
            uint64_t gathered = 0;
            // (read groups whose entries lie on this GPU already -- the records the loader's inflater parsed there, pd_walk_source --
            // are copied from there: nothing crosses the link for them)
            uint64_t fromDevice = 0;
            static const bool deviceSources = [] {
                const char * e = std::getenv("POPDEL_HIP_WALK_SOURCES");
                return !(e && e[0] == '0');
            }();
            c->hWalkDevSrc.resize(nRg);
            std::fill(c->hWalkDevSrc.data(), c->hWalkDevSrc.data() + nRg, static_cast<uint64_t>(0));
            if (wk->rg_sources && deviceSources)
                for (uint32_t r = 0; r < nRg; ++r)
                {
                    const uint64_t n = c->hRgOffsets[r + 1] - c->hRgOffsets[r];
                    if (n == 0 || !wk->rg_sources[r].inflater)
                        continue;
                    uint64_t have = 0, runId = 0;
                    int dev = -1;
                    const uint2 * base = pd::inflaterDeviceRecords(wk->rg_sources[r].inflater, have, dev, runId);
                    if (!base || dev != c->device || runId != wk->rg_sources[r].run_id || wk->rg_sources[r].first_record > have ||
                        n > have - wk->rg_sources[r].first_record)
                        continue;
                    c->hWalkDevSrc[r] = reinterpret_cast<uintptr_t>(base + wk->rg_sources[r].first_record);
                    fromDevice += n;
                }
            for (uint32_t r = 0; r < nRg; ++r)
            {
                const uint64_t n = c->hRgOffsets[r + 1] - c->hRgOffsets[r];
                if (n != 0 && !c->hWalkDevSrc[r] && !wk->rg_entries[r])
                    throw HipError{"a read group's entries were to be taken from an inflater's records on the device, which are not there (another GPU, "
                                   "the inflater has run again, or POPDEL_HIP_WALK_SOURCES=0), and the host gave no table"};
                c->hWalkSrc[r] = n != 0 && !c->hWalkDevSrc[r] && walkGatherWanted() ? deviceViewOfHost(wk->rg_entries[r], n * sizeof(uint2)) : 0u;
                gathered += c->hWalkSrc[r] ? n : 0u;
            }
            if (fromDevice != 0)
            {
                c->dWalkDevSrc.uploadPinned(c->hWalkDevSrc, cs);
                pd::launchGatherDeviceEntries(c->dWalkDevSrc.p, dEntryOffsets.p, c->dWalkEntries.p, nRg, cs);
                PD_HIP(hipGetLastError());
            }
            c->walkFromDevice += fromDevice;
            if (gathered + fromDevice != nEntries)
            {
                const size_t nSlabs = std::min<size_t>(8, std::max<size_t>(1, nRg / 64));
                for (size_t slab = 0; slab < nSlabs; ++slab)
                {
                    const size_t rLo = nRg * slab / nSlabs, rHi = nRg * (slab + 1) / nSlabs;
                    const uint64_t a = c->hRgOffsets[rLo], b = c->hRgOffsets[rHi];
                    uint64_t staged = 0;
                    for (size_t r = rLo; r < rHi; ++r)
                        staged += c->hWalkSrc[r] || c->hWalkDevSrc[r] ? 0u : c->hRgOffsets[r + 1] - c->hRgOffsets[r];
                    if (staged == 0)
                        continue;
                    pdh::parallelFor(rHi - rLo, [&](size_t k) {
                        const size_t r = rLo + k;
                        const uint64_t n = c->hRgOffsets[r + 1] - c->hRgOffsets[r];
                        if (n && !c->hWalkSrc[r] && !c->hWalkDevSrc[r])
                            std::memcpy(c->hWalkEntries.data() + c->hRgOffsets[r], wk->rg_entries[r], n * sizeof(uint2));
                    }, 16);
                    // (the whole slab goes up; the stretches of gathered read groups are written by the kernel behind it)
                    PD_HIP(hipMemcpyAsync(c->dWalkEntries.p + a, c->hWalkEntries.data() + a, (b - a) * sizeof(uint2), hipMemcpyHostToDevice, cs));
                }
            }
            if (gathered != 0)
            {
                // One kernel of a few workgroups reads the tables over the link (pd::launchGatherHostEntries). Measured against it on
                // 4,000 samples x 4 Mb, same box: a copy per read group on the copy engines (POPDEL_HIP_WALK_GATHER=copies) -- 4,000
                // hipMemcpyAsync calls per segment, 0.083 instead of 0.064 s per round.
                static const bool byCopies = [] {
                    const char * e = std::getenv("POPDEL_HIP_WALK_GATHER");
                    return e && e[0] == 'c';
                }();
                if (!byCopies)
                {
                    c->dWalkSrc.uploadPinned(c->hWalkSrc, cs);
                    pd::launchGatherHostEntries(c->dWalkSrc.p, dEntryOffsets.p, c->dWalkEntries.p, nRg, cs);
                    PD_HIP(hipGetLastError());
                }
                else
                    for (uint32_t r = 0; r < nRg; ++r)
                        if (c->hWalkSrc[r])
                            PD_HIP(hipMemcpyAsync(c->dWalkEntries.p + c->hRgOffsets[r], wk->rg_entries[r],
                                                  (c->hRgOffsets[r + 1] - c->hRgOffsets[r]) * sizeof(uint2), hipMemcpyHostToDevice, cs));
            }
            c->walkGathered += gathered;
            c->walkStaged += nEntries - gathered - fromDevice;
        }
        // ---- instances per read group, their offsets (the build's rgOffsets), then the instances and the new state
        pd::WalkInput in;
        in.rgOffsets = dEntryOffsets.p;
        in.entries = c->dWalkEntries.p;
        in.endBorder = c->dWalkBorder.p;
        in.idOffset = c->dWalkIdOffset.p;
        in.switches = c->dWalkFlags.p;
        in.equalRun = c->dWalkFlags.p + nRg;
        pd::WalkParams wp{wk->first_window_pos, wk->step_pos, wk->has_step ? 1u : 0u, W, wk->window_shift};
        c->dWalkCount.ensure(nRg);
        c->dRgOffsets.ensure(nRg + 1);
        PD_HIP(hipMemsetAsync(c->dWalkTotals.p, 0, 5 * sizeof(unsigned long long), cs));
        pd::launchWalkCount(in, walkStateOf(*c, c->walkCur), wp, c->dRg.p, nRg, c->dWalkCount.p, c->dRgOffsets.p,
                            reinterpret_cast<uint64_t *>(c->dWalkTotals.p), reinterpret_cast<unsigned *>(c->dWalkTotals.p + 3), cs);
        PD_HIP(hipGetLastError());
        PD_HIP(hipMemcpyAsync(c->hWalkTotals.data(), c->dWalkTotals.p, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
        PD_HIP(hipStreamSynchronize(cs));
        const uint64_t nInst = c->hWalkTotals[0];
        {
            // the count pass evaluates the predicates of the emit pass: the lists are made long enough before it runs
            // (need[0], need[1]: two 32-bit words at the start of totals[3])
            const unsigned * need = reinterpret_cast<const unsigned *>(c->hWalkTotals.data() + 3);
            if (need[0] > c->walkCapActive || need[1] > c->walkCapPending)
                growWalkState(*c, need[0], need[1]);
        }
        const pd::WalkState cur = walkStateOf(*c, c->walkCur), nxt = walkStateOf(*c, c->walkCur ^ 1);
        for (int k = 0; k < 4; ++k)
            c->dIn32[k].ensure(std::max<uint64_t>(1, nInst));
        pd::launchWalkEmit(in, cur, wp, c->dRg.p, nRg, c->dRgOffsets.p, reinterpret_cast<int32_t *>(c->dIn32[0].p), c->dIn32[1].p, c->dIn32[2].p,
                           c->dIn32[3].p, nxt, reinterpret_cast<unsigned *>(c->dWalkTotals.p + 2), c->dWalkTotals.p + 1, cs);
        PD_HIP(hipGetLastError());
        PD_HIP(hipMemcpyAsync(c->hWalkTotals.data() + 1, c->dWalkTotals.p + 1, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, cs));
        bool built = false;
        const bool deviceBuild = pd::buildLdsBytes(W, 48) <= 160u * 1024u;     // (else: more windows than the build kernels keep in LDS)
        if (nInst != 0 && deviceBuild)
        {
            pd::InstanceArrays ia;
            std::memset(&ia, 0, sizeof(ia));
            ia.wide = 1;
            ia.dev32 = reinterpret_cast<const int32_t *>(c->dIn32[0].p);
            ia.on32 = c->dIn32[1].p;
            ia.off32 = c->dIn32[2].p;
            ia.start32 = c->dIn32[3].p;
            ia.firstPos = wk->first_window_pos;
            ia.shift = wk->window_shift;
            built = buildLayoutOnDevice(*c, ia, nInst, W, *s, nullptr);
        }
        else
            PD_HIP(hipStreamSynchronize(cs));
        // (the build's first synchronisation point lies behind the copy above)
        if (c->hWalkTotals[2] != 0)
            throw HipError{"a read group has more active or pending pairs than the device-side window loop keeps (run with POPDEL_HOST_WALK=1)"};
        // the segment is taken: from here on the state after it is the current one (a failure above leaves the old state)
        c->walkCur ^= 1;
        c->walkActive = c->hWalkTotals[1];
        if (nInst == 0)
        {
            // no pair is active in any window: the (host-side) path for empty segments
            putSpare(c, std::move(s));
            std::vector<uint64_t> zeros(nRg + 1, 0);
            pd_segment empty;
            std::memset(&empty, 0, sizeof(empty));
            empty.first_window_pos = wk->first_window_pos;
            empty.n_windows = W;
            empty.window_shift = wk->window_shift;
            empty.n_readgroups = nRg;
            empty.rg_offsets = zeros.data();
            return pd_segment_submit(c, &empty, ticket);
        }
        if (!built)
        {
            // a segment the device-side layout build does not take: its instances come back and go the way of pd_segment
            putSpare(c, std::move(s));
            std::vector<uint64_t> off(nRg + 1);
            std::vector<int32_t> dv(nInst);
            std::vector<uint32_t> on(nInst), of(nInst), st(nInst);
            PD_HIP(hipMemcpy(off.data(), c->dRgOffsets.p, (nRg + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(dv.data(), c->dIn32[0].p, nInst * sizeof(int32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(on.data(), c->dIn32[1].p, nInst * sizeof(uint32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(of.data(), c->dIn32[2].p, nInst * sizeof(uint32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(st.data(), c->dIn32[3].p, nInst * sizeof(uint32_t), hipMemcpyDeviceToHost));
            pd_segment seg;
            std::memset(&seg, 0, sizeof(seg));
            seg.first_window_pos = wk->first_window_pos;
            seg.n_windows = W;
            seg.window_shift = wk->window_shift;
            seg.n_readgroups = nRg;
            seg.rg_offsets = off.data();
            seg.pair_dev = dv.data();
            seg.pair_start = st.data();
            seg.pair_on = on.data();
            seg.pair_off = of.data();
            return pd_segment_submit(c, &seg, ticket);
        }
        s->scanRg = c->dScanRg.p;
        s->nRealRg = c->nRg;
        if (!s->built)
            PD_HIP(hipEventCreateWithFlags(&s->built, hipEventDisableTiming));
        PD_HIP(hipEventRecord(s->built, cs));
        s->waitBuilt = true;
        handle = c->nextHandle++;
        putStaged(c, handle, std::move(s));
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_walk_submit: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, std::string("pd_walk_submit: ") + e.what());
    }
    catch (...)
    {
        return fail(c, "pd_walk_submit: unknown exception");
    }
    if (int rc = pd_segment_run(c, handle, ticket))
    {
        std::lock_guard<std::mutex> guard(c->stagedLock);
        c->staged.erase(handle);
        return rc;
    }
    return 0;
}

int pd_walk_active(pd_ctx * c, uint64_t * nActive, uint32_t * rg, uint32_t * start, int32_t * dev, uint32_t * id, uint64_t capacity)
{
    return pd_walk_active_ex(c, nActive, rg, start, dev, id, nullptr, capacity);
}

int pd_walk_active_ex(pd_ctx * c, uint64_t * nActive, uint32_t * rg, uint32_t * start, int32_t * dev, uint32_t * id, uint32_t * last, uint64_t capacity)
{
    if (!c || !nActive)
        return fail(c, "pd_walk_active: invalid argument");
    *nActive = c->walkReady ? c->walkActive : 0;
    if (!rg || !start || !dev || !id || *nActive == 0)
        return 0;
    if (capacity < *nActive)
        return 2;
    try
    {
        PD_HIP(hipSetDevice(c->device));
        PD_HIP(hipStreamSynchronize(c->copyStream));
        pd_ctx::WalkBuffers & b = c->walkState[c->walkCur];
        std::vector<uint32_t> n(c->nRg);
        PD_HIP(hipMemcpy(n.data(), b.nActive.p, c->nRg * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t at = 0;
        for (uint32_t r = 0; r < c->nRg; ++r)
        {
            if (n[r] == 0)
                continue;
            if (at + n[r] > capacity)
                return 2;
            const size_t base = static_cast<size_t>(r) * c->walkCapActive;
            PD_HIP(hipMemcpy(start + at, b.aStart.p + base, n[r] * sizeof(uint32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(dev + at, b.aDev.p + base, n[r] * sizeof(int32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(id + at, b.aId.p + base, n[r] * sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (last)
                PD_HIP(hipMemcpy(last + at, b.aLast.p + base, n[r] * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < n[r]; ++k)
                rg[at + k] = r;
            at += n[r];
        }
        *nActive = at;
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_walk_active: " + e.what);
    }
    catch (...)
    {
        return fail(c, "pd_walk_active: unknown exception");
    }
    return 0;
}

int pd_segment_collect(pd_ctx * c, uint64_t ticket, pd_calls ** out)
{
    if (!c || !out)
        return fail(c, "pd_segment_collect: invalid argument");
    *out = nullptr;
    if (c->lanes)
        return laneCollect(c, ticket, out);
    auto rit = c->runs.find(ticket);
    if (rit == c->runs.end())
        return fail(c, "pd_segment_collect: unknown ticket");
    if (rit != c->runs.begin())
        return fail(c, "pd_segment_collect: tickets are collected in the order they were issued");
    Run run = rit->second;
    c->runs.erase(rit);
    StagedSegment * found = findStaged(c, run.handle);
    if (!found)
    {
        (void)hipEventDestroy(run.s1Begin);
        (void)hipEventDestroy(run.s1End);
        if (run.scanDone)
            (void)hipEventDestroy(run.scanDone);
        return fail(c, "pd_segment_collect: segment was unstaged before collect");
    }
    StagedSegment & s = *found;
    hipEvent_t s2Begin = nullptr, s2End = nullptr;
    bool s2BeginRecorded = false;
    PinnedBlock pinned;
    int rc = 0;
    try
    {
        PD_HIP(hipSetDevice(c->device));
        if (!run.launched)              // the collect that should have started it failed
            launchScan(c, run);
        if (run.scanDone)
            PD_HIP(hipStreamWaitEvent(c->stream, run.scanDone, 0));
        Stage1Out & o = c->s1[run.set];
        const uint32_t W = s.nWindows, S = c->nSamples;
        const pd::SegmentView view = viewOf(s);
        PD_HIP(hipEventCreate(&s2Begin));
        PD_HIP(hipEventCreate(&s2End));

        CtxExchange xch(*c);
        if (c->sharded && c->world > 1 && !c->shardSettingsChecked)
        {
            // The contexts of a cohort take the same decisions only if they were set up alike: calling parameters, the byte
            // budget that cuts stage 2 into batches, which lists and pools are formed where. One exchange at the first
            // segment compares a fingerprint of them; ranks that differ fail here instead of hanging in mismatched exchanges.
            uint64_t fp = 1469598103934665603ull;
            auto mix = [&fp](const void * p, size_t n) {
                const uint8_t * b = static_cast<const uint8_t *>(p);
                for (size_t i = 0; i < n; ++i)
                    fp = (fp ^ b[i]) * 1099511628211ull;
            };
            const uint64_t budgetWord = c->stage2Budget;
            const uint32_t words[8] = {c->cfg.iterations, c->cfg.min_len, (uint32_t)c->cfg.somatic, (uint32_t)c->cfg.window_wise_output,
                                       c->optimisticPools ? 1u : 0u, c->deviceLists ? 1u : 0u, c->nSamplesGlobal, (uint32_t)PD_ABI_VERSION};
            mix(&budgetWord, sizeof(budgetWord));
            mix(words, sizeof(words));
            mix(&c->cfg.min_likelihood_ratio, sizeof(double));
            mix(&c->cfg.min_sample_fraction, sizeof(double));
            const size_t per = 256;
            uint8_t * area = reserveArena(*c, per * c->world);
            PinnedBuf<uint64_t> & h = c->hShardCheck;
            h.resize(static_cast<size_t>(c->world) * (per / sizeof(uint64_t)));
            h[0] = fp;
            PD_HIP(hipMemcpyAsync(area + static_cast<size_t>(c->rank) * per, h.data(), sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
            xch.allGather(area, per);
            PD_HIP(hipMemcpyAsync(h.data(), area, per * c->world, hipMemcpyDeviceToHost, c->stream));
            PD_HIP(hipStreamSynchronize(c->stream));
            for (uint32_t r = 0; r < c->world; ++r)
                if (h[static_cast<size_t>(r) * (per / sizeof(uint64_t))] != fp)
                    throw HipError{"the contexts of the sharded cohort were set up differently (calling parameters, POPDEL_HIP_STAGE2_BYTES, "
                                   "POPDEL_HIP_EXACT_POOLS or POPDEL_HIP_HOST_LISTS differ between rank " + std::to_string(c->rank) + " and rank " +
                                   std::to_string(r) + ")"};
            c->shardSettingsChecked = true;
        }
        c->nAllGathers = 0;
        c->allGatherBytes = 0;
        c->xchEventsUsed = 0;
        c->xchHostMs = 0;
        const uint32_t world = c->world, sper = c->samplesPerRank, SG = c->nSamplesGlobal;
        unsigned rowLen = 1;
        while (rowLen < SG)
            rowLen <<= 1;
        std::vector<pd::CellState> cells;
        auto addCell = [&](int window, int initLen) {
            pd::CellState cs;
            std::memset(&cs, 0, sizeof(cs));
            cs.window = window;
            cs.initLen = initLen;
            cs.len = static_cast<unsigned>(initLen);             // (the likelihoods ahead of the EM loop run before the cell's set-up kernel)
            cs.phase = pd::kCellIterating;
            cells.push_back(cs);
        };
        // Flagged windows (some sample of the cohort has a Q3 above the smallest threshold) -> candidate lengths per
        // window -> (window, length) cells. Both lists are formed on the device and the host reads the cells in ONE round
        // trip; the kernels are launched for a capacity of flagged windows and find the number on the device. A sharded
        // cohort exchanges the window flags, then the Q3 values of the flagged windows -- laid out for a capacity guessed
        // from the previous segments (every context sees the same union of flagged windows, so all guess alike); a segment
        // with more flagged windows is listed again with room for them.
        const bool sharded = c->sharded;
        bool listed = false;
        if (W != 0 && c->deviceLists)
        {
            PD_HIP(hipEventRecord(s2Begin, c->stream));
            s2BeginRecorded = true;
            const unsigned capMax = std::min<unsigned>(W, std::max<unsigned>(256u, std::min<unsigned>(4096u, (1u << 28) / rowLen)));
            unsigned capF = sharded ? std::min(capMax, std::max(64u, c->flagCapGuess)) : capMax;
            constexpr unsigned kListCap = 1022;                                 // cells the first copy brings along
            const size_t flagBytes = sharded ? alignUp(static_cast<size_t>(W) * sizeof(int), 256) : 0;     // per context
            for (;;)
            {
                const size_t listRoom = static_cast<size_t>(capF) * pd::kMaxCandidatesPerWindow;
                c->dFlagged.ensure(capF);
                c->dCandHdr.ensure(4 + 2 * std::max<size_t>(listRoom, kListCap));      // (the first copy always brings kListCap cells' room along)
                c->dClusterScratch.ensure(static_cast<size_t>(capF) * rowLen);
                c->dCandLen.ensure(listRoom);
                c->dCandCount.ensure(capF);
                pd::ClusterInput in;
                in.chunkBytes = alignUp(capF * static_cast<size_t>(sper) * 5, 256);
                uint8_t * area = reserveArena(*c, world * (flagBytes + in.chunkBytes));
                const int * flagRows = o.winFlag.p;
                if (sharded)
                {
                    pd::launchCopyBytes(o.winFlag.p, area + static_cast<size_t>(c->rank) * flagBytes, W * sizeof(int), c->stream);
                    xch.allGather(area, flagBytes);
                    flagRows = reinterpret_cast<const int *>(area);
                }
                in.base = area + world * flagBytes;
                in.sper = sper;
                in.world = world;
                in.nFlagged = capF;
                in.nFlaggedDev = c->dCandHdr.p;
                in.minInit = c->dMinInitGlobal.p;
                pd::launchCompactFlagged(flagRows, W, world, flagBytes / sizeof(int), capF, c->dFlagged.p, c->dCandHdr.p, c->stream);
                if (o.screened)
                {
                    // the screen knows WHICH windows may carry a candidate; their exact Q3 values, all samples, now
                    o.exactRest.ensure(1 + static_cast<size_t>(capF) * ((S + 63) / 64));
                    PD_HIP(hipMemsetAsync(o.exactRest.p, 0, sizeof(unsigned), c->stream));
                    pd::launchExactFlagged(view, c->cohort, c->dFlagged.p, capF, c->dCandHdr.p, o.winFlag.p, o.tileTodo.p, o.q3.p, o.flags.p,
                                           o.winFlag.p, o.exactRest.p, c->stream);
                }
                pd::launchPackFlagged(o.q3.p, o.flags.p, c->dFlagged.p, W, S, in, c->rank, c->stream);
                PD_HIP(hipGetLastError());
                xch.allGather(const_cast<uint8_t *>(in.base), in.chunkBytes);
                pd::launchCluster(in, c->uniformThreshold, c->dClusterScratch.p, rowLen, c->dCandLen.p, c->dCandCount.p, c->stream);
                pd::launchListCells(c->dFlagged.p, c->dCandLen.p, c->dCandCount.p, capF, (unsigned)listRoom, c->dCandHdr.p, c->stream);
                PD_HIP(hipGetLastError());
                PinnedBuf<unsigned> & hdr = c->hCandHdr;
                hdr.resize(4 + 2 * kListCap);
                PD_HIP(hipMemcpyAsync(hdr.data(), c->dCandHdr.p, hdr.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
                PD_HIP(hipStreamSynchronize(c->stream));
                checkExchange(*c);
                const unsigned nFlagged = hdr[0], nListed = hdr[1];
                if (getenv("POPDEL_HIP_DEBUG"))
                    fprintf(stderr, "[popdel_hip] %u flagged windows (room for %u), %u candidate cells\n", nFlagged, capF, nListed);
                if (sharded)                                    // room for a quarter more next time
                    c->flagCapGuess = std::min(capMax, (nFlagged + nFlagged / 4 + 32u + 63u) / 64u * 64u);
                if (nFlagged <= capF)
                {
                    if (hdr[2] != 0)
                        throw HipError{"more than 32 candidate deletion lengths in one window"};
                    if (nListed > kListCap)                         // a candidate-rich segment: fetch the rest of the list
                    {
                        hdr.resize(4 + 2 * static_cast<size_t>(nListed));
                        PD_HIP(hipMemcpyAsync(hdr.data(), c->dCandHdr.p, hdr.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
                        PD_HIP(hipStreamSynchronize(c->stream));
                    }
                    for (unsigned i = 0; i < nListed; ++i)
                        addCell((int)hdr[4 + 2 * i], (int)hdr[5 + 2 * i]);
                    listed = true;
                    break;
                }
                if (capF >= capMax)
                    break;          // more flagged windows than the device-side lists are made for: the host-side lists below
                capF = std::min(capMax, (nFlagged + 63u) / 64u * 64u);
            }
        }
        if (!listed)
        {
            PinnedBuf<int> & winFlag = c->hWinFlag;
            winFlag.resize(static_cast<size_t>(W) * world);
            if (!sharded)
            {
                PD_HIP(hipMemcpyAsync(winFlag.data(), o.winFlag.p, W * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            }
            else
            {
                uint8_t * area = reserveArena(*c, static_cast<size_t>(W) * world * sizeof(int));
                pd::launchCopyBytes(o.winFlag.p, area + static_cast<size_t>(c->rank) * W * sizeof(int), W * sizeof(int), c->stream);
                xch.allGather(area, W * sizeof(int));
                PD_HIP(hipMemcpyAsync(winFlag.data(), area, winFlag.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            }
            PD_HIP(hipStreamSynchronize(c->stream));
            std::vector<int> flagged;
            for (uint32_t w = 0; w < W; ++w)
            {
                int any = 0;
                for (uint32_t r = 0; r < world; ++r)
                    any |= winFlag[static_cast<size_t>(r) * W + w];
                if (any)
                    flagged.push_back((int)w);
            }
            if (!s2BeginRecorded)
                PD_HIP(hipEventRecord(s2Begin, c->stream));
            if (!flagged.empty())
            {
                c->hFlagged.resize(flagged.size());
                std::memcpy(c->hFlagged.data(), flagged.data(), flagged.size() * sizeof(int));
                c->dFlagged.uploadPinned(c->hFlagged, c->stream);
                c->dClusterScratch.ensure(static_cast<size_t>(flagged.size()) * rowLen);
                c->dCandLen.ensure(flagged.size() * pd::kMaxCandidatesPerWindow);
                c->dCandCount.ensure(flagged.size());
                pd::ClusterInput in;
                in.chunkBytes = alignUp(flagged.size() * static_cast<size_t>(sper) * 5, 256);
                in.base = reserveArena(*c, in.chunkBytes * world);
                in.sper = sper;
                in.world = world;
                in.nFlagged = (unsigned)flagged.size();
                in.nFlaggedDev = nullptr;
                in.minInit = c->dMinInitGlobal.p;
                if (o.screened)
                {
                    o.exactRest.ensure(1 + std::min<size_t>(32768, flagged.size()) * ((S + 63) / 64));
                    for (size_t f0 = 0; f0 < flagged.size(); f0 += 32768)       // (a grid's y dimension holds 65,535)
                    {
                        PD_HIP(hipMemsetAsync(o.exactRest.p, 0, sizeof(unsigned), c->stream));
                        pd::launchExactFlagged(view, c->cohort, c->dFlagged.p + f0, (unsigned)std::min<size_t>(32768, flagged.size() - f0), nullptr,
                                               o.winFlag.p, o.tileTodo.p, o.q3.p, o.flags.p, o.winFlag.p, o.exactRest.p, c->stream);
                    }
                }
                pd::launchPackFlagged(o.q3.p, o.flags.p, c->dFlagged.p, W, S, in, c->rank, c->stream);
                PD_HIP(hipGetLastError());
                xch.allGather(c->arena, in.chunkBytes);
                pd::launchCluster(in, c->uniformThreshold, c->dClusterScratch.p, rowLen, c->dCandLen.p, c->dCandCount.p, c->stream);
                PD_HIP(hipGetLastError());
                PinnedBuf<int> &candLen = c->hCandLen, &candCount = c->hCandCount;
                candLen.resize(flagged.size() * pd::kMaxCandidatesPerWindow);
                candCount.resize(flagged.size());
                PD_HIP(hipMemcpyAsync(candLen.data(), c->dCandLen.p, candLen.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                PD_HIP(hipMemcpyAsync(candCount.data(), c->dCandCount.p, candCount.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
                PD_HIP(hipStreamSynchronize(c->stream));
                for (size_t f = 0; f < flagged.size(); ++f)
                {
                    if (candCount[f] > pd::kMaxCandidatesPerWindow)
                        throw HipError{"more than 32 candidate deletion lengths in one window"};
                    for (int k = 0; k < candCount[f]; ++k)
                        addCell(flagged[f], candLen[f * pd::kMaxCandidatesPerWindow + k]);
                }
            }
        }
        // Stage 2 runs over the cells in batches that keep its per-(cell, sample) buffers (dense pair image 512 B,
        // triples 24 B, outputs 52 B, ...) within a byte budget: a candidate-rich segment of a large cohort would not
        // fit otherwise (13,000 cells x 50,000 samples = 390 GB). One batch is the normal case.
        std::vector<unsigned> emit;
        uint64_t likUnits = 0;          // (cell, sample) units of the last batch's data-likelihood launch ahead of the loop
        const unsigned nCells = (unsigned)cells.size();
        constexpr uint64_t kHeavyUnits = 5000000;
        std::unique_lock<std::mutex> heavy;
        // (not in a cohort over several ranks: lane k here exchanges with lane k there, and two ranks that hold the lock for
        // different lanes would wait for each other)
        if (c->owner && c->world == 1 && static_cast<uint64_t>(nCells) * S > kHeavyUnits)
            heavy = std::unique_lock<std::mutex>(c->owner->heavyStage2);
        const size_t budget = c->stage2Budget;
        // (from the samples per context of the partition, not from this context's own count: all contexts of a sharded
        // cohort must cut the cell list at the same places, the exchanges run per batch)
        const size_t bytesPerCell = static_cast<size_t>(sper) * (2 * pd::kDenseCap * 2 + 24 + 52 + 2 + 24) + 4096;
        // ((sweep, cell) is a grid's y dimension: at most 65,535 / sweeps cells per batch; pd::kTilesPerSweep)
        // (like bytesPerCell from the partition's samples per context, not from S: a last context with fewer tiles must cut
        // the cell list at the same places)
        const size_t sweeps = ((static_cast<size_t>(sper) + pd::kSamplesPerBlock - 1) / pd::kSamplesPerBlock + 7) / 8 * 8 / pd::kTilesPerSweep + 1;
        const unsigned batchCells = static_cast<unsigned>(
            std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nCells ? nCells : 1, 65535 / sweeps), budget / bytesPerCell)));
        const unsigned nBatches = nCells ? (nCells + batchCells - 1) / batchCells : 0;
        pd::Stage2Buffers b;
        std::memset(&b, 0, sizeof(b));
        c->hCells.resize(std::max<size_t>(1, cells.size()));
        if (!cells.empty())
            std::memcpy(c->hCells.data(), cells.data(), cells.size() * sizeof(pd::CellState));
        c->hIterFlag.resize(4);
        // per-sample results of the emitted cells: device buffer of the current batch, host block of the whole segment
        uint32_t *hPl = nullptr, *hLad = nullptr, *hDad = nullptr, *hFl = nullptr;
        std::vector<uint32_t> accPl, accLad, accDad, accFl;       // only used when there is more than one batch
        uint32_t nLastBatch = 0;                                  // emitted cells of the last batch (still on the device)
        bool tailRecorded = false;                                // s2End / resultsPacked are on the stream
        for (unsigned batch = 0; batch < nBatches; ++batch)
        {
            const unsigned lo = batch * batchCells, hi = std::min(nCells, lo + batchCells), nb = hi - lo;
            b.nCells = nb;
            b.nTiles = (SG + pd::kSamplesPerBlock - 1) / pd::kSamplesPerBlock;
            b.tilesPerRank = sper / pd::kSamplesPerBlock;
            b.rank = c->rank;
            b.world = world;
            b.nSamplesGlobal = SG;
            b.sharded = sharded ? 1u : 0u;
            // exchange areas: xch (header, integer counts, pool records, per-tile sums, RG-0 record) | digit histograms
            b.xchChunk = pd::xchChunkDoubles(b.tilesPerRank, nb);
            const size_t xchBytes = alignUp(static_cast<size_t>(world) * b.xchChunk * sizeof(double), 256);
            const size_t histBytes = sharded ? static_cast<size_t>(world) * nb * 2 * 256 * sizeof(unsigned) : 0;
            uint8_t * area = reserveArena(*c, xchBytes + histBytes);
            b.xch = reinterpret_cast<double *>(area);
            b.selHist = reinterpret_cast<unsigned *>(area + xchBytes);
            c->dCells.ensure(nb);
            c->dDl.ensure(static_cast<size_t>(nb) * S * 3);
            c->dShifts.ensure(static_cast<size_t>(nb) * c->nRg);
            b.ownTiles = (S + pd::kSamplesPerBlock - 1) / pd::kSamplesPerBlock;
            // dense images: one per distinct window of the batch (the cells are ordered by window)
            unsigned nImg = 0;
            for (unsigned i = 0; i < nb; ++i)
            {
                pd::CellState & cs = c->hCells[lo + i];
                if (i == 0 || cs.window != c->hCells[lo + i - 1].window)
                {
                    if (nImg)
                        c->hCells[lo + nImg - 1].imageEnd = i;
                    c->hCells[lo + nImg++].imageFirst = i;          // (image numbers never run ahead of cell numbers)
                }
                cs.image = nImg - 1;
            }
            c->hCells[lo + nImg - 1].imageEnd = nb;
            b.nImages = nImg;
            const size_t waveTiles = static_cast<size_t>(nImg) * b.ownTiles * 4;
            c->dCellDev.ensure(waveTiles * pd::kDenseCap * 64);
            c->dCellStart.ensure(waveTiles * pd::kDenseCap * 64);
            c->dCellN.ensure(static_cast<size_t>(nImg) * S);
            c->dWaveRows.ensure(waveTiles);
            c->dTileMode.ensure(static_cast<size_t>(nImg) * b.ownTiles);
            b.cellDev = c->dCellDev.p;
            b.cellStart = c->dCellStart.p;
            b.cellN = c->dCellN.p;
            b.waveRows = c->dWaveRows.p;
            b.tileMode = c->dTileMode.p;
            b.mt = c->multiTiles;
            b.nRg = c->nRg;
            if (b.mt.n != 0)
            {
                const size_t mWaves = static_cast<size_t>(nImg) * b.mt.nWaves;
                c->dMDev.ensure(mWaves * pd::kDenseCap * 64);
                c->dMStart.ensure(mWaves * pd::kDenseCap * 64);
                c->dMN.ensure(static_cast<size_t>(nImg) * c->nRg);
                c->dMWaveRows.ensure(mWaves);
                c->dRgOutD.ensure(static_cast<size_t>(nb) * c->nRg * 8);
                c->dRgOutU.ensure(static_cast<size_t>(nb) * c->nRg * 12);
                b.mDev = c->dMDev.p;
                b.mStart = c->dMStart.p;
                b.mN = c->dMN.p;
                b.mWaveRows = c->dMWaveRows.p;
                b.rgOutD = c->dRgOutD.p;
                b.rgOutU = c->dRgOutU.p;
            }
            c->dEmit.ensure(static_cast<size_t>(nb) + 1);
            c->dOutPl.ensure(static_cast<size_t>(nb) * S * 13);
            c->dPl.ensure(static_cast<size_t>(nb) * S * 3);
            c->dLad.ensure(static_cast<size_t>(nb) * S * 3);
            c->dDad.ensure(static_cast<size_t>(nb) * S * 5);
            c->dFl.ensure(static_cast<size_t>(nb) * S * 2);
            b.cells = c->dCells.p;
            b.dl = c->dDl.p;
            b.shifts = c->dShifts.p;
            b.iterFlags = c->dIterFlags.p;
            b.pl = c->dPl.p;
            b.lad = c->dLad.p;
            b.dad = c->dDad.p;
            b.fl = c->dFl.p;
            b.minLen = c->cfg.min_len;
            b.maxIterations = c->cfg.iterations;
            b.minLR = c->cfg.min_likelihood_ratio;
            b.minSampleFraction = c->cfg.min_sample_fraction;
            b.somatic = c->cfg.somatic;
            b.windowWise = c->cfg.window_wise_output;
            // Phase A gives, per cell, the number of usable active pairs = capacity of its support pool, and tells whether
            // any tile needs the generic kernels. The first attempt does not wait for either (one host round trip less):
            // pools sized from the segment's average with headroom, generic kernels as earlier segments needed them, both
            // checked on the device (Stage2Buffers::guard); the second attempt, if the check failed, reads the counts.
            // The contexts of a sharded cohort exchange their verdicts with the first exchange of phase B, so all of them
            // repeat a batch or none does; in the second attempt they read the counts of all contexts.
            for (int attempt = 0; attempt < 2; ++attempt)
            {
                const bool optimistic = attempt == 0 && c->optimisticPools;
                unsigned * guard = c->dIterFlags.p + pd::kMaxIterationsCap + 1;
                pd::ZeroRanges zero{};
                zero.add(area, xchBytes);
                zero.add(c->dShifts.p, static_cast<size_t>(nb) * c->nRg * sizeof(int));
                zero.add(c->dTileMode.p, static_cast<size_t>(nImg) * b.ownTiles * sizeof(unsigned));
                zero.add(c->dIterFlags.p, (pd::kMaxIterationsCap + 2) * sizeof(unsigned));     // EM flags and the guard word
                pd::launchZero(zero, c->stream);
                if (attempt != 0)       // (the first attempt's cell records came back into the staging buffer)
                {
                    for (unsigned i = 0; i < nb; ++i)
                    {
                        pd::CellState fresh = cells[lo + i];
                        fresh.image = c->hCells[lo + i].image;
                        fresh.imageFirst = c->hCells[lo + i].imageFirst;
                        fresh.imageEnd = c->hCells[lo + i].imageEnd;
                        c->hCells[lo + i] = fresh;
                    }
                }
                PD_HIP(hipMemcpyAsync(c->dCells.p, c->hCells.data() + lo, nb * sizeof(pd::CellState), hipMemcpyHostToDevice, c->stream));
                bool generic = c->genericSeen;
                b.guard = nullptr;
                b.rangeSelect = 0;
                pd::runStage2Init(view, c->cohort, b, o.flags.p, c->stream);
                PD_HIP(hipGetLastError());
                if (optimistic)
                {
                    const uint64_t perCell = s.nActivePairs * 3 / (2 * static_cast<uint64_t>(W)) + 1024;
                    c->dSuppFirst.ensure(static_cast<size_t>(nb) * perCell + 1);
                    c->dSuppLast.ensure(static_cast<size_t>(nb) * perCell + 1);
                    b.suppCap = std::min(c->dSuppFirst.n, c->dSuppLast.n) - 1;
                    b.genericLaunched = generic ? 1u : 0u;
                    b.rangeSelect = sharded ? 1u : 0u;
                    b.guard = guard;
                }
                else
                {
                    // the integer counts of every context's samples (own ones: pool size; all: generic kernels or not)
                    xch.allGather(b.xch, b.xchChunk * sizeof(double));
                    PinnedBuf<unsigned> & counters = c->hCounters;
                    counters.resize(static_cast<size_t>(world) * nb * 4);
                    for (uint32_t r = 0; r < world; ++r)
                        PD_HIP(hipMemcpyAsync(counters.data() + static_cast<size_t>(r) * nb * 4,
                                              b.xch + static_cast<size_t>(r) * b.xchChunk + pd::kXchHeaderDoubles,
                                              static_cast<size_t>(nb) * 4 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
                    PD_HIP(hipStreamSynchronize(c->stream));
                    uint64_t pool = 0, genericTiles = 0;
                    for (uint32_t r = 0; r < world; ++r)
                        for (unsigned i = 0; i < nb; ++i)
                        {
                            if (r == c->rank)
                                pool += counters[(static_cast<size_t>(r) * nb + i) * 4 + 1];
                            genericTiles += counters[(static_cast<size_t>(r) * nb + i) * 4 + 3];
                        }
                    if (getenv("POPDEL_HIP_DEBUG"))
                        fprintf(stderr, "[popdel_hip] stage 2 batch %u/%u%s: %u cells, %llu (cell, 64-sample wave tile) units of the cohort need the generic kernels\n",
                                batch + 1, nBatches, attempt ? " (repeated)" : "", nb, (unsigned long long)genericTiles);
                    c->dSuppFirst.ensure(pool + 1);
                    c->dSuppLast.ensure(pool + 1);
                    generic = genericTiles != 0;
                    if (generic)
                        c->genericSeen = true;      // from now on the generic kernels ride along (they exit early where not needed)
                }
                b.suppFirst = c->dSuppFirst.p;
                b.suppLast = c->dSuppLast.p;
                if (!c->likBegin)
                {
                    PD_HIP(hipEventCreate(&c->likBegin));
                    PD_HIP(hipEventCreate(&c->likEnd));
                }
                pd::runStage2Main(view, c->cohort, b, o.flags.p, xch, generic, c->hIterFlag.data(), c->likBegin, c->likEnd, c->stream);
                likUnits = static_cast<uint64_t>(nb) * S;
                PD_HIP(hipGetLastError());
                // the columns of the emitted cells are packed before the host has seen which cells those are
                // (the list is formed on the device; room for all cells of the batch)
                pd::launchGatherOut(b, c->dEmit.p, S, c->dOutPl.p, c->stream);
                PD_HIP(hipGetLastError());
                const bool lastBatch = batch + 1 == nBatches;
                if (lastBatch)
                {
                    PD_HIP(hipEventRecord(s2End, c->stream));
                    PD_HIP(hipEventRecord(c->resultsPacked, c->stream));
                    tailRecorded = true;
                }
                PD_HIP(hipMemcpyAsync(c->hCells.data() + lo, c->dCells.p, nb * sizeof(pd::CellState), hipMemcpyDeviceToHost, c->stream));
                PD_HIP(hipMemcpyAsync(c->hIterFlag.data() + 2, c->dEmit.p, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
                c->hIterFlag[1] = 0;
                if (optimistic)
                    PD_HIP(hipMemcpyAsync(c->hIterFlag.data() + 1, guard, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
                PD_HIP(hipEventRecord(c->cellsFetched, c->stream));
                // the scan of a waiting ticket goes onto the stream now, right behind this segment's last kernel: the host
                // reads the cells (and, on the copy stream, the columns travel) while it runs. Not earlier: the looks of the
                // EM loop wait for everything enqueued before them.
                if (lastBatch && nBatches == 1 && !c->runs.empty() && !c->runs.begin()->second.launched)
                    launchScan(c, c->runs.begin()->second);
                PD_HIP(hipEventSynchronize(c->cellsFetched));
                checkExchange(*c);
                if (!optimistic || c->hIterFlag[1] == 0)
                    break;
                ++c->nRepeatedBatches;
            }
            std::memcpy(cells.data() + lo, c->hCells.data() + lo, nb * sizeof(pd::CellState));
            uint32_t nbEmit = 0;
            for (unsigned i = 0; i < nb; ++i)
                if (cells[lo + i].phase == pd::kCellEmit)
                {
                    emit.push_back(lo + i);
                    ++nbEmit;
                }
            const size_t per = static_cast<size_t>(nbEmit) * S;
            if (nbEmit != c->hIterFlag[2])
                throw HipError{"the device packed the columns of " + std::to_string(c->hIterFlag[2]) + " calls, the host counts " +
                               std::to_string(nbEmit)};
            nLastBatch = nbEmit;
            if (nBatches > 1 && nbEmit != 0)
            {
                // not the streamlined path: fetch this batch's columns now, the device buffers are reused by the next one
                PinnedBlock tmp = takePinned(per * 13 * sizeof(uint32_t));
                PD_HIP(hipMemcpyAsync(tmp.p, c->dOutPl.p, per * 13 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                PD_HIP(hipStreamSynchronize(c->stream));
                const uint32_t * t = static_cast<const uint32_t *>(tmp.p);
                accPl.insert(accPl.end(), t, t + per * 3);
                accLad.insert(accLad.end(), t + per * 3, t + per * 6);
                accDad.insert(accDad.end(), t + per * 6, t + per * 11);
                accFl.insert(accFl.end(), t + per * 11, t + per * 13);
                givePinned(tmp);
            }
        }
        const uint32_t n = (uint32_t)emit.size();
        const size_t perSample = static_cast<size_t>(n) * S;
        // columns stay in HBM (pd_ctx_keep_results_on_device) if they are all there: one batch
        const bool keep = c->keepOnDevice && nBatches == 1 && nLastBatch != 0;
        c->keptId = 0;
        if (n != 0 && !keep)
        {
            pinned = takePinned(perSample * 13 * sizeof(uint32_t));
            hPl = static_cast<uint32_t *>(pinned.p);
            hLad = hPl + perSample * 3;
            hDad = hLad + perSample * 3;
            hFl = hDad + perSample * 5;
        }
        if (!tailRecorded)                                                   // (no cells at all)
            PD_HIP(hipEventRecord(s2End, c->stream));
        const bool direct = nBatches == 1 && nLastBatch != 0 && !keep;      // results of the only batch still sit in dOutPl
        // the scan of a waiting ticket is on the stream by now (see the batch loop); with no cells it starts here
        // (enqueued BEFORE the copy: the other order was measured to hold the scan's first packets back until the copy,
        // a blit kernel on this platform, had finished)
        if (!c->runs.empty() && !c->runs.begin()->second.launched)
            launchScan(c, c->runs.begin()->second);
        if (direct)
        {
            // ... while the results travel on the copy stream
            PD_HIP(hipStreamWaitEvent(c->copyStream, c->resultsPacked, 0));
            PD_HIP(hipMemcpyAsync(hPl, c->dOutPl.p, perSample * 13 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->copyStream));
        }
        else if (n != 0 && !keep)
        {
            std::memcpy(hPl, accPl.data(), accPl.size() * sizeof(uint32_t));
            std::memcpy(hLad, accLad.data(), accLad.size() * sizeof(uint32_t));
            std::memcpy(hDad, accDad.data(), accDad.size() * sizeof(uint32_t));
            std::memcpy(hFl, accFl.data(), accFl.size() * sizeof(uint32_t));
        }
        PD_HIP(hipEventSynchronize(s2End));
        if (heavy.owns_lock())
            heavy.unlock();
        PD_HIP(hipStreamSynchronize(c->copyStream));

        float ms1 = 0, ms2 = 0;
        PD_HIP(hipEventElapsedTime(&ms1, run.s1Begin, run.s1End));
        PD_HIP(hipEventElapsedTime(&ms2, s2Begin, s2End));
        c->timing.stage1_ms = ms1;
        c->timing.stage2_ms = ms2;
        c->timing.n_cells = static_cast<uint64_t>(S) * W;
        c->timing.n_active_pairs = s.nActivePairs;
        c->timing.n_pair_chunks = s.nChunks;
        c->timing.n_candidate_cells = nCells;
        c->timing.max_life = s.maxLife;
        c->timing.n_all_gathers = c->nAllGathers;
        {
            // (everything on the stream has completed: s2End was waited for above and the exchanges precede it)
            double ms = c->xchHostMs;
            for (size_t i = 0; i + 1 < c->xchEventsUsed; i += 2)
            {
                float t = 0;
                PD_HIP(hipEventElapsedTime(&t, c->xchEvents[i], c->xchEvents[i + 1]));
                ms += t;
            }
            c->timing.exchange_ms = ms;
        }
        c->timing.lik_first_ms = 0;
        c->timing.lik_first_units = likUnits;
        if (likUnits != 0)
        {
            float t = 0;
            PD_HIP(hipEventElapsedTime(&t, c->likBegin, c->likEnd));
            c->timing.lik_first_ms = t;
        }
        {
            uint64_t nearInteger = 0;
            for (const pd::CellState & cs : cells)
                nearInteger += cs.nearIntegerShifts;
            c->timing.n_near_integer_shifts = nearInteger;
        }
        c->timing.all_gather_bytes = c->allGatherBytes;
        c->timing.n_repeated_batches = c->nRepeatedBatches;

        CallsBox * box = new CallsBox();
        std::memset(&box->pub, 0, sizeof(pd_calls));
        box->pinned = pinned;
        pd_calls * pc = &box->pub;
        pc->n_calls = n;
        pc->n_samples = S;
        std::vector<uint32_t> u(n);
        std::vector<double> d(n);
        std::vector<uint8_t> f(n);
        const uint32_t shift = s.shift, firstPos = s.firstPos;
#define PD_FILL_U(dst, expr) for (uint32_t i = 0; i < n; ++i) { const pd::CellState & cs = cells[emit[i]]; u[i] = (expr); } pc->dst = mallocCopy(u);
        PD_FILL_U(initial_length, (uint32_t)cs.initLen)
        PD_FILL_U(iterations, (uint32_t)cs.iterations)
        PD_FILL_U(deletion_length, cs.len)
        PD_FILL_U(window_position, firstPos + (uint32_t)cs.window * shift - 1u)
        PD_FILL_U(position, c->cfg.window_wise_output ? firstPos + (uint32_t)cs.window * shift - 1u : cs.suppStart)
        PD_FILL_U(end_position, c->cfg.window_wise_output ? 0u : cs.suppEnd)
        PD_FILL_U(near_integer_shifts, cs.nearIntegerShifts)
#undef PD_FILL_U
        for (uint32_t i = 0; i < n; ++i) d[i] = cells[emit[i]].logLR;
        pc->likelihood_ratio = mallocCopy(d);
        for (uint32_t i = 0; i < n; ++i) d[i] = cells[emit[i]].freq;
        pc->frequency = mallocCopy(d);
        for (uint32_t i = 0; i < n; ++i)
        {
            // sample-number filter (reference utils_popdel.h:707-718, genotype_deletion_popdel_call.h:723-724)
            const unsigned dataSamples = SG - cells[emit[i]].lowCovSamples;
            f[i] = (static_cast<double>(dataSamples) / SG >= c->cfg.min_sample_fraction) ? 0 : 4;
        }
        pc->filter = mallocCopy(f);
        pc->pl = hPl;
        pc->lad = hLad;
        pc->dad = hDad;
        pc->first_last = hFl;
        if (keep)
        {
            box->owner = c;
            box->keptId = c->keptId = c->nextKeptId++;
        }
        *out = pc;
    }
    catch (const HipError & e)
    {
        rc = fail(c, "pd_segment_collect: " + e.what);
    }
    catch (const std::exception & e)
    {
        rc = fail(c, std::string("pd_segment_collect: ") + e.what());
    }
    catch (...)
    {
        rc = fail(c, "pd_segment_collect: unknown exception");
    }
    if (rc != 0)
    {
        givePinned(pinned);                 // nobody owns the result block of a failed collect
        *out = nullptr;
        groupFail(c);                       // the other contexts of a local group must not wait for this one
    }
    if (getenv("POPDEL_HIP_MEMSTAT"))
    {
        size_t freeB = 0, totalB = 0;
        (void)hipMemGetInfo(&freeB, &totalB);
        std::fprintf(stderr, "[popdel_hip memory] HBM %.1f MB in %llu allocations (device reports %.1f MB in use), page-locked %.1f MB in %llu\n",
                     g_devBytes.load() / 1048576.0, (unsigned long long)g_devAllocs.load(), (totalB - freeB) / 1048576.0,
                     g_pinnedBytes.load() / 1048576.0, (unsigned long long)g_pinnedAllocs.load());
    }
    if (s2Begin)
        (void)hipEventDestroy(s2Begin);
    if (s2End)
        (void)hipEventDestroy(s2End);
    (void)hipEventDestroy(run.s1Begin);
    (void)hipEventDestroy(run.s1End);
    if (run.scanDone)
        (void)hipEventDestroy(run.scanDone);
    c->lastSet = run.set;
    if (s.ownedBySubmit)
    {
        pd_ctx * o = c->owner ? c->owner : c;
        std::lock_guard<std::mutex> guard(o->stagedLock);
        auto sit = o->staged.find(run.handle);
        if (sit != o->staged.end())
        {
            if (o->spare.size() < 2 + (o->lanes ? o->lanes->workers.size() : 0))
                o->spare.push_back(std::move(sit->second));
            o->staged.erase(sit);
        }
    }
    return rc;
}

void pd_calls_free(pd_calls * pc)
{
    if (!pc)
        return;
    std::free(pc->initial_length);
    std::free(pc->iterations);
    std::free(pc->deletion_length);
    std::free(pc->likelihood_ratio);
    std::free(pc->frequency);
    std::free(pc->window_position);
    std::free(pc->position);
    std::free(pc->end_position);
    std::free(pc->filter);
    std::free(pc->near_integer_shifts);
    CallsBox * box = reinterpret_cast<CallsBox *>(pc);       // pub is the first member
    givePinned(box->pinned);
    delete box;
}

int pd_stage1_last(pd_ctx * c, pd_stage1_view * out)
{
    if (!c || !out)
        return 1;
    if (c->lanes)
    {
        pd_ctx * lane = c->lanes->workers[c->lanes->lastLane]->ctx;
        const int rc = pd_stage1_last(lane, out);
        if (rc != 0)
            c->error = lane->error;
        return rc;
    }
    try
    {
        PD_HIP(hipSetDevice(c->device));
        Stage1Out & o = c->s1[c->lastSet];
        const uint32_t W = o.windows, S = c->nSamples;
        const size_t cells = static_cast<size_t>(W) * S;
        std::vector<int32_t> q(cells);
        std::vector<uint8_t> f(cells);
        PD_HIP(hipStreamSynchronize(c->stream));
        if (o.screened && !o.fullValid)
            return fail(c, "pd_stage1_last: this context screens the windows and takes exact Q3 values in the flagged ones only; call "
                           "pd_ctx_stage1_view(ctx, 1) before the segment is run (or set POPDEL_HIP_SCAN=full) to have every cell's value");
        if (cells)
        {
            PD_HIP(hipMemcpy(q.data(), o.screened ? o.fullQ3.p : o.q3.p, cells * sizeof(int32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(f.data(), o.screened ? o.fullFlags.p : o.flags.p, cells, hipMemcpyDeviceToHost));
        }
        if (o.screened && cells)
        {
            // The view doubles as the check of the screen against the full scan, cell by cell: the flags agree everywhere; every
            // window in which the full scan saw a Q3 above the smallest threshold was flagged by the screen; and in every window
            // the screen flagged the exact pass left the full scan's Q3 for every sample. (In a sharded cohort the exact pass also
            // covers the windows other contexts flagged; the check looks at this context's own flags.)
            std::vector<int32_t> q2(cells);
            std::vector<uint8_t> f2(cells);
            std::vector<int> wf(W), wfFull(W);
            PD_HIP(hipMemcpy(q2.data(), o.q3.p, cells * sizeof(int32_t), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(f2.data(), o.flags.p, cells, hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(wf.data(), o.winFlag.p, W * sizeof(int), hipMemcpyDeviceToHost));
            PD_HIP(hipMemcpy(wfFull.data(), o.fullWinFlag.p, W * sizeof(int), hipMemcpyDeviceToHost));
            if (f != f2)
                return fail(c, "pd_stage1_last: the screen's cell flags differ from the full scan's");
            for (uint32_t w = 0; w < W; ++w)
            {
                if (wfFull[w] && !wf[w])
                    return fail(c, "pd_stage1_last: window " + std::to_string(w) + " holds a Q3 above the threshold and was not flagged by the screen");
                if (!wf[w])
                    continue;
                for (uint32_t smp = 0; smp < S; ++smp)
                    if (q2[static_cast<size_t>(smp) * W + w] != q[static_cast<size_t>(smp) * W + w])
                        return fail(c, "pd_stage1_last: window " + std::to_string(w) + ", sample " + std::to_string(smp) + ": the exact pass left Q3 " +
                                           std::to_string(q2[static_cast<size_t>(smp) * W + w]) + ", the full scan " +
                                           std::to_string(q[static_cast<size_t>(smp) * W + w]));
            }
        }
        // device layout is [sample][window]; the view is [window][sample]
        c->hQ3View.resize(cells);
        c->hFlagsView.resize(cells);
        for (uint32_t s = 0; s < S; ++s)
            for (uint32_t w = 0; w < W; ++w)
            {
                c->hQ3View[static_cast<size_t>(w) * S + s] = q[static_cast<size_t>(s) * W + w];
                c->hFlagsView[static_cast<size_t>(w) * S + s] = f[static_cast<size_t>(s) * W + w];
            }
        out->n_windows = W;
        out->n_samples = S;
        out->q3 = c->hQ3View.data();
        out->flags = c->hFlagsView.data();
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_stage1_last: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, std::string("pd_stage1_last: ") + e.what());
    }
    catch (...)
    {
        return fail(c, "pd_stage1_last: unknown exception");
    }
    return 0;
}

int pd_log_table_view(pd_ctx * c, uint32_t rg, double * value, double * logValue, double * log10Value, uint32_t * n)
{
    if (!c || rg >= c->nRg || !value || !logValue || !log10Value || !n)
        return fail(c, "pd_log_table_view: invalid argument");
    try
    {
        PD_HIP(hipSetDevice(c->device));
        pd::RgConst rc;
        PD_HIP(hipMemcpy(&rc, c->dRg.p + rg, sizeof(rc), hipMemcpyDeviceToHost));
        std::vector<double4> tab(static_cast<size_t>(rc.histLen));
        if (!tab.empty())
            PD_HIP(hipMemcpy(tab.data(), c->dTab.p + rc.histBase, tab.size() * sizeof(double4), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tab.size(); ++i)
        {
            value[i] = tab[i].x;
            logValue[i] = tab[i].y;
            log10Value[i] = tab[i].z;
        }
        *n = static_cast<uint32_t>(tab.size());
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_log_table_view: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, std::string("pd_log_table_view: ") + e.what());
    }
    catch (...)
    {
        return fail(c, "pd_log_table_view: unknown exception");
    }
    return 0;
}

int pd_ctx_stage1_view(pd_ctx * c, int on)
{
    if (!c)
        return 1;
    c->stage1View = on != 0;
    if (c->lanes)
        for (auto & w : c->lanes->workers)
            w->ctx->stage1View = on != 0;
    return 0;
}

int pd_ctx_keep_results_on_device(pd_ctx * c, int on)
{
    if (!c)
        return 1;
    c->keepOnDevice = on != 0;
    if (c->lanes)
        for (auto & w : c->lanes->workers)
            w->ctx->keepOnDevice = on != 0;
    return 0;
}

namespace {
struct MergedBox
{
    pd_merged pub;
    PinnedBlock pinned;
};
}  // namespace

// Do the per-sample columns of `calls` still sit in this context's HBM? (Identity by an id the collect handed out, not by
// address: a later pd_calls may live at a recycled address.)
static bool columnsResident(const pd_ctx * c, const pd_calls * calls)
{
    const CallsBox * box = reinterpret_cast<const CallsBox *>(calls);     // pub is the first member
    return box->owner == c && box->keptId != 0 && box->keptId == c->keptId;
}

int pd_calls_merge(pd_ctx * c, const pd_calls * calls, uint32_t n_groups, const uint32_t * member_offsets, const uint32_t * member_index,
                   const uint32_t * head_index, pd_merged ** out)
{
    if (!c || !calls || !out || (n_groups != 0 && (!member_offsets || !member_index || !head_index)))
        return fail(c, "pd_calls_merge: invalid argument");
    *out = nullptr;
    if (c->lanes)
    {
        // the columns sit in the HBM of the lane that ran the segment
        pd_ctx * lane = reinterpret_cast<const CallsBox *>(calls)->owner;
        if (!lane || lane->owner != c)
            return fail(c, "pd_calls_merge: the per-sample arrays of these calls are not (or no longer) on the device");
        const int rc = pd_calls_merge(lane, calls, n_groups, member_offsets, member_index, head_index, out);
        if (rc != 0)
            c->error = lane->error;
        return rc;
    }
    if (!columnsResident(c, calls))
        return fail(c, "pd_calls_merge: the per-sample arrays of these calls are not (or no longer) on the device");
    PinnedBlock pinned;
    int rc = 0;
    try
    {
        PD_HIP(hipSetDevice(c->device));
        const uint32_t S = calls->n_samples, n = calls->n_calls;
        const uint32_t nMembers = n_groups ? member_offsets[n_groups] : 0;
        for (uint32_t m = 0; m < nMembers; ++m)
            if (member_index[m] >= n)
                return fail(c, "pd_calls_merge: member index out of range");
        for (uint32_t g = 0; g < n_groups; ++g)
            if (head_index[g] >= n || member_offsets[g + 1] <= member_offsets[g])
                return fail(c, "pd_calls_merge: malformed group (every group needs at least one member)");
        const size_t perOut = static_cast<size_t>(n_groups) * S;
        std::unique_ptr<MergedBox> box(new MergedBox());      // (released to the caller on success only)
        std::memset(&box->pub, 0, sizeof(pd_merged));
        box->pub.n_groups = n_groups;
        box->pub.n_samples = S;
        if (n_groups != 0)
        {
            // indices up, kernel, merged columns down
            std::vector<uint32_t> idx;
            idx.insert(idx.end(), member_offsets, member_offsets + n_groups + 1);
            idx.insert(idx.end(), head_index, head_index + n_groups);
            idx.insert(idx.end(), member_index, member_index + nMembers);
            c->dMergeIdx.upload(idx, c->stream);
            c->dMergeOut.ensure(perOut * 13);
            c->dMergeCounters.ensure(n_groups + 1);
            PD_HIP(hipMemsetAsync(c->dMergeCounters.p, 0, (n_groups + 1) * sizeof(unsigned), c->stream));
            const size_t per = static_cast<size_t>(n) * S;
            pd::MergeIn in;
            in.pl = c->dOutPl.p;
            in.lad = in.pl + per * 3;
            in.dad = in.lad + per * 3;
            in.fl = in.dad + per * 5;
            in.memberOffsets = c->dMergeIdx.p;
            in.headIndex = c->dMergeIdx.p + n_groups + 1;
            in.memberIndex = in.headIndex + n_groups;
            in.nSamples = S;
            pd::MergeOut mo;
            mo.pl = c->dMergeOut.p;
            mo.lad = mo.pl + perOut * 3;
            mo.dad = mo.lad + perOut * 3;
            mo.fl = mo.dad + perOut * 5;
            mo.alleles = c->dMergeCounters.p;
            mo.overflow = c->dMergeCounters.p + n_groups;
            if (!pd::launchMergeGroups(in, mo, n_groups, c->stream))
                throw HipError{"the merge kernel cannot get its LDS"};
            PD_HIP(hipGetLastError());
            pinned = takePinned((perOut * 13 + n_groups + 1) * sizeof(uint32_t));
            uint32_t * h = static_cast<uint32_t *>(pinned.p);
            PD_HIP(hipMemcpyAsync(h, c->dMergeOut.p, perOut * 13 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            PD_HIP(hipMemcpyAsync(h + perOut * 13, c->dMergeCounters.p, (n_groups + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            PD_HIP(hipStreamSynchronize(c->stream));
            if (h[perOut * 13 + n_groups] != 0)
            {
                givePinned(pinned);
                c->error = "pd_calls_merge: a LAD/DAD value beyond the median bins; merge this segment on the host";
                return 2;
            }
            box->pub.pl = h;
            box->pub.lad = h + perOut * 3;
            box->pub.dad = h + perOut * 6;
            box->pub.first_last = h + perOut * 11;
            box->pub.alleles = h + perOut * 13;
        }
        box->pinned = pinned;
        *out = &box.release()->pub;
    }
    catch (const HipError & e)
    {
        rc = fail(c, "pd_calls_merge: " + e.what);
    }
    catch (const std::exception & e)
    {
        rc = fail(c, std::string("pd_calls_merge: ") + e.what());
    }
    catch (...)
    {
        rc = fail(c, "pd_calls_merge: unknown exception");
    }
    if (rc != 0)
        givePinned(pinned);
    return rc;
}

void pd_merged_free(pd_merged * m)
{
    if (!m)
        return;
    MergedBox * box = reinterpret_cast<MergedBox *>(m);           // pub is the first member
    givePinned(box->pinned);
    delete box;
}

int pd_calls_fetch_columns(pd_ctx * c, pd_calls * calls)
{
    if (!c || !calls)
        return fail(c, "pd_calls_fetch_columns: invalid argument");
    if (calls->pl)
        return 0;                                   // they are there already
    if (c->lanes)
    {
        pd_ctx * lane = reinterpret_cast<const CallsBox *>(calls)->owner;
        if (!lane || lane->owner != c)
            return fail(c, "pd_calls_fetch_columns: the per-sample arrays of these calls are not (or no longer) on the device");
        const int rc = pd_calls_fetch_columns(lane, calls);
        if (rc != 0)
            c->error = lane->error;
        return rc;
    }
    if (!columnsResident(c, calls))
        return fail(c, "pd_calls_fetch_columns: the per-sample arrays of these calls are not (or no longer) on the device");
    try
    {
        PD_HIP(hipSetDevice(c->device));
        const size_t per = static_cast<size_t>(calls->n_calls) * calls->n_samples;
        CallsBox * box = reinterpret_cast<CallsBox *>(calls);     // pub is the first member
        box->pinned = takePinned(per * 13 * sizeof(uint32_t));
        uint32_t * h = static_cast<uint32_t *>(box->pinned.p);
        PD_HIP(hipMemcpyAsync(h, c->dOutPl.p, per * 13 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        PD_HIP(hipStreamSynchronize(c->stream));
        calls->pl = h;
        calls->lad = h + per * 3;
        calls->dad = h + per * 6;
        calls->first_last = h + per * 11;
    }
    catch (const HipError & e)
    {
        return fail(c, "pd_calls_fetch_columns: " + e.what);
    }
    catch (const std::exception & e)
    {
        return fail(c, std::string("pd_calls_fetch_columns: ") + e.what());
    }
    catch (...)
    {
        return fail(c, "pd_calls_fetch_columns: unknown exception");
    }
    return 0;
}

int pd_last_stage_timing(pd_ctx * c, pd_stage_timing * out)
{
    if (!c || !out)
        return 1;
    *out = c->stageTiming;
    return 0;
}

// pageable -> page-locked on the library's host threads, the way the packing pass of pd_segment_stage() touches memory
double pd_host_copy_gbs(uint64_t bytes)
{
    if (bytes < (1u << 20))
        return 0;
    std::vector<uint8_t> src(bytes);
    void * dst = pd::pinnedAlloc(bytes);
    if (!dst)
        return 0;
    const size_t pieces = 256;
    auto pass = [&](bool write) {
        pdh::parallelFor(pieces, [&](size_t j) {
            const size_t lo = bytes * j / pieces, hi = bytes * (j + 1) / pieces;
            if (write)
                std::memset(src.data() + lo, (int)(j & 0xff), hi - lo);           // (first touch by the threads that read it later)
            else
                std::memcpy(static_cast<uint8_t *>(dst) + lo, src.data() + lo, hi - lo);
        });
    };
    pass(true);
    pass(false);                                                                   // (warm: page tables of the destination)
    const auto t0 = std::chrono::steady_clock::now();
    pass(false);
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pd::pinnedFree(dst);
    return s > 0 ? static_cast<double>(bytes) / s / 1e9 : 0;
}

int pd_last_timing(pd_ctx * c, pd_timing * out)
{
    if (!c || !out)
        return 1;
    *out = c->timing;
    return 0;
}

const char * pd_last_error(const pd_ctx * c) { return c ? c->error.c_str() : g_createError.c_str(); }

// POPDEL_HIP_BLOCKING_SYNC=1: host threads that wait for the device sleep in the driver instead of spinning in the runtime
// (hipDeviceScheduleBlockingSync). For callers short of CPU time, not of latency: `popdel call` sets it -- a sixth of its CPU
// time went into the runtime's wait loops, on a job whose rounds are bound by its CPU quota. Must come before the device's
// context exists; failures are ignored (the default schedule then stays).
static void applyDeviceSchedule()
{
    const char * e = getenv("POPDEL_HIP_BLOCKING_SYNC");
    if (e && e[0] == '1')
        (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    (void)hipGetLastError();
}

int pd_device_warmup(int device)
{
    const bool timed = getenv("POPDEL_HIP_MEMSTAT") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (hipSetDevice(device) != hipSuccess)
        return 1;
    applyDeviceSchedule();
    const double t1 = now();
    void * p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess)       // (the first allocation creates the device context)
        return 1;
    const double t2 = now();
    // ... and the first kernel launch loads the library's code object
    pd::ZeroRanges zero{};
    zero.add(p, 256);
    pd::launchZero(zero, nullptr);
    const bool ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    const double t3 = now();
    (void)hipFree(p);
    if (timed)
        std::fprintf(stderr, "[popdel_hip warm-up] runtime start + device %.3f s, first allocation %.3f s, first kernel (code object) %.3f s\n",
                     t1 - t0, t2 - t1, t3 - t2);
    return ok ? 0 : 1;
}

const char * pd_backend_name(void) { return "hip-gfx950"; }

int pd_abi_version(void) { return PD_ABI_VERSION; }

}  // extern "C"
