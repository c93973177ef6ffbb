// Device-side data layout and kernel launch interface of the MI355X joint-calling backend.
// (Private to libpopdel_hip.so; the public boundary is include/popdel_hip.h.)
#ifndef PD_DEVICE_H_
#define PD_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pd {

// ---- limits --------------------------------------------------------------------------------------------------
constexpr int kMaxCandidatesPerWindow = 32;    // candidate deletion lengths kept per window
constexpr int kMaxIterationsCap = 64;          // visited-table capacity per cell (params.iterations <= this)
constexpr int kStage1Block = 256;              // windows per workgroup in the stage-1 scan
constexpr int kStage1LdsPairs = 3072;          // pair chunks staged in LDS per stage-1 workgroup
constexpr int kSampleTile = 256;               // threads per workgroup in the stage-2 per-sample kernels
constexpr int kDenseCap = 96;                  // active pairs per (cell, sample) the dense stage-2 image holds
constexpr unsigned kDenseHighCov = 0xffffu;    // cellN marker: read group at or above maxLoad (skipped by the reference)
constexpr unsigned kTilesPerSweep = 40;         // sample tiles a per-sample kernel walks for ALL cells before it moves on (see stage2.hip)
constexpr int kSamplesPerBlock = 256;           // samples one such workgroup walks (8 half-waves x 32 samples; 64 and 128 were measured slower)

// x87 `expl` underflows to zero below ln(2^-16446); the reference's "likelihood too small" reset hangs off that.
#define PD_X87_EXP_UNDERFLOW (-11399.498531488860558)
#define PD_LOG_PSEUDO (-23.025850929940457)     // ln(1e-10)

// ---- per read group constants (HBM, read-mostly) ----------------------------------------------------------------
struct RgConst
{
    int      median;
    int      readLength;
    int      histOffset;
    int      histLen;
    int      lowerQ;
    int      upperQ;
    unsigned maxLoad;
    unsigned minInitDelLen;
    double   stddev;
    double   minProb;
    double   logMinProb;    // log(minProb), log10(minProb): filled on the device (buildLogTables)
    double   log10MinProb;
    uint64_t histBase;      // offset (in doubles) of this read group's table in the histogram pool
};

// A read pair chunk: active in window w iff it lies in the chunk range of w and off > w.
struct PairDO
{
    int32_t  dev;
    uint32_t off;
};

// ---- everything the kernels need to find the pairs of a (read group, window) --------------------------------------
struct SegmentView
{
    const PairDO *   pairs;        // [nChunks]  dev + off, read-group major, ordered by `on`
    const uint32_t * start;        // [nChunks]  startPos
    const uint64_t * rgBase;       // [nRg]      first chunk of the read group
    // Chunk index by SLOT: a chunk starting in window a >= 1 has slot a + K - 1; a chunk present from window 0 on
    // (carried over or new) has slot (its end window - 1), so carried-over pairs leave the candidate range exactly when
    // they end. Window w sees the slots [w, w + K): chunks [winStart[w], winStart[w + K]). Row length W + K.
    const uint32_t * winStart;     // [nRg * (W + K)] first chunk (relative to rgBase) whose slot >= x
    // stage-1 image of the chunks, 4 bytes each (round 5; 8 in rounds 1-4, each half duplicated for the packed arithmetic of the
    // full scan, which now duplicates on the fly): the deviation as int16 biased by 0x8000 in the low half, the end window in
    // the high half; only valid for read groups with rgPacked != 0
    const uint32_t * pairsPk;      // [chunks of the stage-1 image]
    const uint32_t * endCount;     // [nRg * (W + 1)] number of chunks of the read group that ended at or before w
    // The stage-1 image is cut with ITS OWN chunk life (maxLifePk <= maxLife) and has its own index: Q3 is an order
    // statistic of integers and does not depend on how the pairs are cut, so the scan takes the life that makes it
    // fastest (fewer slots per cell), while stage 2 keeps the chunk order its sums are formed in. Rows for the real
    // read groups and the scan read groups behind them.
    const uint64_t * rgBasePk;     // [nRg + nScanRg] first chunk of the stream in pairsPk
    const uint32_t * winStartPk;   // [(nRg + nScanRg) * (W + maxLifePk)]
    const uint32_t * endCountPk;   // [(nRg + nScanRg) * (W + 1)]
    uint32_t         maxLifePk;
    const uint8_t *  active8;      // [(nRg + nScanRg) * W] active chunks of the stream in window w = winStartPk[w + K] - endCountPk[w],
                                   // capped at 255 (255: look the two up): all the screen needs of the index per cell (round 5)
    const uint8_t *  rgPacked;     // [nRg + nScanRg] 1 if all deviations fit int16 and W fits 16 bits
    // Samples with several read groups get a "scan read group" appended behind the real ones: the chunks of all their
    // read groups as ONE stream (stage-1 image, winStart and endCount rows only), so that the scan kernel treats them
    // like single-read-group samples. scanRg[s] is the sample's own read group or that appended one. Where one of the
    // real read groups reaches its maxLoad (the reference then drops that read group's pairs) the 512-window block
    // is flagged in scanGeneric and goes to the fallback kernel.
    const uint32_t * scanRg;       // [nSamples]
    const uint8_t *  scanGeneric;  // [nScanRg][ceil(W / 512)], or null if there are no such samples
    uint32_t         nRealRg;
    uint32_t         nWindows;
    uint32_t         maxLife;      // K: a chunk lives at most K windows
    uint32_t         firstPos;
    uint32_t         shift;
};

struct CohortView
{
    const RgConst *  rg;
    const double *   hist;         // histogram pool
    const double4 *  tab;          // per pool entry {value, log(value), log10(value), -}: one 32-byte line per lookup
                                   // (logs computed once on the device with the functions the kernels would call)
    const double2 *  tab2;         // {value, log(value)} only: what the EM kernels look up, four entries per 64-byte
                                   // line (the 32-byte entries cost the data-likelihood kernel 4x the HBM traffic)
    const uint32_t * sampleRg;     // [nSamples + 1]
    uint32_t         nSamples;
    uint32_t         nRg;
    int              minThreshold; // min over read groups of minInitDelLen
};

// How the scan's launch shares the chip with the kernels of the context's other stream (stage 2 of the previous ticket).
struct ScanRoom
{
    enum Mode : unsigned
    {
        kFull = 0,           // one workgroup per (sample, 512-window block): fills every wave slot
        kLdsCap = 1,         // the same launch with unused dynamic LDS so that at most wgPerCu workgroups fit a CU
        kPersistent = 2      // wgPerCu x nCu workgroups that draw their blocks from `counter` (zeroed by the caller)
    };
    unsigned   mode;
    unsigned   wgPerCu;
    unsigned   nCu;
    unsigned * counter;
};

// Input of the clustering kernel: Q3 values and flags of the flagged windows of ALL samples of the cohort, one chunk
// per context (all-gathered when the cohort is sharded): int32 q3[nFlagged][sper], then uint8 flags[nFlagged][sper].
struct ClusterInput
{
    const uint8_t *  base;
    size_t           chunkBytes;
    unsigned         sper;         // samples per context, padded to whole tiles (padding carries "no value")
    unsigned         world;
    unsigned         nFlagged;     // flagged windows; with nFlaggedDev set: the capacity the layout is made for
    const unsigned * nFlaggedDev;  // null, or where the device finds their number (the host has not seen it yet)
    const unsigned * minInit;      // [read groups of the cohort] minInitDelLen, looked up BY SORTED RANK (quirk Q1)
};

// Up to six word-aligned ranges cleared by ONE kernel launch (every hipMemsetAsync is a launch of its own, ~6-10 us apart
// on the stream; a segment needs nine).
struct ZeroRanges
{
    uint32_t * p[6];
    size_t     words[6];
    unsigned   n;
    void add(void * ptr, size_t bytes)
    {
        p[n] = static_cast<uint32_t *>(ptr);
        words[n] = (bytes + 3) / 4;
        ++n;
    }
};

// Called by the launch sequences wherever samples of other contexts are needed: all-gathers `bytesPerRank` bytes per
// context inside the area starting at `area` (layout [world][bytesPerRank]); stream-ordered. No-op for world == 1.
struct Exchange
{
    virtual void allGather(void * area, size_t bytesPerRank) = 0;
    virtual ~Exchange() {}
};

// Where the per-tile sums of (cell c, cohort-wide tile t) live, and the record of read group 0's triple
// ({d0, d1, d2, kind}; kind 1 = high coverage) that rank `r` published for cell c.
struct Stage2Buffers;
__device__ __forceinline__ double * tileSlot(const Stage2Buffers & b, unsigned c, unsigned t);
__device__ __forceinline__ double * rg0Slot(const Stage2Buffers & b, unsigned c, unsigned r);

enum TileMode : unsigned
{
    kTileDense = 0,          // one lane per sample (every sample of the tile has one read group)
    kTileMulti = 1,          // one lane per read group + per-sample combine
    kTileGeneric = 2         // half-wave per sample with on-the-fly compaction (more than kDenseCap pairs somewhere)
};

// Static description of the tiles that contain multi-read-group samples.
struct MultiTiles
{
    const uint32_t * tile;       // [n] sample tile
    const uint32_t * rgLo;       // [n] first read group of the tile's samples
    const uint32_t * rgHi;       // [n] one past the last
    const uint32_t * waveBase;   // [n] first 64-read-group wave tile of the tile in the multi image
    const uint32_t * waveTile;   // [nWaves] multi tile of a wave tile
    const uint32_t * rgSample;   // [nRg] sample (local index) of a read group
    uint32_t         n, nWaves;
};

// ---- stage 2 per-cell state -------------------------------------------------------------------------------------
enum CellPhase : int
{
    kCellDead = 0,
    kCellIterating = 1,
    kCellConverged = 2,       // left the EM loop through the visited-table test
    kCellStopped = 3,         // left the EM loop otherwise (cap, len < minLen, freq == 0)
    kCellFinal = 4,           // passed the post-loop test, final pass runs
    kCellEmit = 5             // call is emitted
};

struct CellState
{
    int      window;
    int      initLen;
    int      phase;
    int      iterations;
    unsigned image;               // dense image of the cell's window, index within the batch
    unsigned imageFirst;          // cells[i].imageFirst / imageEnd: the cells of IMAGE i (rides along with the cells' upload)
    unsigned imageEnd;
    unsigned len, prevLen;
    double   freq, prevFreq;
    double   gt[3], prevGt[3];
    double   logLR, prevLogLR;
    int      restoreZeroShifts;   // "keep previous estimate": shifts go back to zero
    int      nVisited;
    int      visitedLen[kMaxIterationsCap];
    double   visitedFreq[kMaxIterationsCap];
    // reductions written by the per-sample kernels (fixed order: tile 0 .. nTiles-1)
    unsigned suppCapacity;        // upper bound of supporting pairs (a5 "total")
    unsigned suppCount;
    uint64_t suppBase;
    unsigned suppStart, suppEnd;
    unsigned lowCovSamples;
    unsigned nearIntegerShifts;   // reference-shift quotients within 1e-9 of an integer, all samples, all EM iterations (reported only)
    // distributed percentile selection over the support pools (0: starts, 1: ends): radix prefix and remaining rank
    unsigned selPrefix[2];
    unsigned selK[2];
    unsigned selLo[2];            // the selection runs on x - selLo (0, or the cohort-wide minimum of the pool: rangeSelect)
};

// Device buffers and parameters of the stage-2 kernels.
struct Stage2Buffers
{
    CellState * cells;
    unsigned    nCells;
    double *    dl;          // [nCells][S][3]  log-domain data likelihood triples
    int *       shifts;      // [nCells][nRg]   reference shifts
    // Exchange areas. The cohort may be sharded by samples over `world` contexts (one per GPU); every context owns
    // `tilesPerRank` consecutive 256-sample tiles. Whatever crosses samples is written into the context's own chunk of
    // an area laid out [world][chunk] and all-gathered between kernels; per-cell kernels then add the chunks in
    // rank/tile order, so results do not depend on the number of shards (world == 1: no exchange at all).
    // Area `xch`, one chunk per context (xchChunk doubles), always gathered as a whole (it is small, and one region means
    // that things which are ready at the same time travel in ONE exchange):
    //   [kXchHeaderDoubles]      header: word 0 = this context's guard (see `guard` below), rest unused
    //   [nCells][4] unsigned     integer accumulators (a5 count, a5 total, low-coverage samples, generic wave tiles)
    //   [nCells][2][4] unsigned  support pools of the context: {min, max, count, -} of pool 0 (starts) and 1 (ends)
    //   [tilesPerRank][nCells][4] double   per-tile sums
    //   [nCells][4] double       read group 0's record (published by rank 0)
    double *    xch;
    size_t      xchChunk;    // doubles per chunk
    unsigned *  selHist;     // [world][nCells][2][256] digit histograms of the support pools
    unsigned    nTiles;      // tiles of the whole cohort
    unsigned    tilesPerRank;
    unsigned    rank, world;
    unsigned    sharded;     // the cohort is sharded by samples (possibly over ONE context): the exchanged code path runs
    unsigned    nSamplesGlobal;
    // Dense image of the cells' active pairs, built once per cell by cellGatherKernel: for every 64-sample wave tile
    // the deviations (and start positions) transposed to [row j][lane = sample], so that the EM kernels run ONE LANE PER
    // SAMPLE with coalesced loads and no cross-lane work. Tiles the image cannot hold (a multi-read-group sample, more
    // than kDenseCap active pairs below maxLoad) are flagged in tileMode and go through the generic half-wave kernels.
    // The image depends on the window only: it is kept per IMAGE = distinct window of the batch (cells are ordered by
    // window; CellState::image, imageFirst, imageEnd), and the arrays below that say [nImages] are indexed by it.
    unsigned         nImages;
    int16_t *   cellDev;     // [nImages][ownTiles*4][kDenseCap][64]; 16 bits: a pair outside that range sends its tile to
                             // the generic kernels (profiles written with the default size limits stay far inside)
    int16_t *   cellStart;   // same shape; pair starts as offsets from the cell's window position
    uint16_t *  cellN;       // [nImages][S] active pairs of the sample's read group, or kDenseHighCov
    uint8_t *   waveRows;    // [nImages][ownTiles*4] max cellN over the wave tile (rows in use)
    unsigned *  tileMode;    // [nImages][ownTiles] kTileDense / kTileMulti / kTileGeneric (raised with atomicMax)
    // Tiles that contain samples with several read groups ("multi tiles", a static property of the cohort) have a second
    // dense image with ONE LANE PER READ GROUP: rows of 64 read groups instead of 64 samples. Their kernels run the
    // per-pair loops per read group, park the per-read-group results in rgOut and then add them up per sample.
    MultiTiles  mt;
    int16_t *   mDev;        // [nImages][mt.nWaves][kDenseCap][64]
    int16_t *   mStart;      // same shape
    uint16_t *  mN;          // [nImages][nRg] active pairs of the read group, or kDenseHighCov (multi tiles only)
    uint8_t *   mWaveRows;   // [nImages][mt.nWaves]
    double *    rgOutD;      // [nCells][nRg][8]  per-read-group sums (log-likelihoods, log10-likelihoods)
    unsigned *  rgOutU;      // [nCells][nRg][12] per-read-group counts (LAD, DAD, first/last, pairs, flags)
    unsigned    nRg;
    unsigned    ownTiles;    // 256-sample tiles of this context
    // final-pass outputs, [nCells][S][k]
    uint32_t *  pl;
    uint32_t *  lad;
    uint32_t *  dad;
    uint32_t *  fl;
    unsigned *  suppFirst;   // support pools
    unsigned *  suppLast;
    // Launch without the host looking at the pool sizes first (one round trip less per segment): if `guard` is set, the
    // pools hold suppCap entries sized from the segment's average, the generic kernels are launched or not
    // (genericLaunched) by what earlier segments needed, and a sharded cohort selects its support percentiles on
    // x - min(pool) with TWO exchanged digit passes (rangeSelect). suppScanKernel checks the first two against the real
    // counts and publishes the verdict in the header of the context's xch chunk, which travels with the first exchange:
    // cellInitFinishKernel kills every cell if ANY context's check failed; suppRangeKernel does the same if a pool spans
    // 2^16 positions or more. Either raises *guard -- the host (every host of a sharded cohort, they see the same) then
    // repeats the batch the exact way.
    uint64_t    suppCap;
    unsigned    genericLaunched;
    unsigned    rangeSelect;
    unsigned *  guard;
    // parameters
    unsigned *  iterFlags;   // [maxIterations + 1]: 1 if any cell was still iterating after EM iteration i
    unsigned    minLen;
    unsigned    maxIterations;
    double      minLR;
    double      minSampleFraction;
    int         somatic;
    int         windowWise;
};

constexpr unsigned kXchHeaderDoubles = 8;

__device__ __forceinline__ unsigned * rankHeader(const Stage2Buffers & b, unsigned r)
{
    return reinterpret_cast<unsigned *>(b.xch + (size_t)r * b.xchChunk);
}
__device__ __forceinline__ unsigned * counterSlot(const Stage2Buffers & b, unsigned r, unsigned c)
{
    return reinterpret_cast<unsigned *>(b.xch + (size_t)r * b.xchChunk + kXchHeaderDoubles) + (size_t)c * 4;
}
__device__ __forceinline__ unsigned * poolSlot(const Stage2Buffers & b, unsigned r, unsigned c, unsigned pool)
{
    return reinterpret_cast<unsigned *>(b.xch + (size_t)r * b.xchChunk + kXchHeaderDoubles + (size_t)b.nCells * 2) +
           ((size_t)c * 2 + pool) * 4;
}
__device__ __forceinline__ double * tileSlot(const Stage2Buffers & b, unsigned c, unsigned t)
{
    const unsigned r = t / b.tilesPerRank, lt = t - r * b.tilesPerRank;
    return b.xch + (size_t)r * b.xchChunk + kXchHeaderDoubles + (size_t)b.nCells * 6 + ((size_t)lt * b.nCells + c) * 4;
}
__device__ __forceinline__ double * rg0Slot(const Stage2Buffers & b, unsigned c, unsigned r)
{
    return b.xch + (size_t)r * b.xchChunk + kXchHeaderDoubles + (size_t)b.nCells * 6 + ((size_t)b.tilesPerRank * b.nCells + c) * 4;
}
// doubles per context in the area
inline size_t xchChunkDoubles(unsigned tilesPerRank, unsigned nCells)
{
    return kXchHeaderDoubles + (size_t)nCells * 6 + ((size_t)tilesPerRank + 1) * nCells * 4;
}

}  // namespace pd
#endif
