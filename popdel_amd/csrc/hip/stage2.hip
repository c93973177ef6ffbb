// Stage 2 of the joint-calling hot path on gfx950: candidate deletion lengths per window (clustering of the per-sample
// Q3 values), then for every (window, candidate length) cell the EM over deletion length / allele frequency, the final
// genotyping pass and the likelihood-ratio test.
//
// Reference behaviour: genotype_deletion_popdel_call.h:58-87 (clustering), :93-133 (initial allele frequency),
// :179-253 / :255-337 (data likelihoods), :343-380 (genotype priors), :388-462 (deletion-length update),
// :467-485 (allele-frequency update), :490-508 (likelihood ratio), :514-529 (support percentiles), :556-728 (driver);
// utils_popdel.h:1511-1528 (phred-scaled likelihoods), :690-718 (no-data samples, sample-number filter).
//
// Numerics: the reference keeps the per-sample genotype triples as x87 `long double` exp(logL - max), which stay
// non-zero down to e^-11399. FP64 cannot hold that, so the triples live here in the LOG domain (d_k = logL_k - max,
// FP64) and every consumer is rewritten for that representation; the reference's "triple contains a zero" reset is
// reproduced by comparing d_k with the x87 underflow threshold.
//
// Work split: a gather pass compacts the active pairs of every (cell, sample) once into a dense, transposed 16-bit
// image; the EM and final passes then run one lane per sample (one lane per read group in tiles with multi-read-group
// samples) over the rows of that image. A half-wave-per-sample variant with on-the-fly compaction remains for tiles the
// image cannot hold. Sums over samples are reduced per 256-sample tile inside the workgroup and the per-tile partials
// are added in cohort tile order by one wave per cell (lanes fetch, the additions stay sequential), so results depend
// neither on scheduling nor on how the cohort is sharded over GPUs.
#include "pd_device.h"

namespace pd {

// Per-sample kernels: grid.x = sample tile (fast), grid.y = cell. (The transposed order -- same samples, different cells
// dispatched back to back -- was measured 23 % slower on MI355X: all XCDs then pull the same lines at the same time. An
// XCD-aware 1-D order -- XCD x takes the tiles t = x mod 8 and walks all cells of a tile back to back, so that the
// tile's lookup tables stay in that XCD's L2 -- was 13 % slower as well: concurrent workgroups then hammer the same few
// table lines; spreading the samples over the channels matters more here than L2 reuse. What does matter: grid.x a
// multiple of 8, so that tile t always lands on XCD t mod 8 -- see runStage2Main.)
// A large cohort is walked in sweeps of kTilesPerSweep tiles (10,240 samples): grid.x = tiles of a sweep, grid.y = (sweep,
// cell), so that all cells of one sweep are dispatched before the next sweep starts and the lookup tables in use (49 MB
// per sweep for the EM, 98 MB for the final pass) stay in the 256 MB Infinity Cache; walking all tiles per cell, 40,000
// samples (192 / 384 MB of tables) ran stage 2 at 1.88 ms per 10,000 samples instead of 1.26.
// (Both expect the kernel's Stage2Buffers parameter to be called b.)
#define PD_CELL_IDX (blockIdx.y % b.nCells)
#define PD_TILE_IDX ((blockIdx.y / b.nCells) * gridDim.x + blockIdx.x)

// per-tile sums of this context's own tile PD_TILE_IDX
#define PD_OWN_TILE_SLOT(b, c) tileSlot(b, c, (b).rank * (b).tilesPerRank + PD_TILE_IDX)

// The deviations of a batch of rows have all been loaded before the first lookup is issued: an explicit wait, with the
// values tied to it. Left to itself the compiler waits for deviation q just before row q's lookups, and since it cannot
// know whether the (conditional) lookups of the rows before were issued, it words that wait so that it also covers
// them -- the lookups of a batch would again queue behind each other.
#define PD_ALL_LOADS_DONE(arr)                                              \
    do                                                                      \
    {                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    \
        _Pragma("unroll") for (unsigned q_ = 0; q_ < kRowBatch; ++q_)       \
            asm volatile("" : "+v"((arr)[q_]));                            \
    } while (0)

#define PD_LN2 0.69314718055994530942
#define PD_LOG10_2 0.30102999566398119521


__device__ __forceinline__ double lookupI(const RgConst & rc, const double * hist, int deviation)
{
    int i = deviation + rc.median - rc.histOffset;
    if (i <= 0 || i + 1 >= rc.histLen)
        return rc.minProb;
    return hist[rc.histBase + (unsigned)i];
}

// Table position of a deviation, -1 for "outside the table" (min_prob).
__device__ __forceinline__ long long lookupPos(const RgConst & rc, int deviation)
{
    int i = deviation + rc.median - rc.histOffset;
    if (i <= 0 || i + 1 >= rc.histLen)
        return -1;
    return (long long)(rc.histBase + (unsigned)i);
}

// log / log10 of every histogram value, computed with the same device functions the kernels use on the fly.
__global__ void logTableKernel(const double * __restrict__ hist, double4 * __restrict__ tab, double2 * __restrict__ tab2, size_t n,
                               RgConst * __restrict__ rg, unsigned nRg)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const double lg = log(hist[i]);
        tab[i] = make_double4(hist[i], lg, log10(hist[i]), 0.0);
        tab2[i] = make_double2(hist[i], lg);
    }
    if (i < nRg)
    {
        rg[i].logMinProb = log(rg[i].minProb);
        rg[i].log10MinProb = log10(rg[i].minProb);
    }
}

void buildLogTables(const double * hist, double4 * tab, double2 * tab2, size_t n, RgConst * rg, unsigned nRg, hipStream_t st)
{
    const size_t m = n > nRg ? n : nRg;
    hipLaunchKernelGGL(logTableKernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, hist, tab, tab2, n, rg, nRg);
}

__device__ __forceinline__ void priors(double f, int somatic, double * g)
{
    const double pseudo = 0.0000000001;
    if (!somatic)
    {
        g[0] = fmax((1 - f) * (1 - f), pseudo);
        g[1] = fmax(2 * f * (1 - f), pseudo);
        g[2] = fmax(f * f, pseudo);
    }
    else if (f <= 0.4)
    {
        g[0] = fmax(1 - 2 * f + pseudo, pseudo);
        g[1] = fmax(2 * f - 2 * pseudo, pseudo);
        g[2] = pseudo;
    }
    else if (f < 0.75)
    {
        g[0] = pseudo;
        g[1] = 1.;
        g[2] = pseudo;
    }
    else
    {
        g[0] = pseudo;
        g[1] = pseudo;
        g[2] = 1.;
    }
}

// Iteration helper over the active pairs of (read group, window).
struct PairRange
{
    const PairDO *   p;
    const uint32_t * st;
    unsigned         begin, end;
};

__device__ __forceinline__ PairRange pairRange(const SegmentView & seg, unsigned rg, unsigned w)
{
    const unsigned K = seg.maxLife;
    const uint32_t * ws = seg.winStart + (size_t)rg * (seg.nWindows + K);
    PairRange r;
    r.p = seg.pairs + seg.rgBase[rg];
    r.st = seg.start + seg.rgBase[rg];
    r.begin = ws[w];
    r.end = ws[w + K];
    return r;
}

__device__ __forceinline__ unsigned countActive(const PairRange & r, unsigned w)
{
    unsigned c = 0;
    for (unsigned i = r.begin; i < r.end; ++i)
        c += (r.p[i].off > w) ? 1u : 0u;
    return c;
}

// ---- block reductions (256 threads, deterministic) -----------------------------------------------------------------
__device__ __forceinline__ double waveSum(double v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        v += __shfl_down(v, o, 64);
    return v;
}

__device__ __forceinline__ double blockSum(double v, double * scratch)
{
    v = waveSum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
        scratch[wv] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i)
            t += scratch[i];
    return t;     // valid in thread 0
}


// ---- half-wave helpers: 32 lanes work on the pairs of ONE (cell, sample); the two halves of a wave are independent --
__device__ __forceinline__ unsigned long long halfMask() { return 0xffffffffull << (threadIdx.x & 32u); }
__device__ __forceinline__ unsigned halfCount(bool pred) { return (unsigned)__popcll(__ballot(pred) & halfMask()); }
__device__ __forceinline__ double halfSum(double v)
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1)
        v += __shfl_xor(v, o, 32);
    return v;
}
__device__ __forceinline__ unsigned halfMinU(unsigned v)
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1)
        v = min(v, (unsigned)__shfl_xor((int)v, o, 32));
    return v;
}
__device__ __forceinline__ unsigned halfMaxU(unsigned v)
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1)
        v = max(v, (unsigned)__shfl_xor((int)v, o, 32));
    return v;
}
// Number of active pairs of a read group in window w, counted by the half-wave (coalesced 8-byte loads).
__device__ __forceinline__ unsigned halfCountActive(const PairRange & r, unsigned w, unsigned hl)
{
    unsigned n = 0;
    for (unsigned base = r.begin; base < r.end; base += 32u)
    {
        const unsigned i = base + hl;
        n += halfCount(i < r.end && r.p[i].off > w);
    }
    return n;
}

// Per-sample metadata of one 256-sample tile, fetched by 256 lanes in parallel (one dependent-load chain for the whole
// tile instead of one per sample) and kept in LDS while the half-waves walk the samples.
struct SampleMeta
{
    RgConst   rc;        // constants of the sample's first read group
    PairRange r;         // its pair range in the cell's window
    unsigned  rg0, rg1;  // read groups of the sample: [rg0, rg1)
};

__device__ __forceinline__ void loadTileMeta(const SegmentView & seg, const CohortView & co, unsigned w, SampleMeta * meta,
                                             unsigned tile)
{
    const unsigned s = tile * kSamplesPerBlock + threadIdx.x;
    if (threadIdx.x < (unsigned)kSamplesPerBlock && s < co.nSamples)
    {
        SampleMeta m;
        m.rg0 = co.sampleRg[s];
        m.rg1 = co.sampleRg[s + 1];
        m.rc = co.rg[m.rg0];
        m.r = pairRange(seg, m.rg0, w);
        meta[threadIdx.x] = m;
    }
    __syncthreads();
}

// Constants and pair range of read group `rg` of the sample described by `m` (cached for the first read group).
__device__ __forceinline__ void rgOfSample(const SampleMeta & m, const SegmentView & seg, const CohortView & co, unsigned w,
                                           unsigned rg, RgConst & rc, PairRange & r)
{
    if (rg == m.rg0)
    {
        rc = m.rc;
        r = m.r;
    }
    else
    {
        rc = co.rg[rg];
        r = pairRange(seg, rg, w);
    }
}

constexpr int kHalvesPerBlock = kSampleTile / 32;
constexpr unsigned kHalfBufCap = 128;         // compacted active pairs kept in LDS per half-wave

// Compacts the active pairs of (read group, window) into the half-wave's LDS buffer (ballot + rank), returns their
// number n. If n < maxLoad, calls f(dev, start) once per active pair with the pairs spread densely over the 32 lanes,
// so the FP64 work in f runs with ~n/32 lane utilisation instead of (#active / #candidates).
template <bool NeedStart, class F>
__device__ __forceinline__ unsigned halfForEachActive(const PairRange & r, unsigned w, unsigned hl, unsigned maxLoad, int * devBuf,
                                                      unsigned * stBuf, F f)
{
    const unsigned long long hm = halfMask();
    const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
    unsigned n = 0;
    for (unsigned base = r.begin; base < r.end; base += 32u)
    {
        const unsigned i = base + hl;
        bool act = false;
        PairDO p;
        p.dev = 0;
        p.off = 0;
        if (i < r.end)
        {
            p = r.p[i];
            act = p.off > w;
        }
        const unsigned long long bal = __ballot(act) & hm;
        if (act)
        {
            const unsigned slot = n + (unsigned)__popcll(bal & below);
            if (slot < kHalfBufCap)
            {
                devBuf[slot] = p.dev;
                if (NeedStart)
                    stBuf[slot] = r.st[i];
            }
        }
        n += (unsigned)__popcll(bal);
    }
    if (n >= maxLoad)
        return n;
    if (n <= kHalfBufCap)
    {
        for (unsigned j = hl; j < n; j += 32u)
            f(devBuf[j], NeedStart ? stBuf[j] : 0u);
    }
    else
    {
        // More active pairs than the buffer holds (n > 128: only with a maxLoad raised beyond that): the same compaction again,
        // the buffer handed to f whenever another 32 candidates might not fit. f relies on DENSE calls -- lane 0 of the half
        // takes part in every call that any lane takes part in (its class counts, its pool allocation); calling it straight
        // from the candidate loop, as rounds 1-4 did here, broke that: the support pools of such cells were filled at undefined
        // slots and the DAD / LAD counts of lane 0 missed pairs (found by tests/test_gpu_screen.py at 170x, round 5).
        unsigned fill = 0;
        for (unsigned base = r.begin; base < r.end; base += 32u)
        {
            const unsigned i = base + hl;
            bool act = false;
            PairDO p;
            p.dev = 0;
            p.off = 0;
            if (i < r.end)
            {
                p = r.p[i];
                act = p.off > w;
            }
            const unsigned long long bal = __ballot(act) & hm;
            if (act)
            {
                const unsigned slot = fill + (unsigned)__popcll(bal & below);
                devBuf[slot] = p.dev;
                if (NeedStart)
                    stBuf[slot] = r.st[i];
            }
            fill += (unsigned)__popcll(bal);
            if (fill + 32u > kHalfBufCap)
            {
                for (unsigned j = hl; j < fill; j += 32u)
                    f(devBuf[j], NeedStart ? stBuf[j] : 0u);
                fill = 0;
            }
        }
        for (unsigned j = hl; j < fill; j += 32u)
            f(devBuf[j], NeedStart ? stBuf[j] : 0u);
    }
    return n;
}
// Sum of one value per half-wave over the block, in half-wave order (valid in thread 0).
__device__ __forceinline__ double blockSumHalves(double v, double * scratch /* kHalvesPerBlock */)
{
    __syncthreads();
    if ((threadIdx.x & 31u) == 0)
        scratch[threadIdx.x >> 5] = v;
    __syncthreads();
    double t = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < kHalvesPerBlock; ++i)
            t += scratch[i];
    return t;
}

// =====================================================================================================================
// Candidate lengths: sort the valid Q3 values of a flagged window, chain-cluster them (gap < 50), keep cluster means
// above the threshold. The threshold table is indexed by POSITION IN THE SORTED ARRAY (reference :65,72,80).
// One workgroup per flagged window; bitonic sort in a global scratch row (L2 resident).
// =====================================================================================================================
// Emits one cluster (mean above threshold) into the candidate list of flagged window `fw`.
__device__ __forceinline__ void emitCluster(int sum, int n, unsigned threshold, int * candLen, unsigned fw, int & nc)
{
    if (!(sum / n > (int)threshold))
        return;
    const int v = sum / n;
    for (int c = 0; c < nc && c < kMaxCandidatesPerWindow; ++c)
        if (candLen[fw * kMaxCandidatesPerWindow + c] == v)
            return;                                   // std::set semantics
    if (nc < kMaxCandidatesPerWindow)
        candLen[fw * kMaxCandidatesPerWindow + nc] = v;
    ++nc;
}

__device__ __forceinline__ unsigned thresholdOfRanks(const ClusterInput & in, int uniformThreshold, unsigned i0, unsigned i1)
{
    if (uniformThreshold >= 0)
        return (unsigned)uniformThreshold;
    unsigned t = in.minInit[i0];
    for (unsigned i = i0 + 1; i <= i1; ++i)
        t = min(t, in.minInit[i]);
    return t;
}

constexpr int kClusterBins = 8192;

// Copies the Q3 values and flags of the flagged windows out of the [sample][window] stage-1 arrays into this context's
// chunk of the clustering input (window-major, samples padded to `sper` with "no value").
__global__ __launch_bounds__(256) void packFlaggedKernel(const int32_t * __restrict__ q3, const uint8_t * __restrict__ flags,
                                                         const int * __restrict__ flaggedWindows, unsigned W, unsigned nLocal,
                                                         ClusterInput in, unsigned rank)
{
    const unsigned l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= in.sper)
        return;
    const unsigned n = in.nFlaggedDev ? min(*in.nFlaggedDev, in.nFlagged) : in.nFlagged;
    uint8_t * chunk = const_cast<uint8_t *>(in.base) + (size_t)rank * in.chunkBytes;
    for (unsigned fw = blockIdx.y; fw < n; fw += gridDim.y)
    {
        int v = 0;
        uint8_t f = 0;
        if (l < nLocal)
        {
            const size_t cell = (size_t)l * W + (unsigned)flaggedWindows[fw];
            v = q3[cell];
            f = flags[cell];
        }
        reinterpret_cast<int32_t *>(chunk)[(size_t)fw * in.sper + l] = v;
        chunk[(size_t)in.nFlagged * in.sper * 4 + (size_t)fw * in.sper + l] = f;
    }
}

// Calls f(q3) for every sample of the cohort with a valid Q3 in flagged window fw (block-strided).
template <class F>
__device__ __forceinline__ void forEachValidQ3(const ClusterInput & in, unsigned fw, F f)
{
    for (unsigned r = 0; r < in.world; ++r)
    {
        const uint8_t * chunk = in.base + (size_t)r * in.chunkBytes;
        const int32_t * q = reinterpret_cast<const int32_t *>(chunk) + (size_t)fw * in.sper;
        const uint8_t * fl = chunk + (size_t)in.nFlagged * in.sper * 4 + (size_t)fw * in.sper;
        // eight independent (flag, value) loads in flight per lane: one at a time, the walk is a chain of L2 latencies
        // (measured: 75 us for the two walks over 10,000 samples)
        for (unsigned l0 = threadIdx.x; l0 < in.sper; l0 += 8u * blockDim.x)
        {
            uint8_t fv[8];
            int qv[8];
#pragma unroll
            for (unsigned u = 0; u < 8; ++u)
            {
                const unsigned l = l0 + u * blockDim.x;
                fv[u] = l < in.sper ? fl[l] : (uint8_t)0;
                qv[u] = l < in.sper ? q[l] : 0;
            }
#pragma unroll
            for (unsigned u = 0; u < 8; ++u)
                if (fv[u] & 2u)
                    f(qv[u]);
        }
    }
}

// Fast path: the Q3 values of a window span fewer than kClusterBins integers -> counting sort in LDS, clusters are read
// off the occupied bins in ascending order. Otherwise: bitonic sort of the values in a global scratch row.
struct ClusterShared
{
    int      hist[kClusterBins];
    unsigned occ[kClusterBins / 32];
    unsigned nValid;
    int      vMin, vMax;
};

// One flagged window by one workgroup; every return is workgroup-uniform.
__device__ __forceinline__ void clusterWindow(const ClusterInput & in, int uniformThreshold, int * __restrict__ scratch,
                                              unsigned rowLen, int * __restrict__ candLen, int * __restrict__ candCount,
                                              unsigned fw, ClusterShared & sh)
{
    int (&hist)[kClusterBins] = sh.hist;
    unsigned (&occ)[kClusterBins / 32] = sh.occ;
    unsigned & nValid = sh.nValid;
    int &vMin = sh.vMin, &vMax = sh.vMax;
    if (threadIdx.x == 0)
    {
        nValid = 0;
        vMin = 0x7fffffff;
        vMax = (int)0x80000000;
    }
    __syncthreads();
    {
        int lmin = 0x7fffffff, lmax = (int)0x80000000;
        unsigned cnt = 0;
        forEachValidQ3(in, fw, [&](int v) {
            lmin = min(lmin, v);
            lmax = max(lmax, v);
            ++cnt;
        });
        if (cnt)
        {
            atomicMin(&vMin, lmin);
            atomicMax(&vMax, lmax);
            atomicAdd(&nValid, cnt);
        }
    }
    __syncthreads();
    const unsigned m = nValid;
    if (m == 0)
    {
        if (threadIdx.x == 0)
            candCount[fw] = 0;
        return;
    }
    const long long range = (long long)vMax - (long long)vMin + 1;
    if (range <= kClusterBins)
    {
        const int base = vMin;
        for (int i = threadIdx.x; i < (int)range; i += blockDim.x)
            hist[i] = 0;
        __syncthreads();
        forEachValidQ3(in, fw, [&](int v) { atomicAdd(&hist[v - base], 1); });
        __syncthreads();
        // occupancy bitmap, built by all lanes, so that the single lane below only visits occupied bins
        const int nWords = ((int)range + 31) / 32;
        for (int wd = threadIdx.x; wd < nWords; wd += blockDim.x)
        {
            unsigned bits = 0;
            for (int k = 0; k < 32; ++k)
                if (wd * 32 + k < (int)range && hist[wd * 32 + k] != 0)
                    bits |= 1u << k;
            occ[wd] = bits;
        }
        __syncthreads();
        if (threadIdx.x == 0)
        {
            int nc = 0;
            int sum = 0, n = 0, prev = 0;
            unsigned rank = 0, i0 = 0;
            for (int wd = 0; wd < nWords; ++wd)
            {
                for (unsigned bits = occ[wd]; bits != 0; bits &= bits - 1u)
                {
                    const int bin = wd * 32 + (__ffs((int)bits) - 1);
                    const int c = hist[bin];
                    const int v = base + bin;
                    if (n != 0 && !(prev + 50 > v))
                    {
                        emitCluster(sum, n, thresholdOfRanks(in, uniformThreshold, i0, rank - 1), candLen, fw, nc);
                        sum = 0;
                        n = 0;
                        i0 = rank;
                    }
                    sum += v * c;
                    n += c;
                    rank += (unsigned)c;
                    prev = v;
                }
            }
            emitCluster(sum, n, thresholdOfRanks(in, uniformThreshold, i0, rank - 1), candLen, fw, nc);
            candCount[fw] = nc;
        }
        return;
    }
    // ---- wide value range: full sort ----------------------------------------------------------------------------
    int * row = scratch + (size_t)fw * rowLen;
    if (threadIdx.x == 0)
        nValid = 0;
    __syncthreads();
    forEachValidQ3(in, fw, [&](int v) { row[atomicAdd(&nValid, 1u)] = v; });
    __syncthreads();
    unsigned P = 1;
    while (P < m)
        P <<= 1;
    for (unsigned i = m + threadIdx.x; i < P; i += blockDim.x)
        row[i] = 0x7fffffff;
    __syncthreads();
    for (unsigned k = 2; k <= P; k <<= 1)
    {
        for (unsigned j = k >> 1; j > 0; j >>= 1)
        {
            for (unsigned i = threadIdx.x; i < P; i += blockDim.x)
            {
                unsigned ixj = i ^ j;
                if (ixj > i)
                {
                    int a = row[i], b = row[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up)
                    {
                        row[i] = b;
                        row[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0)
    {
        int nc = 0;
        int sum = row[0];
        int n = 1;
        unsigned i0 = 0;
        for (unsigned i = 1; i < m; ++i)
        {
            if (row[i - 1] + 50 > row[i])
            {
                sum += row[i];
                ++n;
            }
            else
            {
                emitCluster(sum, n, thresholdOfRanks(in, uniformThreshold, i0, i - 1), candLen, fw, nc);
                sum = row[i];
                n = 1;
                i0 = i;
            }
        }
        emitCluster(sum, n, thresholdOfRanks(in, uniformThreshold, i0, m - 1), candLen, fw, nc);
        candCount[fw] = nc;
    }
}

// The workgroups take the flagged windows in turn (their number may only be known on the device: ClusterInput).
__global__ __launch_bounds__(1024) void clusterKernel(ClusterInput in, int uniformThreshold, int * __restrict__ scratch,
                                                     unsigned rowLen, int * __restrict__ candLen, int * __restrict__ candCount)
{
    __shared__ ClusterShared sh;
    const unsigned n = in.nFlaggedDev ? min(*in.nFlaggedDev, in.nFlagged) : in.nFlagged;
    for (unsigned fw = blockIdx.x; fw < n; fw += gridDim.x)
    {
        clusterWindow(in, uniformThreshold, scratch, rowLen, candLen, candCount, fw, sh);
        __syncthreads();
    }
}

// The flagged windows of a segment in ascending order (one workgroup): hdr[0] = their number (also if the list, which
// holds `cap`, is too short for them).
// winFlag: `world` rows `rowStride` ints apart (the contexts' window flags after their exchange; one row if unsharded).
__global__ __launch_bounds__(1024) void compactFlaggedKernel(const int * __restrict__ winFlag, unsigned W, unsigned world,
                                                            size_t rowStride, unsigned cap, int * __restrict__ flagged,
                                                            unsigned * __restrict__ hdr)
{
    __shared__ unsigned perWave[16];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned carry = 0;
    for (unsigned base = 0; base < W; base += 1024u)
    {
        const unsigned w = base + threadIdx.x;
        bool v = false;
        if (w < W)
            for (unsigned r = 0; r < world; ++r)
                v = v || winFlag[r * rowStride + w] != 0;
        const unsigned long long bal = __ballot(v);
        if (lane == 0)
            perWave[wave] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = 0, all = 0;
        for (unsigned k = 0; k < 16u; ++k)
        {
            before += k < wave ? perWave[k] : 0u;
            all += perWave[k];
        }
        const unsigned at = carry + before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
        if (v && at < cap)
            flagged[at] = (int)w;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        hdr[0] = carry;
}

// The (window, candidate length) cells in the order the host would list them (flagged window, then candidate):
// hdr[1] = their number, hdr[2] = 1 if a window has more candidates than its list holds, hdr[4 + 2 i] / hdr[5 + 2 i] =
// window and length of cell i < listCap.
__global__ __launch_bounds__(256) void listCellsKernel(const int * __restrict__ flagged, const int * __restrict__ candLen,
                                                       const int * __restrict__ candCount, unsigned cap, unsigned listCap,
                                                       unsigned * __restrict__ hdr)
{
    __shared__ unsigned sh[256];
    __shared__ unsigned tooMany;
    const unsigned n = min(hdr[0], cap);
    if (threadIdx.x == 0)
        tooMany = 0;
    unsigned carry = 0;
    for (unsigned base = 0; base < n; base += 256u)
    {
        const unsigned f = base + threadIdx.x;
        const int cnt = f < n ? candCount[f] : 0;
        const unsigned v = (unsigned)min(cnt, kMaxCandidatesPerWindow);
        sh[threadIdx.x] = v;
        __syncthreads();
        if (cnt > kMaxCandidatesPerWindow)
            tooMany = 1u;
        for (unsigned o = 1; o < 256u; o <<= 1)
        {
            const unsigned t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const unsigned first = carry + sh[threadIdx.x] - v;
        for (unsigned k = 0; k < v; ++k)
            if (first + k < listCap)
            {
                hdr[4 + 2 * (first + k)] = (unsigned)flagged[f];
                hdr[5 + 2 * (first + k)] = (unsigned)candLen[f * kMaxCandidatesPerWindow + k];
            }
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        hdr[1] = carry;
        hdr[2] = tooMany;
    }
}

// =====================================================================================================================
// Cell set-up: first allele-frequency estimate (integer counts, reference :93-133), low-coverage sample count, and
// the dense image of the active pairs (see Stage2Buffers::cellDev).
// The image depends on the window only, and a window usually carries several cells (candidate lengths): it is built once
// per IMAGE = distinct window of the batch (CellState::imageFirst / imageEnd), and only the counts that depend on the candidate
// length are taken per cell. A workgroup takes one (image, 64-sample wave tile): the half-waves compact the active pairs of
// their samples into LDS rows, then the whole workgroup writes the rows out transposed (coalesced).
// =====================================================================================================================
constexpr int kGatherThreads = 512;            // 16 half-waves x 4 samples: the per-sample work is a chain of dependent loads,
                                               // so more half-waves in flight per LDS footprint beat longer loops
constexpr unsigned kGatherCells = 2;           // cells of an image whose counts one walk over the pairs takes (2 is the rule)
__global__ __launch_bounds__(kGatherThreads) void cellGatherKernel(SegmentView seg, CohortView co, Stage2Buffers b,
                                                                   const uint8_t * __restrict__ flags)
{
    __shared__ unsigned acc[2 + kMaxCandidatesPerWindow];  // usable pairs, low-coverage samples, hits per cell of the image
    __shared__ SampleMeta shMeta[64];
    __shared__ short shDev[64][kDenseCap + 1];             // deviations: 16 bits in the dense image (else: generic kernels)
    __shared__ short shSt[64][kDenseCap + 1];              // start positions relative to the cell's window (see stBase)
    __shared__ unsigned shN[64];
    __shared__ unsigned shRows, shGeneric;
    const unsigned img = blockIdx.y, wt = blockIdx.x;              // one workgroup per (image, 64-sample wave tile)
    const unsigned cell0 = b.cells[img].imageFirst, nc = min(b.cells[img].imageEnd - cell0, (unsigned)kMaxCandidatesPerWindow);
    if (threadIdx.x < 2u + (unsigned)kMaxCandidatesPerWindow)
        acc[threadIdx.x] = 0;
    if (threadIdx.x == 0)
    {
        shGeneric = 0;
        shRows = 0;
    }
    const unsigned w = (unsigned)b.cells[cell0].window;
    const unsigned stBase = seg.firstPos + w * seg.shift;  // LDS keeps the pair starts as 16-bit offsets from here
    const unsigned hl = threadIdx.x & 31u, half = threadIdx.x >> 5;
    const unsigned long long hm = halfMask();
    const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
    if (threadIdx.x < 64u && wt * 64u + threadIdx.x < co.nSamples)
    {
        SampleMeta m;
        m.rg0 = co.sampleRg[wt * 64u + threadIdx.x];
        m.rg1 = co.sampleRg[wt * 64u + threadIdx.x + 1];
        m.rc = co.rg[m.rg0];
        m.r = pairRange(seg, m.rg0, w);
        shMeta[threadIdx.x] = m;
    }
    __syncthreads();
    unsigned total = 0, low = 0;
    constexpr unsigned kHalves = kGatherThreads / 32;
    // the cells of the image, kGatherCells per walk over the pairs (the first walk also fills the image)
    for (unsigned c0 = 0; c0 < nc; c0 += kGatherCells)
    {
        const bool fill = c0 == 0u;
        unsigned L[kGatherCells], count[kGatherCells];
#pragma unroll
        for (unsigned q = 0; q < kGatherCells; ++q)
        {
            L[q] = (unsigned)b.cells[cell0 + min(c0 + q, nc - 1u)].initLen;
            count[q] = 0;
        }
        for (unsigned k = 0; k < 64u / kHalves; ++k)
        {
            const unsigned inWt = half + k * kHalves;
            const unsigned s = wt * 64u + inWt;
            if (s >= co.nSamples)                              // uniform per half-wave
            {
                if (hl == 0 && fill)
                    shN[inWt] = 0;
                continue;
            }
            const SampleMeta & meta = shMeta[inWt];
            const bool multi = (meta.rg1 - meta.rg0) != 1u;        // its pairs go into the per-read-group image instead
            bool generic = false;
            bool farStart = false;                                 // a start further than 32 kb from the window: not in 16 bits
            unsigned nDense = 0;
            for (unsigned rg = meta.rg0; rg < meta.rg1; ++rg)
            {
                RgConst rc;
                PairRange r;
                rgOfSample(meta, seg, co, w, rg, rc, r);
                int windowBegin[kGatherCells], windowEnd[kGatherCells];
                unsigned hits[kGatherCells];
#pragma unroll
                for (unsigned q = 0; q < kGatherCells; ++q)
                {
                    windowBegin[q] = max((int)L[q] / 2, (int)floor((L[q] - 2 * rc.stddev) + 0.5));
                    windowEnd[q] = (int)(L[q] + 2 * rc.stddev);
                    hits[q] = 0;
                }
                unsigned n = 0;
                // two 32-chunk groups per round, deviation AND start loaded up front: one memory round trip per ~64 candidates
                // instead of a chain of four (the kernel is bound by exactly that latency)
                for (unsigned base = r.begin; base < r.end; base += 64u)
                {
                    PairDO p[2];
                    unsigned st[2];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                    {
                        const unsigned i = base + 32u * g + hl;
                        p[g].dev = 0;
                        p[g].off = 0;
                        st[g] = 0;
                        if (i < r.end)
                        {
                            p[g] = r.p[i];
                            st[g] = r.st[i];
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                    {
                        const bool act = p[g].off > w;                 // off == 0 for lanes past the end
                        const unsigned long long bal = __ballot(act) & hm;
                        if (act && !multi && fill)
                        {
                            const unsigned pos = n + (unsigned)__popcll(bal & below);
                            if (pos < (unsigned)kDenseCap)
                            {
                                const int rel = (int)(st[g] - stBase);
                                shDev[inWt][pos] = (short)p[g].dev;
                                shSt[inWt][pos] = (short)rel;
                                farStart = farStart || rel < -32768 || rel > 32767 || p[g].dev < -32768 || p[g].dev > 32767;
                            }
                        }
                        n += (unsigned)__popcll(bal);
#pragma unroll
                        for (unsigned q = 0; q < kGatherCells; ++q)
                            hits[q] += halfCount(act && p[g].dev > windowBegin[q] && p[g].dev < windowEnd[q]);
                    }
                }
                if (n >= rc.maxLoad)
                    nDense = kDenseHighCov;
                if (n == 0 || n >= rc.maxLoad)
                    continue;
                if (fill)
                    total += n;
#pragma unroll
                for (unsigned q = 0; q < kGatherCells; ++q)
                    count[q] += hits[q];
                nDense = n;
                if (n > (unsigned)kDenseCap)
                    generic = true;
            }
            if (!fill)
                continue;
            low += (flags[(size_t)s * seg.nWindows + w] & 1u) ? 1u : 0u;
            generic = generic || halfCount(farStart) != 0u;
            if (hl == 0)
            {
                const bool noRows = generic || multi;
                shN[inWt] = noRows ? 0u : nDense;
                b.cellN[(size_t)img * co.nSamples + s] = (uint16_t)(noRows ? 0u : nDense);
                if (generic)
                    atomicMax(&shGeneric, 2u);
                else if (multi)
                    atomicMax(&shGeneric, 1u);
                else if (nDense != kDenseHighCov)
                    atomicMax(&shRows, nDense);
            }
        }
        if (hl == 0)
        {
#pragma unroll
            for (unsigned q = 0; q < kGatherCells; ++q)
                if (c0 + q < nc && count[q])
                    atomicAdd(&acc[2u + c0 + q], count[q]);
        }
    }
    if (hl == 0)
    {
        if (total)
            atomicAdd(&acc[0], total);
        if (low)
            atomicAdd(&acc[1], low);
    }
    __syncthreads();
    // transposed write-out: row j holds the j-th active pair of each of the 64 samples
    const unsigned rows = shRows;
    const size_t wtGlobal = (size_t)img * (b.ownTiles * 4) + wt;
    int16_t * outD = b.cellDev + wtGlobal * (kDenseCap * 64);
    int16_t * outS = b.cellStart + wtGlobal * (kDenseCap * 64);
    for (unsigned idx = threadIdx.x; idx < rows * 64u; idx += blockDim.x)
    {
        const unsigned j = idx >> 6, lane = idx & 63u, nl = shN[lane];
        const bool has = nl != kDenseHighCov && j < nl;
        outD[idx] = has ? shDev[lane][j] : (short)0;
        outS[idx] = has ? shSt[lane][j] : (short)0;
    }
    // every cell of the image: its own hits, the image's usable pairs and low-coverage samples
    for (unsigned q = threadIdx.x; q < nc; q += blockDim.x)
    {
        unsigned * mine = counterSlot(b, b.rank, cell0 + q);
        if (acc[2u + q])
            atomicAdd(&mine[0], acc[2u + q]);
        if (acc[0])
            atomicAdd(&mine[1], acc[0]);
        if (acc[1])
            atomicAdd(&mine[2], acc[1]);
        if (shGeneric == 2u)
            atomicAdd(&mine[3], 1u);                       // tells the host that the generic kernels are needed at all
    }
    if (threadIdx.x == 0)
    {
        b.waveRows[wtGlobal] = (uint8_t)rows;
        if (shGeneric)
        {
            // the whole 256-sample tile leaves the lane-per-sample kernels: kTileMulti (1) or kTileGeneric (2)
            atomicMax(&b.tileMode[(size_t)img * b.ownTiles + wt / 4], shGeneric);
        }
    }
}

// Exclusive scan of the support capacities of this context's own pools (one workgroup, 256 cells per round), and the
// check of an optimistic launch (Stage2Buffers::guard): the verdict goes into the header of the context's xch chunk and
// travels with the first exchange.
__global__ __launch_bounds__(256) void suppScanKernel(Stage2Buffers b)
{
    __shared__ uint64_t sh[256];
    uint64_t carry = 0;
    for (unsigned base = 0; base < b.nCells; base += 256u)
    {
        const unsigned c = base + threadIdx.x;
        const uint64_t v = c < b.nCells ? counterSlot(b, b.rank, c)[1] : 0u;      // usable pairs of this context's samples
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 256u; o <<= 1)
        {
            const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (c < b.nCells)
        {
            b.cells[c].suppCapacity = (unsigned)v;
            b.cells[c].suppBase = carry + sh[threadIdx.x] - v;
        }
        carry += sh[255];
        __syncthreads();
    }
    if (b.guard == nullptr)
        return;
    // launched ahead of the host's look at the counts: do the pools hold, are the generic kernels there if needed?
    __shared__ unsigned bad;
    if (threadIdx.x == 0)
        bad = carry > b.suppCap ? 1u : 0u;
    __syncthreads();
    if (!b.genericLaunched)
        for (unsigned c = threadIdx.x; c < b.nCells; c += 256u)
            if (counterSlot(b, b.rank, c)[3] != 0u)
                atomicOr(&bad, 1u);
    __syncthreads();
    if (threadIdx.x == 0)
        rankHeader(b, b.rank)[0] = bad;
}

__device__ __forceinline__ void loopHead(CellState & cs, const Stage2Buffers & b);

// After the first exchange: first allele-frequency estimate from the integer counts of all contexts (reference :93-133),
// then the head of the first EM iteration. If the optimistic launch failed in ANY context, every cell dies here.
__global__ void cellInitFinishKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.nCells)
        return;
    CellState & cs = b.cells[c];
    unsigned count = 0, total = 0, low = 0, bad = 0;
    for (unsigned r = 0; r < b.world; ++r)
    {
        const unsigned * k = counterSlot(b, r, c);
        count += k[0];
        total += k[1];
        low += k[2];
        bad |= rankHeader(b, r)[0];
    }
    if (b.guard != nullptr && bad)
    {
        cs.phase = kCellDead;
        if (c == 0)
            *b.guard = 1u;
        return;
    }
    cs.lowCovSamples = low;
    cs.suppCount = 0;
    cs.selPrefix[0] = cs.selPrefix[1] = cs.selK[0] = cs.selK[1] = cs.selLo[0] = cs.selLo[1] = 0;
    double freq = (total == 0u) ? 0.0 : (double)count / total;
    cs.freq = freq;
    cs.len = (unsigned)cs.initLen;
    cs.iterations = 0;
    cs.nVisited = 0;
    cs.restoreZeroShifts = 0;
    if (freq == 0)
    {
        cs.phase = kCellDead;
        return;
    }
    priors(freq, b.somatic, cs.gt);
    cs.prevFreq = freq;
    cs.prevLen = cs.len;
    cs.prevGt[0] = cs.gt[0];
    cs.prevGt[1] = cs.gt[1];
    cs.prevGt[2] = cs.gt[2];
    cs.phase = kCellIterating;
    loopHead(cs, b);                        // head of the first iteration
}

// =====================================================================================================================
// Data likelihoods of one sample given (len, reference shifts), EM flavour (reference :179-253).
// mode 0: current estimate (cells in kCellIterating); mode 1: previous estimate with zero shifts (kCellConverged);
// mode 2: the evaluation ahead of the loop -- like 0, but the cell's priors do not exist yet and no term is formed.
// Also accumulates the per-sample term of the allele-frequency update with the cell's current priors.
// =====================================================================================================================
__device__ __forceinline__ void finishTriple(double * d, const double * logL, unsigned nUsed, bool allRefEqDel)
{
    const double mx = fmax(fmax(logL[0], logL[1]), logL[2]);
    d[0] = logL[0] - mx;
    d[1] = logL[1] - mx;
    d[2] = logL[2] - mx;
    bool reset = (d[0] < PD_X87_EXP_UNDERFLOW) || (d[1] < PD_X87_EXP_UNDERFLOW) || (d[2] < PD_X87_EXP_UNDERFLOW);
    if (allRefEqDel && nUsed > 0)
    {
        // ref == del for every pair: in x87 arithmetic the het term sits a hair above the other two
        // (log(2.0) is a double constant), so the three values are NOT equal there.
        d[0] = d[1] = d[2] = 0;
    }
    else if (d[0] == d[1] && d[0] == d[2])
    {
        reset = true;
    }
    if (reset)
    {
        d[0] = 0;
        d[1] = PD_LOG_PSEUDO;
        d[2] = PD_LOG_PSEUDO;
    }
}

__global__ __launch_bounds__(kSampleTile) void dataLikKernel(SegmentView seg, CohortView co, Stage2Buffers b, int mode)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[4];
    __shared__ double shLogL[kSamplesPerBlock][3];
    __shared__ unsigned shUsed[kSamplesPerBlock];          // bit 31: every pair had ref == del
    __shared__ int shDev[kHalvesPerBlock][kHalfBufCap];
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    CellState & cs = b.cells[cellIdx];
    const int wantPhase = mode == 1 ? kCellConverged : kCellIterating;
    if ((mode != 2 && cs.phase != wantPhase) || b.tileMode[(size_t)imgIdx * b.ownTiles + PD_TILE_IDX] != kTileGeneric)
        return;
    const unsigned w = (unsigned)cs.window;
    const int len = mode == 2 ? cs.initLen : (int)(mode == 1 ? cs.prevLen : cs.len);
    const unsigned hl = threadIdx.x & 31u, half = threadIdx.x >> 5;
    __shared__ SampleMeta shMeta[kSamplesPerBlock];
    loadTileMeta(seg, co, w, shMeta, PD_TILE_IDX);
    // phase 1: half-waves walk the pairs, lane = pair
    for (unsigned k = 0; k < (unsigned)(kSamplesPerBlock / kHalvesPerBlock); ++k)
    {
        const unsigned slot = half + k * kHalvesPerBlock;
        const unsigned s = PD_TILE_IDX * kSamplesPerBlock + slot;
        if (s >= co.nSamples)
            break;                                         // uniform per half-wave
        const SampleMeta & meta = shMeta[slot];
        double logL[3] = {0, 0, 0};
        unsigned nUsed = 0;
        bool allEq = true;
        for (unsigned rg = meta.rg0; rg < meta.rg1; ++rg)
        {
            RgConst rc;
            PairRange r;
            rgOfSample(meta, seg, co, w, rg, rc, r);
            const int refShift = (mode != 1) ? b.shifts[(size_t)cellIdx * co.nRg + rg] : 0;
            double g0 = 0, g1 = 0, g2 = 0;
            bool eq = true;
            const unsigned n = halfForEachActive<false>(r, w, hl, rc.maxLoad, shDev[half], nullptr, [&](int dv, unsigned) {
                const long long pr = lookupPos(rc, dv - refShift), pdl = lookupPos(rc, dv - len);
                double ref = rc.minProb, lref = rc.logMinProb, del = rc.minProb, ldel = rc.logMinProb;
                if (pr >= 0)
                {
                    const double2 t = co.tab2[pr];
                    ref = t.x;
                    lref = t.y;
                }
                if (pdl >= 0)
                {
                    const double2 t = co.tab2[pdl];
                    del = t.x;
                    ldel = t.y;
                }
                g0 += lref;
                g1 += log(ref + del) - PD_LN2;
                g2 += ldel;
                eq = eq && (ref == del);
            });
            if (n >= rc.maxLoad)
            {
                if (rg == 0 && b.rank == 0 && hl == 0)
                {
                    double * r0 = rg0Slot(b, cellIdx, 0);
                    r0[0] = r0[1] = r0[2] = 0;
                    r0[3] = 1;
                }
                continue;
            }
            const double rs0 = halfSum(g0), rs1 = halfSum(g1), rs2 = halfSum(g2);
            allEq = allEq && (halfCount(!eq) == 0u);
            nUsed += n;
            logL[0] += rs0;
            logL[1] += rs1;
            logL[2] += rs2;
            if (rg == 0 && b.rank == 0 && hl == 0)
            {
                const double m = fmax(fmax(rs0, rs1), rs2);
                double * r0 = rg0Slot(b, cellIdx, 0);
                r0[0] = rs0 - m;
                r0[1] = rs1 - m;
                r0[2] = rs2 - m;
                r0[3] = 0;
            }
        }
        if (hl == 0)
        {
            shLogL[slot][0] = logL[0];
            shLogL[slot][1] = logL[1];
            shLogL[slot][2] = logL[2];
            shUsed[slot] = nUsed | (allEq ? 0x80000000u : 0u);
        }
    }
    __syncthreads();
    // phase 2: one lane per sample finishes the triple and the allele-frequency term
    double term = 0;
    const unsigned s = PD_TILE_IDX * kSamplesPerBlock + threadIdx.x;
    if (threadIdx.x < (unsigned)kSamplesPerBlock && s < co.nSamples)
    {
        double d[3];
        const double logL[3] = {shLogL[threadIdx.x][0], shLogL[threadIdx.x][1], shLogL[threadIdx.x][2]};
        finishTriple(d, logL, shUsed[threadIdx.x] & 0x7fffffffu, (shUsed[threadIdx.x] >> 31) != 0u);
        double * out = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
        out[0] = d[0];
        out[1] = d[1];
        out[2] = d[2];
        if (mode == 0)
        {
            const double p0 = exp(d[0]) * cs.gt[0];
            const double p1 = exp(d[1]) * cs.gt[1];
            const double p2 = exp(d[2]) * cs.gt[2];
            const double pAll = p0 + p1 + p2;
            term = (p1 + 2 * p2) / pAll;
        }
    }
    if (mode == 0)
    {
        const double t = blockSum(term, red);
        if (threadIdx.x == 0)
            PD_OWN_TILE_SLOT(b, cellIdx)[2] = t;
    }
}

// The same on the dense image: ONE LANE PER SAMPLE walks the rows of its wave tile (coalesced 4-byte loads, no
// cross-lane work). sum_j log(ref_j + del_j) is taken as log(prod_j mantissa_j) + ln2 * sum_j exponent_j: one log per
// sample instead of one per pair (mantissas lie in [0.5, 1), 64 of them cannot underflow).
struct LaneRg
{
    int      base;       // median - histOffset
    int      histLen;
    uint64_t histBase;
    unsigned maxLoad;
    double   minProb, logMinProb, stddev;
};
__device__ __forceinline__ LaneRg laneRg(const RgConst & rc)
{
    LaneRg l;
    l.base = rc.median - rc.histOffset;
    l.histLen = rc.histLen;
    l.histBase = rc.histBase;
    l.maxLoad = rc.maxLoad;
    l.minProb = rc.minProb;
    l.logMinProb = rc.logMinProb;
    l.stddev = rc.stddev;
    return l;
}
__device__ __forceinline__ long long lanePos(const LaneRg & l, int deviation)
{
    const int i = deviation + l.base;
    if (i <= 0 || i + 1 >= l.histLen)
        return -1;
    return (long long)(l.histBase + (unsigned)i);
}

// Table position for the dense loops (clamped to the table's first entry when outside). Rows beyond a lane's own count
// hold zeros or stale values and are masked by `use`. (Loading unconditionally and selecting afterwards was measured
// 1.7x slower: every lane reads its own sample's table, so each lookup costs the L1 a whole cache line.)
__device__ __forceinline__ uint64_t lanePosClamped(const LaneRg & l, int deviation, bool & inside)
{
    const int i = deviation + l.base;
    inside = !(i <= 0 || i + 1 >= l.histLen);
    return l.histBase + (inside ? (unsigned)i : 0u);
}

__global__ __launch_bounds__(kSampleTile) void dataLikDenseKernel(CohortView co, Stage2Buffers b, int mode)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[4];
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    const CellState & cs = b.cells[cellIdx];
    const int wantPhase = mode == 1 ? kCellConverged : kCellIterating;
    if ((mode != 2 && cs.phase != wantPhase) || b.tileMode[(size_t)imgIdx * b.ownTiles + PD_TILE_IDX] != 0)
        return;
    const int len = mode == 2 ? cs.initLen : (int)(mode == 1 ? cs.prevLen : cs.len);
    const unsigned s = PD_TILE_IDX * kSamplesPerBlock + threadIdx.x;
    const bool valid = s < co.nSamples;
    const size_t wtGlobal = (size_t)imgIdx * (b.ownTiles * 4) + PD_TILE_IDX * 4 + (threadIdx.x >> 6);
    const unsigned rows = b.waveRows[wtGlobal];
    const int16_t * col = b.cellDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
    double term = 0;
    unsigned rg = 0, n = 0;
    LaneRg l = {};
    int refShift = 0;
    if (valid)
    {
        rg = co.sampleRg[s];
        l = laneRg(co.rg[rg]);
        n = b.cellN[(size_t)imgIdx * co.nSamples + s];
        refShift = (mode != 1) ? b.shifts[(size_t)cellIdx * co.nRg + rg] : 0;
    }
    const bool highCov = n == kDenseHighCov;
    const unsigned nn = highCov ? 0u : n;
    double g0 = 0, g2 = 0, prod = 1.0;
    int esum = 0;
    bool eq = true;
    // The rows are taken kRowBatch at a time in three steps -- their deviations are loaded, then all their table lookups are
    // issued, then the arithmetic runs -- so that a wave has 2 x kRowBatch gathers in flight. Written row by row (round 2)
    // the compiler has to wait for a row's deviation before it can issue that row's lookups and, with one counter for all
    // outstanding loads, for the PREVIOUS row's lookups with it: a chain of two memory latencies per row, which is what
    // bounded these kernels (0.17 VALU issue, 72 % of the wave cycles parked).
    constexpr unsigned kRowBatch = 8;
    for (unsigned j0 = 0; j0 < rows; j0 += kRowBatch)
    {
        int dvs[kRowBatch];
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
            dvs[q] = col[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
        PD_ALL_LOADS_DONE(dvs);
        double2 trs[kRowBatch], tds[kRowBatch];
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
        {
            const bool use = j0 + q < nn;
            bool inR, inD;
            const uint64_t pr = lanePosClamped(l, dvs[q] - refShift, inR), pdl = lanePosClamped(l, dvs[q] - len, inD);
            // only in-range lookups touch memory: every lane reads its own sample's table, so each one costs a cache line
            trs[q] = make_double2(l.minProb, l.logMinProb);
            tds[q] = trs[q];
            if (use && inR)
                trs[q] = co.tab2[pr];
            if (use && inD)
                tds[q] = co.tab2[pdl];
        }
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
        {
            const bool use = j0 + q < nn;
            const double ref = trs[q].x, lref = trs[q].y, del = tds[q].x, ldel = tds[q].y;
            int e;
            const double m = frexp(ref + del, &e);
            g0 += use ? lref : 0.0;
            g2 += use ? ldel : 0.0;
            prod *= use ? m : 1.0;
            esum += use ? e : 0;
            eq = eq && (!use || ref == del);
        }
    }
    if (valid)
    {
        const double g1 = log(prod) + (double)(esum - (int)nn) * PD_LN2;
        if (rg == 0 && b.rank == 0)
        {
            double * r0 = rg0Slot(b, cellIdx, 0);
            if (highCov)
            {
                r0[0] = r0[1] = r0[2] = 0;
                r0[3] = 1;
            }
            else
            {
                const double m = fmax(fmax(g0, g1), g2);
                r0[0] = g0 - m;
                r0[1] = g1 - m;
                r0[2] = g2 - m;
                r0[3] = 0;
            }
        }
        double d[3];
        const double logL[3] = {g0, nn ? g1 : 0.0, g2};
        finishTriple(d, logL, nn, eq);
        double * out = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
        out[0] = d[0];
        out[1] = d[1];
        out[2] = d[2];
        if (mode == 0)
        {
            const double p0 = exp(d[0]) * cs.gt[0];
            const double p1 = exp(d[1]) * cs.gt[1];
            const double p2 = exp(d[2]) * cs.gt[2];
            const double pAll = p0 + p1 + p2;
            term = (p1 + 2 * p2) / pAll;
        }
    }
    if (mode == 0)
    {
        const double t = blockSum(term, red);
        if (threadIdx.x == 0)
            PD_OWN_TILE_SLOT(b, cellIdx)[2] = t;
    }
}

// The same for the two evaluations that run with ZERO reference shifts -- ahead of the loop (mode 2) and the converged
// cells' previous estimate (mode 1; the reference's prevShifts never leave zero, quirk Q3): the reference-allele lookups of
// a pair then do not depend on the cell, and a window carries two cells on average, so one workgroup takes an IMAGE and
// walks its cells two at a time: deviations and reference-allele lookups once per row, the deletion-allele lookups (mostly
// outside the table: no memory access) and the sums per cell. Per cell the arithmetic and its order are those of
// dataLikDenseKernel: bit-identical results.
__global__ __launch_bounds__(kSampleTile) void dataLikZeroShiftDenseKernel(CohortView co, Stage2Buffers b, int mode)
{
    const unsigned imgIdx = blockIdx.y % b.nImages;
    const unsigned tileIdx = (blockIdx.y / b.nImages) * gridDim.x + blockIdx.x;
    if (tileIdx >= b.ownTiles || b.tileMode[(size_t)imgIdx * b.ownTiles + tileIdx] != 0)
        return;
    const unsigned c0 = b.cells[imgIdx].imageFirst, c1 = b.cells[imgIdx].imageEnd;
    const unsigned s = tileIdx * kSamplesPerBlock + threadIdx.x;
    const bool valid = s < co.nSamples;
    const size_t wtGlobal = (size_t)imgIdx * (b.ownTiles * 4) + tileIdx * 4 + (threadIdx.x >> 6);
    const unsigned rows = b.waveRows[wtGlobal];
    const int16_t * col = b.cellDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
    unsigned rg = 0, n = 0;
    LaneRg l = {};
    if (valid)
    {
        rg = co.sampleRg[s];
        l = laneRg(co.rg[rg]);
        n = b.cellN[(size_t)imgIdx * co.nSamples + s];
    }
    const bool highCov = n == kDenseHighCov;
    const unsigned nn = highCov ? 0u : n;
    for (unsigned ca = c0; ca < c1; ca += 2u)
    {
        const unsigned cb = ca + 1u < c1 ? ca + 1u : ca;       // (an odd cell out is walked alone: its twin is itself, not stored twice)
        const CellState & csA = b.cells[ca];
        const CellState & csB = b.cells[cb];
        const bool doA = mode == 2 || csA.phase == kCellConverged;
        const bool doB = cb != ca && (mode == 2 || csB.phase == kCellConverged);
        if (!doA && !doB)                                      // workgroup-uniform
            continue;
        const int lenA = mode == 2 ? csA.initLen : (int)csA.prevLen;
        const int lenB = mode == 2 ? csB.initLen : (int)csB.prevLen;
        double g0 = 0, g2A = 0, g2B = 0, prodA = 1.0, prodB = 1.0;
        int esumA = 0, esumB = 0;
        bool eqA = true, eqB = true;
        constexpr unsigned kRowBatch = 4;
        for (unsigned j0 = 0; j0 < rows; j0 += kRowBatch)
        {
            int dvs[kRowBatch];
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
                dvs[q] = col[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
            PD_ALL_LOADS_DONE(dvs);
            double2 trs[kRowBatch], tdA[kRowBatch], tdB[kRowBatch];
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
            {
                const bool use = j0 + q < nn;
                bool inR, inA, inB;
                const uint64_t pr = lanePosClamped(l, dvs[q], inR);
                const uint64_t pa = lanePosClamped(l, dvs[q] - lenA, inA), pb = lanePosClamped(l, dvs[q] - lenB, inB);
                trs[q] = make_double2(l.minProb, l.logMinProb);
                tdA[q] = trs[q];
                tdB[q] = trs[q];
                if (use && inR)
                    trs[q] = co.tab2[pr];
                if (use && inA && doA)
                    tdA[q] = co.tab2[pa];
                if (use && inB && doB)
                    tdB[q] = co.tab2[pb];
            }
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
            {
                const bool use = j0 + q < nn;
                const double ref = trs[q].x;
                int eA, eB;
                const double mA = frexp(ref + tdA[q].x, &eA), mB = frexp(ref + tdB[q].x, &eB);
                g0 += use ? trs[q].y : 0.0;
                g2A += use ? tdA[q].y : 0.0;
                g2B += use ? tdB[q].y : 0.0;
                prodA *= use ? mA : 1.0;
                prodB *= use ? mB : 1.0;
                esumA += use ? eA : 0;
                esumB += use ? eB : 0;
                eqA = eqA && (!use || ref == tdA[q].x);
                eqB = eqB && (!use || ref == tdB[q].x);
            }
        }
        if (!valid)
            continue;
        for (int which = 0; which < 2; ++which)
        {
            if (!(which == 0 ? doA : doB))
                continue;
            const unsigned cellIdx = which == 0 ? ca : cb;
            const double g2 = which == 0 ? g2A : g2B, prod = which == 0 ? prodA : prodB;
            const int esum = which == 0 ? esumA : esumB;
            const bool eq = which == 0 ? eqA : eqB;
            const double g1 = log(prod) + (double)(esum - (int)nn) * PD_LN2;
            if (rg == 0 && b.rank == 0)
            {
                double * r0 = rg0Slot(b, cellIdx, 0);
                if (highCov)
                {
                    r0[0] = r0[1] = r0[2] = 0;
                    r0[3] = 1;
                }
                else
                {
                    const double m = fmax(fmax(g0, g1), g2);
                    r0[0] = g0 - m;
                    r0[1] = g1 - m;
                    r0[2] = g2 - m;
                    r0[3] = 0;
                }
            }
            double d[3];
            const double logL[3] = {g0, nn ? g1 : 0.0, g2};
            finishTriple(d, logL, nn, eq);
            double * out = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
            out[0] = d[0];
            out[1] = d[1];
            out[2] = d[2];
        }
    }
}

// =====================================================================================================================
// EM loop head (reference :598-604): loop condition, bookkeeping of the visited table and the previous estimate.
// =====================================================================================================================
__device__ __forceinline__ void loopHead(CellState & cs, const Stage2Buffers & b)
{
    if (cs.phase != kCellIterating)
        return;
    if (!(cs.len >= b.minLen && (unsigned)cs.iterations < b.maxIterations))
    {
        cs.phase = kCellStopped;
        return;
    }
    ++cs.iterations;
    int slot = -1;
    for (int i = 0; i < cs.nVisited; ++i)
        if (cs.visitedLen[i] == (int)cs.len)
            slot = i;
    if (slot < 0)
        slot = cs.nVisited++;
    cs.visitedLen[slot] = (int)cs.len;
    cs.visitedFreq[slot] = cs.freq;
    cs.prevLen = cs.len;
    cs.prevFreq = cs.freq;
    cs.prevGt[0] = cs.gt[0];
    cs.prevGt[1] = cs.gt[1];
    cs.prevGt[2] = cs.gt[2];
}

// New reference shift of one read group (genotype_deletion_popdel_call.h:447-451): the weighted mean deviation of its
// pairs, truncated the way x86 converts, zeroed beyond one standard deviation.
// PD_TEST_SNAP_SHIFTS is a compile-time switch of a TEST build of this file (never of the shipped library; the oracle has
// the matching knob): a quotient within 1e-9 of an integer is first moved onto it. Single-pair and equal-weight read
// groups put the exact quotient on an integer, and which side of it the rounded one lands on is decided by the last bit
// of the weights (Numerics); with the switch on both sides HIP and oracle must agree everywhere.
// `nearInteger` counts the quotients within 1e-9 of an integer -- the truncations whose outcome hangs off the last bit of
// the weights, i.e. where this FP64 evaluation and the reference's x87 one may disagree; the count is reported
// (pd_timing::n_near_integer_shifts), it changes nothing.
__device__ __forceinline__ int referenceShiftOf(double wSumRef, double sumRef, bool rgNan, double stddev, double & nearInteger)
{
    double q = wSumRef / sumRef;
    {
        const double r = rint(q);
        if (fabs(q - r) <= 1e-9 * fmax(1.0, fabs(r)))
        {
            nearInteger += 1.0;
#ifdef PD_TEST_SNAP_SHIFTS
            q = r;
#endif
        }
    }
    int newShift;
    if (rgNan || !(q > -2147483649.0 && q < 2147483648.0))
        newShift = (int)0x80000000;          // what x86 yields for NaN / out-of-range conversions
    else
        newShift = (int)q;
    if (newShift > stddev || newShift < -1 * stddev)
        newShift = 0;
    return newShift;
}

// =====================================================================================================================
// Deletion-length / reference-shift update (reference :388-462). The per-read-group weights come from RG 0's triple
// for EVERY read group (the reference never advances that iterator).
// =====================================================================================================================
__global__ __launch_bounds__(kSampleTile) void updateLenKernel(SegmentView seg, CohortView co, Stage2Buffers b)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[kHalvesPerBlock];
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    const CellState & cs = b.cells[cellIdx];
    if (cs.phase != kCellIterating || b.tileMode[(size_t)imgIdx * b.ownTiles + PD_TILE_IDX] != kTileGeneric)
        return;
    const unsigned w = (unsigned)cs.window;
    const int len = (int)cs.len;
    const unsigned hl = threadIdx.x & 31u, half = threadIdx.x >> 5;
    const double g0 = cs.gt[0], g1 = cs.gt[1], g2 = cs.gt[2];
    // weights of the reference model, shared by all read groups (always RG 0's triple)
    const double * rg0 = rg0Slot(b, cellIdx, 0);           // published by the context that owns read group 0
    bool rgNan = rg0[3] != 0;
    double ea0Rg = 0, ea1Rg = 0;
    if (!rgNan)
    {
        const double aSumRg = exp(rg0[0]) * g0 + exp(rg0[1]) * g1 + exp(rg0[2]) * g2;
        const double l0 = (rg0[0] < PD_X87_EXP_UNDERFLOW) ? -INFINITY : rg0[0];
        const double l1 = (rg0[1] < PD_X87_EXP_UNDERFLOW) ? -INFINITY : rg0[1];
        ea0Rg = exp(l0 + log(g0) - log(aSumRg));
        ea1Rg = exp(l1 + log(g1) - log(aSumRg));
        rgNan = isnan(ea0Rg) || isnan(ea1Rg);
    }
    __shared__ SampleMeta shMeta[kSamplesPerBlock];
    loadTileMeta(seg, co, w, shMeta, PD_TILE_IDX);
    // per-sample weights exp(a1), exp(a2): one lane per sample, shared through LDS
    __shared__ double shEa[kSamplesPerBlock][2];
    __shared__ int shDev[kHalvesPerBlock][kHalfBufCap];
    {
        const unsigned s = PD_TILE_IDX * kSamplesPerBlock + threadIdx.x;
        if (threadIdx.x < (unsigned)kSamplesPerBlock && s < co.nSamples)
        {
            const double * d = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
            const double aSum = exp(d[0]) * g0 + exp(d[1]) * g1 + exp(d[2]) * g2;
            shEa[threadIdx.x][0] = exp(d[1] + log(g1) - log(aSum));
            shEa[threadIdx.x][1] = exp(d[2] + log(g2) - log(aSum));
        }
    }
    __syncthreads();
    double sumDelAcc = 0, wSumDelAcc = 0, nearAcc = 0;
    for (unsigned k = 0; k < (unsigned)(kSamplesPerBlock / kHalvesPerBlock); ++k)
    {
        const unsigned slot = half + k * kHalvesPerBlock;
        const unsigned s = PD_TILE_IDX * kSamplesPerBlock + slot;
        if (s >= co.nSamples)
            break;
        const double ea1 = shEa[slot][0];
        const double ea2 = shEa[slot][1];
        const SampleMeta & meta = shMeta[slot];
        for (unsigned rg = meta.rg0; rg < meta.rg1; ++rg)
        {
            RgConst rc;
            PairRange r;
            rgOfSample(meta, seg, co, w, rg, rc, r);
            int * shiftSlot = b.shifts + (size_t)cellIdx * co.nRg + rg;
            const int refShift = *shiftSlot;
            double sumDel = 0, wSumDel = 0, sumRef = 0, wSumRef = 0;
            const unsigned n = halfForEachActive<false>(r, w, hl, rc.maxLoad, shDev[half], nullptr, [&](int dv, unsigned) {
                const double del = lookupI(rc, co.hist, dv - len);
                const double noDel = lookupI(rc, co.hist, dv - refShift);
                const double probDel = ea1 * del / (del + noDel) + ea2;
                const double probRef = ea1Rg * noDel / (del + noDel) + ea0Rg;
                sumDel += probDel;
                sumRef += probRef;
                wSumDel += probDel * dv;
                wSumRef += probRef * dv;
            });
            if (n >= rc.maxLoad)
                continue;
            sumDel = halfSum(sumDel);
            wSumDel = halfSum(wSumDel);
            sumRef = halfSum(sumRef);
            wSumRef = halfSum(wSumRef);
            sumDelAcc += sumDel;
            wSumDelAcc += wSumDel;
            const int newShift = referenceShiftOf(wSumRef, sumRef, rgNan, rc.stddev, nearAcc);
            if (hl == 0)
                *shiftSlot = newShift;
        }
    }
    const double t0 = blockSumHalves(sumDelAcc, red);
    const double t1 = blockSumHalves(wSumDelAcc, red);
    const double t3 = blockSumHalves(nearAcc, red);           // (every lane of a half counted the same events)
    if (threadIdx.x == 0)
    {
        double * p = PD_OWN_TILE_SLOT(b, cellIdx);
        p[0] = t0;
        p[1] = t1;
        p[3] = t3;          // free during the EM loop (the likelihood-ratio sums use it afterwards); cellLenFinishKernel adds it up
    }
}

// The same on the dense image, one lane per sample.
__global__ __launch_bounds__(kSampleTile) void updateLenDenseKernel(CohortView co, Stage2Buffers b)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[4];
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    const CellState & cs = b.cells[cellIdx];
    if (cs.phase != kCellIterating || b.tileMode[(size_t)imgIdx * b.ownTiles + PD_TILE_IDX] != 0)
        return;
    const int len = (int)cs.len;
    const double g0 = cs.gt[0], g1 = cs.gt[1], g2 = cs.gt[2];
    // what depends on the cell only (seven exp/log calls) is worked out by one lane and shared
    __shared__ double shCell[5];
    if (threadIdx.x == 0)
    {
        const double * rg0 = rg0Slot(b, cellIdx, 0);
        bool nan_ = rg0[3] != 0;
        double e0 = 0, e1 = 0;
        if (!nan_)
        {
            const double aSumRg = exp(rg0[0]) * g0 + exp(rg0[1]) * g1 + exp(rg0[2]) * g2;
            const double l0 = (rg0[0] < PD_X87_EXP_UNDERFLOW) ? -INFINITY : rg0[0];
            const double l1 = (rg0[1] < PD_X87_EXP_UNDERFLOW) ? -INFINITY : rg0[1];
            e0 = exp(l0 + log(g0) - log(aSumRg));
            e1 = exp(l1 + log(g1) - log(aSumRg));
            nan_ = isnan(e0) || isnan(e1);
        }
        shCell[0] = e0;
        shCell[1] = e1;
        shCell[2] = nan_ ? 1.0 : 0.0;
        shCell[3] = log(g1);
        shCell[4] = log(g2);
    }
    __syncthreads();
    const double ea0Rg = shCell[0], ea1Rg = shCell[1], logG1 = shCell[3], logG2 = shCell[4];
    const bool rgNan = shCell[2] != 0.0;
    const unsigned s = PD_TILE_IDX * kSamplesPerBlock + threadIdx.x;
    const bool valid = s < co.nSamples;
    const size_t wtGlobal = (size_t)imgIdx * (b.ownTiles * 4) + PD_TILE_IDX * 4 + (threadIdx.x >> 6);
    const unsigned rows = b.waveRows[wtGlobal];
    const int16_t * col = b.cellDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
    unsigned rg = 0, n = 0;
    LaneRg l = {};
    int refShift = 0;
    double ea1 = 0, ea2 = 0;
    int * shiftSlot = nullptr;
    if (valid)
    {
        rg = co.sampleRg[s];
        l = laneRg(co.rg[rg]);
        n = b.cellN[(size_t)imgIdx * co.nSamples + s];
        shiftSlot = b.shifts + (size_t)cellIdx * co.nRg + rg;
        refShift = *shiftSlot;
        const double * d = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
        const double aSum = exp(d[0]) * g0 + exp(d[1]) * g1 + exp(d[2]) * g2;
        ea1 = exp(d[1] + logG1 - log(aSum));
        ea2 = exp(d[2] + logG2 - log(aSum));
    }
    const bool highCov = n == kDenseHighCov;
    const unsigned nn = highCov ? 0u : n;
    double sumDel = 0, wSumDel = 0, sumRef = 0, wSumRef = 0;
    // (rows in batches: deviations, then all lookups, then the arithmetic -- see dataLikDenseKernel)
    constexpr unsigned kRowBatch = 8;
    for (unsigned j0 = 0; j0 < rows; j0 += kRowBatch)
    {
        int dvs[kRowBatch];
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
            dvs[q] = col[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
        PD_ALL_LOADS_DONE(dvs);
        double dels[kRowBatch], noDels[kRowBatch];
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
        {
            const bool use = j0 + q < nn;
            bool inD, inN;
            const uint64_t pd_ = lanePosClamped(l, dvs[q] - len, inD), pn = lanePosClamped(l, dvs[q] - refShift, inN);
            dels[q] = l.minProb;
            noDels[q] = l.minProb;
            if (use && inD)
                dels[q] = co.hist[pd_];
            if (use && inN)
                noDels[q] = co.hist[pn];
        }
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
        {
            const bool use = j0 + q < nn;
            const int dv = dvs[q];
            const double del = dels[q], noDel = noDels[q];
            const double probDel = ea1 * del / (del + noDel) + ea2;
            const double probRef = ea1Rg * noDel / (del + noDel) + ea0Rg;
            sumDel += use ? probDel : 0.0;
            sumRef += use ? probRef : 0.0;
            wSumDel += use ? probDel * dv : 0.0;
            wSumRef += use ? probRef * dv : 0.0;
        }
    }
    double nearInteger = 0;
    if (valid && !highCov)
    {
        const int newShift = referenceShiftOf(wSumRef, sumRef, rgNan, l.stddev, nearInteger);
        *shiftSlot = newShift;
    }
    const double t0 = blockSum(sumDel, red);
    const double t1 = blockSum(wSumDel, red);
    const double t3 = blockSum(nearInteger, red);
    if (threadIdx.x == 0)
    {
        double * p = PD_OWN_TILE_SLOT(b, cellIdx);
        p[0] = t0;
        p[1] = t1;
        p[3] = t3;          // (see updateLenKernel)
    }
}

// Per-cell kernels that add up per-tile sums run ONE WAVE PER CELL: the lanes fetch the tiles in parallel, the additions
// stay sequential in cohort tile order (bit-identical to a single-lane loop, without its chain of dependent loads).
// Result valid in every lane.
__device__ __forceinline__ void waveTileSums(const Stage2Buffers & b, unsigned c, double & s0, double & s1, double & s2)
{
    s0 = s1 = s2 = 0;
    const unsigned lane = threadIdx.x & 63u;
    for (unsigned base = 0; base < b.nTiles; base += 64u)
    {
        double v0 = 0, v1 = 0, v2 = 0;
        if (base + lane < b.nTiles)
        {
            const double * p = tileSlot(b, c, base + lane);
            v0 = p[0];
            v1 = p[1];
            v2 = p[2];
        }
        const unsigned cnt = min(64u, b.nTiles - base);
        for (unsigned i = 0; i < cnt; ++i)
        {
            s0 += __shfl(v0, (int)i, 64);
            s1 += __shfl(v1, (int)i, 64);
            s2 += __shfl(v2, (int)i, 64);
        }
    }
}

__global__ __launch_bounds__(64) void cellLenFinishKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellIterating)
        return;
    double sumDel, wSumDel, unused;
    waveTileSums(b, c, sumDel, wSumDel, unused);
    // reference shifts of this iteration whose quotient sat within 1e-9 of an integer (counts: the order does not matter)
    double nearInteger = 0;
    for (unsigned t = threadIdx.x; t < b.nTiles; t += 64u)
        nearInteger += tileSlot(b, c, t)[3];
    nearInteger = waveSum(nearInteger);
    if (threadIdx.x != 0)
        return;
    cs.nearIntegerShifts += (unsigned)nearInteger;
    unsigned len;
    if (sumDel == 0)
        len = 0;
    else
    {
        const double l = wSumDel / sumDel;
        len = (l < 0) ? 0u : (unsigned)round(l);
    }
    cs.len = len;
}

// Allele-frequency update and convergence test (reference :623-659).
__global__ __launch_bounds__(64) void cellFreqFinishKernel(Stage2Buffers b, unsigned iteration)
{
    const unsigned c = blockIdx.x;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellIterating)
        return;
    double unused0, unused1, sum;
    waveTileSums(b, c, unused0, unused1, sum);
    if (threadIdx.x != 0)
        return;
    struct Mark
    {
        CellState & cs;
        unsigned *  flag;
        __device__ ~Mark()
        {
            if (cs.phase == kCellIterating)
                *flag = 1u;               // every writer stores the same value
        }
    } mark{cs, b.iterFlags + iteration};
    const double freq = sum / 2.0 / b.nSamplesGlobal;
    cs.freq = freq;
    if (freq == 0)
    {
        cs.phase = kCellStopped;
        return;
    }
    priors(freq, b.somatic, cs.gt);
    for (int i = 0; i < cs.nVisited; ++i)
        if (cs.visitedLen[i] == (int)cs.len && fabs(cs.visitedFreq[i] - freq) <= 0.0001)
        {
            cs.phase = kCellConverged;
            break;
        }
    loopHead(cs, b);                        // head of the next iteration (also retires cells that hit the iteration cap)
}

// =====================================================================================================================
// Likelihood-ratio terms (reference :490-508). which: 0 = current priors, 1 = previous priors.
// =====================================================================================================================
// `which`: 0 = the current priors, 1 = those of the previous estimate; the two sums go to entries 2*which, 2*which+1 of the
// tile slot, so both evaluations of a converged cell travel in ONE exchange.
__global__ __launch_bounds__(kSampleTile) void lrKernel(CohortView co, Stage2Buffers b, int wantPhase, int which)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[4];
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    const CellState & cs = b.cells[cellIdx];
    if (cs.phase != wantPhase)
        return;
    const unsigned s = PD_TILE_IDX * kSamplesPerBlock + threadIdx.x;
    double del = 0, noDel = 0;
    if (threadIdx.x < (unsigned)kSamplesPerBlock && s < co.nSamples)
    {
        const double * d = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
        const double * g = which == 0 ? cs.gt : cs.prevGt;
        const double t0 = exp(d[0]), t1 = exp(d[1]), t2 = exp(d[2]);
        const double p0 = t0 * g[0], p1 = t1 * g[1], p2 = t2 * g[2];
        const double pAll = p0 + p1 + p2;
        const double a0 = p0 / pAll, a1 = p1 / pAll, a2 = p2 / pAll;
        del = log(a0 * t0 + a1 * t1 + a2 * t2);
        noDel = d[0];
    }
    const double s0 = blockSum(del, red);
    const double s1 = blockSum(noDel, red);
    if (threadIdx.x == 0)
    {
        double * p = PD_OWN_TILE_SLOT(b, cellIdx) + 2 * which;
        p[0] = s0;
        p[1] = s1;
    }
}

__device__ __forceinline__ double sumLR(const Stage2Buffers & b, unsigned c)       // by one wave, see waveTileSums
{
    double del, noDel, unused;
    waveTileSums(b, c, del, noDel, unused);
    return del - noDel;
}

// LR of the current estimate (tile-slot entries 0, 1) and of the previous estimate (entries 2, 3, evaluated on the
// likelihoods recomputed at the previous length): keep the better one (reference :640-657).
__global__ __launch_bounds__(64) void cellConvergedKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellConverged)
        return;
    // one wave: lanes fetch the tiles in parallel, the additions stay sequential in cohort tile order (see waveTileSums)
    double s[4] = {0, 0, 0, 0};
    const unsigned lane = threadIdx.x & 63u;
    for (unsigned base = 0; base < b.nTiles; base += 64u)
    {
        double v[4] = {0, 0, 0, 0};
        if (base + lane < b.nTiles)
        {
            const double * p = tileSlot(b, c, base + lane);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = p[k];
        }
        const unsigned cnt = min(64u, b.nTiles - base);
        for (unsigned i = 0; i < cnt; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                s[k] += __shfl(v[k], (int)i, 64);
    }
    if (threadIdx.x != 0)
        return;
    cs.logLR = s[0] - s[1];
    cs.prevLogLR = s[2] - s[3];
    if (cs.prevLogLR > cs.logLR)
    {
        cs.len = cs.prevLen;
        cs.freq = cs.prevFreq;
        cs.restoreZeroShifts = 1;       // prevShifts is the all-zero vector captured before the loop
    }
}

// Post-loop test (reference :661-664).
__global__ void cellPostLoopKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.nCells)
        return;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellConverged && cs.phase != kCellStopped && cs.phase != kCellIterating)
        return;
    if (cs.freq < 0.0000000001 || cs.len < b.minLen)
        cs.phase = kCellDead;
    else
        cs.phase = kCellFinal;
}

// =====================================================================================================================
// Final pass (reference :255-337): data likelihoods once more, log10 sums -> phred-scaled likelihoods, LAD / DAD
// counts, first/last, and the pool of supporting pair starts / ends.
// =====================================================================================================================
__global__ __launch_bounds__(kSampleTile) void finalKernel(SegmentView seg, CohortView co, Stage2Buffers b,
                                                           const uint8_t * __restrict__ flags)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ int shDev[kHalvesPerBlock][kHalfBufCap];
    __shared__ unsigned shSt[kHalvesPerBlock][kHalfBufCap];
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    CellState & cs = b.cells[cellIdx];
    if (cs.phase != kCellFinal || b.tileMode[(size_t)imgIdx * b.ownTiles + PD_TILE_IDX] != kTileGeneric)
        return;
    const unsigned w = (unsigned)cs.window;
    const int len = (int)cs.len;
    const unsigned hl = threadIdx.x & 31u, half = threadIdx.x >> 5;
    __shared__ SampleMeta shMeta[kSamplesPerBlock];
    loadTileMeta(seg, co, w, shMeta, PD_TILE_IDX);
    for (unsigned k = 0; k < (unsigned)(kSamplesPerBlock / kHalvesPerBlock); ++k)
    {
        const unsigned s = PD_TILE_IDX * kSamplesPerBlock + half + k * kHalvesPerBlock;
        if (s >= co.nSamples)
            break;
        double logL[3] = {0, 0, 0}, gtLogs[3] = {0, 0, 0};
        unsigned lad[3] = {0, 0, 0}, dad[5] = {0, 0, 0, 0, 0};
        unsigned nUsed = 0;
        bool allEq = true;
        int delLower = 0x7fffffff, delUpper = 0;
        unsigned mn = 0xffffffffu, mx = 0;
        const SampleMeta & meta = shMeta[half + k * kHalvesPerBlock];
        for (unsigned rg = meta.rg0; rg < meta.rg1; ++rg)
        {
            RgConst rc;
            PairRange r;
            rgOfSample(meta, seg, co, w, rg, rc, r);
            const int refShift = cs.restoreZeroShifts ? 0 : b.shifts[(size_t)cellIdx * co.nRg + rg];
            const int rgLower = (int)((unsigned)len - (unsigned)rc.lowerQ);
            const int rgUpper = (int)((unsigned)len + (unsigned)rc.upperQ);
            double l0 = 0, l1 = 0, l2 = 0, t0 = 0, t1 = 0, t2 = 0;
            bool eq = true;
            unsigned rgMn = 0xffffffffu, rgMx = 0;
            unsigned rgLad[3] = {0, 0, 0}, rgDad[5] = {0, 0, 0, 0, 0};
            const unsigned n = halfForEachActive<true>(r, w, hl, rc.maxLoad, shDev[half], shSt[half], [&](int dv, unsigned st) {
                int cls;
                if (dv > rc.upperQ)
                    cls = (dv < rgLower) ? 2 : ((dv <= rgUpper) ? 3 : 4);
                else
                    cls = (dv < rgUpper) ? 0 : 1;
                const long long pr = lookupPos(rc, dv - refShift), pdl = lookupPos(rc, dv - len);
                double4 tr = make_double4(rc.minProb, rc.logMinProb, rc.log10MinProb, 0.0), td = tr;
                if (pr >= 0)
                    tr = co.tab[pr];
                if (pdl >= 0)
                    td = co.tab[pdl];
                const double ref = tr.x, del = td.x;
                const int lcls = (ref >= 2 * del) ? 0 : ((del >= 2 * ref) ? 2 : 1);
                l0 += tr.y;
                t0 += tr.z;
                l1 += log(ref + del) - PD_LN2;
                t1 += log10(ref + del) - PD_LOG10_2;
                l2 += td.y;
                t2 += td.z;
                eq = eq && (ref == del);
                const int ins = max(0, dv + rc.median - 2 * rc.readLength);
                rgMn = min(rgMn, st);
                rgMx = max(rgMx, st + (unsigned)ins);
                // class counts: lane 0 takes part in every iteration that has any lane, so its totals are complete
#pragma unroll
                for (int c = 0; c < 5; ++c)
                    rgDad[c] += halfCount(cls == c);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    rgLad[c] += halfCount(lcls == c);
            });
            if (n >= rc.maxLoad)
                continue;
            delLower = rgLower;
            delUpper = rgUpper;
            logL[0] += halfSum(l0);
            logL[1] += halfSum(l1);
            logL[2] += halfSum(l2);
            gtLogs[0] += halfSum(t0);
            gtLogs[1] += halfSum(t1);
            gtLogs[2] += halfSum(t2);
            allEq = allEq && (halfCount(!eq) == 0u);
            nUsed += n;
            mn = min(mn, rgMn);
            mx = max(mx, rgMx);
#pragma unroll
            for (int c = 0; c < 5; ++c)
                dad[c] += rgDad[c];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                lad[c] += rgLad[c];
        }
        mn = halfMinU(mn);
        mx = halfMaxU(mx);
        if (mn == 0xffffffffu)
            mn = 0;
        // supporting pairs use the borders of the LAST usable read group for all read groups of the sample
        for (unsigned rg = meta.rg0; rg < meta.rg1; ++rg)
        {
            RgConst rc;
            PairRange r;
            rgOfSample(meta, seg, co, w, rg, rc, r);
            halfForEachActive<true>(r, w, hl, rc.maxLoad, shDev[half], shSt[half], [&](int dv, unsigned st) {
                const bool supp = dv >= delLower && dv <= delUpper;
                const unsigned long long bal = __ballot(supp) & halfMask();
                const unsigned cnt = (unsigned)__popcll(bal);
                if (cnt)                                              // same for all lanes in here
                {
                    unsigned slot0 = 0;
                    if (hl == 0)
                        slot0 = atomicAdd(&cs.suppCount, cnt);
                    slot0 = (unsigned)__shfl((int)slot0, 0, 32);
                    if (supp)
                    {
                        const unsigned rank = (unsigned)__popcll(bal & ((1ull << (threadIdx.x & 63u)) - 1ull));
                        b.suppFirst[cs.suppBase + slot0 + rank] = st;
                        b.suppLast[cs.suppBase + slot0 + rank] = st + (unsigned)max(0, dv + rc.median - 2 * rc.readLength);
                    }
                }
            });
        }
        if (hl != 0)
            continue;
        double d[3];
        double gl[3] = {gtLogs[0], gtLogs[1], gtLogs[2]};
        if (nUsed == 0)
        {
            // sum(gtLogs) == 0: reference-like triple, gtLogs stay untouched (all zero)
            d[0] = 0;
            d[1] = PD_LOG_PSEUDO;
            d[2] = PD_LOG_PSEUDO;
        }
        else
        {
            const double maxGt = fmax(fmax(gl[0], gl[1]), gl[2]);
            gl[0] -= maxGt;
            gl[1] -= maxGt;
            gl[2] -= maxGt;
            if (allEq)
            {
                gl[0] = gl[1] = gl[2] = 0;      // x87: a hair apart, not equal; all phred values round to 0
            }
            else if (gl[0] == gl[1] && gl[0] == gl[2])
            {
                gl[0] = 0;
                gl[1] = -10;
                gl[2] = -10;
            }
            finishTriple(d, logL, nUsed, allEq);
        }
        const size_t o = (size_t)cellIdx * co.nSamples + s;
        b.dl[o * 3 + 0] = d[0];
        b.dl[o * 3 + 1] = d[1];
        b.dl[o * 3 + 2] = d[2];
        // phred-scaled genotype likelihoods; exp() of log10-based values as in the reference
        uint32_t pl[3];
        {
            const double gTot = log10(exp(gl[0]) + exp(gl[1]) + exp(gl[2]));
            const double p0 = -10 * (gl[0] - gTot);
            const double p1 = -10 * (gl[1] - gTot);
            const double p2 = -10 * (gl[2] - gTot);
            const double mnP = fmin(fmin(p0, p1), p2);
            pl[0] = (uint32_t)round(p0 - mnP);
            pl[1] = (uint32_t)round(p1 - mnP);
            pl[2] = (uint32_t)round(p2 - mnP);
        }
        if (flags[(size_t)s * seg.nWindows + w] & 1u)
            pl[0] = pl[1] = pl[2] = 0;
        for (int c = 0; c < 3; ++c)
        {
            b.pl[o * 3 + c] = pl[c];
            b.lad[o * 3 + c] = lad[c];
        }
        for (int c = 0; c < 5; ++c)
            b.dad[o * 5 + c] = dad[c];
        b.fl[o * 2 + 0] = mn;
        b.fl[o * 2 + 1] = mx;
    }
}

// The final pass on the dense image, one lane per sample. log / log10 of the heterozygous term again come from one
// mantissa product per sample; the supporting pairs are appended to the pool with one atomic per wave.
// NarrowTable: gather the EM kernels' 16-byte entries and take log10 here (candidate-rich batches: the 32-byte table of 10,000
// samples is 320 MB, twice what the last-level cache holds beside the image -- 10.3 GB of HBM per launch through it, 5.3 GB
// this way, same time); else the 32-byte entries {value, log, log10} (sparse batches: 221 us per launch against 260).
template <bool NarrowTable>
__global__ __launch_bounds__(kSampleTile) void finalDenseKernel(CohortView co, Stage2Buffers b, const uint8_t * __restrict__ flags,
                                                                unsigned nWindows, unsigned firstPos, unsigned shift)
{
    if (PD_TILE_IDX >= b.ownTiles)      // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    const unsigned cellIdx = PD_CELL_IDX;
    [[maybe_unused]] const unsigned imgIdx = b.cells[cellIdx].image;          // dense image of the cell's window
    CellState & cs = b.cells[cellIdx];
    if (cs.phase != kCellFinal || b.tileMode[(size_t)imgIdx * b.ownTiles + PD_TILE_IDX] != 0)
        return;
    const unsigned w = (unsigned)cs.window;
    const unsigned stBase = firstPos + w * shift;          // the image keeps pair starts as 16-bit offsets from the window
    const int len = (int)cs.len;
    const unsigned s = PD_TILE_IDX * kSamplesPerBlock + threadIdx.x;
    const bool valid = s < co.nSamples;
    const size_t wtGlobal = (size_t)imgIdx * (b.ownTiles * 4) + PD_TILE_IDX * 4 + (threadIdx.x >> 6);
    const unsigned rows = b.waveRows[wtGlobal];
    const int16_t * col = b.cellDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
    const int16_t * colSt = b.cellStart + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
    unsigned rg = 0, n = 0;
    RgConst rc = {};
    int refShift = 0;
    if (valid)
    {
        rg = co.sampleRg[s];
        rc = co.rg[rg];
        n = b.cellN[(size_t)imgIdx * co.nSamples + s];
        refShift = cs.restoreZeroShifts ? 0 : b.shifts[(size_t)cellIdx * co.nRg + rg];
    }
    const bool highCov = n == kDenseHighCov;
    const unsigned nn = highCov ? 0u : n;
    const int rgLower = (int)((unsigned)len - (unsigned)rc.lowerQ);
    const int rgUpper = (int)((unsigned)len + (unsigned)rc.upperQ);
    double l0 = 0, l2 = 0, t0 = 0, t2 = 0, prod = 1.0;
    int esum = 0;
    bool eq = true;
    unsigned mn = 0xffffffffu, mx = 0, nSupp = 0;
    unsigned lad[3] = {0, 0, 0}, dad[5] = {0, 0, 0, 0, 0};
    // (rows in batches: deviations and starts, then all lookups, then the arithmetic -- see dataLikDenseKernel)
    constexpr unsigned kRowBatch = 6;
    for (unsigned j0 = 0; j0 < rows; j0 += kRowBatch)
    {
    int dvs[kRowBatch], sts[kRowBatch];
#pragma unroll
    for (unsigned q = 0; q < kRowBatch; ++q)
        dvs[q] = col[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
    PD_ALL_LOADS_DONE(dvs);
    // (log10 by the function that fills the 32-byte table, logTableKernel: the same bits)
    double2 trs2[kRowBatch], tds2[kRowBatch];
    double4 trs[kRowBatch], tds[kRowBatch];
    bool inR[kRowBatch], inD[kRowBatch];
#pragma unroll
    for (unsigned q = 0; q < kRowBatch; ++q)
    {
        const bool use = j0 + q < nn;
        const long long pr = lookupPos(rc, dvs[q] - refShift), pdl = lookupPos(rc, dvs[q] - len);
        inR[q] = use && pr >= 0;
        inD[q] = use && pdl >= 0;
        if constexpr (NarrowTable)
        {
            trs2[q] = make_double2(rc.minProb, rc.logMinProb);
            tds2[q] = trs2[q];
            if (inR[q])
                trs2[q] = co.tab2[pr];
            if (inD[q])
                tds2[q] = co.tab2[pdl];
        }
        else
        {
            trs[q] = make_double4(rc.minProb, rc.logMinProb, rc.log10MinProb, 0.0);
            tds[q] = trs[q];
            if (inR[q])
                trs[q] = co.tab[pr];
            if (inD[q])
                tds[q] = co.tab[pdl];
        }
    }
    // (the starts are wanted by the arithmetic only: loaded behind the lookups, so that no lookup is issued behind a wait for them)
#pragma unroll
    for (unsigned q = 0; q < kRowBatch; ++q)
        sts[q] = colSt[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
    if constexpr (NarrowTable)
    {
#pragma unroll
        for (unsigned q = 0; q < kRowBatch; ++q)
        {
            trs[q] = make_double4(trs2[q].x, trs2[q].y, inR[q] ? log10(trs2[q].x) : rc.log10MinProb, 0.0);
            tds[q] = make_double4(tds2[q].x, tds2[q].y, inD[q] ? log10(tds2[q].x) : rc.log10MinProb, 0.0);
        }
    }
#pragma unroll
    for (unsigned q = 0; q < kRowBatch; ++q)
    {
        const bool use = j0 + q < nn;       // (rows past the image's are padding like rows past the lane's own count)
        const int dv = dvs[q];
        const unsigned st = stBase + (unsigned)sts[q];
        int cls;
        if (dv > rc.upperQ)
            cls = (dv < rgLower) ? 2 : ((dv <= rgUpper) ? 3 : 4);
        else
            cls = (dv < rgUpper) ? 0 : 1;
        const double4 tr = trs[q], td = tds[q];
        const double ref = tr.x, del = td.x;
        const int lcls = (ref >= 2 * del) ? 0 : ((del >= 2 * ref) ? 2 : 1);
        int e;
        const double m = frexp(ref + del, &e);
        l0 += use ? tr.y : 0.0;
        t0 += use ? tr.z : 0.0;
        l2 += use ? td.y : 0.0;
        t2 += use ? td.z : 0.0;
        prod *= use ? m : 1.0;
        esum += use ? e : 0;
        eq = eq && (!use || ref == del);
        const unsigned last = st + (unsigned)max(0, dv + rc.median - 2 * rc.readLength);
        mn = use ? min(mn, st) : mn;
        mx = use ? max(mx, last) : mx;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            dad[c] += (use && cls == c) ? 1u : 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            lad[c] += (use && lcls == c) ? 1u : 0u;
        nSupp += (use && dv >= rgLower && dv <= rgUpper) ? 1u : 0u;
    }
    }
    // supporting pairs: one pool allocation per wave, lanes fill their own stretch
    {
        unsigned incl = nSupp;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
            if ((int)(threadIdx.x & 63u) >= o)
                incl += t;
        }
        const unsigned waveTotal = (unsigned)__shfl((int)incl, 63, 64);
        if (waveTotal)                                        // wave-uniform
        {
            unsigned base0 = 0;
            if ((threadIdx.x & 63u) == 0)
                base0 = atomicAdd(&cs.suppCount, waveTotal);
            base0 = (unsigned)__shfl((int)base0, 0, 64);
            unsigned at = base0 + incl - nSupp;
            if (nSupp)
                for (unsigned j = 0; j < nn; ++j)
                {
                    const int dv = col[j * 64];
                    if (dv >= rgLower && dv <= rgUpper)
                    {
                        const unsigned st = stBase + (unsigned)(int)colSt[j * 64];
                        b.suppFirst[cs.suppBase + at] = st;
                        b.suppLast[cs.suppBase + at] = st + (unsigned)max(0, dv + rc.median - 2 * rc.readLength);
                        ++at;
                    }
                }
        }
    }
    if (!valid)
        return;
    if (mn == 0xffffffffu)
        mn = 0;
    double d[3];
    double gl[3] = {0, 0, 0};
    if (nn == 0)
    {
        // sum(gtLogs) == 0: reference-like triple, gtLogs stay untouched (all zero)
        d[0] = 0;
        d[1] = PD_LOG_PSEUDO;
        d[2] = PD_LOG_PSEUDO;
    }
    else
    {
        const double l1 = log(prod) + (double)(esum - (int)nn) * PD_LN2;
        gl[0] = t0;
        gl[1] = log10(prod) + (double)(esum - (int)nn) * PD_LOG10_2;
        gl[2] = t2;
        const double maxGt = fmax(fmax(gl[0], gl[1]), gl[2]);
        gl[0] -= maxGt;
        gl[1] -= maxGt;
        gl[2] -= maxGt;
        if (eq)
        {
            gl[0] = gl[1] = gl[2] = 0;      // x87: a hair apart, not equal; all phred values round to 0
        }
        else if (gl[0] == gl[1] && gl[0] == gl[2])
        {
            gl[0] = 0;
            gl[1] = -10;
            gl[2] = -10;
        }
        const double logL[3] = {l0, l1, l2};
        finishTriple(d, logL, nn, eq);
    }
    const size_t o = (size_t)cellIdx * co.nSamples + s;
    b.dl[o * 3 + 0] = d[0];
    b.dl[o * 3 + 1] = d[1];
    b.dl[o * 3 + 2] = d[2];
    uint32_t pl[3];
    {
        const double gTot = log10(exp(gl[0]) + exp(gl[1]) + exp(gl[2]));
        const double p0 = -10 * (gl[0] - gTot);
        const double p1 = -10 * (gl[1] - gTot);
        const double p2 = -10 * (gl[2] - gTot);
        const double mnP = fmin(fmin(p0, p1), p2);
        pl[0] = (uint32_t)round(p0 - mnP);
        pl[1] = (uint32_t)round(p1 - mnP);
        pl[2] = (uint32_t)round(p2 - mnP);
    }
    if (flags[(size_t)s * nWindows + w] & 1u)
        pl[0] = pl[1] = pl[2] = 0;
    for (int c = 0; c < 3; ++c)
    {
        b.pl[o * 3 + c] = pl[c];
        b.lad[o * 3 + c] = lad[c];
    }
    for (int c = 0; c < 5; ++c)
        b.dad[o * 5 + c] = dad[c];
    b.fl[o * 2 + 0] = mn;
    b.fl[o * 2 + 1] = mx;
}

// =====================================================================================================================
// Multi tiles (samples with several read groups): the same three passes with ONE LANE PER READ GROUP over a second
// dense image, then one lane per sample adds up the results of its read groups in read-group order.
// =====================================================================================================================
// Dense image of the multi tiles: one workgroup per (cell, 64-read-group wave tile).
__global__ __launch_bounds__(kGatherThreads) void multiGatherKernel(SegmentView seg, CohortView co, Stage2Buffers b)
{
    __shared__ short shDev[64][kDenseCap + 1];             // deviations: 16 bits in the dense image (else: generic kernels)
    __shared__ short shSt[64][kDenseCap + 1];              // start positions relative to the cell's window (see stBase)
    __shared__ unsigned shN[64];
    __shared__ unsigned shRows;
    const unsigned imgIdx = blockIdx.y, mw = blockIdx.x;           // one workgroup per (image, 64-read-group wave tile)
    const unsigned cellIdx = b.cells[imgIdx].imageFirst;           // (the image's first cell: window, and where "generic" is counted)
    const unsigned mtI = b.mt.waveTile[mw];
    if (b.tileMode[(size_t)imgIdx * b.ownTiles + b.mt.tile[mtI]] != kTileMulti)
        return;
    const unsigned rgBase = b.mt.rgLo[mtI] + (mw - b.mt.waveBase[mtI]) * 64u, rgHi = b.mt.rgHi[mtI];
    const unsigned w = (unsigned)b.cells[cellIdx].window;
    const unsigned stBase = seg.firstPos + w * seg.shift;
    const unsigned hl = threadIdx.x & 31u, half = threadIdx.x >> 5;
    const unsigned long long hm = halfMask();
    const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
    if (threadIdx.x == 0)
        shRows = 0;
    __syncthreads();
    constexpr unsigned kHalves = kGatherThreads / 32;
    for (unsigned k = 0; k < 64u / kHalves; ++k)
    {
        const unsigned inWt = half + k * kHalves;
        const unsigned rg = rgBase + inWt;
        if (rg >= rgHi)                                    // uniform per half-wave
        {
            if (hl == 0)
                shN[inWt] = 0;
            continue;
        }
        const PairRange r = pairRange(seg, rg, w);
        unsigned n = 0;
        bool farStart = false;
        for (unsigned base = r.begin; base < r.end; base += 64u)
        {
            PairDO p[2];
            unsigned st[2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
            {
                const unsigned i = base + 32u * g + hl;
                p[g].dev = 0;
                p[g].off = 0;
                st[g] = 0;
                if (i < r.end)
                {
                    p[g] = r.p[i];
                    st[g] = r.st[i];
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
            {
                const bool act = p[g].off > w;
                const unsigned long long bal = __ballot(act) & hm;
                if (act)
                {
                    const unsigned pos = n + (unsigned)__popcll(bal & below);
                    if (pos < (unsigned)kDenseCap)
                    {
                        const int rel = (int)(st[g] - stBase);
                        shDev[inWt][pos] = (short)p[g].dev;
                        shSt[inWt][pos] = (short)rel;
                        farStart = farStart || rel < -32768 || rel > 32767 || p[g].dev < -32768 || p[g].dev > 32767;
                    }
                }
                n += (unsigned)__popcll(bal);
            }
        }
        if (halfCount(farStart) != 0u && hl == 0)          // cannot be represented here: the tile goes to the generic kernels
        {
            atomicMax(&b.tileMode[(size_t)imgIdx * b.ownTiles + b.mt.tile[mtI]], (unsigned)kTileGeneric);
            atomicAdd(&counterSlot(b, b.rank, cellIdx)[3], 1u);
        }
        const unsigned nDense = (n >= co.rg[rg].maxLoad) ? kDenseHighCov : n;   // n <= kDenseCap otherwise (else: kTileGeneric)
        if (hl == 0)
        {
            shN[inWt] = nDense;
            b.mN[(size_t)imgIdx * b.nRg + rg] = (uint16_t)nDense;
            if (nDense != kDenseHighCov)
                atomicMax(&shRows, nDense);
        }
    }
    __syncthreads();
    const unsigned rows = shRows;
    const size_t wtGlobal = (size_t)imgIdx * b.mt.nWaves + mw;
    int16_t * outD = b.mDev + wtGlobal * (kDenseCap * 64);
    int16_t * outS = b.mStart + wtGlobal * (kDenseCap * 64);
    for (unsigned idx = threadIdx.x; idx < rows * 64u; idx += blockDim.x)
    {
        const unsigned j = idx >> 6, lane = idx & 63u, nl = shN[lane];
        const bool has = nl != kDenseHighCov && j < nl;
        outD[idx] = has ? shDev[lane][j] : (short)0;
        outS[idx] = has ? shSt[lane][j] : (short)0;
    }
    if (threadIdx.x == 0)
        b.mWaveRows[wtGlobal] = (uint8_t)rows;
}

// flag word of a read group's parked result
#define PD_RGF_HIGHCOV 0x80000000u
#define PD_RGF_EQ 0x40000000u

__global__ __launch_bounds__(kSampleTile) void dataLikMultiKernel(CohortView co, Stage2Buffers b, int mode)
{
    if (blockIdx.x >= b.mt.n)           // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[4];
    const unsigned cellIdx = blockIdx.y, mtI = blockIdx.x, tileIdx = b.mt.tile[mtI];
    const unsigned imgIdx = b.cells[cellIdx].image;
    const CellState & cs = b.cells[cellIdx];
    const int wantPhase = mode == 1 ? kCellConverged : kCellIterating;
    if ((mode != 2 && cs.phase != wantPhase) || b.tileMode[(size_t)imgIdx * b.ownTiles + tileIdx] != kTileMulti)
        return;
    const int len = mode == 2 ? cs.initLen : (int)(mode == 1 ? cs.prevLen : cs.len);
    const unsigned rgLo = b.mt.rgLo[mtI], rgHi = b.mt.rgHi[mtI];
    // phase 1: one lane per read group
    for (unsigned base = rgLo; base < rgHi; base += kSampleTile)
    {
        const unsigned rg = base + threadIdx.x;
        const bool valid = rg < rgHi;
        const size_t wtGlobal = (size_t)imgIdx * b.mt.nWaves + b.mt.waveBase[mtI] + ((base - rgLo) >> 6) + (threadIdx.x >> 6);
        const unsigned rows = (base - rgLo) + (threadIdx.x & ~63u) < rgHi - rgLo ? b.mWaveRows[wtGlobal] : 0u;
        const int16_t * col = b.mDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
        unsigned n = 0;
        LaneRg l = {};
        int refShift = 0;
        if (valid)
        {
            l = laneRg(co.rg[rg]);
            n = b.mN[(size_t)imgIdx * b.nRg + rg];
            refShift = (mode != 1) ? b.shifts[(size_t)cellIdx * co.nRg + rg] : 0;
        }
        const bool highCov = n == kDenseHighCov;
        const unsigned nn = highCov ? 0u : n;
        double g0 = 0, g2 = 0, prod = 1.0;
        int esum = 0;
        bool eq = true;
        // (rows in batches: deviations, then all lookups, then the arithmetic -- see dataLikDenseKernel)
        constexpr unsigned kRowBatch = 8;
        for (unsigned j0 = 0; j0 < rows; j0 += kRowBatch)
        {
            int dvs[kRowBatch];
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
                dvs[q] = col[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
            PD_ALL_LOADS_DONE(dvs);
            double2 trs[kRowBatch], tds[kRowBatch];
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
            {
                const bool use = j0 + q < nn;
                bool inR, inD;
                const uint64_t pr = lanePosClamped(l, dvs[q] - refShift, inR), pdl = lanePosClamped(l, dvs[q] - len, inD);
                trs[q] = make_double2(l.minProb, l.logMinProb);
                tds[q] = trs[q];
                if (use && inR)
                    trs[q] = co.tab2[pr];
                if (use && inD)
                    tds[q] = co.tab2[pdl];
            }
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
            {
                const bool use = j0 + q < nn;
                const double2 tr = trs[q], td = tds[q];
                int e;
                const double m = frexp(tr.x + td.x, &e);
                g0 += use ? tr.y : 0.0;
                g2 += use ? td.y : 0.0;
                prod *= use ? m : 1.0;
                esum += use ? e : 0;
                eq = eq && (!use || tr.x == td.x);
            }
        }
        if (valid)
        {
            const double g1 = nn ? log(prod) + (double)(esum - (int)nn) * PD_LN2 : 0.0;
            double * od = b.rgOutD + ((size_t)cellIdx * b.nRg + rg) * 8;
            od[0] = g0;
            od[1] = g1;
            od[2] = g2;
            b.rgOutU[((size_t)cellIdx * b.nRg + rg) * 12] = nn | (highCov ? PD_RGF_HIGHCOV : 0u) | (eq ? PD_RGF_EQ : 0u);
            if (rg == 0 && b.rank == 0)
            {
                double * r0 = rg0Slot(b, cellIdx, 0);
                if (highCov)
                {
                    r0[0] = r0[1] = r0[2] = 0;
                    r0[3] = 1;
                }
                else
                {
                    const double m = fmax(fmax(g0, g1), g2);
                    r0[0] = g0 - m;
                    r0[1] = g1 - m;
                    r0[2] = g2 - m;
                    r0[3] = 0;
                }
            }
        }
    }
    __syncthreads();                                       // the parked results are read by other lanes of this workgroup
    // phase 2: one lane per sample
    double term = 0;
    const unsigned s = tileIdx * kSamplesPerBlock + threadIdx.x;
    if (s < co.nSamples)
    {
        double logL[3] = {0, 0, 0};
        unsigned nUsed = 0;
        bool allEq = true;
        for (unsigned rg = co.sampleRg[s]; rg < co.sampleRg[s + 1]; ++rg)
        {
            const unsigned f = b.rgOutU[((size_t)cellIdx * b.nRg + rg) * 12];
            if (f & PD_RGF_HIGHCOV)
                continue;
            const double * od = b.rgOutD + ((size_t)cellIdx * b.nRg + rg) * 8;
            logL[0] += od[0];
            logL[1] += od[1];
            logL[2] += od[2];
            nUsed += f & 0xffffu;
            allEq = allEq && (f & PD_RGF_EQ) != 0u;
        }
        double d[3];
        finishTriple(d, logL, nUsed, allEq);
        double * out = b.dl + ((size_t)cellIdx * co.nSamples + s) * 3;
        out[0] = d[0];
        out[1] = d[1];
        out[2] = d[2];
        if (mode == 0)
        {
            const double p0 = exp(d[0]) * cs.gt[0];
            const double p1 = exp(d[1]) * cs.gt[1];
            const double p2 = exp(d[2]) * cs.gt[2];
            const double pAll = p0 + p1 + p2;
            term = (p1 + 2 * p2) / pAll;
        }
    }
    if (mode == 0)
    {
        const double t = blockSum(term, red);
        if (threadIdx.x == 0)
            tileSlot(b, cellIdx, b.rank * b.tilesPerRank + tileIdx)[2] = t;
    }
}

__global__ __launch_bounds__(kSampleTile) void updateLenMultiKernel(CohortView co, Stage2Buffers b)
{
    if (blockIdx.x >= b.mt.n)           // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ double red[4];
    const unsigned cellIdx = blockIdx.y, mtI = blockIdx.x, tileIdx = b.mt.tile[mtI];
    const unsigned imgIdx = b.cells[cellIdx].image;
    const CellState & cs = b.cells[cellIdx];
    if (cs.phase != kCellIterating || b.tileMode[(size_t)imgIdx * b.ownTiles + tileIdx] != kTileMulti)
        return;
    const int len = (int)cs.len;
    const double g0 = cs.gt[0], g1 = cs.gt[1], g2 = cs.gt[2];
    // what depends on the cell only (seven exp/log calls) is worked out by one lane and shared
    __shared__ double shCell[5];
    if (threadIdx.x == 0)
    {
        const double * rg0 = rg0Slot(b, cellIdx, 0);
        bool nan_ = rg0[3] != 0;
        double e0 = 0, e1 = 0;
        if (!nan_)
        {
            const double aSumRg = exp(rg0[0]) * g0 + exp(rg0[1]) * g1 + exp(rg0[2]) * g2;
            const double l0 = (rg0[0] < PD_X87_EXP_UNDERFLOW) ? -INFINITY : rg0[0];
            const double l1 = (rg0[1] < PD_X87_EXP_UNDERFLOW) ? -INFINITY : rg0[1];
            e0 = exp(l0 + log(g0) - log(aSumRg));
            e1 = exp(l1 + log(g1) - log(aSumRg));
            nan_ = isnan(e0) || isnan(e1);
        }
        shCell[0] = e0;
        shCell[1] = e1;
        shCell[2] = nan_ ? 1.0 : 0.0;
        shCell[3] = log(g1);
        shCell[4] = log(g2);
    }
    __syncthreads();
    const double ea0Rg = shCell[0], ea1Rg = shCell[1], logG1 = shCell[3], logG2 = shCell[4];
    const bool rgNan = shCell[2] != 0.0;
    const unsigned rgLo = b.mt.rgLo[mtI], rgHi = b.mt.rgHi[mtI];
    double sumDelAcc = 0, wSumDelAcc = 0;                  // over the read groups this lane handles, in read-group order
    double nearAcc = 0;
    for (unsigned base = rgLo; base < rgHi; base += kSampleTile)
    {
        const unsigned rg = base + threadIdx.x;
        const bool valid = rg < rgHi;
        const size_t wtGlobal = (size_t)imgIdx * b.mt.nWaves + b.mt.waveBase[mtI] + ((base - rgLo) >> 6) + (threadIdx.x >> 6);
        const unsigned rows = (base - rgLo) + (threadIdx.x & ~63u) < rgHi - rgLo ? b.mWaveRows[wtGlobal] : 0u;
        const int16_t * col = b.mDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
        unsigned n = 0;
        LaneRg l = {};
        int refShift = 0;
        double ea1 = 0, ea2 = 0;
        int * shiftSlot = nullptr;
        if (valid)
        {
            l = laneRg(co.rg[rg]);
            n = b.mN[(size_t)imgIdx * b.nRg + rg];
            shiftSlot = b.shifts + (size_t)cellIdx * co.nRg + rg;
            refShift = *shiftSlot;
            const double * d = b.dl + ((size_t)cellIdx * co.nSamples + b.mt.rgSample[rg]) * 3;
            const double aSum = exp(d[0]) * g0 + exp(d[1]) * g1 + exp(d[2]) * g2;
            ea1 = exp(d[1] + logG1 - log(aSum));
            ea2 = exp(d[2] + logG2 - log(aSum));
        }
        const bool highCov = n == kDenseHighCov;
        const unsigned nn = highCov ? 0u : n;
        double sumDel = 0, wSumDel = 0, sumRef = 0, wSumRef = 0;
        constexpr unsigned kRowBatch = 8;
        for (unsigned j0 = 0; j0 < rows; j0 += kRowBatch)
        {
            int dvs[kRowBatch];
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
                dvs[q] = col[min(j0 + q, (unsigned)kDenseCap - 1u) * 64];
            PD_ALL_LOADS_DONE(dvs);
            double dels[kRowBatch], noDels[kRowBatch];
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
            {
                const bool use = j0 + q < nn;
                bool inD, inN;
                const uint64_t pd_ = lanePosClamped(l, dvs[q] - len, inD), pn = lanePosClamped(l, dvs[q] - refShift, inN);
                dels[q] = l.minProb;
                noDels[q] = l.minProb;
                if (use && inD)
                    dels[q] = co.hist[pd_];
                if (use && inN)
                    noDels[q] = co.hist[pn];
            }
#pragma unroll
            for (unsigned q = 0; q < kRowBatch; ++q)
            {
                const bool use = j0 + q < nn;
                const int dv = dvs[q];
                const double del = dels[q], noDel = noDels[q];
                const double probDel = ea1 * del / (del + noDel) + ea2;
                const double probRef = ea1Rg * noDel / (del + noDel) + ea0Rg;
                sumDel += use ? probDel : 0.0;
                sumRef += use ? probRef : 0.0;
                wSumDel += use ? probDel * dv : 0.0;
                wSumRef += use ? probRef * dv : 0.0;
            }
        }
        if (valid && !highCov)
        {
            const int newShift = referenceShiftOf(wSumRef, sumRef, rgNan, l.stddev, nearAcc);
            *shiftSlot = newShift;
            sumDelAcc += sumDel;
            wSumDelAcc += wSumDel;
        }
    }
    const double t0 = blockSum(sumDelAcc, red);
    const double t1 = blockSum(wSumDelAcc, red);
    const double t3 = blockSum(nearAcc, red);
    if (threadIdx.x == 0)
    {
        double * p = tileSlot(b, cellIdx, b.rank * b.tilesPerRank + tileIdx);
        p[0] = t0;
        p[1] = t1;
        p[3] = t3;          // (see updateLenKernel)
    }
}

__global__ __launch_bounds__(kSampleTile) void finalMultiKernel(CohortView co, Stage2Buffers b, const uint8_t * __restrict__ flags,
                                                                unsigned nWindows, unsigned firstPos, unsigned shift)
{
    if (blockIdx.x >= b.mt.n)           // grid.x is padded to a multiple of 8 (runStage2Main)
        return;
    __shared__ unsigned shBorder[kSamplesPerBlock];        // per sample: its last read group below maxLoad, or ~0
    const unsigned cellIdx = blockIdx.y, mtI = blockIdx.x, tileIdx = b.mt.tile[mtI];
    const unsigned imgIdx = b.cells[cellIdx].image;
    CellState & cs = b.cells[cellIdx];
    if (cs.phase != kCellFinal || b.tileMode[(size_t)imgIdx * b.ownTiles + tileIdx] != kTileMulti)
        return;
    const unsigned w = (unsigned)cs.window;
    const unsigned stBase = firstPos + w * shift;
    const int len = (int)cs.len;
    const unsigned s = tileIdx * kSamplesPerBlock + threadIdx.x;
    // phase 0: the supporting-pair borders of a sample come from its LAST usable read group (quirk Q7)
    {
        unsigned border = 0xffffffffu;
        if (s < co.nSamples)
            for (unsigned rg = co.sampleRg[s]; rg < co.sampleRg[s + 1]; ++rg)
                if (b.mN[(size_t)imgIdx * b.nRg + rg] != kDenseHighCov)
                    border = rg;
        shBorder[threadIdx.x] = border;
    }
    __syncthreads();
    const unsigned rgLo = b.mt.rgLo[mtI], rgHi = b.mt.rgHi[mtI];
    // phase 1: one lane per read group
    for (unsigned base = rgLo; base < rgHi; base += kSampleTile)
    {
        const unsigned rg = base + threadIdx.x;
        const bool valid = rg < rgHi;
        const size_t wtGlobal = (size_t)imgIdx * b.mt.nWaves + b.mt.waveBase[mtI] + ((base - rgLo) >> 6) + (threadIdx.x >> 6);
        const unsigned rows = (base - rgLo) + (threadIdx.x & ~63u) < rgHi - rgLo ? b.mWaveRows[wtGlobal] : 0u;
        const int16_t * col = b.mDev + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
        const int16_t * colSt = b.mStart + wtGlobal * (kDenseCap * 64) + (threadIdx.x & 63u);
        unsigned n = 0;
        RgConst rc = {};
        int refShift = 0;
        int delLower = 0x7fffffff, delUpper = 0;
        if (valid)
        {
            rc = co.rg[rg];
            n = b.mN[(size_t)imgIdx * b.nRg + rg];
            refShift = cs.restoreZeroShifts ? 0 : b.shifts[(size_t)cellIdx * co.nRg + rg];
            const unsigned border = shBorder[b.mt.rgSample[rg] - tileIdx * kSamplesPerBlock];
            if (border != 0xffffffffu)
            {
                const RgConst & rb = co.rg[border];
                delLower = (int)((unsigned)len - (unsigned)rb.lowerQ);
                delUpper = (int)((unsigned)len + (unsigned)rb.upperQ);
            }
        }
        const bool highCov = n == kDenseHighCov;
        const unsigned nn = highCov ? 0u : n;
        const int rgLower = (int)((unsigned)len - (unsigned)rc.lowerQ);
        const int rgUpper = (int)((unsigned)len + (unsigned)rc.upperQ);
        double l0 = 0, l2 = 0, t0 = 0, t2 = 0, prod = 1.0;
        int esum = 0;
        bool eq = true;
        unsigned mn = 0xffffffffu, mx = 0, nSupp = 0;
        unsigned lad[3] = {0, 0, 0}, dad[5] = {0, 0, 0, 0, 0};
#pragma unroll 2
        for (unsigned j = 0; j < rows; ++j)
        {
            const bool use = j < nn;
            const int dv = col[j * 64];
            const unsigned st = stBase + (unsigned)(int)colSt[j * 64];
            int cls;
            if (dv > rc.upperQ)
                cls = (dv < rgLower) ? 2 : ((dv <= rgUpper) ? 3 : 4);
            else
                cls = (dv < rgUpper) ? 0 : 1;
            const long long pr = lookupPos(rc, dv - refShift), pdl = lookupPos(rc, dv - len);
            double4 tr = make_double4(rc.minProb, rc.logMinProb, rc.log10MinProb, 0.0), td = tr;
            if (use && pr >= 0)
                tr = co.tab[pr];
            if (use && pdl >= 0)
                td = co.tab[pdl];
            const double ref = tr.x, del = td.x;
            const int lcls = (ref >= 2 * del) ? 0 : ((del >= 2 * ref) ? 2 : 1);
            int e;
            const double m = frexp(ref + del, &e);
            l0 += use ? tr.y : 0.0;
            t0 += use ? tr.z : 0.0;
            l2 += use ? td.y : 0.0;
            t2 += use ? td.z : 0.0;
            prod *= use ? m : 1.0;
            esum += use ? e : 0;
            eq = eq && (!use || ref == del);
            const unsigned last = st + (unsigned)max(0, dv + rc.median - 2 * rc.readLength);
            mn = use ? min(mn, st) : mn;
            mx = use ? max(mx, last) : mx;
#pragma unroll
            for (int c = 0; c < 5; ++c)
                dad[c] += (use && cls == c) ? 1u : 0u;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                lad[c] += (use && lcls == c) ? 1u : 0u;
            nSupp += (use && dv >= delLower && dv <= delUpper) ? 1u : 0u;
        }
        // supporting pairs: one pool allocation per wave, lanes fill their own stretch
        {
            unsigned incl = nSupp;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
            {
                const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
                if ((int)(threadIdx.x & 63u) >= o)
                    incl += t;
            }
            const unsigned waveTotal = (unsigned)__shfl((int)incl, 63, 64);
            if (waveTotal)                                        // wave-uniform
            {
                unsigned base0 = 0;
                if ((threadIdx.x & 63u) == 0)
                    base0 = atomicAdd(&cs.suppCount, waveTotal);
                base0 = (unsigned)__shfl((int)base0, 0, 64);
                unsigned at = base0 + incl - nSupp;
                if (nSupp)
                    for (unsigned j = 0; j < nn; ++j)
                    {
                        const int dv = col[j * 64];
                        if (dv >= delLower && dv <= delUpper)
                        {
                            const unsigned st = stBase + (unsigned)(int)colSt[j * 64];
                            b.suppFirst[cs.suppBase + at] = st;
                            b.suppLast[cs.suppBase + at] = st + (unsigned)max(0, dv + rc.median - 2 * rc.readLength);
                            ++at;
                        }
                    }
            }
        }
        if (valid)
        {
            double * od = b.rgOutD + ((size_t)cellIdx * b.nRg + rg) * 8;
            unsigned * ou = b.rgOutU + ((size_t)cellIdx * b.nRg + rg) * 12;
            od[0] = l0;
            od[1] = nn ? log(prod) + (double)(esum - (int)nn) * PD_LN2 : 0.0;
            od[2] = l2;
            od[3] = t0;
            od[4] = nn ? log10(prod) + (double)(esum - (int)nn) * PD_LOG10_2 : 0.0;
            od[5] = t2;
            ou[0] = nn | (highCov ? PD_RGF_HIGHCOV : 0u) | (eq ? PD_RGF_EQ : 0u);
            for (int c = 0; c < 3; ++c)
                ou[1 + c] = lad[c];
            for (int c = 0; c < 5; ++c)
                ou[4 + c] = dad[c];
            ou[9] = mn;
            ou[10] = mx;
        }
    }
    __syncthreads();
    // phase 2: one lane per sample
    if (s >= co.nSamples)
        return;
    double logL[3] = {0, 0, 0}, gl[3] = {0, 0, 0};
    unsigned lad[3] = {0, 0, 0}, dad[5] = {0, 0, 0, 0, 0};
    unsigned nUsed = 0, mn = 0xffffffffu, mx = 0;
    bool allEq = true;
    for (unsigned rg = co.sampleRg[s]; rg < co.sampleRg[s + 1]; ++rg)
    {
        const unsigned * ou = b.rgOutU + ((size_t)cellIdx * b.nRg + rg) * 12;
        if (ou[0] & PD_RGF_HIGHCOV)
            continue;
        const double * od = b.rgOutD + ((size_t)cellIdx * b.nRg + rg) * 8;
        for (int k = 0; k < 3; ++k)
        {
            logL[k] += od[k];
            gl[k] += od[3 + k];
            lad[k] += ou[1 + k];
        }
        for (int k = 0; k < 5; ++k)
            dad[k] += ou[4 + k];
        nUsed += ou[0] & 0xffffu;
        allEq = allEq && (ou[0] & PD_RGF_EQ) != 0u;
        mn = min(mn, ou[9]);
        mx = max(mx, ou[10]);
    }
    if (mn == 0xffffffffu)
        mn = 0;
    double d[3];
    if (nUsed == 0)
    {
        // sum(gtLogs) == 0: reference-like triple, gtLogs stay untouched (all zero)
        d[0] = 0;
        d[1] = PD_LOG_PSEUDO;
        d[2] = PD_LOG_PSEUDO;
        gl[0] = gl[1] = gl[2] = 0;
    }
    else
    {
        const double maxGt = fmax(fmax(gl[0], gl[1]), gl[2]);
        gl[0] -= maxGt;
        gl[1] -= maxGt;
        gl[2] -= maxGt;
        if (allEq)
        {
            gl[0] = gl[1] = gl[2] = 0;      // x87: a hair apart, not equal; all phred values round to 0
        }
        else if (gl[0] == gl[1] && gl[0] == gl[2])
        {
            gl[0] = 0;
            gl[1] = -10;
            gl[2] = -10;
        }
        finishTriple(d, logL, nUsed, allEq);
    }
    const size_t o = (size_t)cellIdx * co.nSamples + s;
    b.dl[o * 3 + 0] = d[0];
    b.dl[o * 3 + 1] = d[1];
    b.dl[o * 3 + 2] = d[2];
    uint32_t pl[3];
    {
        const double gTot = log10(exp(gl[0]) + exp(gl[1]) + exp(gl[2]));
        const double p0 = -10 * (gl[0] - gTot);
        const double p1 = -10 * (gl[1] - gTot);
        const double p2 = -10 * (gl[2] - gTot);
        const double mnP = fmin(fmin(p0, p1), p2);
        pl[0] = (uint32_t)round(p0 - mnP);
        pl[1] = (uint32_t)round(p1 - mnP);
        pl[2] = (uint32_t)round(p2 - mnP);
    }
    if (flags[(size_t)s * nWindows + w] & 1u)
        pl[0] = pl[1] = pl[2] = 0;
    for (int c = 0; c < 3; ++c)
    {
        b.pl[o * 3 + c] = pl[c];
        b.lad[o * 3 + c] = lad[c];
    }
    for (int c = 0; c < 5; ++c)
        b.dad[o * 5 + c] = dad[c];
    b.fl[o * 2 + 0] = mn;
    b.fl[o * 2 + 1] = mx;
}

// Percentiles of the support pools (reference :514-529) by 8-bit radix selection. The pool of a cell is spread over the
// contexts of a sharded cohort, so every pass is: digit histogram of the own pool (suppHistKernel) -> exchange -> sum of
// the histograms, pick the digit (suppPickKernel). Pool 0: pair starts (80th percentile), pool 1: pair ends (20th).
// Two flavours. Exact: four passes over the raw values. Range-limited (Stage2Buffers::rangeSelect, the rule): the pool
// holds positions of one deletion's supporting pairs, a few thousand bases wide, so the contexts first exchange
// {min, max, count} of their pools (suppMinMaxKernel; the record rides in the xch area together with the likelihood-ratio
// sums of the final estimate), and the selection runs on x - min over the two low bytes only: three exchanges instead of
// five. A pool that spans 2^16 positions or more raises the guard and the batch is repeated the exact way.
__global__ __launch_bounds__(256) void suppMinMaxKernel(Stage2Buffers b)
{
    __shared__ unsigned shLo[4], shHi[4];
    const unsigned c = blockIdx.x, pool = blockIdx.y;
    const CellState & cs = b.cells[c];
    unsigned * out = poolSlot(b, b.rank, c, pool);
    const unsigned n = cs.phase == kCellFinal ? cs.suppCount : 0u;
    unsigned lo = 0xffffffffu, hi = 0;
    const unsigned * v = (pool == 0 ? b.suppFirst : b.suppLast) + cs.suppBase;
    for (unsigned i0 = threadIdx.x; i0 < n; i0 += 8u * blockDim.x)
    {
        unsigned x[8];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u)
            x[u] = (i0 + u * blockDim.x < n) ? v[i0 + u * blockDim.x] : v[i0];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u)
        {
            lo = min(lo, x[u]);
            hi = max(hi, x[u]);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    }
    if ((threadIdx.x & 63u) == 0)
    {
        shLo[threadIdx.x >> 6] = lo;
        shHi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        out[0] = min(min(shLo[0], shLo[1]), min(shLo[2], shLo[3]));
        out[1] = max(max(shHi[0], shHi[1]), max(shHi[2], shHi[3]));
        out[2] = n;
        out[3] = 0;
    }
}

// After the exchange of the pool records: cohort-wide minimum, count and rank of the percentile per pool.
__global__ void suppRangeKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.nCells)
        return;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellFinal)
        return;
    for (unsigned pool = 0; pool < 2; ++pool)
    {
        unsigned lo = 0xffffffffu, hi = 0, n = 0;
        for (unsigned r = 0; r < b.world; ++r)
        {
            const unsigned * p = poolSlot(b, r, c, pool);
            if (p[2] == 0)
                continue;
            lo = min(lo, p[0]);
            hi = max(hi, p[1]);
            n += p[2];
        }
        if (pool == 0)
            cs.suppCapacity = n;                           // cohort-wide number of supporting pairs (parked here)
        cs.selPrefix[pool] = 0;
        cs.selLo[pool] = n ? lo : 0u;
        cs.selK[pool] = n ? (unsigned)round((double)(n - 1) * (pool == 0 ? 0.8 : (1 - 0.8))) : 0u;
        if (n != 0 && hi - lo >= 65536u && b.guard != nullptr)
            *b.guard = 1u;                                 // (every context sees the same records and does the same)
    }
}

__global__ __launch_bounds__(256) void suppHistKernel(Stage2Buffers b, int pass)
{
    __shared__ unsigned hist[256];
    const unsigned c = blockIdx.x, pool = blockIdx.y;
    const CellState & cs = b.cells[c];
    if (cs.phase != kCellFinal)
        return;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned * v = (pool == 0 ? b.suppFirst : b.suppLast) + cs.suppBase;
    const unsigned n = cs.suppCount;                       // own pool; replaced by the cohort-wide count by suppFinishKernel
    const int shift = pass * 8;
    const unsigned mask = pass == 3 ? 0u : (0xffffffffu << (shift + 8)), prefix = cs.selPrefix[pool], lo = cs.selLo[pool];
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
    {
        const unsigned x = v[i] - lo;
        if ((x & mask) == prefix)
            atomicAdd(&hist[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    b.selHist[(((size_t)b.rank * b.nCells + c) * 2 + pool) * 256 + threadIdx.x] = hist[threadIdx.x];
}

// `first`: this is the first pass of the exact flavour -- the histogram also yields the cohort-wide count and with it the
// rank of the percentile (the range-limited flavour has both from suppRangeKernel).
__global__ __launch_bounds__(256) void suppPickKernel(Stage2Buffers b, int pass, int first)
{
    __shared__ unsigned hist[256];
    const unsigned c = blockIdx.x, pool = blockIdx.y;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellFinal)
        return;
    unsigned h = 0;
    for (unsigned r = 0; r < b.world; ++r)
        h += b.selHist[(((size_t)r * b.nCells + c) * 2 + pool) * 256 + threadIdx.x];
    hist[threadIdx.x] = h;
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    unsigned k = cs.selK[pool];
    if (first)
    {
        unsigned n = 0;
        for (int d = 0; d < 256; ++d)
            n += hist[d];
        if (pool == 0)
            cs.suppCapacity = n;                           // cohort-wide number of supporting pairs (parked here)
        if (n == 0)
        {
            cs.selPrefix[pool] = 0;
            cs.selK[pool] = 0;
            return;
        }
        k = (unsigned)round((double)(n - 1) * (pool == 0 ? 0.8 : (1 - 0.8)));
    }
    else if (cs.suppCapacity == 0)
    {
        return;
    }
    unsigned acc = 0, digit = 0;
    for (digit = 0; digit < 256; ++digit)
    {
        if (acc + hist[digit] > k)
            break;
        acc += hist[digit];
    }
    cs.selPrefix[pool] |= digit << (pass * 8);
    cs.selK[pool] = k - acc;
}

// The same selection in ONE launch when the cohort is not sharded (the whole pool is local): the passes run inside the
// workgroup and the percentiles are published directly (what suppFinishKernel does after the exchanged passes). The pool
// holds positions of one deletion's supporting pairs -- a few thousand bases wide, every upper byte identical -- so the
// selection works on x - min(pool) and only on the bytes the range max - min needs (two, normally): the passes over the
// constant upper bytes were the expensive ones (all 256 lanes adding to ONE LDS counter). One histogram per wave.
__global__ __launch_bounds__(256) void suppSelectLocalKernel(Stage2Buffers b)
{
    __shared__ unsigned hist[4][256];
    __shared__ unsigned shPrefix, shK, waveTot[4], shLo[4], shHi[4];
    const unsigned c = blockIdx.x, pool = blockIdx.y;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellFinal)
        return;
    const unsigned * v = (pool == 0 ? b.suppFirst : b.suppLast) + cs.suppBase;
    const unsigned n = cs.suppCount;
    const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (n == 0)
    {
        if (threadIdx.x == 0)
            (pool == 0 ? cs.suppStart : cs.suppEnd) = 0u;
        return;
    }
    unsigned lo = 0xffffffffu, hi = 0;
    // (eight loads in flight per lane: with one, the walks over the pool are a chain of L2 latencies)
    for (unsigned i0 = threadIdx.x; i0 < n; i0 += 8u * blockDim.x)
    {
        unsigned x[8];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u)
            x[u] = (i0 + u * blockDim.x < n) ? v[i0 + u * blockDim.x] : v[i0];
#pragma unroll
        for (unsigned u = 0; u < 8; ++u)
        {
            lo = min(lo, x[u]);
            hi = max(hi, x[u]);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
    {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o, 64));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o, 64));
    }
    if (lane == 0)
    {
        shLo[wv] = lo;
        shHi[wv] = hi;
    }
    if (threadIdx.x == 0)
    {
        shPrefix = 0;
        shK = (unsigned)round((double)(n - 1) * (pool == 0 ? 0.8 : (1 - 0.8)));
    }
    __syncthreads();
    lo = min(min(shLo[0], shLo[1]), min(shLo[2], shLo[3]));
    hi = max(max(shHi[0], shHi[1]), max(shHi[2], shHi[3]));
    const unsigned range = hi - lo;
    const int topPass = range >= (1u << 24) ? 3 : (range >= (1u << 16) ? 2 : (range >= (1u << 8) ? 1 : 0));
    for (int pass = topPass; pass >= 0; --pass)
    {
        for (int q = 0; q < 4; ++q)
            hist[q][threadIdx.x] = 0;
        __syncthreads();
        const int shift = pass * 8;
        const unsigned mask = pass == 3 ? 0u : (0xffffffffu << (shift + 8)), prefix = shPrefix;
        for (unsigned i0 = threadIdx.x; i0 < n; i0 += 8u * blockDim.x)
        {
            unsigned y[8];
#pragma unroll
            for (unsigned u = 0; u < 8; ++u)
                y[u] = (i0 + u * blockDim.x < n) ? v[i0 + u * blockDim.x] - lo : 0xffffffffu;
#pragma unroll
            for (unsigned u = 0; u < 8; ++u)
                if (i0 + u * blockDim.x < n && (y[u] & mask) == prefix)
                    atomicAdd(&hist[wv][(y[u] >> shift) & 255u], 1u);
        }
        __syncthreads();
        // first digit whose inclusive prefix count exceeds k: block-wide inclusive scan, one lane per digit
        const unsigned h = hist[0][threadIdx.x] + hist[1][threadIdx.x] + hist[2][threadIdx.x] + hist[3][threadIdx.x];
        const unsigned k = shK;
        unsigned incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= (unsigned)o)
                incl += up;
        }
        if (lane == 63u)
            waveTot[wv] = incl;
        __syncthreads();
        for (unsigned q = 0; q < wv; ++q)
            incl += waveTot[q];
        const unsigned digit = (unsigned)__syncthreads_count(incl <= k);       // < 256: the total exceeds k
        if (threadIdx.x == digit)
        {
            shPrefix = prefix | (digit << shift);
            shK = k - (incl - h);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        (pool == 0 ? cs.suppStart : cs.suppEnd) = lo + shPrefix;
}

// After the last pass: publish the percentiles and the cohort-wide support count.
__global__ void suppFinishKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.nCells)
        return;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellFinal)
        return;
    cs.suppCount = cs.suppCapacity;
    cs.suppStart = cs.suppCount ? cs.selLo[0] + cs.selPrefix[0] : 0u;
    cs.suppEnd = cs.suppCount ? cs.selLo[1] + cs.selPrefix[1] : 0u;
}

// Likelihood-ratio test and call emission (reference :685-724).
__global__ __launch_bounds__(64) void cellEmitKernel(Stage2Buffers b)
{
    const unsigned c = blockIdx.x;
    CellState & cs = b.cells[c];
    if (cs.phase != kCellFinal)
        return;
    const bool noSupport = cs.suppCount == 0 || (cs.suppStart == 0 && cs.suppEnd == 0);      // wave-uniform
    const double lr = noSupport ? 0.0 : sumLR(b, c);
    if (threadIdx.x != 0)
        return;
    if (noSupport)
    {
        cs.phase = kCellDead;
        return;
    }
    cs.logLR = lr;
    cs.phase = (cs.logLR >= b.minLR) ? kCellEmit : kCellDead;
}

// The emitted cells of a batch in cell order (one workgroup): emitCells[0] = their number, emitCells[1..] = their indices.
__global__ __launch_bounds__(256) void emitCompactKernel(Stage2Buffers b, unsigned * __restrict__ emitCells)
{
    __shared__ unsigned sh[256];
    unsigned carry = 0;
    for (unsigned base = 0; base < b.nCells; base += 256u)
    {
        const unsigned c = base + threadIdx.x;
        const unsigned v = (c < b.nCells && b.cells[c].phase == kCellEmit) ? 1u : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 256u; o <<= 1)
        {
            const unsigned t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (v)
            emitCells[carry + sh[threadIdx.x]] = c;            // (inclusive count: slot 0 is the total)
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        emitCells[0] = carry;
}

// Pack the per-sample outputs of the emitted cells: [n][S][3] PL | [n][S][3] LAD | [n][S][5] DAD | [n][S][2] first/last,
// n = emitCells[0]. Launched for every cell of the batch before the host knows n; the surplus workgroups leave at once.
__global__ void gatherOutKernel(Stage2Buffers b, const unsigned * __restrict__ emitCells, unsigned nSamples,
                                uint32_t * __restrict__ out)
{
    const unsigned e = blockIdx.y;
    const unsigned n = emitCells[0];
    if (e >= n)
        return;
    const unsigned c = emitCells[1 + e];
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nSamples)
        return;
    const size_t per = (size_t)n * nSamples;
    uint32_t * pl = out;
    uint32_t * lad = pl + per * 3;
    uint32_t * dad = lad + per * 3;
    uint32_t * fl = dad + per * 5;
    const size_t src = (size_t)c * nSamples + s, dst = (size_t)e * nSamples + s;
    for (int k = 0; k < 3; ++k)
    {
        pl[dst * 3 + k] = b.pl[src * 3 + k];
        lad[dst * 3 + k] = b.lad[src * 3 + k];
    }
    for (int k = 0; k < 5; ++k)
        dad[dst * 5 + k] = b.dad[src * 5 + k];
    fl[dst * 2] = b.fl[src * 2];
    fl[dst * 2 + 1] = b.fl[src * 2 + 1];
}

// ---- host-side launch sequence ---------------------------------------------------------------------------------------
void launchPackFlagged(const int32_t * q3, const uint8_t * flags, const int * flaggedWindows, unsigned W, unsigned nLocal,
                       const ClusterInput & in, unsigned rank, hipStream_t st)
{
    const unsigned rows = in.nFlaggedDev ? (in.nFlagged < 64u ? in.nFlagged : 64u) : in.nFlagged;
    hipLaunchKernelGGL(packFlaggedKernel, dim3((in.sper + 255) / 256, rows), dim3(256), 0, st, q3, flags, flaggedWindows, W,
                       nLocal, in, rank);
}

void launchCompactFlagged(const int * winFlag, unsigned W, unsigned world, size_t rowStride, unsigned cap, int * flagged,
                          unsigned * hdr, hipStream_t st)
{
    hipLaunchKernelGGL(compactFlaggedKernel, dim3(1), dim3(1024), 0, st, winFlag, W, world, rowStride, cap, flagged, hdr);
}

void launchListCells(const int * flagged, const int * candLen, const int * candCount, unsigned cap, unsigned listCap,
                     unsigned * hdr, hipStream_t st)
{
    hipLaunchKernelGGL(listCellsKernel, dim3(1), dim3(256), 0, st, flagged, candLen, candCount, cap, listCap, hdr);
}

__global__ __launch_bounds__(256) void zeroRangesKernel(ZeroRanges r)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (unsigned k = 0; k < r.n; ++k)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.words[k]; i += stride)
            r.p[k][i] = 0u;
}

void launchZero(const ZeroRanges & r, hipStream_t st)
{
    size_t most = 0;
    for (unsigned k = 0; k < r.n; ++k)
        most = most > r.words[k] ? most : r.words[k];
    if (most == 0)
        return;
    const unsigned groups = (unsigned)((most + 1023) / 1024 < 1024 ? (most + 1023) / 1024 : 1024);
    hipLaunchKernelGGL(zeroRangesKernel, dim3(groups), dim3(256), 0, st, r);
}

__global__ void copyBytesKernel(const uint8_t * __restrict__ src, uint8_t * __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// A plain copy kernel: hipMemcpyAsync(DeviceToDevice) in the middle of a kernel sequence was measured to cost ~0.1 ms.
void launchCopyBytes(const void * src, void * dst, size_t n, hipStream_t st)
{
    if (n)
        hipLaunchKernelGGL(copyBytesKernel, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, st,
                           static_cast<const uint8_t *>(src), static_cast<uint8_t *>(dst), n);
}

// One wave that does nothing for `ticks` of the constant-rate wall clock: stands in for the wire time of an exchange when
// the sample-sharded path is measured with one rank (POPDEL_HIP_XCH_DELAY_US; bench.py's latency model).
__global__ void spinKernel(unsigned long long ticks)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(8);
}

void launchSpin(unsigned long long ticks, hipStream_t st)
{
    hipLaunchKernelGGL(spinKernel, dim3(1), dim3(64), 0, st, ticks);
}

void launchCluster(const ClusterInput & in, int uniformThreshold, int * scratch, unsigned rowLen, int * candLen, int * candCount,
                   hipStream_t st)
{
    const unsigned groups = in.nFlaggedDev ? (in.nFlagged < 256u ? in.nFlagged : 256u) : in.nFlagged;
    hipLaunchKernelGGL(clusterKernel, dim3(groups), dim3(1024), 0, st, in, uniformThreshold, scratch, rowLen, candLen,
                       candCount);
}

// Phase A: the dense images of the cells' active pairs and the integer counts (own samples) in the context's xch chunk.
// The caller has cleared the xch area, shifts, tileMode and iterFlags[0 .. maxIterations + 1] (one launchZero).
void runStage2Init(const SegmentView & seg, const CohortView & co, Stage2Buffers & b, const uint8_t * flags, hipStream_t st)
{
    if (b.nCells == 0)
        return;
    hipLaunchKernelGGL(cellGatherKernel, dim3(b.ownTiles * 4, b.nImages), dim3(kGatherThreads), 0, st, seg, co, b, flags);
    if (b.mt.n != 0)
        hipLaunchKernelGGL(multiGatherKernel, dim3(b.mt.nWaves, b.nImages), dim3(kGatherThreads), 0, st, seg, co, b);
}

// Phase B: first allele-frequency estimate, EM loop, convergence comparison, final pass, support percentiles,
// likelihood-ratio test. Every context of a sharded cohort runs the same sequence; the per-cell kernels see identical
// inputs everywhere (all-gathered per-tile sums), so all contexts take the same decisions, including the early exit.
// Exchanges of a sharded cohort (the xch area travels as a whole each time): counts + read group 0's first record | two
// per EM iteration | both likelihood-ratio evaluations of the converged cells | pool records + the final estimate's
// likelihood-ratio sums | two digit histograms (four, without the pool records' range, in the exact flavour).
void runStage2Main(const SegmentView & seg, const CohortView & co, Stage2Buffers & b, const uint8_t * flags, Exchange & x,
                   bool anyGenericTile, unsigned * pinnedWord, hipEvent_t likBegin, hipEvent_t likEnd, hipStream_t st)
{
    if (b.nCells == 0)
        return;
    const unsigned ownTiles = (co.nSamples + kSamplesPerBlock - 1) / kSamplesPerBlock;
    // Workgroups go to the 8 XCDs round-robin by their linear index (x fastest): with grid.x a multiple of 8 an XCD always
    // sees the same sample tiles, whatever the cell, and its L2 (4 MB) keeps those samples' lookup tables -- with any
    // other width every XCD sees every tile (48 MB of tables for 10,000 samples). Measured: 9,984 samples (39 tiles)
    // 118 / 97 / 232 us for dataLikDense / updateLenDense / finalDense against 64 / 65 / 174 us with 10,000 (40 tiles).
    // The surplus workgroups leave at once.
    // Small cohorts are left as they are (measured: 10 tiles -3 % with the padding, 13 tiles equal, 17 tiles +3.5 %,
    // 21 tiles +9 %, 25 tiles +13 %, 29 tiles +24 %, 39 tiles +21 %: their tables fit the caches either way and the surplus
    // workgroups cost more than they save).
    const unsigned padded = ownTiles <= 12u ? ownTiles : (ownTiles + 7u) / 8u * 8u;
    const unsigned sweeps = (padded + kTilesPerSweep - 1u) / kTilesPerSweep;         // (the host keeps nCells x sweeps <= 65,535)
    const dim3 perSample(padded < kTilesPerSweep ? padded : kTilesPerSweep, b.nCells * sweeps), tile(kSampleTile);
    const dim3 perCell((b.nCells + 127) / 128), cellBlock(128), wavePerCell(b.nCells);
    const size_t xchBytes = b.xchChunk * sizeof(double);
    // every (cell, tile) is served by exactly one of the two kernels of a pair (tileMode)
    const dim3 perMulti(b.mt.n <= 12u ? b.mt.n : (b.mt.n + 7u) / 8u * 8u, b.nCells);   // (same XCD-stable tile order)
    const dim3 perImage(perSample.x, b.nImages * sweeps);
    auto dataLik = [&](int mode) {
        if (mode != 0)      // zero reference shifts: the cells of a window share their reference-allele lookups
            hipLaunchKernelGGL(dataLikZeroShiftDenseKernel, perImage, tile, 0, st, co, b, mode);
        else
            hipLaunchKernelGGL(dataLikDenseKernel, perSample, tile, 0, st, co, b, mode);
        if (b.mt.n != 0)
            hipLaunchKernelGGL(dataLikMultiKernel, perMulti, tile, 0, st, co, b, mode);
        if (anyGenericTile)
            hipLaunchKernelGGL(dataLikKernel, perSample, tile, 0, st, seg, co, b, mode);
    };
    // pool bases (and the check of an optimistic launch) from the own counts; the likelihoods ahead of the loop do not
    // depend on the first frequency estimate: they run before the counts are exchanged, and read group 0's record
    // travels with them
    hipLaunchKernelGGL(suppScanKernel, dim3(1), dim3(256), 0, st, b);
    // (timed: this launch of the data-likelihood kernels takes EVERY cell of the batch x every sample -- a known number of
    // units, which the launches inside the loop, over the cells still iterating, do not have: pd_timing::lik_first_ms)
    (void)hipEventRecord(likBegin, st);
    dataLik(2);
    (void)hipEventRecord(likEnd, st);
    x.allGather(b.xch, xchBytes);
    hipLaunchKernelGGL(cellInitFinishKernel, perCell, cellBlock, 0, st, b);
    for (unsigned it = 0; it < b.maxIterations; ++it)
    {
        hipLaunchKernelGGL(updateLenDenseKernel, perSample, tile, 0, st, co, b);
        if (b.mt.n != 0)
            hipLaunchKernelGGL(updateLenMultiKernel, perMulti, tile, 0, st, co, b);
        if (anyGenericTile)
            hipLaunchKernelGGL(updateLenKernel, perSample, tile, 0, st, seg, co, b);
        x.allGather(b.xch, xchBytes);
        hipLaunchKernelGGL(cellLenFinishKernel, wavePerCell, dim3(64), 0, st, b);
        dataLik(0);
        x.allGather(b.xch, xchBytes);
        hipLaunchKernelGGL(cellFreqFinishKernel, wavePerCell, dim3(64), 0, st, b, it);
        // Most cells leave the loop after 2-4 iterations: look at the "someone is still iterating" flag now and then
        // instead of launching the remaining (empty) iterations (a look costs a round trip of ~50 us, an empty iteration ~20
        // -- plus two exchanges in a sharded cohort, which therefore looks after every iteration from the second on).
        if (it == 3 || it == 5 || it == 8 || it == 11 || (b.sharded && it >= 1))
        {
            *pinnedWord = 1;                               // page-locked: no bounce copy for four bytes
            (void)hipMemcpyAsync(pinnedWord, b.iterFlags + it, sizeof(unsigned), hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            if (!*pinnedWord)
                break;
        }
    }
    // cells that left through the visited-table test: compare with the previous estimate
    hipLaunchKernelGGL(lrKernel, perSample, tile, 0, st, co, b, (int)kCellConverged, 0);
    dataLik(1);
    hipLaunchKernelGGL(lrKernel, perSample, tile, 0, st, co, b, (int)kCellConverged, 1);
    x.allGather(b.xch, xchBytes);
    hipLaunchKernelGGL(cellConvergedKernel, wavePerCell, dim3(64), 0, st, b);
    hipLaunchKernelGGL(cellPostLoopKernel, perCell, cellBlock, 0, st, b);
    if (static_cast<uint64_t>(b.nCells) * co.nSamples > 2000000ull)
        hipLaunchKernelGGL(finalDenseKernel<true>, perSample, tile, 0, st, co, b, flags, seg.nWindows, seg.firstPos, seg.shift);
    else
        hipLaunchKernelGGL(finalDenseKernel<false>, perSample, tile, 0, st, co, b, flags, seg.nWindows, seg.firstPos, seg.shift);
    if (b.mt.n != 0)
        hipLaunchKernelGGL(finalMultiKernel, perMulti, tile, 0, st, co, b, flags, seg.nWindows, seg.firstPos, seg.shift);
    if (anyGenericTile)
        hipLaunchKernelGGL(finalKernel, perSample, tile, 0, st, seg, co, b, flags);
    // the likelihood-ratio sums of the final estimate need the final pass only: they are formed now and travel with the
    // first exchange of the percentile selection
    hipLaunchKernelGGL(lrKernel, perSample, tile, 0, st, co, b, (int)kCellFinal, 0);
    if (!b.sharded)
    {
        hipLaunchKernelGGL(suppSelectLocalKernel, dim3(b.nCells, 2), dim3(256), 0, st, b);
    }
    else
    {
        if (b.rangeSelect)
            hipLaunchKernelGGL(suppMinMaxKernel, dim3(b.nCells, 2), dim3(256), 0, st, b);
        x.allGather(b.xch, xchBytes);
        if (b.rangeSelect)
            hipLaunchKernelGGL(suppRangeKernel, perCell, cellBlock, 0, st, b);
        const int topPass = b.rangeSelect ? 1 : 3;
        for (int pass = topPass; pass >= 0; --pass)
        {
            hipLaunchKernelGGL(suppHistKernel, dim3(b.nCells, 2), dim3(256), 0, st, b, pass);
            x.allGather(b.selHist, (size_t)b.nCells * 2 * 256 * sizeof(unsigned));
            hipLaunchKernelGGL(suppPickKernel, dim3(b.nCells, 2), dim3(256), 0, st, b, pass, (!b.rangeSelect && pass == 3) ? 1 : 0);
        }
        hipLaunchKernelGGL(suppFinishKernel, perCell, cellBlock, 0, st, b);
    }
    hipLaunchKernelGGL(cellEmitKernel, wavePerCell, dim3(64), 0, st, b);
}

// emitCells: [1 + nCells] (count, then indices); out: room for nCells x nSamples x 13 words
void launchGatherOut(const Stage2Buffers & b, unsigned * emitCells, unsigned nSamples, uint32_t * out, hipStream_t st)
{
    if (b.nCells == 0)
        return;
    hipLaunchKernelGGL(emitCompactKernel, dim3(1), dim3(256), 0, st, b, emitCells);
    hipLaunchKernelGGL(gatherOutKernel, dim3((nSamples + 255) / 256, b.nCells), dim3(256), 0, st, b, emitCells, nSamples, out);
}

}  // namespace pd
