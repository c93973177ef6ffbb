// Stage 1 of the joint-calling hot path on gfx950: for every (sample, window) cell gather the insert-size deviations
// of the active read pairs and take their upper-half median (Q3).
//
// Reference behaviour: initialize_deletion_lengths() first half (genotype_deletion_popdel_call.h:33-57) with
// ChromosomeProfile::getActiveReadsDeviations() (profile_structure_popdel_call.h:1443-1478) and upperHalfMedian()
// (genotype_deletion_popdel_call.h:15-27).
//
// Mapping: one lane per cell, the 64 lanes of a wave are 64 CONSECUTIVE windows of ONE sample, a 256-thread
// workgroup covers 256 consecutive windows. Consecutive windows share most of their pairs, so the workgroup stages the
// union of their pair chunks ONCE from HBM into LDS (coalesced 8-byte loads); HBM traffic is the compact pair
// stream (8 B per pair chunk) instead of the expanded [sample x window x pair] tensor. Each lane then pulls its
// candidates from LDS into a register array and sorts it with a straight-line Batcher network (VALU min/max, no
// cross-lane traffic, no divergence); inactive slots hold +INF.
#include "pd_device.h"
#include "sortnet_gen.h"

namespace pd {

__device__ __forceinline__ int q3FromRankPair(unsigned n, int a, int b)
{
    // n >= 4: pos = (3n + 2 + n%2)/4 - 1 = l + k/4; value = floor((1-k/4)*a + (k/4)*b + 0.5)  (exact in integers)
    unsigned pos4 = 3u * n + 2u + (n & 1u) - 4u;
    long long k = pos4 & 3u;
    long long x = (4 - k) * (long long)a + k * (long long)b + 2;
    return (int)(x >> 2);
}

__device__ __forceinline__ void q3Ranks(unsigned n, unsigned & l0, unsigned & l1)
{
    if (n < 4)
    {
        l0 = n - 1;
        l1 = n - 1;
    }
    else
    {
        unsigned pos4 = 3u * n + 2u + (n & 1u) - 4u;
        l0 = pos4 >> 2;
        l1 = l0 + 1;
    }
}

// Generic cell evaluation straight from HBM: any number of read groups per sample, any number of candidates.
// Order statistics by rank counting (O(n^2)); used for multi-read-group samples and oversized tiles.
__device__ void stage1CellGeneric(const SegmentView & seg, const CohortView & co, unsigned s, unsigned w, int & q3Out,
                                  unsigned & flagsOut)
{
    const unsigned K = seg.maxLife;
    const unsigned wlo = (w + 1 >= K) ? (w + 1 - K) : 0u;
    unsigned total = 0, usable = 0;
    for (unsigned rg = co.sampleRg[s]; rg < co.sampleRg[s + 1]; ++rg)
    {
        const uint32_t * ws = seg.winStart + (size_t)rg * (seg.nWindows + 1);
        const PairDO * p = seg.pairs + seg.rgBase[rg];
        unsigned c = 0;
        for (unsigned i = ws[wlo]; i < ws[w + 1]; ++i)
            c += (p[i].off > w) ? 1u : 0u;
        total += c;
        if (c < co.rg[rg].maxLoad)
            usable += c;
    }
    q3Out = 0;
    flagsOut = 0;
    if (total < 2u)
    {
        flagsOut = 1u;
        return;
    }
    if (usable == 0)
        return;
    unsigned l0, l1;
    q3Ranks(usable, l0, l1);
    int a = 0, b = 0;
    // rank of every usable pair among all usable pairs (ties broken by enumeration order)
    unsigned outerIdx = 0;
    for (unsigned rg = co.sampleRg[s]; rg < co.sampleRg[s + 1]; ++rg)
    {
        const uint32_t * ws = seg.winStart + (size_t)rg * (seg.nWindows + 1);
        const PairDO * p = seg.pairs + seg.rgBase[rg];
        unsigned c = 0;
        for (unsigned i = ws[wlo]; i < ws[w + 1]; ++i)
            c += (p[i].off > w) ? 1u : 0u;
        if (c >= co.rg[rg].maxLoad)
            continue;
        for (unsigned i = ws[wlo]; i < ws[w + 1]; ++i)
        {
            if (!(p[i].off > w))
                continue;
            const int v = p[i].dev;
            unsigned rank = 0, innerIdx = 0;
            for (unsigned rg2 = co.sampleRg[s]; rg2 < co.sampleRg[s + 1]; ++rg2)
            {
                const uint32_t * ws2 = seg.winStart + (size_t)rg2 * (seg.nWindows + 1);
                const PairDO * p2 = seg.pairs + seg.rgBase[rg2];
                unsigned c2 = 0;
                for (unsigned j = ws2[wlo]; j < ws2[w + 1]; ++j)
                    c2 += (p2[j].off > w) ? 1u : 0u;
                if (c2 >= co.rg[rg2].maxLoad)
                    continue;
                for (unsigned j = ws2[wlo]; j < ws2[w + 1]; ++j)
                {
                    if (!(p2[j].off > w))
                        continue;
                    const int u = p2[j].dev;
                    rank += (u < v || (u == v && innerIdx < outerIdx)) ? 1u : 0u;
                    ++innerIdx;
                }
            }
            if (rank == l0)
                a = v;
            if (rank == l1)
                b = v;
            ++outerIdx;
        }
    }
    q3Out = (usable < 4) ? a : q3FromRankPair(usable, a, b);
    flagsOut = 2u;
}

#define PD_CE(i, j)                 \
    {                               \
        const int lo_ = min(r[i], r[j]); \
        const int hi_ = max(r[i], r[j]); \
        r[i] = lo_;                 \
        r[j] = hi_;                 \
    }

template <int C>
__device__ __forceinline__ void sortRegs(int (&r)[C]);
template <>
__device__ __forceinline__ void sortRegs<16>(int (&r)[16]) { PD_SORTNET_16(PD_CE) }
template <>
__device__ __forceinline__ void sortRegs<32>(int (&r)[32]) { PD_SORTNET_32(PD_CE) }
template <>
__device__ __forceinline__ void sortRegs<64>(int (&r)[64]) { PD_SORTNET_64(PD_CE) }
template <>
__device__ __forceinline__ void sortRegs<128>(int (&r)[128]) { PD_SORTNET_128(PD_CE) }

// Single read group, candidates already in LDS: registers + sorting network.
template <int C>
__device__ __forceinline__ void stage1CellFast(const PairDO * lds, unsigned a, unsigned cnt, unsigned w, unsigned maxLoad,
                                               int & q3Out, unsigned & flagsOut)
{
    int r[C];
    unsigned n = 0;
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
        const PairDO p = lds[a + c];
        const bool act = ((unsigned)c < cnt) && (p.off > w);
        r[c] = act ? p.dev : 0x7fffffff;
        n += act ? 1u : 0u;
    }
    q3Out = 0;
    flagsOut = 0;
    if (n < 2u)
    {
        flagsOut = 1u;
        return;
    }
    if (n >= maxLoad)
        return;                       // high coverage: counted, but contributes no value
    sortRegs<C>(r);
    unsigned l0, l1;
    q3Ranks(n, l0, l1);
    int va = 0, vb = 0;
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
        va = ((unsigned)c == l0) ? r[c] : va;
        vb = ((unsigned)c == l1) ? r[c] : vb;
    }
    q3Out = (n < 4) ? va : q3FromRankPair(n, va, vb);
    flagsOut = 2u;
}

__device__ __forceinline__ unsigned waveMaxU32(unsigned v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
}

__global__ __launch_bounds__(kStage1Block) void stage1Kernel(SegmentView seg, CohortView co, int32_t * __restrict__ q3,
                                                             uint8_t * __restrict__ flags, int * __restrict__ winFlag)
{
    __shared__ PairDO ldsPairs[kStage1LdsPairs + 128];
    __shared__ uint32_t ldsWs[kStage1Block + 64 + 2];

    const unsigned s = blockIdx.y;
    const unsigned w0 = blockIdx.x * kStage1Block;
    const unsigned W = seg.nWindows;
    const unsigned K = seg.maxLife;
    const unsigned tid = threadIdx.x;
    const unsigned w = w0 + tid;
    const unsigned rg0 = co.sampleRg[s];
    const bool singleRg = (co.sampleRg[s + 1] - rg0) == 1u;

    int q = 0;
    unsigned fl = 0;
    bool done = false;

    if (singleRg && K <= 64u)
    {
        const uint32_t * ws = seg.winStart + (size_t)rg0 * (W + 1);
        const unsigned wlo0 = (w0 + 1 >= K) ? (w0 + 1 - K) : 0u;        // first window whose chunks matter
        const unsigned whi = min(W, w0 + kStage1Block);                  // ws index of the range end
        const unsigned nWs = whi - wlo0 + 1;
        for (unsigned i = tid; i < nWs; i += kStage1Block)
            ldsWs[i] = ws[wlo0 + i];
        __syncthreads();
        const unsigned A = ldsWs[0];
        const unsigned nPairs = ldsWs[nWs - 1] - A;
        if (nPairs <= (unsigned)kStage1LdsPairs)                          // block-uniform
        {
            const PairDO * src = seg.pairs + seg.rgBase[rg0] + A;
            for (unsigned i = tid; i < nPairs; i += kStage1Block)
                ldsPairs[i] = src[i];
            __syncthreads();
            unsigned a = 0, cnt = 0;
            if (w < W)
            {
                const unsigned wlo = (w + 1 >= K) ? (w + 1 - K) : 0u;
                a = ldsWs[wlo - wlo0] - A;
                cnt = ldsWs[w + 1 - wlo0] - A - a;
            }
            const unsigned cmax = waveMaxU32(cnt);                        // wave-uniform network size
            const unsigned maxLoad = co.rg[rg0].maxLoad;
            if (cmax <= 128u)
            {
                done = true;
                if (w < W)
                {
                    if (cmax <= 16u)
                        stage1CellFast<16>(ldsPairs, a, cnt, w, maxLoad, q, fl);
                    else if (cmax <= 32u)
                        stage1CellFast<32>(ldsPairs, a, cnt, w, maxLoad, q, fl);
                    else if (cmax <= 64u)
                        stage1CellFast<64>(ldsPairs, a, cnt, w, maxLoad, q, fl);
                    else
                        stage1CellFast<128>(ldsPairs, a, cnt, w, maxLoad, q, fl);
                }
            }
        }
    }
    if (w >= W)
        return;
    if (!done)
        stage1CellGeneric(seg, co, s, w, q, fl);
    const size_t cell = (size_t)s * W + w;
    q3[cell] = q;
    flags[cell] = (uint8_t)fl;
    if ((fl & 2u) && q > co.minThreshold)
        winFlag[w] = 1;                 // benign race: every writer stores the same value
}

void launchStage1(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                  hipStream_t stream)
{
    dim3 grid((seg.nWindows + kStage1Block - 1) / kStage1Block, co.nSamples);
    hipLaunchKernelGGL(stage1Kernel, grid, dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag);
}

}  // namespace pd
