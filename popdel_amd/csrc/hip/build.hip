// Device-side construction of the staged segment layout (SegmentView): pair instances {dev, start, on, off} as the host
// hands them over are cut into chunks of at most K windows, filed by slot and indexed by window -- what the reference does
// incrementally with ChromosomeProfile::add() (profile_structure_popdel_call.h:1084-1112) for every record it loads
// (load_profile_popdel_call.h:463-480). The instances cross PCIe once, compact (8 bytes each when they fit 16-bit
// fields, 16 otherwise); both chunk images, the slot index and the end counts are built here.
//
// The result is bit-identical to the host-side builder in capi.hip (buildChunks, kept for segments this one does not
// take): chunks of a slot stay in the order of the instances (stable), which is also the order stage 2 adds up the pairs
// of a cell in -- so results do not depend on who built the layout.
//
//   lifeStatsKernel   lifetime histogram (picks K) and the sum of active pair-windows
//   countKernel       per entity: chunks per slot -> winStart row (prefix sums), chunk end counts -> endCount row, densest
//                     512-window block, 16-bit fitness, order check
//   baseScanKernel    exclusive scan of the entities' chunk totals -> rgBase
//   scatterKernel     per entity, ONE WAVE walking the instances in order: deterministic stable placement of every chunk
//   scanGenericKernel 512-window blocks of multi-read-group samples in which a read group reaches maxLoad
#include "pd_build.h"

namespace pd {

struct Instance
{
    int32_t  dev;
    uint32_t start, on, off;
};

__device__ __forceinline__ Instance loadInstance(const InstanceArrays & a, uint64_t i)
{
    Instance r;
    if (a.wide)
    {
        r.dev = a.dev32[i];
        r.on = a.on32[i];
        r.off = a.off32[i];
        r.start = a.start32[i];
    }
    else
    {
        r.dev = a.dev16[i];
        r.on = a.on16[i];
        r.off = a.off16[i];
        r.start = a.firstPos + r.on * a.shift - a.back16[i];
    }
    return r;
}

__device__ __forceinline__ void entityRange(const BuildEntities & en, unsigned e, unsigned & rgLo, unsigned & rgHi)
{
    if (e < en.nRg)
    {
        rgLo = e;
        rgHi = e + 1;
    }
    else
    {
        const unsigned s = en.scanSample[e - en.nRg];
        rgLo = en.sampleRg[s];
        rgHi = en.sampleRg[s + 1];
    }
}

__device__ __forceinline__ unsigned slotOf(unsigned a, unsigned chunkOff, unsigned K)
{
    return a == 0u ? min(chunkOff, K) - 1u : a + K - 1u;
}

// ---------------------------------------------------------------------------------------------------------------------
// One workgroup per read group: running sum of the `on` differences, 256 instances per round.
__global__ __launch_bounds__(256) void expandKernel(CompactInstances in, const uint64_t * __restrict__ rgOffsets, unsigned W,
                                                    uint16_t * __restrict__ on16, uint16_t * __restrict__ off16,
                                                    uint16_t * __restrict__ back16)
{
    __shared__ unsigned perWave[4];
    const unsigned rg = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lo = rgOffsets[rg], hi = rgOffsets[rg + 1];
    const unsigned excLo = in.excRgStart[rg], excHi = in.excRgStart[rg + 1];
    unsigned carry = 0;
    for (uint64_t blk = lo; blk < hi; blk += 256u)
    {
        const uint64_t i = blk + threadIdx.x;
        unsigned d = 0, l = 0, bk = 0;
        if (i < hi)
        {
            d = in.onDelta[i];
            l = in.life[i];
            bk = in.back[i];
            if (d == 255u || l == 255u || bk == 255u)
            {
                // (exceptions: a pair that was active before the segment began, a long-lived pair, 255 windows or more
                // without a new pair -- a per cent of the instances; the read group's part of the list is short)
                unsigned a = excLo, b = excHi;
                while (a < b)
                {
                    const unsigned m = (a + b) >> 1;
                    if (in.excIndex[m] < (uint32_t)i)
                        a = m + 1u;
                    else
                        b = m;
                }
                d = in.excOnDelta[a];
                l = in.excLife[a];
                bk = in.excBack[a];
            }
        }
        unsigned incl = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const unsigned t = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= (unsigned)o)
                incl += t;
        }
        if (lane == 63u)
            perWave[wave] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
        for (unsigned k = 0; k < 4u; ++k)
        {
            before += k < wave ? perWave[k] : 0u;
            all += perWave[k];
        }
        const unsigned on = carry + before + incl;
        if (i < hi)
        {
            on16[i] = (uint16_t)min(on, 0xffffu);
            off16[i] = (uint16_t)min(on + l, W);
            back16[i] = (uint16_t)bk;
        }
        carry += all;
        __syncthreads();
    }
}

void launchExpand(const CompactInstances & in, const uint64_t * rgOffsets, unsigned nRg, unsigned W, uint16_t * on16, uint16_t * off16,
                  uint16_t * back16, hipStream_t st)
{
    if (nRg)
        hipLaunchKernelGGL(expandKernel, dim3(nRg), dim3(256), 0, st, in, rgOffsets, W, on16, off16, back16);
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lifeStatsKernel(InstanceArrays in, uint64_t n, unsigned W, unsigned long long * hist /* [67] */)
{
    __shared__ unsigned shHist[66];
    __shared__ unsigned long long shSum;
    if (threadIdx.x < 66u)
        shHist[threadIdx.x] = 0;
    if (threadIdx.x == 0)
        shSum = 0;
    __syncthreads();
    unsigned long long sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const unsigned on = in.wide ? in.on32[i] : in.on16[i];
        const unsigned off = min(in.wide ? in.off32[i] : (unsigned)in.off16[i], W);
        if (off <= on)
            continue;
        const unsigned life = off - on;
        sum += life;
        atomicAdd(&shHist[min(life, 65u)], 1u);
    }
    atomicAdd(&shSum, sum);
    __syncthreads();
    if (threadIdx.x < 66u && shHist[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], (unsigned long long)shHist[threadIdx.x]);
    if (threadIdx.x == 0 && shSum)
        atomicAdd(&hist[66], shSum);
}

// ---------------------------------------------------------------------------------------------------------------------
// Block-wide inclusive prefix sum of the LDS array v[0 .. n) in place (256 threads): every thread adds up its own
// contiguous stretch, the 256 stretch totals are scanned once, then every thread rewrites its stretch -- three barriers
// and one 8-step scan, whatever n is (scanning 256 elements at a time cost ~18 barriers per 256 elements).
__device__ __forceinline__ void blockPrefixInPlace(unsigned * v, unsigned n, unsigned * scratch /* [256] */)
{
    const unsigned per = (n + 255u) / 256u;
    const unsigned lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    unsigned sum = 0;
    for (unsigned i = lo; i < hi; ++i)
        sum += v[i];
    scratch[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned o = 1; o < 256u; o <<= 1)
    {
        const unsigned t = threadIdx.x >= o ? scratch[threadIdx.x - o] : 0u;
        __syncthreads();
        scratch[threadIdx.x] += t;
        __syncthreads();
    }
    unsigned run = scratch[threadIdx.x] - sum;            // total of the stretches before this one
    for (unsigned i = lo; i < hi; ++i)
    {
        run += v[i];
        v[i] = run;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void countKernel(InstanceArrays in, BuildEntities en, unsigned W, unsigned K, CountOut out)
{
    extern __shared__ unsigned sh[];
    unsigned * cnt = sh;                       // [W + K]   cnt[x] = chunks with slot x - 1 (shifted by one for the prefix sum)
    unsigned * ec = sh + (W + K);              // [W + 1]   chunks ending at w
    unsigned * scratch = ec + (W + 1);         // [256]
    __shared__ unsigned shFits, shUnordered;
    const unsigned e = blockIdx.x;
    for (unsigned i = threadIdx.x; i < W + K + W + 1; i += 256u)
        sh[i] = 0;
    if (threadIdx.x == 0)
    {
        shFits = (W <= 0xffffu) ? 1u : 0u;
        shUnordered = 0;
    }
    __syncthreads();
    unsigned rgLo, rgHi;
    entityRange(en, e, rgLo, rgHi);
    bool fits = true;
    unsigned unordered = 0;
    for (unsigned rg = rgLo; rg < rgHi; ++rg)
    {
        const uint64_t lo = en.rgOffsets[rg], hi = en.rgOffsets[rg + 1];
        for (uint64_t i = lo + threadIdx.x; i < hi; i += 256u)
        {
            const Instance p = loadInstance(in, i);
            if (i > lo)
            {
                const unsigned prevOn = in.wide ? in.on32[i - 1] : (unsigned)in.on16[i - 1];
                unordered += prevOn > p.on ? 1u : 0u;
            }
            const unsigned off = min(p.off, W);
            fits = fits && p.dev >= -32767 && p.dev <= 32766;
            for (unsigned a = p.on; a < off; a += K)
            {
                const unsigned chunkOff = min(off, a + K);
                atomicAdd(&cnt[slotOf(a, chunkOff, K) + 1u], 1u);
                atomicAdd(&ec[chunkOff], 1u);
            }
        }
    }
    if (!fits)
        shFits = 0;                            // every writer stores the same value
    if (unordered)
        atomicAdd(&shUnordered, unordered);
    __syncthreads();
    blockPrefixInPlace(cnt, W + K, scratch);   // cnt[x] = first chunk whose slot >= x (the slots run 0 .. W + K - 2)
    blockPrefixInPlace(ec, W + 1, scratch);
    uint32_t * ws = out.winStart + (size_t)e * (W + K);
    for (unsigned i = threadIdx.x; i < W + K; i += 256u)
        ws[i] = cnt[i];
    uint32_t * eo = out.endCount + (size_t)e * (W + 1);
    for (unsigned i = threadIdx.x; i <= W; i += 256u)
        eo[i] = ec[i];
    // densest 512-window block (the scan kernel stages ws[w0] .. ws[min(W, w0 + 512) + K - 1])
    unsigned m = 0;
    for (unsigned w0 = threadIdx.x * 2u * kStage1Block; w0 < W; w0 += 256u * 2u * kStage1Block)
    {
        const unsigned whi = min(W, w0 + 2u * kStage1Block);
        m = max(m, cnt[whi + K - 1u] - cnt[w0]);
    }
    scratch[threadIdx.x] = m;
    __syncthreads();
    for (unsigned o = 128u; o > 0u; o >>= 1)
    {
        if (threadIdx.x < o)
            scratch[threadIdx.x] = max(scratch[threadIdx.x], scratch[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        out.entChunks[e] = cnt[W + K - 1u];
        out.entMax[e] = scratch[0];
        out.rgPacked[e] = (uint8_t)shFits;
        if (shUnordered)
            atomicAdd(out.problems, shUnordered);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// rgBase[e] = chunks of the entities before e; totals[0] = chunks of the real read groups, totals[1] = all chunks.
__global__ __launch_bounds__(256) void baseScanKernel(const uint32_t * __restrict__ entChunks, unsigned nEnt, unsigned nRg,
                                                      uint64_t * __restrict__ rgBase, uint64_t * __restrict__ totals)
{
    __shared__ uint64_t sh[256];
    uint64_t carry = 0;
    for (unsigned base = 0; base < nEnt; base += 256u)
    {
        const unsigned e = base + threadIdx.x;
        const uint64_t v = e < nEnt ? entChunks[e] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 256u; o <<= 1)
        {
            const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        const uint64_t excl = carry + sh[threadIdx.x] - v;
        if (e < nEnt)
        {
            rgBase[e] = excl;
            if (e == nRg)
                totals[0] = excl;
        }
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        totals[1] = carry;
        if (nEnt == nRg)
            totals[0] = carry;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per entity walks its instances in order, 64 at a time. The host builder files the chunks of a slot in instance
// order (a cursor per slot, instances visited one after the other). Here: instances are ordered by `on`, so within a
// block of 64 the instances with equal `on` are neighbours, and the m-th chunk of an instance (first window on + m K) has
// slot on + (m + 1) K - 1: all m-th chunks with the same slot form one run of neighbouring lanes -- their places are the
// slot's cursor plus their rank in the run. Within a slot the chunks with larger m come from smaller `on`, i.e. from
// earlier instances: classes are handled in descending m. Chunks present from window 0 on (on == 0, m == 0) are filed
// under their end window instead, which is not ordered: one ballot per possible slot (K of them) ranks those.
__global__ __launch_bounds__(64) void scatterKernel(InstanceArrays in, BuildEntities en, unsigned W, unsigned K, ScatterOut out)
{
    extern __shared__ unsigned cursor[];       // [W + K - 1] next free place of every slot, relative to rgBase[e]
    const unsigned e = blockIdx.x, lane = threadIdx.x;
    const uint32_t * ws = out.winStart + (size_t)e * (W + K);
    for (unsigned i = lane; i < W + K - 1u; i += 64u)
        cursor[i] = ws[i];
    __syncthreads();
    const bool real = e < en.nRg && out.pairs != nullptr;       // stage-2 arrays wanted (and this is a real read group)
    const bool packed = out.pairsPk != nullptr;                 // stage-1 image wanted
    const uint64_t base = out.rgBase[e];
    const unsigned long long below = (1ull << lane) - 1ull;
    unsigned rgLo, rgHi;
    entityRange(en, e, rgLo, rgHi);
    for (unsigned rg = rgLo; rg < rgHi; ++rg)
    {
        const uint64_t lo = en.rgOffsets[rg], hi = en.rgOffsets[rg + 1];
        // (the instances of the next block are fetched while this one is placed: the walk is a chain of dependent LDS
        // steps, the loads must not be part of it)
        Instance nextP;
        nextP.dev = 0;
        nextP.start = 0;
        nextP.on = 0xffffffffu;
        nextP.off = 0;
        if (lo + lane < hi)
            nextP = loadInstance(in, lo + lane);
        for (uint64_t blk = lo; blk < hi; blk += 64u)
        {
            const uint64_t i = blk + lane;
            const bool have = i < hi;
            const Instance p = nextP;
            nextP.dev = 0;
            nextP.start = 0;
            nextP.on = 0xffffffffu;
            nextP.off = 0;
            if (i + 64u < hi)
                nextP = loadInstance(in, i + 64u);
            const unsigned off = have ? min(p.off, W) : 0u;
            const unsigned nCh = (have && off > p.on) ? (off - p.on + K - 1u) / K : 0u;
            // runs of equal `on` among neighbouring lanes
            const unsigned prevOn = (unsigned)__shfl_up((int)p.on, 1, 64);
            const bool head = lane == 0u || prevOn != p.on;
            const unsigned long long heads = __ballot(head);
            const unsigned runStart = 63u - (unsigned)__clzll((long long)(heads & ((below << 1) | 1ull)));
            const unsigned long long after = heads & ~((below << 1) | 1ull);
            const unsigned runEnd = after ? (unsigned)__ffsll((long long)after) - 1u : 64u;       // first lane of the next run
            const unsigned long long runMask = (runEnd >= 64u ? ~0ull : ((1ull << runEnd) - 1ull)) & ~((1ull << runStart) - 1ull);
            unsigned maxCh = nCh;
            for (int o = 32; o > 0; o >>= 1)
                maxCh = max(maxCh, (unsigned)__shfl_xor((int)maxCh, o, 64));
            const uint32_t d16 = ((uint32_t)p.dev & 0xffffu) ^ 0x8000u;      // biased: see stage1.hip
            for (unsigned mm = maxCh; mm-- > 0u;)
            {
                const bool valid = mm < nCh;
                const unsigned a = p.on + mm * K;
                const unsigned chunkOff = valid ? min(off, a + K) : 0u;
                const bool zero = valid && a == 0u;            // present from window 0 on: filed under its end
                unsigned place = 0;
                // ordinary chunks: one run per slot
                {
                    const bool ord = valid && !zero;
                    const unsigned long long ordMask = __ballot(ord);
                    if (ord)
                    {
                        const unsigned slot = a + K - 1u;
                        const unsigned long long mine = ordMask & runMask;
                        const unsigned rank = (unsigned)__popcll(mine & below);
                        const unsigned cur = cursor[slot];
                        place = cur + rank;
                        if (rank + 1u == (unsigned)__popcll(mine))
                            cursor[slot] = cur + rank + 1u;    // the last of the run moves the cursor (reads above precede it)
                    }
                }
                if (__ballot(zero))
                {
                    const unsigned slot = zero ? min(chunkOff, K) - 1u : 0xffffffffu;
                    for (unsigned sIdx = 0; sIdx < K; ++sIdx)
                    {
                        const unsigned long long m = __ballot(zero && slot == sIdx);
                        if (zero && slot == sIdx)
                        {
                            const unsigned rank = (unsigned)__popcll(m & below);
                            const unsigned cur = cursor[sIdx];
                            place = cur + rank;
                            if (rank + 1u == (unsigned)__popcll(m))
                                cursor[sIdx] = cur + rank + 1u;
                        }
                    }
                }
                if (valid)
                {
                    if (real)
                    {
                        PairDO q;
                        q.dev = p.dev;
                        q.off = chunkOff;
                        out.pairs[base + place] = q;
                        out.start[base + place] = p.start;
                    }
                    if (packed)
                        out.pairsPk[base + place] = d16 | ((chunkOff & 0xffffu) << 16);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Where a real read group of a multi-read-group sample reaches its maxLoad, the reference drops that read group's pairs:
// those 512-window blocks of the sample go to the fallback scan kernel (it knows the read groups).
__global__ __launch_bounds__(256) void scanGenericKernel(BuildEntities en, const RgConst * __restrict__ rgConst,
                                                         const uint32_t * __restrict__ winStart, const uint32_t * __restrict__ endCount,
                                                         unsigned W, unsigned K, uint8_t * __restrict__ scanGeneric, unsigned blocksPerSample)
{
    const unsigned j = blockIdx.x;
    const unsigned s = en.scanSample[j];
    for (unsigned rg = en.sampleRg[s]; rg < en.sampleRg[s + 1]; ++rg)
    {
        const uint32_t * ws = winStart + (size_t)rg * (W + K);
        const uint32_t * ec = endCount + (size_t)rg * (W + 1);
        const unsigned maxLoad = rgConst[rg].maxLoad;
        for (unsigned w = threadIdx.x; w < W; w += 256u)
            if (ws[w + K] - ec[w] >= maxLoad)
                scanGeneric[(size_t)j * blocksPerSample + w / (2u * kStage1Block)] = 1;
    }
}

// ---- launchers --------------------------------------------------------------------------------------------------------
void launchLifeStats(const InstanceArrays & in, uint64_t n, unsigned W, unsigned long long * hist, hipStream_t st)
{
    if (n == 0)
        return;
    const unsigned blocks = (unsigned)((n + 256ull * 16 - 1) / (256ull * 16) < 4096 ? (n + 256ull * 16 - 1) / (256ull * 16) : 4096);
    hipLaunchKernelGGL(lifeStatsKernel, dim3(blocks), dim3(256), 0, st, in, n, W, hist);
}

size_t buildLdsBytes(unsigned W, unsigned K) { return ((size_t)(W + K) + (W + 1) + 256) * sizeof(unsigned); }

bool launchCount(const InstanceArrays & in, const BuildEntities & en, unsigned W, unsigned K, const CountOut & out, hipStream_t st)
{
    const size_t lds = buildLdsBytes(W, K);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(countKernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(countKernel, dim3(en.nRg + en.nScan), dim3(256), lds, st, in, en, W, K, out);
    return true;
}

void launchBaseScan(const uint32_t * entChunks, unsigned nEnt, unsigned nRg, uint64_t * rgBase, uint64_t * totals, hipStream_t st)
{
    hipLaunchKernelGGL(baseScanKernel, dim3(1), dim3(256), 0, st, entChunks, nEnt, nRg, rgBase, totals);
}

bool launchScatter(const InstanceArrays & in, const BuildEntities & en, unsigned W, unsigned K, const ScatterOut & out, hipStream_t st)
{
    const size_t lds = (size_t)(W + K) * sizeof(unsigned);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(scatterKernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(scatterKernel, dim3(en.nRg + en.nScan), dim3(64), lds, st, in, en, W, K, out);
    return true;
}

void launchScanGeneric(const BuildEntities & en, const RgConst * rgConst, const uint32_t * winStart, const uint32_t * endCount, unsigned W,
                       unsigned K, uint8_t * scanGeneric, unsigned blocksPerSample, hipStream_t st)
{
    if (en.nScan)
        hipLaunchKernelGGL(scanGenericKernel, dim3(en.nScan), dim3(256), 0, st, en, rgConst, winStart, endCount, W, K, scanGeneric,
                           blocksPerSample);
}

}  // namespace pd
