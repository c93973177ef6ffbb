// The window loop of `popdel call` in closed form, on the device: from the loader's slot tables (start entries of the
// read slot as the host filed them, one run per read group) to the pair instances {dev, start, on, off} the layout build
// (build.hip) takes -- the walk itself (reference: ChromosomeProfile::initializeActiveReads / nextWindow / nextStartWindow
// / nextEndWindow, profile_structure_popdel_call.h:1120-1213, driven by processSegment(), workflow_popdel.h:58-86) is not
// run anywhere: what it does to a pair is a function of the segment's shape (position of window 0, position of the
// loader's extra step, number of windows), of the pair's start and last window, and of which end slot the pair's end
// record was filed under. The host plans the shape from the heads and tails of its tables (ChromProfile::planSegment);
// the per-pair evaluation and the state that goes from segment to segment -- the pairs still active, the start entries no
// window has reached -- live here. Host reference of this file: pdh::WalkReference (chrom_profile.cpp), which
// tests/test_walk_closed_form.py checks against the window loop itself.
//
//   walkCountKernel   per read group: instances the segment will log
//   walkScanKernel    exclusive scan -> instance offsets of the read groups (the build's rgOffsets), total
//   walkEmitKernel    per read group: the instances in the order the walk logs them (carried pairs by id, then
//                     activations in entry order), the new active and pending lists
#include <algorithm>
#include <cstdlib>

#include "pd_walk.h"

namespace pd {

namespace {

struct Cand
{
    uint32_t start;
    int32_t  dev;
    uint32_t id;
    uint32_t lastWin;
    uint32_t next;      // its end record sits in the slot after the end read slot
};

// (the end tables file a pair under the 30-bp window of its end, whatever the window shift: EndSlots::add, posToWin(endPos, 30))
__device__ __forceinline__ uint32_t lastWindowOf(uint32_t start, int32_t dev, const RgConst & rc)
{
    const int inner = dev + rc.median - 2 * rc.readLength;
    const uint32_t endPos = start + (inner > 0 ? (uint32_t)inner : 0u);
    return endPos / 30u * 30u;
}

// window index at which a pair whose end record sits in the read slot leaves; >= n: not within this segment
__device__ __forceinline__ uint32_t offOf(const Cand & p, const WalkParams & w)
{
    if (p.next)
        return w.nWindows;
    const uint32_t over = p.lastWin + w.shift;
    if (w.hasStep && over < w.stepPos)
        return 0u;
    const uint32_t j = over < w.firstPos ? 1u : (over - w.firstPos) / w.shift + 1u;
    return min(j, w.nWindows);
}

__device__ __forceinline__ uint32_t onOf(const Cand & p, const WalkParams & w, bool equalRun)
{
    if (w.hasStep && p.start <= w.stepPos)
        return 0u;
    if (equalRun && p.start == w.firstPos)
        return 0u;
    return p.start <= w.firstPos ? 1u : (p.start - w.firstPos + w.shift - 1u) / w.shift;
}

// The k-th pair the read group's walk looks at: first the pairs still active (after the loader's switches), then the
// candidates for activation -- the pending entries of the old read slot, or the entries of the new one.
struct RgView
{
    unsigned nActive;        // carried pairs (after the switch rule)
    unsigned nCand;
    bool     fresh;          // candidates are the new slot's entries
    uint64_t entryLo;
};

// Block-wide ordered compaction: every thread with `flag` gets base + (number of flagged threads before it); returns the
// block's total through `total` (uniform). 256 threads.
__device__ __forceinline__ unsigned blockRank(bool flag, unsigned * shWave, unsigned & total)
{
    const unsigned long long bal = __ballot(flag);
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0)
        shWave[wave] = (unsigned)__popcll(bal);
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (unsigned k = 0; k < 4u; ++k)
    {
        before += k < wave ? shWave[k] : 0u;
        all += shWave[k];
    }
    total = all;
    return before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
}

// Loads pair k of the read group's sequence (see RgView). Carried pairs of the OLD state pass the switch rule first:
// `keep` tells whether the pair is still there.
template <bool Count>
__device__ void walkReadGroup(const WalkInput & in, const WalkState & st, const WalkParams & w, const RgConst * rgConst, unsigned rg,
                              unsigned * shWave, unsigned * shAcc, const uint64_t * instOffsets, uint32_t * instCount,
                              int32_t * outDev, uint32_t * outOn, uint32_t * outOff, uint32_t * outStart, WalkState & nxt,
                              unsigned * problems, unsigned long long * activeTotal)
{
    const unsigned sw = in.switches[rg];
    const bool equalRun = in.equalRun[rg] != 0;
    const RgConst rc = rgConst[rg];
    const unsigned nOldA = st.nActive[rg], nOldP = st.nPending[rg];
    const size_t aBase = (size_t)rg * st.capActive, pBase = (size_t)rg * st.capPending;
    const uint64_t eLo = in.rgOffsets[rg], eHi = in.rgOffsets[rg + 1];
    const unsigned nNew = sw != 0 ? (unsigned)(eHi - eLo) : 0u;
    const unsigned nCand = sw != 0 ? nNew : nOldP;
    uint64_t instAt = Count ? 0 : instOffsets[rg];
    unsigned nA = 0, nP = 0, nInst = 0;
    if (threadIdx.x == 0)
        shAcc[0] = 0;
    // ---- carried pairs
    for (unsigned base = 0; base < nOldA; base += blockDim.x)
    {
        const unsigned k = base + threadIdx.x;
        Cand p = {};
        bool there = false;
        if (k < nOldA)
        {
            p.start = st.aStart[aBase + k];
            p.dev = st.aDev[aBase + k];
            p.id = st.aId[aBase + k];
            p.lastWin = st.aLast[aBase + k] & 0x7fffffffu;
            p.next = st.aLast[aBase + k] >> 31;
            // a switch drains the end read slot: pairs whose end record sat there are gone, pairs filed ahead move into it
            there = sw == 0 || (sw == 1 && p.next != 0);
            if (sw != 0)
                p.next = 0;
        }
        const uint32_t off = there ? offOf(p, w) : 0u;
        const bool logs = there && off > 0u, stays = there && off >= w.nWindows;
        unsigned tot;
        const unsigned ri = blockRank(logs, shWave, tot);
        if (!Count && logs)
        {
            outDev[instAt + ri] = p.dev;
            outStart[instAt + ri] = p.start;
            outOn[instAt + ri] = 0u;
            outOff[instAt + ri] = off;
        }
        instAt += tot;
        nInst += tot;
        const unsigned ra = blockRank(stays, shWave, tot);
        if (!Count && stays)
        {
            if (nA + ra < nxt.capActive)
            {
                nxt.aStart[aBase + nA + ra] = p.start;
                nxt.aDev[aBase + nA + ra] = p.dev;
                nxt.aId[aBase + nA + ra] = p.id;
                nxt.aLast[aBase + nA + ra] = p.lastWin | (p.next << 31);
            }
        }
        nA += tot;
    }
    // ---- candidates for activation, in entry order
    for (unsigned base = 0; base < nCand; base += blockDim.x)
    {
        const unsigned k = base + threadIdx.x;
        Cand p = {};
        const bool have = k < nCand;
        if (have)
        {
            if (sw != 0)
            {
                const uint2 e = in.entries[eLo + k];
                p.start = e.x;
                p.dev = (int32_t)e.y;
                p.id = in.idOffset[rg] + k;
                p.lastWin = lastWindowOf(p.start, p.dev, rc);
                p.next = p.lastWin >= in.endBorder[rg] ? 1u : 0u;
            }
            else
            {
                p.start = st.pStart[pBase + k];
                p.dev = st.pDev[pBase + k];
                p.id = st.pId[pBase + k];
                p.lastWin = st.pLast[pBase + k] & 0x7fffffffu;
                p.next = st.pLast[pBase + k] >> 31;
            }
        }
        const uint32_t on = have ? onOf(p, w, equalRun) : 0u;
        const bool reached = have && on < w.nWindows;
        const uint32_t off = reached ? max(offOf(p, w), on) : 0u;
        const bool logs = reached && on < off, stays = reached && off >= w.nWindows, waits = have && !reached;
        unsigned tot;
        const unsigned ri = blockRank(logs, shWave, tot);
        if (!Count && logs)
        {
            outDev[instAt + ri] = p.dev;
            outStart[instAt + ri] = p.start;
            outOn[instAt + ri] = on;
            outOff[instAt + ri] = off;
        }
        instAt += tot;
        nInst += tot;
        const unsigned ra = blockRank(stays, shWave, tot);
        if (!Count && stays && nA + ra < nxt.capActive)
        {
            nxt.aStart[aBase + nA + ra] = p.start;
            nxt.aDev[aBase + nA + ra] = p.dev;
            nxt.aId[aBase + nA + ra] = p.id;
            nxt.aLast[aBase + nA + ra] = p.lastWin | (p.next << 31);
        }
        nA += tot;
        const unsigned rp = blockRank(waits, shWave, tot);
        if (!Count && waits && nP + rp < nxt.capPending)
        {
            nxt.pStart[pBase + nP + rp] = p.start;
            nxt.pDev[pBase + nP + rp] = p.dev;
            nxt.pId[pBase + nP + rp] = p.id;
            nxt.pLast[pBase + nP + rp] = p.lastWin | (p.next << 31);
        }
        nP += tot;
    }
    if (threadIdx.x == 0)
    {
        if (Count)
        {
            instCount[rg] = nInst;
            // what the lists of the state AFTER the segment must hold (the host grows them before the emit pass)
            atomicMax(problems, nA);
            atomicMax(problems + 1, nP);
        }
        else
        {
            nxt.nActive[rg] = min(nA, nxt.capActive);
            nxt.nPending[rg] = min(nP, nxt.capPending);
            if (nA > nxt.capActive)
                atomicOr(problems, 1u);
            if (nP > nxt.capPending)
                atomicOr(problems, 2u);
            if (nA)
                atomicAdd(activeTotal, (unsigned long long)nA);
        }
    }
}

__global__ __launch_bounds__(256) void walkCountKernel(WalkInput in, WalkState st, WalkParams w, const RgConst * rgConst,
                                                       uint32_t * instCount, unsigned * need)
{
    __shared__ unsigned shWave[4], shAcc[1];
    WalkState none = st;
    walkReadGroup<true>(in, st, w, rgConst, blockIdx.x, shWave, shAcc, nullptr, instCount, nullptr, nullptr, nullptr, nullptr, none,
                        need, nullptr);
}

__global__ __launch_bounds__(256) void walkEmitKernel(WalkInput in, WalkState st, WalkParams w, const RgConst * rgConst,
                                                      const uint64_t * instOffsets, int32_t * outDev, uint32_t * outOn, uint32_t * outOff,
                                                      uint32_t * outStart, WalkState nxt, unsigned * problems,
                                                      unsigned long long * activeTotal)
{
    __shared__ unsigned shWave[4], shAcc[1];
    walkReadGroup<false>(in, st, w, rgConst, blockIdx.x, shWave, shAcc, instOffsets, nullptr, outDev, outOn, outOff, outStart, nxt,
                         problems, activeTotal);
}

// instOffsets[0 .. nRg]: exclusive scan of the counts; total[0] = their sum
__global__ __launch_bounds__(256) void walkScanKernel(const uint32_t * __restrict__ counts, unsigned nRg, uint64_t * __restrict__ offsets,
                                                      uint64_t * __restrict__ total)
{
    __shared__ uint64_t sh[256];
    uint64_t carry = 0;
    for (unsigned base = 0; base < nRg; base += 256u)
    {
        const unsigned e = base + threadIdx.x;
        const uint64_t v = e < nRg ? counts[e] : 0u;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (unsigned o = 1; o < 256u; o <<= 1)
        {
            const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0u;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (e < nRg)
            offsets[e] = carry + sh[threadIdx.x] - v;
        carry += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        offsets[nRg] = carry;
        total[0] = carry;
    }
}

}  // namespace

void launchWalkCount(const WalkInput & in, const WalkState & st, const WalkParams & w, const RgConst * rgConst, unsigned nRg,
                     uint32_t * instCount, uint64_t * instOffsets, uint64_t * total, unsigned * need, hipStream_t stream)
{
    hipLaunchKernelGGL(walkCountKernel, dim3(nRg), dim3(256), 0, stream, in, st, w, rgConst, instCount, need);
    hipLaunchKernelGGL(walkScanKernel, dim3(1), dim3(256), 0, stream, instCount, nRg, instOffsets, total);
}

void launchWalkEmit(const WalkInput & in, const WalkState & st, const WalkParams & w, const RgConst * rgConst, unsigned nRg,
                    const uint64_t * instOffsets, int32_t * outDev, uint32_t * outOn, uint32_t * outOff, uint32_t * outStart,
                    const WalkState & nxt, unsigned * problems, unsigned long long * activeTotal, hipStream_t stream)
{
    hipLaunchKernelGGL(walkEmitKernel, dim3(nRg), dim3(256), 0, stream, in, st, w, rgConst, instOffsets, outDev, outOn, outOff, outStart, nxt,
                       problems, activeTotal);
}

namespace {
__global__ __launch_bounds__(256) void gatherHostEntriesKernel(const uint64_t * __restrict__ src, const uint64_t * __restrict__ rgOffsets,
                                                               uint2 * __restrict__ entries, unsigned nRg)
{
    for (unsigned r = blockIdx.x; r < nRg; r += gridDim.x)
    {
        const uint64_t a = src[r];
        if (a == 0)
            continue;
        const uint64_t lo = rgOffsets[r], n = rgOffsets[r + 1] - lo;
        const uint2 * from = reinterpret_cast<const uint2 *>(a);
        uint2 * to = entries + lo;
        // (table blocks and `entries` are 16-byte aligned as wholes; a read group's run begins on any 8-byte entry)
        uint64_t i = 0;
        if (n != 0 && ((reinterpret_cast<uintptr_t>(from) ^ reinterpret_cast<uintptr_t>(to)) & 8u) == 0)
        {
            const uint64_t head = (reinterpret_cast<uintptr_t>(from) & 8u) ? 1u : 0u;
            if (threadIdx.x == 0 && head)
                to[0] = from[0];
            const uint64_t pairs = (n - head) / 2;
            const uint4 * f4 = reinterpret_cast<const uint4 *>(from + head);
            uint4 * t4 = reinterpret_cast<uint4 *>(to + head);
            for (uint64_t k = threadIdx.x; k < pairs; k += 256u)
                t4[k] = f4[k];
            i = head + 2 * pairs;
        }
        for (uint64_t k = i + threadIdx.x; k < n; k += 256u)
            to[k] = from[k];
    }
}
}  // namespace

// Few workgroups: the link needs ~100 KB of requests in flight, not the chip. `popdel call` on 4,000 samples x 4 Mb, warmed
// rounds on one box: one workgroup per read group (4,000) 0.0638 s per round -- they hold workgroup slots for the length of the
// transfer and the loader's gunzip kernels beside them wait --, 512: 0.0651, 128: 0.0654, 32: 0.0611 (the staging copy: 0.0637).
static unsigned gatherGroups()
{
    static const unsigned n = [] {
        const char * e = std::getenv("POPDEL_HIP_GATHER_GROUPS");
        const int v = e ? std::atoi(e) : 0;
        return v > 0 ? (unsigned)v : 32u;
    }();
    return n;
}

void launchGatherHostEntries(const uint64_t * src, const uint64_t * rgOffsets, uint2 * entries, unsigned nRg, hipStream_t stream)
{
    if (nRg == 0)
        return;
    hipLaunchKernelGGL(gatherHostEntriesKernel, dim3(std::min(nRg, gatherGroups())), dim3(256), 0, stream, src, rgOffsets, entries, nRg);
}

void launchGatherDeviceEntries(const uint64_t * src, const uint64_t * rgOffsets, uint2 * entries, unsigned nRg, hipStream_t stream)
{
    if (nRg == 0)
        return;
    hipLaunchKernelGGL(gatherHostEntriesKernel, dim3(nRg), dim3(256), 0, stream, src, rgOffsets, entries, nRg);
}

}  // namespace pd
