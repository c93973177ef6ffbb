// Stage 1 of the joint-calling hot path on gfx950: for every (sample, window) cell gather the insert-size deviations
// of the active read pairs and take their upper-half median (Q3).
//
// Reference behaviour: initialize_deletion_lengths() first half (genotype_deletion_popdel_call.h:33-57) with
// ChromosomeProfile::getActiveReadsDeviations() (profile_structure_popdel_call.h:1443-1478) and upperHalfMedian()
// (genotype_deletion_popdel_call.h:15-27).
//
// Mapping (main kernel): one lane per PAIR of adjacent windows of one sample, both cells packed into the 16-bit halves of
// the lane's registers; a 256-thread workgroup covers 512 consecutive windows. Consecutive windows share most of their
// pairs, so the workgroup stages the union of their pair chunks ONCE from HBM into LDS (coalesced 8-byte loads); HBM
// traffic is the compact chunk stream instead of the expanded [sample x window x pair] tensor. Each lane then pulls its
// candidates from LDS into a register array -- inactive slots become -INF / +INF fillers such that the wanted order
// statistics land at static positions -- and sorts it with a straight-line Batcher network (packed VALU min/max, no
// cross-lane traffic, no divergence) pruned to those positions. Kernels: stage1Kernel (up to 64 candidates, four LDS
// capacities), stage1BigKernel (up to 128), stage1FallbackKernel (32-bit, one window per lane, or rank counting).
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
    unsigned total = 0, usable = 0;
    for (unsigned rg = co.sampleRg[s]; rg < co.sampleRg[s + 1]; ++rg)
    {
        const uint32_t * ws = seg.winStart + (size_t)rg * (seg.nWindows + K);
        const PairDO * p = seg.pairs + seg.rgBase[rg];
        unsigned c = 0;
        for (unsigned i = ws[w]; i < ws[w + K]; ++i)
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
        const uint32_t * ws = seg.winStart + (size_t)rg * (seg.nWindows + K);
        const PairDO * p = seg.pairs + seg.rgBase[rg];
        unsigned c = 0;
        for (unsigned i = ws[w]; i < ws[w + K]; ++i)
            c += (p[i].off > w) ? 1u : 0u;
        if (c >= co.rg[rg].maxLoad)
            continue;
        for (unsigned i = ws[w]; i < ws[w + K]; ++i)
        {
            if (!(p[i].off > w))
                continue;
            const int v = p[i].dev;
            unsigned rank = 0, innerIdx = 0;
            for (unsigned rg2 = co.sampleRg[s]; rg2 < co.sampleRg[s + 1]; ++rg2)
            {
                const uint32_t * ws2 = seg.winStart + (size_t)rg2 * (seg.nWindows + K);
                const PairDO * p2 = seg.pairs + seg.rgBase[rg2];
                unsigned c2 = 0;
                for (unsigned j = ws2[w]; j < ws2[w + K]; ++j)
                    c2 += (p2[j].off > w) ? 1u : 0u;
                if (c2 >= co.rg[rg2].maxLoad)
                    continue;
                for (unsigned j = ws2[w]; j < ws2[w + K]; ++j)
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

// ---- packed 16-bit variant: two cells per lane ----------------------------------------------------------------------
// Deviations that fit int16 are sorted two at a time with v_pk_min_u16 / v_pk_max_u16: the low half of every register
// belongs to window w, the high half to window w + 1. The staged image holds the deviations BIASED by 0x8000 (order
// preserving map of int16 onto uint16), so that the two fillers are the all-zeros / all-ones patterns: -INF = 0x0000,
// +INF = 0xFFFF -- a filler is then simply a sign mask, no constant to mix in. The number of active pairs n(w) is known
// before the values are loaded (pairs started minus pairs ended, both per window prefix counts), so the inactive slots
// can be filled with p(w) copies of -INF and +INF otherwise such that the wanted order statistics are those of the STATIC
// ranks T and T+1: a selection network for exactly these two ranks replaces the sorter.
typedef unsigned short v2s __attribute__((ext_vector_type(2)));

#define PD_CE16(i, j)                                          \
    {                                                          \
        const v2s lo_ = __builtin_elementwise_min(r[i], r[j]); \
        const v2s hi_ = __builtin_elementwise_max(r[i], r[j]); \
        r[i] = lo_;                                            \
        r[j] = hi_;                                            \
    }

// Selection instead of sorting: sortnet_gen.h's PD_SELNET_<C> leaves the order statistics of ranks T and T + 1 on the
// wires PD_SELNET_<C>_LO / _HI (group sorts + truncated odd-even merges, only comparators in the cone of the two).
template <int C>
struct SelNet16;
#define PD_SELNET16(C)                                                                      \
    template <>                                                                             \
    struct SelNet16<C>                                                                      \
    {                                                                                       \
        static constexpr int kLo = PD_SELNET_##C##_LO, kHi = PD_SELNET_##C##_HI;            \
        static __device__ __forceinline__ void run(v2s (&r)[C]) { PD_SELNET_##C(PD_CE16) } \
    };
PD_SELNET16(32)
PD_SELNET16(40)
PD_SELNET16(48)
PD_SELNET16(56)
PD_SELNET16(64)
PD_SELNET16(80)
PD_SELNET16(96)
PD_SELNET16(112)
PD_SELNET16(128)
#undef PD_SELNET16

__device__ __forceinline__ unsigned rankL0(unsigned n)
{
    return (n < 4u) ? (n - 1u) : ((3u * n + 2u + (n & 1u) - 4u) >> 2);
}

constexpr int kPackedWindows = 2 * kStage1Block;          // windows per workgroup
constexpr int kPackedLdsRound = 2304;                      // 9 x 256 chunks are copied per round (512 windows x ~3.1 + K x 3 at 30x)
// LDS capacities (chunks per workgroup) the scan kernel is instantiated for; the launch picks the smallest that holds the
// densest 512-window block of the segment (deeper coverage = more chunks per block), blocks beyond the largest fall back.
constexpr int kPackedLdsSmall = 2176, kPackedLdsMedium = 3328, kPackedLdsLarge = 4864, kPackedLdsHuge = 6912;
constexpr int kPackedWsLen = kPackedWindows + 64 + 2;

// Packed-16-bit helpers on 32-bit registers (low half = window w, high half = window w + 1).
__device__ __forceinline__ unsigned pkSub(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (v2s)(__builtin_bit_cast(v2s, a) - __builtin_bit_cast(v2s, b)));
}
__device__ __forceinline__ unsigned pkAdd(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, (v2s)(__builtin_bit_cast(v2s, a) + __builtin_bit_cast(v2s, b)));
}
__device__ __forceinline__ unsigned pkSignMask(unsigned a)          // 0xFFFF in every half that is negative
{
    return __builtin_bit_cast(unsigned, (v2s)(__builtin_bit_cast(v2s, a) >> 15));
}

// A staged chunk is one dword: deviation (biased int16) in the low half, (segment-relative) end window in the high half. The
// packed arithmetic below wants each of them in BOTH halves of a register: one v_perm_b32 each (rounds 1-4 kept the chunks
// expanded, 8 bytes each, in HBM and LDS: twice the bytes for the screen, which reads every chunk once and duplicates nothing).
struct PairPk
{
    unsigned devdev;
    unsigned offoff;
};
__device__ __forceinline__ PairPk expandChunk(unsigned e)
{
    PairPk p;
    p.devdev = __builtin_amdgcn_perm(e, e, 0x01000100u);
    p.offoff = __builtin_amdgcn_perm(e, e, 0x03020302u);
    return p;
}

// Two ADJACENT windows per lane (w in the low halves, w + 1 in the high halves). Their candidate ranges overlap almost
// completely, so every LDS entry is read once and feeds both cells: slot c of the lane is entry (aA + c) for both.
// All per-slot logic (active test, range test, filler choice, filler count) is packed 16-bit arithmetic on both cells
// at once: a slot costs 7 VALU for the two cells.
// The per-slot logic below is written with the packed instructions spelled out: left to itself hipcc canonicalises
// "(a - b) >> 15" on <2 x i16> into a compare + select per half (v_cmp_*_sdwa, v_cndmask, v_perm: ~16 VALU per slot
// instead of 7). These are pure register-to-register VALU ops, so the asm statements need no waits or clobbers.
__device__ __forceinline__ unsigned asmPkSub(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned asmPkAdd(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned asmPkAddS(unsigned a, unsigned sconst)      // second operand from an SGPR
{
    unsigned r;
    asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(sconst));
    return r;
}
__device__ __forceinline__ unsigned asmPkSign(unsigned a)                       // 0xFFFF in every negative half
{
    unsigned r;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}

template <int C>
__device__ __forceinline__ void stage1PairFast(const uint32_t * lds, unsigned aA, unsigned wPk, unsigned cntPk, unsigned nA,
                                               unsigned nB, int & vaA, int & vbA, int & vaB, int & vbB)
{
    constexpr unsigned T = (3u * C + 2u - 4u) >> 2;        // l0(C): 23 / 29 / 35 / 41 / 47 for 32 / 40 / 48 / 56 / 64 slots
    // number of -INF fillers that moves rank l0(n) to position T
    const unsigned pA = (nA >= 2u) ? (T - rankL0(nA)) : 0u;
    const unsigned pB = (nB >= 2u) ? (T - rankL0(nB)) : 0u;
    const unsigned one = 0x00010001u, minusOne = 0xFFFFFFFFu;
    const unsigned w1Pk = asmPkAddS(wPk, one);             // w + 1
    unsigned rPk = asmPkAddS(cntPk, minusOne);             // cnt - 1 - c: negative once the slot is out of range
    unsigned vPk = pA | (pB << 16);                        // p - #inactive slots so far: >= 0 -> this filler is -INF
    v2s r[C];
#pragma unroll
    for (int c = 0; c < C; ++c)
    {
        const PairPk e = expandChunk(lds[aA + c]);
        // INACTIVE <=> off <= w (off - (w + 1) negative) or c >= cnt: 7 VALU per slot for the two cells (+ 2 for the expansion)
        // (a native formulation -- vector types, shift amount hidden from the optimiser -- compiled to 8 and ran 7 % slower)
        const unsigned mI = asmPkSign(asmPkSub(e.offoff, w1Pk) | rPk);
        vPk = asmPkAdd(vPk, mI);                           // one filler more
        const unsigned fill = asmPkSign(vPk);              // 0x0000 = -INF while within the first p, 0xFFFF = +INF after
        const unsigned x = (fill & mI) | (e.devdev & ~mI);
        rPk = asmPkAddS(rPk, minusOne);
        r[c] = __builtin_bit_cast(v2s, x);
    }
    SelNet16<C>::run(r);
    // back from the biased representation
    vaA = (int)(short)(r[SelNet16<C>::kLo].x ^ 0x8000u);
    vbA = (int)(short)(r[SelNet16<C>::kHi].x ^ 0x8000u);
    vaB = (int)(short)(r[SelNet16<C>::kLo].y ^ 0x8000u);
    vbB = (int)(short)(r[SelNet16<C>::kHi].y ^ 0x8000u);
}

// Same interpolation for int16 inputs: everything fits 32 bits.
__device__ __forceinline__ int q3FromRankPair16(unsigned n, int a, int b)
{
    const int k = (int)((3u * n + 2u + (n & 1u)) & 3u);       // (pos4 - 4) & 3 == pos4 & 3
    return ((4 - k) * a + k * b + 2) >> 2;
}

__device__ __forceinline__ void storeCell(unsigned n, unsigned maxLoad, int va, int vb, size_t cell, unsigned w, int minThreshold,
                                          int32_t * q3, uint8_t * flags, int * winFlag)
{
    int q = 0;
    unsigned fl = 0;
    if (n < 2u)
        fl = 1u;
    else if (n < maxLoad)
    {
        q = (n < 4u) ? va : q3FromRankPair16(n, va, vb);
        fl = 2u;
    }
    q3[cell] = q;
    flags[cell] = (uint8_t)fl;
    if ((fl & 2u) && q > minThreshold)
        winFlag[w] = 1;                 // benign race: every writer stores the same value
}

// ---- the scan of one 512-window block of one sample -------------------------------------------------------------------
// kBig = false: candidate counts up to 64 (networks of 32..64 inputs); kBig = true: up to 128 (64..128 inputs; deeper
// coverage), run by a second kernel over the blocks the first one could not take. Returns 0 if the block was done,
// 1 if it must go to the 32-bit fallback kernel, 2 (kBig = false only) if it should be retried with kBig = true.
// All return paths are workgroup-uniform.
template <int kLdsPairs, bool kBig>
__device__ __forceinline__ int scanBlock(const SegmentView & seg, const CohortView & co, int32_t * __restrict__ q3,
                                         uint8_t * __restrict__ flags, int * __restrict__ winFlag, unsigned s, unsigned bx,
                                         unsigned blocksPerSample, uint32_t * ldsPairs, uint32_t * ldsWs, unsigned short * ldsN,
                                         unsigned short * unitOrder, unsigned (*classCount)[kStage1Block / 64])
{
    constexpr unsigned kMaxCand = kBig ? 128u : 64u;
    const unsigned w0 = bx * kPackedWindows;
    const unsigned W = seg.nWindows;
    const unsigned K = seg.maxLifePk;            // chunk life of the stage-1 image (its own, see SegmentView)
    const unsigned tid = threadIdx.x;
    const unsigned rg0 = seg.scanRg[s];          // the (possibly appended) read group whose stream holds all pairs of s
    const bool rgOverloaded = rg0 >= seg.nRealRg && seg.scanGeneric[(size_t)(rg0 - seg.nRealRg) * blocksPerSample + bx] != 0;
    if (!(K <= 64u && seg.rgPacked[rg0]) || rgOverloaded)
        return 1;
    const uint32_t * ws = seg.winStartPk + (size_t)rg0 * (W + K);
    const unsigned wlo0 = w0;                                        // first slot the block's windows see
    const unsigned whi = min(W, w0 + kPackedWindows);                // one past the block's last window
    const unsigned nWs = whi - w0 + K;                               // slots w0 .. whi + K - 1 (the last one is the end entry)
    for (unsigned i = tid; i < nWs; i += kStage1Block)
        ldsWs[i] = ws[wlo0 + i];
    // pairs that ended at or before the lane's windows (host-built prefix counts): n(w) = started - ended
    const unsigned wA = w0 + 2u * tid, wB = wA + 1u;
    const uint32_t * ec = seg.endCountPk + (size_t)rg0 * (W + 1);
    const unsigned endedA = (wA < W) ? ec[wA] : 0u;
    const unsigned endedB = (wB < W) ? ec[wB] : 0u;
    __syncthreads();
    const unsigned A0 = ldsWs[0];
    const unsigned nPairs = ldsWs[nWs - 1] - A0;
    const bool tooMany = nPairs > (unsigned)kLdsPairs;               // workgroup-uniform
    // The LDS image is the HBM image: the chunks of the block are copied by LDS-DMA (global_load_lds_dwordx4: 64 lanes x
    // 16 B = 256 chunks per wave instruction, no VGPR round trip, no ds_write, no per-element address arithmetic; the
    // register-staged copy cost ~100 VALU per wave). The source is aligned down to 16 bytes: LDS entry i is stream entry
    // A + i with A = A0 - mis; the last instruction may copy up to 255 chunks beyond the block's (both arrays are padded).
    const size_t g0 = seg.rgBasePk[rg0] + A0;
    const unsigned mis = (unsigned)(g0 & 3u);
    const unsigned A = A0 - mis;
    if (!tooMany)
    {
        const uint32_t * src = seg.pairsPk + (g0 - mis);
        const unsigned nCopy = nPairs + mis;
        const unsigned lane = tid & 63u;
        for (unsigned base = (tid >> 6) * 256u; base < nCopy; base += (unsigned)(kStage1Block / 64) * 256u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + base + 4u * lane),
                                             (__attribute__((address_space(3))) void *)(ldsPairs + base), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // lane t owns the adjacent windows w0 + 2t and w0 + 2t + 1; B's range is expressed in A's slot coordinates
    unsigned aA = 0, cntA = 0, endB = 0;
    if (wA < W)
    {
        aA = ldsWs[wA - wlo0] - A;
        cntA = ldsWs[wA + K - wlo0] - A - aA;
    }
    if (wB < W)
        endB = ldsWs[wB + K - wlo0] - A - aA;
    const unsigned cntU = max(cntA, endB);
    const int over = __syncthreads_or(cntU > kMaxCand);            // also orders the LDS writes above
    if (tooMany || over)
    {
        // the first kernel hands the block to the second one if that has a chance (128 candidates, the largest LDS stage)
        if (!kBig && nPairs <= (unsigned)kPackedLdsHuge && !__syncthreads_or(cntU > 128u))
            return 2;
        return 1;
    }
    // active pairs per window; parked in LDS because the lanes are re-dealt below
    ldsN[2u * tid] = (unsigned short)((wA < W) ? (ldsWs[wA + K - wlo0] - endedA) : 0u);
    ldsN[2u * tid + 1u] = (unsigned short)((wB < W) ? (ldsWs[wB + K - wlo0] - endedB) : 0u);
    // Re-deal the 256 window pairs to the lanes ordered by candidate count (largest first): every wave then runs the
    // smallest network that fits ITS lanes instead of the one that fits the worst lane of an arbitrary 64.
    constexpr unsigned kClasses = 6;
    constexpr unsigned kStep = kBig ? 16u : 8u, kTop = kBig ? 128u : 64u;        // class bounds: top, top - step, ...
    {
        const unsigned cls = (wA >= W) ? 5u
                                       : (cntU > kTop - kStep ? 0u
                                                              : (cntU > kTop - 2 * kStep ? 1u : (cntU > kTop - 3 * kStep ? 2u : (cntU > kTop - 4 * kStep ? 3u : 4u))));
        const unsigned long long below = (1ull << (tid & 63u)) - 1ull;
        unsigned rank = 0;
#pragma unroll
        for (unsigned k = 0; k < kClasses; ++k)
        {
            const unsigned long long bal = __ballot(cls == k);
            if (cls == k)
                rank = (unsigned)__popcll(bal & below);
            if ((tid & 63u) == 0u)
                classCount[k][tid >> 6] = (unsigned)__popcll(bal);
        }
        __syncthreads();
        // units ahead of this lane's (class, wave): the 24 counts are the same for every lane, so they are pulled into
        // scalar registers and added up by the scalar unit; what is left per lane is the choice by its class
        // (each lane adding up the 24 counts itself cost 110 VALU per wave)
        const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
        unsigned offset = 0, run = 0;
#pragma unroll
        for (unsigned k = 0; k < kClasses; ++k)
        {
            unsigned within = 0, total = 0;
#pragma unroll
            for (unsigned v = 0; v < kStage1Block / 64; ++v)
            {
                const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)classCount[k][v]);
                within += v < wv ? c : 0u;
                total += c;
            }
            offset = cls == k ? run + within : offset;
            run += total;
        }
        unitOrder[offset + rank] = (unsigned short)tid;
        __syncthreads();
    }
    const unsigned u = unitOrder[tid];
    const unsigned uA = w0 + 2u * u, uB = uA + 1u;
    if (uA < W)
    {
        unsigned xA, xCntA, xEndB = 0;
        xA = ldsWs[uA - wlo0] - A;
        xCntA = ldsWs[uA + K - wlo0] - A - xA;
        if (uB < W)
            xEndB = ldsWs[uB + K - wlo0] - A - xA;
        const unsigned nA = ldsN[2u * u];
        const unsigned nB = ldsN[2u * u + 1u];
        const unsigned cmax = waveMaxU32(max(xCntA, xEndB));         // wave-uniform network size
        // entries ahead of B's range ended before w + 1 (chunk life <= K), so B needs no lower range test
        const unsigned wPk = uA | (uB << 16);                        // absolute window numbers (W fits 16 bits here)
        const unsigned cntPk = xCntA | (xEndB << 16);
        int vaA, vbA, vaB, vbB;
        if (kBig)
        {
            if (cmax <= 64u)
                stage1PairFast<64>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else if (cmax <= 80u)
                stage1PairFast<80>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else if (cmax <= 96u)
                stage1PairFast<96>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else if (cmax <= 112u)
                stage1PairFast<112>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else
                stage1PairFast<128>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
        }
        else
        {
            if (cmax <= 32u)
                stage1PairFast<32>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else if (cmax <= 40u)
                stage1PairFast<40>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else if (cmax <= 48u)
                stage1PairFast<48>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else if (cmax <= 56u)
                stage1PairFast<56>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
            else
                stage1PairFast<64>(ldsPairs, xA, wPk, cntPk, nA, nB, vaA, vbA, vaB, vbB);
        }
        const unsigned maxLoad = co.rg[rg0].maxLoad;
        storeCell(nA, maxLoad, vaA, vbA, (size_t)s * W + uA, uA, co.minThreshold, q3, flags, winFlag);
        if (uB < W)
            storeCell(nB, maxLoad, vaB, vbB, (size_t)s * W + uB, uB, co.minThreshold, q3, flags, winFlag);
    }
    return 0;
}

// Appends the two 256-window tiles of block (s, bx) to the list of the 32-bit fallback kernel (lane 0).
__device__ __forceinline__ void deferToFallback(unsigned s, unsigned bx, unsigned W, unsigned tilesPerSample, uint8_t * tileTodo,
                                                unsigned * todoList, bool defer)
{
    if (threadIdx.x != 0)
        return;
    uint8_t * todo = tileTodo + (size_t)s * tilesPerSample + 2u * bx;
    const bool secondTile = (bx * kPackedWindows + kStage1Block) < W;
    todo[0] = defer ? 1 : 0;
    if (secondTile)
        todo[1] = defer ? 1 : 0;
    if (!defer)
        return;
    // todoList[0] counts the entries; the order of the list does not matter
    const unsigned tile = s * tilesPerSample + 2u * bx;
    const unsigned at = atomicAdd(&todoList[0], secondTile ? 2u : 1u);
    todoList[1 + at] = tile;
    if (secondTile)
        todoList[2 + at] = tile + 1;
}

// Main scan kernel: one workgroup per (512-window block, sample); int16 deviations, at most 64 candidates per cell.
// Blocks with up to 128 candidates are appended to bigList for stage1BigKernel, everything else it cannot take to
// todoList (256-window tiles) for stage1FallbackKernel.
template <int kLdsPairs>
__device__ __forceinline__ void scanKernelBody(const SegmentView & seg, const CohortView & co, int32_t * __restrict__ q3,
                                               uint8_t * __restrict__ flags, int * __restrict__ winFlag, uint8_t * __restrict__ tileTodo,
                                               unsigned * __restrict__ todoList, unsigned * __restrict__ bigList, unsigned tilesPerSample)
{
    __shared__ __attribute__((aligned(16))) uint32_t ldsPairs[kLdsPairs + 256];
    __shared__ uint32_t ldsWs[kPackedWsLen];
    __shared__ unsigned short ldsN[kPackedWindows];
    __shared__ unsigned short unitOrder[kStage1Block];
    __shared__ unsigned classCount[6][kStage1Block / 64];
    const int st = scanBlock<kLdsPairs, false>(seg, co, q3, flags, winFlag, blockIdx.y, blockIdx.x, gridDim.x, ldsPairs, ldsWs, ldsN,
                                               unitOrder, classCount);
    if (st == 2)
    {
        if (threadIdx.x == 0)
        {
            deferToFallback(blockIdx.y, blockIdx.x, seg.nWindows, tilesPerSample, tileTodo, todoList, false);
            bigList[1 + atomicAdd(&bigList[0], 1u)] = blockIdx.y * gridDim.x + blockIdx.x;
        }
        return;
    }
    deferToFallback(blockIdx.y, blockIdx.x, seg.nWindows, tilesPerSample, tileTodo, todoList, st == 1);
}

// The instantiation for ordinary coverage (smallest LDS stage: 7 workgroups fit a CU) is compiled for 7 waves per SIMD:
// 72 VGPRs, with the 64-input network spilling a few values to scratch. A/B on one box: 6 waves (75 VGPRs, what the
// compiler picks by itself) 1.380 ms, 8 waves (64 VGPRs, 88 B scratch) 1.311 ms, 7 waves 1.29 ms -- the scan alternates
// load, barrier and compute phases, more resident waves keep the VALU fed across them.
__global__ __launch_bounds__(kStage1Block) __attribute__((amdgpu_waves_per_eu(7, 7))) void stage1Kernel(
    SegmentView seg, CohortView co, int32_t * __restrict__ q3, uint8_t * __restrict__ flags, int * __restrict__ winFlag,
    uint8_t * __restrict__ tileTodo, unsigned * __restrict__ todoList, unsigned * __restrict__ bigList, unsigned tilesPerSample)
{
    scanKernelBody<kPackedLdsSmall>(seg, co, q3, flags, winFlag, tileTodo, todoList, bigList, tilesPerSample);
}

// The same with larger LDS stages (deeper coverage): occupancy is bounded by LDS there, registers are left alone.
template <int kLdsPairs>
__global__ __launch_bounds__(kStage1Block) void stage1KernelDeep(SegmentView seg, CohortView co, int32_t * __restrict__ q3,
                                                                 uint8_t * __restrict__ flags, int * __restrict__ winFlag,
                                                                 uint8_t * __restrict__ tileTodo, unsigned * __restrict__ todoList,
                                                                 unsigned * __restrict__ bigList, unsigned tilesPerSample)
{
    scanKernelBody<kLdsPairs>(seg, co, q3, flags, winFlag, tileTodo, todoList, bigList, tilesPerSample);
}

// The same scan as a PERSISTENT grid (round 4): wgPerCu x CUs workgroups that draw their (sample, block) from a counter.
// A launch of 140,000 workgroups holds every wave slot of the chip for its whole 1.1 ms; this one leaves the slots, VGPRs
// and LDS of 7 - wgPerCu workgroups per CU free for as long as it runs, which is where the short stage-2 kernels of the
// PREVIOUS segment run (on the context's other stream) instead of waiting for the scan to drain. Same registers, same
// LDS, same results; blocks are drawn in the order of the plain launch (x fastest).
__global__ __launch_bounds__(kStage1Block) __attribute__((amdgpu_waves_per_eu(7, 7))) void stage1KernelPersistent(
    SegmentView seg, CohortView co, int32_t * __restrict__ q3, uint8_t * __restrict__ flags, int * __restrict__ winFlag,
    uint8_t * __restrict__ tileTodo, unsigned * __restrict__ todoList, unsigned * __restrict__ bigList, unsigned tilesPerSample,
    unsigned blocksPerSample, unsigned nBlocks, unsigned * __restrict__ counter)
{
    __shared__ __attribute__((aligned(16))) uint32_t ldsPairs[kPackedLdsSmall + 256];
    __shared__ uint32_t ldsWs[kPackedWsLen];
    __shared__ unsigned short ldsN[kPackedWindows];
    __shared__ unsigned short unitOrder[kStage1Block];
    __shared__ unsigned classCount[6][kStage1Block / 64];
    __shared__ unsigned drawn;
    for (;;)
    {
        __syncthreads();                                           // LDS reuse across blocks (and of `drawn`)
        if (threadIdx.x == 0)
            drawn = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned id = drawn;
        if (id >= nBlocks)
            return;
        const unsigned s = id / blocksPerSample, bx = id - s * blocksPerSample;
        const int st = scanBlock<kPackedLdsSmall, false>(seg, co, q3, flags, winFlag, s, bx, blocksPerSample, ldsPairs, ldsWs, ldsN,
                                                         unitOrder, classCount);
        if (st == 2)
        {
            if (threadIdx.x == 0)
            {
                deferToFallback(s, bx, seg.nWindows, tilesPerSample, tileTodo, todoList, false);
                bigList[1 + atomicAdd(&bigList[0], 1u)] = id;
            }
        }
        else
            deferToFallback(s, bx, seg.nWindows, tilesPerSample, tileTodo, todoList, st == 1);
    }
}

// Second pass over the blocks with 65..128 candidates somewhere (deeper coverage): same scan with networks of up to
// 128 inputs and the largest LDS stage; 128 value registers per lane, hence its own kernel (the register budget of a
// kernel is that of its largest path). Workgroups stride over the list.
__global__ __launch_bounds__(kStage1Block) void stage1BigKernel(SegmentView seg, CohortView co, int32_t * __restrict__ q3,
                                                                uint8_t * __restrict__ flags, int * __restrict__ winFlag,
                                                                uint8_t * __restrict__ tileTodo, unsigned * __restrict__ todoList,
                                                                const unsigned * __restrict__ bigList, unsigned tilesPerSample,
                                                                unsigned blocksPerSample)
{
    __shared__ __attribute__((aligned(16))) uint32_t ldsPairs[kPackedLdsHuge + 256];
    __shared__ uint32_t ldsWs[kPackedWsLen];
    __shared__ unsigned short ldsN[kPackedWindows];
    __shared__ unsigned short unitOrder[kStage1Block];
    __shared__ unsigned classCount[6][kStage1Block / 64];
    const unsigned nBig = bigList[0];
    for (unsigned k = blockIdx.x; k < nBig; k += gridDim.x)
    {
        const unsigned id = bigList[1 + k], s = id / blocksPerSample, bx = id % blocksPerSample;
        __syncthreads();                                           // LDS reuse across blocks
        const int st = scanBlock<kPackedLdsHuge, true>(seg, co, q3, flags, winFlag, s, bx, blocksPerSample, ldsPairs, ldsWs, ldsN,
                                                       unitOrder, classCount);
        if (st != 0)
            deferToFallback(s, bx, seg.nWindows, tilesPerSample, tileTodo, todoList, true);
    }
}

// Tiles the main kernel skipped: 32-bit 64- or 128-input network where the LDS stage still fits, otherwise rank counting
// straight from HBM. Workgroups stride over the compact list the main kernel appended to (an empty list costs a launch).
__global__ __launch_bounds__(kStage1Block) void stage1FallbackKernel(SegmentView seg, CohortView co, int32_t * __restrict__ q3,
                                                                     uint8_t * __restrict__ flags, int * __restrict__ winFlag,
                                                                     const unsigned * __restrict__ todoList,
                                                                     unsigned tilesPerSample)
{
    __shared__ PairDO ldsPairs[kStage1LdsPairs + 128];
    __shared__ uint32_t ldsWs[kStage1Block + 64 + 2];
    const unsigned nTodo = todoList[0];
    for (unsigned k = blockIdx.x; k < nTodo; k += gridDim.x)
    {
        const unsigned tile = todoList[1 + k];
        const unsigned s = tile / tilesPerSample;
        const unsigned w0 = (tile % tilesPerSample) * kStage1Block;
        const unsigned W = seg.nWindows;
        const unsigned tid = threadIdx.x;
        const unsigned w = w0 + tid;
        // the sample's own read group or its scan read group (all pairs of the sample as one stream, packed image only)
        const unsigned rg0 = seg.scanRg[s];
        const bool merged = rg0 >= seg.nRealRg;
        const unsigned K = merged ? seg.maxLifePk : seg.maxLife;     // each image with its own chunk life and index
        const bool rgOverloaded = merged && seg.scanGeneric[(size_t)(rg0 - seg.nRealRg) * ((W + 2u * kStage1Block - 1u) / (2u * kStage1Block)) +
                                                            w0 / (2u * kStage1Block)] != 0;
        int q = 0;
        unsigned fl = 0;
        bool done = false;
        __syncthreads();                                           // LDS reuse across tiles
        if (K <= 64u && !rgOverloaded && (!merged || seg.rgPacked[rg0]))
        {
            const uint32_t * ws = (merged ? seg.winStartPk : seg.winStart) + (size_t)rg0 * (W + K);
            const unsigned wlo0 = w0;
            const unsigned whi = min(W, w0 + kStage1Block);
            const unsigned nWs = whi - w0 + K;
            for (unsigned i = tid; i < nWs; i += kStage1Block)
                ldsWs[i] = ws[wlo0 + i];
            __syncthreads();
            const unsigned A = ldsWs[0];
            const unsigned nPairs = ldsWs[nWs - 1] - A;
            if (nPairs <= (unsigned)kStage1LdsPairs)
            {
                if (!merged)
                {
                    const PairDO * src = seg.pairs + seg.rgBase[rg0] + A;
                    for (unsigned i = tid; i < nPairs; i += kStage1Block)
                        ldsPairs[i] = src[i];
                }
                else
                {
                    const uint32_t * src = seg.pairsPk + seg.rgBasePk[rg0] + A;
                    for (unsigned i = tid; i < nPairs; i += kStage1Block)
                    {
                        const uint32_t e = src[i];
                        PairDO p;
                        p.dev = (int)(short)((e & 0xffffu) ^ 0x8000u);        // the packed image is biased
                        p.off = e >> 16;
                        ldsPairs[i] = p;
                    }
                }
                __syncthreads();
                unsigned a = 0, cnt = 0;
                if (w < W)
                {
                    a = ldsWs[w - wlo0] - A;
                    cnt = ldsWs[w + K - wlo0] - A - a;
                }
                const unsigned cmax = waveMaxU32(cnt);
                if (cmax <= 128u)
                {
                    done = true;
                    if (w < W)
                    {
                        if (cmax <= 64u)
                            stage1CellFast<64>(ldsPairs, a, cnt, w, co.rg[rg0].maxLoad, q, fl);
                        else
                            stage1CellFast<128>(ldsPairs, a, cnt, w, co.rg[rg0].maxLoad, q, fl);
                    }
                }
            }
        }
        if (w < W)
        {
            if (!done)
                stage1CellGeneric(seg, co, s, w, q, fl);
            const size_t cell = (size_t)s * W + w;
            q3[cell] = q;
            flags[cell] = (uint8_t)fl;
            if ((fl & 2u) && q > co.minThreshold)
                winFlag[w] = 1;
        }
    }
}

// ---- the SCREEN (round 5): which windows can have a candidate at all, without taking a single order statistic --------
// What stage 1 is for (reference initialize_deletion_lengths(), genotype_deletion_popdel_call.h:33-87): a window yields
// candidate lengths only if some sample's Q3 exceeds a threshold, and every threshold is at least T = co.minThreshold.
// Q3 interpolates the order statistics a = x(l0) <= b = x(l0 + 1) of the n active deviations with weights (4 - k, k) / 4:
//     Q3 > T  <=  a > T  <=>  c >= n - l0          (c = number of active deviations > T)
//     Q3 <= T <=  b <= T <=>  c <= n - l0 - 2      (k = 0 or n < 4: Q3 = a, the first line decides alone)
// and in between (c == n - l0 - 1) the window is a MAYBE. So the scan only has to COUNT, per cell, the active pairs whose
// deviation exceeds T -- and those are rare (T sits at several standard deviations): instead of every window looking at
// its ~3 K candidate chunks (each chunk is looked at K times, 22 packed VALU operations per look with the selection
// network), every CHUNK is looked at once -- one coalesced 16-byte load per two chunks straight from HBM, one compare --
// and the few that exceed T add one to the LDS counters of the windows they are active in (slot found by bisection in the
// block's window index). n(w) comes from the prefix counts as before. The kernel is bound by reading the chunk stream.
// Windows flagged here (sure or maybe) get their exact Q3 -- for all samples, as the clustering needs them -- from
// exactFlaggedKernel below, which runs the selection networks for those windows only; a maybe that turns out at or below T
// stays in the flagged list and yields no candidate (every cluster threshold is >= T), exactly like a window whose Q3
// values exceed T but not their cluster's threshold. flags[] are written for every cell (they follow from n alone).
// Blocks the screen cannot take (deviations beyond 16 bits, a multi-read-group sample with a read group at its maxLoad,
// K > 64) go to stage1FallbackKernel through the same list as before, which leaves exact Q3, flags and window flags.
__device__ __forceinline__ bool screenMaybe(unsigned n, unsigned c)
{
    // n >= 2, n < maxLoad
    if (n < 4u)
        return c >= 1u;                                 // Q3 is the largest value
    const unsigned pos4 = 3u * n + 2u + (n & 1u) - 4u;
    const unsigned l0 = pos4 >> 2, k = pos4 & 3u;
    const unsigned need = n - l0;                       // c >= need: a > T
    return k == 0u ? c >= need : c + 1u >= need;        // c == need - 1: only b > T, decided by the exact pass
}

// A workgroup takes kScreenBlocks adjacent 512-window blocks of one sample (1,024 windows, ~3,200 chunks = 13 KB of
// stream): every global load it needs -- the active counts, the chunks themselves (four 16-byte loads per lane, four chunks
// each, held in registers) -- is issued before its first barrier. Lane t owns the windows t, t + 256, ... of the workgroup
// (byte loads and stores of a wave are contiguous). Measured per 10,000 x 6,667 launch: 512 / 1,024 / 2,048 windows per
// workgroup 0.288 / 0.217 / 0.255 ms.
constexpr int kScreenBlocks = 2;
constexpr int kScreenWindows = kScreenBlocks * kPackedWindows;
constexpr int kScreenPre = 4;                     // chunk loads per lane issued ahead of the barrier (4 chunks each)

// n(w) of every (stream, window) as one byte (capped at 255), from the index the build leaves: the screen reads 1 B per cell
// instead of the two 4-byte index rows (which were 0.53 of its 1.42 GB per 10,000 x 6,667 launch). Part of staging.
__global__ __launch_bounds__(256) void active8Kernel(const uint32_t * __restrict__ winStartPk, const uint32_t * __restrict__ endCountPk,
                                                     unsigned W, unsigned K, uint8_t * __restrict__ out)
{
    const unsigned e = blockIdx.y, w = blockIdx.x * 256u + threadIdx.x;
    if (w >= W)
        return;
    const unsigned n = winStartPk[(size_t)e * (W + K) + w + K] - endCountPk[(size_t)e * (W + 1) + w];
    out[(size_t)e * W + w] = (uint8_t)min(n, 255u);
}

void launchActive8(const uint32_t * winStartPk, const uint32_t * endCountPk, unsigned nEntities, unsigned W, unsigned K, uint8_t * out,
                   hipStream_t stream)
{
    if (nEntities == 0 || W == 0)
        return;
    hipLaunchKernelGGL(active8Kernel, dim3((W + 255u) / 256u, nEntities), dim3(256), 0, stream, winStartPk, endCountPk, W, K, out);
}

__global__ __launch_bounds__(kStage1Block) void screenKernel(SegmentView seg, CohortView co, uint8_t * __restrict__ flags,
                                                             int * __restrict__ winFlag, uint8_t * __restrict__ tileTodo,
                                                             unsigned * __restrict__ todoList, unsigned tilesPerSample,
                                                             unsigned blocksPerSample)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    __shared__ unsigned ldsCnt[kScreenWindows];
    const unsigned s = blockIdx.y, bx0 = blockIdx.x * kScreenBlocks;
    const unsigned w0 = bx0 * kPackedWindows;
    const unsigned W = seg.nWindows;
    const unsigned K = seg.maxLifePk;
    const unsigned tid = threadIdx.x;
    const unsigned rg0 = seg.scanRg[s];
    bool take = K <= 64u && seg.rgPacked[rg0];                                   // workgroup-uniform
    if (rg0 >= seg.nRealRg)
        for (unsigned k = 0; k < (unsigned)kScreenBlocks && bx0 + k < blocksPerSample; ++k)
            take = take && seg.scanGeneric[(size_t)(rg0 - seg.nRealRg) * blocksPerSample + bx0 + k] == 0;
    for (unsigned k = 0; k < (unsigned)kScreenBlocks && bx0 + k < blocksPerSample; ++k)
        deferToFallback(s, bx0 + k, W, tilesPerSample, tileTodo, todoList, !take);
    if (!take)
        return;
    const uint32_t * ws = seg.winStartPk + (size_t)rg0 * (W + K);
    const unsigned whi = min(W, w0 + kScreenWindows);
    const unsigned nWs = whi - w0 + K;
    // (the stream's extent: uniform addresses, scalar loads)
    const unsigned A0 = ws[w0];
    const unsigned nPairs = ws[w0 + nWs - 1u] - A0;
    const size_t g0 = seg.rgBasePk[rg0] + A0;
    const unsigned mis = (unsigned)(g0 & 3u);
    const unsigned nCopy = nPairs + mis;                                        // stream entries [A0 - mis, A0 + nPairs)
    const v4u * src = reinterpret_cast<const v4u *>(seg.pairsPk + (g0 - mis));
    const unsigned nLoads = (nCopy + 3u) >> 2;
    v4u pre[kScreenPre];
#pragma unroll
    for (int k = 0; k < kScreenPre; ++k)
    {
        const unsigned i = tid + (unsigned)k * kStage1Block;
        pre[k] = __builtin_nontemporal_load(src + min(i, nLoads ? nLoads - 1u : 0u));        // (clamped: the value is ignored)
    }
    const uint8_t * n8 = seg.active8 + (size_t)rg0 * W;
    unsigned nAct[kScreenWindows / kStage1Block];
#pragma unroll
    for (int k = 0; k < kScreenWindows / kStage1Block; ++k)
    {
        const unsigned w = w0 + tid + (unsigned)k * kStage1Block;
        nAct[k] = w < W ? n8[w] : 0u;
        ldsCnt[tid + (unsigned)k * kStage1Block] = 0u;
    }
    const int T = co.minThreshold;
    __syncthreads();
    // a chunk that exceeds T (rare): one more active big pair in every window it is active in (slot x sees windows (x - K, x]);
    // its slot by bisection in the stream's window index, read where it lies
    auto hit = [&](unsigned j, unsigned chunk) {
        if (j < mis || j >= nCopy)
            return;
        const unsigned pos = A0 - mis + j;
        unsigned lo = 0u, hi = nWs - 1u;                 // ws[w0 + lo] <= pos < ws[w0 + hi]
        while (hi - lo > 1u)
        {
            const unsigned mid = (lo + hi) >> 1;
            if (ws[w0 + mid] <= pos)
                lo = mid;
            else
                hi = mid;
        }
        const int slot = (int)(w0 + lo);
        const int off = (int)(chunk >> 16);
        const int first = max((int)w0, slot - (int)K + 1);
        const int last = min(min(slot, off - 1), (int)whi - 1);
        for (int w = first; w <= last; ++w)
            atomicAdd(&ldsCnt[w - (int)w0], 1u);
    };
    auto look = [&](unsigned i, const v4u & v) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((int)(short)((v[k] & 0xffffu) ^ 0x8000u) > T)
                hit(4u * i + (unsigned)k, v[k]);
    };
#pragma unroll
    for (int k = 0; k < kScreenPre; ++k)
    {
        const unsigned i = tid + (unsigned)k * kStage1Block;
        if (i < nLoads)
            look(i, pre[k]);
    }
    for (unsigned i = tid + (unsigned)kScreenPre * kStage1Block; i < nLoads; i += kStage1Block)         // (deeper coverage)
        look(i, __builtin_nontemporal_load(src + i));
    __syncthreads();
    const unsigned maxLoad = co.rg[rg0].maxLoad;
    const uint32_t * ec = seg.endCountPk + (size_t)rg0 * (W + 1);
#pragma unroll
    for (int k = 0; k < kScreenWindows / kStage1Block; ++k)
    {
        const unsigned w = w0 + tid + (unsigned)k * kStage1Block;
        if (w < W)
        {
            unsigned n = nAct[k];
            if (n == 255u)                              // (capped: the two index words say how many)
                n = ws[w + K] - ec[w];
            const unsigned fl = n < 2u ? 1u : (n < maxLoad ? 2u : 0u);
            flags[(size_t)s * W + w] = (uint8_t)fl;
            if (fl == 2u && screenMaybe(n, ldsCnt[w - w0]))
                winFlag[w] = 1;
        }
    }
}

// Exact Q3 of ALL samples in the flagged windows (what the clustering reads): one lane per sample takes the window PAIR
// (w & ~1, w | 1) of a flagged window w straight from the chunk stream in HBM -- its candidates are one contiguous run
// of ~3 K chunks -- into the register array of the scan's selection networks. flagged[] holds nFlagged (or *nFlaggedDev)
// window numbers; an odd window whose even neighbour is flagged too is left to that one. Cells in blocks the screen
// handed to the fallback kernel are exact already. Wave tiles with more than 64 candidates somewhere are appended to
// `rest` (count in rest[0]) for exactFlaggedRestKernel.
__device__ __forceinline__ bool exactLaneSetup(const SegmentView & seg, unsigned s, unsigned wA, unsigned tilesPerSample,
                                               const uint8_t * tileTodo, unsigned & rg0, unsigned & xA, unsigned & cntA, unsigned & endB,
                                               unsigned & nA, unsigned & nB)
{
    const unsigned W = seg.nWindows, K = seg.maxLifePk, wB = wA + 1u;
    if (tileTodo[(size_t)s * tilesPerSample + wA / kStage1Block] != 0)
        return false;                                   // the fallback kernel has done this tile
    rg0 = seg.scanRg[s];
    const uint32_t * ws = seg.winStartPk + (size_t)rg0 * (W + K);
    const uint32_t * ec = seg.endCountPk + (size_t)rg0 * (W + 1);
    xA = ws[wA];
    const unsigned eA = ws[wA + K];
    cntA = eA - xA;
    nA = eA - ec[wA];
    endB = 0u;
    nB = 0u;
    if (wB < W)
    {
        const unsigned eB = ws[wB + K];
        endB = eB - xA;
        nB = eB - ec[wB];
    }
    return true;
}

template <int C>
__device__ __forceinline__ void exactPairStore(const SegmentView & seg, const CohortView & co, unsigned s, unsigned rg0, unsigned wA,
                                               unsigned xA, unsigned cntA, unsigned endB, unsigned nA, unsigned nB,
                                               int32_t * __restrict__ q3, uint8_t * __restrict__ flags, int * __restrict__ winFlag)
{
    const unsigned W = seg.nWindows, wB = wA + 1u;
    const uint32_t * stream = seg.pairsPk + seg.rgBasePk[rg0];
    int vaA, vbA, vaB, vbB;
    stage1PairFast<C>(stream, xA, wA | (wB << 16), cntA | (endB << 16), nA, nB, vaA, vbA, vaB, vbB);
    const unsigned maxLoad = co.rg[rg0].maxLoad;
    storeCell(nA, maxLoad, vaA, vbA, (size_t)s * W + wA, wA, co.minThreshold, q3, flags, winFlag);
    if (wB < W)
        storeCell(nB, maxLoad, vaB, vbB, (size_t)s * W + wB, wB, co.minThreshold, q3, flags, winFlag);
}

__global__ __launch_bounds__(256) void exactFlaggedKernel(SegmentView seg, CohortView co, const int * __restrict__ flagged,
                                                          unsigned nFlagged, const unsigned * __restrict__ nFlaggedDev,
                                                          const int * __restrict__ ownWinFlag, const uint8_t * __restrict__ tileTodo,
                                                          unsigned tilesPerSample, int32_t * __restrict__ q3, uint8_t * __restrict__ flags,
                                                          int * __restrict__ winFlag, unsigned * __restrict__ rest)
{
    // (the launch is made for a CAPACITY of flagged windows the host has not seen: the grid's y strides over the real number)
    const unsigned nF = nFlaggedDev ? min(*nFlaggedDev, nFlagged) : nFlagged;
    const unsigned s = blockIdx.x * 256u + threadIdx.x;
    for (unsigned f = blockIdx.y; f < nF; f += gridDim.y)
    {
        const unsigned w = (unsigned)flagged[f];
        if ((w & 1u) && f != 0u && (unsigned)flagged[f - 1u] == w - 1u)
            continue;                                       // the even neighbour's lanes take the pair
        const unsigned wA = w & ~1u;
        unsigned rg0 = 0, xA = 0, cntA = 0, endB = 0, nA = 0, nB = 0;
        const bool mine = s < co.nSamples && exactLaneSetup(seg, s, wA, tilesPerSample, tileTodo, rg0, xA, cntA, endB, nA, nB);
        const unsigned cnt = mine ? max(cntA, endB) : 0u;
        const unsigned cmax = waveMaxU32(cnt);
        if (cmax > 64u)
        {
            if ((threadIdx.x & 63u) == 0u)
                rest[1u + atomicAdd(&rest[0], 1u)] = f * ((co.nSamples + 63u) / 64u) + (s >> 6);
            continue;
        }
        if (!mine)
            continue;
        if (cmax <= 32u)
            exactPairStore<32>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
        else if (cmax <= 40u)
            exactPairStore<40>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
        else if (cmax <= 48u)
            exactPairStore<48>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
        else if (cmax <= 56u)
            exactPairStore<56>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
        else
            exactPairStore<64>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
    }
}

// The wave tiles with 65 or more candidates in some lane (deeper coverage): networks of up to 128 inputs, and rank
// counting from the 32-bit image for lanes beyond that.
__global__ __launch_bounds__(64) void exactFlaggedRestKernel(SegmentView seg, CohortView co, const int * __restrict__ flagged,
                                                             const uint8_t * __restrict__ tileTodo, unsigned tilesPerSample,
                                                             int32_t * __restrict__ q3, uint8_t * __restrict__ flags,
                                                             int * __restrict__ winFlag, const unsigned * __restrict__ rest)
{
    const unsigned nRest = rest[0], wavesPerWindow = (co.nSamples + 63u) / 64u, W = seg.nWindows;
    for (unsigned k = blockIdx.x; k < nRest; k += gridDim.x)
    {
        const unsigned id = rest[1u + k], f = id / wavesPerWindow, s = (id % wavesPerWindow) * 64u + threadIdx.x;
        const unsigned wA = (unsigned)flagged[f] & ~1u, wB = wA + 1u;
        unsigned rg0 = 0, xA = 0, cntA = 0, endB = 0, nA = 0, nB = 0;
        const bool mine = s < co.nSamples && exactLaneSetup(seg, s, wA, tilesPerSample, tileTodo, rg0, xA, cntA, endB, nA, nB);
        const unsigned cnt = mine ? max(cntA, endB) : 0u;
        const bool network = cnt <= 128u;
        const unsigned cmax = waveMaxU32(network ? cnt : 0u);
        if (mine && network)
        {
            if (cmax <= 64u)
                exactPairStore<64>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
            else if (cmax <= 80u)
                exactPairStore<80>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
            else if (cmax <= 96u)
                exactPairStore<96>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
            else if (cmax <= 112u)
                exactPairStore<112>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
            else
                exactPairStore<128>(seg, co, s, rg0, wA, xA, cntA, endB, nA, nB, q3, flags, winFlag);
        }
        else if (mine)
        {
            for (unsigned w = wA; w <= wB && w < W; ++w)
            {
                int q;
                unsigned fl;
                stage1CellGeneric(seg, co, s, w, q, fl);
                q3[(size_t)s * W + w] = q;
                flags[(size_t)s * W + w] = (uint8_t)fl;
                if ((fl & 2u) && q > co.minThreshold)
                    winFlag[w] = 1;
            }
        }
    }
}

// room: how the scan shares the chip with the other stream's kernels (ScanRoom, pd_device.h); ordinary coverage only --
// the deeper LDS stages are bounded by LDS already and launch as before.
void launchStage1(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                  uint8_t * tileTodo, unsigned * todoList, unsigned * bigList, unsigned maxBlockChunks, const ScanRoom & room,
                  hipStream_t stream)
{
    const unsigned tilesPerSample = (seg.nWindows + kStage1Block - 1) / kStage1Block;
    dim3 grid((seg.nWindows + kPackedWindows - 1) / kPackedWindows, co.nSamples);
    // maxBlockChunks: chunks of the densest 512-window block of (99 % of) the streams, host-computed at staging time
    if (maxBlockChunks <= (unsigned)kPackedLdsSmall && room.mode == ScanRoom::kPersistent)
    {
        const unsigned nBlocks = grid.x * grid.y;
        const unsigned groups = min(nBlocks, room.wgPerCu * room.nCu);
        hipLaunchKernelGGL(stage1KernelPersistent, dim3(groups), dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag,
                           tileTodo, todoList, bigList, tilesPerSample, grid.x, nBlocks, room.counter);
    }
    else if (maxBlockChunks <= (unsigned)kPackedLdsSmall && room.mode == ScanRoom::kLdsCap)
    {
        // unused dynamic LDS on top of the kernel's own so that at most wgPerCu workgroups fit the CU's 160 KB
        constexpr unsigned kOwn = (kPackedLdsSmall + 256) * 4 + kPackedWsLen * 4 + kPackedWindows * 2 + kStage1Block * 2 + 6 * (kStage1Block / 64) * 4;
        const unsigned want = 163840u / (room.wgPerCu + 1u) + 1536u;
        const unsigned pad = want > kOwn ? want - kOwn : 0u;
        hipLaunchKernelGGL(stage1Kernel, grid, dim3(kStage1Block), pad, stream, seg, co, q3, flags, winFlag, tileTodo,
                           todoList, bigList, tilesPerSample);
    }
    else if (maxBlockChunks <= (unsigned)kPackedLdsSmall)
        hipLaunchKernelGGL(stage1Kernel, grid, dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag, tileTodo,
                           todoList, bigList, tilesPerSample);
    else if (maxBlockChunks <= (unsigned)kPackedLdsMedium)
        hipLaunchKernelGGL(stage1KernelDeep<kPackedLdsMedium>, grid, dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag, tileTodo,
                           todoList, bigList, tilesPerSample);
    else if (maxBlockChunks <= (unsigned)kPackedLdsLarge)
        hipLaunchKernelGGL(stage1KernelDeep<kPackedLdsLarge>, grid, dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag, tileTodo,
                           todoList, bigList, tilesPerSample);
    else
        hipLaunchKernelGGL(stage1KernelDeep<kPackedLdsHuge>, grid, dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag, tileTodo,
                           todoList, bigList, tilesPerSample);
}

// The blocks the main kernel left for the 128-candidate networks, then the tiles left for the 32-bit paths.
void launchStage1Fallback(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                          uint8_t * tileTodo, unsigned * todoList, const unsigned * bigList, hipStream_t stream)
{
    const unsigned tilesPerSample = (seg.nWindows + kStage1Block - 1) / kStage1Block;
    const unsigned blocksPerSample = (seg.nWindows + kPackedWindows - 1) / kPackedWindows;
    const unsigned nBlocks = blocksPerSample * co.nSamples;
    hipLaunchKernelGGL(stage1BigKernel, dim3(nBlocks < 1024u ? nBlocks : 1024u), dim3(kStage1Block), 0, stream, seg, co, q3, flags,
                       winFlag, tileTodo, todoList, bigList, tilesPerSample, blocksPerSample);
    const unsigned nTiles = tilesPerSample * co.nSamples;
    const unsigned blocks = nTiles < 2048u ? nTiles : 2048u;
    hipLaunchKernelGGL(stage1FallbackKernel, dim3(blocks), dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag, todoList,
                       tilesPerSample);
}

// The screen over the whole segment, then the 32-bit fallback kernel for the blocks it handed on.
void launchScreen(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag, uint8_t * tileTodo,
                  unsigned * todoList, hipStream_t stream)
{
    const unsigned tilesPerSample = (seg.nWindows + kStage1Block - 1) / kStage1Block;
    const unsigned blocksPerSample = (seg.nWindows + kPackedWindows - 1) / kPackedWindows;
    dim3 grid((blocksPerSample + kScreenBlocks - 1) / kScreenBlocks, co.nSamples);
    hipLaunchKernelGGL(screenKernel, grid, dim3(kStage1Block), 0, stream, seg, co, flags, winFlag, tileTodo, todoList, tilesPerSample,
                       blocksPerSample);
}

void launchScreenFallback(const SegmentView & seg, const CohortView & co, int32_t * q3, uint8_t * flags, int * winFlag,
                          const unsigned * todoList, hipStream_t stream)
{
    const unsigned tilesPerSample = (seg.nWindows + kStage1Block - 1) / kStage1Block;
    const unsigned nTiles = tilesPerSample * co.nSamples;
    const unsigned blocks = nTiles < 2048u ? nTiles : 2048u;
    hipLaunchKernelGGL(stage1FallbackKernel, dim3(blocks), dim3(kStage1Block), 0, stream, seg, co, q3, flags, winFlag, todoList,
                       tilesPerSample);
}

// rest: room for 1 + nFlagged x ceil(nSamples / 64) words, rest[0] zeroed by the caller
void launchExactFlagged(const SegmentView & seg, const CohortView & co, const int * flagged, unsigned nFlagged,
                        const unsigned * nFlaggedDev, const int * ownWinFlag, const uint8_t * tileTodo, int32_t * q3, uint8_t * flags,
                        int * winFlag, unsigned * rest, hipStream_t stream)
{
    if (nFlagged == 0 || co.nSamples == 0)
        return;
    const unsigned tilesPerSample = (seg.nWindows + kStage1Block - 1) / kStage1Block;
    // y: the flagged windows as far as they are known (a capacity: 128 rows stride over the real number -- a launch of
    // capacity x tiles workgroups that leave at once cost 45 us per segment)
    const unsigned rows = nFlaggedDev ? min(nFlagged, 128u) : min(nFlagged, 65535u);
    hipLaunchKernelGGL(exactFlaggedKernel, dim3((co.nSamples + 255u) / 256u, rows), dim3(256), 0, stream, seg, co, flagged, nFlagged,
                       nFlaggedDev, ownWinFlag, tileTodo, tilesPerSample, q3, flags, winFlag, rest);
    hipLaunchKernelGGL(exactFlaggedRestKernel, dim3(256), dim3(64), 0, stream, seg, co, flagged, tileTodo, tilesPerSample, q3, flags,
                       winFlag, rest);
}

}  // namespace pd
