// Gunzip on the device (round 4, prototype of SURVEY 8(f) row f1's first stage): the gzip members of the profiles -- one per
// 10,000-bp index region and sample, 3-4 KB each, 200,000 of them per 200-kb segment of a 10,000-sample cohort -- are
// independent, so one LANE inflates one member (inflate_core.h), eight members to a wave: a segment is 25,000 waves of work. Host counterpart:
// profile_io.cpp's MemberSource (libdeflate / zlib, one member per call); reference: the zlib stream behind
// seqan/stream/iostream_zip_impl.h:336-349, fed from the index jump of insert_histogram_popdel.h:531-562.
//
// pd_inflate_members() takes host buffers, copies the compressed bytes up (page-locked staging, so that the copy runs at
// PCIe rate), inflates, copies the result back and reports the three times: what `popdel call` would pay per segment if
// its decode phase handed compressed members to the device instead of inflating them on the host threads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pinned_host.h"
#include "inflate_core.h"
#include "popdel_hip.h"
#include "pd_walk.h"

namespace {

// Workgroups of ONE wave that holds L members (one per lane, lanes L..63 absent): the decoders' tables and the code lengths
// of a dynamic block header take L x 582 bytes of LDS (inflate_core.h; L x 1,116 until round 6: twice the members in flight per CU). Workgroups stride over the
// batches of L members. What a small L buys: a wave executes the UNION of its lanes' paths through the decoder (literal /
// match / header / refill), so fewer members per wave is fewer instructions per wave, and the smaller table set lets more
// waves share a SIMD -- a wave here is a chain of dependent instructions with nothing of its own to hide their latencies.
// Measured on MI355X (tools/inflate_lanes_sweep.py, profiles/r05_j_inflate_lanes_sweep.md), 21,000 members (the chunk a decoder
// thread of `popdel call` hands over at 4,000 samples) / 200,000 members (a 10,000-sample segment): 64 per wave 9.5 / 52.6 ms,
// 32: 8.4 / 46.3, 16: 7.4 / 51.4, 8: 6.3 / 42.9, 4: 10.1 / 62.5 (too many waves for the chip to hold at once); with the
// eight-byte loads of the far copies (inflate_core.h) and the code lengths in LDS, 8 per wave: 5.6 / 38.7 ms. Round 6: the symbol loop
// rewritten so that the lanes of a wave share ONE instruction stream whatever their members hold (inflate_core.h, inflateCodes):
// the union is no longer what a wave pays for, and more members per wave are fewer instructions per member -- 8 per wave 3.73 /
// 25.4 ms, 16: 3.25 / 19.6, 32: - / 22.1, 4: - / 48.3.
template <unsigned L>
__global__ __launch_bounds__(L) void inflateMembersKernel(const uint8_t * __restrict__ data, const uint64_t * __restrict__ inOff,
                                                          uint32_t n, uint8_t * __restrict__ out,
                                                          const uint64_t * __restrict__ outOff, uint32_t * __restrict__ status,
                                                          uint64_t * __restrict__ outLen, unsigned nBatches)
{
    __shared__ uint8_t symbols[pdinf::kSymbols * L];
    __shared__ uint32_t highs[pdinf::kHighWords * L];
    __shared__ uint16_t metas[pdinf::kMeta * L];
    __shared__ uint8_t lens[pdinf::kLengthBytes * L];
    __shared__ uint16_t kinds[pdinf::kSymbolKinds];
    pdinf::StridedTablesOf<L> t;
    t.symbols = symbols + threadIdx.x;
    t.highs = highs + threadIdx.x;
    t.metas = metas + threadIdx.x;
    t.lengths = lens + threadIdx.x;
    t.kinds = kinds;
    for (unsigned i = threadIdx.x; i < pdinf::kSymbolKinds; i += L)
        kinds[i] = pdinf::symbolKind(i);
    __syncthreads();
    for (unsigned batch = blockIdx.x; batch < nBatches; batch += gridDim.x)
    {
        const unsigned m = batch * L + threadIdx.x;
        if (m >= n)
            continue;
        uint64_t got = 0, used = 0;
        uint32_t crc = 0;
        const uint64_t a = inOff[m], b = outOff[m];
        status[m] = pdinf::inflateGzipMember(data + a, inOff[m + 1] - a, out + b, outOff[m + 1] - b, &got, &used, t, nullptr, &crc);
        outLen[m] = got | (uint64_t)crc << 32;          // (the trailer's CRC rides along to crcMembersKernel, which takes it out again)
    }
}

// The CRC-32 of what the members inflated to, against their trailers (RFC 1952). Sixteen lanes per member: its output is cut into
// pieces of 512 bytes counted from its END (the first piece takes what is left over), a lane takes the CRC of a piece -- eight
// bytes per load, byte by byte through the table --, and the pieces are joined in order the way zlib's crc32_combine does it:
// crc(A B) = crc(A) * x^(8 |B|) + crc(B) over GF(2) modulo the CRC's polynomial, |B| = 512 for every piece but the first, so
// the factor is ONE constant. (One lane per member, the first form of this kernel, left the chip to 400 waves with a 5,000-byte
// chain each: 0.43 ms per 27,000 members, 0.66 in `popdel call`; in the symbol loop, where eight -- now sixteen -- lanes of a wave
// work, the same table look-ups cost a multiple of the issue slots.) Members the decoder refused keep their status.
constexpr uint32_t kCrcPoly = 0xedb88320u;
constexpr uint32_t crcMultiply(uint32_t a, uint32_t b)         // a * b mod the polynomial, bit-reflected like the CRC itself (zlib: multmodp)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m != 0u; m >>= 1)
    {
        if (a & m)
            p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}
constexpr uint32_t crcPowerOfX(uint64_t bits)                  // x^bits
{
    uint32_t p = 0x80000000u, square = 0x40000000u;            // x^0, x^1
    for (; bits != 0u; bits >>= 1)
    {
        if (bits & 1u)
            p = crcMultiply(square, p);
        square = crcMultiply(square, square);
    }
    return p;
}
constexpr unsigned kCrcPiece = 512, kCrcLanes = 16;
constexpr uint32_t kCrcShiftPiece = crcPowerOfX(8u * kCrcPiece);

__global__ __launch_bounds__(256) void crcMembersKernel(const uint8_t * __restrict__ out, const uint64_t * __restrict__ outOff, uint32_t n,
                                                        uint32_t * __restrict__ status, uint64_t * __restrict__ outLen)
{
    __shared__ uint32_t crcTab[256];
    crcTab[threadIdx.x] = pdinf::crcTableEntry(threadIdx.x);
    __syncthreads();
    const uint32_t m = blockIdx.x * (256u / kCrcLanes) + threadIdx.x / kCrcLanes, lane = threadIdx.x % kCrcLanes;
    if (m >= n)
        return;
    const uint64_t packed = outLen[m];
    const uint32_t len = (uint32_t)packed;
    const bool fine = status[m] == pdinf::kOk;
    const uint32_t nPieces = fine ? (len + kCrcPiece - 1u) / kCrcPiece : 0u;
    const uint32_t firstLen = nPieces ? len - kCrcPiece * (nPieces - 1u) : 0u;
    const uint8_t * p = out + outOff[m];
    uint32_t all = 0;                                           // (the CRC of nothing)
    for (uint32_t round = 0; round * kCrcLanes < nPieces; ++round)
    {
        const uint32_t j = round * kCrcLanes + lane;
        uint32_t c = 0xffffffffu;
        if (j < nPieces)
        {
            const uint8_t * q = p + (j == 0u ? 0u : firstLen + kCrcPiece * (j - 1u));
            const uint32_t pieceLen = j == 0u ? firstLen : kCrcPiece;
            uint32_t i = 0;
            for (; i + 8u <= pieceLen; i += 8u)
            {
                uint64_t v;
                __builtin_memcpy(&v, q + i, 8);
#pragma unroll
                for (unsigned k = 0; k < 8u; ++k, v >>= 8)
                    c = crcTab[(c ^ (uint32_t)v) & 0xffu] ^ (c >> 8);
            }
            for (; i < pieceLen; ++i)
                c = crcTab[(c ^ q[i]) & 0xffu] ^ (c >> 8);
        }
        c ^= 0xffffffffu;
        for (uint32_t k = 0; k < kCrcLanes && round * kCrcLanes + k < nPieces; ++k)
        {
            const uint32_t piece = __shfl(c, (int)k, (int)kCrcLanes);
            all = (round | k) == 0u ? piece : crcMultiply(kCrcShiftPiece, all) ^ piece;
        }
    }
    if (lane == 0u)
    {
        outLen[m] = len;
        if (fine && all != (uint32_t)(packed >> 32))
            status[m] = pdinf::kBadCrc;
    }
}

// Members per wave (4 | 8 | 16 | 32) and the waves of that shape a CU holds (the runtime's occupancy figure: LDS and VGPRs).
// Sixteen unless POPDEL_HIP_INFLATE_LANES says otherwise (measurements).
template <unsigned L>
unsigned wavesPerCuOf()
{
    static const unsigned n = [] {
        int blocks = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, inflateMembersKernel<L>, (int)L, 0) != hipSuccess || blocks < 1)
            blocks = 1;
        return (unsigned)blocks;
    }();
    return n;
}
unsigned wavesPerCu(unsigned lanes)
{
    return lanes == 4u ? wavesPerCuOf<4u>() : lanes == 8u ? wavesPerCuOf<8u>() : lanes == 16u ? wavesPerCuOf<16u>() : wavesPerCuOf<32u>();
}

unsigned membersPerWave()
{
    static const unsigned fixed = [] {
        const char * e = std::getenv("POPDEL_HIP_INFLATE_LANES");
        const unsigned v = e ? (unsigned)std::atoi(e) : 0u;
        return (v == 4u || v == 8u || v == 16u || v == 32u) ? v : 0u;
    }();
    return fixed ? fixed : 16u;
}

// (returns the number of workgroups launched)
unsigned launchInflate(unsigned nCu, hipStream_t st, const uint8_t * data, const uint64_t * inOff, uint32_t n, uint8_t * out,
                       const uint64_t * outOff, uint32_t * status, uint64_t * outLen)
{
    const unsigned lanes = membersPerWave();
    const unsigned nBatches = (n + lanes - 1u) / lanes;
    const unsigned groups = std::min(nBatches, wavesPerCu(lanes) * nCu);
    switch (lanes)
    {
    case 4u: hipLaunchKernelGGL(inflateMembersKernel<4u>, dim3(groups), dim3(4u), 0, st, data, inOff, n, out, outOff, status, outLen, nBatches); break;
    case 8u: hipLaunchKernelGGL(inflateMembersKernel<8u>, dim3(groups), dim3(8u), 0, st, data, inOff, n, out, outOff, status, outLen, nBatches); break;
    case 16u: hipLaunchKernelGGL(inflateMembersKernel<16u>, dim3(groups), dim3(16u), 0, st, data, inOff, n, out, outOff, status, outLen, nBatches); break;
    default: hipLaunchKernelGGL(inflateMembersKernel<32u>, dim3(groups), dim3(32u), 0, st, data, inOff, n, out, outOff, status, outLen, nBatches); break;
    }
    hipLaunchKernelGGL(crcMembersKernel, dim3((n + 256u / kCrcLanes - 1u) / (256u / kCrcLanes)), dim3(256u), 0, st, out, outOff, n, status, outLen);
    return groups;
}

// ---- parse + 256 -> 30 split of the inflated blocks (round 4) ---------------------------------------------------------------
// What the loader does with the bytes of a block (reference: readWindow(), popdel_profile/window_podel.h:211-251, then
// convertWindow(), :314-418, per 256-bp window; host counterpart: readWindowHead() + readWindowBodyFlat(), profile_io.cpp):
// a block is a run of 256-bp windows {chrom u32, beginPos u32, n u32, n x {offset u8, deviation i32}} (one read group per
// file: the rule; files with several are parsed on the host), a record's position is beginPos + offset, and the records of a
// window -- position-sorted -- are cut into the 30-bp windows of the global grid that have content. One lane per block, two
// passes: counts (-> offsets by a scan), then the arrays the host's replay reads -- per 256-bp window {chrom, beginPos, first
// 30-bp window, their number}, per 30-bp window {chrom, beginPos, first record} and its record count, the records {pos, dev}.
// Anything the host's reader would stumble over (a window cut off by the end of the block, a window without records, records
// out of order or spread over more than ten 30-bp windows) flags the block: its sample is parsed by the host, which reports
// what the reference reports.
struct ParsedWindow     // (= pdh::DecodedRaw)
{
    int32_t  chrom, beginPos;
    uint32_t splitFailed, firstWin, nWin;
};
// Lean form (round 6; pd_inflater_run_summarized with lean != 0): per 256-bp window where its records lie and the positions of
// the first and the last -- the 30-bp windows are the runs of equal position / 30 among them (the first record of a window
// anchors the 30-grid at a multiple of 30, convertWindow), which the host reads off the records where it needs them: at the two
// ends of what a round files. No call windows, no record counts per window: a third less to copy back.
struct LeanWindow       // (= pdh::LeanRaw)
{
    int32_t  chrom, beginPos;
    uint32_t firstRec, nRecs, firstPos, lastPos;
};
struct CallWindow       // (= pdh::FlatWindow)
{
    int32_t  chrom, beginPos;
    uint32_t firstRec;
};
struct BlockCounts
{
    uint32_t raws, wins, recs, flags;
};

// (unaligned: the hardware takes them; eight bytes at p cover a record {offset u8, deviation i32} and three bytes of what
// follows -- the buffers end with slack, growDevice())
__device__ __forceinline__ uint32_t loadU32(const uint8_t * p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ uint64_t loadU64(const uint8_t * p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Blocks per wave of the two walking kernels: a lane walks its block alone, record after record -- a chain of loads that a wave
// of 64 such lanes spreads over 64 cache lines per instruction; sixteen per wave give the chip four times the waves to
// switch between (parseCount + parseWrite per 1,024-sample chunk: 0.56 + 1.03 ms with 64).
constexpr unsigned kParseLanes = 16;

template <bool Write, bool Lean = false>
__device__ __forceinline__ BlockCounts walkBlock(const uint8_t * p, const uint8_t * end, ParsedWindow * raws, CallWindow * wins,
                                                 uint32_t * counts, uint2 * recs, uint32_t winBase, uint32_t recBase, uint2 * span = nullptr)
{
    BlockCounts c = {0, 0, 0, 0};
    while (p < end)
    {
        if (end - p < 12)
        {
            c.flags = 1;
            break;
        }
        const uint64_t head = loadU64(p);
        const uint32_t chrom = (uint32_t)head, begin = (uint32_t)(head >> 32), n = loadU32(p + 8);
        if (n == 0 || (uint64_t)(end - p - 12) < (uint64_t)n * 5u)
        {
            c.flags = 1;
            break;
        }
        p += 12;
        const uint32_t base = ((begin + p[0]) / 30u) * 30u;
        uint32_t prev = 0, boundary = 0, firstWin = c.wins, inWin = 0;
        const uint32_t firstPos = begin + p[0];
        for (uint32_t i = 0; i < n; ++i, p += 5)
        {
            const uint64_t rec = loadU64(p);
            const uint32_t pos = begin + ((uint32_t)rec & 0xffu);
            if (pos < prev)
                c.flags = 1;
            prev = pos;
            if (pos >= boundary)
            {
                const uint32_t winBegin = base + ((pos - base) / 30u) * 30u;
                if (Write && !Lean)
                {
                    if (inWin)
                        counts[c.wins - 1] = inWin;
                    wins[c.wins] = CallWindow{(int32_t)chrom, (int32_t)winBegin, recBase + c.recs + i};
                }
                inWin = 0;
                ++c.wins;
                boundary = winBegin + 30u;
            }
            ++inWin;
            if (Write)
                recs[c.recs + i] = make_uint2(pos, (uint32_t)(rec >> 8));
        }
        if (Write && !Lean)
        {
            counts[c.wins - 1] = inWin;
            raws[c.raws] = ParsedWindow{(int32_t)chrom, (int32_t)begin, 0u, winBase + firstWin, c.wins - firstWin};
        }
        if (Write && Lean)
            reinterpret_cast<LeanWindow *>(raws)[c.raws] = LeanWindow{(int32_t)chrom, (int32_t)begin, recBase + c.recs, n, firstPos, prev};
        if ((prev - base) / 30u >= 10u)
            c.flags = 1;
        if (!Write && span)
        {
            span->x = min(span->x, firstPos);
            span->y = max(span->y, prev);
        }
        ++c.raws;
        c.recs += n;
    }
    return c;
}

__global__ __launch_bounds__(kParseLanes) void parseCountKernel(const uint8_t * __restrict__ data, const uint64_t * __restrict__ outOff,
                                                       const uint64_t * __restrict__ outLen, const uint32_t * __restrict__ status,
                                                       const uint8_t * __restrict__ wanted, uint32_t n, BlockCounts * __restrict__ counts,
                                                       uint2 * __restrict__ spans)
{
    const uint32_t m = blockIdx.x * kParseLanes + threadIdx.x;
    if (m >= n)
        return;
    BlockCounts c = {0, 0, 0, 2};
    uint2 span = make_uint2(0xffffffffu, 0u);   // (smallest, largest start position among the block's records)
    if (wanted[m] && status[m] == 0 && outLen[m] == outOff[m + 1] - outOff[m])
        c = walkBlock<false>(data + outOff[m], data + outOff[m] + outLen[m], nullptr, nullptr, nullptr, nullptr, 0, 0, &span);
    if (c.flags)
        c.raws = c.wins = c.recs = 0;           // (nothing is written for a flagged block)
    counts[m] = c;
    if (spans)
        spans[m] = span;
}

// the kept blocks' records packed: range r = records [src[r], src[r] + n[r]) of the chunk's array -> packed[dst[r] ..)
struct PackRange
{
    uint64_t src, dst;
    uint32_t n, pad;
};
__global__ __launch_bounds__(256) void packRecordsKernel(const uint2 * __restrict__ recs, const PackRange * __restrict__ ranges, uint2 * __restrict__ packed)
{
    const PackRange r = ranges[blockIdx.x];
    const uint2 * from = recs + r.src;
    uint2 * to = packed + r.dst;
    for (uint32_t i = threadIdx.x; i < r.n; i += 256u)
        to[i] = from[i];
}

// exclusive scans of the three counts over the blocks (offsets[k][0 .. n]); one workgroup of 1,024: every thread adds up a
// contiguous stretch of blocks, the stretches' sums are scanned in LDS, every thread writes its stretch's offsets (round 6: the
// 256-wide form walked a chunk's 28,000 blocks in 110 rounds of three LDS scans, 0.31 ms per chunk)
constexpr unsigned kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void parseScanKernel(const BlockCounts * __restrict__ counts, uint32_t n, uint64_t * __restrict__ rawOff,
                                                                uint64_t * __restrict__ winOff, uint64_t * __restrict__ recOff)
{
    __shared__ uint64_t sh[3][kScanThreads];
    const uint32_t per = (n + kScanThreads - 1u) / kScanThreads;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint64_t v[3] = {0, 0, 0};
    for (uint32_t e = lo; e < hi; ++e)
    {
        v[0] += counts[e].raws;
        v[1] += counts[e].wins;
        v[2] += counts[e].recs;
    }
    for (int k = 0; k < 3; ++k)
        sh[k][threadIdx.x] = v[k];
    __syncthreads();
    for (unsigned o = 1; o < kScanThreads; o <<= 1)
    {
        uint64_t t[3];
        for (int k = 0; k < 3; ++k)
            t[k] = threadIdx.x >= o ? sh[k][threadIdx.x - o] : 0u;
        __syncthreads();
        for (int k = 0; k < 3; ++k)
            sh[k][threadIdx.x] += t[k];
        __syncthreads();
    }
    uint64_t at[3];
    for (int k = 0; k < 3; ++k)
        at[k] = sh[k][threadIdx.x] - v[k];          // (exclusive: what lies before this thread's stretch)
    for (uint32_t e = lo; e < hi; ++e)
    {
        rawOff[e] = at[0];
        winOff[e] = at[1];
        recOff[e] = at[2];
        at[0] += counts[e].raws;
        at[1] += counts[e].wins;
        at[2] += counts[e].recs;
    }
    if (threadIdx.x == kScanThreads - 1u)
    {
        rawOff[n] = sh[0][kScanThreads - 1u];
        winOff[n] = sh[1][kScanThreads - 1u];
        recOff[n] = sh[2][kScanThreads - 1u];
    }
}

__global__ __launch_bounds__(kParseLanes) void parseWriteKernel(const uint8_t * __restrict__ data, const uint64_t * __restrict__ outOff,
                                                       const uint64_t * __restrict__ outLen, const BlockCounts * __restrict__ counts,
                                                       const uint32_t * __restrict__ sampleFirst, const uint64_t * __restrict__ rawOff,
                                                       const uint64_t * __restrict__ winOff, const uint64_t * __restrict__ recOff, uint32_t n,
                                                       ParsedWindow * __restrict__ raws, CallWindow * __restrict__ wins,
                                                       uint32_t * __restrict__ winCounts, uint2 * __restrict__ recs, uint32_t lean)
{
    const uint32_t m = blockIdx.x * kParseLanes + threadIdx.x;
    if (m >= n || counts[m].flags)
        return;
    // indices are relative to the block's SAMPLE (its first block is sampleFirst[m])
    const uint32_t f = sampleFirst[m];
    if (lean)
        walkBlock<true, true>(data + outOff[m], data + outOff[m] + outLen[m],
                              reinterpret_cast<ParsedWindow *>(reinterpret_cast<LeanWindow *>(raws) + rawOff[m]), nullptr, nullptr,
                              recs + recOff[m], 0u, (uint32_t)(recOff[m] - recOff[f]));
    else
        walkBlock<true>(data + outOff[m], data + outOff[m] + outLen[m], raws + rawOff[m], wins + winOff[m], winCounts + winOff[m],
                        recs + recOff[m], (uint32_t)(winOff[m] - winOff[f]), (uint32_t)(recOff[m] - recOff[f]));
}

// ---- the parsed records summarised block by block for the loader's tables (round 6; pd_inflater_run_summarized) ----------------
// One workgroup per block: its records are counted by the window they begin in and the window they end in (LDS, then one
// saturating byte per window), with the extremes and the two order flags. The host's twin: pdh::summarizeRun() (chrom_profile.cpp).
constexpr unsigned kSummaryThreads = 128;
constexpr unsigned kSummaryRoomMax = 12000;         // windows per block, starts and ends together: 4 bytes each in LDS

__global__ __launch_bounds__(kSummaryThreads) void summarizeBlocksKernel(const uint2 * __restrict__ recs, const uint64_t * __restrict__ recOff,
                                                                         const BlockCounts * __restrict__ blocks, const int32_t * __restrict__ innerOffset,
                                                                         uint32_t startsRoom, uint32_t endsRoom,
                                                                         pd_block_summary * __restrict__ summaries, uint8_t * __restrict__ counts)
{
    extern __shared__ unsigned shCounts[];          // [startsRoom + endsRoom]
    __shared__ unsigned shMinEnd, shMaxEnd, shFlags;
    const uint32_t m = blockIdx.x;
    const uint2 * r = recs + recOff[m];
    const unsigned n = blocks[m].flags ? 0u : (unsigned)(recOff[m + 1] - recOff[m]);
    const int inner = innerOffset[m];
    const unsigned room = startsRoom + endsRoom;
    auto endBinOf = [&](uint2 rec) {
        const int d = (int)rec.y + inner;
        return (rec.x + (d > 0 ? (unsigned)d : 0u)) / 30u;
    };
    pd_block_summary out;
    out.n = n;
    out.first_pos = out.last_pos = out.bin_base = out.n_bins = out.min_end_bin = out.max_end_bin = out.first_end_bin = out.last_end_bin = out.flags = 0;
    uint8_t * row = counts + (size_t)m * room;
    if (n == 0)
    {
        if (threadIdx.x == 0)
            summaries[m] = out;
        return;                                     // (its counts are not looked at)
    }
    if (threadIdx.x == 0)
    {
        shMinEnd = 0xffffffffu;
        shMaxEnd = 0;
        shFlags = 1u | 2u;
    }
    for (unsigned b = threadIdx.x; b < room; b += kSummaryThreads)
        shCounts[b] = 0;
    __syncthreads();
    const unsigned base = r[0].x / 30u;
    unsigned myMin = 0xffffffffu, myMax = 0, myFlags = 0;       // myFlags: bits to CLEAR (1, 2) or to set (8)
    for (unsigned i = threadIdx.x; i < n; i += kSummaryThreads)
    {
        const uint2 rec = r[i];
        const unsigned sb = rec.x / 30u, eb = endBinOf(rec);
        if (i > 0)
        {
            const uint2 before = r[i - 1];
            if (rec.x < before.x)
                myFlags |= 1u;
            if (eb < endBinOf(before))
                myFlags |= 2u;
        }
        myMin = min(myMin, eb);
        myMax = max(myMax, eb);
        if (sb < base || sb - base >= startsRoom || eb < base || eb - base >= endsRoom)
            myFlags |= 8u;
        else
        {
            atomicAdd(&shCounts[sb - base], 1u);
            atomicAdd(&shCounts[startsRoom + eb - base], 1u);
        }
    }
    atomicMin(&shMinEnd, myMin);
    atomicMax(&shMaxEnd, myMax);
    if (myFlags & 3u)
        atomicAnd(&shFlags, ~(myFlags & 3u));
    if (myFlags & 8u)
        atomicOr(&shFlags, 8u);
    __syncthreads();
    bool saturated = false;
    for (unsigned b = threadIdx.x; b < room; b += kSummaryThreads)
    {
        const unsigned c = shCounts[b];
        saturated = saturated || c >= 255u;
        row[b] = (uint8_t)min(c, 255u);
    }
    if (saturated)
        atomicOr(&shFlags, 4u);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        out.first_pos = r[0].x;
        out.last_pos = r[n - 1].x;
        out.bin_base = base;
        out.min_end_bin = shMinEnd;
        out.max_end_bin = shMaxEnd;
        out.first_end_bin = endBinOf(r[0]);
        out.last_end_bin = endBinOf(r[n - 1]);
        out.flags = shFlags;
        out.n_bins = (shFlags & 8u) ? 0u : shMaxEnd - base + 1u;
        summaries[m] = out;
    }
}

struct Fail
{
    std::string what;
};
#define PD_INF_HIP(expr)                                                                                     \
    do                                                                                                        \
    {                                                                                                         \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            throw Fail{std::string(#expr) + ": " + hipGetErrorString(e_)};                                    \
    } while (0)

// (per calling thread: several inflaters run on threads of their own, each reads the message of ITS last failed call)
thread_local std::string g_inflateError;

}  // namespace

// A persistent inflater: stream, device buffers and page-locked staging on both sides kept from call to call (what the
// loader of `popdel call` uses: a chunk of samples' members per call, several inflaters in turn so that one chunk is on the
// device while the next is gathered).
struct pd_inflater
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t timeBegin = nullptr, timeEnd = nullptr;   // around the kernels when the caller asks for kernel_ms (created on first use)
    hipEvent_t finished = nullptr;      // blocking-sync event: the calling thread sleeps instead of spinning on the stream (the host's
                                        // cores are what `popdel call` is short of)
    unsigned nCu = 1;
    void *dData = nullptr, *dOut = nullptr, *dInOff = nullptr, *dOutOff = nullptr, *dStatus = nullptr, *dLen = nullptr;
    size_t capData = 0, capOut = 0, capMembers = 0;
    void *hIn = nullptr, *hOut = nullptr, *hMeta = nullptr;     // page-locked: compressed bytes in, inflated bytes out, status + lengths
    size_t capHIn = 0, capHOut = 0, capHMeta = 0;
    // parsed blocks (pd_inflater_run_parsed): per-block tables and the four arrays, device and page-locked host
    void *dWanted = nullptr, *dSampleFirst = nullptr, *dCounts = nullptr, *dOffsets = nullptr, *dRaws = nullptr, *dWins = nullptr, *dWinCounts = nullptr, *dRecs = nullptr;
    size_t capWanted = 0, capSampleFirst = 0, capCounts = 0, capOffsets = 0, capRaws = 0, capWins = 0, capWinCounts = 0, capRecs = 0;
    void *hOffsets = nullptr, *hFlags = nullptr, *hRaws = nullptr, *hWins = nullptr, *hWinCounts = nullptr, *hRecs = nullptr;
    size_t capHOffsets = 0, capHFlags = 0, capHRaws = 0, capHWins = 0, capHWinCounts = 0, capHRecs = 0;
    uint64_t nRecsLast = 0;             // records the last parsed chunk left in dRecs
    uint64_t runId = 0;                 // parsed chunks run so far
    // records copied back for some blocks only (pd_summarize_options::keep_below / keep_from)
    void *dSpans = nullptr, *hSpans = nullptr, *dRanges = nullptr, *hRanges = nullptr, *dPacked = nullptr;
    size_t capDSpans = 0, capHSpans = 0, capDRanges = 0, capHRanges = 0, capDPacked = 0;
    std::vector<uint64_t> recordIndex;
    // blocks summarised (pd_inflater_run_summarized): inner offsets up, summaries and counts back
    void *dInner = nullptr, *dSummaries = nullptr, *dBlockCounts = nullptr, *hSummaries = nullptr, *hBlockCounts = nullptr;
    size_t capDInner = 0, capDSummaries = 0, capDBlockCounts = 0, capHSummaries = 0, capHBlockCounts = 0;
    std::string error;
};

namespace {

void growDevice(void *& p, size_t & cap, size_t want)
{
    if (want <= cap)
        return;
    if (p)
        (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const size_t take = want + want / 4 + 4096;
    PD_INF_HIP(hipMalloc(&p, take));
    cap = take;
}

void growPinned(void *& p, size_t & cap, size_t want)
{
    if (want <= cap)
        return;
    if (p)
        pd::pinnedFree(p);
    p = nullptr;
    cap = 0;
    const size_t take = want + want / 4 + 4096;
    p = pd::pinnedAlloc(take);
    if (!p)
        throw Fail{"page-locked host memory: " + std::to_string(take) + " bytes not to be had"};
    cap = take;
}

}  // namespace

namespace pd {
const uint2 * inflaterDeviceRecords(pd_inflater * f, uint64_t & count, int & device, uint64_t & runId)
{
    runId = f ? f->runId : 0;
    count = f ? f->nRecsLast : 0;
    device = f ? f->device : -1;
    return f ? static_cast<const uint2 *>(f->dRecs) : nullptr;
}
}  // namespace pd

extern "C" pd_inflater * pd_inflater_create(int device)
{
    pd_inflater * f = new (std::nothrow) pd_inflater();
    if (!f)
        return nullptr;
    try
    {
        f->device = device;
        PD_INF_HIP(hipSetDevice(device));
        int nCu = 0;
        PD_INF_HIP(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, device));
        f->nCu = (unsigned)std::max(1, nCu);
        PD_INF_HIP(hipStreamCreate(&f->stream));
        PD_INF_HIP(hipEventCreateWithFlags(&f->finished, hipEventBlockingSync | hipEventDisableTiming));
    }
    catch (const Fail & e)
    {
        g_inflateError = "pd_inflater_create: " + e.what;
        pd_inflater_destroy(f);
        return nullptr;
    }
    return f;
}

extern "C" void pd_inflater_destroy(pd_inflater * f)
{
    if (!f)
        return;
    (void)hipSetDevice(f->device);
    for (void * p : {f->dData, f->dOut, f->dInOff, f->dOutOff, f->dStatus, f->dLen, f->dWanted, f->dSampleFirst, f->dCounts, f->dOffsets, f->dRaws,
                     f->dWins, f->dWinCounts, f->dRecs, f->dInner, f->dSummaries, f->dBlockCounts, f->dSpans, f->dRanges, f->dPacked})
        if (p)
            (void)hipFree(p);
    for (void * p : {f->hIn, f->hOut, f->hMeta, f->hOffsets, f->hFlags, f->hRaws, f->hWins, f->hWinCounts, f->hRecs, f->hSummaries, f->hBlockCounts, f->hSpans, f->hRanges})
        if (p)
            pd::pinnedFree(p);
    for (hipEvent_t e : {f->finished, f->timeBegin, f->timeEnd})
        if (e)
            (void)hipEventDestroy(e);
    if (f->stream)
        (void)hipStreamDestroy(f->stream);
    delete f;
}

extern "C" uint8_t * pd_inflater_input(pd_inflater * f, uint64_t bytes)
{
    if (!f)
        return nullptr;
    try
    {
        PD_INF_HIP(hipSetDevice(f->device));
        growPinned(f->hIn, f->capHIn, bytes + pdinf::kInputSlack);
    }
    catch (const Fail & e)
    {
        f->error = "pd_inflater_input: " + e.what;
        return nullptr;
    }
    return static_cast<uint8_t *>(f->hIn);
}

static int runInflater(pd_inflater * f, const uint64_t * member_offsets, uint32_t n_members, const uint64_t * out_offsets,
                       const uint32_t * sample_first, const uint8_t * parse_member, const uint8_t ** out, const uint32_t ** status,
                       const uint64_t ** out_lengths, pd_parsed_blocks * parsed, double * kernel_ms, const int32_t * inner_offset = nullptr,
                       uint32_t starts_room = 0, uint32_t ends_room = 0, pd_block_summaries * summaries = nullptr, uint32_t lean = 0,
                       const uint32_t * keep_below = nullptr, const uint32_t * keep_from = nullptr)
{
    if (!f || !member_offsets || !out_offsets || !out || !status || !out_lengths || n_members == 0 || (parsed && (!sample_first || !parse_member)))
        return 1;
    if (summaries && (!parsed || !inner_offset || starts_room == 0 || ends_room == 0 || starts_room + ends_room > kSummaryRoomMax))
    {
        f->error = "pd_inflater_run_summarized: invalid argument (no inner offsets, or more than 12,000 windows of room per block)";
        return 1;
    }
    try
    {
        PD_INF_HIP(hipSetDevice(f->device));
        const uint64_t inBytes = member_offsets[n_members], outBytes = out_offsets[n_members];
        if (inBytes + pdinf::kInputSlack > f->capHIn)
            throw Fail{"the members were not gathered into pd_inflater_input()'s buffer"};
        growDevice(f->dData, f->capData, inBytes + pdinf::kInputSlack);
        growDevice(f->dOut, f->capOut, outBytes + pdinf::kOutputSlack);      // (the gunzip kernel's match loads; the parser reads a record's five bytes with one eight-byte load, walkBlock())
        if (n_members + 1 > f->capMembers)
        {
            size_t a = 0, b = 0, c = 0, d = 0;
            for (void * p : {f->dInOff, f->dOutOff, f->dStatus, f->dLen})
                if (p)
                    (void)hipFree(p);
            f->dInOff = f->dOutOff = f->dStatus = f->dLen = nullptr;
            f->capMembers = 0;
            const size_t m = n_members + n_members / 4 + 1024;
            growDevice(f->dInOff, a, m * sizeof(uint64_t));
            growDevice(f->dOutOff, b, m * sizeof(uint64_t));
            growDevice(f->dStatus, c, m * sizeof(uint32_t));
            growDevice(f->dLen, d, m * sizeof(uint64_t));
            f->capMembers = m;
        }
        growPinned(f->hMeta, f->capHMeta, static_cast<size_t>(n_members) * (sizeof(uint64_t) + sizeof(uint32_t)) + 16);
        hipStream_t st = f->stream;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (kernel_ms)
        {
            if (!f->timeBegin)
                PD_INF_HIP(hipEventCreate(&f->timeBegin));
            if (!f->timeEnd)
                PD_INF_HIP(hipEventCreate(&f->timeEnd));
            e0 = f->timeBegin;
            e1 = f->timeEnd;
        }
        PD_INF_HIP(hipMemcpyAsync(f->dInOff, member_offsets, (n_members + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        PD_INF_HIP(hipMemcpyAsync(f->dOutOff, out_offsets, (n_members + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        PD_INF_HIP(hipMemcpyAsync(f->dData, f->hIn, inBytes, hipMemcpyHostToDevice, st));
        if (e0)
            PD_INF_HIP(hipEventRecord(e0, st));
        launchInflate(f->nCu, st, static_cast<const uint8_t *>(f->dData),
                      static_cast<const uint64_t *>(f->dInOff), n_members, static_cast<uint8_t *>(f->dOut),
                      static_cast<const uint64_t *>(f->dOutOff), static_cast<uint32_t *>(f->dStatus), static_cast<uint64_t *>(f->dLen));
        PD_INF_HIP(hipGetLastError());
        if (e1)
            PD_INF_HIP(hipEventRecord(e1, st));
        uint64_t * hLen = static_cast<uint64_t *>(f->hMeta);
        uint32_t * hStatus = reinterpret_cast<uint32_t *>(hLen + n_members);
        bool needBytes = true;
        if (parsed)
        {
            // counts per block -> offsets -> (sizes to the host) -> the arrays
            growDevice(f->dWanted, f->capWanted, n_members);
            growDevice(f->dSampleFirst, f->capSampleFirst, static_cast<size_t>(n_members) * sizeof(uint32_t));
            growDevice(f->dCounts, f->capCounts, static_cast<size_t>(n_members) * sizeof(BlockCounts));
            growDevice(f->dOffsets, f->capOffsets, 3 * (static_cast<size_t>(n_members) + 1) * sizeof(uint64_t));
            growPinned(f->hOffsets, f->capHOffsets, 3 * (static_cast<size_t>(n_members) + 1) * sizeof(uint64_t));
            growPinned(f->hFlags, f->capHFlags, static_cast<size_t>(n_members) * sizeof(BlockCounts));
            PD_INF_HIP(hipMemcpyAsync(f->dWanted, parse_member, n_members, hipMemcpyHostToDevice, st));
            PD_INF_HIP(hipMemcpyAsync(f->dSampleFirst, sample_first, static_cast<size_t>(n_members) * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            uint64_t * dRawOff = static_cast<uint64_t *>(f->dOffsets), *dWinOff = dRawOff + n_members + 1, *dRecOff = dWinOff + n_members + 1;
            const bool partial = summaries && keep_below && keep_from;
            if (partial)
            {
                growDevice(f->dSpans, f->capDSpans, static_cast<size_t>(n_members) * sizeof(uint2));
                growPinned(f->hSpans, f->capHSpans, static_cast<size_t>(n_members) * sizeof(uint2));
            }
            hipLaunchKernelGGL(parseCountKernel, dim3((n_members + kParseLanes - 1) / kParseLanes), dim3(kParseLanes), 0, st, static_cast<const uint8_t *>(f->dOut),
                               static_cast<const uint64_t *>(f->dOutOff), static_cast<const uint64_t *>(f->dLen),
                               static_cast<const uint32_t *>(f->dStatus), static_cast<const uint8_t *>(f->dWanted), n_members,
                               static_cast<BlockCounts *>(f->dCounts), partial ? static_cast<uint2 *>(f->dSpans) : nullptr);
            hipLaunchKernelGGL(parseScanKernel, dim3(1), dim3(kScanThreads), 0, st, static_cast<const BlockCounts *>(f->dCounts), n_members, dRawOff, dWinOff, dRecOff);
            PD_INF_HIP(hipGetLastError());
            PD_INF_HIP(hipMemcpyAsync(f->hOffsets, f->dOffsets, 3 * (static_cast<size_t>(n_members) + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
            PD_INF_HIP(hipMemcpyAsync(f->hFlags, f->dCounts, static_cast<size_t>(n_members) * sizeof(BlockCounts), hipMemcpyDeviceToHost, st));
            if (partial)
                PD_INF_HIP(hipMemcpyAsync(f->hSpans, f->dSpans, static_cast<size_t>(n_members) * sizeof(uint2), hipMemcpyDeviceToHost, st));
            PD_INF_HIP(hipEventRecord(f->finished, st));
            PD_INF_HIP(hipEventSynchronize(f->finished));
            const uint64_t * hRawOff = static_cast<const uint64_t *>(f->hOffsets), *hWinOff = hRawOff + n_members + 1, *hRecOff = hWinOff + n_members + 1;
            const uint64_t nRaws = hRawOff[n_members], nWins = hWinOff[n_members], nRecs = hRecOff[n_members];
            f->nRecsLast = nRecs;
            ++f->runId;
            const size_t rawBytes = lean ? sizeof(LeanWindow) : sizeof(ParsedWindow);
            growDevice(f->dRaws, f->capRaws, std::max<uint64_t>(1, nRaws) * rawBytes);
            growDevice(f->dRecs, f->capRecs, std::max<uint64_t>(1, nRecs) * sizeof(uint2));
            growPinned(f->hRaws, f->capHRaws, std::max<uint64_t>(1, nRaws) * rawBytes);
            // which blocks' records go back: all, or those with a record below keep_below / at or beyond keep_from
            uint64_t nBack = nRecs;
            std::vector<PackRange> ranges;
            if (partial)
            {
                const uint2 * span = static_cast<const uint2 *>(f->hSpans);
                f->recordIndex.assign(n_members, ~static_cast<uint64_t>(0));
                nBack = 0;
                for (uint32_t m = 0; m < n_members; ++m)
                {
                    const uint64_t n = hRecOff[m + 1] - hRecOff[m];
                    if (n == 0 || !(span[m].x < keep_below[m] || span[m].y >= keep_from[m]))
                        continue;
                    f->recordIndex[m] = nBack;
                    if (!ranges.empty() && ranges.back().src + ranges.back().n == hRecOff[m] && ranges.back().n + n < 0x7fffffffu)
                        ranges.back().n += static_cast<uint32_t>(n);
                    else
                        ranges.push_back(PackRange{hRecOff[m], nBack, static_cast<uint32_t>(n), 0u});
                    nBack += n;
                }
            }
            growPinned(f->hRecs, f->capHRecs, std::max<uint64_t>(1, nBack) * sizeof(uint2));
            if (!lean)
            {
                growDevice(f->dWins, f->capWins, std::max<uint64_t>(1, nWins) * sizeof(CallWindow));
                growDevice(f->dWinCounts, f->capWinCounts, std::max<uint64_t>(1, nWins) * sizeof(uint32_t));
                growPinned(f->hWins, f->capHWins, std::max<uint64_t>(1, nWins) * sizeof(CallWindow));
                growPinned(f->hWinCounts, f->capHWinCounts, std::max<uint64_t>(1, nWins) * sizeof(uint32_t));
            }
            hipLaunchKernelGGL(parseWriteKernel, dim3((n_members + kParseLanes - 1) / kParseLanes), dim3(kParseLanes), 0, st, static_cast<const uint8_t *>(f->dOut),
                               static_cast<const uint64_t *>(f->dOutOff), static_cast<const uint64_t *>(f->dLen),
                               static_cast<const BlockCounts *>(f->dCounts), static_cast<const uint32_t *>(f->dSampleFirst), dRawOff, dWinOff, dRecOff,
                               n_members, static_cast<ParsedWindow *>(f->dRaws), static_cast<CallWindow *>(f->dWins),
                               static_cast<uint32_t *>(f->dWinCounts), static_cast<uint2 *>(f->dRecs), lean);
            PD_INF_HIP(hipGetLastError());
            PD_INF_HIP(hipMemcpyAsync(f->hRaws, f->dRaws, nRaws * rawBytes, hipMemcpyDeviceToHost, st));
            if (!lean)
            {
                PD_INF_HIP(hipMemcpyAsync(f->hWins, f->dWins, nWins * sizeof(CallWindow), hipMemcpyDeviceToHost, st));
                PD_INF_HIP(hipMemcpyAsync(f->hWinCounts, f->dWinCounts, nWins * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            }
            if (!partial)
                PD_INF_HIP(hipMemcpyAsync(f->hRecs, f->dRecs, nRecs * sizeof(uint2), hipMemcpyDeviceToHost, st));
            else if (!ranges.empty())
            {
                growDevice(f->dRanges, f->capDRanges, ranges.size() * sizeof(PackRange));
                growPinned(f->hRanges, f->capHRanges, ranges.size() * sizeof(PackRange));
                growDevice(f->dPacked, f->capDPacked, nBack * sizeof(uint2));
                std::memcpy(f->hRanges, ranges.data(), ranges.size() * sizeof(PackRange));
                PD_INF_HIP(hipMemcpyAsync(f->dRanges, f->hRanges, ranges.size() * sizeof(PackRange), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(packRecordsKernel, dim3(static_cast<unsigned>(ranges.size())), dim3(256), 0, st, static_cast<const uint2 *>(f->dRecs),
                                   static_cast<const PackRange *>(f->dRanges), static_cast<uint2 *>(f->dPacked));
                PD_INF_HIP(hipGetLastError());
                PD_INF_HIP(hipMemcpyAsync(f->hRecs, f->dPacked, nBack * sizeof(uint2), hipMemcpyDeviceToHost, st));
            }
            if (summaries)
            {
                const size_t room = static_cast<size_t>(starts_room) + ends_room;
                growDevice(f->dInner, f->capDInner, static_cast<size_t>(n_members) * sizeof(int32_t));
                growDevice(f->dSummaries, f->capDSummaries, static_cast<size_t>(n_members) * sizeof(pd_block_summary));
                growDevice(f->dBlockCounts, f->capDBlockCounts, static_cast<size_t>(n_members) * room);
                growPinned(f->hSummaries, f->capHSummaries, static_cast<size_t>(n_members) * sizeof(pd_block_summary));
                growPinned(f->hBlockCounts, f->capHBlockCounts, static_cast<size_t>(n_members) * room);
                PD_INF_HIP(hipMemcpyAsync(f->dInner, inner_offset, static_cast<size_t>(n_members) * sizeof(int32_t), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(summarizeBlocksKernel, dim3(n_members), dim3(kSummaryThreads), room * sizeof(unsigned), st,
                                   static_cast<const uint2 *>(f->dRecs), dRecOff, static_cast<const BlockCounts *>(f->dCounts),
                                   static_cast<const int32_t *>(f->dInner), starts_room, ends_room, static_cast<pd_block_summary *>(f->dSummaries),
                                   static_cast<uint8_t *>(f->dBlockCounts));
                PD_INF_HIP(hipGetLastError());
                PD_INF_HIP(hipMemcpyAsync(f->hSummaries, f->dSummaries, static_cast<size_t>(n_members) * sizeof(pd_block_summary), hipMemcpyDeviceToHost, st));
                PD_INF_HIP(hipMemcpyAsync(f->hBlockCounts, f->dBlockCounts, static_cast<size_t>(n_members) * room, hipMemcpyDeviceToHost, st));
                summaries->block_summaries = static_cast<const pd_block_summary *>(f->hSummaries);
                summaries->block_counts = static_cast<const uint8_t *>(f->hBlockCounts);
                summaries->starts_room = starts_room;
                summaries->ends_room = ends_room;
                summaries->record_index = partial ? f->recordIndex.data() : nullptr;
            }
            // the inflated bytes themselves travel only if some block is left to the host's parser
            const BlockCounts * fl = static_cast<const BlockCounts *>(f->hFlags);
            needBytes = false;
            for (uint32_t m = 0; m < n_members && !needBytes; ++m)
                needBytes = fl[m].flags != 0;
            parsed->profile_windows = f->hRaws;
            parsed->call_windows = lean ? nullptr : f->hWins;
            parsed->call_window_records = lean ? nullptr : static_cast<const uint32_t *>(f->hWinCounts);
            parsed->records = f->hRecs;
            parsed->profile_window_offsets = hRawOff;
            parsed->call_window_offsets = hWinOff;
            parsed->record_offsets = hRecOff;
            parsed->block_flags = reinterpret_cast<const uint32_t *>(f->hFlags);
            parsed->bytes_copied = needBytes ? 1 : 0;
        }
        if (needBytes && outBytes)
        {
            // (page-locked room for the inflated bytes only when somebody will read them: with every block parsed on the device
            // nobody does)
            growPinned(f->hOut, f->capHOut, outBytes);
            PD_INF_HIP(hipMemcpyAsync(f->hOut, f->dOut, outBytes, hipMemcpyDeviceToHost, st));
        }
        PD_INF_HIP(hipMemcpyAsync(hLen, f->dLen, n_members * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        PD_INF_HIP(hipMemcpyAsync(hStatus, f->dStatus, n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PD_INF_HIP(hipEventRecord(f->finished, st));
        PD_INF_HIP(hipEventSynchronize(f->finished));
        if (kernel_ms)
        {
            float ms = 0;
            PD_INF_HIP(hipEventElapsedTime(&ms, e0, e1));
            *kernel_ms = ms;
        }
        *out = static_cast<const uint8_t *>(f->hOut);
        *status = hStatus;
        *out_lengths = hLen;
    }
    catch (const Fail & e)
    {
        f->error = "pd_inflater_run: " + e.what;
        g_inflateError = f->error;
        return 2;
    }
    return 0;
}

extern "C" int pd_inflater_run(pd_inflater * f, const uint64_t * member_offsets, uint32_t n_members, const uint64_t * out_offsets,
                               const uint8_t ** out, const uint32_t ** status, const uint64_t ** out_lengths, double * kernel_ms)
{
    return runInflater(f, member_offsets, n_members, out_offsets, nullptr, nullptr, out, status, out_lengths, nullptr, kernel_ms);
}

extern "C" int pd_inflater_run_parsed(pd_inflater * f, const uint64_t * member_offsets, uint32_t n_members, const uint64_t * out_offsets,
                                      const uint32_t * sample_first, const uint8_t * parse_member, const uint8_t ** out,
                                      const uint32_t ** status, const uint64_t ** out_lengths, pd_parsed_blocks * parsed, double * kernel_ms)
{
    if (!parsed)
        return 1;
    return runInflater(f, member_offsets, n_members, out_offsets, sample_first, parse_member, out, status, out_lengths, parsed, kernel_ms);
}

extern "C" int pd_inflater_run_summarized(pd_inflater * f, const uint64_t * member_offsets, uint32_t n_members, const uint64_t * out_offsets,
                                          const uint32_t * sample_first, const uint8_t * parse_member, const pd_summarize_options * opt,
                                          const uint8_t ** out, const uint32_t ** status, const uint64_t ** out_lengths, pd_parsed_blocks * parsed,
                                          pd_block_summaries * summaries, double * kernel_ms)
{
    if (!parsed || !summaries || !opt || ((opt->keep_below == nullptr) != (opt->keep_from == nullptr)))
        return 1;
    return runInflater(f, member_offsets, n_members, out_offsets, sample_first, parse_member, out, status, out_lengths, parsed, kernel_ms,
                       opt->inner_offset, opt->starts_room, opt->ends_room, summaries, opt->lean, opt->keep_below, opt->keep_from);
}

extern "C" uint64_t pd_inflater_run_id(pd_inflater * f) { return f ? f->runId : 0; }
extern "C" const char * pd_inflate_last_error(void) { return g_inflateError.c_str(); }
extern "C" const char * pd_inflater_last_error(pd_inflater * f) { return f ? f->error.c_str() : "pd_inflater_last_error: no inflater"; }

extern "C" int pd_inflate_members(int device, const uint8_t * data, const uint64_t * member_offsets, uint32_t n_members, uint8_t * out,
                                  const uint64_t * out_offsets, uint32_t * status, uint64_t * out_lengths, uint32_t repeat,
                                  pd_inflate_timing * timing)
{
    if (!data || !member_offsets || !out || !out_offsets || !status || !out_lengths || n_members == 0)
    {
        g_inflateError = "pd_inflate_members: invalid argument";
        return 1;
    }
    void *dData = nullptr, *dOut = nullptr, *dInOff = nullptr, *dOutOff = nullptr, *dStatus = nullptr, *dLen = nullptr;
    void *hIn = nullptr, *hOut = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = nullptr;
    int rc = 0;
    try
    {
        PD_INF_HIP(hipSetDevice(device));
        const uint64_t inBytes = member_offsets[n_members], outBytes = out_offsets[n_members];
        int nCu = 0;
        PD_INF_HIP(hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, device));
        unsigned groups = 0;
        PD_INF_HIP(hipStreamCreate(&st));
        for (hipEvent_t & e : ev)
            PD_INF_HIP(hipEventCreate(&e));
        PD_INF_HIP(hipMalloc(&dData, inBytes + pdinf::kInputSlack));
        PD_INF_HIP(hipMalloc(&dOut, outBytes + pdinf::kOutputSlack));
        PD_INF_HIP(hipMalloc(&dInOff, (n_members + 1) * sizeof(uint64_t)));
        PD_INF_HIP(hipMalloc(&dOutOff, (n_members + 1) * sizeof(uint64_t)));
        PD_INF_HIP(hipMalloc(&dStatus, n_members * sizeof(uint32_t)));
        PD_INF_HIP(hipMalloc(&dLen, n_members * sizeof(uint64_t)));
        PD_INF_HIP(hipHostMalloc(&hIn, inBytes + 8, hipHostMallocDefault));
        PD_INF_HIP(hipHostMalloc(&hOut, std::max<uint64_t>(outBytes, 1), hipHostMallocDefault));
        std::memcpy(hIn, data, inBytes);
        PD_INF_HIP(hipMemcpyAsync(dInOff, member_offsets, (n_members + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        PD_INF_HIP(hipMemcpyAsync(dOutOff, out_offsets, (n_members + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        PD_INF_HIP(hipStreamSynchronize(st));
        PD_INF_HIP(hipEventRecord(ev[0], st));
        PD_INF_HIP(hipMemcpyAsync(dData, hIn, inBytes, hipMemcpyHostToDevice, st));
        PD_INF_HIP(hipEventRecord(ev[1], st));
        const uint32_t times = std::max<uint32_t>(1, repeat);
        for (uint32_t k = 0; k < times; ++k)
            groups = launchInflate((unsigned)std::max(1, nCu), st, static_cast<const uint8_t *>(dData),
                                   static_cast<const uint64_t *>(dInOff), n_members, static_cast<uint8_t *>(dOut),
                                   static_cast<const uint64_t *>(dOutOff), static_cast<uint32_t *>(dStatus), static_cast<uint64_t *>(dLen));
        PD_INF_HIP(hipGetLastError());
        PD_INF_HIP(hipEventRecord(ev[2], st));
        PD_INF_HIP(hipMemcpyAsync(hOut, dOut, outBytes, hipMemcpyDeviceToHost, st));
        PD_INF_HIP(hipEventRecord(ev[3], st));
        PD_INF_HIP(hipMemcpyAsync(status, dStatus, n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        PD_INF_HIP(hipMemcpyAsync(out_lengths, dLen, n_members * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        PD_INF_HIP(hipStreamSynchronize(st));
        std::memcpy(out, hOut, outBytes);
        if (timing)
        {
            float a = 0, b = 0, c = 0;
            PD_INF_HIP(hipEventElapsedTime(&a, ev[0], ev[1]));
            PD_INF_HIP(hipEventElapsedTime(&b, ev[1], ev[2]));
            PD_INF_HIP(hipEventElapsedTime(&c, ev[2], ev[3]));
            timing->h2d_ms = a;
            timing->kernel_ms = b / times;
            timing->d2h_ms = c;
            timing->in_bytes = inBytes;
            timing->out_bytes = outBytes;
            timing->workgroups = groups;
        }
    }
    catch (const Fail & f)
    {
        g_inflateError = "pd_inflate_members: " + f.what;
        rc = 2;
    }
    for (void * p : {dData, dOut, dInOff, dOutOff, dStatus, dLen})
        if (p)
            (void)hipFree(p);
    for (void * p : {hIn, hOut})
        if (p)
            (void)hipHostFree(p);
    for (hipEvent_t e : ev)
        if (e)
            (void)hipEventDestroy(e);
    if (st)
        (void)hipStreamDestroy(st);
    return rc;
}
