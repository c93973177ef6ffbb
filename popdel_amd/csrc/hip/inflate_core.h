// DEFLATE (RFC 1951) decoder for ONE gzip member (RFC 1952), written so that one LANE can run it: no heap, no recursion,
// canonical Huffman decoding from per-length offsets and a symbol list (582 bytes of table per decoder instead of the
// kilobytes a lookup-table decoder wants -- the decoders of a wave share its LDS, and how many of them a CU holds is how many
// members are in flight; a table of the next nine bits per code was built and measured slower on the device: the rare longer
// codes stall the lanes beside them).
//
// What it replaces: the gunzip of the per-index-region gzip members of a profile (reference: the writer ends a member per
// 10,000-bp region, workflow_popdel.h:201-241; the reader jumps to one, insert_histogram_popdel.h:531-562, and inflates it
// through seqan/stream/iostream_zip_impl.h:336-349 -- zlib). Same accept / reject behaviour as zlib's inflate on the block
// level: over-subscribed and incomplete code sets are refused (a single one-bit code is allowed), a missing end-of-block
// code is refused, distances beyond the output produced so far are refused; CRC-32 and ISIZE of the trailer are checked.
//
// The same source is compiled for the host (tests/: against zlib, bit for bit, including damaged and truncated members)
// and for the device (inflate.hip).
#ifndef PD_INFLATE_CORE_H_
#define PD_INFLATE_CORE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define PD_INF_FN __host__ __device__ __forceinline__
#else
#define PD_INF_FN inline
#endif

namespace pdinf {

enum Status : uint32_t
{
    kOk = 0,
    kBadHeader = 1,        // not a gzip member with deflate data
    kTruncated = 2,        // input ends inside the member
    kBadBlock = 3,         // reserved block type, stored-block length check, bad dynamic header
    kBadCode = 4,          // over-subscribed / incomplete code set, code not in the set, symbol out of range
    kBadDistance = 5,      // back-reference before the start of the output
    kOutputFull = 6,       // more output than the caller has room for
    kBadCrc = 7,
    kBadLength = 8         // ISIZE does not match
};

constexpr unsigned kMaxBits = 15, kLitLenCodes = 288, kDistCodes = 30, kCodeLengths = 320;
// A decoder's tables. The symbols of the two codes in canonical order, one byte each (kSymbols: the literal / length code's 288,
// then the distance code's -- or the code-length code's -- 30) with the ninth bit of the literal / length symbols in a bit
// map beside them (kHighWords x 32 bits: position p's bit is bit p % 32 of word p / 32; the distance symbols' word stays zero);
// per code and code length one 16-bit word (kMeta: 0..15 the literal / length code, 16..31 the other): the number of symbols
// of that length while they are counted, the place of the length's next symbol while they are listed -- so that afterwards it
// is where the length's symbols END and the difference to the length before is their number --, and for the symbol loop
// the index of the length's first symbol minus its first code (buildFast); the code lengths of a block header, four bits each.
constexpr unsigned kSymbols = 318, kDistSymbolsAt = 288, kHighWords = 10, kMeta = 32, kLengthBytes = kCodeLengths / 2;

// What a symbol stands for, in sixteen bits: bits 0-3 the number of extra bits behind it, bits 4-11 a multiplier m -- the length
// is 3 + (m << extra) + the extra bits, the distance 1 + (m << extra) + the extra bits --, bit 12: a literal, bit 13: the end of
// the block, bit 14: no symbol of the format (286, 287; distances 30, 31). Entries 0..287: the literal / length symbols,
// 288..319: the distance symbols. Lengths: 3..10 for x = symbol - 257 < 8 (m = x), then four symbols per extra-bit count
// (m = 4 + x % 4), 258 for x = 28 (m = 255); distances: 1..4 for x < 4 (m = x), then two symbols per extra-bit count (m = 2 + x % 2).
constexpr unsigned kSymbolKinds = 320;
constexpr unsigned kInputSlack = 16, kOutputSlack = 16;      // bytes the device's loops may READ beyond the last member's input / output
constexpr uint32_t kKindLiteral = 1u << 12, kKindEnd = 1u << 13, kKindNone = 1u << 14;
PD_INF_FN uint16_t symbolKind(unsigned i)
{
    if (i < 256u)
        return (uint16_t)kKindLiteral;
    if (i == 256u)
        return (uint16_t)kKindEnd;
    const bool dist = i >= 288u;
    const unsigned x = dist ? i - 288u : i - 257u;
    if (x >= (dist ? 30u : 29u))
        return (uint16_t)kKindNone;
    if (!dist && x == 28u)
        return (uint16_t)(255u << 4);
    const unsigned q = dist ? 1u : 2u, group = x >> q;
    if (group < 2u)
        return (uint16_t)(x << 4);
    return (uint16_t)((((1u << q) + (x & ((1u << q) - 1u))) << 4) | (group - 1u));
}

// Plain-array tables (host; one decoder).
struct FlatTables
{
    uint8_t  symbol[kSymbols];
    uint32_t highBits[kHighWords];
    uint16_t metaWord[kMeta];
    uint8_t  lengths[kLengthBytes];
    PD_INF_FN uint8_t & sym(unsigned i) { return symbol[i]; }
    PD_INF_FN uint32_t & high(unsigned i) { return highBits[i]; }
    PD_INF_FN uint16_t & meta(unsigned i) { return metaWord[i]; }
    PD_INF_FN uint8_t & lenPair(unsigned i) { return lengths[i]; }
    PD_INF_FN uint32_t kind(unsigned i) const { return symbolKind(i); }
};

// Tables of `Stride` decoders interleaved element by element (device: element i of lane l at base[i * Stride + l]; lanes that look at
// the same entry hit consecutive LDS banks).
template <unsigned Stride>
struct StridedTablesOf
{
    uint8_t *  symbols;    // + lane, all of them
    uint32_t * highs;
    uint16_t * metas;
    uint8_t *  lengths;
    const uint16_t * kinds; // symbolKind() of 0 .. kSymbolKinds - 1, one table for the decoders of a workgroup
    PD_INF_FN uint8_t & sym(unsigned i) { return symbols[i * Stride]; }
    PD_INF_FN uint32_t & high(unsigned i) { return highs[i * Stride]; }
    PD_INF_FN uint16_t & meta(unsigned i) { return metas[i * Stride]; }
    PD_INF_FN uint8_t & lenPair(unsigned i) { return lengths[i * Stride]; }
    PD_INF_FN uint32_t kind(unsigned i) const { return kinds[i]; }
};

// (the code length of symbol i of a block header)
template <class Tab>
PD_INF_FN unsigned lengthOf(Tab & tab, unsigned i)
{
    return (tab.lenPair(i >> 1) >> (4u * (i & 1u))) & 15u;
}
template <class Tab>
PD_INF_FN void setLength(Tab & tab, unsigned i, unsigned v)
{
    uint8_t & pair = tab.lenPair(i >> 1);
    pair = (uint8_t)((i & 1u) ? (pair & 0x0fu) | (v << 4) : (pair & 0xf0u) | v);
}
// (symbol at position p of the lists: the literal / length code's from 0, the other code's from kDistSymbolsAt)
template <class Tab>
PD_INF_FN unsigned symbolAt(Tab & tab, unsigned p)
{
    return tab.sym(p) | (((tab.high(p >> 5) >> (p & 31u)) & 1u) << 8);
}

struct BitReader
{
    const uint8_t * in;
    uint32_t pos, end;     // pos: first byte that is neither in `buf` nor in `ahead` (a member is far below 4 GB: inflateGzipMember)
    uint64_t buf;
    unsigned cnt;
    uint32_t ahead;        // the next four input bytes, loaded when the previous word went into `buf`: the load's latency
    bool     haveAhead;    // passes while the decoder works through the bits it has (a lane has nothing else to hide it)
    bool     shortInput;

    PD_INF_FN void start(const uint8_t * data, uint32_t from, uint32_t to)
    {
        in = data;
        pos = from;
        end = to;
        buf = 0;
        cnt = 0;
        ahead = 0;
        haveAhead = false;
        shortInput = false;
        fetchAhead();
    }
    PD_INF_FN void fetchAhead()
    {
        haveAhead = pos + 4u <= end;
        if (haveAhead)
        {
            __builtin_memcpy(&ahead, in + pos, 4);
            pos += 4u;
        }
    }
    PD_INF_FN void refill()
    {
        if (cnt <= 32u)
        {
            if (haveAhead)
            {
                buf |= (uint64_t)ahead << cnt;
                cnt += 32u;
                fetchAhead();
            }
            else
            {
                while (cnt <= 56u && pos < end)
                {
                    buf |= (uint64_t)in[pos++] << cnt;
                    cnt += 8u;
                }
            }
        }
    }
    // n <= 16
    PD_INF_FN unsigned bits(unsigned n)
    {
        refill();
        if (cnt < n)
        {
            shortInput = true;
            return 0u;
        }
        const unsigned v = (unsigned)(buf & ((1ull << n) - 1ull));
        buf >>= n;
        cnt -= n;
        return v;
    }
    PD_INF_FN void alignToByte()
    {
        const unsigned drop = cnt & 7u;
        buf >>= drop;
        cnt -= drop;
    }
    // offset of the first byte that has not been consumed (call after alignToByte)
    PD_INF_FN uint32_t bytePos() const { return pos - (haveAhead ? 4u : 0u) - (cnt >> 3); }
};

// One symbol of a canonical code, bit by bit (the code-length code of a block header; `set`: 0 the literal / length slots, 1 the
// other ones; the lists as buildCode leaves them): -1 if the bits are no code of the set, -2 if the input ends.
template <class Tab>
PD_INF_FN int decodeSymbol(BitReader & br, Tab & tab, unsigned set)
{
    br.refill();
    int code = 0, first = 0, index = 0;
    for (unsigned len = 1; len <= kMaxBits; ++len)
    {
        if (br.cnt == 0u)
        {
            br.shortInput = true;
            return -2;
        }
        code |= (int)(br.buf & 1ull);
        br.buf >>= 1;
        --br.cnt;
        const int count = (int)tab.meta(16u * set + len) - index;
        if (code - count < first)
            return (int)symbolAt(tab, (set ? kDistSymbolsAt : 0u) + (unsigned)(index + (code - first)));
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Lists the symbols of a code in canonical order from the code lengths of symbols from .. from + n (see the tables above).
// Returns 0 for a complete set, > 0 for an incomplete one (codes left over), < 0 for an over-subscribed one; *zeros = symbols
// without a code.
template <class Tab>
PD_INF_FN int buildCode(Tab & tab, unsigned from, unsigned n, unsigned set, unsigned * zeros)
{
    const unsigned metaAt = 16u * set, listAt = set ? kDistSymbolsAt : 0u;
    for (unsigned len = 0; len <= kMaxBits; ++len)
        tab.meta(metaAt + len) = 0;
    for (unsigned s = 0; s < n; ++s)
        ++tab.meta(metaAt + lengthOf(tab, from + s));
    *zeros = tab.meta(metaAt);
    tab.meta(metaAt) = 0;
    int left = 1;
    if (*zeros == n)
        return 0;                       // no codes at all: complete, and decoding will fail
    unsigned run = 0;
    bool over = false;
    for (unsigned len = 1; len <= kMaxBits; ++len)
    {
        const unsigned count = tab.meta(metaAt + len);
        left <<= 1;
        left -= (int)count;
        over = over || left < 0;
        tab.meta(metaAt + len) = (uint16_t)run;
        run += count;
    }
    if (over)
    {
        for (unsigned len = 1; len <= kMaxBits; ++len)
            tab.meta(metaAt + len) = 0;
        return -1;
    }
    if (!set)
        for (unsigned i = 0; i < kHighWords - 1u; ++i)
            tab.high(i) = 0;
    for (unsigned s = 0; s < n; ++s)
    {
        const unsigned l = lengthOf(tab, from + s);
        if (l != 0u)
        {
            const unsigned at = listAt + tab.meta(metaAt + l)++;
            tab.sym(at) = (uint8_t)s;
            if (s >> 8)
                tab.high(at >> 5) |= 1u << (at & 31u);
        }
    }
    return left;
}

// A canonical code as the symbol loop wants it: limit[len] = one past the last code of length len, left-aligned to 15 bits
// (non-decreasing in len, at most 0x8000), in REGISTERS, two to a word: pair[j] holds limit[2j + 1] + 0x7fff in its low half and
// limit[2j + 2] + 0x7fff in its high half (the sixteenth half: 0xffff). With the next fifteen bits v in both halves of a word,
// pair[j] - v has bit 15 of a half set where v < limit (no borrow crosses the halves: limit + 0x7fff - v >= 0), so the length of
// the code is 17 minus the number of such bits -- a subtraction, a mask and a bit count per two lengths instead of a compare, a
// wait and a conditional add per length. The meta word of a length becomes the index of the length's first symbol minus its
// first code (biased by 0x8000). One symbol then costs a bit reversal, these and two dependent table reads.
struct FastCode
{
    uint32_t pair[(kMaxBits + 1) / 2];
};

template <class Tab>
PD_INF_FN void buildFast(Tab & tab, unsigned set, FastCode & fc)
{
    uint32_t first = 0, index = 0, low = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (unsigned len = 1; len <= kMaxBits; ++len)
    {
        const uint32_t end = tab.meta(16u * set + len), count = end - index;
        tab.meta(16u * set + len) = (uint16_t)(0x8000u + index - first);
        first += count;
        const uint32_t biased = (first << (kMaxBits - len)) + 0x7fffu;
        if (len & 1u)
            low = biased;
        else
            fc.pair[len / 2u - 1u] = low | (biased << 16);
        index = end;
        first <<= 1;
    }
    fc.pair[kMaxBits / 2u] = low | 0xffff0000u;
}

PD_INF_FN uint32_t reverse15(uint32_t v)
{
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(v) >> 17;
#else
    uint32_t r = 0;
    for (int i = 0; i < 15; ++i)
        r |= ((v >> i) & 1u) << (14 - i);
    return r;
#endif
}

struct Output
{
    uint8_t * out;
    uint32_t  n, cap;
    PD_INF_FN void put(unsigned b) { out[n++] = (uint8_t)b; }
};

// The input as the symbol loop reads it: the 32-bit word the bit cursor stands in and the one behind it, and the bits left in
// the member. One symbol with its extra bits is at most 28 bits: ONE look at the next 32 bits serves a whole trip of the loop.
// The word after these two is loaded at the beginning of EVERY trip and taken at its end if the cursor has crossed into the
// second word by then (mostly it has not: the load is repeated, out of the cache) -- its latency passes behind the decoding,
// and no register that waits for memory lives across the loop's back edge (the compiler would wait for it there, and for the
// stores in front of it). What lies beyond the end of the input may be looked at (zeros on the host, whatever follows on the
// device) and is never consumed: `left` tells when a symbol took bits that were not there.
struct WordReader
{
    const uint8_t * in;
    uint32_t end, at;      // bytes; at: where w0 was read from
    uint32_t w0, w1, bo;
    int32_t  left;         // bits of input not yet consumed
    PD_INF_FN uint32_t word(uint32_t p) const
    {
        uint32_t v = 0;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        __builtin_memcpy(&v, in + p, 4);        // (the device's input ends with sixteen bytes of slack, kInputSlack: what lies beyond a
                                                //  member is never consumed -- `left` --, only looked at)
#else
        if (p + 4u <= end)
            __builtin_memcpy(&v, in + p, 4);
        else
            for (unsigned i = 0; i < 4u && p + i < end; ++i)
                v |= (uint32_t)in[p + i] << (8u * i);
#endif
        return v;
    }
    PD_INF_FN void start(const uint8_t * data, uint32_t to, uint32_t bitPos)
    {
        in = data;
        end = to;
        at = (bitPos >> 5) << 2;
        bo = bitPos & 31u;
        w0 = word(at);
        w1 = word(at + 4u);
        left = (int32_t)(8u * to - bitPos);
    }
    PD_INF_FN uint32_t upcoming() const { return word(at + 8u); }
    PD_INF_FN uint32_t peek() const
    {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_alignbit(w1, w0, bo);
#else
        return (uint32_t)((((uint64_t)w1 << 32) | w0) >> bo);
#endif
    }
    PD_INF_FN void consume(unsigned n, uint32_t next)      // n < 32; next: upcoming() of this position
    {
        bo += n;
        left -= (int32_t)n;
        if (bo >= 32u)
        {
            bo -= 32u;
            w0 = w1;
            w1 = next;
            at += 4u;
        }
    }
    PD_INF_FN uint32_t bitPos() const { return 8u * at + bo; }
};

PD_INF_FN uint32_t bitField(uint32_t v, unsigned from, unsigned n)      // n < 32, from < 32
{
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, from, n);
#else
    return (v >> from) & ((1u << n) - 1u);
#endif
}

// The symbols of one block. A trip of the loop decodes ONE symbol -- a literal / length symbol or, behind a length, the distance
// symbol: which code the lane reads is a matter of its state, the instructions are the same, so that the lanes of a wave (each
// with a member of its own, each somewhere else in it) walk through one instruction stream instead of taking turns on the
// literal, the length and the distance path -- and emits what the trip BEFORE it produced, up to four bytes a time: a literal is
// a copy of one byte out of a register, a match one of bytes loaded eight at a time (as many of them as the distance says are
// there) -- the load goes out at the beginning of the trip that emits the match, behind the bytes of the item before it, and has
// the symbol's decoding to arrive. A wave's time here is its instruction count (four waves a SIMD keep it issuing): what a symbol
// stands for comes out of a table (symbolKind), every reason to stop is one branch. (The CRC is not taken here: a pass of
// its own over the output does that with every lane of a wave busy, crcMembersKernel / inflateGzipMember.)
template <class Tab>
PD_INF_FN uint32_t inflateCodes(BitReader & br, Output & o, Tab & tab)
{
    FastCode lenCode, distCode;
    buildFast(tab, 0, lenCode);
    buildFast(tab, 1, distCode);
    WordReader wr;
    wr.start(br.in, br.end, 8u * (br.pos - (br.haveAhead ? 4u : 0u)) - br.cnt);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt(0x0f70);     // (vmcnt(0) here: the compiler would wait for the first two words at the loop's head, trip after trip)
#endif
    uint32_t rc = kOk, matchLen = 0;
    bool distNext = false;
    uint32_t itemBytes = 0, itemDist = 0, itemLiteral = 0;     // the item waiting to be emitted: itemDist = 0 for a literal
    for (;;)
    {
        // the loads of the trip: the input word that may be needed at its end, and the eight bytes at the source of the match
        // that is waiting to be emitted (a literal: the eight bytes at the cursor, not looked at). On the device through inline
        // assembly, the wait for them placed by hand behind the decoding: no register that waits for memory lives across the
        // loop's back edge or a branch's join, where the compiler would wait at once.
        uint32_t upcoming;
        uint64_t itemChunk;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("global_load_dword %0, %2, off offset:8\n\tglobal_load_dwordx2 %1, %3, off"
                     : "=&v"(upcoming), "=&v"(itemChunk) : "v"(wr.in + wr.at), "v"(o.out + (o.n - itemDist)) : "memory");
#else
        upcoming = wr.upcoming();
        itemChunk = 0;
        for (unsigned i = 0; i < 8u && i < itemDist; ++i)
            itemChunk |= (uint64_t)o.out[o.n - itemDist + i] << (8u * i);
#endif
        const uint32_t window = wr.peek();
        const uint32_t v = reverse15(window & 0x7fffu);                    // (what lies beyond the input is looked at, never consumed)
        const uint32_t both = v * 0x10001u;
        unsigned below = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (unsigned j = 0; j < (kMaxBits + 1) / 2; ++j)
            below += (unsigned)__builtin_popcount(((distNext ? distCode.pair[j] : lenCode.pair[j]) - both) & 0x80008000u);
        const bool noCode = below < 2u;                                     // (17 - below > 15)
        const unsigned len = noCode ? kMaxBits : 17u - below;
        const uint32_t code = v >> (kMaxBits - len);
        const uint32_t base = tab.meta((distNext ? 16u : 0u) + len);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        const uint32_t index = (code + base - 0x8000u) & 0xffffu;           // (without a code: anything -- LDS reads out of range deliver zeros)
#else
        const uint32_t index = noCode ? 0u : (code + base - 0x8000u) & 0xffffu;
#endif
        const uint32_t sym = symbolAt(tab, (distNext ? kDistSymbolsAt : 0u) + index);
        const uint32_t kind = tab.kind((distNext ? 288u : 0u) + sym);
        const unsigned extra = kind & 15u;
        const uint32_t value = (((kind >> 4) & 0xffu) << extra) + (distNext ? 1u : 3u) + bitField(window, len, extra);
        const bool literal = (kind & kKindLiteral) != 0u, endOfBlock = (kind & kKindEnd) != 0u;
        const unsigned used = len + extra;
        const bool truncated = wr.left < (int32_t)used;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(upcoming), "+v"(itemChunk) : : "memory");
#endif
        wr.consume(used, upcoming);
        // the item of the trip before
        if (itemBytes != 0u)
        {
            uint64_t chunk = itemDist != 0u ? itemChunk : (uint64_t)itemLiteral;
            unsigned avail = itemDist != 0u && itemDist < 8u ? itemDist : 8u;
            const bool roomy = o.n + itemBytes + 4u <= o.cap;
            for (;;)
            {
                unsigned k = itemBytes < 4u ? itemBytes : 4u;
                k = k < avail ? k : avail;
                const uint32_t wd = (uint32_t)chunk;
                if (roomy)                                          // (the bytes beyond k are written again by what follows)
                {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("global_store_dword %0, %1, off" : : "v"(o.out + o.n), "v"(wd) : "memory");
#else
                    __builtin_memcpy(o.out + o.n, &wd, 4);
#endif
                }
                else
                {
                    for (unsigned i = 0; i < k; ++i)
                        o.out[o.n + i] = (uint8_t)(wd >> (8u * i));
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_s_waitcnt(0x0f70);             // (vmcnt(0): nothing of the compiler's own in flight at the loop's head)
#endif
                }
                chunk >>= 8u * k;
                avail -= k;
                o.n += k;
                itemBytes -= k;
                if (itemBytes == 0u)
                    break;
                if (avail == 0u)            // (a match longer than its distance, or than eight bytes: what has been written since is there now)
                {
                    __builtin_memcpy(&chunk, o.out + (o.n - itemDist), 8);
                    avail = itemDist < 8u ? itemDist : 8u;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_s_waitcnt(0x0f70);
#endif
                }
            }
        }
        // ... and the one of this trip
        const uint32_t bytes = literal ? 1u : distNext ? matchLen : 0u;
        const bool noSymbol = noCode | ((kind & kKindNone) != 0u);
        const bool tooFar = distNext & (value > o.n);
        const bool full = o.n + bytes > o.cap;
        if (noSymbol | truncated | endOfBlock | tooFar | full)
        {
            // (the end of the input first: what the device sees beyond it is not what the host sees, and must not decide)
            rc = truncated ? kTruncated : noSymbol ? kBadCode : endOfBlock ? kOk : tooFar ? kBadDistance : kOutputFull;
            break;
        }
        itemBytes = bytes;
        itemDist = distNext ? value : 0u;
        itemLiteral = sym;
        matchLen = value;
        distNext = !literal & !distNext;
    }
    // the reader of the block headers goes on where the symbols ended
    const uint32_t bitPos = wr.bitPos();
    br.start(br.in, bitPos >> 3, br.end);
    if ((bitPos & 7u) != 0u)
        br.bits(bitPos & 7u);
    return rc;
}

// The raw deflate stream behind a gzip header; on success the reader stands behind the final block.
template <class Tab>
PD_INF_FN uint32_t inflateStream(BitReader & br, Output & o, Tab & tab)
{
    // order in which the code-length code's lengths are transmitted
    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    unsigned last;
    do
    {
        last = br.bits(1);
        const unsigned type = br.bits(2);
        if (br.shortInput)
            return kTruncated;
        if (type == 0u)
        {
            br.alignToByte();
            const unsigned len = br.bits(16), nlen = br.bits(16);
            if (br.shortInput)
                return kTruncated;
            if (len != (~nlen & 0xffffu))
                return kBadBlock;
            if (o.n + len > o.cap)
                return kOutputFull;
            for (unsigned i = 0; i < len; ++i)
            {
                const unsigned b = br.bits(8);
                if (br.shortInput)
                    return kTruncated;
                o.put(b);
            }
        }
        else if (type == 1u)
        {
            unsigned zeros;
            for (unsigned s = 0; s < 288u; ++s)
                setLength(tab, s, s < 144u ? 8u : s < 256u ? 9u : s < 280u ? 7u : 8u);
            buildCode(tab, 0, 288, 0, &zeros);
            for (unsigned s = 0; s < 30u; ++s)
                setLength(tab, 288u + s, 5u);
            buildCode(tab, 288, 30, 1, &zeros);
            const uint32_t rc = inflateCodes(br, o, tab);
            if (rc != kOk)
                return rc;
        }
        else if (type == 2u)
        {
            const unsigned nlen = br.bits(5) + 257u, ndist = br.bits(5) + 1u, ncode = br.bits(4) + 4u;
            if (br.shortInput)
                return kTruncated;
            if (nlen > 286u || ndist > 30u)
                return kBadBlock;
            // the code-length code, built in the distance slots (free until the real tables are built)
            unsigned zeros;
            for (unsigned i = 0; i < 19u; ++i)
                setLength(tab, i, 0u);
            for (unsigned i = 0; i < ncode; ++i)
                setLength(tab, order[i], br.bits(3));
            if (br.shortInput)
                return kTruncated;
            if (buildCode(tab, 0, 19, 1, &zeros) != 0)
                return kBadCode;        // the code-length code must be complete
            unsigned idx = 0;
            while (idx < nlen + ndist)
            {
                const int sym = decodeSymbol(br, tab, 1);
                if (sym < 0)
                    return sym == -2 ? kTruncated : kBadCode;
                if (sym < 16)
                {
                    setLength(tab, idx++, (unsigned)sym);
                    continue;
                }
                unsigned value = 0, rep;
                if (sym == 16)
                {
                    if (idx == 0u)
                        return kBadBlock;
                    value = lengthOf(tab, idx - 1u);
                    rep = 3u + br.bits(2);
                }
                else if (sym == 17)
                    rep = 3u + br.bits(3);
                else
                    rep = 11u + br.bits(7);
                if (br.shortInput)
                    return kTruncated;
                if (idx + rep > nlen + ndist)
                    return kBadBlock;
                while (rep--)
                    setLength(tab, idx++, value);
            }
            if (lengthOf(tab, 256) == 0u)
                return kBadCode;        // no end-of-block code
            // (the code lengths sit in one run: literal/length codes first, distance codes behind them)
            // (the distance code goes first: its lengths lie behind the others and the list of the literal / length code does not touch them)
            int err = buildCode(tab, nlen, ndist, 1, &zeros);
            if (err != 0 && (err < 0 || ndist != zeros + tab.meta(16u + 1u)))
                return kBadCode;        // incomplete sets are only allowed as a single one-bit code
            err = buildCode(tab, 0, nlen, 0, &zeros);
            if (err != 0 && (err < 0 || nlen != zeros + tab.meta(1u)))
                return kBadCode;
            const uint32_t rc = inflateCodes(br, o, tab);
            if (rc != kOk)
                return rc;
        }
        else
            return kBadBlock;
    } while (!last);
    return kOk;
}

// One gzip member at in[0 .. inLen): header, deflate stream, trailer. Output to out[0 .. outCap); *outLen = bytes written
// (also on failure: what had been produced), *used = bytes of input the member occupies (on success).
template <class Tab>
PD_INF_FN uint32_t inflateGzipMember(const uint8_t * in, uint64_t inLen, uint8_t * out, uint64_t outCap, uint64_t * outLen,
                                     uint64_t * used, Tab & tab, const uint32_t * crcTable, uint32_t * crcOut = nullptr)
{
    *outLen = 0;
    *used = 0;
    if (inLen > 0xffff0000ull)
        return kBadHeader;          // (positions are kept in 32 bits: nothing the profiles hold comes near)
    if (inLen < 18u)
        return inLen >= 3u && !(in[0] == 0x1f && in[1] == 0x8b && in[2] == 8) ? kBadHeader : kTruncated;
    if (in[0] != 0x1f || in[1] != 0x8b || in[2] != 8 || (in[3] & 0xe0u) != 0u)
        return kBadHeader;
    const unsigned flg = in[3];
    uint64_t p = 10;
    if (flg & 4u)
    {
        if (p + 2u > inLen)
            return kTruncated;
        p += 2u + ((uint64_t)in[p] | ((uint64_t)in[p + 1] << 8));
    }
    for (unsigned field = 0; field < 2u; ++field)          // FNAME, FCOMMENT: zero-terminated
        if (flg & (8u << field))
        {
            while (p < inLen && in[p] != 0)
                ++p;
            ++p;
        }
    if (flg & 2u)
        p += 2u;
    if (p >= inLen)
        return kTruncated;
    BitReader br;
    br.start(in, (uint32_t)p, (uint32_t)inLen);
    Output o;
    o.out = out;
    o.n = 0;
    o.cap = outCap > 0xffff0000ull ? 0xffff0000u : (uint32_t)outCap;
    tab.high(kHighWords - 1u) = 0;          // (the distance symbols' ninth bits: never set)
    const uint32_t rc = inflateStream(br, o, tab);
    *outLen = o.n;
    if (rc != kOk)
        return rc;
    br.alignToByte();
    const uint64_t t = br.bytePos();
    if (t + 8u > inLen)
        return kTruncated;
    const uint32_t crc = (uint32_t)in[t] | ((uint32_t)in[t + 1] << 8) | ((uint32_t)in[t + 2] << 16) | ((uint32_t)in[t + 3] << 24);
    const uint32_t isize = (uint32_t)in[t + 4] | ((uint32_t)in[t + 5] << 8) | ((uint32_t)in[t + 6] << 16) | ((uint32_t)in[t + 7] << 24);
    if (isize != (uint32_t)o.n)
        return kBadLength;
    if (crcOut != nullptr)
        *crcOut = crc;                  // (the caller takes the CRC of the output: crcMembersKernel)
    else
    {
        uint32_t c = 0xffffffffu;
        for (uint32_t i = 0; i < o.n; ++i)
            c = crcTable[(c ^ out[i]) & 0xffu] ^ (c >> 8);
        if (crc != (c ^ 0xffffffffu))
            return kBadCrc;
    }
    *used = t + 8u;
    return kOk;
}

// The CRC-32 table of RFC 1952 (polynomial edb88320).
PD_INF_FN uint32_t crcTableEntry(unsigned n)
{
    uint32_t c = n;
    for (int k = 0; k < 8; ++k)
        c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
    return c;
}

}  // namespace pdinf
#endif
