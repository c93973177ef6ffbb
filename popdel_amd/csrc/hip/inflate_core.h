// DEFLATE (RFC 1951) decoder for ONE gzip member (RFC 1952), written so that one LANE can run it: no heap, no recursion,
// canonical Huffman decoding from per-length counts and a symbol list (a few hundred 16-bit words of table per decoder
// instead of the kilobytes a lookup-table decoder wants -- the decoders of a wave share its LDS; a table of the next nine
// bits per code was built and measured slower on the device: the rare longer codes stall the lanes beside them), the last eight output
// bytes in a register so that the short back-references the profiles are made of (the zero / ff bytes of the 32-bit
// deviations, five bytes back) cost no memory round trip.
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
// 16-bit table words per decoder: lenCount[16] lenSymbol[288] distCount[16] distSymbol[30] offs[16] lenBase[16] distBase[16]
constexpr unsigned kTabLenCount = 0, kTabLenSymbol = 16, kTabDistCount = 304, kTabDistSymbol = 320, kTabOffs = 350, kTabLenBase = 366,
                   kTabDistBase = 382, kTabWords = 398;

// Plain-array tables (host; one decoder).
struct FlatTables
{
    uint16_t word[kTabWords];
    uint8_t  length[kCodeLengths];
    PD_INF_FN uint16_t & w(unsigned i) { return word[i]; }
    PD_INF_FN uint8_t & len(unsigned i) { return length[i]; }
};

// Tables of `Stride` decoders interleaved word by word (device: word i of lane l at base[i * Stride + l]; lanes that look at the same
// entry hit consecutive LDS banks), code lengths likewise in a global workspace.
template <unsigned Stride>
struct StridedTablesOf
{
    uint16_t * base;       // + lane
    uint8_t *  lengths;    // + lane
    PD_INF_FN uint16_t & w(unsigned i) { return base[i * Stride]; }
    PD_INF_FN uint8_t & len(unsigned i) { return lengths[i * Stride]; }
};

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

// One symbol of a canonical code (counts at countAt + 1 .. 15, symbols at symbolAt): -1 if the bits are no code of the set,
// -2 if the input ends.
template <class Tab>
PD_INF_FN int decodeSymbol(BitReader & br, Tab & tab, unsigned countAt, unsigned symbolAt)
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
        const int count = tab.w(countAt + len);
        if (code - count < first)
            return tab.w(symbolAt + (unsigned)(index + (code - first)));
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// Builds counts and symbol list from code lengths tab.len(from .. from + n). Returns 0 for a complete set, > 0 for an
// incomplete one (codes left over), < 0 for an over-subscribed one.
template <class Tab>
PD_INF_FN int buildCode(Tab & tab, unsigned from, unsigned n, unsigned countAt, unsigned symbolAt)
{
    for (unsigned len = 0; len <= kMaxBits; ++len)
        tab.w(countAt + len) = 0;
    for (unsigned s = 0; s < n; ++s)
        ++tab.w(countAt + tab.len(from + s));
    if (tab.w(countAt) == n)
        return 0;                       // no codes at all: complete, and decoding will fail
    int left = 1;
    for (unsigned len = 1; len <= kMaxBits; ++len)
    {
        left <<= 1;
        left -= (int)tab.w(countAt + len);
        if (left < 0)
            return left;
    }
    tab.w(kTabOffs + 1) = 0;
    for (unsigned len = 1; len < kMaxBits; ++len)
        tab.w(kTabOffs + len + 1) = (uint16_t)(tab.w(kTabOffs + len) + tab.w(countAt + len));
    for (unsigned s = 0; s < n; ++s)
    {
        const unsigned l = tab.len(from + s);
        if (l != 0u)
            tab.w(symbolAt + tab.w(kTabOffs + l)++) = (uint16_t)s;
    }
    return left;
}

// A canonical code as the symbol loop wants it: limit[len] = one past the last code of length len, left-aligned to 15 bits
// (non-decreasing in len), in REGISTERS (every use below is unrolled); base[len] = index of the length's first symbol minus
// its first code, in the table words at baseAt (biased by 0x8000). One symbol then costs a bit reversal, fifteen compares and
// two dependent table reads instead of one table read per code bit.
struct FastCode
{
    uint32_t limit[kMaxBits + 1];
};

template <class Tab>
PD_INF_FN void buildFast(Tab & tab, unsigned countAt, unsigned baseAt, FastCode & fc)
{
    uint32_t first = 0, index = 0;
    fc.limit[0] = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (unsigned len = 1; len <= kMaxBits; ++len)
    {
        const uint32_t count = tab.w(countAt + len);
        tab.w(baseAt + len) = (uint16_t)(0x8000u + index - first);
        first += count;
        fc.limit[len] = first << (kMaxBits - len);
        index += count;
        first <<= 1;
    }
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

// -1: the bits are no code of the set; -2: the input ends inside the code
template <class Tab>
PD_INF_FN int decodeFast(BitReader & br, Tab & tab, const FastCode & fc, unsigned baseAt, unsigned symbolAt)
{
    br.refill();
    const uint32_t v = reverse15((uint32_t)br.buf & 0x7fffu);          // (bits beyond the input read as zeros)
    unsigned len = 1;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (unsigned k = 1; k <= kMaxBits; ++k)
        len += v >= fc.limit[k] ? 1u : 0u;
    if (len > kMaxBits)
        return -1;
    if (len > br.cnt)
    {
        br.shortInput = true;
        return -2;
    }
    br.buf >>= len;
    br.cnt -= len;
    const uint32_t code = v >> (kMaxBits - len);
    return tab.w(symbolAt + ((code + tab.w(baseAt + len) - 0x8000u) & 0xffffu));
}

struct Output
{
    uint8_t * out;
    uint32_t  n, cap;
    uint64_t  tail;        // the last eight bytes written, most recent in the low byte
    uint32_t  crc;
    const uint32_t * crcTable;
    PD_INF_FN void put(unsigned b)
    {
        out[n++] = (uint8_t)b;
        tail = (tail << 8) | b;
        crc = crcTable[(crc ^ b) & 0xffu] ^ (crc >> 8);
    }
};

template <class Tab>
PD_INF_FN uint32_t inflateCodes(BitReader & br, Output & o, Tab & tab)
{
    FastCode lenCode, distCode;
    buildFast(tab, kTabLenCount, kTabLenBase, lenCode);
    buildFast(tab, kTabDistCount, kTabDistBase, distCode);
    for (;;)
    {
        int sym = decodeFast(br, tab, lenCode, kTabLenBase, kTabLenSymbol);
        if (sym < 0)
            return sym == -2 ? kTruncated : kBadCode;
        if (sym < 256)
        {
            if (o.n >= o.cap)
                return kOutputFull;
            o.put((unsigned)sym);
            continue;
        }
        if (sym == 256)
            return kOk;
        sym -= 257;
        if (sym >= 29)
            return kBadCode;
        // length: 3..10 for symbols 0..7, then four symbols per extra-bit count, 258 for the last
        unsigned len;
        if (sym < 8)
            len = 3u + (unsigned)sym;
        else if (sym == 28)
            len = 258u;
        else
        {
            const unsigned ext = ((unsigned)sym >> 2) - 1u;
            len = 3u + ((4u + ((unsigned)sym & 3u)) << ext) + br.bits(ext);
        }
        const int dsym = decodeFast(br, tab, distCode, kTabDistBase, kTabDistSymbol);
        if (dsym < 0)
            return dsym == -2 ? kTruncated : kBadCode;
        if (dsym >= 30)
            return kBadCode;
        unsigned dist;
        if (dsym < 4)
            dist = 1u + (unsigned)dsym;
        else
        {
            const unsigned ext = ((unsigned)dsym >> 1) - 1u;
            dist = 1u + ((2u + ((unsigned)dsym & 1u)) << ext) + br.bits(ext);
        }
        if (br.shortInput)
            return kTruncated;
        if (dist > o.n)
            return kBadDistance;
        if (o.n + len > o.cap)
            return kOutputFull;
        if (dist <= 8u)
        {
            const unsigned sh = 8u * (dist - 1u);
            for (unsigned i = 0; i < len; ++i)
                o.put((unsigned)(o.tail >> sh) & 0xffu);
        }
        else
        {
            // more than eight bytes back: the eight bytes at the source are all there, one load serves up to eight of the
            // match's bytes (70 % of a profile block's bytes come this way, four to a match)
            unsigned left = len;
            while (left != 0u)
            {
                uint64_t v;
                __builtin_memcpy(&v, o.out + (o.n - dist), 8);
                const unsigned take = left < 8u ? left : 8u;
                for (unsigned i = 0; i < take; ++i, v >>= 8)
                    o.put((unsigned)v & 0xffu);
                left -= take;
            }
        }
    }
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
            for (unsigned s = 0; s < 144u; ++s) tab.len(s) = 8;
            for (unsigned s = 144u; s < 256u; ++s) tab.len(s) = 9;
            for (unsigned s = 256u; s < 280u; ++s) tab.len(s) = 7;
            for (unsigned s = 280u; s < 288u; ++s) tab.len(s) = 8;
            buildCode(tab, 0, 288, kTabLenCount, kTabLenSymbol);
            for (unsigned s = 0; s < 30u; ++s) tab.len(288u + s) = 5;
            buildCode(tab, 288, 30, kTabDistCount, kTabDistSymbol);
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
            for (unsigned i = 0; i < 19u; ++i)
                tab.len(i) = 0;
            for (unsigned i = 0; i < ncode; ++i)
                tab.len(order[i]) = (uint8_t)br.bits(3);
            if (br.shortInput)
                return kTruncated;
            if (buildCode(tab, 0, 19, kTabDistCount, kTabDistSymbol) != 0)
                return kBadCode;        // the code-length code must be complete
            unsigned idx = 0;
            while (idx < nlen + ndist)
            {
                const int sym = decodeSymbol(br, tab, kTabDistCount, kTabDistSymbol);
                if (sym < 0)
                    return sym == -2 ? kTruncated : kBadCode;
                if (sym < 16)
                {
                    tab.len(idx++) = (uint8_t)sym;
                    continue;
                }
                unsigned value = 0, rep;
                if (sym == 16)
                {
                    if (idx == 0u)
                        return kBadBlock;
                    value = tab.len(idx - 1u);
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
                    tab.len(idx++) = (uint8_t)value;
            }
            if (tab.len(256) == 0)
                return kBadCode;        // no end-of-block code
            // (the code lengths sit in one run: literal/length codes first, distance codes behind them)
            int err = buildCode(tab, 0, nlen, kTabLenCount, kTabLenSymbol);
            if (err != 0 && (err < 0 || nlen != (unsigned)tab.w(kTabLenCount) + tab.w(kTabLenCount + 1)))
                return kBadCode;        // incomplete sets are only allowed as a single one-bit code
            err = buildCode(tab, nlen, ndist, kTabDistCount, kTabDistSymbol);
            if (err != 0 && (err < 0 || ndist != (unsigned)tab.w(kTabDistCount) + tab.w(kTabDistCount + 1)))
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
                                     uint64_t * used, Tab & tab, const uint32_t * crcTable)
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
    o.tail = 0;
    o.crc = 0xffffffffu;
    o.crcTable = crcTable;
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
    if (crc != (o.crc ^ 0xffffffffu))
        return kBadCrc;
    if (isize != (uint32_t)o.n)
        return kBadLength;
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
