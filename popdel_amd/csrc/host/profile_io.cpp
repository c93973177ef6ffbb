#include "profile_io.h"

#include <functional>

#include <dlfcn.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace pdh {
namespace {

constexpr size_t kIndexStart = 7 + 2 * sizeof(uint32_t);

class RawSource : public ByteSource
{
public:
    explicit RawSource(FILE * fp) : fp_(fp) {}
    size_t read(void * dst, size_t n) override { return fread(dst, 1, n, fp_); }
private:
    FILE * fp_;
};

// Inflates a run of concatenated gzip members starting at the current file offset. The reader asks for a few bytes at a
// time (window header, record count, ~25 records of 5 bytes); handing those requests to inflate() one by one keeps zlib on
// its byte-wise path (its fast loop wants 258 bytes of room), so the stream is inflated in blocks of 16 KB into a buffer
// of its own and the requests are served from there (a bigger block would inflate too far beyond the window at which the
// loader stops reading a round: a 200-kb slot is ~100 KB of records).
class GzSource : public ByteSource
{
public:
    explicit GzSource(FILE * fp) : fp_(fp), in_(1 << 16), out_(1 << 14)
    {
        std::memset(&zs_, 0, sizeof(zs_));
        if (inflateInit2(&zs_, 15 + 32) != Z_OK)
            throw ProfileError("[PopDel] zlib initialisation failed.");
    }
    // the same over bytes in memory (what MemberSource hands over when it meets a member it does not take)
    GzSource(const unsigned char * data, size_t size) : fp_(nullptr), out_(1 << 14), mem_(data), memLeft_(size)
    {
        std::memset(&zs_, 0, sizeof(zs_));
        if (inflateInit2(&zs_, 15 + 32) != Z_OK)
            throw ProfileError("[PopDel] zlib initialisation failed.");
    }
    ~GzSource() override { inflateEnd(&zs_); }

    size_t read(void * dst, size_t n) override
    {
        unsigned char * to = static_cast<unsigned char *>(dst);
        size_t done = 0;
        while (done < n)
        {
            if (outPos_ == outEnd_ && !refill())
                break;
            const size_t take = std::min(n - done, outEnd_ - outPos_);
            std::memcpy(to + done, out_.data() + outPos_, take);
            outPos_ += take;
            done += take;
        }
        return done;
    }

private:
    // the next block of inflated bytes; false at the end of the data
    bool refill()
    {
        outPos_ = outEnd_ = 0;
        if (finished_)
            return false;
        zs_.next_out = out_.data();
        zs_.avail_out = static_cast<uInt>(out_.size());
        while (zs_.avail_out != 0 && !finished_)
        {
            if (zs_.avail_in == 0 && fp_ == nullptr)
            {
                const size_t got = std::min<size_t>(memLeft_, static_cast<size_t>(1) << 30);
                if (got == 0)
                {
                    finished_ = true;
                    break;
                }
                zs_.next_in = const_cast<unsigned char *>(mem_);
                zs_.avail_in = static_cast<uInt>(got);
                mem_ += got;
                memLeft_ -= got;
            }
            else if (zs_.avail_in == 0)
            {
                size_t got = fread(in_.data(), 1, in_.size(), fp_);
                if (got == 0)
                {
                    finished_ = true;
                    break;
                }
                zs_.next_in = in_.data();
                zs_.avail_in = static_cast<uInt>(got);
            }
            int rc = inflate(&zs_, Z_NO_FLUSH);
            if (rc == Z_STREAM_END)
            {
                // Next member (if any) follows immediately.
                if (inflateReset(&zs_) != Z_OK)
                    throw ProfileError("[PopDel] zlib reset failed.");
                continue;
            }
            if (rc == Z_BUF_ERROR && zs_.avail_in == 0)
                continue;
            if (rc != Z_OK)
            {
                // Bytes that are not a gzip member end the stream, like an EOF.
                finished_ = true;
                break;
            }
        }
        outEnd_ = out_.size() - zs_.avail_out;
        return outEnd_ != 0;
    }

    FILE * fp_;
    std::vector<unsigned char> in_, out_;
    const unsigned char * mem_ = nullptr;
    size_t memLeft_ = 0;
    size_t outPos_ = 0, outEnd_ = 0;
    z_stream zs_;
    bool finished_ = false;
};

// libdeflate, if the system has it (Ubuntu's libtiff brings libdeflate.so.0 along; nothing here links it): its gzip decoder
// takes one whole member at a time -- which is how profiles are written, one member of a few kilobytes per 10-kb index
// region -- and runs 1.6 times as fast as zlib's inflate() on them (289 against 176 MB/s; word-wise bit buffer, CRC-32 by
// carry-less multiplication). Inflate is a third of `popdel call`'s host time.
struct Libdeflate
{
    void * (*alloc)() = nullptr;
    void (*release)(void *) = nullptr;
    // libdeflate_gzip_decompress_ex: 0 success, 1 bad data, 2 short output, 3 insufficient space
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    bool usable = false;
};

const Libdeflate & libdeflate()
{
    static const Libdeflate lib = [] {
        Libdeflate l;
        if (getenv("POPDEL_ZLIB_INFLATE"))                      // (A/B switch: the same bytes either way)
            return l;
        void * h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h)
            return l;
        l.alloc = reinterpret_cast<void * (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        l.release = reinterpret_cast<void (*)(void *)>(dlsym(h, "libdeflate_free_decompressor"));
        l.gunzip = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *)>(
            dlsym(h, "libdeflate_gzip_decompress_ex"));
        l.usable = l.alloc && l.release && l.gunzip;
        return l;
    }();
    return lib;
}

// The stream of gzip members through libdeflate, member by member, reading the file where the page cache holds it (the rest
// of the file from the current offset is mapped; the loader stops asking long before its end). A member libdeflate does
// not take -- damaged, truncated, larger than kMaxMember, or not a gzip member at all -- goes through zlib together with
// everything behind it (GzSource over the mapped bytes), which delivers what can be decoded and then ends the stream: the
// behaviour on bad data is zlib's, byte for byte.
class MemberSource : public ByteSource
{
public:
    static constexpr size_t kMaxMember = static_cast<size_t>(64) << 20;

    // nullptr if libdeflate is not there, the file cannot be mapped or does not continue with a gzip member here
    static MemberSource * open(FILE * fp)
    {
        if (!libdeflate().usable)
            return nullptr;
        const off_t at = ftello(fp);
        struct stat st;
        if (at < 0 || fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < at + 18)
            return nullptr;
        const long page = sysconf(_SC_PAGESIZE);
        const off_t from = at - at % page;
        const size_t length = static_cast<size_t>(st.st_size - from);
        void * map = mmap(nullptr, length, PROT_READ, MAP_PRIVATE, fileno(fp), from);
        if (map == MAP_FAILED)
            return nullptr;
        const unsigned char * data = static_cast<const unsigned char *>(map) + (at - from);
        if (data[0] != 0x1f || data[1] != 0x8b)
        {
            munmap(map, length);
            return nullptr;
        }
        void * d = libdeflate().alloc();
        if (!d)
        {
            munmap(map, length);
            return nullptr;
        }
        return new MemberSource(map, length, data, static_cast<size_t>(st.st_size - at), d);
    }
    ~MemberSource() override
    {
        zlib_.reset();
        libdeflate().release(decoder_);
        munmap(map_, length_);
    }

    size_t read(void * dst, size_t n) override
    {
        if (zlib_)
            return zlib_->read(dst, n);
        unsigned char * to = static_cast<unsigned char *>(dst);
        size_t done = 0;
        while (done < n)
        {
            if (pos_ == end_ && !nextMember())
            {
                if (zlib_)
                    return done + zlib_->read(to + done, n - done);
                break;
            }
            const size_t take = std::min(n - done, end_ - pos_);
            std::memcpy(to + done, out_.data() + pos_, take);
            pos_ += take;
            done += take;
        }
        return done;
    }

private:
    MemberSource(void * map, size_t length, const unsigned char * data, size_t size, void * decoder)
        : map_(map), length_(length), in_(data), inLeft_(size), decoder_(decoder), out_(1 << 16)
    {
    }
    // the next member with content; false at the end of the data or when zlib has taken over (zlib_ set)
    bool nextMember()
    {
        pos_ = end_ = 0;
        while (inLeft_ != 0)
        {
            size_t usedIn = 0, gotOut = 0;
            const int rc = libdeflate().gunzip(decoder_, in_, inLeft_, out_.data(), out_.size(), &usedIn, &gotOut);
            if (rc == 0)
            {
                in_ += usedIn;
                inLeft_ -= usedIn;
                if (gotOut == 0)
                    continue;
                end_ = gotOut;
                return true;
            }
            if (rc == 3 && out_.size() < kMaxMember)
            {
                out_.resize(out_.size() * 4);
                continue;
            }
            zlib_.reset(new GzSource(in_, inLeft_));
            inLeft_ = 0;
            return false;
        }
        return false;
    }

    void * map_;
    size_t length_;
    const unsigned char * in_;
    size_t inLeft_;
    void * decoder_;
    std::vector<unsigned char> out_;
    size_t pos_ = 0, end_ = 0;
    std::unique_ptr<GzSource> zlib_;
};

template <typename T>
bool readPod(FILE * fp, T & v)
{
    return fread(&v, sizeof(T), 1, fp) == 1;
}

[[noreturn]] void corrupted(const std::string & path, const char * what)
{
    throw ProfileError("[PopDel] Corrupted profile \"" + path + "\": " + what);
}

std::string readName(FILE * fp, const std::string & path, const char * lenMsg, const char * nameMsg, const char * termMsg)
{
    uint32_t nameLen = 0;
    if (!readPod(fp, nameLen) || nameLen == 0)
        corrupted(path, lenMsg);
    std::string s(nameLen - 1, '\0');
    char term = 1;
    if ((nameLen > 1 && fread(&s[0], 1, nameLen - 1, fp) != nameLen - 1) || fread(&term, 1, 1, fp) != 1)
        corrupted(path, nameMsg);
    if (term != '\0')
        corrupted(path, termMsg);
    return s;
}

}  // namespace

ProfileFile::ProfileFile(const std::string & path) : path_(path)
{
    fp_ = fopen(path.c_str(), "rb");
    if (!fp_)
        throw ProfileError("[PopDel] Could not open file '" + path + "' for reading.");
}

ProfileFile::~ProfileFile()
{
    if (fp_)
        fclose(fp_);
}

void ProfileFile::readHeader(ProfileHeader & hdr, bool dropContigs)
{
    fseeko(fp_, 0, SEEK_SET);
    char magic[7];
    if (fread(magic, 1, 7, fp_) != 7)
        corrupted(path_, "Unable to read magic string.");
    if (std::memcmp(magic, "POPDEL\1", 7) != 0)
        corrupted(path_, "Magic string is wrong.");
    uint32_t regionSize = 0, numRegions = 0;
    if (!readPod(fp_, regionSize))
        corrupted(path_, "Unable to read index region size.");
    if (!readPod(fp_, numRegions))
        corrupted(path_, "Unable to read index size.");
    hdr.indexRegionSize = regionSize;
    hdr.numRegions = numRegions;
    fseeko(fp_, static_cast<off_t>(numRegions) * sizeof(uint64_t), SEEK_CUR);

    uint32_t numRg = 0;
    if (!readPod(fp_, numRg))
        corrupted(path_, "Unable to read number of read groups.");
    if (numRg == 0)
        corrupted(path_, "Profile seems to have 0 read groups.");
    hdr.readGroups.assign(numRg, std::string());
    hdr.histograms.assign(numRg, InsertHistogram());
    for (uint32_t i = 0; i < numRg; ++i)
    {
        hdr.readGroups[i] = readName(fp_, path_, "Unable to read length of read group name.",
                                     "Unable to read read group name.",
                                     "Expecting read group name to be null-terminated.");
        InsertHistogram & h = hdr.histograms[i];
        uint32_t median = 0, readLen = 0, histStart = 0, histEnd = 0;
        double sd = 0;
        if (!readPod(fp_, median))
            corrupted(path_, "Unable to read median insert size.");
        if (!readPod(fp_, sd))
            corrupted(path_, "Unable to read standard deviation of insert size.");
        if (!readPod(fp_, readLen))
            corrupted(path_, "Unable to read read length.");
        if (!readPod(fp_, histStart))
            corrupted(path_, "Unable to read offset of insert size histogram.");
        if (!readPod(fp_, histEnd))
            corrupted(path_, "Unable to read insert size histogram size.");
        h.median = median;
        h.stddev = sd;
        h.readLength = readLen;
        h.offset = static_cast<int>(histStart);
        h.values.assign(histEnd - histStart, 0.0);
        if (!h.values.empty() && fread(h.values.data(), sizeof(double), h.values.size(), fp_) != h.values.size())
            corrupted(path_, "Unable to read value from histogram.");
    }
    uint32_t numContigs = 0;
    if (!readPod(fp_, numContigs))
        corrupted(path_, "Unable to read number of contigs.");
    if (!dropContigs)
    {
        hdr.contigNames.assign(numContigs, std::string());
        hdr.contigLengths.assign(numContigs, 0);
    }
    for (uint32_t i = 0; i < numContigs; ++i)
    {
        if (!dropContigs)
        {
            hdr.contigNames[i] = readName(fp_, path_, "Unable to read length of contig name.",
                                          "Unable to read contig name.", "Expecting contig name to be null-terminated.");
            if (!readPod(fp_, hdr.contigLengths[i]))
                corrupted(path_, "Unable to read contig length.");
        }
        else
        {
            uint32_t nameLen = 0;
            if (!readPod(fp_, nameLen))
                corrupted(path_, "Unable to read length of contig name.");
            if (fseeko(fp_, static_cast<off_t>(nameLen) + sizeof(int32_t), SEEK_CUR) != 0)
                corrupted(path_, "Unable to skip contig name and length.");
        }
    }
}

void ProfileFile::jumpToRegion(const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths,
                               unsigned indexRegionSize, const std::string & contig, int32_t beginPos)
{
    uint64_t slot = 0;
    for (size_t i = 0; i < contigNames.size(); ++i)
    {
        if (contig == contigNames[i])
            break;
        slot += static_cast<uint64_t>(contigLengths[i] / static_cast<int32_t>(indexRegionSize)) + 1;
    }
    slot += static_cast<uint64_t>(beginPos / static_cast<int32_t>(indexRegionSize));
    uint64_t offset = 0;
    fseeko(fp_, static_cast<off_t>(kIndexStart + slot * sizeof(uint64_t)), SEEK_SET);
    if (!readPod(fp_, offset))
        offset = 0;
    fseeko(fp_, static_cast<off_t>(offset), SEEK_SET);
}

bool ProfileFile::blockOffsets(const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths,
                               unsigned indexRegionSize, const std::string & contig, int32_t beginPos, int32_t endPos,
                               std::vector<uint64_t> & offsets, std::vector<uint64_t> * perRegion)
{
    return blockOffsetsOf(fileno(fp_), contigNames, contigLengths, indexRegionSize, contig, beginPos, endPos, offsets, perRegion);
}

bool ProfileFile::blockOffsetsOf(int fd, const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths,
                                 unsigned indexRegionSize, const std::string & contig, int32_t beginPos, int32_t endPos,
                                 std::vector<uint64_t> & offsets, std::vector<uint64_t> * perRegion, uint64_t * fileSize)
{
    offsets.clear();
    if (perRegion)
        perRegion->clear();
    uint32_t numRegions = 0;
    if (pread(fd, &numRegions, sizeof(numRegions), 7 + sizeof(uint32_t)) != static_cast<ssize_t>(sizeof(numRegions)) || numRegions == 0)
        return false;
    uint64_t base = 0;
    size_t c = 0;
    for (; c < contigNames.size(); ++c)
    {
        if (contig == contigNames[c])
            break;
        base += static_cast<uint64_t>(contigLengths[c] / static_cast<int32_t>(indexRegionSize)) + 1;
    }
    if (c == contigNames.size() || beginPos < 0 || endPos < beginPos)
        return false;
    const uint64_t lastOfContig = base + static_cast<uint64_t>(contigLengths[c] / static_cast<int32_t>(indexRegionSize));
    const uint64_t s0 = base + static_cast<uint64_t>(beginPos / static_cast<int32_t>(indexRegionSize));
    uint64_t s1 = std::min(lastOfContig, base + static_cast<uint64_t>(endPos / static_cast<int32_t>(indexRegionSize))) + 1;     // one past
    if (s0 >= numRegions)
        return false;
    s1 = std::min<uint64_t>(s1, numRegions);
    std::vector<uint64_t> raw(static_cast<size_t>(s1 - s0) + 1, 0);
    const size_t have = static_cast<size_t>(std::min<uint64_t>(s1 + 1, numRegions) - s0);
    if (pread(fd, raw.data(), have * sizeof(uint64_t), static_cast<off_t>(kIndexStart + s0 * sizeof(uint64_t))) !=
        static_cast<ssize_t>(have * sizeof(uint64_t)))
        return false;
    struct stat st;
    if (fstat(fd, &st) != 0)
        return false;
    if (fileSize)
        *fileSize = static_cast<uint64_t>(st.st_size);
    if (have < raw.size())
        raw.back() = static_cast<uint64_t>(st.st_size);          // (behind the last region of the file)
    for (uint64_t o : raw)
    {
        if (o > static_cast<uint64_t>(st.st_size) || (!offsets.empty() && o < offsets.back()))
            return false;                                       // (not an index this reader understands)
        if (offsets.empty() || o != offsets.back())
            offsets.push_back(o);
    }
    if (perRegion)
        *perRegion = raw;
    return offsets.size() >= 1;
}

namespace {
// `size` bytes that are there already, then whatever `next()` opens (called once, when they are used up)
class PrefetchedSource : public ByteSource
{
public:
    PrefetchedSource(const unsigned char * data, size_t size, std::function<std::unique_ptr<ByteSource>()> next)
        : p_(data), left_(size), next_(std::move(next))
    {
    }
    size_t read(void * dst, size_t n) override
    {
        unsigned char * to = static_cast<unsigned char *>(dst);
        size_t done = std::min(n, left_);
        if (done)
        {
            std::memcpy(to, p_, done);
            p_ += done;
            left_ -= done;
        }
        if (done < n)
        {
            if (!rest_ && next_)
            {
                rest_ = next_();
                next_ = nullptr;
            }
            if (rest_)
                done += rest_->read(to + done, n - done);
        }
        return done;
    }

    const unsigned char * peek(size_t & avail) override
    {
        avail = left_;
        return left_ ? p_ : nullptr;
    }
    void consume(size_t n) override
    {
        p_ += n;
        left_ -= n;
    }

private:
    const unsigned char * p_;
    size_t left_;
    std::function<std::unique_ptr<ByteSource>()> next_;
    std::unique_ptr<ByteSource> rest_;
};
}  // namespace

std::unique_ptr<ByteSource> ProfileFile::openPrefetched(const unsigned char * data, size_t size, uint64_t resumeOffset, bool forceZlib)
{
    return std::unique_ptr<ByteSource>(new PrefetchedSource(data, size, [this, resumeOffset, forceZlib]() -> std::unique_ptr<ByteSource> {
        struct stat st;
        if (fstat(fileno(fp_), &st) != 0 || resumeOffset >= static_cast<uint64_t>(st.st_size))
            return nullptr;                                     // (the end of the data)
        fseeko(fp_, static_cast<off_t>(resumeOffset), SEEK_SET);
        return openSource(false, forceZlib);
    }));
}

std::unique_ptr<ByteSource> ProfileFile::openSource(bool uncompressed, bool forceZlib)
{
    if (uncompressed)
        return std::unique_ptr<ByteSource>(new RawSource(fp_));
    if (!forceZlib)
        if (MemberSource * m = MemberSource::open(fp_))
            return std::unique_ptr<ByteSource>(m);
    return std::unique_ptr<ByteSource>(new GzSource(fp_));
}

bool readWindow(ByteSource & src, ProfileWindow & win, unsigned numReadGroups)
{
    win.chrom = 0;
    win.beginPos = 0;
    for (auto & r : win.records)
        r.clear();
    uint32_t chrom = 0, begin = 0;
    if (src.read(&chrom, 4) != 4)
        return false;
    if (src.read(&begin, 4) != 4)
        throw ProfileError("[PopDel] Unable to read contig position.");
    win.chrom = static_cast<int32_t>(chrom);
    win.beginPos = static_cast<int32_t>(begin);
    win.records.resize(numReadGroups);
    static thread_local std::vector<unsigned char> buf;        // (one heap block per thread instead of one per window)
    for (unsigned rg = 0; rg < numReadGroups; ++rg)
    {
        uint32_t n = 0;
        if (src.read(&n, 4) != 4)
            throw ProfileError("[PopDel] Unable to read number of read pairs.");
        std::vector<PairRecord> & recs = win.records[rg];
        recs.resize(n);
        buf.resize(static_cast<size_t>(n) * 5);
        if (n != 0 && src.read(buf.data(), buf.size()) != buf.size())
            throw ProfileError("[PopDel] Unable to read insert size deviation.");
        for (uint32_t i = 0; i < n; ++i)
        {
            const unsigned char * p = buf.data() + static_cast<size_t>(i) * 5;
            uint32_t dev;
            std::memcpy(&dev, p + 1, 4);
            recs[i].pos = static_cast<uint32_t>(p[0]) + begin;
            recs[i].dev = dev;
        }
    }
    return true;
}

bool readWindowHead(ByteSource & src, WindowHead & head)
{
    uint32_t chrom = 0, begin = 0;
    size_t avail = 0;
    if (const unsigned char * p = src.peek(avail))
        if (avail >= 8)
        {
            std::memcpy(&chrom, p, 4);
            std::memcpy(&begin, p + 4, 4);
            src.consume(8);
            head.chrom = static_cast<int32_t>(chrom);
            head.beginPos = static_cast<int32_t>(begin);
            return true;
        }
    if (src.read(&chrom, 4) != 4)
        return false;
    if (src.read(&begin, 4) != 4)
        throw ProfileError("[PopDel] Unable to read contig position.");
    head.chrom = static_cast<int32_t>(chrom);
    head.beginPos = static_cast<int32_t>(begin);
    return true;
}

namespace {
// the 5-byte records of one read group of the window, raw; the thread's scratch buffer holds them until the next call
const unsigned char * readRawRecords(ByteSource & src, uint32_t & n)
{
    static thread_local std::vector<unsigned char> buf;
    n = 0;
    {
        // (bytes that sit in memory already -- blocks the device has inflated -- are parsed where they are)
        size_t avail = 0;
        if (const unsigned char * p = src.peek(avail))
            if (avail >= 4)
            {
                std::memcpy(&n, p, 4);
                const size_t body = static_cast<size_t>(n) * 5;
                if (avail - 4 >= body)
                {
                    src.consume(4 + body);
                    return p + 4;
                }
                n = 0;
            }
    }
    if (src.read(&n, 4) != 4)
        throw ProfileError("[PopDel] Unable to read number of read pairs.");
    if (buf.size() < static_cast<size_t>(n) * 5)
        buf.resize(static_cast<size_t>(n) * 5);
    if (n != 0 && src.read(buf.data(), static_cast<size_t>(n) * 5) != static_cast<size_t>(n) * 5)
        throw ProfileError("[PopDel] Unable to read insert size deviation.");
    return buf.data();
}
}  // namespace

void skipWindowBody(ByteSource & src, unsigned numReadGroups)
{
    uint32_t n;
    for (unsigned rg = 0; rg < numReadGroups; ++rg)
        (void)readRawRecords(src, n);
}

bool readWindowBodyFlat(ByteSource & src, const WindowHead & head, unsigned numReadGroups, FlatWindows & wins, FlatCounts & counts,
                        FlatRecords & recs, unsigned oldSize, unsigned newSize)
{
    const uint32_t begin = static_cast<uint32_t>(head.beginPos);
    if (numReadGroups == 1)
    {
        // One read group (the rule): the records are position-sorted, so the 30-bp windows are runs of the record array as
        // it is -- converted in place behind `recs`, cut into windows on the way.
        uint32_t n;
        const unsigned char * p = readRawRecords(src, n);
        if (n == 0)
            return false;
        const size_t first = recs.size(), wins0 = wins.size(), counts0 = counts.size();
        recs.resize(first + n);
        PairRecord * o = recs.data() + first;
        const size_t slots = static_cast<size_t>(std::ceil(static_cast<double>(oldSize + newSize) / newSize));
        const unsigned base = ((static_cast<uint32_t>(p[0]) + begin) / newSize) * newSize;
        bool ordered = true;
        uint32_t prev = 0;
        uint32_t boundary = 0;          // first position beyond the current 30-bp window (0: none yet)
        uint32_t * count = nullptr;
        // one pass: the record array as it is, cut into 30-bp windows on the way (a record out of order undoes the cuts)
        for (uint32_t i = 0; i < n; ++i, p += 5)
        {
            uint32_t dev;
            std::memcpy(&dev, p + 1, 4);
            const uint32_t pos = static_cast<uint32_t>(p[0]) + begin;
            o[i].pos = pos;
            o[i].dev = dev;
            ordered = ordered && pos >= prev;
            prev = pos;
            if (pos >= boundary)
            {
                const uint32_t winBegin = base + ((pos - base) / newSize) * newSize;
                wins.push_back(FlatWindow{head.chrom, static_cast<int32_t>(winBegin), static_cast<uint32_t>(first + i)});
                counts.push_back(0);
                count = &counts.back();
                boundary = winBegin + newSize;
            }
            ++*count;
        }
        if (ordered && (o[n - 1].pos - base) / newSize < slots)
            return true;
        wins.resize(wins0);
        counts.resize(counts0);
        // (out of order: the general path reports or handles it)
        ProfileWindow win;
        win.chrom = head.chrom;
        win.beginPos = head.beginPos;
        win.records.resize(1);
        win.records[0].assign(o, o + n);
        recs.resize(first);
        return splitWindowFlat(win, wins, counts, recs, oldSize, newSize);
    }
    static thread_local ProfileWindow win;
    win.chrom = head.chrom;
    win.beginPos = head.beginPos;
    win.records.resize(numReadGroups);
    for (unsigned rg = 0; rg < numReadGroups; ++rg)
    {
        uint32_t n;
        const unsigned char * p = readRawRecords(src, n);
        std::vector<PairRecord> & r = win.records[rg];
        r.resize(n);
        for (uint32_t i = 0; i < n; ++i, p += 5)
        {
            std::memcpy(&r[i].dev, p + 1, 4);
            r[i].pos = static_cast<uint32_t>(p[0]) + begin;
        }
    }
    return splitWindowFlat(win, wins, counts, recs, oldSize, newSize);
}

bool splitWindow(const ProfileWindow & in, std::vector<ProfileWindow> & out, unsigned oldSize, unsigned newSize)
{
    const size_t rgNum = in.records.size();
    const size_t slots = static_cast<size_t>(std::ceil(static_cast<double>(oldSize + newSize) / newSize));
    unsigned base = std::numeric_limits<unsigned>::max();
    for (size_t rg = 0; rg < rgNum; ++rg)
        if (!in.records[rg].empty() && in.records[rg][0].pos < base)
            base = in.records[rg][0].pos;        // records are position-sorted: element 0 is the minimum
    base = (base / newSize) * newSize;

    std::vector<std::vector<unsigned>> counts(rgNum, std::vector<unsigned>(slots, 0));
    std::vector<char> hasContent(slots, 0);
    for (size_t rg = 0; rg < rgNum; ++rg)
        for (const PairRecord & r : in.records[rg])
        {
            size_t idx = (r.pos - base) / newSize;
            if (idx >= slots)
                throw ProfileError("[PopDel] Profile window is not position-sorted.");
            hasContent[idx] = 1;
            ++counts[rg][idx];
        }
    std::vector<int> slotToOut(slots, -1);
    unsigned total = 0;
    for (size_t i = 0; i < slots; ++i)
        if (hasContent[i])
            slotToOut[i] = static_cast<int>(total++);
    out.clear();
    if (total == 0)
        return false;
    out.resize(total);
    for (size_t i = 0; i < slots; ++i)
    {
        if (slotToOut[i] < 0)
            continue;
        ProfileWindow & w = out[static_cast<size_t>(slotToOut[i])];
        w.chrom = in.chrom;
        w.beginPos = static_cast<int32_t>(base + i * newSize);
        w.records.resize(rgNum);
        for (size_t rg = 0; rg < rgNum; ++rg)
            w.records[rg].reserve(counts[rg][i]);
    }
    for (size_t rg = 0; rg < rgNum; ++rg)
        for (const PairRecord & r : in.records[rg])
            out[static_cast<size_t>(slotToOut[(r.pos - base) / newSize])].records[rg].push_back(r);
    return true;
}

bool splitWindowFlat(const ProfileWindow & in, FlatWindows & wins, FlatCounts & counts, FlatRecords & recs, unsigned oldSize,
                     unsigned newSize)
{
    const size_t rgNum = in.records.size();
    const size_t slots = static_cast<size_t>(std::ceil(static_cast<double>(oldSize + newSize) / newSize));
    unsigned base = std::numeric_limits<unsigned>::max();
    for (size_t rg = 0; rg < rgNum; ++rg)
        if (!in.records[rg].empty() && in.records[rg][0].pos < base)
            base = in.records[rg][0].pos;
    base = (base / newSize) * newSize;
    if (rgNum == 1)
    {
        // One read group (the rule): the records are position-sorted, so the 30-bp windows are runs of the record array as
        // it is -- one pass, no scatter. (A record out of order ends this path; the general one below reports or handles it.)
        const std::vector<PairRecord> & rs = in.records[0];
        if (rs.empty())
            return false;
        const size_t wins0 = wins.size(), counts0 = counts.size();
        const uint32_t first = static_cast<uint32_t>(recs.size());
        size_t prev = static_cast<size_t>(-1);
        bool ordered = true;
        for (size_t i = 0; i < rs.size(); ++i)
        {
            const size_t idx = (rs[i].pos - base) / newSize;
            if (idx != prev)
            {
                if (idx >= slots || (prev != static_cast<size_t>(-1) && idx < prev))
                {
                    ordered = false;
                    break;
                }
                wins.push_back(FlatWindow{in.chrom, static_cast<int32_t>(base + idx * newSize), first + static_cast<uint32_t>(i)});
                counts.push_back(0);
                prev = idx;
            }
            ++counts.back();
        }
        if (ordered)
        {
            recs.insert(recs.end(), rs.begin(), rs.end());
            return true;
        }
        wins.resize(wins0);
        counts.resize(counts0);
    }
    // per slot and read group: number of records (scratch on the stack for the usual sizes: ten slots, a few read groups)
    constexpr size_t kStack = 160;
    uint32_t perSlotStack[kStack], cursorStack[kStack];
    std::vector<uint32_t> perSlotHeap, cursorHeap;
    uint32_t *perSlot = perSlotStack, *cursor = cursorStack;
    if (slots * rgNum > kStack)
    {
        perSlotHeap.assign(slots * rgNum, 0);
        cursorHeap.assign(slots * rgNum, 0);
        perSlot = perSlotHeap.data();
        cursor = cursorHeap.data();
    }
    else
    {
        std::fill(perSlot, perSlot + slots * rgNum, 0u);
        std::fill(cursor, cursor + slots * rgNum, 0u);
    }
    char hasContentStack[64] = {0};
    std::vector<char> hasContentHeap;
    char * hasContent = hasContentStack;
    if (slots > 64)
    {
        hasContentHeap.assign(slots, 0);
        hasContent = hasContentHeap.data();
    }
    for (size_t rg = 0; rg < rgNum; ++rg)
        for (const PairRecord & r : in.records[rg])
        {
            size_t idx = (r.pos - base) / newSize;
            if (idx >= slots)
                throw ProfileError("[PopDel] Profile window is not position-sorted.");
            hasContent[idx] = 1;
            ++perSlot[idx * rgNum + rg];
        }
    size_t total = 0;
    for (size_t i = 0; i < slots; ++i)
        total += hasContent[i] ? 1 : 0;
    if (total == 0)
        return false;
    // layout: windows in slot order; within a window read groups in order
    uint32_t next = static_cast<uint32_t>(recs.size());
    for (size_t i = 0; i < slots; ++i)
    {
        if (!hasContent[i])
            continue;
        wins.push_back(FlatWindow{in.chrom, static_cast<int32_t>(base + i * newSize), next});
        for (size_t rg = 0; rg < rgNum; ++rg)
        {
            counts.push_back(perSlot[i * rgNum + rg]);
            cursor[i * rgNum + rg] = next;
            next += perSlot[i * rgNum + rg];
        }
    }
    recs.resize(next);
    for (size_t rg = 0; rg < rgNum; ++rg)
        for (const PairRecord & r : in.records[rg])
        {
            size_t idx = (r.pos - base) / newSize;
            recs[cursor[idx * rgNum + rg]++] = r;
        }
    return true;
}

}  // namespace pdh
