// Reader for PopDel's binary profile format ("POPDEL\1"): header, region index, window records, and the
// 256-bp -> 30-bp window split used by `popdel call`.
// Behavioural reference: insert_histogram_popdel.h:334-525 (readProfileHeader), :531-562 (jumpToRegion);
// popdel_profile/window_podel.h:210-251 (readWindow), :314-418 (convertWindow);
// seqan/stream/iostream_zip_impl.h:336-349 (concatenated gzip members are read through).
#ifndef PDH_PROFILE_IO_H_
#define PDH_PROFILE_IO_H_

#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "histogram.h"
#include "huge_pool.h"

namespace pdh {

struct ProfileError : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

struct ProfileHeader
{
    unsigned indexRegionSize = 0;
    unsigned numRegions = 0;
    std::vector<std::string> readGroups;
    std::vector<InsertHistogram> histograms;   // raw counts as stored in the file
    std::vector<std::string> contigNames;
    std::vector<int32_t> contigLengths;
};

// One (position, deviation) record; pos is the absolute end position of the forward read.
struct PairRecord
{
    uint32_t pos;
    uint32_t dev;     // int32 deviation kept in an unsigned, as the reference does
};

struct ProfileWindow
{
    int32_t chrom = -1;
    int32_t beginPos = -1;
    std::vector<std::vector<PairRecord>> records;   // per read group of the file
    bool empty() const { return records.empty(); }
};

// Sequential byte source positioned inside a profile: plain file bytes or an inflating view of them.
class ByteSource
{
public:
    virtual ~ByteSource() {}
    // Reads exactly n bytes unless the data ends; returns the number of bytes delivered.
    virtual size_t read(void * dst, size_t n) = 0;
    // A source that has its next bytes in memory may show them (`avail` of them at the returned address, valid as long as the
    // source's buffer lives) and be told how many were taken: the parser then works on them in place. Default: nothing to show.
    virtual const unsigned char * peek(size_t & avail)
    {
        avail = 0;
        return nullptr;
    }
    virtual void consume(size_t) {}
};

class ProfileFile
{
public:
    explicit ProfileFile(const std::string & path);
    ~ProfileFile();
    ProfileFile(const ProfileFile &) = delete;
    ProfileFile & operator=(const ProfileFile &) = delete;

    const std::string & path() const { return path_; }
    // Parse the header from the start of the file. With dropContigs the contig table is skipped.
    void readHeader(ProfileHeader & hdr, bool dropContigs = false);
    // Seek to the first data block of the index region that contains `beginPos` on `contig`.
    void jumpToRegion(const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths,
                      unsigned indexRegionSize, const std::string & contig, int32_t beginPos);
    // Byte source starting at the current file offset. Compressed profiles are inflated with libdeflate where the system
    // has it (member by member, the file mapped), else -- or with forceZlib, or POPDEL_ZLIB_INFLATE=1 -- with zlib; the
    // bytes delivered are the same, on damaged files too.
    std::unique_ptr<ByteSource> openSource(bool uncompressed, bool forceZlib = false);

    FILE * handle() { return fp_; }

    // File offsets that cut the data of [beginPos, endPos] on `contig` into the blocks the index knows: the offsets of the
    // index regions from the one that holds beginPos to the one behind the one that holds endPos (clipped to the contig's
    // regions; the end of the file behind the last region), duplicates removed -- a compressed profile holds one gzip member
    // per non-empty index region, so consecutive offsets bracket one member each. Leaves the file position alone.
    // Returns false if the index cannot be read. perRegion (optional): the same offsets region by region, duplicates kept --
    // perRegion[d] is where a jump to the d-th region behind beginPos's lands.
    bool blockOffsets(const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths, unsigned indexRegionSize,
                      const std::string & contig, int32_t beginPos, int32_t endPos, std::vector<uint64_t> & offsets,
                      std::vector<uint64_t> * perRegion = nullptr);
    // The same on a descriptor the caller keeps open (the loader's gathering passes ask every sample every round: opening and
    // closing the file each time was a sixteenth of a round's CPU time).
    // (*fileSize: the file's size, so that the caller sees whether the last offset is the end of the file)
    static bool blockOffsetsOf(int fd, const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths,
                               unsigned indexRegionSize, const std::string & contig, int32_t beginPos, int32_t endPos,
                               std::vector<uint64_t> & offsets, std::vector<uint64_t> * perRegion = nullptr, uint64_t * fileSize = nullptr);
    // Byte source over `size` bytes already inflated (the blocks from the jump position on) that goes on reading the file at
    // `resumeOffset` when they are used up -- what openSource() would deliver from the jump position, with the first blocks
    // inflated by someone else (the device).
    std::unique_ptr<ByteSource> openPrefetched(const unsigned char * data, size_t size, uint64_t resumeOffset, bool forceZlib = false);

private:
    std::string path_;
    FILE * fp_ = nullptr;
};

// Read the next 256-bp window record. Returns false at end of data (before the contig id).
bool readWindow(ByteSource & src, ProfileWindow & win, unsigned numReadGroups);

// Split a 256-bp window into 30-bp windows on the global 30-grid; only windows with content are produced.
// Returns false if the window holds no records at all.
bool splitWindow(const ProfileWindow & in, std::vector<ProfileWindow> & out, unsigned oldSize = 256, unsigned newSize = 30);

// Flat variant for bulk decoding: appends the 30-bp windows of `in` to `wins` (begin positions), their per-read-group
// record counts to `counts` (numReadGroups entries per window) and the records to `recs` (window by window, read group by
// read group, file order within). Same windows and order as splitWindow(). Returns false if `in` holds no records.
struct FlatWindow
{
    int32_t  chrom;
    int32_t  beginPos;
    uint32_t firstRec;     // index into the record array
};
// (the arrays of a decoded stream: hundreds of kilobytes per sample and round, see huge_pool.h)
using FlatWindows = HugeVector<FlatWindow>;
using FlatCounts = HugeVector<uint32_t>;
using FlatRecords = HugeVector<PairRecord>;
bool splitWindowFlat(const ProfileWindow & in, FlatWindows & wins, FlatCounts & counts, FlatRecords & recs, unsigned oldSize = 256,
                     unsigned newSize = 30);

// readWindow() + splitWindowFlat() in two steps for bulk decoding, so that a window the caller does not want is not
// converted and the records of a one-read-group file go from the inflated bytes to `recs` in one pass. Same data, same
// errors (a truncated body throws where readWindow() would).
struct WindowHead
{
    int32_t chrom = 0;
    int32_t beginPos = 0;
};
bool readWindowHead(ByteSource & src, WindowHead & head);                       // false at the end of the data
void skipWindowBody(ByteSource & src, unsigned numReadGroups);
// false if the window holds no records (nothing appended)
bool readWindowBodyFlat(ByteSource & src, const WindowHead & head, unsigned numReadGroups, FlatWindows & wins, FlatCounts & counts,
                        FlatRecords & recs, unsigned oldSize = 256, unsigned newSize = 30);

inline unsigned posToWin(unsigned pos, unsigned windowSize = 30u) { return (pos / windowSize) * windowSize; }

}  // namespace pdh
#endif
