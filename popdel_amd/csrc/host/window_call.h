// One (window, candidate) deletion call as produced by the genotyping backend, and the post-kernel steps that
// turn a segment's window calls into VCF records.
// Behavioural reference: utils_popdel.h:30-124 (Dad, Call), :132-232 (filter bits), :237-654 (unifyCalls and
// helpers); popdel_call/vcfout_popdel_call.h:12-205 (header / record text), :207-291 (output bundle).
#ifndef PDH_WINDOW_CALL_H_
#define PDH_WINDOW_CALL_H_

#include <array>
#include <cstdint>
#include <fstream>
#include <string>
#include <vector>

#include "call_params.h"

namespace pdh {

struct WindowCall
{
    uint32_t initialLength = 0;
    uint32_t iterations = 0;
    uint32_t deletionLength = 0;
    double   likelihoodRatio = 0.0;
    double   frequency = 0.0;
    unsigned windowPosition = 0;
    unsigned position = 0;
    unsigned endPosition = 0;
    unsigned significantWindows = 0;
    uint8_t  filter = 0;
    std::vector<std::array<uint32_t, 3>> pl;         // phred-scaled GT likelihoods (ref, het, hom)
    std::vector<std::array<uint32_t, 3>> lad;        // likelihood-derived allelic depth
    std::vector<std::array<uint32_t, 5>> dad;        // distribution-derived allelic depth
    std::vector<std::array<uint32_t, 2>> firstLast;
    uint32_t sourceIndex = 0;                        // position in the backend's call list (pd_calls) of the segment
};

enum CallFilter : uint8_t
{
    kFilterLR = 1,
    kFilterCoverage = 2,
    kFilterSamples = 4,
    kFilterGT0 = 8,
    kFilterRelWinCov = 16,
    kFilterInvalid = 255
};

inline bool passesAll(const WindowCall & c) { return (c.filter & (kFilterLR | kFilterCoverage | kFilterSamples | kFilterGT0 | kFilterRelWinCov)) == 0; }

// Merge the window calls of one segment into variant calls (in place). Returns false if nothing is to be written.
bool unifyCalls(std::vector<WindowCall> & calls, double stddev, double minRelWinCover, bool outputFailed);

// The same in its two halves (reference utils_popdel.h:567-654 / :446-501, :339-359). unifyCallsScalar() sorts the calls
// and runs the grouping pass over call-level fields (positions, lengths, LR, filters -- accumulator quirks included); it
// leaves, per merged variant, the head and the member windows whose per-sample columns are to be merged (positions in the
// sorted vector). The per-sample half fills the heads' PL / LAD / DAD columns and allele frequency: regenotypeOnHost(),
// or the genotyping backend on the device (pd_calls_merge) when the calls carry no per-sample columns. keepValidCalls()
// then drops everything that was merged away.
struct MergeGroup
{
    size_t head;
    std::vector<size_t> members;
};
bool unifyCallsScalar(std::vector<WindowCall> & calls, double stddev, double minRelWinCover, bool outputFailed,
                      std::vector<MergeGroup> & groups, size_t & firstKept);
void regenotypeOnHost(std::vector<WindowCall> & calls, const std::vector<MergeGroup> & groups);
void keepValidCalls(std::vector<WindowCall> & calls, size_t firstKept);

class VcfOutput
{
public:
    VcfOutput(const std::string & outfile, const std::vector<std::string> & inputFiles,
              const std::vector<std::string> & contigNames, const std::vector<int32_t> & contigLengths);
    void appendContigName(const std::string & name);
    void unlockContigName();
    void writeCalls(const std::vector<WindowCall> & calls, const CallParams & params);
    void flush() { out_.flush(); }

private:
    std::ofstream out_;
    std::vector<std::string> contigNames_;
    unsigned rID_ = 0;
    bool locked_ = false;
};

// Text of one VCF data line (without trailing newline).
std::string formatRecord(const QuantileTable & q, const WindowCall & call, const std::string & contig);

}  // namespace pdh
#endif
