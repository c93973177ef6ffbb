// Parameters of `popdel call`, its command line, and the start-up calculations that feed the kernels.
// Behavioural reference: popdel_call/parameter_parsing_popdel_call.h:141-210 (PopDelCallParameters),
// :231-369 (options); popdel_call/parameter_calculation_popdel_call.h:16-204; popdel_call/load_profile_popdel_call.h:52-190
// (regions of interest); insert_histogram_popdel.h:992-1150 (loading the per-read-group histograms).
#ifndef PDH_CALL_PARAMS_H_
#define PDH_CALL_PARAMS_H_

#include <cstdint>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "histogram.h"

namespace pdh {

struct Region
{
    std::string seqName;
    int32_t beginPos = -1;     // 0-based, inclusive
    int32_t endPos = -1;       // 0-based, exclusive
};

// LR -> probability that the call is correct; used for the QUAL column only.
class QuantileTable
{
public:
    explicit QuantileTable(double prior);
    double probability(double lr) const;
private:
    std::map<double, double> m_;
};

struct CallParams
{
    std::vector<std::string> inputFiles;
    std::string roiFile;
    std::string maxLoadFile;
    std::string outfile = "popdel.vcf";
    unsigned iterations = 15;
    double   prior = 0.0001;
    double   minimumLikelihoodRatio = 0;
    std::map<std::string, unsigned> readGroups;          // read-group id -> global index
    std::vector<std::vector<unsigned>> rgs;              // per sample: its read-group indices
    std::vector<InsertHistogram> histograms;             // per read group, conditioned
    bool modRgByFileName = false;
    bool representativeContigs = true;
    std::vector<unsigned> minInitDelLengths;
    unsigned minLen = std::numeric_limits<unsigned>::max();
    double   minRelWinCover = 0.5;
    unsigned windowSize = 30;
    unsigned windowShift = 0;
    unsigned windowBuffer = 200000;
    bool smoothing = true;
    std::vector<std::string> roiList;
    std::vector<Region> allRois;
    size_t nextRoi = 0;                                  // index into allRois
    bool outputFailed = false;
    QuantileTable quantileTable{0.0001};                 // always built from the default prior
    unsigned fileCount = 0;
    unsigned sampleNum = 0;
    std::vector<std::vector<std::string>> contigNames;   // one table per file (or one in total)
    std::vector<std::vector<int32_t>> contigLengths;
    unsigned maxDeletionSize = 10000;                    // parsed, never used by `call`
    double   minSampleFraction = 0.1;
    double   meanStddev = 0.0;
    bool windowWiseOutput = false;
    std::vector<unsigned> indexRegionSizes;
    unsigned defaultMaxLoad = 100;
    std::vector<unsigned> maxLoad;
    unsigned pseudoCountFraction = 500;
    bool uncompressedIn = false;
    bool somatic = false;
};

void printStatus(const std::string & msg);
inline void printStatus(const std::ostringstream & msg) { printStatus(msg.str()); }

// Returns -1 to continue, 0 after help/version output, 1 on a command-line error.
int parseCallCommandLine(CallParams & params, int argc, const char ** argv);

// Loads all profile headers, conditions the histograms and derives thresholds / regions of interest.
void loadAndCalculateParameters(CallParams & params);

bool parseRegion(Region & reg, const std::string & s);
int roundHalfUp(double d);
std::string sampleNameOf(const std::string & path);

}  // namespace pdh
#endif
