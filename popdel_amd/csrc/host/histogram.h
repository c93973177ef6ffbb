// Insert-size histogram of one read group and its start-up conditioning (host side).
// Behavioural reference: insert_histogram_popdel.h:19-43 (Histogram), :609-645 (calculateQuantiles),
// :718-753 (normalisation), :759-780 (smoothing), :806-821 (minimum probability), :884-891 (densityScale),
// :974-986 (processHistogram), :1157-1163 (I()).
#ifndef PDH_HISTOGRAM_H_
#define PDH_HISTOGRAM_H_

#include <cstddef>
#include <cstdint>
#include <vector>

namespace pdh {

struct InsertHistogram
{
    std::vector<double> values;        // index 0 and the last index are reserved for irregular sizes
    double   minProb = 0.0;
    int      offset = 0;               // insert size of index 0
    double   stddev = 0.0;
    unsigned median = 0;
    unsigned readLength = 0;
    unsigned lowerQuantileDist = 0;    // |median - 1 % quantile|
    unsigned upperQuantileDist = 0;    // |99 % quantile - median|
    unsigned windowSize = 0;
};

// Density lookup for an insert-size deviation; out-of-table deviations get minProb.
inline double lookupDensity(const InsertHistogram & h, int deviation)
{
    int i = deviation + static_cast<int>(h.median) - h.offset;
    if (i <= 0 || i + 1 >= static_cast<int>(h.values.size()))
        return h.minProb;
    return h.values[static_cast<size_t>(i)];
}

// Full conditioning chain applied once per read group when `popdel call` starts.
void conditionHistogram(InsertHistogram & h, unsigned windowSize, bool smoothing, unsigned pseudoCountFraction);

}  // namespace pdh
#endif
