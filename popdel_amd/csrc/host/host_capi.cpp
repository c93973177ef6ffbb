// Small C exports of host-side start-up code for Python callers (bench.py / tests build synthetic cohorts in memory
// and need exactly the histogram conditioning that `popdel call` applies when it loads a profile header).
#include <cstring>
#include <vector>

#include "histogram.h"

extern "C" {

// In: raw counts `values[n]` with table offset / median / stddev / read length. Out: conditioned densities written
// back into `values`, plus min_prob and the two quantile distances.
int pdh_condition_histogram(double * values, unsigned n, int offset, unsigned median, double stddev, unsigned readLength,
                            int smoothing, unsigned pseudoCountFraction, double * minProbOut, unsigned * lowerQOut,
                            unsigned * upperQOut)
{
    if (!values || n < 3 || !minProbOut || !lowerQOut || !upperQOut)
        return 1;
    pdh::InsertHistogram h;
    h.values.assign(values, values + n);
    h.offset = offset;
    h.median = median;
    h.stddev = stddev;
    h.readLength = readLength;
    pdh::conditionHistogram(h, 256, smoothing != 0, pseudoCountFraction);
    std::memcpy(values, h.values.data(), n * sizeof(double));
    *minProbOut = h.minProb;
    *lowerQOut = h.lowerQuantileDist;
    *upperQOut = h.upperQuantileDist;
    return 0;
}

}  // extern "C"
