#include "histogram.h"

#include <cmath>
#include <cstdlib>

namespace pdh {
namespace {

// Gaussian-weighted moving average over +-20 positions. Edge windows are divided by the weight of the
// full kernel, not of the part inside the table (reference insert_histogram_popdel.h:759-780).
void smooth(InsertHistogram & h)
{
    const int len = static_cast<int>(h.values.size());
    std::vector<double> out(static_cast<size_t>(len));
    for (int i = 0; i < len; ++i)
    {
        double weighted = 0, norm = 0;
        for (int j = -20; j <= 20; ++j)
        {
            double k = std::exp(-j * j / 40.0);
            if (i + j >= 0 && i + j < len)
                weighted += k * h.values[static_cast<size_t>(i + j)];
            norm += k;
        }
        out[static_cast<size_t>(i)] = weighted / norm;
    }
    h.values.swap(out);
}

// 1 % / 99 % quantile distances from the median on the inner part of the table. The running total is an
// `unsigned` that is truncated after every addition, and the element at which the lower quantile is found is
// added a second time when the upper-quantile scan starts (reference insert_histogram_popdel.h:609-645).
void quantileDistances(InsertHistogram & h)
{
    const size_t n = h.values.size();
    if (n < 3)
        return;
    unsigned total = 0;
    for (size_t i = 1; i + 1 < n; ++i)
        total = static_cast<unsigned>(static_cast<double>(total) + h.values[i]);
    const double lower = total * 0.01;
    const double upper = total * 0.99;
    double sum = 0;
    size_t i = 1;
    for (; i + 1 < n; ++i)
    {
        sum += h.values[i];
        if (sum >= lower)
        {
            h.lowerQuantileDist = static_cast<unsigned>(std::abs(static_cast<int>(i) + h.offset - static_cast<int>(h.median)));
            break;
        }
    }
    for (; i + 1 < n; ++i)
    {
        sum += h.values[i];
        if (sum >= upper)
        {
            h.upperQuantileDist = static_cast<unsigned>(std::abs(static_cast<int>(i) + h.offset - static_cast<int>(h.median)));
            break;
        }
    }
}

// Counts -> densities; the divisor is again a truncating unsigned accumulator over *all* entries
// (reference insert_histogram_popdel.h:884-891).
void toDensity(std::vector<double> & v)
{
    unsigned total = 0;
    for (double x : v)
        total = static_cast<unsigned>(static_cast<double>(total) + x);
    for (double & x : v)
        x /= total;
}

// Window-size normalisation of the inner entries. Index 1 is paired with insert size `offset`
// (reference insert_histogram_popdel.h:718-753).
void normaliseForWindow(InsertHistogram & h)
{
    const size_t n = h.values.size();
    unsigned insertSize = static_cast<unsigned>(h.offset);
    for (size_t i = 1; i + 1 < n; ++i, ++insertSize)
    {
        int inner = static_cast<int>(insertSize - 2u * h.readLength);
        if (inner < 1)
            inner = 1;
        h.values[i] *= static_cast<double>(h.windowSize + inner - 1) / h.windowSize;
    }
}

void floorAtMinimum(InsertHistogram & h, unsigned pseudoCountFraction)
{
    double mx = 0;
    for (double x : h.values)
        if (x > mx)
            mx = x;
    h.minProb = mx / pseudoCountFraction;
    for (double & x : h.values)
        if (x < h.minProb)
            x = h.minProb;
}

}  // namespace

void conditionHistogram(InsertHistogram & h, unsigned windowSize, bool smoothing, unsigned pseudoCountFraction)
{
    h.windowSize = windowSize;
    if (smoothing)
        smooth(h);
    quantileDistances(h);
    toDensity(h.values);
    normaliseForWindow(h);
    floorAtMinimum(h, pseudoCountFraction);
}

}  // namespace pdh
