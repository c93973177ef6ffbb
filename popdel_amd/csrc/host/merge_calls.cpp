#include <algorithm>
#include <cmath>

#include "window_call.h"

namespace pdh {
namespace {

bool delSizeSimilar(unsigned a, unsigned b, double stddev, double f = 0.5)
{
    unsigned l = std::min(a, b), r = std::max(a, b);
    if (l + 2 * stddev >= r)
        return true;
    return l >= f * r;
}

bool inDelRange(const WindowCall & a, const WindowCall & b, double stddev, double f = 4)
{
    return (b.position - a.position < (std::min(a.deletionLength, b.deletionLength) + f * stddev));
}

// If one of the two spans is shorter than its estimated deletion and both lie close together, the head's end is
// moved to the other call's end (note the mutation of `a`).
bool extendIfShort(WindowCall & a, const WindowCall & b, double stddev)
{
    unsigned aSpan = a.endPosition - a.position;
    unsigned bSpan = b.endPosition - b.position;
    if ((aSpan < a.deletionLength || bSpan < b.deletionLength) && inDelRange(a, b, stddev))
    {
        a.endPosition = b.endPosition;
        return true;
    }
    return false;
}

bool enoughOverlap(const WindowCall & a, const WindowCall & b, double stddev, double f = 0.25)
{
    unsigned aSpan = a.endPosition - a.position;
    unsigned bSpan = b.endPosition - b.position;
    unsigned minLen = std::min(aSpan, bSpan);
    unsigned left = std::max(a.position, b.position);
    unsigned right = std::min(a.position + aSpan, b.position + bSpan);
    int overlap = static_cast<int>(right - left);
    if (overlap >= f * minLen)
        return true;
    return overlap + 2 * stddev >= minLen;
}

bool similarCalls(WindowCall & a, WindowCall & b, double stddev)
{
    if (!delSizeSimilar(a.deletionLength, b.deletionLength, stddev))
        return false;
    if (enoughOverlap(a, b, stddev))
        return true;
    return extendIfShort(a, b, stddev);
}

bool callOrder(const WindowCall & l, const WindowCall & r)
{
    if (l.position != r.position)
        return l.position < r.position;
    if (l.deletionLength != r.deletionLength)
        return l.deletionLength < r.deletionLength;
    return l.likelihoodRatio > r.likelihoodRatio;
}

unsigned upperMedianAndClear(std::vector<unsigned> & v)
{
    std::sort(v.begin(), v.end());
    unsigned m = v[v.size() / 2];
    v.clear();
    return m;
}

// Working state of unifyCalls() that survives from one merge group to the next.
struct MergeState
{
    std::vector<std::array<unsigned, 3>> genotypes;
    std::vector<unsigned> startPositions;
    std::vector<unsigned> sizeEstimates;
    std::vector<std::array<std::vector<unsigned>, 3>> lads;
    std::vector<std::array<std::vector<unsigned>, 5>> dads;
    long double lr = 0;
    unsigned callCount = 1;
    unsigned winCount = 1;
    unsigned significantWindows = 1;
};

// Average the PLs and take per-component medians of LAD/DAD over the windows of [start, end) that lie inside the
// deletion. Returns false if no window qualifies.
bool regenotype(std::vector<WindowCall> & calls, size_t start, size_t end, MergeState & st)
{
    WindowCall & head = calls[start];
    const size_t S = st.genotypes.size();
    unsigned count = 0;
    for (size_t g = start; g < end; ++g)
    {
        const WindowCall & c = calls[g];
        if (c.windowPosition > head.position && c.windowPosition - 30 < head.position + head.deletionLength)
        {
            for (size_t s = 0; s < S; ++s)
            {
                for (int k = 0; k < 3; ++k)
                {
                    st.genotypes[s][k] += c.pl[s][k];
                    st.lads[s][k].push_back(c.lad[s][k]);
                }
                for (int k = 0; k < 5; ++k)
                    st.dads[s][k].push_back(c.dad[s][k]);
            }
            ++count;
        }
    }
    if (count == 0)
        return false;
    for (size_t s = 0; s < S; ++s)
    {
        std::array<unsigned, 3> & g = st.genotypes[s];
        double minGt = std::min(std::min(g[0], g[1]), g[2]);
        double ref = static_cast<double>(g[0] - minGt) / count;
        double het = static_cast<double>(g[1] - minGt) / count;
        double hom = static_cast<double>(g[2] - minGt) / count;
        g = {0, 0, 0};
        head.pl[s][0] = static_cast<uint32_t>(std::round(ref));
        head.pl[s][1] = static_cast<uint32_t>(std::round(het));
        head.pl[s][2] = static_cast<uint32_t>(std::round(hom));
        {
            size_t m = st.lads[s][0].size() / 2;
            for (int k = 0; k < 3; ++k)
            {
                std::sort(st.lads[s][k].begin(), st.lads[s][k].end());
                head.lad[s][k] = st.lads[s][k][m];
                st.lads[s][k].clear();
            }
        }
        {
            size_t m = st.dads[s][0].size() / 2;
            for (int k = 0; k < 5; ++k)
            {
                std::sort(st.dads[s][k].begin(), st.dads[s][k].end());
                head.dad[s][k] = st.dads[s][k][m];
                st.dads[s][k].clear();
            }
        }
        // Break ties between neighbouring genotypes.
        if (head.pl[s][1] == head.pl[s][2])
        {
            if (het > hom)
                ++head.pl[s][1];
            else
                ++head.pl[s][2];
        }
        else if (head.pl[s][0] == head.pl[s][1])
        {
            if (ref > het)
                ++head.pl[s][0];
            else
                ++head.pl[s][1];
        }
    }
    return true;
}

void frequencyFromGenotypes(WindowCall & c)
{
    unsigned alleles = 0;
    for (const auto & p : c.pl)
    {
        if (p[0] == 0)
            continue;
        alleles += (p[1] == 0) ? 1 : 2;
    }
    c.frequency = static_cast<double>(alleles) / (c.pl.size() * 2);
}

void mergeRange(std::vector<WindowCall> & calls, size_t start, size_t last, MergeState & st, double r)
{
    WindowCall & head = calls[start];
    head.position = upperMedianAndClear(st.startPositions);
    head.deletionLength = upperMedianAndClear(st.sizeEstimates);
    head.likelihoodRatio = static_cast<double>(st.lr / st.winCount);
    if (regenotype(calls, start, last, st))
    {
        frequencyFromGenotypes(head);
        head.significantWindows = st.significantWindows;
        if (30.0 * st.significantWindows / head.deletionLength < r)
            head.filter |= kFilterRelWinCov;
        st.winCount = 1;
        st.significantWindows = 1;
        st.lr = 0.0;
        ++st.callCount;
    }
    else
    {
        head.filter = kFilterInvalid;
    }
}

}  // namespace

bool unifyCalls(std::vector<WindowCall> & calls, double stddev, double r, bool outputFailed)
{
    if (calls.size() <= 1u)
        return false;
    std::sort(calls.begin(), calls.end(), callOrder);
    size_t cur = 0;
    const size_t last = calls.size() - 1;
    if (!outputFailed)
    {
        while (!passesAll(calls[cur]))
        {
            if (cur == last)
                return false;
            ++cur;
        }
        if (cur == last)
            return false;
    }
    const size_t firstGood = cur;
    MergeState st;
    st.genotypes.assign(calls[cur].pl.size(), {0, 0, 0});
    st.lads.resize(calls[cur].lad.size());
    st.dads.resize(calls[cur].dad.size());
    st.startPositions.push_back(calls[cur].position);
    st.sizeEstimates.push_back(calls[cur].deletionLength);
    st.lr = calls[cur].likelihoodRatio;

    size_t it = firstGood + 1;
    while (true)
    {
        if (similarCalls(calls[cur], calls[it], stddev))
        {
            if (passesAll(calls[it]))
            {
                st.startPositions.push_back(calls[it].position);
                st.sizeEstimates.push_back(calls[it].deletionLength);
                ++st.significantWindows;
            }
            ++st.winCount;
            st.lr += calls[it].likelihoodRatio;
            calls[it].filter = kFilterInvalid;
            if (it == last)
            {
                if (!st.startPositions.empty())
                    mergeRange(calls, cur, it, st, r);
                break;
            }
        }
        else
        {
            if (st.winCount != 1 && !st.startPositions.empty())
                mergeRange(calls, cur, it, st, r);
            else
                calls[cur].filter = kFilterInvalid;
            cur = it;
        }
        if (it != last)
        {
            ++it;
        }
        else
        {
            if (st.winCount == 1)
            {
                --st.callCount;
                calls[cur].filter = kFilterInvalid;
            }
            break;
        }
    }
    std::vector<WindowCall> kept;
    for (size_t i = firstGood; i <= last; ++i)
        if (calls[i].filter != kFilterInvalid)
            kept.push_back(std::move(calls[i]));
    calls.swap(kept);
    return true;
}

}  // namespace pdh
