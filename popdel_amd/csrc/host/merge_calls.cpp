#include <algorithm>
#include <cmath>

#include "parallel.h"
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
    std::vector<unsigned> startPositions;
    std::vector<unsigned> sizeEstimates;
    long double lr = 0;
    unsigned callCount = 1;
    unsigned winCount = 1;
    unsigned significantWindows = 1;
};

// The windows of [start, end) that lie inside the head's deletion: the members whose per-sample columns are merged
// (reference utils_popdel.h:454-470).
std::vector<size_t> groupMembers(const std::vector<WindowCall> & calls, size_t start, size_t end)
{
    const WindowCall & head = calls[start];
    std::vector<size_t> members;
    for (size_t g = start; g < end; ++g)
    {
        const WindowCall & c = calls[g];
        if (c.windowPosition > head.position && c.windowPosition - 30 < head.position + head.deletionLength)
            members.push_back(g);
    }
    return members;
}

// Average the PLs and take per-component medians of LAD/DAD over the member windows. The per-sample part is independent
// from sample to sample and runs on the host threads.
void regenotype(std::vector<WindowCall> & calls, const MergeGroup & grp)
{
    WindowCall & head = calls[grp.head];
    const std::vector<size_t> & members = grp.members;
    const size_t S = head.pl.size();
    const unsigned count = static_cast<unsigned>(members.size());
    const size_t m = count / 2;                                    // upper median
    parallelFor(S, [&](size_t s) {
        std::array<unsigned, 3> g = {0, 0, 0};
        static thread_local std::vector<unsigned> col;
        col.resize(count);
        auto medianOf = [&](auto pick) {
            for (unsigned i = 0; i < count; ++i)
                col[i] = pick(calls[members[i]]);
            std::nth_element(col.begin(), col.begin() + m, col.end());
            return col[m];
        };
        for (size_t idx : members)
            for (int k = 0; k < 3; ++k)
                g[k] += calls[idx].pl[s][k];
        double minGt = std::min(std::min(g[0], g[1]), g[2]);
        double ref = static_cast<double>(g[0] - minGt) / count;
        double het = static_cast<double>(g[1] - minGt) / count;
        double hom = static_cast<double>(g[2] - minGt) / count;
        std::array<uint32_t, 3> pl = {static_cast<uint32_t>(std::round(ref)), static_cast<uint32_t>(std::round(het)),
                                      static_cast<uint32_t>(std::round(hom))};
        std::array<uint32_t, 3> lad;
        std::array<uint32_t, 5> dad;
        for (int k = 0; k < 3; ++k)
            lad[k] = medianOf([&](const WindowCall & c) { return c.lad[s][k]; });
        for (int k = 0; k < 5; ++k)
            dad[k] = medianOf([&](const WindowCall & c) { return c.dad[s][k]; });
        // Break ties between neighbouring genotypes.
        if (pl[1] == pl[2])
        {
            if (het > hom)
                ++pl[1];
            else
                ++pl[2];
        }
        else if (pl[0] == pl[1])
        {
            if (ref > het)
                ++pl[0];
            else
                ++pl[1];
        }
        // the head is a member of its own group in general: write only after all members have been read for this sample
        head.pl[s] = pl;
        head.lad[s] = lad;
        head.dad[s] = dad;
    }, 256);
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

void mergeRange(std::vector<WindowCall> & calls, size_t start, size_t last, MergeState & st, double r, std::vector<MergeGroup> & groups)
{
    WindowCall & head = calls[start];
    head.position = upperMedianAndClear(st.startPositions);
    head.deletionLength = upperMedianAndClear(st.sizeEstimates);
    head.likelihoodRatio = static_cast<double>(st.lr / st.winCount);
    // the per-sample columns of the head and its allele frequency are filled in afterwards (regenotype on the host, or
    // the backend's merge on the device): nothing below or in the grouping that follows reads them
    std::vector<size_t> members = groupMembers(calls, start, last);
    if (!members.empty())
    {
        groups.push_back(MergeGroup{start, std::move(members)});
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

bool unifyCallsScalar(std::vector<WindowCall> & calls, double stddev, double r, bool outputFailed, std::vector<MergeGroup> & groups,
                      size_t & firstKept)
{
    groups.clear();
    firstKept = 0;
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
                    mergeRange(calls, cur, it, st, r, groups);
                break;
            }
        }
        else
        {
            if (st.winCount != 1 && !st.startPositions.empty())
                mergeRange(calls, cur, it, st, r, groups);
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
    firstKept = firstGood;
    return true;
}

void regenotypeOnHost(std::vector<WindowCall> & calls, const std::vector<MergeGroup> & groups)
{
    for (const MergeGroup & g : groups)
    {
        regenotype(calls, g);
        frequencyFromGenotypes(calls[g.head]);
    }
}

void keepValidCalls(std::vector<WindowCall> & calls, size_t firstKept)
{
    std::vector<WindowCall> kept;
    for (size_t i = firstKept; i < calls.size(); ++i)
        if (calls[i].filter != kFilterInvalid)
            kept.push_back(std::move(calls[i]));
    calls.swap(kept);
}

bool unifyCalls(std::vector<WindowCall> & calls, double stddev, double r, bool outputFailed)
{
    std::vector<MergeGroup> groups;
    size_t firstKept = 0;
    if (!unifyCallsScalar(calls, stddev, r, outputFailed, groups, firstKept))
        return false;
    regenotypeOnHost(calls, groups);
    keepValidCalls(calls, firstKept);
    return true;
}

}  // namespace pdh
