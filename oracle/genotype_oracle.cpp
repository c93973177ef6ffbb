// ORACLE -- TEST INFRASTRUCTURE ONLY. Not part of the product; nothing under popdel_amd/ may link or call this.
//
// CPU restatement (plain C++, x87 `long double` where the reference uses it) of PopDel's per-window joint-calling
// algorithm, exposed through the same C ABI as the HIP library (include/popdel_hip.h) so that tests can run the
// identical host driver against either implementation and compare.
//
// Parity pin: the reference itself cannot be built in this image under the project rules (it needs htslib, which is
// absent). The restatement is pinned against outputs that the survey stage captured from the unmodified reference
// (tests/golden/README.md lists them and their provenance); see DESIGN.md "Oracle".
//
// Every function cites the reference lines it follows (paths relative to the reference tree).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "popdel_hip.h"
#include "oracle_side_channel.h"
#include "chrom_profile.h"      // pdh::WalkReference (POPDEL_ORACLE_WALK, see pd_walk_submit below)

namespace {

struct ReadGroup
{
    unsigned median = 0, readLength = 0;
    double stddev = 0, minProb = 0;
    int offset = 0;
    std::vector<double> values;
    unsigned lowerQuantileDist = 0, upperQuantileDist = 0;
    unsigned maxLoad = 0, minInitDelLen = 0;
};

struct LTriple
{
    long double i1 = 0, i2 = 0, i3 = 0;
};

struct DTriple
{
    double i1 = 0, i2 = 0, i3 = 0;
};

struct ActivePair
{
    int dev;
    unsigned start;
};

// insert_histogram_popdel.h:1157-1163
inline double I(const ReadGroup & h, int deviation)
{
    int i = deviation + static_cast<int>(h.median) - h.offset;
    if (i <= 0 || i + 1 >= static_cast<int>(h.values.size()))
        return h.minProb;
    return h.values[static_cast<size_t>(i)];
}

// utils_popdel.h:1410-1413
inline int roundHalfUp(double d) { return static_cast<int>(std::floor(d + 0.5)); }

// Conversion of a double to int the way x86 does it for values C++ leaves undefined.
inline int toIntX86(double d)
{
    if (!(d > -2147483649.0 && d < 2147483648.0))
        return INT_MIN;
    return static_cast<int>(d);
}

struct OracleCall
{
    uint32_t initialLength, iterations, deletionLength;
    double likelihoodRatio, frequency;
    uint32_t windowPosition, position, endPosition;
    uint8_t filter;
    uint32_t nearIntegerShifts;       // instrumentation (pd_calls::near_integer_shifts), not a field of the reference's Call
    std::vector<uint32_t> pl, lad, dad, firstLast;
};

}  // namespace

struct pd_ctx
{
    pd_config cfg;
    std::vector<ReadGroup> rgs;
    std::vector<uint32_t> sampleOffsets;
    std::string error;
    // current window state
    std::vector<std::vector<ActivePair>> active;      // per read group
    unsigned currentPos = 0;
    // results
    std::vector<OracleCall> calls;
    std::vector<int32_t> q3;
    std::vector<uint8_t> flags;
    uint32_t lastWindows = 0;
    uint64_t ticket = 0;
    // staged segments (deep copies), see pd_segment_stage()
    struct Staged
    {
        pd_segment seg;
        std::vector<uint64_t> offsets;
        std::vector<int32_t> dev;
        std::vector<uint32_t> start, on, off;
    };
    std::map<uint64_t, Staged> staged;
    uint64_t nextHandle = 1;
    // The reference's active sets themselves (only when the driver feeds the mutation log, oracle_side_channel.h): real
    // std::unordered_set<uint32_t> per read group, alive across segments like the reference's, iterated for the order
    // in which the pairs of a window are visited.
    std::vector<std::unordered_set<uint32_t>> refActive;
    pd_timing timing;
    std::unique_ptr<pdh::WalkReference> walk;      // POPDEL_ORACLE_WALK: the closed-form walk's state from segment to segment

    unsigned nSamples() const { return static_cast<unsigned>(sampleOffsets.size() - 1); }
    bool isHighCov(unsigned rg) const { return active[rg].size() >= rgs[rg].maxLoad; }
};

namespace {

std::string g_createError;

// genotype_deletion_popdel_call.h:15-27
int upperHalfMedian(std::vector<int> & values)
{
    std::sort(values.begin(), values.end());
    unsigned n = static_cast<unsigned>(values.size());
    if (n == 0)
        return 0;
    if (n < 4)
        return roundHalfUp(values[n - 1]);
    double pos = (3.0 * n + 2.0 + (n % 2)) / 4.0 - 1.0;
    unsigned l = static_cast<unsigned>(pos);
    double r = pos - l;
    return roundHalfUp((1 - r) * values[l] + r * values[l + 1]);
}

// profile_structure_popdel_call.h:1443-1478
unsigned activeDeviations(const pd_ctx & c, std::vector<int> & deviations, unsigned sample, unsigned minCov)
{
    unsigned readCount = 0, highCovReads = 0;
    for (uint32_t rg = c.sampleOffsets[sample]; rg < c.sampleOffsets[sample + 1]; ++rg)
    {
        unsigned n = static_cast<unsigned>(c.active[rg].size());
        readCount += n;
        if (n >= c.rgs[rg].maxLoad)
            highCovReads += n;
    }
    if (readCount < minCov)
        return readCount;
    deviations.reserve(readCount - highCovReads);
    for (uint32_t rg = c.sampleOffsets[sample]; rg < c.sampleOffsets[sample + 1]; ++rg)
    {
        if (c.isHighCov(rg))
            continue;
        for (const ActivePair & p : c.active[rg])
            deviations.push_back(p.dev);
    }
    return readCount;
}

// genotype_deletion_popdel_call.h:33-87
std::set<int> initializeDeletionLengths(pd_ctx & c, std::vector<char> & lowCoverage, int32_t * q3Out, uint8_t * flagOut)
{
    const unsigned S = c.nSamples();
    std::vector<int> deviations;
    deviations.reserve(S);
    for (unsigned i = 0; i < S; ++i)
    {
        std::vector<int> sampleValues;
        unsigned cov = activeDeviations(c, sampleValues, i, 2u);
        if (q3Out)
        {
            q3Out[i] = 0;
            flagOut[i] = 0;
        }
        if (cov < 2u)
        {
            lowCoverage[i] = 1;
            if (flagOut)
                flagOut[i] |= 1;
            continue;
        }
        if (sampleValues.empty())
            continue;
        int q = upperHalfMedian(sampleValues);
        deviations.push_back(q);
        if (q3Out)
        {
            q3Out[i] = q;
            flagOut[i] |= 2;
        }
    }
    std::set<int> lengths;
    if (deviations.empty())
        return lengths;
    std::sort(deviations.begin(), deviations.end());
    // The threshold array is indexed by position in the sorted array (not by read group).
    int sum = deviations[0];
    int n = 1;
    unsigned threshold = c.rgs[0].minInitDelLen;
    for (unsigned i = 1; i < deviations.size(); ++i)
    {
        if (deviations[i - 1] + 50 > deviations[i])
        {
            sum += deviations[i];
            ++n;
            threshold = std::min(threshold, c.rgs[i].minInitDelLen);
        }
        else
        {
            if (sum / n > static_cast<int>(threshold))
                lengths.insert(sum / n);
            sum = deviations[i];
            n = 1;
            threshold = c.rgs[i].minInitDelLen;
        }
    }
    if (sum / n > static_cast<int>(threshold))
        lengths.insert(sum / n);
    return lengths;
}

// genotype_deletion_popdel_call.h:93-133
double initializeAlleleFrequency(const pd_ctx & c, unsigned deletionLength)
{
    unsigned globalCount = 0, globalTotal = 0;
    for (unsigned s = 0; s < c.nSamples(); ++s)
    {
        unsigned count = 0, total = 0;
        for (uint32_t rg = c.sampleOffsets[s]; rg < c.sampleOffsets[s + 1]; ++rg)
        {
            unsigned n = static_cast<unsigned>(c.active[rg].size());
            if (n == 0 || n >= c.rgs[rg].maxLoad)
                continue;
            total += n;
            int windowBegin = std::max(static_cast<int>(deletionLength) / 2, roundHalfUp(deletionLength - 2 * c.rgs[rg].stddev));
            int windowEnd = static_cast<int>(deletionLength + 2 * c.rgs[rg].stddev);
            for (const ActivePair & p : c.active[rg])
                if (p.dev > windowBegin && p.dev < windowEnd)
                    ++count;
        }
        globalCount += count;
        globalTotal += total;
    }
    if (globalTotal == 0u)
        return 0.0;
    return static_cast<double>(globalCount) / globalTotal;
}

void referenceFallback(LTriple & res)
{
    res.i1 = 1;
    res.i2 = 0.0000000001;
    res.i3 = 0.0000000001;
}

// genotype_deletion_popdel_call.h:179-253
LTriple dataLikelihoods(const pd_ctx & c, std::vector<LTriple> & rgWise, unsigned sample, unsigned deletionLength,
                        const std::vector<int> & referenceShifts)
{
    LTriple logL;
    for (uint32_t rg = c.sampleOffsets[sample]; rg < c.sampleOffsets[sample + 1]; ++rg)
    {
        LTriple & cur = rgWise[rg];
        cur = LTriple();
        if (c.isHighCov(rg))
            continue;
        const int refShift = referenceShifts[rg];
        const ReadGroup & hist = c.rgs[rg];
        for (const ActivePair & p : c.active[rg])
        {
            long double refLikelihood = I(hist, p.dev - refShift);
            long double delLikelihood = I(hist, p.dev - static_cast<int>(deletionLength));
            long double g0 = logl(refLikelihood);
            long double g1 = logl(refLikelihood + delLikelihood) - std::log(2.0);
            long double g2 = logl(delLikelihood);
            cur.i1 += g0;
            cur.i2 += g1;
            cur.i3 += g2;
            logL.i1 += g0;
            logL.i2 += g1;
            logL.i3 += g2;
        }
        long double mx = std::max(std::max(cur.i1, cur.i2), cur.i3);
        cur.i1 = expl(cur.i1 - mx);
        cur.i2 = expl(cur.i2 - mx);
        cur.i3 = expl(cur.i3 - mx);
    }
    long double mx = std::max(std::max(logL.i1, logL.i2), logL.i3);
    LTriple res;
    res.i1 = expl(logL.i1 - mx);
    res.i2 = expl(logL.i2 - mx);
    res.i3 = expl(logL.i3 - mx);
    if (res.i1 == 0 || res.i2 == 0 || res.i3 == 0)
        referenceFallback(res);
    if (res.i1 == res.i2 && res.i1 == res.i3)
        referenceFallback(res);
    return res;
}

// genotype_deletion_popdel_call.h:137-172
void assignDad(uint32_t * dad, int deviation, int refUpper, int delLower, int delUpper)
{
    if (deviation > refUpper)
    {
        if (deviation < delLower)
            ++dad[2];               // between
        else if (deviation <= delUpper)
            ++dad[3];               // alt
        else
            ++dad[4];               // right
    }
    else
    {
        if (deviation < delUpper)
            ++dad[0];               // ref
        else
            ++dad[1];               // both
    }
}

// genotype_deletion_popdel_call.h:255-337 with profile_structure_popdel_call.h:1520-1586
LTriple finalDataLikelihoods(const pd_ctx & c, LTriple & gtLogs, uint32_t * lad, uint32_t * dad, uint32_t * firstLast,
                             std::vector<unsigned> & suppFirst, std::vector<unsigned> & suppLast, unsigned sample,
                             unsigned deletionLength, const std::vector<int> & referenceShifts)
{
    LTriple logL;
    gtLogs = LTriple();
    int delLowerBorder = INT_MAX;
    int delUpperBorder = 0;
    for (uint32_t rg = c.sampleOffsets[sample]; rg < c.sampleOffsets[sample + 1]; ++rg)
    {
        if (c.isHighCov(rg))
            continue;
        const ReadGroup & hist = c.rgs[rg];
        const int refShift = referenceShifts[rg];
        delLowerBorder = static_cast<int>(deletionLength - hist.lowerQuantileDist);
        delUpperBorder = static_cast<int>(deletionLength + hist.upperQuantileDist);
        for (const ActivePair & p : c.active[rg])
        {
            assignDad(dad, p.dev, static_cast<int>(hist.upperQuantileDist), delLowerBorder, delUpperBorder);
            long double refLikelihood = I(hist, p.dev - refShift);
            long double delLikelihood = I(hist, p.dev - static_cast<int>(deletionLength));
            if (refLikelihood >= 2 * delLikelihood)
                ++lad[0];
            else if (delLikelihood >= 2 * refLikelihood)
                ++lad[2];
            else
                ++lad[1];
            logL.i1 += logl(refLikelihood);
            gtLogs.i1 += log10l(refLikelihood);
            logL.i2 += logl(refLikelihood + delLikelihood) - std::log(2.0);
            gtLogs.i2 += log10l(refLikelihood + delLikelihood) - std::log10(2.0);
            logL.i3 += logl(delLikelihood);
            gtLogs.i3 += log10l(delLikelihood);
        }
    }
    // lowest start / highest end of all active pairs of the sample
    {
        unsigned mn = UINT_MAX, mx = 0;
        for (uint32_t rg = c.sampleOffsets[sample]; rg < c.sampleOffsets[sample + 1]; ++rg)
        {
            if (c.isHighCov(rg))
                continue;
            const ReadGroup & hist = c.rgs[rg];
            for (const ActivePair & p : c.active[rg])
            {
                int insertSize = p.dev + static_cast<int>(hist.median);
                insertSize = std::max(0, insertSize - 2 * static_cast<int>(hist.readLength));
                unsigned lastEnd = p.start + static_cast<unsigned>(insertSize);
                mn = std::min(mn, p.start);
                mx = std::max(mx, lastEnd);
            }
        }
        if (mn == UINT_MAX)
            mn = 0;
        firstLast[0] = mn;
        firstLast[1] = mx;
    }
    // supporting pairs: the borders are those of the sample's last non-high-coverage read group
    for (uint32_t rg = c.sampleOffsets[sample]; rg < c.sampleOffsets[sample + 1]; ++rg)
    {
        if (c.isHighCov(rg))
            continue;
        const ReadGroup & hist = c.rgs[rg];
        for (const ActivePair & p : c.active[rg])
        {
            if (p.dev >= delLowerBorder && p.dev <= delUpperBorder)
            {
                int insertSize = std::max(0, p.dev + static_cast<int>(hist.median) - 2 * static_cast<int>(hist.readLength));
                suppFirst.push_back(p.start);
                suppLast.push_back(p.start + static_cast<unsigned>(insertSize));
            }
        }
    }
    LTriple res;
    if (gtLogs.i1 + gtLogs.i2 + gtLogs.i3 == 0.0)
    {
        referenceFallback(res);
        return res;
    }
    long double maxGt = std::max(std::max(gtLogs.i1, gtLogs.i2), gtLogs.i3);
    long double maxDl = std::max(std::max(logL.i1, logL.i2), logL.i3);
    gtLogs.i1 -= maxGt;
    gtLogs.i2 -= maxGt;
    gtLogs.i3 -= maxGt;
    if (gtLogs.i1 == gtLogs.i2 && gtLogs.i1 == gtLogs.i3)
    {
        gtLogs.i1 = 0;
        gtLogs.i2 = -10;
        gtLogs.i3 = -10;
    }
    res.i1 = expl(logL.i1 - maxDl);
    res.i2 = expl(logL.i2 - maxDl);
    res.i3 = expl(logL.i3 - maxDl);
    if (res.i1 == 0 || res.i2 == 0 || res.i3 == 0)
        referenceFallback(res);
    if (res.i1 == res.i2 && res.i1 == res.i3)
        referenceFallback(res);
    return res;
}

// genotype_deletion_popdel_call.h:343-380
DTriple genotypePriors(double f, bool somatic)
{
    const double pseudo = 0.0000000001;
    DTriple g;
    if (!somatic)
    {
        g.i1 = std::max((1 - f) * (1 - f), pseudo);
        g.i2 = std::max(2 * f * (1 - f), pseudo);
        g.i3 = std::max(f * f, pseudo);
    }
    else if (f <= 0.4)
    {
        g.i1 = std::max(1 - 2 * f + pseudo, pseudo);
        g.i2 = std::max(2 * f - 2 * pseudo, pseudo);
        g.i3 = pseudo;
    }
    else if (f < 0.75)
    {
        g.i1 = pseudo;
        g.i2 = 1.;
        g.i3 = pseudo;
    }
    else
    {
        g.i1 = pseudo;
        g.i2 = pseudo;
        g.i3 = 1.;
    }
    return g;
}

static bool g_snapShifts = false;    // POPDEL_SNAP_SHIFTS, read per segment
static uint32_t g_nearIntegerShifts = 0;   // quotients within 1e-9 of an integer since the current cell's EM began (reported per call)
static int g_perturbFreq = 0;        // POPDEL_ORACLE_PERTURB_FREQ, read per segment
static bool g_traceKnife = false;   // set while the window named by POPDEL_TRACE_POS is processed

// genotype_deletion_popdel_call.h:388-462. The per-read-group weights are always taken from rgWise[0].
unsigned updateDeletionLength(const pd_ctx & c, const std::vector<LTriple> & dataL, const std::vector<LTriple> & rgWise,
                              const DTriple & gt, int deletionLength, std::vector<int> & referenceShifts)
{
    double sumDel = 0, weightedSumDel = 0;
    const LTriple & rgDl = rgWise[0];
    for (unsigned s = 0; s < c.nSamples(); ++s)
    {
        const LTriple & dl = dataL[s];
        double aSum = dl.i1 * gt.i1 + dl.i2 * gt.i2 + dl.i3 * gt.i3;
        double a1 = logl(dl.i2) + std::log(gt.i2) - std::log(aSum);
        double a2 = logl(dl.i3) + std::log(gt.i3) - std::log(aSum);
        for (uint32_t rg = c.sampleOffsets[s]; rg < c.sampleOffsets[s + 1]; ++rg)
        {
            if (c.isHighCov(rg))
                continue;
            double sumRef = 0, weightedSumRef = 0;
            double aSumRg = rgDl.i1 * gt.i1 + rgDl.i2 * gt.i2 + rgDl.i3 * gt.i3;
            long double l = logl(rgDl.i1);
            long double g = std::log(gt.i1);
            long double a = std::log(aSumRg);
            long double a0Rg = l + g - a;
            long double a1Rg = logl(rgDl.i2) + std::log(gt.i2) - std::log(aSumRg);
            const ReadGroup & hist = c.rgs[rg];
            for (const ActivePair & p : c.active[rg])
            {
                double del = I(hist, p.dev - deletionLength);
                double noDel = I(hist, p.dev - referenceShifts[rg]);
                double probDel = std::exp(a1) * del / (del + noDel) + std::exp(a2);
                double probRef = expl(a1Rg) * noDel / (del + noDel) + expl(a0Rg);
                sumDel += probDel;
                sumRef += probRef;
                weightedSumDel += probDel * p.dev;
                weightedSumRef += probRef * p.dev;
            }
            double quotient = weightedSumRef / sumRef;
            {
                const double r = std::rint(quotient);
                if (std::fabs(quotient - r) <= 1e-9 * std::max(1.0, std::fabs(r)))
                    ++g_nearIntegerShifts;
            }
            if (g_snapShifts)
            {
                // Test knob POPDEL_SNAP_SHIFTS (the test build libpopdel_hip_snap.so does the same): a quotient within 1e-9 of an integer is
                // moved onto it, which takes the last-bit dependence out of the truncation below.
                const double r = std::rint(quotient);
                if (std::fabs(quotient - r) <= 1e-9 * std::max(1.0, std::fabs(r)))
                    quotient = r;
            }
            referenceShifts[rg] = toIntX86(quotient);
            if (g_traceKnife)
            {
                // debugging aid: quotients within a few ulp of an integer are the order-/x87-sensitive truncations
                const double q = weightedSumRef / sumRef, r = std::nearbyint(q);
                if (r != 0 && std::fabs(q - r) <= 64 * 2.220446049250313e-16 * std::fabs(r))
                    fprintf(stderr, "[oracle trace]    knife edge: rg %u pairs %zu quotient %.17g (sigma %.3f)\n", rg,
                            c.active[rg].size(), q, hist.stddev);
            }
            if (referenceShifts[rg] > hist.stddev || referenceShifts[rg] < -1 * hist.stddev)
                referenceShifts[rg] = 0;
        }
    }
    if (sumDel == 0)
        return 0;
    double len = weightedSumDel / sumDel;
    if (len < 0)
        return 0;
    return static_cast<unsigned>(std::round(len));
}

// genotype_deletion_popdel_call.h:467-485
double updateAlleleFrequency(const std::vector<LTriple> & dataL, const DTriple & gt)
{
    double sum = 0;
    for (const LTriple & d : dataL)
    {
        double p0 = d.i1 * gt.i1;
        double p1 = d.i2 * gt.i2;
        double p2 = d.i3 * gt.i3;
        double pAll = p0 + p1 + p2;
        sum += (p1 + 2 * p2) / pAll;
    }
    return sum / 2.0 / dataL.size();
}

// genotype_deletion_popdel_call.h:490-508
double likelihoodRatio(const std::vector<LTriple> & dataL, const DTriple & gt)
{
    double del = 0, noDel = 0;
    for (const LTriple & d : dataL)
    {
        double p0 = d.i1 * gt.i1;
        double p1 = d.i2 * gt.i2;
        double p2 = d.i3 * gt.i3;
        double pAll = p0 + p1 + p2;
        double a0 = p0 / pAll;
        double a1 = p1 / pAll;
        double a2 = p2 / pAll;
        del += logl(a0 * d.i1 + a1 * d.i2 + a2 * d.i3);
        noDel += logl(d.i1);
    }
    return del - noDel;
}

// utils_popdel.h:1511-1528: exp() is applied to log10-based values.
void phredGL(uint32_t * pl, const LTriple & gtLogs)
{
    long double g0 = gtLogs.i1, g1 = gtLogs.i2, g2 = gtLogs.i3;
    const long double gTot = log10l(expl(g0) + expl(g1) + expl(g2));
    g0 = -10 * (g0 - gTot);
    g1 = -10 * (g1 - gTot);
    g2 = -10 * (g2 - gTot);
    const long double minPhred = std::min(std::min(g0, g1), g2);
    pl[0] = static_cast<uint32_t>(roundl(g0 - minPhred));
    pl[1] = static_cast<uint32_t>(roundl(g1 - minPhred));
    pl[2] = static_cast<uint32_t>(roundl(g2 - minPhred));
}

// genotype_deletion_popdel_call.h:536-730
void genotypeWindow(pd_ctx & c, int32_t * q3Out, uint8_t * flagOut)
{
    const unsigned S = c.nSamples();
    std::vector<char> lowCoverage(S, 0);
    std::set<int> lengths = initializeDeletionLengths(c, lowCoverage, q3Out, flagOut);
    if (lengths.empty())
        return;
    std::vector<int> referenceShifts(c.rgs.size());
    static const char * tracePos = getenv("POPDEL_TRACE_POS");   // debugging aid: per-iteration trace of one window
    g_traceKnife = tracePos && *tracePos && static_cast<uint32_t>(atol(tracePos)) == c.currentPos;
    for (int initLen : lengths)
    {
        std::fill(referenceShifts.begin(), referenceShifts.end(), 0);
        g_nearIntegerShifts = 0;
        std::map<int, double> visited;
        double freq = initializeAlleleFrequency(c, static_cast<unsigned>(initLen));
        if (freq == 0)
            continue;
        std::vector<LTriple> rgWise(c.rgs.size());
        std::vector<LTriple> dataL(S);
        for (unsigned i = 0; i < S; ++i)
            dataL[i] = dataLikelihoods(c, rgWise, i, static_cast<unsigned>(initLen), referenceShifts);
        DTriple gt = genotypePriors(freq, c.cfg.somatic != 0);
        unsigned len = static_cast<unsigned>(initLen);
        double prevFreq = freq;
        unsigned prevLen = len;
        DTriple prevGt = gt;
        const std::vector<int> prevShifts = referenceShifts;      // captured once, never refreshed
        uint32_t iterations = 0;
        while (len >= c.cfg.min_len && iterations < c.cfg.iterations)
        {
            ++iterations;
            visited[static_cast<int>(len)] = freq;
            prevLen = len;
            prevFreq = freq;
            prevGt = gt;
            len = updateDeletionLength(c, dataL, rgWise, gt, static_cast<int>(len), referenceShifts);
            for (unsigned i = 0; i < S; ++i)
                dataL[i] = dataLikelihoods(c, rgWise, i, len, referenceShifts);
            freq = updateAlleleFrequency(dataL, gt);
            if (g_perturbFreq != 0)
            {
                // Test knob POPDEL_ORACLE_PERTURB_FREQ (tests/test_oracle_golden.py): the allele frequency -- a sum over all samples
                // of terms derived from x87 exp/log results -- moved by ONE unit in the last place (1: up, 2: down, 3:
                // pseudo-random direction). Shows which outputs of the reference's algorithm hang off the last bit of that sum.
                static unsigned lcg = 12345u;
                lcg = lcg * 1664525u + 1013904223u;
                const bool up = g_perturbFreq == 1 || (g_perturbFreq == 3 && ((lcg >> 8) & 1u));
                freq = std::nextafter(freq, up ? 1.0 : 0.0);
            }
            if (freq == 0)
                break;
            gt = genotypePriors(freq, c.cfg.somatic != 0);
            if (tracePos && *tracePos && static_cast<uint32_t>(atol(tracePos)) == c.currentPos)
            {
                long shiftSum = 0, shiftNonZero = 0;
                for (int sft : referenceShifts)
                {
                    shiftSum += sft;
                    shiftNonZero += sft != 0;
                }
                fprintf(stderr, "[oracle trace] pos %u init %d it %u len %u freq %.17g shifts sum %ld nonzero %ld\n", c.currentPos,
                        initLen, iterations, len, freq, shiftSum, shiftNonZero);
            }
            auto v = visited.find(static_cast<int>(len));
            if (v != visited.end() && std::fabs(v->second - freq) <= 0.0001)
            {
                double logLR = likelihoodRatio(dataL, gt);
                for (unsigned i = 0; i < S; ++i)
                    dataL[i] = dataLikelihoods(c, rgWise, i, prevLen, prevShifts);
                double prevLogLR = likelihoodRatio(dataL, prevGt);
                if (prevLogLR > logLR)
                {
                    len = prevLen;
                    freq = prevFreq;
                    referenceShifts = prevShifts;
                }
                break;
            }
        }
        if (freq < 0.0000000001 || len < c.cfg.min_len)
            continue;
        std::vector<LTriple> gtLogs(S);
        OracleCall call;
        call.lad.assign(static_cast<size_t>(S) * 3, 0);
        call.dad.assign(static_cast<size_t>(S) * 5, 0);
        call.firstLast.assign(static_cast<size_t>(S) * 2, 0);
        std::vector<unsigned> suppFirst, suppLast;
        for (unsigned i = 0; i < S; ++i)
            dataL[i] = finalDataLikelihoods(c, gtLogs[i], &call.lad[i * 3], &call.dad[i * 5], &call.firstLast[i * 2], suppFirst,
                                            suppLast, i, len, referenceShifts);
        // 80 % / 20 % order statistics of the supporting starts / ends (genotype_deletion_popdel_call.h:514-529)
        if (suppFirst.empty())
            continue;
        unsigned li = static_cast<unsigned>(std::round(static_cast<double>(suppFirst.size() - 1) * 0.8));
        unsigned ri = static_cast<unsigned>(std::round(static_cast<double>(suppLast.size() - 1) * (1 - 0.8)));
        std::sort(suppFirst.begin(), suppFirst.end());
        std::sort(suppLast.begin(), suppLast.end());
        unsigned suppStart = suppFirst[li], suppEnd = suppLast[ri];
        if (suppStart == 0 && suppEnd == 0)
            continue;
        double logLR = likelihoodRatio(dataL, gt);
        if (!(logLR >= c.cfg.min_likelihood_ratio))
            continue;
        call.initialLength = static_cast<uint32_t>(initLen);
        call.iterations = iterations;
        call.nearIntegerShifts = g_nearIntegerShifts;
        call.deletionLength = len;
        call.likelihoodRatio = logLR;
        call.frequency = freq;
        call.windowPosition = c.currentPos - 1;
        if (c.cfg.window_wise_output)
        {
            call.position = c.currentPos - 1;
            call.endPosition = 0;
        }
        else
        {
            call.position = suppStart;
            call.endPosition = suppEnd;
        }
        call.pl.assign(static_cast<size_t>(S) * 3, 0);
        for (unsigned i = 0; i < S; ++i)
            phredGL(&call.pl[i * 3], gtLogs[i]);
        call.filter = 0;
        unsigned dataSamples = S;
        for (unsigned i = 0; i < S; ++i)
        {
            if (lowCoverage[i])
            {
                call.pl[i * 3] = call.pl[i * 3 + 1] = call.pl[i * 3 + 2] = 0;
                --dataSamples;
            }
        }
        if (!(static_cast<double>(dataSamples) / S >= c.cfg.min_sample_fraction))
            call.filter |= 4;
        c.calls.push_back(std::move(call));
    }
}

template <typename T>
T * copyArray(const std::vector<T> & v)
{
    T * p = static_cast<T *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (!v.empty())
        std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

extern "C" {

int pd_ctx_create(const pd_config * cfg, const pd_readgroup * rgs, uint32_t n_readgroups, const uint32_t * sample_rg_offsets,
                  uint32_t n_samples, int /*device*/, pd_ctx ** out)
{
    if (!cfg || !rgs || !sample_rg_offsets || !out || n_readgroups == 0 || n_samples == 0)
    {
        g_createError = "pd_ctx_create: invalid argument";
        return 1;
    }
    if (sample_rg_offsets[0] != 0 || sample_rg_offsets[n_samples] != n_readgroups)
    {
        g_createError = "pd_ctx_create: sample_rg_offsets do not cover the read groups";
        return 1;
    }
    pd_ctx * c = new pd_ctx();
    c->cfg = *cfg;
    c->rgs.resize(n_readgroups);
    for (uint32_t i = 0; i < n_readgroups; ++i)
    {
        ReadGroup & r = c->rgs[i];
        r.median = rgs[i].median;
        r.readLength = rgs[i].read_length;
        r.stddev = rgs[i].stddev;
        r.minProb = rgs[i].min_prob;
        r.offset = rgs[i].hist_offset;
        r.values.assign(rgs[i].hist_values, rgs[i].hist_values + rgs[i].hist_len);
        r.lowerQuantileDist = rgs[i].lower_quantile_dist;
        r.upperQuantileDist = rgs[i].upper_quantile_dist;
        r.maxLoad = rgs[i].max_load;
        r.minInitDelLen = rgs[i].min_init_del_len;
    }
    c->sampleOffsets.assign(sample_rg_offsets, sample_rg_offsets + n_samples + 1);
    c->active.resize(n_readgroups);
    *out = c;
    return 0;
}

void pd_ctx_destroy(pd_ctx * ctx) { delete ctx; }

// Mutation log and pair ids of the next segment (oracle_side_channel.h); consumed by the next runSegment().
struct PendingOrder
{
    bool present = false;
    std::vector<uint64_t> opOffsets;
    std::vector<pd_oracle_op> ops;
    std::vector<uint32_t> ids;
};
static PendingOrder g_order;

extern "C" void pd_oracle_next_segment_order(uint32_t n_readgroups, const uint64_t * rg_op_offsets, const pd_oracle_op * ops,
                                             const uint32_t * instance_ids, uint64_t n_instances)
{
    g_order.present = true;
    g_order.opOffsets.assign(rg_op_offsets, rg_op_offsets + n_readgroups + 1);
    g_order.ops.assign(ops, ops + rg_op_offsets[n_readgroups]);
    g_order.ids.assign(instance_ids, instance_ids + n_instances);
}

static int runSegment(pd_ctx * c, const pd_segment * seg, uint64_t * ticket)
{
    if (!c || !seg || seg->n_readgroups != c->rgs.size())
    {
        if (c)
            c->error = "pd_segment_submit: read-group count mismatch";
        return 1;
    }
    const unsigned S = c->nSamples();
    c->calls.clear();
    c->lastWindows = seg->n_windows;
    c->q3.assign(static_cast<size_t>(seg->n_windows) * S, 0);
    c->flags.assign(static_cast<size_t>(seg->n_windows) * S, 0);
    std::vector<uint64_t> next(c->rgs.size());
    for (size_t rg = 0; rg < c->rgs.size(); ++rg)
    {
        next[rg] = seg->rg_offsets[rg];
        c->active[rg].clear();
    }
    // Active lists are kept as instance indices so that pairs can be retired by their `off` window.
    g_snapShifts = getenv("POPDEL_SNAP_SHIFTS") != nullptr;
    g_perturbFreq = getenv("POPDEL_ORACLE_PERTURB_FREQ") ? atoi(getenv("POPDEL_ORACLE_PERTURB_FREQ")) : 0;
    std::vector<std::vector<uint64_t>> activeIdx(c->rgs.size());
    // the reference's visiting order, if the driver supplied the mutation log of this segment
    const bool refOrder = g_order.present;
    PendingOrder order;
    std::vector<std::unordered_map<uint32_t, uint64_t>> instanceOfId;
    std::vector<uint64_t> nextOp;
    if (refOrder)
    {
        std::swap(order, g_order);
        g_order = PendingOrder();
        if (order.opOffsets.size() != c->rgs.size() + 1 || order.ids.size() != seg->rg_offsets[seg->n_readgroups])
        {
            c->error = "pd_segment_submit: the mutation log does not belong to this segment";
            return 1;
        }
        c->refActive.resize(c->rgs.size());
        instanceOfId.resize(c->rgs.size());
        nextOp.assign(order.opOffsets.begin(), order.opOffsets.end() - 1);
        for (size_t rg = 0; rg < c->rgs.size(); ++rg)
            for (uint64_t i = seg->rg_offsets[rg]; i < seg->rg_offsets[rg + 1]; ++i)
                instanceOfId[rg][order.ids[i]] = i;       // a pair has one instance per segment
    }
    for (uint32_t w = 0; w < seg->n_windows; ++w)
    {
        for (size_t rg = 0; rg < c->rgs.size(); ++rg)
        {
            if (refOrder)
            {
                std::unordered_set<uint32_t> & set = c->refActive[rg];
                while (nextOp[rg] < order.opOffsets[rg + 1] && order.ops[nextOp[rg]].window <= w)
                {
                    const pd_oracle_op & op = order.ops[nextOp[rg]++];
                    if (op.add)
                        set.emplace(op.id);
                    else
                        set.erase(op.id);
                }
                std::vector<ActivePair> & act = c->active[rg];
                act.clear();
                for (uint32_t id : set)
                {
                    auto it = instanceOfId[rg].find(id);
                    if (it == instanceOfId[rg].end() || !(seg->pair_on[it->second] <= w && seg->pair_off[it->second] > w))
                    {
                        c->error = "pd_segment_submit: the replayed active set disagrees with the segment's instances";
                        return 1;
                    }
                    act.push_back(ActivePair{seg->pair_dev[it->second], seg->pair_start[it->second]});
                }
                size_t expect = 0;
                for (uint64_t i = seg->rg_offsets[rg]; i < seg->rg_offsets[rg + 1] && seg->pair_on[i] <= w; ++i)
                    expect += seg->pair_off[i] > w;
                if (expect != act.size())
                {
                    c->error = "pd_segment_submit: the replayed active set misses pairs of the segment";
                    return 1;
                }
                continue;
            }
            std::vector<uint64_t> & idx = activeIdx[rg];
            size_t keep = 0;
            for (size_t k = 0; k < idx.size(); ++k)
                if (seg->pair_off[idx[k]] > w)
                    idx[keep++] = idx[k];
            idx.resize(keep);
            while (next[rg] < seg->rg_offsets[rg + 1] && seg->pair_on[next[rg]] <= w)
            {
                if (seg->pair_off[next[rg]] > w)
                    idx.push_back(next[rg]);
                ++next[rg];
            }
            std::vector<ActivePair> & act = c->active[rg];
            act.resize(idx.size());
            for (size_t k = 0; k < idx.size(); ++k)
                act[k] = ActivePair{seg->pair_dev[idx[k]], seg->pair_start[idx[k]]};
        }
        c->currentPos = seg->first_window_pos + w * seg->window_shift;
        genotypeWindow(*c, &c->q3[static_cast<size_t>(w) * S], &c->flags[static_cast<size_t>(w) * S]);
    }
    *ticket = ++c->ticket;
    return 0;
}


int pd_segment_submit(pd_ctx * c, const pd_segment * seg, uint64_t * ticket) { return runSegment(c, seg, ticket); }

int pd_segment_stage(pd_ctx * c, const pd_segment * seg, uint64_t * handle)
{
    if (!c || !seg || !handle || seg->n_readgroups != c->rgs.size())
        return 1;
    pd_ctx::Staged st;
    const uint64_t total = seg->rg_offsets[seg->n_readgroups];
    st.offsets.assign(seg->rg_offsets, seg->rg_offsets + seg->n_readgroups + 1);
    st.dev.assign(seg->pair_dev, seg->pair_dev + total);
    st.start.assign(seg->pair_start, seg->pair_start + total);
    st.on.assign(seg->pair_on, seg->pair_on + total);
    st.off.assign(seg->pair_off, seg->pair_off + total);
    st.seg = *seg;
    *handle = c->nextHandle++;
    pd_ctx::Staged & dst = c->staged[*handle];
    dst = std::move(st);
    dst.seg.rg_offsets = dst.offsets.data();
    dst.seg.pair_dev = dst.dev.data();
    dst.seg.pair_start = dst.start.data();
    dst.seg.pair_on = dst.on.data();
    dst.seg.pair_off = dst.off.data();
    return 0;
}

int pd_segment_run(pd_ctx * c, uint64_t handle, uint64_t * ticket)
{
    if (!c || !c->staged.count(handle))
        return 1;
    return runSegment(c, &c->staged[handle].seg, ticket);
}

int pd_segment_unstage(pd_ctx * c, uint64_t handle) { return (c && c->staged.erase(handle)) ? 0 : 1; }

int pd_last_timing(pd_ctx * c, pd_timing * out)
{
    if (!c || !out)
        return 1;
    std::memset(out, 0, sizeof(*out));
    return 0;
}

int pd_segment_collect(pd_ctx * c, uint64_t ticket, pd_calls ** out)
{
    if (!c || !out || ticket != c->ticket)
    {
        if (c)
            c->error = "pd_segment_collect: unknown ticket";
        return 1;
    }
    const uint32_t n = static_cast<uint32_t>(c->calls.size());
    const uint32_t S = c->nSamples();
    pd_calls * pc = static_cast<pd_calls *>(std::calloc(1, sizeof(pd_calls)));
    pc->n_calls = n;
    pc->n_samples = S;
    std::vector<uint32_t> u(n);
    std::vector<double> d(n);
    std::vector<uint8_t> b(n);
#define PD_FIELD_U(dst, member) for (uint32_t i = 0; i < n; ++i) u[i] = c->calls[i].member; pc->dst = copyArray(u);
    PD_FIELD_U(initial_length, initialLength)
    PD_FIELD_U(iterations, iterations)
    PD_FIELD_U(deletion_length, deletionLength)
    PD_FIELD_U(window_position, windowPosition)
    PD_FIELD_U(position, position)
    PD_FIELD_U(end_position, endPosition)
    PD_FIELD_U(near_integer_shifts, nearIntegerShifts)
#undef PD_FIELD_U
    for (uint32_t i = 0; i < n; ++i) d[i] = c->calls[i].likelihoodRatio;
    pc->likelihood_ratio = copyArray(d);
    for (uint32_t i = 0; i < n; ++i) d[i] = c->calls[i].frequency;
    pc->frequency = copyArray(d);
    for (uint32_t i = 0; i < n; ++i) b[i] = c->calls[i].filter;
    pc->filter = copyArray(b);
    auto flatten = [&](size_t per, std::vector<uint32_t> OracleCall::*m) {
        std::vector<uint32_t> all(static_cast<size_t>(n) * S * per);
        for (uint32_t i = 0; i < n; ++i)
            std::memcpy(all.data() + static_cast<size_t>(i) * S * per, (c->calls[i].*m).data(), S * per * sizeof(uint32_t));
        return copyArray(all);
    };
    pc->pl = flatten(3, &OracleCall::pl);
    pc->lad = flatten(3, &OracleCall::lad);
    pc->dad = flatten(5, &OracleCall::dad);
    pc->first_last = flatten(2, &OracleCall::firstLast);
    *out = pc;
    return 0;
}

void pd_calls_free(pd_calls * pc)
{
    if (!pc)
        return;
    std::free(pc->initial_length);
    std::free(pc->iterations);
    std::free(pc->deletion_length);
    std::free(pc->likelihood_ratio);
    std::free(pc->frequency);
    std::free(pc->window_position);
    std::free(pc->position);
    std::free(pc->end_position);
    std::free(pc->filter);
    std::free(pc->near_integer_shifts);
    std::free(pc->pl);
    std::free(pc->lad);
    std::free(pc->dad);
    std::free(pc->first_last);
    std::free(pc);
}

int pd_stage1_last(pd_ctx * c, pd_stage1_view * out)
{
    if (!c || !out)
        return 1;
    out->n_windows = c->lastWindows;
    out->n_samples = c->nSamples();
    out->q3 = c->q3.data();
    out->flags = c->flags.data();
    return 0;
}

const char * pd_last_error(const pd_ctx * c) { return c ? c->error.c_str() : g_createError.c_str(); }

// Same partition rule as the product library; the oracle itself always sees the whole cohort (tests compare the
// sharded product run with it).
uint32_t pd_shard_samples(uint32_t n_samples_global, uint32_t world)
{
    if (world == 0)
        return 0;
    const uint32_t tiles = (n_samples_global + 255u) / 256u;
    return (tiles + world - 1) / world * 256u;
}

int pd_ctx_set_shard(pd_ctx * c, const pd_shard * sh)
{
    if (!c || !sh)
        return 1;
    if (sh->world != 1 || sh->rank != 0)
    {
        c->error = "pd_ctx_set_shard: the oracle runs the whole cohort in one context";
        return 1;
    }
    return 0;
}

int pd_local_group_create(uint32_t, pd_local_group ** out)
{
    if (out)
        *out = nullptr;
    return 1;                                     // the oracle runs the whole cohort in one context
}

void pd_local_group_destroy(pd_local_group *) {}

int pd_ctx_set_shard_local(pd_ctx * c, const pd_shard *, pd_local_group *)
{
    if (c)
        c->error = "pd_ctx_set_shard_local: the oracle runs the whole cohort in one context";
    return 1;
}

int pd_log_table_view(pd_ctx * c, uint32_t rg, double * value, double * logValue, double * log10Value, uint32_t * n)
{
    if (!c || rg >= c->rgs.size() || !value || !logValue || !log10Value || !n)
        return 1;
    const ReadGroup & h = c->rgs[rg];
    for (size_t i = 0; i < h.values.size(); ++i)
    {
        value[i] = h.values[i];
        logValue[i] = static_cast<double>(logl(h.values[i]));          // what the reference computes per pair, rounded to double
        log10Value[i] = static_cast<double>(log10l(h.values[i]));
    }
    *n = static_cast<uint32_t>(h.values.size());
    return 0;
}

// per-sample merging on the device: not offered by the oracle (the host driver then merges on the host)
int pd_ctx_keep_results_on_device(pd_ctx *, int) { return 1; }
int pd_calls_merge(pd_ctx *, const pd_calls *, uint32_t, const uint32_t *, const uint32_t *, const uint32_t *, pd_merged ** out)
{
    if (out)
        *out = nullptr;
    return 1;
}
int pd_calls_fetch_columns(pd_ctx *, pd_calls *) { return 1; }
void pd_merged_free(pd_merged *) {}

int pd_rccl_unique_id(uint8_t *) { return 1; }

int pd_device_warmup(int) { return 0; }

// The oracle takes segments as pair instances (pd_segment): the host walks the windows itself. With POPDEL_ORACLE_WALK=1 it
// also accepts the loader's tables (pd_walk) and derives the instances with the host's closed-form evaluation of the window
// loop (pdh::WalkReference, the host reference of walk.hip) -- a TEST path for the hand-over protocol of the call driver
// (planned segments, the hand-over thread, leaving and re-entering the mode) on machines without a GPU and under the
// sanitizers; the checker of the closed form itself is the walked loop (POPDEL_WALK_CHECK, tests/test_walk_closed_form.py).
// (device-only entry points of ABI 11: the oracle declines)
int pd_inflate_members(int, const uint8_t *, const uint64_t *, uint32_t, uint8_t *, const uint64_t *, uint32_t *, uint64_t *, uint32_t,
                       pd_inflate_timing *) { return 1; }
const char * pd_inflate_last_error(void) { return "the oracle backend has no device"; }
int pd_ctx_stage1_view(pd_ctx * c, int) { return c ? 0 : 1; }
int pd_last_stage_timing(pd_ctx *, pd_stage_timing * out)
{
    if (out)
        std::memset(out, 0, sizeof(*out));
    return out ? 0 : 1;
}
double pd_host_copy_gbs(uint64_t) { return 0; }
int pd_host_regions_offer(const void * const *, const uint64_t *, uint32_t) { return 0; }
int pd_walk_entry_counts(pd_ctx *, uint64_t * g, uint64_t * s, uint64_t * l)
{
    if (g) *g = 0;
    if (s) *s = 0;
    if (l) *l = 0;
    return 0;
}
pd_inflater * pd_inflater_create(int) { return nullptr; }
const char * pd_inflater_last_error(pd_inflater *) { return "the oracle backend has no device"; }
void pd_inflater_destroy(pd_inflater *) {}
uint8_t * pd_inflater_input(pd_inflater *, uint64_t) { return nullptr; }
int pd_inflater_run(pd_inflater *, const uint64_t *, uint32_t, const uint64_t *, const uint8_t **, const uint32_t **, const uint64_t **, double *) { return 1; }
int pd_ctx_set_shard_lanes(pd_ctx *, const pd_shard *, const pd_shard *, uint32_t) { return 1; }
int pd_inflater_run_parsed(pd_inflater *, const uint64_t *, uint32_t, const uint64_t *, const uint32_t *, const uint8_t *, const uint8_t **, const uint32_t **,
                           const uint64_t **, pd_parsed_blocks *, double *) { return 1; }
uint64_t pd_inflater_run_id(pd_inflater *) { return 0; }
int pd_inflater_run_summarized(pd_inflater *, const uint64_t *, uint32_t, const uint64_t *, const uint32_t *, const uint8_t *, const pd_summarize_options *,
                               const uint8_t **, const uint32_t **, const uint64_t **, pd_parsed_blocks *, pd_block_summaries *, double *) { return 1; }

int pd_walk_source_counts(pd_ctx *, uint64_t * from_device)
{
    if (from_device)
        *from_device = 0;
    return 0;
}
int pd_walk_supported(const pd_ctx *) { return getenv("POPDEL_ORACLE_WALK") ? 1 : 0; }
int pd_walk_submit(pd_ctx * c, const pd_walk * wk, uint64_t * ticket)
{
    if (!c || !wk || !ticket || !getenv("POPDEL_ORACLE_WALK"))
    {
        if (c)
            c->error = "pd_walk_submit: the oracle takes pair instances (pd_segment_submit)";
        return 1;
    }
    if (wk->n_readgroups != c->rgs.size())
    {
        c->error = "pd_walk_submit: read-group count mismatch";
        return 1;
    }
    const uint32_t nRg = wk->n_readgroups;
    if (!c->walk)
        c->walk.reset(new pdh::WalkReference(nRg));
    pdh::WalkShape shape;
    shape.firstPos = wk->first_window_pos;
    shape.stepPos = wk->step_pos;
    shape.nWindows = wk->n_windows;
    shape.hasStep = wk->has_step != 0;
    std::vector<uint64_t> offsets(nRg + 1, 0);
    std::vector<int32_t> dev;
    std::vector<uint32_t> start, on, off;
    std::vector<pdh::PairInstance> inst;
    for (uint32_t r = 0; r < nRg; ++r)
    {
        pdh::StartSlots::Slot slot;
        if (wk->switches[r] != 0)
        {
            slot.entries.resize(wk->rg_counts[r]);
            for (uint32_t i = 0; i < wk->rg_counts[r]; ++i)
                slot.entries[i] = pdh::StartEntry{wk->rg_entries[r][i].start, wk->rg_entries[r][i].dev};
            slot.endBorder = wk->end_border[r];
            slot.idOffset = wk->id_offset[r];
        }
        inst.clear();
        c->walk->segment(r, shape, wk->window_shift, wk->switches[r], wk->equal_run[r] != 0, wk->switches[r] ? &slot : nullptr,
                         c->rgs[r].median, c->rgs[r].readLength, inst);
        for (const pdh::PairInstance & p : inst)
        {
            dev.push_back(p.dev);
            start.push_back(p.start);
            on.push_back(p.on);
            off.push_back(p.off);
        }
        offsets[r + 1] = dev.size();
    }
    pd_segment seg;
    std::memset(&seg, 0, sizeof(seg));
    seg.first_window_pos = wk->first_window_pos;
    seg.n_windows = wk->n_windows;
    seg.window_shift = wk->window_shift;
    seg.n_readgroups = nRg;
    seg.rg_offsets = offsets.data();
    seg.pair_dev = dev.data();
    seg.pair_start = start.data();
    seg.pair_on = on.data();
    seg.pair_off = off.data();
    return runSegment(c, &seg, ticket);
}
int pd_walk_active_ex(pd_ctx * c, uint64_t * n, uint32_t * rg, uint32_t * start, int32_t * dev, uint32_t * id, uint32_t * last, uint64_t capacity);
int pd_walk_active(pd_ctx * c, uint64_t * n, uint32_t * rg, uint32_t * start, int32_t * dev, uint32_t * id, uint64_t capacity)
{
    return pd_walk_active_ex(c, n, rg, start, dev, id, nullptr, capacity);
}
int pd_walk_active_ex(pd_ctx * c, uint64_t * n, uint32_t * rg, uint32_t * start, int32_t * dev, uint32_t * id, uint32_t * last, uint64_t capacity)
{
    if (!c || !n)
        return 1;
    uint64_t total = 0;
    if (c->walk)
        for (unsigned r = 0; r < c->rgs.size(); ++r)
            for (const pdh::WalkReference::Pair & p : c->walk->active(r))
            {
                if (rg && total < capacity)
                {
                    rg[total] = r;
                    start[total] = p.start;
                    dev[total] = p.dev;
                    id[total] = p.id;
                    if (last)
                        last[total] = p.lastWin | (p.next ? 0x80000000u : 0u);
                }
                ++total;
            }
    *n = total;
    return (rg && total > capacity) ? 2 : 0;
}

int pd_ctx_set_shard_rccl(pd_ctx * c, const pd_shard *, const uint8_t *)
{
    if (c)
        c->error = "pd_ctx_set_shard_rccl: the oracle runs the whole cohort in one context";
    return 1;
}

const char * pd_backend_name(void) { return "oracle-cpu-longdouble"; }

int pd_abi_version(void) { return PD_ABI_VERSION; }

}  // extern "C"
