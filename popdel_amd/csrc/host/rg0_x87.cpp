#include "rg0_x87.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "parallel.h"

namespace pdh {

namespace {

pd::X87 toX87(long double v)
{
    // 80-bit extended format: 64-bit mantissa with explicit integer bit, then sign + 15-bit exponent
    unsigned char raw[16] = {0};
    std::memcpy(raw, &v, 10);
    uint64_t m;
    uint16_t se;
    std::memcpy(&m, raw, 8);
    std::memcpy(&se, raw + 8, 2);
    if (m == 0)
        return pd::x87Zero();
    pd::X87 r;
    r.pad = 0;
    const int ex = se & 0x7fff;
    if (ex == 0)
    {
        const int lz = __builtin_clzll(m);
        r.m = m << lz;
        r.e = -16382 - lz;
    }
    else
    {
        r.m = m;
        r.e = ex - 16383;
    }
    return r;
}

}  // namespace

void Rg0Service::setReadGroup(const pd_readgroup & rg)
{
    median_ = rg.median;
    maxLoad_ = rg.max_load;
    offset_ = rg.hist_offset;
    minProb_ = rg.min_prob;
    values_.assign(rg.hist_values, rg.hist_values + rg.hist_len);
}

void Rg0Service::copySegment(Segment & out, const pd_segment & seg)
{
    const uint64_t lo = seg.rg_offsets[0], hi = seg.rg_offsets[1];
    out.dev.assign(seg.pair_dev + lo, seg.pair_dev + hi);
    out.on.assign(seg.pair_on + lo, seg.pair_on + hi);
    out.off.assign(seg.pair_off + lo, seg.pair_off + hi);
    out.maxLife = 0;
    for (size_t i = 0; i < out.on.size(); ++i)
        if (out.off[i] > out.on[i])
            out.maxLife = std::max(out.maxLife, out.off[i] - out.on[i]);
}

void Rg0Service::beginBatch(unsigned nCells)
{
    cache_.assign(nCells, Cached());
}

// insert_histogram_popdel.h:1157-1163
double Rg0Service::lookup(int deviation) const
{
    const int i = deviation + static_cast<int>(median_) - offset_;
    if (i <= 0 || i + 1 >= static_cast<int>(values_.size()))
        return minProb_;
    return values_[static_cast<size_t>(i)];
}

// genotype_deletion_popdel_call.h:186-236 for read group 0: the per-read-group triple exp(g - max(g)), the pairs visited
// in the order of the segment's instance list (pd_segment: non-decreasing `on`)
void Rg0Service::triple(const Segment & seg, const Rg0Query & q, long double * t) const
{
    const uint32_t w = static_cast<uint32_t>(q.window);
    // active: on <= w < off. Instances are ordered by `on`; none lives longer than maxLife windows.
    const size_t hi = std::upper_bound(seg.on.begin(), seg.on.end(), w) - seg.on.begin();
    const uint32_t firstOn = w + 1 > seg.maxLife ? w + 1 - seg.maxLife : 0;
    size_t lo = std::lower_bound(seg.on.begin(), seg.on.begin() + hi, firstOn) - seg.on.begin();
    size_t n = 0;
    for (size_t i = lo; i < hi; ++i)
        n += seg.off[i] > w;
    long double c1 = 0, c2 = 0, c3 = 0;
    if (n < maxLoad_)                                   // at or above maxLoad the reference skips the read group: zeros
    {
        for (size_t i = lo; i < hi; ++i)
        {
            if (seg.off[i] <= w)
                continue;
            const long double refLikelihood = lookup(seg.dev[i] - q.shift0);
            const long double delLikelihood = lookup(seg.dev[i] - static_cast<int>(q.len));
            const long double g0 = logl(refLikelihood);
            const long double g1 = logl(refLikelihood + delLikelihood) - std::log(2.0);
            const long double g2 = logl(delLikelihood);
            c1 += g0;
            c2 += g1;
            c3 += g2;
        }
        const long double mx = std::max(std::max(c1, c2), c3);
        t[0] = expl(c1 - mx);
        t[1] = expl(c2 - mx);
        t[2] = expl(c3 - mx);
    }
    else
    {
        t[0] = t[1] = t[2] = 0;
    }
}

// genotype_deletion_popdel_call.h:421-431
void Rg0Service::weights(const Segment & seg, const Rg0Query * q, unsigned n, Rg0Weights * out)
{
    parallelFor(n, [&](size_t i) {
        Rg0Weights & o = out[i];
        std::memset(&o, 0, sizeof(o));
        if (!q[i].active)
            return;
        Cached & c = cache_[i];
        if (!c.valid || c.len != q[i].len || c.shift0 != q[i].shift0)
        {
            triple(seg, q[i], c.t);
            c.valid = true;
            c.len = q[i].len;
            c.shift0 = q[i].shift0;
        }
        const double g1 = q[i].gt[0], g2 = q[i].gt[1], g3 = q[i].gt[2];
        const double aSumRg = c.t[0] * g1 + c.t[1] * g2 + c.t[2] * g3;
        const long double l = logl(c.t[0]);
        const long double g = std::log(g1);
        const long double a = std::log(aSumRg);
        const long double a0Rg = l + g - a;
        const long double a1Rg = logl(c.t[1]) + std::log(g2) - std::log(aSumRg);
        const long double e0 = expl(a0Rg), e1 = expl(a1Rg);
        if (!std::isfinite(e0) || !std::isfinite(e1) || e0 < 0 || e1 < 0)
        {
            o.invalid = 1;
            return;
        }
        o.expA0 = toX87(e0);
        o.expA1 = toX87(e1);
        o.expA0d = static_cast<double>(e0);
        o.expA1d = static_cast<double>(e1);
    }, 512);     // a segment has a few dozen cells as a rule: no threads then
}

}  // namespace pdh
