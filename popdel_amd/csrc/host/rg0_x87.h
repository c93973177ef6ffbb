// Read group 0's weights of the reference-shift update, evaluated on the host FPU.
//
// The reference takes the weights exp(a0Rg), exp(a1Rg) of EVERY read group from the data-likelihood triple of read
// group 0 (genotype_deletion_popdel_call.h:401, 424-431: the iterator is never advanced) and evaluates them in x87
// `long double` through logl/expl. Whether a reference shift int(sum(p * dev) / sum(p)) lands on an integer or one below
// is decided by the last bit of those two numbers wherever the exact quotient IS an integer (single-pair read groups,
// equally weighted pairs), and the x87 transcendental instructions behind logl/expl cannot be reproduced on the GPU.
// They are two numbers per (window, candidate) cell and EM iteration, derived from the handful of pairs of ONE read
// group, so the library works them out here, with the reference's own operations in the reference's order, and hands them to the
// kernels as 64-bit mantissa + exponent (pd::X87); everything per sample and per pair stays on the device.
#ifndef PDH_RG0_X87_H_
#define PDH_RG0_X87_H_

#include <cstdint>
#include <vector>

#include "popdel_hip.h"
#include "x87soft.h"

namespace pdh {

using pd::Rg0Query;
using pd::Rg0Weights;

class Rg0Service
{
public:
    void setReadGroup(const pd_readgroup & rg);
    // pair instances of read group 0 in one segment (arrays are copied)
    struct Segment
    {
        std::vector<int32_t>  dev;
        std::vector<uint32_t> on, off;
        uint32_t maxLife = 0;
    };
    static void copySegment(Segment & out, const pd_segment & seg);
    // one batch of cells: `slot` i keeps what was computed for cell i last time (same length and shift: same triple)
    void beginBatch(unsigned nCells);
    void weights(const Segment & seg, const Rg0Query * q, unsigned n, Rg0Weights * out);

private:
    struct Cached
    {
        bool        valid = false;
        uint32_t    len = 0;
        int32_t     shift0 = 0;
        long double t[3] = {0, 0, 0};
    };
    double lookup(int deviation) const;
    void triple(const Segment & seg, const Rg0Query & q, long double * t) const;
    uint32_t median_ = 0, maxLoad_ = 0;
    int32_t offset_ = 0;
    double minProb_ = 0;
    std::vector<double> values_;
    std::vector<Cached> cache_;
};

}  // namespace pdh
#endif
