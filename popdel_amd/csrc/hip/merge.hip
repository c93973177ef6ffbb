// Per-sample part of call merging on the device: what the reference's setGenotypes() (utils_popdel.h:446-501) and
// setFreqFromGTs() (:339-359) do for every merged variant -- per sample the mean of the members' phred-scaled likelihoods
// (rounded, with the reference's tie bumps), per-component upper medians of LAD / DAD over the member windows, the
// allele count behind the variant's AF -- over the per-sample result arrays of a collected segment while they are still
// in HBM. The scalar grouping of the window calls (which calls form a variant: utils_popdel.h:567-654, leaking
// accumulators and all) stays with the host; it only needs the call-level fields. Only the merged columns cross PCIe
// (26 MB instead of 474 MB on the candidate-dense bench segment).
//
// One lane per sample (the arrays are [call][sample][k]: lanes of a wave read neighbouring samples, coalesced), a
// workgroup of 64 samples per merged group. Medians by counting: the values are pair counts of one sample in one window
// (< maxLoad per read group), so each lane keeps a 128-bin histogram per component in LDS (8 components x 128 bins x 64
// lanes x 2 B = 128 KB) and walks it; a value beyond the bins raises a flag and the caller merges that segment on the host.
#include "pd_build.h"

namespace pd {

constexpr int kMergeBins = 128;
constexpr int kMergeLanes = 64;

__global__ __launch_bounds__(kMergeLanes) void mergeGroupsKernel(MergeIn in, MergeOut out)
{
    extern __shared__ unsigned short hist[];                 // [8][kMergeBins][kMergeLanes]
    const unsigned g = blockIdx.y, lane = threadIdx.x, S = in.nSamples;
    const unsigned s = blockIdx.x * kMergeLanes + lane;
    for (unsigned i = lane; i < 8u * kMergeBins * kMergeLanes; i += kMergeLanes)
        hist[i] = 0;
    __syncthreads();
    const unsigned m0 = in.memberOffsets[g], m1 = in.memberOffsets[g + 1], count = m1 - m0;
    unsigned sum[3] = {0, 0, 0};
    bool over = false;
    if (s < S)
    {
        for (unsigned m = m0; m < m1; ++m)
        {
            const size_t row = (size_t)in.memberIndex[m] * S + s;
            const uint32_t * p = in.pl + row * 3;
            sum[0] += p[0];
            sum[1] += p[1];
            sum[2] += p[2];
            const uint32_t * l = in.lad + row * 3;
            const uint32_t * d = in.dad + row * 5;
#pragma unroll
            for (int k = 0; k < 8; ++k)
            {
                const uint32_t v = k < 3 ? l[k] : d[k - 3];
                over = over || v >= (uint32_t)kMergeBins;
                ++hist[((unsigned)k * kMergeBins + min(v, (uint32_t)kMergeBins - 1u)) * kMergeLanes + lane];
            }
        }
    }
    unsigned add = 0;
    if (s < S && count != 0)
    {
        // reference :485-498: PL = round((sum - min) / count), ties between neighbouring genotypes bumped
        const double minGt = (double)min(min(sum[0], sum[1]), sum[2]);
        const double ref = ((double)sum[0] - minGt) / count, het = ((double)sum[1] - minGt) / count, hom = ((double)sum[2] - minGt) / count;
        uint32_t pl[3] = {(uint32_t)round(ref), (uint32_t)round(het), (uint32_t)round(hom)};
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
        const size_t o = (size_t)g * S + s;
        out.pl[o * 3] = pl[0];
        out.pl[o * 3 + 1] = pl[1];
        out.pl[o * 3 + 2] = pl[2];
        // upper medians (sorted[count / 2], reference :394-421)
        const unsigned want = count / 2u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            unsigned acc = 0, v = 0;
            for (; v < (unsigned)kMergeBins; ++v)
            {
                acc += hist[((unsigned)k * kMergeBins + v) * kMergeLanes + lane];
                if (acc > want)
                    break;
            }
            if (k < 3)
                out.lad[o * 3 + k] = v;
            else
                out.dad[o * 5 + (k - 3)] = v;
        }
        const size_t h = (size_t)in.headIndex[g] * S + s;
        out.fl[o * 2] = in.fl[h * 2];
        out.fl[o * 2 + 1] = in.fl[h * 2 + 1];
        // reference :339-359: PL(ref) == 0 -> no allele (this includes samples without data), PL(het) == 0 -> one, else two
        add = pl[0] == 0u ? 0u : (pl[1] == 0u ? 1u : 2u);
    }
    // one atomic per wave
    for (int o = 32; o > 0; o >>= 1)
        add += (unsigned)__shfl_xor((int)add, o, 64);
    if (lane == 0 && add)
        atomicAdd(&out.alleles[g], add);
    if (over)
        *out.overflow = 1u;                     // every writer stores the same value
}

bool launchMergeGroups(const MergeIn & in, const MergeOut & out, unsigned nGroups, hipStream_t st)
{
    const size_t lds = (size_t)8 * kMergeBins * kMergeLanes * sizeof(unsigned short);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(mergeGroupsKernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(mergeGroupsKernel, dim3((in.nSamples + kMergeLanes - 1) / kMergeLanes, nGroups), dim3(kMergeLanes), lds, st, in, out);
    return true;
}

}  // namespace pd
