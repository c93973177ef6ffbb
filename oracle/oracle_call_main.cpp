// ORACLE -- TEST INFRASTRUCTURE ONLY.
// `popdel call` host driver linked against the CPU oracle instead of the HIP library. Besides writing the VCF it can
// dump (a) every window call at full precision and (b) per-window active-set checksums, in the formats of the
// survey-stage captures under tests/golden/, so that the restatement can be compared with the reference's outputs.
//   POPDEL_ORACLE_ACTIVE_DUMP=<file>  per window: pos \t count:sum(startPos) per read group
//   POPDEL_ORACLE_CALL_DUMP=<file>    per window call: scalars at 17 significant digits + per-sample integers
//   POPDEL_ORACLE_SET_ORDER=1         visit the pairs of a window in the reference's std::unordered_set order
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "call_driver.h"
#include "oracle_side_channel.h"

int main(int argc, const char ** argv)
{
    pdh::SegmentObserver obs;
    FILE * act = nullptr;
    if (const char * path = getenv("POPDEL_ORACLE_ACTIVE_DUMP"))
        act = fopen(path, "w");
    if (act)
    {
        obs.onSegment = [act](unsigned firstPos, unsigned nWindows, const std::vector<std::vector<pdh::PairInstance>> & log) {
            std::vector<std::vector<long long>> cnt(log.size(), std::vector<long long>(nWindows + 1, 0));
            std::vector<std::vector<long long>> sum(log.size(), std::vector<long long>(nWindows + 1, 0));
            for (size_t rg = 0; rg < log.size(); ++rg)
                for (const pdh::PairInstance & p : log[rg])
                {
                    cnt[rg][p.on] += 1;
                    cnt[rg][p.off] -= 1;
                    sum[rg][p.on] += p.start;
                    sum[rg][p.off] -= p.start;
                }
            std::vector<long long> c(log.size(), 0), s(log.size(), 0);
            for (unsigned w = 0; w < nWindows; ++w)
            {
                fprintf(act, "%u", firstPos + 30 * w);
                for (size_t rg = 0; rg < log.size(); ++rg)
                {
                    c[rg] += cnt[rg][w];
                    s[rg] += sum[rg][w];
                    fprintf(act, "\t%lld:%lld", c[rg], s[rg]);
                }
                fputc('\n', act);
            }
        };
    }
    FILE * cd = nullptr;
    if (const char * path = getenv("POPDEL_ORACLE_CALL_DUMP"))
        cd = fopen(path, "w");
    if (cd)
    {
        obs.onWindowCalls = [cd](const std::vector<pdh::WindowCall> & calls) {
            for (const pdh::WindowCall & c : calls)
            {
                fprintf(cd, "CALL win=%u pos=%u end=%u init=%u it=%u len=%u lr=%.17g af=%.17g filt=%d", c.windowPosition,
                        c.position, c.endPosition, c.initialLength, c.iterations, c.deletionLength, c.likelihoodRatio,
                        c.frequency, static_cast<int>(c.filter));
                for (size_t s = 0; s < c.pl.size(); ++s)
                    fprintf(cd, " | %u,%u,%u;%u,%u,%u;%u,%u,%u,%u,%u;%u,%u", c.pl[s][0], c.pl[s][1], c.pl[s][2], c.lad[s][0],
                            c.lad[s][1], c.lad[s][2], c.dad[s][0], c.dad[s][1], c.dad[s][2], c.dad[s][3], c.dad[s][4],
                            c.firstLast[s][0], c.firstLast[s][1]);
                fputc('\n', cd);
            }
        };
    }
    const bool setOrder = getenv("POPDEL_ORACLE_SET_ORDER") != nullptr;
    if (setOrder)
    {
        obs.onActiveOps = [](const std::vector<std::vector<pdh::ActiveOp>> & ops, const std::vector<std::vector<pdh::PairInstance>> & log) {
            std::vector<uint64_t> offsets(ops.size() + 1, 0);
            std::vector<pd_oracle_op> flat;
            for (size_t rg = 0; rg < ops.size(); ++rg)
            {
                for (const pdh::ActiveOp & o : ops[rg])
                    flat.push_back(pd_oracle_op{o.window, o.id, o.add});
                offsets[rg + 1] = flat.size();
            }
            std::vector<uint32_t> ids;
            for (const auto & v : log)
                for (const pdh::PairInstance & p : v)
                    ids.push_back(p.id);
            pd_oracle_next_segment_order(static_cast<uint32_t>(ops.size()), offsets.data(), flat.data(), ids.data(), ids.size());
        };
    }
    int rc = pdh::runPopdelCall(argc, argv, (act || cd || setOrder) ? &obs : nullptr);
    if (act)
        fclose(act);
    if (cd)
        fclose(cd);
    return rc;
}
