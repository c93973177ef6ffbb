// Host driver of `popdel call`: segment loop, profile loading, hand-off to the genotyping backend through the
// C ABI of include/popdel_hip.h, call merging and VCF output.
// Behavioural reference: workflow_popdel.h:28-54, :58-86 (processSegment), :92-136 (ROI bookkeeping),
// :256-371 (popdel_call); popdel_call/load_profile_popdel_call.h:220-667 (loader).
#ifndef PDH_CALL_DRIVER_H_
#define PDH_CALL_DRIVER_H_

#include <functional>
#include <vector>

#include "call_params.h"
#include "chrom_profile.h"
#include "window_call.h"

namespace pdh {

// Optional observer used by tests: called once per segment with the window event log.
struct SegmentObserver
{
    std::function<void(unsigned firstWindowPos, unsigned nWindows, const std::vector<std::vector<PairInstance>> &)> onSegment;
    // called with the raw window calls of a segment, before merging
    std::function<void(const std::vector<WindowCall> &)> onWindowCalls;
    // if set: called once per segment, before it is submitted, with the mutations of the active sets since the previous
    // segment in the reference's order (ActiveOp) and the instance log (whose entries carry the reference's pair ids)
    std::function<void(const std::vector<std::vector<ActiveOp>> &, const std::vector<std::vector<PairInstance>> &)> onActiveOps;
};

// Runs the whole `popdel call` command. Returns the process exit code.
// quickExit: end the process right after the output is complete (command-line use), skipping the teardown.
int runPopdelCall(int argc, const char ** argv, SegmentObserver * observer = nullptr, bool quickExit = false);

}  // namespace pdh
#endif
