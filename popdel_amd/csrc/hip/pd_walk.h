// Interface between walk.hip (the window loop in closed form, on the device) and capi.hip.
#ifndef PD_WALK_H_
#define PD_WALK_H_

#include "pd_device.h"

struct pd_inflater;

namespace pd {

// Shape of a segment (pdh::WalkShape): position of window 0, position of the loader's extra step ahead of it, windows.
struct WalkParams
{
    uint32_t firstPos, stepPos, hasStep, nWindows, shift;
};

// The loader's tables for one segment, in device memory (see pd_walk in include/popdel_hip.h).
struct WalkInput
{
    const uint64_t * rgOffsets;     // [nRg + 1] new start-slot entries per read group
    const uint2 *    entries;       // {start, deviation (bits of the int32)}
    const uint32_t * endBorder;     // [nRg]
    const uint32_t * idOffset;      // [nRg]
    const uint8_t *  switches;      // [nRg] 0, 1, 2 (= two or more)
    const uint8_t *  equalRun;      // [nRg]
};

// What goes from segment to segment, per read group: the pairs still active (aLast: last window, bit 31 = the end record
// sits in the slot after the end read slot) and the start entries of the read slot that no window has reached.
struct WalkState
{
    uint32_t * aStart;
    int32_t *  aDev;
    uint32_t * aId;
    uint32_t * aLast;
    uint32_t * nActive;     // [nRg]
    uint32_t * pStart;
    int32_t *  pDev;
    uint32_t * pId;
    uint32_t * pLast;
    uint32_t * nPending;    // [nRg]
    uint32_t   capActive, capPending;
};

// instances per read group -> instCount, their exclusive scan -> instOffsets[0 .. nRg], sum -> total[0]; need[0] / need[1]
// (cleared by the caller): the longest active / pending list the state after the segment will hold
void launchWalkCount(const WalkInput & in, const WalkState & st, const WalkParams & w, const RgConst * rgConst, unsigned nRg,
                     uint32_t * instCount, uint64_t * instOffsets, uint64_t * total, unsigned * need, hipStream_t stream);
// the instances (wide arrays of pd::InstanceArrays) and the state after the segment (nxt; must not alias st);
// problems: bit 0 = an active list overflowed, bit 1 = a pending list; activeTotal += pairs left active
void launchWalkEmit(const WalkInput & in, const WalkState & st, const WalkParams & w, const RgConst * rgConst, unsigned nRg,
                    const uint64_t * instOffsets, int32_t * outDev, uint32_t * outOn, uint32_t * outOff, uint32_t * outStart,
                    const WalkState & nxt, unsigned * problems, unsigned long long * activeTotal, hipStream_t stream);

// Entries fetched by the device from the host's tables themselves: src[rg] is the device-visible address of the read group's
// entries in page-locked host memory (0: none / taken care of by a staging copy), rgOffsets as in WalkInput; one workgroup
// walks a read group's entries, 16 bytes per lane and trip (the link wants many requests in flight: nRg workgroups of 256).
void launchGatherHostEntries(const uint64_t * src, const uint64_t * rgOffsets, uint2 * entries, unsigned nRg, hipStream_t stream);
// ... and from device memory (the records an inflater parsed, pd_walk_source): the same copy with a workgroup per read group
void launchGatherDeviceEntries(const uint64_t * src, const uint64_t * rgOffsets, uint2 * entries, unsigned nRg, hipStream_t stream);
// (inflate.hip) the records array the inflater's last parsed chunk left on the device: base, count, device
const uint2 * inflaterDeviceRecords(::pd_inflater * inflater, uint64_t & count, int & device, uint64_t & runId);

}  // namespace pd
#endif
