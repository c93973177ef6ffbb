// Device-side construction of the staged segment layout: interface between build.hip (kernels) and capi.hip (staging).
#ifndef PD_BUILD_H_
#define PD_BUILD_H_

#include "pd_device.h"

namespace pd {

// Pair instances in device memory as uploaded: narrow (16-bit fields; start as distance back from the position of the
// instance's first window) or wide (the caller's arrays as they are).
struct InstanceArrays
{
    const int16_t *  dev16;
    const uint16_t * on16;
    const uint16_t * off16;
    const uint16_t * back16;      // firstPos + on * shift - start
    const int32_t *  dev32;
    const uint32_t * on32;
    const uint32_t * off32;
    const uint32_t * start32;
    int              wide;
    uint32_t         firstPos, shift;
};

// Entities: the real read groups, then the scan read groups of the samples that have several read groups (they cover the
// instances of all read groups of their sample, read group after read group).
struct BuildEntities
{
    const uint64_t * rgOffsets;    // [nRg + 1] instance offsets of the real read groups
    const uint32_t * sampleRg;     // [nSamples + 1]
    const uint32_t * scanSample;   // [nScan] sample of a scan read group
    uint32_t         nRg, nScan;
};

struct CountOut
{
    uint32_t * winStart;     // [nEnt][W + K]
    uint32_t * endCount;     // [nEnt][W + 1]
    uint32_t * entChunks;    // [nEnt]
    uint32_t * entMax;       // [nEnt] chunks of the entity's densest 512-window block
    uint8_t *  rgPacked;     // [nEnt]
    unsigned * problems;     // [0]: instances out of `on` order
};

struct ScatterOut
{
    PairDO *         pairs;       // [real chunks]   stage-2 arrays, or null
    uint32_t *       start;       // [real chunks]
    uint32_t *       pairsPk;     // [all chunks]    stage-1 image (dev16 biased | end window << 16), or null
    const uint64_t * rgBase;
    const uint32_t * winStart;
};

// Compact upload (5 bytes per instance): the deviation in 16 bits; `on` as the difference to the previous instance of the
// read group, the life off - on and the distance of the pair start behind its first window in one byte each, 255 = look
// the instance up in the exception list (ascending instance index; excRgStart[rg] = first exception of read group rg).
// expandKernel turns them into the on16 / off16 / back16 arrays of the narrow format in device memory.
struct CompactInstances
{
    const uint8_t *  onDelta;      // [nInst]
    const uint8_t *  life;         // [nInst]
    const uint8_t *  back;         // [nInst]
    const uint32_t * excRgStart;   // [nRg + 1]
    const uint32_t * excIndex;     // [nExc] instance index, ascending
    const uint32_t * excOnDelta;   // [nExc]
    const uint32_t * excLife;      // [nExc]
    const uint32_t * excBack;      // [nExc]
};
void launchExpand(const CompactInstances & in, const uint64_t * rgOffsets, unsigned nRg, unsigned W, uint16_t * on16, uint16_t * off16,
                  uint16_t * back16, hipStream_t st);
void launchLifeStats(const InstanceArrays & in, uint64_t n, unsigned W, unsigned long long * hist, hipStream_t st);
size_t buildLdsBytes(unsigned W, unsigned K);
bool launchCount(const InstanceArrays & in, const BuildEntities & en, unsigned W, unsigned K, const CountOut & out, hipStream_t st);
void launchBaseScan(const uint32_t * entChunks, unsigned nEnt, unsigned nRg, uint64_t * rgBase, uint64_t * totals, hipStream_t st);
bool launchScatter(const InstanceArrays & in, const BuildEntities & en, unsigned W, unsigned K, const ScatterOut & out, hipStream_t st);
void launchScanGeneric(const BuildEntities & en, const RgConst * rgConst, const uint32_t * winStart, const uint32_t * endCount, unsigned W,
                       unsigned K, uint8_t * scanGeneric, unsigned blocksPerSample, hipStream_t st);

// merge.hip: per-sample part of call merging on the device
struct MergeIn
{
    const uint32_t * pl;
    const uint32_t * lad;
    const uint32_t * dad;
    const uint32_t * fl;
    const uint32_t * memberOffsets;
    const uint32_t * memberIndex;
    const uint32_t * headIndex;
    unsigned         nSamples;
};
struct MergeOut
{
    uint32_t * pl;
    uint32_t * lad;
    uint32_t * dad;
    uint32_t * fl;
    unsigned * alleles;
    unsigned * overflow;
};
bool launchMergeGroups(const MergeIn & in, const MergeOut & out, unsigned nGroups, hipStream_t st);

}  // namespace pd
#endif
