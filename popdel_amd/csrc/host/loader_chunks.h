// How the device loader of `popdel call` cuts a round's readers into chunks (one gunzip + parse launch each), and how many
// samples a batch of the replay may span (call_driver.cpp: GunzipPrefetch; tests/test_abi_and_host.py checks the relation).
#ifndef PDH_LOADER_CHUNKS_H_
#define PDH_LOADER_CHUNKS_H_

#include <algorithm>
#include <cstddef>
#include <cstdlib>

namespace pdh {

#ifndef PDH_LOADER_INFLATERS
#define PDH_LOADER_INFLATERS 3          // (measurements: -DPDH_LOADER_INFLATERS=n builds a binary with n inflaters per loader)
#endif
constexpr size_t kLoaderInflaters = PDH_LOADER_INFLATERS;

inline size_t loaderEnvOr(const char * name, size_t dflt)
{
    const char * e = getenv(name);
    const long v = e ? atol(e) : 0;
    return v >= 64 ? static_cast<size_t>(v) : dflt;
}

// Items per chunk: a round's readers are dealt to the three inflaters (a thread each) in three equal chunks -- the gunzip
// kernel is a chain of dependent instructions per member and wants many members per launch (a launch of 21,000 takes 5.4 ms,
// one of 200,000 takes 37), and with the staging copy of the walk gone the device's work on the blocks is what a warmed
// round of `popdel call` waits for. Measured on 4,000 samples x 4 Mb, same box (first chunk : the others, seconds per
// warmed round): 256 : 1024 (rounds 4-5: five chunks) 0.0604, 512 : 2048 0.0546, 1344 : 1344 (three) 0.0545, 1024 : 3072
// 0.0574, 512 : 4096 0.0615, 2048 : 2048 (two) 0.0563; round 6, same cohort, records staying on the device: three chunks 0.029,
// two 0.035, one 0.046. Cohorts beyond 52,224 readers take more chunks, an inflater's buffers go round as before.
// POPDEL_LOADER_CHUNK / POPDEL_LOADER_FIRST_CHUNK fix the sizes (measurements).
inline size_t loaderChunkFor(size_t readers)
{
    static const size_t fixed = loaderEnvOr("POPDEL_LOADER_CHUNK", 0);
    if (fixed)
        return fixed;
    // (round 6: at most 17,408 items per chunk -- three chunks take the 50,000 profiles of BASELINE's largest cohort, so that a round
    // runs each inflater once and the records it parsed are still on the device when the round's segment is handed over
    // (pd_walk_source). With lean arrays and records copied back for the ends of a stretch only a chunk of that size is ~1 GB
    // page-locked and ~5 GB on the device; rounds 4-5 capped at 4,096.)
    const size_t per = ((readers + kLoaderInflaters - 1) / kLoaderInflaters + 63) / 64 * 64;
    return std::min<size_t>(17408, std::max<size_t>(256, per));
}
inline size_t loaderFirstChunkFor(size_t readers)
{
    static const size_t fixed = loaderEnvOr("POPDEL_LOADER_FIRST_CHUNK", 0);
    return fixed ? fixed : loaderChunkFor(readers);
}
// Samples a batch of the replay may span: the first two chunks. An inflater's buffers are recycled once the replay has passed
// the chunk three before it, and the replay releases a batch as a whole -- a batch that touched FOUR consecutive chunks would
// wait for a chunk whose buffers it holds itself. The loader chunks a round's READERS, at most the samples: with the limit
// taken for the number of samples, a batch of samples holds at most that many readers and so touches at most three chunks
// of any round (two, or two and the tail of a third).
inline size_t loaderBatchLimitFor(size_t samples) { return loaderFirstChunkFor(samples) + loaderChunkFor(samples); }

}  // namespace pdh
#endif
