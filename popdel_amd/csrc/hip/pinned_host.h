// Page-locked host memory for the library's staging and result buffers. hipHostMalloc() page-locks 4-KB pages one by one:
// 55-90 ms per 400 MB on the GPU boxes -- 0.6 s for the buffers a `popdel call` run sets up in its first rounds, 0.2 s for a
// context's first staged segment. An anonymous mapping aligned to 2 MB and advised as transparent huge pages, faulted in by
// a few threads and then registered with the runtime (hipHostRegister) is there in 4 ms per 400 MB and copies at the same
// rate (57 GB/s either way; measured with tools/pin_probe in round 4). Where the system grants no huge pages the mapping is
// ordinary memory and registering it costs what hipHostMalloc costs. Small blocks and POPDEL_HIP_PINNED=malloc take hipHostMalloc.
#ifndef PD_PINNED_HOST_H_
#define PD_PINNED_HOST_H_
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>
#ifdef __linux__
#include <sys/mman.h>
#endif

namespace pd {

struct PinnedRegistry
{
    struct Region
    {
        void * raw = nullptr;       // nullptr: a hipHostMalloc block
        size_t mapped = 0;
    };
    std::mutex lock;
    std::unordered_map<void *, Region> regions;
    static PinnedRegistry & instance()
    {
        static PinnedRegistry * r = new PinnedRegistry();       // (never destroyed: blocks may be freed by static destructors)
        return *r;
    }
};

// nullptr on failure (hipGetLastError() cleared)
inline void * pinnedAlloc(size_t bytes)
{
    PinnedRegistry & reg = PinnedRegistry::instance();
    void * p = nullptr;
#ifdef __linux__
    static const bool viaMalloc = [] {
        const char * e = getenv("POPDEL_HIP_PINNED");
        return e && e[0] == 'm';
    }();
    constexpr size_t kHuge = 2u << 20;
    if (!viaMalloc && bytes >= (4u << 20))
    {
        const size_t want = (bytes + kHuge - 1) / kHuge * kHuge;
        void * raw = mmap(nullptr, want + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw != MAP_FAILED)
        {
            char * al = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + kHuge - 1) / kHuge * kHuge);
            (void)madvise(al, want, MADV_HUGEPAGE);
            // first touch from a few threads (a huge page is zeroed by whoever faults it in)
            const unsigned threads = want >= (64u << 20) ? 4u : 1u;
            std::vector<std::thread> pool;
            size_t mine = want;             // (what no helper thread takes is touched here)
            try
            {
                for (unsigned k = threads - 1; k >= 1; --k)
                {
                    pool.emplace_back([=] {
                        for (size_t o = want / threads * k; o < want / threads * (k + 1); o += 4096)
                            al[o] = 0;
                    });
                    mine = want / threads * k;
                }
            }
            catch (...)
            {
            }
            for (size_t o = 0; o < mine; o += 4096)
                al[o] = 0;
            for (std::thread & t : pool)
                t.join();
            if (hipHostRegister(al, want, hipHostRegisterDefault) == hipSuccess)
            {
                std::lock_guard<std::mutex> guard(reg.lock);
                reg.regions[al] = PinnedRegistry::Region{raw, want + kHuge};
                return al;
            }
            (void)hipGetLastError();
            munmap(raw, want + kHuge);
        }
    }
#endif
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess)
    {
        (void)hipGetLastError();
        return nullptr;
    }
    std::lock_guard<std::mutex> guard(reg.lock);
    reg.regions[p] = PinnedRegistry::Region{nullptr, 0};
    return p;
}

inline void pinnedFree(void * p)
{
    if (!p)
        return;
    PinnedRegistry & reg = PinnedRegistry::instance();
    PinnedRegistry::Region r;
    {
        std::lock_guard<std::mutex> guard(reg.lock);
        auto it = reg.regions.find(p);
        if (it == reg.regions.end())
        {
            (void)hipHostFree(p);       // (not ours: allocated before this header took over)
            return;
        }
        r = it->second;
        reg.regions.erase(it);
    }
    if (!r.raw)
    {
        (void)hipHostFree(p);
        return;
    }
    (void)hipHostUnregister(p);
#ifdef __linux__
    munmap(r.raw, r.mapped);
#endif
}

}  // namespace pd
#endif
