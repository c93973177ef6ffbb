// Memory for the big per-sample arrays of `popdel call` (window tables, decoded record streams): a gigabyte and more for a
// thousand samples, touched once per round by whichever thread takes the sample. From the general heap that is 4-KB pages --
// a page fault per page on first use, a TLB miss per page in every pass, and at exit 0.2 s in which the kernel takes the
// pages back one by one. HugeAlloc<T> serves blocks of 32 KB and more from 2-MB-aligned regions that are advised as
// transparent huge pages (512 times fewer pages to fault in, map and free; where the system does not grant them the regions
// are ordinary memory), everything smaller from the heap.
//
// Freed blocks go to a per-size free list and are handed out again (sizes are rounded to four steps per power of two, so
// the slots of different samples and rounds meet in the same lists); regions are never returned before the process ends.
#ifndef PDH_HUGE_POOL_H_
#define PDH_HUGE_POOL_H_

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <new>
#include <unordered_map>
#include <utility>
#include <vector>
#ifdef __linux__
#include <sys/mman.h>
#endif

namespace pdh {

class HugePool
{
public:
    static constexpr size_t kMinBytes = 32u * 1024u;           // smaller requests go to the heap
    static constexpr size_t kRegionBytes = 64u * 1024u * 1024u;
    static constexpr size_t kHugePage = 2u * 1024u * 1024u;

    static HugePool & instance()
    {
        static HugePool * pool = new HugePool();               // (never destroyed: blocks may be freed by static destructors)
        return *pool;
    }

    // the size a request is served with: four steps per power of two
    static size_t classOf(size_t bytes)
    {
        size_t step = kMinBytes / 4;
        while (step * 8 < bytes)
            step *= 2;
        return (bytes + step - 1) / step * step;
    }

    void * allocate(size_t bytes)
    {
        const size_t size = classOf(bytes);
        std::lock_guard<std::mutex> guard(lock_);
        auto it = free_.find(size);
        if (it != free_.end() && !it->second.empty())
        {
            void * p = it->second.back();
            it->second.pop_back();
            return p;
        }
        if (left_ < size)
            newRegion(size);
        void * p = cursor_;
        cursor_ += size;
        left_ -= size;
        return p;
    }

    void release(void * p, size_t bytes)
    {
        std::lock_guard<std::mutex> guard(lock_);
        free_[classOf(bytes)].push_back(p);
    }

    // The regions handed out so far (begin, bytes): for a process about to end, which gives their pages back from several
    // threads at once instead of leaving them to the kernel's single-threaded exit path (discardRegion()).
    std::vector<std::pair<char *, size_t>> regions()
    {
        std::lock_guard<std::mutex> guard(lock_);
        return regions_;
    }
    static void discardRegion(char * begin, size_t bytes)
    {
#ifdef __linux__
        (void)madvise(begin, bytes, MADV_DONTNEED);
#else
        (void)begin;
        (void)bytes;
#endif
    }

private:
    HugePool() {}
    void newRegion(size_t atLeast)
    {
        // (what is left of the current region stays unused: less than one request)
        const size_t bytes = (std::max(atLeast, kRegionBytes) + kHugePage - 1) / kHugePage * kHugePage;
#ifdef __linux__
        void * map = mmap(nullptr, bytes + kHugePage, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (map == MAP_FAILED)
            throw std::bad_alloc();
        char * aligned = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(map) + kHugePage - 1) / kHugePage * kHugePage);
        (void)madvise(aligned, bytes, MADV_HUGEPAGE);
#else
        char * aligned = static_cast<char *>(std::malloc(bytes));
        if (!aligned)
            throw std::bad_alloc();
#endif
        cursor_ = aligned;
        left_ = bytes;
        regions_.emplace_back(aligned, bytes);
    }

    std::mutex lock_;
    std::unordered_map<size_t, std::vector<void *>> free_;
    char * cursor_ = nullptr;
    size_t left_ = 0;
    std::vector<std::pair<char *, size_t>> regions_;
};

template <typename T>
struct HugeAlloc
{
    using value_type = T;
    HugeAlloc() {}
    template <typename U>
    HugeAlloc(const HugeAlloc<U> &) {}
    T * allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < HugePool::kMinBytes)
            return static_cast<T *>(::operator new(bytes));
        return static_cast<T *>(HugePool::instance().allocate(bytes));
    }
    void deallocate(T * p, size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < HugePool::kMinBytes)
            ::operator delete(p);
        else
            HugePool::instance().release(p, bytes);
    }
    template <typename U>
    bool operator==(const HugeAlloc<U> &) const { return true; }
    template <typename U>
    bool operator!=(const HugeAlloc<U> &) const { return false; }
};

template <typename T>
using HugeVector = std::vector<T, HugeAlloc<T>>;

}  // namespace pdh
#endif
