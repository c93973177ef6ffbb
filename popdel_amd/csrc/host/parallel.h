// Minimal fork-join helper for the host side (profile decoding, window stepping and staging are independent per read
// group / sample; results never depend on the number of threads).
#ifndef PDH_PARALLEL_H_
#define PDH_PARALLEL_H_

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace pdh {

inline unsigned hostThreads()
{
    static const unsigned n = [] {
        if (const char * e = getenv("POPDEL_THREADS"))
        {
            int v = atoi(e);
            if (v >= 1)
                return static_cast<unsigned>(v);
        }
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0)
            hw = 1;
        return hw > 64u ? 64u : hw;
    }();
    return n;
}

// Calls fn(i) for i in [0, n) on up to hostThreads() threads (dynamic chunks). Exceptions are re-thrown on the caller.
// Threads are spawned per call on purpose: a persistent pool (condition-variable hand-off to parked workers) was
// measured 10-20 % SLOWER on the 256-core GPU host for the memory-bound staging passes.
// `maxThreads` (0: no extra limit) caps the threads of passes that were measured to get SLOWER with more of them.
template <typename F>
void parallelFor(size_t n, F fn, size_t grain = 1, unsigned maxThreads = 0)
{
    unsigned threads = static_cast<unsigned>(std::min<size_t>(hostThreads(), (n + grain - 1) / grain));
    if (maxThreads != 0)
        threads = std::min(threads, maxThreads);
    if (threads <= 1)
    {
        for (size_t i = 0; i < n; ++i)
            fn(i);
        return;
    }
    std::atomic<size_t> next(0);
    std::exception_ptr error;
    std::mutex errorLock;
    auto worker = [&]() {
        try
        {
            for (;;)
            {
                const size_t begin = next.fetch_add(grain);
                if (begin >= n)
                    break;
                const size_t end = std::min(n, begin + grain);
                for (size_t i = begin; i < end; ++i)
                    fn(i);
            }
        }
        catch (...)
        {
            std::lock_guard<std::mutex> g(errorLock);
            if (!error)
                error = std::current_exception();
            next.store(n);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(threads - 1);
    for (unsigned t = 1; t < threads; ++t)
        pool.emplace_back(worker);
    worker();
    for (std::thread & t : pool)
        t.join();
    if (error)
        std::rethrow_exception(error);
}

}  // namespace pdh
#endif
