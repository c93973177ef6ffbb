// Minimal fork-join helper for the host side (profile decoding, window stepping and staging are independent per read
// group / sample; results never depend on the number of threads).
#ifndef PDH_PARALLEL_H_
#define PDH_PARALLEL_H_

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>
#ifdef __linux__
#include <sched.h>
#endif

namespace pdh {

// CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU-time quota if there is one
// (cgroup v2 cpu.max / v1 cpu.cfs_quota_us). The GPU boxes show 256 logical CPUs and grant a job 16 of them: more
// threads than that only get throttled.
inline unsigned usableCpus()
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0)
        hw = 1;
#ifdef __linux__
    {
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0)
        {
            const int n = CPU_COUNT(&set);
            if (n > 0 && static_cast<unsigned>(n) < hw)
                hw = static_cast<unsigned>(n);
        }
    }
    auto quotaOf = [](const char * path, bool v2) -> double {
        FILE * f = fopen(path, "r");
        if (!f)
            return 0;
        char a[64] = {0};
        long long period = 100000;
        double cpus = 0;
        if (v2)
        {
            if (fscanf(f, "%63s %lld", a, &period) >= 1 && a[0] != 'm' && period > 0)
                cpus = atof(a) / static_cast<double>(period);
        }
        else
        {
            long long q = -1;
            if (fscanf(f, "%lld", &q) == 1 && q > 0)
            {
                FILE * g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
                if (g)
                {
                    if (fscanf(g, "%lld", &period) != 1 || period <= 0)
                        period = 100000;
                    fclose(g);
                }
                cpus = static_cast<double>(q) / static_cast<double>(period);
            }
        }
        fclose(f);
        return cpus;
    };
    double q = quotaOf("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0)
        q = quotaOf("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
    if (q >= 1 && q < hw)
        hw = static_cast<unsigned>(q + 0.5);
#endif
    return hw;
}

// Keeps the process on the CPUs of ONE memory node when it is free to run on several (the GPU boxes: two sockets, 256
// logical CPUs, of which a job's quota pays for 16): the per-sample tables are written by one round's threads and read by
// the next round's, and threads are started per pass wherever the scheduler finds room -- across sockets every other access
// to them is remote (`popdel call` on 1,000 x 1 Mb: 8 % less CPU time, less run-to-run spread). The node is the one the
// calling thread runs on; an affinity mask that already lies inside one node is left alone, POPDEL_NO_NODE_PIN=1 turns it off.
inline void stayOnOneMemoryNode()
{
#ifdef __linux__
    if (getenv("POPDEL_NO_NODE_PIN"))
        return;
    cpu_set_t mine;
    if (sched_getaffinity(0, sizeof(mine), &mine) != 0)
        return;
    const int here = sched_getcpu();
    if (here < 0)
        return;
    for (int node = 0; node < 64; ++node)
    {
        char path[96];
        std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        FILE * f = fopen(path, "r");
        if (!f)
            break;
        cpu_set_t of;
        CPU_ZERO(&of);
        int a = 0, b = 0;
        char sep = 0;
        bool mineIsHere = false;
        // "0-63,128-191"
        while (fscanf(f, "%d", &a) == 1)
        {
            b = a;
            int c = fgetc(f);
            if (c == '-')
            {
                if (fscanf(f, "%d", &b) != 1)
                    break;
                c = fgetc(f);
            }
            for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu)
                CPU_SET(cpu, &of);
            if (here >= a && here <= b)
                mineIsHere = true;
            sep = static_cast<char>(c);
            if (sep != ',')
                break;
        }
        fclose(f);
        if (!mineIsHere)
            continue;
        cpu_set_t both;
        CPU_AND(&both, &mine, &of);
        if (CPU_COUNT(&both) > 0 && CPU_COUNT(&both) < CPU_COUNT(&mine))
            (void)sched_setaffinity(0, sizeof(both), &both);
        return;
    }
#endif
}

inline unsigned hostThreads()
{
    static const unsigned n = [] {
        if (const char * e = getenv("POPDEL_THREADS"))
        {
            int v = atoi(e);
            if (v >= 1)
                return static_cast<unsigned>(v);
        }
        // a quarter more than the usable CPUs: the passes are memory-bound and fault pages in, a few extra threads fill the
        // stalls (GPU box, 16 usable CPUs, popdel call on 1000 samples, round 2: 16 threads 1.84 s, 32 and 64 threads 1.62 s;
        // round 5, 4,000 samples x 4 Mb with the round bound by the CPU quota, seconds per warmed round / CPU-seconds per round,
        // threads started per pass: 16 threads 0.0543 / 0.77, 24: 0.0513 / 0.82, 32: 0.0535 / 0.85, 48: 0.0564 / 0.88; with the
        // parked workers: 20: 0.0478 / 0.73, 24: 0.0500 / 0.77, 28: 0.0523 / 0.83 -- every thread a pass uses costs)
        const unsigned hw = usableCpus() + usableCpus() / 4u;
        return hw > 64u ? 64u : hw;
    }();
    return n;
}

// Calls fn(i) for i in [0, n) on up to hostThreads() threads (dynamic chunks). Exceptions are re-thrown on the caller.
// Round 3 measured a persistent pool (condition-variable hand-off to parked workers) 10-20 % SLOWER than threads spawned per
// call on the 256-core GPU host for the memory-bound staging passes, with 64 threads and no CPU quota in the way; with 24
// threads on 16 CPUs the pool below wins by 3 % (detail::Pool) -- passes that find it taken still spawn.
// `maxThreads` (0: no extra limit) caps the threads of passes that were measured to get SLOWER with more of them.
// The threads get 256-KB stacks (their bodies are loops over table entries and bytes; nothing here recurses): glibc keeps the
// stacks of finished threads for re-use up to 40 MB in all, which is five default-sized ones -- a round of `popdel call`
// starts a thousand threads and spent a tenth of its CPU time mapping and unmapping their stacks.
namespace detail {
inline const pthread_attr_t * smallStack()
{
    static const pthread_attr_t * attr = [] {
        static pthread_attr_t a;
        if (pthread_attr_init(&a) != 0 || pthread_attr_setstacksize(&a, 256u * 1024u) != 0)
            return static_cast<const pthread_attr_t *>(nullptr);
        return static_cast<const pthread_attr_t *>(&a);
    }();
    return attr;
}
template <typename W>
void * runWorker(void * w)
{
    (*static_cast<W *>(w))();
    return nullptr;
}
}  // namespace detail

namespace detail {
// Parked workers for the passes of ONE caller at a time: whoever finds the pool taken -- another thread's pass is running on
// it, or the caller is itself one of its workers -- starts threads as before (POPDEL_THREAD_POOL=0: always). Round 5, `popdel
// call` on 4,000 samples x 4 Mb under the job's CPU quota, same box, seconds / CPU-seconds per warmed round: 0.0507 / 0.803
// with threads started per pass, 0.0494 / 0.768 with the pool (a pass of the round loop finds it free, the loaders' and the
// hand-over thread's passes mostly start their own).
class Pool
{
public:
    static Pool & instance()
    {
        static Pool * p = new Pool();       // (never destroyed: its workers are parked for the life of the process)
        return *p;
    }
    static bool wanted()
    {
        static const bool on = [] {
            const char * e = getenv("POPDEL_THREAD_POOL");
            return !(e && e[0] == '0');
        }();
        return on;
    }
    // runs job() on up to `helpers` parked workers and on the caller; false: the pool is taken (nothing was run)
    template <typename W>
    bool run(W & job, unsigned helpers)
    {
        if (busy_.exchange(true, std::memory_order_acquire))
            return false;
        {
            std::unique_lock<std::mutex> lock(m_);
            while (workers_.size() < helpers)
            {
                pthread_t id;
                if (pthread_create(&id, smallStack(), &Pool::park, this) != 0)
                    break;
                workers_.push_back(id);
            }
            helpers = std::min<unsigned>(helpers, static_cast<unsigned>(workers_.size()));
            call_ = [](void * w) { (*static_cast<W *>(w))(); };
            arg_ = &job;
            wanted_ = helpers;
            running_ = helpers;
            ++generation_;
        }
        wake_.notify_all();
        job();
        {
            std::unique_lock<std::mutex> lock(m_);
            done_.wait(lock, [this] { return running_ == 0; });
            wanted_ = 0;
        }
        busy_.store(false, std::memory_order_release);
        return true;
    }

private:
    static void * park(void * self)
    {
        Pool & p = *static_cast<Pool *>(self);
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(p.m_);
        for (;;)
        {
            p.wake_.wait(lock, [&] { return p.generation_ != seen && p.wanted_ != 0; });
            seen = p.generation_;
            --p.wanted_;
            void (*call)(void *) = p.call_;
            void * arg = p.arg_;
            lock.unlock();
            call(arg);
            lock.lock();
            if (--p.running_ == 0)
                p.done_.notify_one();
        }
        return nullptr;
    }
    std::atomic<bool> busy_{false};
    std::mutex m_;
    std::condition_variable wake_, done_;
    std::vector<pthread_t> workers_;
    void (*call_)(void *) = nullptr;
    void * arg_ = nullptr;
    unsigned wanted_ = 0, running_ = 0;
    uint64_t generation_ = 0;
};
}  // namespace detail

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
    if (detail::Pool::wanted() && detail::Pool::instance().run(worker, threads - 1))
    {
        if (error)
            std::rethrow_exception(error);
        return;
    }
    std::vector<pthread_t> pool;
    pool.reserve(threads - 1);
    for (unsigned t = 1; t < threads; ++t)
    {
        pthread_t id;
        if (pthread_create(&id, detail::smallStack(), &detail::runWorker<decltype(worker)>, &worker) != 0)
            break;              // (no more threads to be had: the ones there are take the work)
        pool.push_back(id);
    }
    worker();
    for (pthread_t id : pool)
        (void)pthread_join(id, nullptr);
    if (error)
        std::rethrow_exception(error);
}

}  // namespace pdh
#endif
