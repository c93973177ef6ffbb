// Diagnosis: where the process's CPU time goes (POPDEL_CPU_PROFILE=<file>). A profiling timer interrupts whichever thread is
// running every millisecond of CPU time; the handler notes the interrupted instruction's address. At the end the addresses
// are written as "<samples> <module> <offset>" (tools/cpu_profile_symbols.py turns them into function names with nm, on any
// machine that has the same binaries). No symbols, no unwinding: leaf functions only -- enough to see what a round's CPU
// quota is spent on.
#ifndef PDH_CPU_SAMPLER_H_
#define PDH_CPU_SAMPLER_H_
#ifdef __linux__
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <dlfcn.h>
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>

namespace pdh {

class CpuSampler
{
public:
    static void start()
    {
        const char * path = getenv("POPDEL_CPU_PROFILE");
        if (!path || !*path || samples())
            return;
        samples() = static_cast<uintptr_t *>(calloc(kCap, sizeof(uintptr_t)));
        if (!samples())
            return;
        struct sigaction sa;
        sa.sa_sigaction = &CpuSampler::onTick;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_SIGINFO | SA_RESTART;
        sigaction(SIGPROF, &sa, nullptr);
        struct itimerval tv;
        tv.it_interval.tv_sec = tv.it_value.tv_sec = 0;
        tv.it_interval.tv_usec = tv.it_value.tv_usec = 1000;
        setitimer(ITIMER_PROF, &tv, nullptr);
    }
    static void stopAndWrite()
    {
        const char * path = getenv("POPDEL_CPU_PROFILE");
        if (!path || !samples())
            return;
        struct itimerval off = {{0, 0}, {0, 0}};
        setitimer(ITIMER_PROF, &off, nullptr);
        const size_t n = std::min<size_t>(count().load(), kCap);
        std::map<std::pair<std::string, uintptr_t>, size_t> hist;
        for (size_t i = 0; i < n; ++i)
        {
            Dl_info info;
            std::string module = "?";
            uintptr_t off = samples()[i];
            if (dladdr(reinterpret_cast<void *>(samples()[i]), &info) && info.dli_fname)
            {
                module = info.dli_fname;
                off -= reinterpret_cast<uintptr_t>(info.dli_fbase);
            }
            ++hist[{module, off}];
        }
        if (FILE * f = fopen(path, "w"))
        {
            fprintf(f, "# %zu samples, one per millisecond of CPU time\n", n);
            for (const auto & e : hist)
                fprintf(f, "%zu %s %lx\n", e.second, e.first.first.c_str(), static_cast<unsigned long>(e.first.second));
            fclose(f);
        }
    }

private:
    static constexpr size_t kCap = 1u << 20;
    static uintptr_t *& samples()
    {
        static uintptr_t * p = nullptr;
        return p;
    }
    static std::atomic<size_t> & count()
    {
        static std::atomic<size_t> c{0};
        return c;
    }
    static void onTick(int, siginfo_t *, void * context)
    {
        const size_t i = count().fetch_add(1, std::memory_order_relaxed);
        if (i < kCap)
            samples()[i] = static_cast<uintptr_t>(static_cast<ucontext_t *>(context)->uc_mcontext.gregs[REG_RIP]);
    }
};

}  // namespace pdh
#endif
#endif
