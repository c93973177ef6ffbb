// GPU box: what page-locking host memory costs -- hipHostMalloc against a huge-page mapping registered with the runtime -- and that both copy
// at the same rate (pinned_host.h). hipcc --offload-arch=gfx950 -O2 -o pin_probe tools/pin_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipSetDevice(0);
    const size_t bytes = 400u << 20;
    void * d = nullptr;
    hipMalloc(&d, bytes);
    hipStream_t st;
    hipStreamCreate(&st);
    for (int rep = 0; rep < 2; ++rep)
    {
        double t0 = now();
        void * p = nullptr;
        hipHostMalloc(&p, bytes, hipHostMallocDefault);
        double t1 = now();
        hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        double t2 = now();
        hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        double t3 = now();
        hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        double t4 = now();
        printf("hipHostMalloc: alloc %.3f s, D2H first %.3f s, D2H again %.3f s (%.1f GB/s), H2D %.3f s\n", t1 - t0, t2 - t1, t3 - t2, bytes / (t3 - t2) / 1e9, t4 - t3);
        hipHostFree(p);
        t0 = now();
        const size_t huge = 2u << 20;
        char * raw = static_cast<char *>(mmap(nullptr, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        char * al = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + huge - 1) / huge * huge);
        madvise(al, bytes, MADV_HUGEPAGE);
        std::vector<std::thread> th;
        for (int k = 0; k < 8; ++k)
            th.emplace_back([=] { for (size_t o = bytes / 8 * k; o < bytes / 8 * (k + 1); o += 4096) al[o] = 1; });
        for (auto & t : th) t.join();
        t1 = now();
        hipError_t e = hipHostRegister(al, bytes, hipHostRegisterDefault);
        t2 = now();
        hipMemcpyAsync(al, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        t3 = now();
        hipMemcpyAsync(al, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        t4 = now();
        hipMemcpyAsync(d, al, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
        double t5 = now();
        printf("huge pages + hipHostRegister(%d): map+touch %.3f s, register %.3f s, D2H first %.3f s, D2H again %.3f s (%.1f GB/s), H2D %.3f s\n", (int)e, t1 - t0, t2 - t1, t3 - t2, t4 - t3, bytes / (t4 - t3) / 1e9, t5 - t4);
        t0 = now();
        hipHostUnregister(al);
        munmap(raw, bytes + huge);
        printf("  unregister + munmap %.3f s\n", now() - t0);
        // without touching first
        t0 = now();
        raw = static_cast<char *>(mmap(nullptr, bytes + huge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        al = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(raw) + huge - 1) / huge * huge);
        madvise(al, bytes, MADV_HUGEPAGE);
        e = hipHostRegister(al, bytes, hipHostRegisterDefault);
        t1 = now();
        hipMemcpyAsync(al, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st);
        t2 = now();
        printf("huge pages untouched + register(%d) %.3f s, D2H first %.3f s\n", (int)e, t1 - t0, t2 - t1);
        hipHostUnregister(al);
        munmap(raw, bytes + huge);
    }
    return 0;
}
