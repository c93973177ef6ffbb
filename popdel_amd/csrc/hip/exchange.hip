// The fourth exchange carrier (round 5): all-gather between the contexts of ONE process -- one per GPU, `popdel call` with
// POPDEL_DEVICES -- without the host in the data path. The three older carriers: host callbacks (torch.distributed), the
// library's own ncclAllGather on the stream, and peer copies bracketed by barriers of the host threads (two stream
// synchronisations and two barriers per exchange: what this one replaces).
//
// Every context owns an exchange arena (same layout everywhere: [world][bytesPerRank] at the same offset) and a small
// flag block in fine-grained memory. Exchange number k of a context, ONE kernel launch on the context's stream, one
// workgroup per peer p:
//   1. ack   store k into p's ack word for me: everything I enqueued before this kernel -- in particular the kernels that READ
//            exchange k - 1 out of my arena -- has completed (stream order), so p may overwrite its chunk in my arena;
//      wait  until p's ack for exchange k has arrived here (p is done reading what I wrote for k - 1);
//   2. push  copy my chunk into p's arena through the peer-mapped pointer, fence at system scope, store k into p's
//            data word for me;
//   3. wait  until p's data word here says k: p's chunk is in my arena.
// The kernel that consumes the gathered area follows in stream order. No host barrier, no stream synchronisation, no
// library launch: two one-way flag latencies over the link per exchange. The exchanges of a context are issued in the same order
// on every context (the collective sequence is the same everywhere), k is just their running number.
// A wait gives up after `timeoutTicks` of the 100-MHz wall clock and raises *err (page-locked host word) instead of hanging
// the device: the host fails the segment. (How that can happen: two contexts on ONE GPU whose streams share a hardware
// queue -- the waiting kernel then sits in front of the kernel it waits for. One context per GPU, the case this is made
// for, has no such dependence; tests on one GPU raise GPU_MAX_HW_QUEUES.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pd_exchange.h"

namespace pd {

__device__ __forceinline__ void flagStore(unsigned * p, unsigned v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool flagWait(const unsigned * p, unsigned want, unsigned long long timeoutTicks, unsigned * err, unsigned code)
{
    const unsigned long long t0 = wall_clock64();
    for (;;)
    {
        // (>= in wrap-around arithmetic: the peer may be an exchange ahead with its ack)
        if ((int)(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) >= 0)
            return true;
        if (wall_clock64() - t0 > timeoutTicks)
        {
            __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ __launch_bounds__(256) void flagExchangeKernel(FlagExchange x)
{
    unsigned p = blockIdx.x;
    if (p >= x.rank)
        ++p;                                            // my peers in rank order, myself left out
    __shared__ int ok;
    if (threadIdx.x == 0)
    {
        flagStore(x.peerFlags[p] + kXchAckWords + x.rank, x.seq);
        ok = flagWait(x.myFlags + kXchAckWords + p, x.seq, x.timeoutTicks, x.err, 1u) ? 1 : 0;
    }
    __syncthreads();
    if (ok)
    {
        const uint4 * src = reinterpret_cast<const uint4 *>(x.myArena + x.offset + (size_t)x.rank * x.bytesPerRank);
        uint4 * dst = reinterpret_cast<uint4 *>(x.peerArena[p] + x.offset + (size_t)x.rank * x.bytesPerRank);
        const size_t n16 = x.bytesPerRank / 16u;        // (chunks are multiples of 256 bytes)
        for (size_t i = threadIdx.x; i < n16; i += blockDim.x)
            dst[i] = src[i];
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        flagStore(x.peerFlags[p] + x.rank, x.seq);      // (also after a failed wait: the peer must not hang on us)
        flagWait(x.myFlags + p, x.seq, x.timeoutTicks, x.err, 2u);
    }
}

void launchFlagExchange(const FlagExchange & x, hipStream_t st)
{
    if (x.world < 2)
        return;
    hipLaunchKernelGGL(flagExchangeKernel, dim3(x.world - 1), dim3(256), 0, st, x);
}

}  // namespace pd
