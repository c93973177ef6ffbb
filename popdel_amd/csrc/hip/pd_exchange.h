// Device-side all-gather between the contexts of one process (exchange.hip). Private to libpopdel_hip.so.
#ifndef PD_EXCHANGE_H_
#define PD_EXCHANGE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pd {

constexpr unsigned kXchMaxWorld = 16;
constexpr unsigned kXchAckWords = kXchMaxWorld;     // flag block: [0 .. 16) data words by sender, [16 .. 32) ack words by sender

struct FlagExchange
{
    uint8_t *  myArena;
    uint8_t *  peerArena[kXchMaxWorld];     // by rank (own entry unused)
    unsigned * myFlags;                     // this context's flag block (fine-grained, on its device)
    unsigned * peerFlags[kXchMaxWorld];     // the peers' flag blocks
    size_t     offset, bytesPerRank;
    unsigned   rank, world, seq;
    unsigned long long timeoutTicks;        // of wall_clock64() (100 MHz)
    unsigned * err;                         // page-locked host word: 0, or 1 / 2 = a wait for an ack / for data gave up
};

void launchFlagExchange(const FlagExchange & x, hipStream_t st);

}  // namespace pd
#endif
