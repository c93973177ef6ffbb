// ORACLE -- TEST INFRASTRUCTURE ONLY. Side channel between the oracle-backed host driver (oracle_call_main.cpp) and the oracle:
// the reference visits the active pairs of a read group in the iteration order of a std::unordered_set<uint32_t> of pair
// ids (profile_structure_popdel_call.h:1037), which depends on the whole history of insertions and erasures. The host
// driver records that history (pdh::ActiveOp) and the ids of a segment's pair instances; handed over here before the
// segment is submitted, the oracle replays it on real std::unordered_sets and walks the pairs in exactly that order --
// only floating-point sums depend on it.
#ifndef PD_ORACLE_SIDE_CHANNEL_H_
#define PD_ORACLE_SIDE_CHANNEL_H_
#include <stdint.h>

struct pd_oracle_op
{
    uint32_t window;     // first window index of the segment genotyped after the mutation
    uint32_t id;
    uint32_t add;        // 1: emplace, 0: erase
};

// ops[rg_op_offsets[rg] .. rg_op_offsets[rg + 1]) are the mutations of read group rg since the previous segment, in
// order; instance_ids[i] is the id of pair instance i of the NEXT segment submitted (pd_segment order).
extern "C" void pd_oracle_next_segment_order(uint32_t n_readgroups, const uint64_t * rg_op_offsets, const pd_oracle_op * ops,
                                             const uint32_t * instance_ids, uint64_t n_instances);
#endif
