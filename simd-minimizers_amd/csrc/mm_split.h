// mm_split.h — the expander of the split path (mm_split.hip): interface to the launcher in mm_fused.hip.
#pragma once
#include "mm_fused_impl.h"

namespace mm {

struct ExpandParams {
    const uint8_t *dump;               // tiles x dump_stride bytes (lane counts, then list rows)
    uint32_t dump_stride;
    unsigned long long *tile_status;   // kSplitValid | kSplitOverflow | rows << 32 | count, one word per tile
    uint32_t n_tiles;
    uint32_t S, NB;                    // windows per lane / per tile
    uint32_t list_cap;
    uint32_t mode_sub;                 // 1 for minimizer positions (list entries are element indices), 0 for syncmers
    uint32_t sk_shift;                 // kSkShift of the window size (super-k-mer entries)
    uint32_t win_begin;
    RedoEntry *redo_list;              // tiles whose lists overflowed: walked again by fused_kernel in redo mode
    uint32_t *redo_n;
    const unsigned long long *carry;   // outputs before this run
    uint32_t debug;
    OutParams out;
};

// E persistent workgroups on `stream`; 8-bit or 16-bit list entries, with or without super-k-mer indices
int launch_expand(const ExpandParams &p, bool e8, bool sk, uint32_t workgroups, uint32_t lds_bytes, hipStream_t stream);

}  // namespace mm
