// Reads-mode instantiations of the fused kernel (one lane per short read, minimizer positions);
// same window sizes as the sequence-mode list: 1..16, odd 17..33, 41, 51, canonical and forward.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedReadsInstance *fused_reads_instances_a(int *count) {
    static const FusedReadsInstance kInst[] = {
        MM_READS_INST(51, true, true),
        MM_READS_INST(31, true, true),
        MM_READS_INST(27, true, true),
        MM_READS_INST(23, true, true),
        MM_READS_INST(19, true, true),
        MM_READS_INST(16, true, true),
        MM_READS_INST(14, true, true),
        MM_READS_INST(13, true, true),
        MM_READS_INST(10, true, true),
        MM_READS_INST(9, true, true),
        MM_READS_INST(6, true, true),
        MM_READS_INST(5, true, true),
        MM_READS_INST(2, true, true),
        MM_READS_INST(1, true, true),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
