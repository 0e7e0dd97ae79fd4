// Reads-mode instantiations of the fused kernel (one lane per short read, minimizer positions);
// same window sizes as the sequence-mode list: 1..16, odd 17..33, 41, 51, canonical and forward.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedReadsInstance *fused_reads_instances_c(int *count) {
    static const FusedReadsInstance kInst[] = {
        MM_READS_INST(41, true, true),
        MM_READS_INST(33, true, true),
        MM_READS_INST(29, true, true),
        MM_READS_INST(25, true, true),
        MM_READS_INST(21, true, true),
        MM_READS_INST(17, true, true),
        MM_READS_INST(15, true, true),
        MM_READS_INST(12, true, true),
        MM_READS_INST(11, true, true),
        MM_READS_INST(8, true, true),
        MM_READS_INST(7, true, true),
        MM_READS_INST(4, true, true),
        MM_READS_INST(3, true, true),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
