// Reads-mode instantiations of the fused kernel (one lane per short read, minimizer positions);
// same window sizes as the sequence-mode list: 1..16, odd 17..33, 41, 51, canonical and forward.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedReadsInstance *fused_reads_instances_b(int *count) {
    static const FusedReadsInstance kInst[] = {
        MM_READS_INST(51, false, false),
        MM_READS_INST(31, false, false),
        MM_READS_INST(27, false, false),
        MM_READS_INST(23, false, false),
        MM_READS_INST(19, false, false),
        MM_READS_INST(16, false, false),
        MM_READS_INST(14, false, false),
        MM_READS_INST(13, false, false),
        MM_READS_INST(10, false, false),
        MM_READS_INST(9, false, false),
        MM_READS_INST(6, false, false),
        MM_READS_INST(5, false, false),
        MM_READS_INST(2, false, false),
        MM_READS_INST(1, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
