// Reads-mode instances of the fused kernel (minimizers only) for the even window sizes 18..32 (round 3).
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedReadsInstance *fused_reads_instances_e(int *count) {
    static const FusedReadsInstance kInst[] = {
        MM_READS_INST(18, true, true),
        MM_READS_INST(18, false, false),
        MM_READS_INST(20, true, true),
        MM_READS_INST(20, false, false),
        MM_READS_INST(22, true, true),
        MM_READS_INST(22, false, false),
        MM_READS_INST(24, true, true),
        MM_READS_INST(24, false, false),
        MM_READS_INST(26, true, true),
        MM_READS_INST(26, false, false),
        MM_READS_INST(28, true, true),
        MM_READS_INST(28, false, false),
        MM_READS_INST(30, true, true),
        MM_READS_INST(30, false, false),
        MM_READS_INST(32, true, true),
        MM_READS_INST(32, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
