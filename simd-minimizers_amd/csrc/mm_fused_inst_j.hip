// Explicit instantiations of the fused kernel (split over several files so that the
// build parallelises); the launcher in mm_fused.hip looks them up through fused_instances_j().
// Round 3: the even window sizes 18..32 (18, 20 here), which used to be compiled at first use.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedInstance *fused_instances_j(int *count) {
    static const FusedInstance kInst[] = {
        MM_FUSED_INST(18, true, true),
        MM_FUSED_INST(18, false, false),
        MM_FUSED_INST(20, true, true),
        MM_FUSED_INST(20, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
