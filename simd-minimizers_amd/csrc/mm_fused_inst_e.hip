// Explicit instantiations of the fused kernel (split over several files so that the
// build parallelises); the launcher in mm_fused.hip looks them up through fused_instances_e().
// Generated list: window sizes 1..16, odd 17..33, 41, 51, canonical and forward.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedInstance *fused_instances_e(int *count) {
    static const FusedInstance kInst[] = {
        MM_FUSED_INST(33, true, true),
        MM_FUSED_INST(25, false, false),
        MM_FUSED_INST(16, true, true),
        MM_FUSED_INST(12, false, false),
        MM_FUSED_INST(7, true, true),
        MM_FUSED_INST(3, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
