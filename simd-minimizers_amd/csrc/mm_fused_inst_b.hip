// Explicit instantiations of the fused kernel (split over several files so that the
// build parallelises); the launcher in mm_fused.hip looks them up through fused_instances_b().
// Generated list: window sizes 1..16, odd 17..33, 41, 51, canonical and forward.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedInstance *fused_instances_b(int *count) {
    static const FusedInstance kInst[] = {
        MM_FUSED_INST(51, false, false),
        MM_FUSED_INST(27, true, true),
        MM_FUSED_INST(19, false, false),
        MM_FUSED_INST(13, true, true),
        MM_FUSED_INST(9, false, false),
        MM_FUSED_INST(4, true, true),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
