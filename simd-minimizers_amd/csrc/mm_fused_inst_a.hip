// Explicit instantiations of the fused kernel (split over several files so that the
// build parallelises); the launcher in mm_fused.hip looks them up through fused_instances_a().
// Generated list: window sizes 1..16, odd 17..33, 41, 51, canonical and forward.
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const FusedInstance *fused_instances_a(int *count) {
    static const FusedInstance kInst[] = {
        MM_FUSED_INST(51, true, true),
        MM_FUSED_INST(29, false, false),
        MM_FUSED_INST(19, true, true),
        MM_FUSED_INST(14, false, false),
        MM_FUSED_INST(9, true, true),
        MM_FUSED_INST(5, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
