// Prebuilt walk kernels of the split path (walk_kernel, mm_fused_impl.h) for the window sizes the BASELINE
// configurations use; every other plan's walk kernel is specialised at first use (mm_jit.hip).
#include "mm_fused_impl.h"
#include "mm_fused_inst.h"

namespace mm {

const WalkInstance *walk_instances(int *count) {
    static const WalkInstance kInst[] = {
        MM_WALK_INST(11, false, false, 0, false),
        MM_WALK_INST(11, true, true, 0, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
