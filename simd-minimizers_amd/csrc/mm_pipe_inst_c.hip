// Explicit instantiations of the pipelined kernel (mm_fused_pipe.h), split over several files so that
// the build parallelises; the launcher in mm_fused.hip looks them up through pipe_instances_c().
// Window sizes 14 and 3, canonical and forward.
#include "mm_fused_inst.h"

namespace mm {

const PipeInstance *pipe_instances_c(int *count) {
    static const PipeInstance kInst[] = {
        MM_PIPE_INST(14, true, true),
        MM_PIPE_INST(14, false, false),
        MM_PIPE_INST(3, true, true),
        MM_PIPE_INST(3, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
