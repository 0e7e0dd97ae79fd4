// Explicit instantiations of the pipelined kernel (mm_fused_pipe.h), split over several files so that
// the build parallelises; the launcher in mm_fused.hip looks them up through pipe_instances_b().
// Window sizes 15 and 2, canonical and forward.
#include "mm_fused_inst.h"

namespace mm {

const PipeInstance *pipe_instances_b(int *count) {
    static const PipeInstance kInst[] = {
        MM_PIPE_INST(15, true, true),
        MM_PIPE_INST(15, false, false),
        MM_PIPE_INST(2, true, true),
        MM_PIPE_INST(2, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
