// Explicit instantiations of the pipelined kernel (mm_fused_pipe.h), split over several files so that
// the build parallelises; the launcher in mm_fused.hip looks them up through pipe_instances_g().
// Window sizes 10 and 7, canonical and forward.
#include "mm_fused_inst.h"

namespace mm {

const PipeInstance *pipe_instances_g(int *count) {
    static const PipeInstance kInst[] = {
        MM_PIPE_INST(10, true, true),
        MM_PIPE_INST(10, false, false),
        MM_PIPE_INST(7, true, true),
        MM_PIPE_INST(7, false, false),
    };
    *count = (int)(sizeof(kInst) / sizeof(kInst[0]));
    return kInst;
}

}  // namespace mm
