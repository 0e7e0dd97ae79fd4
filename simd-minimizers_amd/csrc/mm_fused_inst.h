// mm_fused_inst.h — table entry describing one specialisation of the fused kernel.
#pragma once
#include "mm_fused_impl.h"

namespace mm {

using FusedKernelFn = void (*)(const FusedParams);

struct FusedInstance {
    uint32_t w;
    bool canon;    // canonical windows (strand vote)
    bool hash_rc;  // canonical hasher (forward + reverse-complement hash)
    FusedKernelFn fn[4];  // minimizers, closed syncmers, open syncmers, minimizers + super-k-mers
};

#define MM_FUSED_INST(W, C, R)                                                              \
    {                                                                                       \
        W, C, R, {                                                                          \
            &fused_kernel<W, C, R, 0, false, false>, &fused_kernel<W, C, R, 1, false, false>, \
                &fused_kernel<W, C, R, 2, false, false>, &fused_kernel<W, C, R, 0, true, false> \
        }                                                                                   \
    }

const FusedInstance *fused_instances_a(int *count);
const FusedInstance *fused_instances_b(int *count);
const FusedInstance *fused_instances_c(int *count);
const FusedInstance *fused_instances_d(int *count);
const FusedInstance *fused_instances_e(int *count);
const FusedInstance *fused_instances_f(int *count);
const FusedInstance *fused_instances_g(int *count);
const FusedInstance *fused_instances_h(int *count);
const FusedInstance *fused_instances_i(int *count);
const FusedInstance *fused_instances_j(int *count);
const FusedInstance *fused_instances_k(int *count);
const FusedInstance *fused_instances_l(int *count);
const FusedInstance *fused_instances_m(int *count);

// reads-mode instances (minimizers only), mm_fused_inst_reads_*.hip
struct FusedReadsInstance {
    uint32_t w;
    bool canon;
    bool hash_rc;
    FusedKernelFn fn;
};
#define MM_READS_INST(W, C, R) \
    { W, C, R, &fused_kernel<W, C, R, 0, false, true> }
const FusedReadsInstance *fused_reads_instances_a(int *count);
const FusedReadsInstance *fused_reads_instances_b(int *count);
const FusedReadsInstance *fused_reads_instances_c(int *count);
const FusedReadsInstance *fused_reads_instances_d(int *count);
const FusedReadsInstance *fused_reads_instances_e(int *count);

// walk kernels of the split path (walk_kernel), mm_walk_inst.hip
struct WalkInstance {
    uint32_t w;
    bool canon, hash_rc;
    uint32_t mode;
    bool sk;
    FusedKernelFn fn;
};
#define MM_WALK_INST(W, C, R, MODE, SK) \
    { W, C, R, MODE, SK, &walk_kernel<W, C, R, MODE, SK> }
const WalkInstance *walk_instances(int *count);

}  // namespace mm
