#include "../../simd-minimizers_amd/csrc/mm_fused_impl.h"
namespace mm {
template __global__ void fused_kernel<WW, true, true, 0, false, false>(const FusedParams);
}
