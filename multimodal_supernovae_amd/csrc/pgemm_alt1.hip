// Instantiations of the plane GEMM kernel templates (pgemm_kernels.h) that csrc/pgemm.hip launches but does not compile itself:
// the 2 x 4 wave layout (msn_set_pgemm_variant(0)) and the 16 x 16 x 32 form (variant 2) of the 3-plane NT kernel.  A translation unit of their own so that they compile beside pgemm.hip.
#include "pgemm_kernels.h"

namespace msn {
template __global__ void pgemm_nt_kernel<3, 128, true, true, 2, 4, false>(const PgemmArgs);
template __global__ void pgemm_nt_kernel<3, 128, false, true, 2, 4, false>(const PgemmArgs);
template __global__ void pgemm_nt_kernel<3, 128, true, true, 4, 2, false, false, true>(const PgemmArgs);
template __global__ void pgemm_nt_kernel<3, 128, false, true, 4, 2, false, false, true>(const PgemmArgs);
}  // namespace msn
