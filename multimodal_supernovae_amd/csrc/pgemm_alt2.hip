// Instantiations of the plane GEMM kernel templates (pgemm_kernels.h) that csrc/pgemm.hip launches but does not compile itself:
// the 2-plane bf16 forms (bf16x3p) and the fp16 forms (msn_pgemm_nt_f16 / _tn_f16).  A translation unit of their own so that they compile beside pgemm.hip.
#include "pgemm_kernels.h"

namespace msn {
template __global__ void pgemm_nt_kernel<2, 128, true, false, 2, 4, false>(const PgemmArgs);
template __global__ void pgemm_nt_kernel<2, 128, false, false, 2, 4, false>(const PgemmArgs);
template __global__ void pgemm_nt_kernel<2, 128, false, true, 4, 2, false, true>(const PgemmArgs);
template __global__ void pgemm_tn_kernel<2, 128, true, false, 2, 4>(const PgemmArgs);
template __global__ void pgemm_tn_kernel<2, 128, false, false, 2, 4>(const PgemmArgs);
template __global__ void pgemm_tn_kernel<2, 128, true, true, 2, 4, true>(const PgemmArgs);
template __global__ void pgemm_tn_kernel<2, 128, false, true, 2, 4, true>(const PgemmArgs);
}  // namespace msn
