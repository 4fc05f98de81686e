/* libmsn_hip.so -- C-ABI of the MI355X (gfx950) contrastive hot path.
 *
 * Drop-in boundary for the training step of ThomasHelfer/multimodal-supernovae:
 * encoders -> projection -> L2 normalise -> all-pairs similarity -> symmetric InfoNCE
 * -> backward -> RAdam.  The reference has no FFI of its own (it is pure PyTorch); each
 * entry point below replaces the stock-ATen op sequence of the cited reference lines
 * (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success, MSN_ERR_SHAPE for a rejected argument (nothing
 *    is launched) or MSN_ERR_HIP for a failed launch; msn_last_error() gives the text;
 *  - all pointers are DEVICE pointers to fp32 unless stated (masks: 1 byte / element);
 *    matrices are row-major with explicit leading dimensions where strided use is allowed;
 *  - no allocation, no synchronisation: kernels are enqueued on `stream` (a hipStream_t);
 *    scratch memory is caller-owned (`ws`, sized by the matching *_workspace_bytes);
 *  - results are deterministic (no floating-point atomics anywhere).
 */
#ifndef MSN_HIP_H
#define MSN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSN_OK 0
#define MSN_ERR_SHAPE 1
#define MSN_ERR_HIP 2

typedef void* msn_stream_t; /* hipStream_t */

int msn_version(void);
const char* msn_last_error(void);
/* number of visible HIP devices, or -1 when the HIP runtime cannot be initialised */
int msn_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 * Replaces nn.Linear / 1x1-conv / patch-conv forward and both of their backward products:
 *   src/transformer_utils.py:45-47,89 (q/k/v, unifyheads), :102-106 (ff), :251 (projection),
 *   src/models_multimodal.py:54-56,75,85-88 (ConvMixer convs / head), :853-856 (MLP), :277 etc.
 *
 *   C[M,N] = epilogue( opA(A)[M,K] . opB(B)[K,N] + bias[N] )
 *
 * opA = MSN_OP_N: A stored [M][lda>=K];  MSN_OP_T: A stored [K][lda>=M].
 * opB = MSN_OP_N: B stored [K][ldb>=N];  MSN_OP_T: B stored [N][ldb>=K].
 *   forward  y = x W^T + b : (N, T);   dgrad dx = dy W : (N, N);   wgrad dW = dy^T x : (T, N).
 * epilogue:
 *   MSN_EPI_NONE      C = acc + bias
 *   MSN_EPI_RELU      C = max(acc + bias, 0)
 *   MSN_EPI_GELU      C = gelu(acc + bias) (erf form); if aux != NULL also aux = acc + bias
 *   MSN_EPI_RELU_BWD  C = acc * (aux > 0)            (aux = forward ReLU output)
 *   MSN_EPI_GELU_BWD  C = acc * gelu'(aux)           (aux = forward pre-activation)
 *   MSN_EPI_ADD       C = acc + bias + aux           (residual add)
 * bias may be NULL.  `ws` is only needed for opA = T (split-K wgrad; see workspace_bytes).
 */
#define MSN_OP_N 0
#define MSN_OP_T 1
#define MSN_EPI_NONE 0
#define MSN_EPI_RELU 1
#define MSN_EPI_GELU 2
#define MSN_EPI_RELU_BWD 3
#define MSN_EPI_GELU_BWD 4
#define MSN_EPI_ADD 5

size_t msn_sgemm_workspace_bytes(int opA, int opB, int64_t M, int64_t N, int64_t K);
int msn_sgemm(int opA, int opB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
              const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int epilogue,
              float* aux, int64_t ldaux, void* ws, size_t ws_bytes, msn_stream_t stream);

/* out[n] = sum_m X[m][n]  (bias gradients).  ws >= msn_colsum_workspace_bytes(M, N). */
size_t msn_colsum_workspace_bytes(int64_t M, int64_t N);
int msn_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes,
               msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused symmetric InfoNCE ("clip_loss") for one modality pair -- src/loss.py:14-38.
 * The N x N logit matrix S = exp(log_scale) * E2 . E1^T + bias is never materialised.
 *
 * Row-sharded over ranks: this rank owns rows [q_offset, q_offset + b) of both modalities
 * (E1_loc (b1 x D), E2_loc (b2 x D)) and holds the all-gathered matrices E1_all (n1 x D),
 * E2_all (n2 x D).  Single process: *_loc == *_all, q_offset = 0.  n = min(n1, n2) (:31).
 *   fwd: lse_row[i] = LSE_j S[q_offset+i][j]   (b2 values; rows of S come from E2)
 *        lse_col[i] = LSE_j S[j][q_offset+i]   (b1 values; columns of S come from E1)
 *        loss[0]    = 1/(2n) * sum_{local i, q_offset+i < n} (lse_row + lse_col - 2 S_ii)
 *                     (this rank's share; the total loss is the sum over ranks)
 *   bwd: needs the LSE vectors of ALL rows (all-gathered: lse_row_all (n2), lse_col_all (n1))
 *        and the upstream gradient `grad_out` (device scalar).  Writes dE1_loc, dE2_loc (the
 *        complete gradient of the global loss w.r.t. the local rows -- no reduce-scatter) and
 *        dscale_dbias[2] = this rank's share of d/dlog_scale and d/dbias (sum over ranks).
 * log_scale, bias, grad_out are DEVICE scalars (no host synchronisation).  D in {8,16,32,64,128}.
 */
size_t msn_infonce_workspace_bytes(int b1, int b2, int n1, int n2, int D);
int msn_infonce_fwd(const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                    const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2,
                    int D, int q_offset, const float* log_scale, const float* bias, float* lse_row,
                    float* lse_col, float* loss, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_infonce_bwd(const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                    const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2,
                    int D, int q_offset, const float* log_scale, const float* bias,
                    const float* lse_row_all, const float* lse_col_all, const float* grad_out,
                    float* dE1_loc, int64_t ldd1, float* dE2_loc, int64_t ldd2, float* dscale_dbias,
                    void* ws, size_t ws_bytes, msn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSN_HIP_H */
