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
 *  - the msn_set_* entry points are process-wide measurement / test switches (unsynchronised statics of the library): set them
 *    once, before work is enqueued, from the thread that enqueues it -- they are not thread-safe (INTEGRATION.md, round-4 notes);
 *    the data path keeps no mutable global state besides the mutex-protected arrival-counter slices of the work-list kernels;
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
/* Measurement: one wave that stamps the shader-cycle and the 100-MHz real-time counters `microseconds` apart on `stream`:
 * out2[0] = shader cycles, out2[1] = real-time ticks (device memory) -> the clock the chip holds while whatever else runs meanwhile
 * (bench.py launches it on a side stream beside a training step: roofline.shader_clock_ghz). */
int msn_clock_probe(unsigned long long* out2, int microseconds, msn_stream_t stream);
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
 *   MSN_EPI_GELU      C = gelu(acc + bias) (erf form); if aux != NULL also aux = gelu'(acc + bias)
 *   MSN_EPI_RELU_BWD  C = acc * (aux > 0)            (aux = forward ReLU output)
 *   MSN_EPI_GELU_BWD  C = acc * aux                  (aux = the gelu' saved by the forward epilogue)
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
/* precision of the inner products (inputs / outputs are fp32 in every mode):
 *   MSN_PREC_F32     exact fp32 on v_mfma_f32_32x32x2_f32 (default; 157 TFLOP/s peak)
 *   MSN_PREC_BF16X3  each operand split hi + lo into two bf16, three v_mfma_f32_32x32x16_bf16 products with
 *                    fp32 accumulation: ~1e-5 relative error per product, 5.3x the fp32 matrix rate
 *   MSN_PREC_BF16    operands rounded to bf16, one product (BASELINE cfg5 "bf16 on MFMA") */
#define MSN_PREC_F32 0
#define MSN_PREC_BF16X3 1
#define MSN_PREC_BF16 2

size_t msn_sgemm_workspace_bytes(int opA, int opB, int64_t M, int64_t N, int64_t K);
/* Two fp32 kernel families sit behind msn_sgemm.
 *   mode 0: register-staged global->LDS copies with a distance-1 prefetch; takes every shape.
 *   mode 3 (default): LDS-DMA pipeline (global_load_lds, no staging registers, no ds_write), 4 waves per
 *           workgroup, 2 x 32-deep K ring, two workgroups per CU.  Used where it applies (16-B aligned
 *           operands, K % 32 == 0); other shapes take mode 0.
 *   mode 1 / 2: LDS-DMA with 8 waves per workgroup and a 3 x 32 / 2 x 64 deep ring (one workgroup per CU).
 *   mode 4: persistent workgroups of 4 multiplying + 2 loader waves (the loaders issue the ring's LDS-DMA for the
 *           whole sequence of tiles, so epilogue stores never sit in front of an operand wait); 128-wide tiles only.
 *           Measured equal to mode 3 on long K and slower on K <= 384: kept as an alternative.
 * All modes give bit-identical results (same k order per accumulator).  Process-wide. */
int msn_set_gemm_variant(int mode);
/* Tail split (default on): a product whose 128 x 128 tiles do not fill a whole number of rounds of the chip's
 * 512 resident workgroups has the tiles of the last, partly filled round cut into K-slabs; msn_sgemm_workspace_bytes
 * covers the slabs.  enabled = 1 (default): the workgroup that stores a tile's last slab (device-scope arrival
 * counter in module memory, one slice per stream) sums the slabs in slab order and applies the epilogue inside the
 * same launch; 2: a finishing launch does (bit-identical: same order); 0: no slabs.  Results of the tail tiles
 * differ from the unsplit order in the last bits.  Process-wide. */
int msn_set_gemm_tail_split(int enabled);
/* Measurement switch: tile width of products with N > 64: 0 = planned (default), 64, 128. */
int msn_set_gemm_tile_n(int bn);
int msn_sgemm(int opA, int opB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
              const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int epilogue,
              float* aux, int64_t ldaux, int precision, void* ws, size_t ws_bytes, msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Work-list launch: up to 3 independent products in ONE persistent launch (gemm_list.hip).  The (tile, K-step) pairs
 * of all products form one sequence that is cut into equal ranges, one per resident workgroup; a tile cut by a range
 * boundary is finished by the last of its contributors to arrive (slabs of raw accumulators summed in k order:
 * deterministic).  Built for the 128 - 512 rows per GPU that strong scaling of the global batch leaves a rank, where a
 * single product cannot fill the chip's 512 workgroup slots, and for the two backward products of a Linear
 * (src/transformer_utils.py:45-47,89,102-106: dX = dY . W and dW = dY^T . X share dY and are independent) -- the weight
 * gradient needs no split count and no reduction launch.  `colsum` (opA = T only, may be NULL): also out[M] = column
 * sums of A, i.e. the bias gradient of the same Linear.  Products the kernel does not take (N or M <= 64, K % 32 != 0,
 * unaligned operands, precision != fp32) are issued one by one through msn_sgemm / msn_wgrad_bias -- same results as
 * calling those.  Against the one-by-one path the list kernel's results differ in the last bits (another k order).
 * msn_sgemm itself takes the list kernel for an opA = N product of at most 1024 128 x 128 tiles and K >= 1024 whose last round of
 * workgroups is under-filled (the balance gained outweighs the slabs of the cut tiles).
 * msn_set_gemm_list(mode): 1 = all of the above (default); 0 = no work-list launch at all (lists one by one; measurements, bit
 * comparisons); 2 = lists only, single products never; 3 = lists + every single product the kernel can take (tests).  Process-wide. */
typedef struct msn_gemm_desc {
    int opA, opB;
    int64_t M, N, K;
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    float* C;
    int64_t ldc;
    const float* bias;
    int epilogue;
    float* aux;
    int64_t ldaux;
    float* colsum;
} msn_gemm_desc;
size_t msn_sgemm_list_workspace_bytes(int n, const msn_gemm_desc* products);
int msn_sgemm_list(int n, const msn_gemm_desc* products, int precision, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_set_gemm_list(int mode);
/* Zero the arrival counters of the in-kernel tile finishes (tail split, work-list launch) of the current device: they
 * are zero between launches by construction, but a launch that faulted or was aborted half-way leaves them dirty. */
int msn_reset_gemm_counters(msn_stream_t stream);

/* Weight + bias gradient of a Linear in one launch (the backward of torch.nn.functional.linear as used at
 * ref src/transformer_utils.py:33-36, :103-107 and every nn.Linear of src/models_multimodal.py):
 *   dW[M x N] = dY^T X,   db[M] = column sums of dY,   dY: K x M (row stride lddy), X: K x N (row stride ldx).
 * On the LDS-DMA kernels the column sums ride on the A fragments the product already holds (no second pass over
 * dY); other shapes / precisions run msn_sgemm + msn_colsum.  Workspace: msn_wgrad_bias_workspace_bytes. */
size_t msn_wgrad_bias_workspace_bytes(int64_t M, int64_t N, int64_t K);
int msn_wgrad_bias(int64_t M, int64_t N, int64_t K, const float* dY, int64_t lddy, const float* X, int64_t ldx,
                   float* dW, int64_t lddw, float* db, int precision, void* ws, size_t ws_bytes,
                   msn_stream_t stream);

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
 * log_scale, bias, grad_out are DEVICE scalars (no host synchronisation).  Any 1 <= D <= 256 (the reference takes
 * any enc_dim, src/models_multimodal.py:101): tiles are 8/16/32/64/128/256 columns wide, zero beyond column D.
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

/* SigLIP-style loss with the reference's sign convention -- sigmoid_loss, src/loss.py:68-83:
 *   Z = -(E2 . E1^T) exp(log_scale) + bias (fp32);  loss = mean_ij softplus(z_ij Z_ij) evaluated in fp64,
 *   z = +1 on the diagonal, -1 elsewhere (mean over all n x n entries).  Same row-sharded calling
 *   convention and workspace as msn_infonce_*; both modalities have b local / n global rows.
 *   fwd writes this rank's share of the loss; bwd writes dE1_loc, dE2_loc and (dlog_scale, dbias) shares. */
int msn_sigmoid_loss_fwd(const float* E1_loc, int64_t ld1, const float* E2_loc, int64_t ld2, int b,
                         const float* E1_all, int64_t ld1a, const float* E2_all, int64_t ld2a, int n, int D,
                         int q_offset, const float* log_scale, const float* bias, float* loss, void* ws,
                         size_t ws_bytes, msn_stream_t stream);
int msn_sigmoid_loss_bwd(const float* E1_loc, int64_t ld1, const float* E2_loc, int64_t ld2, int b,
                         const float* E1_all, int64_t ld1a, const float* E2_all, int64_t ld2a, int n, int D,
                         int q_offset, const float* log_scale, const float* bias, const float* grad_out,
                         float* dE1_loc, int64_t ldd1, float* dE2_loc, int64_t ldd2, float* dscale_dbias,
                         void* ws, size_t ws_bytes, msn_stream_t stream);

/* Retrieval rank for the validation "AUC" -- get_ROC_data, src/utils.py:380-411: for unit-norm rows,
 * rank[i] = #{ j != i : <E2_i, E1_j> > <E2_i, E1_i> } (position of the true partner in row i's similarity
 * ranking; the reference sorts every row in a host loop).  workspace: msn_infonce_workspace_bytes(n,n,n,n,D). */
int msn_retrieval_rank(const float* E1, int64_t ld1, const float* E2, int64_t ld2, int n, int D, int* rank,
                       void* ws, size_t ws_bytes, msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (rows x cols, cols % 4 == 0, cols <= 1024), eps inside the
 * square root -- nn.LayerNorm as used at src/transformer_utils.py:97-98,111,114 (post-norm; the
 * residual sum is produced by the preceding GEMM's MSN_EPI_ADD epilogue).
 * fwd writes y and the per-row mean / rstd; bwd consumes them and returns dx (+ `add`, the
 * gradient of a residual branch around the norm, when non-NULL), dgamma, dbeta.
 */
int msn_layernorm_fwd(const float* x, int64_t ldx, int64_t rows, int cols, const float* gamma,
                      const float* beta, float eps, float* y, int64_t ldy, float* mean, float* rstd,
                      msn_stream_t stream);
size_t msn_layernorm_bwd_workspace_bytes(int64_t rows, int cols);
int msn_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int cols,
                      const float* mean, const float* rstd, const float* gamma, const float* add,
                      int64_t ldadd, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws,
                      size_t ws_bytes, msn_stream_t stream);

/* y = x / ||x||_2 per row, NO epsilon (src/models_multimodal.py:279,286,293,304); inv_norm[r] = 1/||x_r||.
 * bwd: dx = inv_norm * (dy - y <y, dy>). */
int msn_l2norm_fwd(const float* x, int64_t ldx, int64_t rows, int cols, float* y, int64_t ldy,
                   float* inv_norm, msn_stream_t stream);
int msn_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, int64_t rows, int cols,
                   const float* inv_norm, float* dx, int64_t lddx, msn_stream_t stream);

/* Token embedding of an irregular time series -- src/transformer_utils.py:166-176 and :214-231:
 *   out[b,t,c] = x[b,t] * w[c] + bw[c] + (c even ? sin : cos)(t[b,t] * omega[c/2]) + band[t / (T/nband)][c]
 * omega[k] = exp(-2k ln(norm) / e) is passed in (e/2 floats); band (nband x e) only when nband > 1.
 * bwd returns dw, dbw (e each) and dband (nband x e; NULL when nband == 1); x, t get no gradient. */
int msn_time_embed_fwd(const float* x, const float* t, int64_t B, int T, int e, const float* w,
                       const float* bw, const float* omega, const float* band, int nband, float* out,
                       msn_stream_t stream);
size_t msn_time_embed_bwd_workspace_bytes(int64_t B, int e, int nband);
int msn_time_embed_bwd(const float* dy, const float* x, int64_t B, int T, int e, int nband, float* dw,
                       float* dbw, float* dband, void* ws, size_t ws_bytes, msn_stream_t stream);

/* Masked pooling over tokens -- src/transformer_utils.py:234-239.  mask: (B, T) bytes (torch.bool).
 * mode MSN_POOL_MEAN: sum_t x*mask / sum_t mask (writes count[b]; an all-false row gives NaN as
 * the reference does); MSN_POOL_MAX: max_t (x*mask) over ALL t (writes argmax (B, e) int32). */
#define MSN_POOL_MEAN 0
#define MSN_POOL_MAX 1
int msn_masked_pool_fwd(const float* x, const uint8_t* mask, int64_t B, int T, int e, int mode, float* out,
                        int* argmax, float* count, msn_stream_t stream);
int msn_masked_pool_bwd(const float* dout, const uint8_t* mask, int64_t B, int T, int e, int mode,
                        const int* argmax, const float* count, float* dx, msn_stream_t stream);
/* y[r,:] = x[r,:] * mask[r]  (the `x * mask[:, :, None]` of :235; its own backward) */
int msn_mask_tokens(const float* x, const uint8_t* mask, int64_t rows, int e, float* y, msn_stream_t stream);

/* dst[r][0..cols) += src[r][0..cols) for r < rows, both row-strided (cols, ldd, lds multiples of 4; 16-byte aligned).
 * The class-token read-out of the build-defined ViT: only token 0 of the last block feeds the head, so that block's
 * attention output / MLP run on the B class rows alone and their input gradient is added back into row 0 of every
 * sample of the dense (B, T, e) gradient (no reference counterpart: the ViT is build-defined). */
int msn_add_rows(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t rows, int cols, msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32-GRADE products on the bf16 matrix cores from resident bf16 PLANES (pgemm.hip) -- the same nn.Linear products
 * msn_sgemm multiplies (ref src/transformer_utils.py:45-47, 89, 102-106, 251; src/models_multimodal.py:277), for the wide
 * layers of the build-defined ViT towers.  An fp32 matrix is kept in HBM as `planes` bf16 planes, x = p0 + p1 (+ p2) with
 * p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1): three planes hold an fp32 value EXACTLY (3 x 8 significand
 * bits, fp32 exponent range).  A product sums the plane products with pa + pb < planes on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation: planes = 3 -> 6 MFMA products, terms dropped <= 2^-26 |a||b| (fp32 grade); planes = 2 -> 3 products,
 * dropped <= 2^-17 |a||b|.
 * Plane matrix of a logical R x C matrix ("blocked planes"): 32-row x 16-column blocks, block (rb, cb) = `planes`
 * consecutive 1-KB images [32][16] bf16; byte offset of (r, c, plane) = (((r / 32) * CB + c / 16) * planes + plane) * 1024
 * + (r % 32) * 32 + (c % 16) * 2, CB = 2 * ceil(C / 32); rows / columns past R / C are zero.  msn_plane_bytes gives its size.
 *   msn_plane_split:  fp32 (R x C, row stride ldx) -> plane matrix; transposed = 1 writes the planes of the C x R
 *                     transpose (weights for the input-gradient products).  colsum (nullable, untransposed only):
 *                     out[c] = sum_r x[r][c] (bias gradients); ws >= msn_plane_split_colsum_workspace_bytes.
 *   msn_plane_merge:  plane matrix -> fp32 (tests).
 *   msn_pgemm_nt:     C[M][N] = epilogue(A[M][K] . B[N][K]^T + bias); A, B plane matrices (M x K, N x K).  C is fp32
 *                     (row stride ldc) or, with c_planes = 1, the plane matrix of the M x N result (N % 16 == 0) -- the
 *                     operand of the next product, split in the epilogue.  Epilogues as msn_sgemm (MSN_EPI_*; aux fp32).
 *                     colsum_out (nullable): column sums of the values written to C; ws >= msn_pgemm_nt_colsum_workspace_bytes.
 *   msn_pgemm_tn:     C[N][K] (fp32) = sum_m A[m][N]^T . B[m][K] (weight gradient dY^T . X); A, B plane matrices with the
 *                     reduction on the rows (M x N, M x K); reduction split over workgroups, fixed-order slab sums;
 *                     ws >= msn_pgemm_tn_workspace_bytes.
 * Results are deterministic. */
size_t msn_plane_bytes(int64_t R, int64_t C, int planes);
size_t msn_plane_split_colsum_workspace_bytes(int64_t R, int64_t C);
int msn_plane_split(const float* x, int64_t ldx, int64_t R, int64_t C, int planes, int transposed, void* out, float* colsum,
                    void* ws, size_t ws_bytes, msn_stream_t stream);
/* msn_plane_split of several matrices in ONE launch (per 64 items): the weights of every block of a tower, or their
 * transposes.  items: host array; out of item i holds msn_plane_bytes(R, C, planes) bytes (transposed: of (C, R)). */
typedef struct msn_split_item {
    const float* x;
    int64_t ldx, R, C;
    int transposed;
    void* out;
} msn_split_item;
int msn_plane_split_list(int n, const msn_split_item* items, int planes, msn_stream_t stream);
int msn_plane_merge(const void* planes_in, int planes, int64_t R, int64_t C, float* y, int64_t ldy, msn_stream_t stream);
size_t msn_pgemm_nt_colsum_workspace_bytes(int64_t M, int N);
/* Workspace of msn_pgemm_nt: the column sums' partials (want_colsum) or the slabs of the TAIL split -- the tiles that do not
 * fill a round of the 256 persistent workgroups are cut into K-segments, one per workgroup, and a finishing launch sums
 * the segments in K order and applies the epilogue (fp32 outputs, epilogues NONE / RELU / ADD; deterministic).  Without a
 * workspace (NULL / too small) the tail tiles are multiplied whole.  msn_set_pgemm_tail_split(0) turns the split off. */
size_t msn_pgemm_nt_workspace_bytes(int64_t M, int N, int K, int planes, int c_planes, int epilogue, int want_colsum);
int msn_set_pgemm_tail_split(int enabled);
int msn_pgemm_nt(int64_t M, int N, int K, int planes, const void* A, const void* B, void* C, int64_t ldc, int c_planes,
                 const float* bias, int epilogue, float* aux, int64_t ldaux, float* colsum_out, void* ws, size_t ws_bytes,
                 msn_stream_t stream);
size_t msn_pgemm_tn_workspace_bytes(int64_t M, int N, int K, int planes);
int msn_pgemm_tn(int64_t M, int N, int K, int planes, const void* A, const void* B, float* C, int64_t ldc, void* ws,
                 size_t ws_bytes, msn_stream_t stream);
/* nn.LayerNorm as msn_layernorm_fwd / _bwd, writing the result as a plane matrix (the next product's operand): forward y
 * (y_planes; y fp32 optional, may be NULL), backward dx (fp32 AND planes; dx_colsum nullable = column sums of dx, the bias
 * gradient of the Linear below a residual add).  Workspace of the backward: msn_layernorm_bwd_workspace_bytes * 3 / 2. */
int msn_layernorm_fwd_planes(const float* x, int64_t ldx, int64_t rows, int cols, const float* gamma, const float* beta,
                             float eps, int planes, void* y_planes, float* y, int64_t ldy, float* mean, float* rstd,
                             msn_stream_t stream);
/* (With y == NULL and 260..400 columns, from 32 768 rows on, both directions take whole 32-row blocks per workgroup and write each
 * block's plane images as contiguous memory; y, dx and every plane byte are the same as the row-at-a-time kernels', the backward's
 * dgamma / dbeta / dx_colsum the same terms summed in another fixed order.) */
int msn_layernorm_bwd_planes(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int cols,
                             const float* mean, const float* rstd, const float* gamma, const float* add, int64_t ldadd,
                             float* dx, int64_t lddx, int planes, void* dx_planes, float* dgamma, float* dbeta,
                             float* dx_colsum, void* ws, size_t ws_bytes, msn_stream_t stream);
/* Self-attention forward over a packed qkv matrix (rows = (sample, token), columns q | k | v; ref src/transformer_utils.py:36-89)
 * whose output is written twice by the one kernel: `out` (fp32, (B T) x (H head_dim), row stride ldo -- the backward reads it) and
 * `out_planes` (msn_plane_bytes(B T, H head_dim, planes) bytes, 16-byte aligned): the operand of the output projection, bit for bit
 * what msn_plane_split(out) would write.  T <= 128 tokens, head_dim in {16, 32, 48, 64}, 16-byte aligned rows with strides % 4 == 0;
 * lse as msn_attention_fwd ([B][H][T][2]).  MSN_ERR_SHAPE otherwise (the caller runs msn_attention_fwd + msn_plane_split). */
int msn_attention_fwd_planes(const float* qkv, int64_t ldqkv, const uint8_t* key_mask, int B, int H, int T, int head_dim,
                             float scale, float* out, int64_t ldo, float* lse, int planes, void* out_planes, msn_stream_t stream);
/* Backward of the ViT blocks' self-attention (msn_attention_bwd on the packed q | k | v matrix of msn_pgemm_nt's qkv
 * product: qkv (B T x ldqkv >= 3 H hd), out / dout (B T x H hd), lse as msn_attention_fwd wrote it), writing the gradient
 * dqkv (B T x 3 H hd) as a PLANE matrix -- the operand of the two products that consume it -- and, colsum_out non-NULL,
 * its column sums (the bias gradient of the qkv projection; workspace msn_attention_bwd_planes_workspace_bytes).  One
 * launch, one pass over the operands (see msn_set_attention_fused).  T <= 128, hd in {16, 32, 48, 64}, key_mask (B, T)
 * bytes or NULL.  Replaces msn_attention_bwd + msn_plane_split for the build-defined ViT (no reference counterpart:
 * the reference's image ViT is torch's nn.MultiheadAttention inside its build-defined encoder). */
size_t msn_attention_bwd_planes_workspace_bytes(int B, int H, int head_dim);
int msn_attention_bwd_planes(const float* qkv, int64_t ldqkv, const uint8_t* key_mask, int B, int H, int T, int head_dim,
                             float scale, const float* out, int64_t ldo, const float* lse, const float* dout, int64_t ldd,
                             int planes, void* dqkv_planes, float* colsum_out, void* ws, size_t ws_bytes,
                             msn_stream_t stream);
/* ---- the same products from TWO fp16 planes per operand (3 MFMA products instead of 6): x 2^e = h0 + h1 carries 22 significand
 * bits (bf16 planes: 8 per plane, so fp32 grade takes three), at half the matrix work.  fp16's exponent range is what this
 * costs: every plane matrix has a power-of-two scale chosen from its largest magnitude (msn_plane_split_f16 makes a pass for
 * it: m 2^e in [2^13, 2^14)); `scale` is two device floats: [0] = 2^-e, which the products multiply into their sums (exact),
 * [1] = the bits of m.  reuse_scale = 1 splits with the scale already in `scale` (a weight matrix and its transpose share one).
 * Products: v_mfma_f32_32x32x16_f16, two accumulator sets, K chunks as msn_pgemm_nt; fp32 results only.  Measured against fp64
 * (tests/test_pgemm_gpu.py::test_fp32_grade_gate_*[f16x3-*]): 0.3 - 0.6 x the native fp32 kernel's error on normal, wide-range,
 * tiny and huge data, but 2.2 - 2.7 x on cancellation-heavy data (22-bit operands) -- NOT fp32 grade by the 1.5 x gate, so an
 * opt-in of this interface only; 1.4 - 1.6 x the rate of the six-product form (profiles/r04_pgemm_f16.txt).  Layout of the
 * plane matrix: as msn_plane_split with planes = 2. */
int msn_plane_split_f16(const float* x, int64_t ldx, int64_t R, int64_t C, int transposed, void* out, float* scale,
                        int reuse_scale, float* colsum, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_pgemm_nt_f16(int64_t M, int N, int K, const void* A, const float* scaleA, const void* B, const float* scaleB, float* C,
                     int64_t ldc, const float* bias, int epilogue, float* aux, int64_t ldaux, float* colsum_out, void* ws,
                     size_t ws_bytes, msn_stream_t stream);   /* workspace: msn_pgemm_nt_workspace_bytes(.., planes 2, ..) */
int msn_pgemm_tn_f16(int64_t M, int N, int K, const void* A, const float* scaleA, const void* B, const float* scaleB, float* C,
                     int64_t ldc, void* ws, size_t ws_bytes, msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The feed-forward half of the reference's TransformerBlock for the NARROW towers -- z = x + Linear(4e -> e)(ReLU(Linear(e -> 4e)(x))),
 * ref src/transformer_utils.py:102-106, 114; emb 32 / hidden 128 (the spectrum transformer) -- as ONE kernel per direction whose
 * 4e-wide hidden activations never reach HBM (csrc/ffn_planes.hip): fp32-grade arithmetic on the bf16 matrix cores (three planes per
 * operand, six products, as msn_pgemm_*), both weights resident in LDS as the plane matrices msn_plane_split writes:
 *   w1_planes  = planes of ff.0.weight [4e][e];  w2t_planes = planes of ff.2.weight TRANSPOSED [4e][e] (msn_plane_split, transposed = 1).
 *   msn_ffn_fwd:  z[M][e] = x + relu(x W1^T + c1) W2^T + c2.
 *   msn_ffn_bwd:  dx = dz + ((dz W2) o (pre > 0)) W1 (the hidden tile is RECOMPUTED with the forward's instructions: the same bits, no
 *                 mask is stored), dw1 [4e][e], dc1 [4e], dw2 [e][4e], dc2 [e] (per-workgroup partials in ws, summed in a fixed order
 *                 by a finishing launch: deterministic); ws >= msn_ffn_bwd_workspace_bytes.
 * msn_ffn_supported(M, emb, hidden): the shapes this build takes (emb 32, hidden 128); every other block keeps msn_sgemm. */
int msn_ffn_supported(int64_t M, int emb, int hidden);
int msn_ffn_fwd(const float* x, int64_t ldx, int64_t M, int emb, int hidden, const void* w1_planes, const void* w2t_planes,
                const float* c1, const float* c2, float* z, int64_t ldz, msn_stream_t stream);
size_t msn_ffn_bwd_workspace_bytes(int64_t M, int emb, int hidden);
int msn_ffn_bwd(const float* x, int64_t ldx, const float* dz, int64_t lddz, int64_t M, int emb, int hidden, const void* w1_planes,
                const void* w2t_planes, const float* c1, float* dx, int64_t lddx, float* dw1, float* dc1, float* dw2, float* dc2,
                void* ws, size_t ws_bytes, msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * bf16-RESIDENT products for BASELINE.json configs[4] (ViT-B/16 "bf16 on MFMA"; build-defined encoder, no reference
 * counterpart -- the reference has no mixed precision): operands are bf16 in HBM (uint16 bit patterns), accumulation fp32.
 *   msn_bgemm_nt:  C[M][N] = epi(A[M][K] . B[N][K]^T + bias); K % 64 == 0, N % 4 == 0, 16-byte aligned rows.
 *                  epilogue 0 none (C fp32 or bf16), 1 GELU (aux <- bf16 pre-activation, C <- bf16 gelu), 2 GELU' (C <-
 *                  acc * gelu'(aux bf16); C fp32 or bf16), 3 ADD (C fp32 <- acc + bias + aux fp32).
 *   msn_bgemm_tn:  C[N][K] (fp32) = sum_m A[m][N]^T . B[m][K]  (weight gradient dY^T . X), reduction split over m with
 *                  fixed-order slab sums; ws >= msn_bgemm_tn_workspace_bytes.
 *   msn_cast_bf16 / msn_cast_bf16_transposed: y = bf16(x) (n % 8 == 0) / y[c][r] = bf16(x[r][c]).
 *   msn_bcolsum:   out[n] = sum_m X[m][n] for bf16 X (bias gradients).
 *   msn_layernorm_fwd_bf16 / msn_layernorm_bwd_bf16: nn.LayerNorm as msn_layernorm_fwd / _bwd, writing y as bf16 /
 *                  writing a bf16 copy of dx next to the fp32 one. */
int msn_bgemm_nt(int64_t M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                 int c_bf16, const float* bias, int epilogue, void* aux, int64_t ldaux, float* colsum_out, void* ws,
                 size_t ws_bytes, msn_stream_t stream);
  /* one persistent workgroup per CU walks the 256 x 256 tiles and keeps its LDS ring of K-tiles running across tile boundaries (one
     workgroup per tile below 257 tiles).  colsum_out (nullable): out[n] = sum_m C[m][n] from the epilogue (the
                 bias gradient of the Linear whose output gradient C is); ws >= msn_bgemm_nt_colsum_workspace_bytes(M, N) */
size_t msn_bgemm_nt_colsum_workspace_bytes(int64_t M, int N);
size_t msn_bgemm_tn_workspace_bytes(int64_t M, int N, int K);
int msn_bgemm_tn(int64_t M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc,
                 void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_cast_bf16(const float* x, int64_t n, void* y, msn_stream_t stream);
int msn_cast_bf16_transposed(const float* x, int R, int C, void* y, msn_stream_t stream);
/* msn_cast_bf16 / msn_cast_bf16_transposed of a LIST of contiguous (R, C) fp32 matrices in ONE launch (the bf16-resident trunk: four
 * weights per block forward, four transposed copies backward): y (R, C) or, transposed != 0, (C, R), bf16, round to nearest even. */
typedef struct msn_cast_item {
    const float* x;
    int64_t R, C;
    int transposed;
    void* y;
} msn_cast_item;
int msn_cast_bf16_list(int n, const msn_cast_item* items, msn_stream_t stream);
size_t msn_bcolsum_workspace_bytes(int64_t M, int N);
int msn_bcolsum(const void* X, int64_t ldx, int64_t M, int N, float* out, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_layernorm_fwd_bf16(const float* x, int64_t ldx, int64_t rows, int cols, const float* gamma, const float* beta,
                           float eps, void* y_bf16, int64_t ldy, float* mean, float* rstd, msn_stream_t stream);
/* Self-attention on the bf16 matrix cores, 64-wide heads, T <= 256 tokens, no key mask (the cfg5 ViT-B/16 attention;
 * build-defined, scale 1/sqrt(head_dim)): qkv is the (B*T, ld >= 3*H*64) bf16 matrix [q | k | v] the packed projection
 * writes, out / dout (B*T, H*64) bf16, lse (B, H, T) fp32; bwd writes dqkv in qkv's layout (bf16); delta = (B, H, T) scratch. */
int msn_attention_bf16_fwd(const void* qkv, int64_t ld, int B, int H, int T, float scale, void* out, int64_t ldo, float* lse,
                           msn_stream_t stream);
int msn_attention_bf16_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t ldd,
                           const float* lse, int B, int H, int T, float scale, void* dqkv, float* delta, float* colsum_out,
                           float* colsum_ws, msn_stream_t stream);   /* colsum_out (3*H*64, nullable): column sums of dqkv = the
                           bias gradient of the packed projection; colsum_ws: B x 3*H*64 floats */
int msn_layernorm_bwd_bf16(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int cols,
                           const float* mean, const float* rstd, const float* gamma, const float* add, int64_t ldadd,
                           float* dx, int64_t lddx, void* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, int dy_is_bf16,
                           void* ws, size_t ws_bytes, msn_stream_t stream);   /* dx_colsum (nullable): column sums of dx = the bias
                           gradient of the Linear feeding this LayerNorm's residual add; ws >= 1.5 x msn_layernorm_bwd_workspace_bytes
                           then.  dy_is_bf16: dy points at bf16 values, lddy in bf16 elements (the input-gradient product of cfg5
                           writes its result as bf16: half the bytes out of the GEMM epilogue and into this kernel) */

/* ------------------------------------------------------------------------------------------
 * Fused multi-head attention, exact fp32, no T x T tensor in memory -- SelfAttention.forward,
 * src/transformer_utils.py:36-89 (scale = 1/sqrt(emb); key-padding scores REPLACED by -1e7), also
 * the 1-query nn.MultiheadAttention pooling (:240-246) and the build-defined ViT blocks.
 * q: (B, Tq, H*hd) rows `ldq` apart, batches `q_bstride` apart (0 = one query shared by the batch);
 * k, v: (B, Tk, H*hd); head h lives in columns h*hd .. h*hd+hd-1 (so a packed q|k|v buffer works by
 * pointer offset + ld = 3*H*hd).  key_mask: (B, Tk) bytes or NULL.  hd <= 128.
 * lse: (B, H, Tq, 2) = (row max, log of the exp-sum), kept apart because a fully padded sample has
 * max = -1e7 where fp32 cannot hold the sum.  bwd recomputes the probabilities from lse; `delta` is
 * (B, H, Tq) scratch.
 */
int msn_attention_fwd(const float* q, int64_t ldq, int64_t q_bstride, const float* k, int64_t ldk,
                      int64_t k_bstride, const float* v, int64_t ldv, int64_t v_bstride,
                      const uint8_t* key_mask, int B, int H, int Tq, int Tk, int head_dim, float scale,
                      float* out, int64_t ldo, int64_t o_bstride, float* lse, msn_stream_t stream);
int msn_attention_bwd(const float* q, int64_t ldq, int64_t q_bstride, const float* k, int64_t ldk,
                      int64_t k_bstride, const float* v, int64_t ldv, int64_t v_bstride,
                      const uint8_t* key_mask, int B, int H, int Tq, int Tk, int head_dim, float scale,
                      const float* out, int64_t ldo, int64_t o_bstride, const float* lse,
                      const float* dout, int64_t ldd, int64_t d_bstride, float* delta, float* dq,
                      int64_t lddq, int64_t dq_bstride, float* dk, int64_t lddk, int64_t dk_bstride,
                      float* dv, int64_t lddv, int64_t dv_bstride, msn_stream_t stream);
/* Attention of ONE query per (sample, head) over T keys: the class-token row of the LAST block of the build-defined ViT (the head
 * reads token 0 only; the arithmetic is SelfAttention of ref src/transformer_utils.py:36-89 for a single query, no mask).
 * q / out / dout / dq: (B, H * 64) fp32 rows; kv / dkv: (B * T, >= 2 * H * 64) rows, keys in columns [0, H 64), values in
 * [H 64, 2 H 64), fp32 (kv_bf16 = 0) or bf16 (1: the bf16-resident tower; dkv is written in the same type, round to nearest
 * even); probs: (B, H, T) fp32, written by the forward and read by the backward.  A wave per pair, eight lanes on a key's 64
 * columns, every K and V row read once per direction (csrc/cls_attention.hip).  head_dim must be 64 and T <= 256
 * (msn_cls_attention_supported); every row 16-byte aligned.  Everything else stays with msn_attention_fwd / _bwd, which run this
 * shape as a 16-row query tile with fifteen rows of padding. */
int msn_cls_attention_supported(int T, int head_dim);
int msn_cls_attention_fwd(const float* q, int64_t ldq, const void* kv, int64_t ldkv, int kv_bf16, int B, int H, int T, int head_dim,
                          float scale, float* out, int64_t ldo, float* probs, msn_stream_t stream);
int msn_cls_attention_bwd(const float* q, int64_t ldq, const void* kv, int64_t ldkv, int kv_bf16, int B, int H, int T, int head_dim,
                          float scale, const float* out, int64_t ldo, const float* probs, const float* dout, int64_t ldd, float* dq,
                          int64_t lddq, void* dkv, int64_t lddkv, msn_stream_t stream);
/* Two implementations sit behind msn_attention_*: vector-ALU kernels (head widths up to 32, any length; the
 * reference-native 8-wide heads, widths that are not a multiple of 4, unaligned operands) and matrix-core
 * kernels (v_mfma_f32_16x16x4_f32; any length -- chunked beyond 128 tokens; every head width that is a
 * multiple of 4 up to 128, run as the next multiple of 16 with zero columns: the ViT towers, the
 * reference's 16-wide spectrum heads -- beyond 128 tokens on the bf16 planes, msn_set_attention_planes --
 * and its default 128-wide heads (emb 256 / 2 heads)).  A head WIDER THAN 32 must be a multiple of 4 on
 * 16-byte aligned rows (MSN_ERR_SHAPE otherwise: pad the heads with zero columns -- the Python binding does;
 * the 64- / 128-wide vector-ALU instantiations of rounds 1-4 spilled up to 2 KB per lane and are gone).
 * mode 0 = automatic (default: matrix cores for widths >= 16 where they apply), 1 = vector-ALU wherever it
 * exists (widths up to 32), 2 = matrix cores whenever applicable, narrow heads included (measured no faster
 * there).  Process-wide; meant for tests. */
int msn_set_attention_path(int mode);
/* Self-attention backward over up to 128 tokens with heads up to 64 wide (the ViT towers): 1 (default) = ONE launch that
 * holds Q, K, V and dO of a (sample, head) in LDS together -- one pass over the operands, delta never in memory; 0 = the
 * dQ kernel followed by the dK,dV kernel (same products in the same order); 3 = the one-pass kernel without the shared
 * recomputation (up to 4 full tiles -- the ViT towers -- the default computes every score tile ONCE for dQ, dK and dV: the dQ
 * parts of the four key-tile waves meet in LDS; 80 instead of 112 MFMAs per tile pair).  Process-wide; measurements and tests. */
int msn_set_attention_fused(int on);
/* Sequences of more than 128 tokens with heads up to 16 wide (the reference's spectrum transformer on 1024-bin spectra /
 * 220-step series: emb 32, 2 heads -- src/transformer_utils.py:36-89): fp32-GRADE attention on the bf16 matrix cores
 * (csrc/attention_planes.hip).  Q, K, V, dO are split into three bf16 planes as they are staged into LDS (exact: three 8-bit
 * pieces of the fp32 significand), probabilities and score gradients are split in registers, and every product is the six
 * plane products of the plane GEMMs (v_mfma_f32_16x16x32_bf16; the head-dimension products pack two planes into one
 * instruction's 32 k).  Same arguments, statistics layout and results (to fp32 rounding) as the exact-fp32 matrix-core
 * kernels it replaces; the accuracy gate is tests/test_attention_planes_gpu.py (error against fp64 <= 1.5 x theirs).
 * The backward is ONE pass for 256 or more (sample, head) pairs -- every score tile computed once, dS transposed through LDS for
 * the dQ product, a workgroup per pair walking all key blocks and summing dq over them in place (a small launch computes
 * delta = rowsum(dO o O) first) -- and the dQ kernel followed by the dK,dV kernel below that.
 * mode: 0 = off (the v_mfma_f32_16x16x4_f32 kernels), 1 = on (default), 3 = on with the two-kernel backward everywhere, 5 = on
 * with the one-pass backward everywhere (tests, A/B runs).  Heads narrower than 16 take this path only under
 * msn_set_attention_path(2) (measured equal to the vector-ALU kernels there).
 * Process-wide, not thread-safe (as every msn_set_* switch). */
int msn_set_attention_planes(int mode);

/* ------------------------------------------------------------------------------------------
 * ConvMixer image tower pieces -- src/models_multimodal.py:38-95, channels-last token matrices
 * X[(b, i, j)][c].  The patch conv (stride = kernel = p, no bias, :54-56) and the 1x1 convs (:75)
 * are msn_sgemm on these matrices with the GELU epilogue.
 */
/* img (B, C, H, W) -> patches [(b, i, j)][(c, u, v)], grid = floor(H/p) x floor(W/p) (extra pixels unused) */
int msn_patchify(const float* img, int B, int C, int H, int W, int p, float* patches, msn_stream_t stream);
/* its adjoint: d img from d patches (zeros outside the grid) */
int msn_unpatchify(const float* dpatches, int B, int C, int H, int W, int p, float* dimg, msn_stream_t stream);

/* nn.BatchNorm2d over the rows of x (rows x C), :58,70,77.  training != 0: two-pass batch statistics,
 * `mean` / `rstd` (C each) written, running_mean / running_var updated in place when non-NULL
 * (momentum; unbiased variance) ; training == 0: running statistics are used.  y = bn(x) (+ residual),
 * then ReLU when relu != 0 (build-defined ResNet blocks).
 * bwd: dx = dL/dx, additionally multiplied by dact[.] when dact != NULL (conv -> GELU -> BN order:
 * dact is the gelu' that msn_sgemm's GELU epilogue / msn_dwconv_gelu_fwd saved). */
size_t msn_bn_workspace_bytes(int64_t rows, int C);
int msn_batchnorm_fwd(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                      int training, float momentum, float* running_mean, float* running_var,
                      const float* residual, int relu, float* y, float* mean, float* rstd, void* ws,
                      size_t ws_bytes, msn_stream_t stream);
/* dmasked = dy * (y > 0): gradient through a trailing ReLU (relu != 0 above), given its output y */
int msn_relu_mask(const float* dy, const float* y, int64_t total, float* dmasked, msn_stream_t stream);
int msn_batchnorm_bwd(const float* dy, const float* x, const float* dact, int64_t rows, int C,
                      const float* mean, const float* rstd, const float* gamma, int training, float* dx,
                      float* dgamma, float* dbeta, void* ws, size_t ws_bytes, msn_stream_t stream);
/* The same for BatchNorm + ReLU (y = relu(bn(x)) saved): dy is masked by y > 0 inside both passes. */
int msn_batchnorm_relu_bwd(const float* dy, const float* y, const float* x, int64_t rows, int C, const float* mean,
                           const float* rstd, const float* gamma, int training, float* dx, float* dgamma, float* dbeta,
                           void* ws, size_t ws_bytes, msn_stream_t stream);

/* Synchronised BatchNorm for data-parallel replicas (per-replica statistics differ from the
 * single-process statistics at the global batch).  The same two-pass statistics as msn_batchnorm_fwd / _bwd, cut at
 * the points where the CALLER all-reduces (SUM, RCCL) C floats (forward, twice) or 2C floats (backward):
 *   fwd: colsum(center = NULL) -> all-reduce -> mean_from_sum(count = global rows)
 *        -> colsum(center = mean) -> all-reduce -> rstd_from_sqdev (running statistics from the global batch)
 *        -> msn_batchnorm_apply
 *   bwd: bwd_sums (sums[0..C) = sum dy = the LOCAL d beta, sums[C..2C) = sum dy * xhat = the LOCAL d gamma; parameter
 *        gradients stay local sums, the gradient all-reduce adds the other ranks) -> all-reduce of a copy
 *        -> bwd_apply with the global sums and count. */
int msn_bn_colsum(const float* x, int64_t rows, int C, const float* center, float* out, void* ws, size_t ws_bytes,
                  msn_stream_t stream);
int msn_bn_mean_from_sum(const float* sum, int64_t count, int C, float* mean, msn_stream_t stream);
int msn_bn_rstd_from_sqdev(const float* sqdev, int64_t count, int C, float eps, float momentum, const float* mean,
                           float* running_mean, float* running_var, float* rstd, msn_stream_t stream);
int msn_batchnorm_apply(const float* x, int64_t rows, int C, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, const float* residual, int relu, float* y,
                        msn_stream_t stream);
int msn_bn_bwd_sums(const float* dy, const float* x, int64_t rows, int C, const float* mean, const float* rstd,
                    float* sums, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_bn_bwd_apply(const float* dy, const float* x, const float* dact, int64_t rows, int64_t count, int C,
                     const float* mean, const float* rstd, const float* gamma, const float* sums, float* dx,
                     msn_stream_t stream);

/* depthwise k x k conv, padding='same', + bias, + GELU (:65-69): x (B, gh, gw, C) channels-last,
 * w (C, 1, k, k).  fwd writes act = gelu(pre) and dact = gelu'(pre) with pre = conv + bias.
 * bwd: dx = conv^T(dpre) (+ add), dw (C,1,k,k), dbias (C; may be NULL). */
int msn_dwconv_gelu_fwd(const float* x, const float* w, const float* bias, int B, int gh, int gw, int C,
                        int k, float* dact, float* act, msn_stream_t stream);
size_t msn_dwconv_bwd_workspace_bytes(int B, int C, int k);
int msn_dwconv_bwd(const float* dpre, const float* x, const float* w, int B, int gh, int gw, int C, int k,
                   const float* add, float* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                   msn_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-tensor RAdam step with torch.optim.RAdam semantics (betas, eps, coupled L2 weight
 * decay, rectification once rho_t > 5) -- the optimiser of src/models_multimodal.py:306-310.
 * table: DEVICE array of n_tensors records of five 64-bit words {p*, g*, m*, v*, numel};
 * step is the 1-based update count.  One launch for the whole model; 28 B of HBM traffic / parameter.
 */
int msn_radam_step(const void* table, int n_tensors, int64_t max_numel, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int64_t step, msn_stream_t stream);
/* The same step for a training step recorded in a HIP graph: hyper[8] (device) = {lr, beta1, beta2, eps, weight_decay,
 * -, -, -}, step_counter[1] (device int64: steps taken so far).  Every launch increments the counter and derives
 * 1 / (1 - beta1^t) and the rectification term from it ON the device, so replays need no host write. */
int msn_radam_step_dev(const void* table, int n_tensors, int64_t max_numel, float* hyper, long long* step_counter,
                       msn_stream_t stream);

/* Channels-last convolution plumbing for the build-defined ResNet-18 / 1-D CNN encoders (not in the
 * reference): cols[(b,oh,ow)][(c,u,v)] = x[b, oh*sh+u-ph, ow*sw+v-pw, c] (0 outside), column order equal to
 * the flattening of a (C_out, C_in, kh, kw) weight, so conv = msn_sgemm(cols, W) ; col2im is its adjoint
 * (deterministic gather); a 1-D conv is the H = 1 case.  Max pooling keeps the arg-max pixel (int32). */
int msn_im2col(const float* x, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
               float* cols, msn_stream_t stream);
int msn_col2im(const float* dcols, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
               float* dx, msn_stream_t stream);
/* The same with TAP-MAJOR columns cols[(b,oh,ow)][(u,v,c)] (channels fastest: both kernels move 16-byte channel groups;
 * C % 4 == 0, 16-byte aligned tensors) and the weight re-laid to match: msn_conv_weight_relayout(to_tap = 1) copies
 * torch's (C_out, C_in, kh*kw) weight to (C_out, kh*kw, C_pad) with zeros in the channels C_in .. C_pad-1; to_tap = 0
 * is the inverse (for the weight gradient; the padding channels are dropped); to_tap = 2 gives (kh*kw, C_out, C_in)
 * (C_pad == C_in), the weight operand of msn_conv2d_dgrad.  msn_pad_channels widens a channels-last
 * tensor (rows x C -> rows x Cp, zeros added): a 3-channel image stem runs as a 4-channel convolution. */
int msn_im2col_tap(const float* x, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                   float* cols, msn_stream_t stream);
int msn_col2im_tap(const float* dcols, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int ph, int pw,
                   float* dx, msn_stream_t stream);
int msn_conv_weight_relayout(const float* src, int64_t co, int ci, int ci_pad, int taps, int to_tap, float* dst,
                             msn_stream_t stream);
int msn_pad_channels(const float* x, int64_t rows, int C, int Cp, float* out, msn_stream_t stream);

/* Implicit-GEMM convolution on channels-last fp32 tensors (the build-defined ResNet-18 / 1-D CNN encoders; no
 * reference counterpart): y[(b,oh,ow)][co] = sum_{u,v,c} x[b, oh*sh+u-ph, ow*sw+v-pw, c] * w_tap[co][(u,v,c)] without a
 * column matrix in memory -- the GEMM kernel's LDS-DMA lanes compute their own source addresses and padding taps read
 * a page of zeros.  msn_conv2d_implicit_ok says whether a shape is taken (C and C_out multiples of 32 and > 32,
 * images up to 512 x 512, B*OH*OW a multiple of 32); other shapes go through msn_im2col_tap + msn_sgemm.
 *   fwd:   w_tap = msn_conv_weight_relayout(to_tap = 1) of the (C_out, C, kh, kw) weight; epilogue NONE or RELU.
 *   dgrad: stride 1 only; dx[(b,h,w)][c] from dy[(b,oh,ow)][co] and w_tco = relayout(to_tap = 2): [(u,v,co)][c].
 *   wgrad: dw_tap[co][(u,v,c)] (relayout(to_tap = 0) gives torch's layout back) and, when dbias != NULL, dbias[co] =
 *          column sums of dy from the same launch; split over the pixels, slabs in `ws`.
 * `ws` of msn_conv2d_workspace_bytes(...) bytes serves all three. */
int msn_conv2d_implicit_ok(int B, int H, int W, int C, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
size_t msn_conv2d_workspace_bytes(int B, int H, int W, int C, int Cout, int kh, int kw, int sh, int sw, int ph, int pw);
int msn_conv2d_fwd(const float* x, int B, int H, int W, int C, const float* w_tap, int Cout, int kh, int kw, int sh,
                   int sw, int ph, int pw, const float* bias, int epilogue, float* y, void* ws, size_t ws_bytes,
                   msn_stream_t stream);
int msn_conv2d_dgrad(const float* dy, int B, int H, int W, int C, const float* w_tco, int Cout, int kh, int kw, int ph,
                     int pw, float* dx, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_conv2d_wgrad(const float* dy, const float* x, int B, int H, int W, int C, int Cout, int kh, int kw, int sh,
                     int sw, int ph, int pw, float* dw_tap, float* dbias, void* ws, size_t ws_bytes,
                     msn_stream_t stream);
int msn_maxpool2d_fwd(const float* x, int B, int H, int W, int C, int k, int s, int p, float* y, int* argmax,
                      msn_stream_t stream);
int msn_maxpool2d_bwd(const float* dy, const int* argmax, int B, int H, int W, int C, int k, int s, int p,
                      float* dx, msn_stream_t stream);

/* nn.Dropout in train mode (src/transformer_utils.py:108,113,116,140; src/models_multimodal.py:71,78,87,850):
 * y = keep(seed, i) ? x / (1 - p) : 0 (+ residual), keep from a counter-based hash of (seed, element index), so the
 * same call applied to the gradient reproduces the mask and nothing is stored.  In place allowed. */
int msn_dropout(const float* x, int64_t n, float p, uint64_t seed, const float* residual, float* y,
                msn_stream_t stream);
/* For a training step recorded in a HIP graph: the seed is seed_base[0] (DEVICE memory) + seed_offset, and
 * msn_seed_advance (one launch per recorded step) moves the base, so every replay draws new masks while a forward
 * call and its backward twin (same offset) still agree. */
int msn_dropout_dev(const float* x, int64_t n, float p, const uint64_t* seed_base, uint64_t seed_offset,
                    const float* residual, float* y, msn_stream_t stream);
int msn_seed_advance(uint64_t* seed_base, msn_stream_t stream);

/* On-device form of NoisyDataLoader.__iter__ (src/dataloader.py:88-287), the step right before the path:
 *   images: out = rot90^{rot[b]}( img + (2 u - 1) * noise_level * std(img) )   (B, C, S, S), std = torch.std of the
 *           whole batch; u = uniform [0,1) field, rot = quarter turns per sample (both supplied by the caller's RNG);
 *   series: out = x + g * err * noise_level   (g = standard-normal field). */
size_t msn_augment_workspace_bytes(void);
int msn_augment_images(const float* img, const float* u, const int* rot, int64_t B, int C, int S,
                       float noise_level, float* out, void* ws, size_t ws_bytes, msn_stream_t stream);
int msn_augment_series(const float* x, const float* g, const float* err, int64_t n, float noise_level,
                       float* out, msn_stream_t stream);

/* Masked MSE of the masked-light-curve pretraining objective (src/models_pretraining.py:201-231):
 * stats[0] = mean over {i : select[i]} of (pred[i] - target[i])^2, stats[1] = number of selected elements;
 * bwd: dpred = grad_out * 2 (pred - target) / count on the selected elements, 0 elsewhere. */
int msn_masked_mse_fwd(const float* pred, const float* target, const uint8_t* select, int64_t n, float* stats,
                       msn_stream_t stream);
int msn_masked_mse_bwd(const float* pred, const float* target, const uint8_t* select, int64_t n,
                       const float* stats, const float* grad_out, float* dpred, msn_stream_t stream);

/* feat[r] = (x[r]*m, t[r]*inv_norm*m, m, 0), m = mask[r]: the 4 input channels of the build-defined 1-D CNN
 * encoder for light curves / spectra (value, normalised time / wavelength, validity, pad). */
int msn_series_features(const float* x, const float* t, const uint8_t* mask, int64_t rows, float inv_norm,
                        float* feat, msn_stream_t stream);

/* Build-defined ViT image encoder (not in the reference; fills its `image_encoder` slot):
 * tok[b][0] = cls + pos[0], tok[b][1+i] = patch[b][i] + pos[1+i]  with T = 1 + n_patches, and the
 * compaction dpatch[b][i] = dtok[b][1+i] used by the backward. */
int msn_vit_tokens_fwd(const float* patch, const float* cls, const float* pos, int64_t B, int T, int e,
                       float* tok, msn_stream_t stream);
int msn_vit_tokens_bwd(const float* dtok, int64_t B, int T, int e, float* dpatch, msn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSN_HIP_H */
