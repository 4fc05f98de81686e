// ConvMixer image tower pieces (ref src/models_multimodal.py:38-95), channels-last.
//
// Activations live as token matrices X[(b, i, j)][c] (rows = B*gh*gw grid cells, cols = dim), so
// the patch convolution (stride = kernel = p, no bias) and the 1x1 convolutions are plain GEMMs
// on msn_sgemm (GELU fused in its epilogue), and everything else here is a streaming kernel:
//   patchify / un-patchify          image (B,C,H,W) <-> patch rows [(b,i,j)][(c,u,v)]
//   depthwise k x k 'same' conv     + bias + GELU fused, forward / dX / dW
//   BatchNorm2d over (B, H, W)      two-pass batch statistics (+ running-stat update), apply
//                                   (+ fused residual add), backward reduce + apply (+ fused GELU')
// Column reductions (BN statistics, d gamma / d beta, depthwise dW) are block partials in caller
// scratch followed by a fixed-order final pass: deterministic, no atomics.
#include <algorithm>
#include <math.h>

#include "msn_common.h"

namespace msn {

constexpr int RED_BLOCKS_MAX = 1024;   // 4 workgroups per CU on a narrow (<= 64-channel) matrix
constexpr int RED_CP = 64;                 // columns per block (threads along c)
constexpr int RED_RG = 256 / RED_CP;       // row groups per block

__global__ void patchify_kernel(const float* __restrict__ img, int B, int C, int H, int W, int p, int gh, int gw,
                                float* __restrict__ out) {
    const int64_t K = (int64_t)C * p * p, total = (int64_t)B * gh * gw * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / K;
        const int kk = (int)(i % K);
        const int v = kk % p, u = (kk / p) % p, c = kk / (p * p);
        const int gj = (int)(row % gw), gi = (int)((row / gw) % gh);
        const int64_t b = row / ((int64_t)gw * gh);
        out[i] = img[((b * C + c) * H + gi * p + u) * W + gj * p + v];
    }
}
__global__ void unpatchify_kernel(const float* __restrict__ dpatch, int B, int C, int H, int W, int p, int gh, int gw,
                                  float* __restrict__ dimg) {
    const int64_t total = (int64_t)B * C * H * W, K = (int64_t)C * p * p;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C);
        const int64_t b = i / ((int64_t)W * H * C);
        const int gi = y / p, gj = x / p;
        float v = 0.f;  // pixels beyond the floor(H/p) x floor(W/p) grid are never read by the conv
        if (gi < gh && gj < gw)
            v = dpatch[((b * gh + gi) * gw + gj) * K + ((int64_t)c * p + (y - gi * p)) * p + (x - gj * p)];
        dimg[i] = v;
    }
}

// ---- generic two-value column reduction over rows: part[blockIdx.x][which][c] -------------------
// f(row, c) -> (a, b); launched with grid (nblocks, cdiv(C, RED_CP)), 256 threads.
template <typename F>
__device__ __forceinline__ void col_reduce2(int64_t rows, int C, float* __restrict__ part, F f) {
    __shared__ float red[2][RED_RG][RED_CP];
    const int c = blockIdx.y * RED_CP + (threadIdx.x % RED_CP);
    const int rg = threadIdx.x / RED_CP;
    // four rows in flight per thread, each with its own pair of partial sums (one load per iteration and a single
    // dependent chain left a 64-channel BatchNorm at 1.6 TB/s); fixed combination order: deterministic
    float sa = 0.f, sb = 0.f;
    if (c < C) {
        const int64_t step = (int64_t)gridDim.x * RED_RG;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
        int64_t r = (int64_t)blockIdx.x * RED_RG + rg;
        for (; r + 3 * step < rows; r += 4 * step) {
            float x0, y0, x1, y1, x2, y2, x3, y3;
            f(r, c, x0, y0);
            f(r + step, c, x1, y1);
            f(r + 2 * step, c, x2, y2);
            f(r + 3 * step, c, x3, y3);
            a0 += x0, b0 += y0, a1 += x1, b1 += y1, a2 += x2, b2 += y2, a3 += x3, b3 += y3;
        }
        for (; r < rows; r += step) {
            float x0, y0;
            f(r, c, x0, y0);
            a0 += x0, b0 += y0;
        }
        sa = (a0 + a1) + (a2 + a3);
        sb = (b0 + b1) + (b2 + b3);
    }
    red[0][rg][threadIdx.x % RED_CP] = sa;
    red[1][rg][threadIdx.x % RED_CP] = sb;
    __syncthreads();
    if (rg == 0 && c < C) {
        float ta = 0.f, tb = 0.f;
        for (int k = 0; k < RED_RG; ++k) {
            ta += red[0][k][threadIdx.x];
            tb += red[1][k][threadIdx.x];
        }
        part[((int64_t)blockIdx.x * 2 + 0) * C + c] = ta;
        part[((int64_t)blockIdx.x * 2 + 1) * C + c] = tb;
    }
}

__global__ __launch_bounds__(256) void bn_sum_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                     float* __restrict__ part) {
    col_reduce2(rows, C, part, [&](int64_t r, int c, float& a, float& b) { a = x[r * C + c]; b = 0.f; });
}
__global__ __launch_bounds__(256) void bn_sqdev_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       int64_t rows, int C, float* __restrict__ part) {
    col_reduce2(rows, C, part, [&](int64_t r, int c, float& a, float& b) {
        const float d = x[r * C + c] - mean[c];
        a = d * d;
        b = 0.f;
    });
}
// Sum of the per-block partials of one channel: the finish kernels run 1024 threads = 16 groups x 64 channels, every
// group walks a sixteenth of the blocks with 4 independent loads in flight, groups combined through LDS in a fixed order
// (a single thread per channel walking all blocks was ~60 us of dependent loads per BatchNorm).
__device__ __forceinline__ float slab_sum16(const float* __restrict__ part, int64_t stride, int nblocks, int c, int C,
                                            float (*red)[64]) {
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int b = g;
        for (; b + 48 < nblocks; b += 64) {
            s0 += part[(int64_t)b * stride + c];
            s1 += part[(int64_t)(b + 16) * stride + c];
            s2 += part[(int64_t)(b + 32) * stride + c];
            s3 += part[(int64_t)(b + 48) * stride + c];
        }
        for (; b < nblocks; b += 16) s0 += part[(int64_t)b * stride + c];
    }
    __syncthreads();
    red[g][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    return t;
}
// stage 1 finish: mean = sum / rows
__global__ __launch_bounds__(1024) void bn_mean_finish_kernel(const float* __restrict__ part, int nblocks, int64_t rows,
                                                              int C, float* __restrict__ mean) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = slab_sum16(part, 2 * (int64_t)C, nblocks, c, C, red);
    if (threadIdx.x < 64 && c < C) mean[c] = s / (float)rows;
}
// stage 2 finish: biased var -> rstd; running stats (unbiased var) as nn.BatchNorm2d in train mode
__global__ __launch_bounds__(1024) void bn_var_finish_kernel(const float* __restrict__ part, int nblocks, int64_t rows,
                                                             int C, float eps, const float* __restrict__ mean,
                                                             float* __restrict__ rstd, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float momentum) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = slab_sum16(part, 2 * (int64_t)C, nblocks, c, C, red);
    if (threadIdx.x >= 64 || c >= C) return;
    const float var = s / (float)rows;
    rstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
        const float unbiased = rows > 1 ? s / (float)(rows - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}
__global__ void bn_eval_rstd_kernel(const float* __restrict__ running_var, int C, float eps, float* __restrict__ rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) rstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// y = (x - mean) * rstd * gamma + beta (+ res), optionally followed by ReLU
__global__ void bn_apply_kernel(const float* __restrict__ x, int64_t rows, int C, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const float* __restrict__ res, int relu,
                                float* __restrict__ y) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = (x[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
        if (res) v += res[i];
        if (relu) v = fmaxf(v, 0.f);
        y[i] = v;
    }
}
// dm = dy * (y > 0): the gradient after a trailing ReLU (also what flows into the residual branch)
__global__ void relu_mask_kernel(const float* __restrict__ dy, const float* __restrict__ y, int64_t total,
                                 float* __restrict__ dm) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        dm[i] = y[i] > 0.f ? dy[i] : 0.f;
}
// partial (sum dy, sum dy * xhat)
// (ymask != null: BatchNorm was followed by ReLU -- the incoming gradient counts where the saved output is positive)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, int64_t rows, int C,
                                                            float* __restrict__ part) {
    col_reduce2(rows, C, part, [&](int64_t r, int c, float& a, float& b) {
        float d = dy[r * C + c];
        if (ymask) d = ymask[r * C + c] > 0.f ? d : 0.f;
        a = d;
        b = d * (x[r * C + c] - mean[c]) * rstd[c];
    });
}
__global__ __launch_bounds__(1024) void bn_bwd_finish_kernel(const float* __restrict__ part, int nblocks, int C,
                                                             float* __restrict__ dbeta, float* __restrict__ dgamma) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float sa = slab_sum16(part, 2 * (int64_t)C, nblocks, c, C, red);
    const float sb = slab_sum16(part + C, 2 * (int64_t)C, nblocks, c, C, red);
    if (threadIdx.x < 64 && c < C) {
        dbeta[c] = sa;
        dgamma[c] = sb;
    }
}
// dx = gamma * rstd * (dy - [train](dbeta + xhat * dgamma) / rows), then * pre[] (= saved gelu') when non-null
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ ymask,
                                    const float* __restrict__ x, const float* __restrict__ pre, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ dbeta, const float* __restrict__ dgamma, int64_t rows,
                                    int64_t count, int C, int training, float* __restrict__ dx) {
    const int64_t total = rows * C;
    const float inv_n = 1.f / (float)count;   // count = rows, or the rows of ALL ranks under synchronised BN
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float d = dy[i];
        if (ymask) d = ymask[i] > 0.f ? d : 0.f;
        if (training) {
            const float xhat = (x[i] - mean[c]) * rstd[c];
            d -= (dbeta[c] + xhat * dgamma[c]) * inv_n;
        }
        d *= gamma[c] * rstd[c];
        if (pre) d *= pre[i];   // pre[] holds gelu'(pre-activation)
        dx[i] = d;
    }
}

// ---- depthwise conv, channels-last (B, gh, gw, C), weight (C, 1, k, k), 'same' padding ------------
struct DwGeom { int B, gh, gw, C, k, pad; };

__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  DwGeom g, float* __restrict__ pre, float* __restrict__ act) {
    const int64_t total = (int64_t)g.B * g.gh * g.gw * g.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        const int j = (int)((i / g.C) % g.gw), ii = (int)((i / ((int64_t)g.C * g.gw)) % g.gh);
        const int64_t b = i / ((int64_t)g.C * g.gw * g.gh);
        float s = bias ? bias[c] : 0.f;
        for (int u = 0; u < g.k; ++u) {
            const int yy = ii + u - g.pad;
            if (yy < 0 || yy >= g.gh) continue;
            for (int v = 0; v < g.k; ++v) {
                const int xx = j + v - g.pad;
                if (xx < 0 || xx >= g.gw) continue;
                s = fmaf(w[((int64_t)c * g.k + u) * g.k + v], x[((b * g.gh + yy) * g.gw + xx) * g.C + c], s);
            }
        }
        const float cdf = 0.5f * (1.f + erff(s * 0.70710678118654752f));
        pre[i] = cdf + s * 0.39894228040143268f * __expf(-0.5f * s * s);   // gelu'(pre), consumed by the BN backward
        act[i] = s * cdf;
    }
}
// dx[b,y,x,c] = sum_{u,v} w[c,u,v] * dpre[b, y-u+pad, x-v+pad, c] (+ add)
__global__ void dwconv_bwd_dx_kernel(const float* __restrict__ dpre, const float* __restrict__ w, DwGeom g,
                                     const float* __restrict__ add, float* __restrict__ dx) {
    const int64_t total = (int64_t)g.B * g.gh * g.gw * g.C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        const int xx = (int)((i / g.C) % g.gw), yy = (int)((i / ((int64_t)g.C * g.gw)) % g.gh);
        const int64_t b = i / ((int64_t)g.C * g.gw * g.gh);
        float s = add ? add[i] : 0.f;
        for (int u = 0; u < g.k; ++u) {
            const int oi = yy - u + g.pad;
            if (oi < 0 || oi >= g.gh) continue;
            for (int v = 0; v < g.k; ++v) {
                const int oj = xx - v + g.pad;
                if (oj < 0 || oj >= g.gw) continue;
                s = fmaf(w[((int64_t)c * g.k + u) * g.k + v], dpre[((b * g.gh + oi) * g.gw + oj) * g.C + c], s);
            }
        }
        dx[i] = s;
    }
}
// partial dW: part[blockIdx.x][c][u][v] = sum over this block's samples;  dbias partial alongside
__global__ __launch_bounds__(256) void dwconv_bwd_dw_kernel(const float* __restrict__ dpre, const float* __restrict__ x,
                                                            DwGeom g, float* __restrict__ part) {
    const int kk = g.k * g.k;
    const int nout = g.C * (kk + 1);
    for (int o = threadIdx.x; o < nout; o += blockDim.x) {
        const int c = o / (kk + 1), t = o % (kk + 1);
        float s = 0.f;
        for (int64_t b = blockIdx.x; b < g.B; b += gridDim.x) {
            const float* dp = dpre + b * g.gh * g.gw * g.C;
            const float* xp = x + b * g.gh * g.gw * g.C;
            if (t == kk) {
                for (int q = 0; q < g.gh * g.gw; ++q) s += dp[(int64_t)q * g.C + c];
            } else {
                const int u = t / g.k, v = t % g.k;
                for (int i = 0; i < g.gh; ++i) {
                    const int yy = i + u - g.pad;
                    if (yy < 0 || yy >= g.gh) continue;
                    for (int j = 0; j < g.gw; ++j) {
                        const int xx = j + v - g.pad;
                        if (xx < 0 || xx >= g.gw) continue;
                        s = fmaf(dp[((int64_t)i * g.gw + j) * g.C + c], xp[((int64_t)yy * g.gw + xx) * g.C + c], s);
                    }
                }
            }
        }
        part[(int64_t)blockIdx.x * nout + o] = s;
    }
}
__global__ void dwconv_bwd_dw_finish_kernel(const float* __restrict__ part, int nblocks, int C, int kk,
                                            float* __restrict__ dw, float* __restrict__ dbias) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int nout = C * (kk + 1);
    if (o >= nout) return;
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += part[(int64_t)b * nout + o];
    const int c = o / (kk + 1), t = o % (kk + 1);
    if (t == kk) { if (dbias) dbias[c] = s; }
    else dw[(int64_t)c * kk + t] = s;
}

static int red_blocks(int64_t rows) { return (int)std::min<int64_t>(cdiv(rows, 4 * RED_RG), RED_BLOCKS_MAX); }
static unsigned ew_grid(int64_t total) { return (unsigned)std::min<int64_t>(cdiv(total, 256), 4096); }

}  // namespace msn

using namespace msn;

extern "C" int msn_patchify(const float* img, int B, int C, int H, int W, int p, float* patches, msn_stream_t stream) {
    MSN_REQUIRE(img && patches && B > 0 && C > 0 && p > 0 && H >= p && W >= p, "msn_patchify: bad arguments");
    const int gh = H / p, gw = W / p;
    hipLaunchKernelGGL(patchify_kernel, dim3(ew_grid((int64_t)B * gh * gw * C * p * p)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), img, B, C, H, W, p, gh, gw, patches);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_unpatchify(const float* dpatches, int B, int C, int H, int W, int p, float* dimg, msn_stream_t stream) {
    MSN_REQUIRE(dimg && dpatches && B > 0 && C > 0 && p > 0 && H >= p && W >= p, "msn_unpatchify: bad arguments");
    hipLaunchKernelGGL(unpatchify_kernel, dim3(ew_grid((int64_t)B * C * H * W)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dpatches, B, C, H, W, p, H / p, W / p, dimg);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" size_t msn_bn_workspace_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    return sizeof(float) * 2 * (size_t)C * RED_BLOCKS_MAX;
}

// training != 0: batch statistics (mean, rstd written; running stats updated in place when given).
// training == 0: mean := running_mean (must be passed as `mean`), rstd from running_var.
extern "C" int msn_batchnorm_fwd(const float* x, int64_t rows, int C, const float* gamma, const float* beta, float eps,
                                 int training, float momentum, float* running_mean, float* running_var,
                                 const float* residual, int relu, float* y, float* mean, float* rstd, void* ws,
                                 size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0, "msn_batchnorm_fwd: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned cb = (unsigned)cdiv(C, 64);
    if (training) {
        MSN_REQUIRE(ws && ws_bytes >= msn_bn_workspace_bytes(rows, C), "msn_batchnorm_fwd: workspace too small");
        float* part = static_cast<float*>(ws);
        const int nb = red_blocks(rows);
        const dim3 grid(nb, (unsigned)cdiv(C, RED_CP));
        hipLaunchKernelGGL(bn_sum_kernel, grid, dim3(256), 0, st, x, rows, C, part);
        hipLaunchKernelGGL(bn_mean_finish_kernel, dim3(cb), dim3(1024), 0, st, part, nb, rows, C, mean);
        hipLaunchKernelGGL(bn_sqdev_kernel, grid, dim3(256), 0, st, x, mean, rows, C, part);
        hipLaunchKernelGGL(bn_var_finish_kernel, dim3(cb), dim3(1024), 0, st, part, nb, rows, C, eps, mean, rstd,
                           running_mean, running_var, momentum);
    } else {
        MSN_REQUIRE(running_mean && running_var, "msn_batchnorm_fwd: eval mode needs the running statistics");
        if (hipMemcpyAsync(mean, running_mean, sizeof(float) * C, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_error("msn_batchnorm_fwd: copy of running_mean failed");
            return MSN_ERR_HIP;
        }
        hipLaunchKernelGGL(bn_eval_rstd_kernel, dim3(cb), dim3(64), 0, st, running_var, C, eps, rstd);
    }
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, st, x, rows, C, mean, rstd, gamma, beta,
                       residual, relu, y);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// dx = BN backward (then * gelu'(pre) when pre != NULL); dgamma, dbeta always written.
static int batchnorm_bwd_impl(const float* dy, const float* ymask, const float* x, const float* pre, int64_t rows, int C,
                              const float* mean, const float* rstd, const float* gamma, int training, float* dx,
                              float* dgamma, float* dbeta, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && rows > 0 && C > 0,
                "msn_batchnorm_bwd: bad arguments");
    MSN_REQUIRE(ws && ws_bytes >= msn_bn_workspace_bytes(rows, C), "msn_batchnorm_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const int nb = red_blocks(rows);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb, (unsigned)cdiv(C, RED_CP)), dim3(256), 0, st, dy, ymask, x, mean, rstd,
                       rows, C, part);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)cdiv(C, 64)), dim3(1024), 0, st, part, nb, C, dbeta, dgamma);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, st, dy, ymask, x, pre, mean, rstd, gamma,
                       dbeta, dgamma, rows, rows, C, training, dx);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_batchnorm_bwd(const float* dy, const float* x, const float* pre, int64_t rows, int C,
                                 const float* mean, const float* rstd, const float* gamma, int training, float* dx,
                                 float* dgamma, float* dbeta, void* ws, size_t ws_bytes, msn_stream_t stream) {
    return batchnorm_bwd_impl(dy, nullptr, x, pre, rows, C, mean, rstd, gamma, training, dx, dgamma, dbeta, ws, ws_bytes, stream);
}
// BatchNorm followed by ReLU: y = relu(bn(x)) was saved; the gradient dy is masked by y > 0 inside both passes (no
// separate masking pass, no masked copy of dy)
extern "C" int msn_batchnorm_relu_bwd(const float* dy, const float* y, const float* x, int64_t rows, int C,
                                      const float* mean, const float* rstd, const float* gamma, int training, float* dx,
                                      float* dgamma, float* dbeta, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(y, "msn_batchnorm_relu_bwd: null output");
    return batchnorm_bwd_impl(dy, y, x, nullptr, rows, C, mean, rstd, gamma, training, dx, dgamma, dbeta, ws, ws_bytes, stream);
}

// ---- synchronised (data-parallel) BatchNorm: the same two-pass statistics cut at the points where the caller
// all-reduces C (forward, twice) or 2C (backward) floats across the ranks -------------------------------------------
__global__ __launch_bounds__(1024) void bn_sum_finish_kernel(const float* __restrict__ part, int nblocks, int C,
                                                             float* __restrict__ out) {
    __shared__ float red[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const float s = slab_sum16(part, 2 * (int64_t)C, nblocks, c, C, red);
    if (threadIdx.x < 64 && c < C) out[c] = s;
}
__global__ void bn_mean_from_sum_kernel(const float* __restrict__ sum, int64_t count, int C, float* __restrict__ mean) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) mean[c] = sum[c] / (float)count;
}
__global__ void bn_rstd_from_sqdev_kernel(const float* __restrict__ sqdev, int64_t count, int C, float eps,
                                          const float* __restrict__ mean, float* __restrict__ rstd,
                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                          float momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = sqdev[c], var = s / (float)count;
    rstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) {
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
        const float unbiased = count > 1 ? s / (float)(count - 1) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

extern "C" int msn_bn_colsum(const float* x, int64_t rows, int C, const float* center, float* out, void* ws,
                             size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(x && out && rows > 0 && C > 0, "msn_bn_colsum: bad arguments");
    MSN_REQUIRE(ws && ws_bytes >= msn_bn_workspace_bytes(rows, C), "msn_bn_colsum: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const int nb = red_blocks(rows);
    const dim3 grid(nb, (unsigned)cdiv(C, RED_CP));
    if (center)
        hipLaunchKernelGGL(bn_sqdev_kernel, grid, dim3(256), 0, st, x, center, rows, C, part);
    else
        hipLaunchKernelGGL(bn_sum_kernel, grid, dim3(256), 0, st, x, rows, C, part);
    hipLaunchKernelGGL(bn_sum_finish_kernel, dim3((unsigned)cdiv(C, 64)), dim3(1024), 0, st, part, nb, C, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_bn_mean_from_sum(const float* sum, int64_t count, int C, float* mean, msn_stream_t stream) {
    MSN_REQUIRE(sum && mean && count > 0 && C > 0, "msn_bn_mean_from_sum: bad arguments");
    hipLaunchKernelGGL(bn_mean_from_sum_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       sum, count, C, mean);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_bn_rstd_from_sqdev(const float* sqdev, int64_t count, int C, float eps, float momentum,
                                      const float* mean, float* running_mean, float* running_var, float* rstd,
                                      msn_stream_t stream) {
    MSN_REQUIRE(sqdev && mean && rstd && count > 0 && C > 0, "msn_bn_rstd_from_sqdev: bad arguments");
    MSN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "msn_bn_rstd_from_sqdev: running stats come in pairs");
    hipLaunchKernelGGL(bn_rstd_from_sqdev_kernel, dim3((unsigned)cdiv(C, 64)), dim3(64), 0,
                       static_cast<hipStream_t>(stream), sqdev, count, C, eps, mean, rstd, running_mean, running_var,
                       momentum);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_batchnorm_apply(const float* x, int64_t rows, int C, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, const float* residual, int relu, float* y,
                                   msn_stream_t stream) {
    MSN_REQUIRE(x && mean && rstd && gamma && beta && y && rows > 0 && C > 0, "msn_batchnorm_apply: bad arguments");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, static_cast<hipStream_t>(stream), x, rows,
                       C, mean, rstd, gamma, beta, residual, relu, y);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_bn_bwd_sums(const float* dy, const float* x, int64_t rows, int C, const float* mean,
                               const float* rstd, float* sums, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(dy && x && mean && rstd && sums && rows > 0 && C > 0, "msn_bn_bwd_sums: bad arguments");
    MSN_REQUIRE(ws && ws_bytes >= msn_bn_workspace_bytes(rows, C), "msn_bn_bwd_sums: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const int nb = red_blocks(rows);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb, (unsigned)cdiv(C, RED_CP)), dim3(256), 0, st, dy, (const float*)nullptr, x, mean, rstd,
                       rows, C, part);
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((unsigned)cdiv(C, 64)), dim3(1024), 0, st, part, nb, C, sums, sums + C);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_bn_bwd_apply(const float* dy, const float* x, const float* dact, int64_t rows, int64_t count, int C,
                                const float* mean, const float* rstd, const float* gamma, const float* sums, float* dx,
                                msn_stream_t stream) {
    MSN_REQUIRE(dy && x && mean && rstd && gamma && sums && dx && rows > 0 && count >= rows && C > 0,
                "msn_bn_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * C)), dim3(256), 0, static_cast<hipStream_t>(stream), dy,
                       (const float*)nullptr, x, dact, mean, rstd, gamma, sums, sums + C, rows, count, C, 1, dx);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

static int dw_geom(const char* who, int B, int gh, int gw, int C, int k, DwGeom* g) {
    MSN_REQUIRE(B > 0 && gh > 0 && gw > 0 && C > 0 && k > 0, "%s: bad sizes", who);
    *g = DwGeom{B, gh, gw, C, k, (k - 1) / 2};  // torch padding='same': left pad = (k-1)//2
    return MSN_OK;
}

extern "C" int msn_dwconv_gelu_fwd(const float* x, const float* w, const float* bias, int B, int gh, int gw, int C,
                                   int k, float* pre, float* act, msn_stream_t stream) {
    DwGeom g;
    if (int rc = dw_geom("msn_dwconv_gelu_fwd", B, gh, gw, C, k, &g)) return rc;
    MSN_REQUIRE(x && w && pre && act, "msn_dwconv_gelu_fwd: null pointer");
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(ew_grid((int64_t)B * gh * gw * C)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, w, bias, g, pre, act);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" size_t msn_dwconv_bwd_workspace_bytes(int B, int C, int k) {
    if (B <= 0 || C <= 0 || k <= 0) return 0;
    return sizeof(float) * (size_t)std::min(B, RED_BLOCKS_MAX) * (size_t)C * (size_t)(k * k + 1);
}

// dx = conv^T(dpre) (+ add);  dw (C, k, k), dbias (C) reduced over the batch.
extern "C" int msn_dwconv_bwd(const float* dpre, const float* x, const float* w, int B, int gh, int gw, int C, int k,
                              const float* add, float* dx, float* dw, float* dbias, void* ws, size_t ws_bytes,
                              msn_stream_t stream) {
    DwGeom g;
    if (int rc = dw_geom("msn_dwconv_bwd", B, gh, gw, C, k, &g)) return rc;
    MSN_REQUIRE(dpre && x && w && dx && dw, "msn_dwconv_bwd: null pointer");
    MSN_REQUIRE(ws && ws_bytes >= msn_dwconv_bwd_workspace_bytes(B, C, k), "msn_dwconv_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dwconv_bwd_dx_kernel, dim3(ew_grid((int64_t)B * gh * gw * C)), dim3(256), 0, st, dpre, w, g, add, dx);
    const int nb = std::min(B, RED_BLOCKS_MAX);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(dwconv_bwd_dw_kernel, dim3(nb), dim3(256), 0, st, dpre, x, g, part);
    const int nout = C * (k * k + 1);
    hipLaunchKernelGGL(dwconv_bwd_dw_finish_kernel, dim3((unsigned)cdiv(nout, 128)), dim3(128), 0, st, part, nb, C, k * k,
                       dw, dbias);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_relu_mask(const float* dy, const float* y, int64_t total, float* dmasked, msn_stream_t stream) {
    MSN_REQUIRE(dy && y && dmasked && total > 0, "msn_relu_mask: bad arguments");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(ew_grid(total)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, total,
                       dmasked);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
