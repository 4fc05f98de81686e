// Shared helpers for libmsn_hip.so (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/msn_hip.h"

namespace msn {

constexpr int kWave = 64;

// Last error text, retrievable through msn_last_error(); one slot per host thread.
void set_error(const char* fmt, ...);

// Every entry point returns 0 on success; on a bad shape / null pointer it records a message
// and returns MSN_ERR_SHAPE without launching anything.
#define MSN_REQUIRE(cond, ...)                     \
    do {                                           \
        if (!(cond)) {                             \
            ::msn::set_error(__VA_ARGS__);         \
            return MSN_ERR_SHAPE;                  \
        }                                          \
    } while (0)

#define MSN_LAUNCH_CHECK()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            ::msn::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,       \
                             hipGetErrorString(e__));                             \
            return MSN_ERR_HIP;                                                   \
        }                                                                         \
    } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Column sums from partial rows (pgemm.hip): part = [nparts][N] floats followed by COLSUM_SLICES x N floats of scratch; the
// order of the additions is fixed (two passes over fixed slices), so the sums are reproducible.
constexpr int COLSUM_SLICES = 32;
int colsum_finish(float* part, int nparts, int N, float* out, hipStream_t st);

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum through LDS; `red` holds >= blockDim.x/64 floats. Every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// Exact (erf) GELU, as torch.nn.GELU() default, and its derivative.
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// gelu(x) AND gelu'(x) (erf form) from ONE error-function evaluation, branch-free, no libm: the plane GEMM's GELU epilogue
// evaluates 64 of these per lane with nothing to overlap them (erff twice + expf cost a third of a K = 384 tile's matrix time).
//   |z| <= 1 (z = x / sqrt 2):  erf(z) = z P(z^2), P of degree 6            (max abs error 1.6e-7, fp32 Horner)
//   |z| >  1:  erfc(|z|) = exp(-z^2) t Q(t), t = 1 / (1 + 0.4 |z|), Q of degree 6   (max abs error 6.2e-8)
// least-squares fits on Chebyshev nodes, checked in fp32 arithmetic against scipy (tools/fit_gelu.py); exp(-z^2) is shared
// with the density term of the derivative.  Phi is formed as 1/2 + erf/2, 1 - erfc/2 or erfc/2 by sign, so the negative
// tail keeps its relative accuracy.
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
    const float z = x * 0.70710678118654752f, az = fabsf(z), s = z * z;
    const float e = __expf(-s);
    float p = 7.933350570965558e-05f;
    p = fmaf(p, s, -0.0008034805068746209f);
    p = fmaf(p, s, 0.005191213916987181f);
    p = fmaf(p, s, -0.02685539796948433f);
    p = fmaf(p, s, 0.11283625662326813f);
    p = fmaf(p, s, -0.3761262893676758f);
    p = fmaf(p, s, 1.128379225730896f);
    const float phi_s = fmaf(0.5f * z, p, 0.5f);                       // 1/2 + erf(z) / 2
    const float t = __builtin_amdgcn_rcpf(fmaf(0.4f, az, 1.f));         // (1 ulp: the correctly rounded quotient cost ten instructions and changes no result bound -- tools/fit_gelu.py)
    float q = -0.08732129633426666f;
    q = fmaf(q, t, 0.14394809305667877f);
    q = fmaf(q, t, 0.15008124709129333f);
    q = fmaf(q, t, 0.1063244417309761f);
    q = fmaf(q, t, 0.24356801807880402f);
    q = fmaf(q, t, 0.2167593091726303f);
    q = fmaf(q, t, 0.22653643786907196f);
    const float hc = 0.5f * q * t * e;                                 // erfc(|z|) / 2
    const float phi_l = x > 0.f ? 1.f - hc : hc;
    const float phi = az <= 1.f ? phi_s : phi_l;
    g = x * phi;
    dg = fmaf(x * 0.39894228040143268f, e, phi);
}

}  // namespace msn
