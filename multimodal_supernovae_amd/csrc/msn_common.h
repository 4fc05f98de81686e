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

}  // namespace msn
