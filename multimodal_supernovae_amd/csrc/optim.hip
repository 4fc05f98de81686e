// Fused multi-tensor RAdam step (torch.optim.RAdam semantics: L2 weight decay folded into the
// gradient, bias-corrected first moment, variance rectification once rho_t > 5) -- the optimiser the
// reference builds in configure_optimizers, src/models_multimodal.py:306-310.
// HBM-bound: reads p, g, m, v and writes p, m, v = 28 B / parameter; one launch for the whole model
// through a device table of per-tensor descriptors (blockIdx.y = tensor).
#include <algorithm>

#include "msn_common.h"

namespace msn {

struct RadamTensor {  // 5 x 8 bytes, uploaded by the host as int64 words
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
};

// hyper != NULL: the seven scalars come from device memory {lr, beta1, beta2, eps, weight_decay, inv_c1, rect_scale}
// (a launch recorded in a HIP graph is replayed with the step-dependent values of the replay, not of the capture)
__global__ void radam_kernel(const RadamTensor* __restrict__ table, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float inv_c1, float rect_scale /* rect * sqrt(c2), 0 = unrectified */,
                             const float* __restrict__ hyper) {
    if (hyper) {
        lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], weight_decay = hyper[4], inv_c1 = hyper[5],
        rect_scale = hyper[6];
    }
    const RadamTensor t = table[blockIdx.y];
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    auto upd = [&](float& p, float g, float& m, float& v) {
        g = fmaf(weight_decay, p, g);
        m = beta1 * m + (1.f - beta1) * g;
        v = beta2 * v + (1.f - beta2) * g * g;
        const float mh = m * inv_c1;
        if (rect_scale > 0.f) p -= lr * mh * (rect_scale / (sqrtf(v) + eps));
        else p -= lr * mh;
    };
    const int64_t n4 = vec ? t.n / 4 : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4*>(t.p)[i], m = reinterpret_cast<float4*>(t.m)[i],
               v = reinterpret_cast<float4*>(t.v)[i];
        const float4 g = reinterpret_cast<const float4*>(t.g)[i];
        upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
        reinterpret_cast<float4*>(t.p)[i] = p;
        reinterpret_cast<float4*>(t.m)[i] = m;
        reinterpret_cast<float4*>(t.v)[i] = v;
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += (int64_t)gridDim.x * blockDim.x)
        upd(t.p[i], t.g[i], t.m[i], t.v[i]);
}

}  // namespace msn

using namespace msn;

// table: device array of n_tensors x {p, g, m, v, numel} (int64 words).  step >= 1 is the 1-based
// count of this update (the same for every tensor, as in the reference's single parameter group).
extern "C" int msn_radam_step(const void* table, int n_tensors, int64_t max_numel, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int64_t step, msn_stream_t stream) {
    MSN_REQUIRE(table && n_tensors > 0 && n_tensors <= 65535 && max_numel > 0 && step >= 1, "msn_radam_step: bad arguments");
    const double b1 = beta1, b2 = beta2;
    const double c1 = 1.0 - pow(b1, (double)step), c2 = 1.0 - pow(b2, (double)step);
    const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
    const double rho_t = rho_inf - 2.0 * (double)step * pow(b2, (double)step) / c2;
    double rect_scale = 0.0;
    if (rho_t > 5.0)
        rect_scale = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) * sqrt(c2);
    const unsigned gx = (unsigned)std::min<int64_t>(cdiv(max_numel, 4 * 256), 1024);
    hipLaunchKernelGGL(radam_kernel, dim3(gx ? gx : 1, n_tensors), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const RadamTensor*>(table), lr, beta1, beta2, eps, weight_decay, (float)(1.0 / c1),
                       (float)rect_scale, static_cast<const float*>(nullptr));
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// Step-dependent scalars computed ON the device from a device-resident step counter: a recorded launch needs no host
// write between replays (a pinned-buffer refresh would race with the copy node of a replay still in flight).
__global__ void radam_prepare_kernel(float* __restrict__ hyper, long long* __restrict__ step_counter) {
    const long long step = ++step_counter[0];
    const double b1 = hyper[1], b2 = hyper[2];
    const double c1 = 1.0 - pow(b1, (double)step), c2 = 1.0 - pow(b2, (double)step);
    const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
    const double rho_t = rho_inf - 2.0 * (double)step * pow(b2, (double)step) / c2;
    double rect_scale = 0.0;
    if (rho_t > 5.0)
        rect_scale = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t)) * sqrt(c2);
    hyper[5] = (float)(1.0 / c1);
    hyper[6] = (float)rect_scale;
}

// The same step for a training step recorded in a HIP graph: hyper[8] (device) = {lr, beta1, beta2, eps, weight_decay,
// -, -, -} and step_counter[1] (device, the number of steps taken so far); every launch increments the counter and
// derives 1 / (1 - beta1^t) and the rectification term from it on the device (same double-precision formulas).
extern "C" int msn_radam_step_dev(const void* table, int n_tensors, int64_t max_numel, float* hyper,
                                  long long* step_counter, msn_stream_t stream) {
    MSN_REQUIRE(table && hyper && step_counter && n_tensors > 0 && n_tensors <= 65535 && max_numel > 0,
                "msn_radam_step_dev: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(radam_prepare_kernel, dim3(1), dim3(1), 0, st, hyper, step_counter);
    const unsigned gx = (unsigned)std::min<int64_t>(cdiv(max_numel, 4 * 256), 1024);
    hipLaunchKernelGGL(radam_kernel, dim3(gx ? gx : 1, n_tensors), dim3(256), 0, st,
                       static_cast<const RadamTensor*>(table), 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, hyper);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
