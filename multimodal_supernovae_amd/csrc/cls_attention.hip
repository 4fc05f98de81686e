// Attention of ONE query per (sample, head) over T keys -- the class-token row of the build-defined ViT's LAST block
// (encoders.py: the head reads token 0 only, so of that block's attention only this row is evaluated; the arithmetic is
// SelfAttention of ref src/transformer_utils.py:36-89 for one query: scores * scale, softmax over the keys, @ values; no mask).
//
// The products are 2 x T x 64 multiply-adds per pair: nothing for a matrix core to do (msn_attention_fwd / _bwd run the shape as a
// 16-row query tile with fifteen rows of padding: 0.66 ms per launch at 512 x 12 pairs over 197 keys, 0.9 TB/s of its K / V rows).
// These kernels are the bytes: a wave per (sample, head), eight lanes side by side on the 64 columns of a key's row (16 bytes of bf16
// or 2 x 16 bytes of fp32 per lane), eight keys per instruction; every K and V row is read ONCE per direction -- the forward keeps
// the scores in registers (T <= 256: 32 per lane), the backward the probabilities the forward saved (B H T floats) -- and the
// gradient rows are written once.  K | V rows are fp32 (the plane / fp32 towers) or bf16 (the bf16-resident tower: its key | value
// projection writes bf16 and the gradient goes back as bf16 into msn_bgemm_nt / msn_bgemm_tn).
#include <math.h>

#include "msn_common.h"

namespace msn {
namespace {

typedef unsigned short u16;
constexpr int HD = 64;          // head width (eight lanes x eight columns)
constexpr int MAXIT = 32;       // key groups of eight per wave: T <= 256

template <typename T>
__device__ __forceinline__ void load8(float (&v)[8], const T* p);
template <>
__device__ __forceinline__ void load8<float>(float (&v)[8], const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<u16>(float (&v)[8], const u16* p) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[2 * i] = __uint_as_float(w[i] << 16), v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void store8<u16>(u16* p, const float (&v)[8]) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t ab = {v[2 * i], v[2 * i + 1]};
        w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bf16x2_t));      // round to nearest even
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
// sum over the eight lanes of a key (lane & 7), and over the eight key groups of the wave (lane >> 3)
__device__ __forceinline__ float sum_parts(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    return v + __shfl_xor(v, 4, 64);
}
__device__ __forceinline__ float sum_groups(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float max_groups(float v) {
    v = fmaxf(v, __shfl_xor(v, 8, 64));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}

struct ClsArgs {
    const float* q; int64_t ldq;
    const void* kv; int64_t ldkv;
    int B, H, T;
    float scale;
    float* out; int64_t ldo;        // forward: written; backward: read
    float* P;                       // [B][H][T]: forward written, backward read
    const float* dout; int64_t ldd;
    float* dq; int64_t lddq;
    void* dkv; int64_t lddkv;
};

template <typename KT>
__global__ __launch_bounds__(256) void cls_attn_fwd_kernel(const ClsArgs p) {
    const int lane = threadIdx.x & 63, pt = lane & 7, kg = lane >> 3;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= p.B * p.H) return;
    const int b = pair / p.H, h = pair % p.H, e = p.H * HD;
    const int nit = (p.T + 7) >> 3;
    float qv[8];
    load8<float>(qv, p.q + (int64_t)b * p.ldq + h * HD + 8 * pt);
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] *= p.scale;
    const KT* rows = static_cast<const KT*>(p.kv) + (int64_t)b * p.T * p.ldkv + h * HD + 8 * pt;
    float s[MAXIT];
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        s[it] = -INFINITY;
        if (it < nit) {
            const int j = 8 * it + kg;
            float kx[8];
            load8<KT>(kx, rows + (int64_t)(j < p.T ? j : 0) * p.ldkv);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf(qv[i], kx[i], d);
            d = sum_parts(d);
            s[it] = j < p.T ? d : -INFINITY;
            m = fmaxf(m, s[it]);
        }
    }
    m = max_groups(m);
    float l = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (it < nit) {
            s[it] = __expf(s[it] - m);          // exp(-inf) = 0 beyond the sequence
            l += s[it];
        }
    l = sum_groups(l);                          // (the eight lanes of a key hold the same value: summed over the key groups only)
    const float inv = 1.f / l;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* prow = p.P + (int64_t)pair * p.T;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (it < nit) {
            const int j = 8 * it + kg;
            const float pj = s[it] * inv;
            if (pt == 0 && j < p.T) prow[j] = pj;
            float vx[8];
            load8<KT>(vx, rows + (int64_t)(j < p.T ? j : 0) * p.ldkv + e);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(pj, vx[i], acc[i]);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = sum_groups(acc[i]);
    if (kg == 0) store8<float>(p.out + (int64_t)b * p.ldo + h * HD + 8 * pt, acc);
}

template <typename KT>
__global__ __launch_bounds__(256) void cls_attn_bwd_kernel(const ClsArgs p) {
    const int lane = threadIdx.x & 63, pt = lane & 7, kg = lane >> 3;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= p.B * p.H) return;
    const int b = pair / p.H, h = pair % p.H, e = p.H * HD;
    const int nit = (p.T + 7) >> 3;
    float qv[8], dv8[8], ov[8];
    load8<float>(qv, p.q + (int64_t)b * p.ldq + h * HD + 8 * pt);
    load8<float>(dv8, p.dout + (int64_t)b * p.ldd + h * HD + 8 * pt);
    load8<float>(ov, p.out + (int64_t)b * p.ldo + h * HD + 8 * pt);
    float delta = 0.f;                          // sum_j p_j dP_j = dout . out
#pragma unroll
    for (int i = 0; i < 8; ++i) delta = fmaf(dv8[i], ov[i], delta);
    delta = sum_parts(delta);
    const KT* rows = static_cast<const KT*>(p.kv) + (int64_t)b * p.T * p.ldkv + h * HD + 8 * pt;
    KT* drows = static_cast<KT*>(p.dkv) + (int64_t)b * p.T * p.lddkv + h * HD + 8 * pt;
    const float* prow = p.P + (int64_t)pair * p.T;
    float ds[MAXIT];
    // values: dP_j = dout . v_j, dv_j = p_j dout, ds_j = p_j (dP_j - delta)
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        ds[it] = 0.f;
        if (it < nit) {
            const int j = 8 * it + kg;
            const bool ok = j < p.T;
            const float pj = ok ? prow[j] : 0.f;
            float vx[8];
            load8<KT>(vx, rows + (int64_t)(ok ? j : 0) * p.ldkv + e);
            float dp = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) dp = fmaf(dv8[i], vx[i], dp);
            dp = sum_parts(dp);
            ds[it] = pj * (dp - delta);
            float g[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = pj * dv8[i];
            if (ok) store8<KT>(drows + (int64_t)j * p.lddkv + e, g);
        }
    }
    // keys: dq += ds_j k_j, dk_j = scale ds_j q
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
        if (it < nit) {
            const int j = 8 * it + kg;
            const bool ok = j < p.T;
            float kx[8];
            load8<KT>(kx, rows + (int64_t)(ok ? j : 0) * p.ldkv);
            const float w = ds[it] * p.scale;
            float g[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = fmaf(ds[it], kx[i], acc[i]);
                g[i] = w * qv[i];
            }
            if (ok) store8<KT>(drows + (int64_t)j * p.lddkv, g);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = sum_groups(acc[i]) * p.scale;
    if (kg == 0) store8<float>(p.dq + (int64_t)b * p.lddq + h * HD + 8 * pt, acc);
}

bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

}  // namespace
}  // namespace msn

using namespace msn;

extern "C" int msn_cls_attention_supported(int T, int head_dim) { return head_dim == HD && T >= 1 && T <= 8 * MAXIT ? 1 : 0; }

extern "C" int msn_cls_attention_fwd(const float* q, int64_t ldq, const void* kv, int64_t ldkv, int kv_bf16, int B, int H, int T,
                                     int head_dim, float scale, float* out, int64_t ldo, float* probs, msn_stream_t stream) {
    MSN_REQUIRE(q && kv && out && probs && B > 0 && H > 0, "msn_cls_attention_fwd: empty operand");
    MSN_REQUIRE(msn_cls_attention_supported(T, head_dim), "msn_cls_attention_fwd: head_dim %d (must be 64), T = %d (1 .. 256)", head_dim, T);
    const int e = H * HD, es = kv_bf16 ? 2 : 4;
    MSN_REQUIRE(ldq >= e && ldo >= e && ldkv >= 2 * e && ldq % 4 == 0 && ldo % 4 == 0 && (ldkv * es) % 16 == 0 && aligned16(q) &&
                    aligned16(kv) && aligned16(out),
                "msn_cls_attention_fwd: rows must be 16-byte aligned (ldq %lld, ldkv %lld, ldo %lld)", (long long)ldq, (long long)ldkv,
                (long long)ldo);
    ClsArgs a = {};
    a.q = q, a.ldq = ldq, a.kv = kv, a.ldkv = ldkv, a.B = B, a.H = H, a.T = T, a.scale = scale, a.out = out, a.ldo = ldo, a.P = probs;
    const dim3 grid((unsigned)((B * H + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kv_bf16) hipLaunchKernelGGL(cls_attn_fwd_kernel<u16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(cls_attn_fwd_kernel<float>, grid, block, 0, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_cls_attention_bwd(const float* q, int64_t ldq, const void* kv, int64_t ldkv, int kv_bf16, int B, int H, int T,
                                     int head_dim, float scale, const float* out, int64_t ldo, const float* probs, const float* dout,
                                     int64_t ldd, float* dq, int64_t lddq, void* dkv, int64_t lddkv, msn_stream_t stream) {
    MSN_REQUIRE(q && kv && out && probs && dout && dq && dkv && B > 0 && H > 0, "msn_cls_attention_bwd: empty operand");
    MSN_REQUIRE(msn_cls_attention_supported(T, head_dim), "msn_cls_attention_bwd: head_dim %d (must be 64), T = %d (1 .. 256)", head_dim, T);
    const int e = H * HD, es = kv_bf16 ? 2 : 4;
    MSN_REQUIRE(ldq >= e && ldo >= e && ldd >= e && lddq >= e && ldkv >= 2 * e && lddkv >= 2 * e && ldq % 4 == 0 && ldo % 4 == 0 &&
                    ldd % 4 == 0 && lddq % 4 == 0 && (ldkv * es) % 16 == 0 && (lddkv * es) % 16 == 0 && aligned16(q) && aligned16(kv) &&
                    aligned16(out) && aligned16(dout) && aligned16(dq) && aligned16(dkv),
                "msn_cls_attention_bwd: rows must be 16-byte aligned");
    ClsArgs a = {};
    a.q = q, a.ldq = ldq, a.kv = kv, a.ldkv = ldkv, a.B = B, a.H = H, a.T = T, a.scale = scale;
    a.out = const_cast<float*>(out), a.ldo = ldo, a.P = const_cast<float*>(probs), a.dout = dout, a.ldd = ldd, a.dq = dq, a.lddq = lddq;
    a.dkv = dkv, a.lddkv = lddkv;
    const dim3 grid((unsigned)((B * H + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (kv_bf16) hipLaunchKernelGGL(cls_attn_bwd_kernel<u16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(cls_attn_bwd_kernel<float>, grid, block, 0, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
