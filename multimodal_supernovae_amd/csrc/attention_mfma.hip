// Matrix-core attention (sequences up to 256 tokens; head_dim 16..64 in multiples of 16: the ViT image
// tower; head_dim 4 / 8 / 12: the light-curve transformer, run as 16-wide heads whose missing columns
// are zeros in LDS / registers only).  Exact fp32 on v_mfma_f32_16x16x4_f32.
//
// One workgroup per (batch, head); one wave per 16-row tile of the "fixed" operand, every wave
// sweeping the 16-row tiles of the "streamed" operand, which sits in LDS for the whole workgroup:
//   forward / dQ kernel : fixed = a query tile (fragments in registers), streamed = K and V
//   dK,dV kernel        : fixed = a key tile,                            streamed = Q and dO
// Every product is oriented so that the accumulator of one MFMA chain is directly the A operand of
// the next chain (no transposes, no LDS round trip for P or dS):
//   scores   T[srow][fcol] = sum_d Streamed[srow][d] * Fixed[fcol][d]   (A = LDS rows, B = registers)
//            -> lane (c = lane & 15, g = lane >> 4) holds T[4g + r][c], r = 0..3
//   outputs  O[fcol][d]   += sum_srow T[srow][fcol] * Streamed2[srow][d] (A = T registers: the MFMA k
//            index is the lane group g, which is exactly how T is spread; B = one LDS row per g)
// Softmax statistics of a query are lane-local in the forward / dQ kernels (query = lane column;
// 4 registers per tile + two cross-group shuffles per reduction).  The forward makes two passes over
// the key tiles (row maximum, then exponentials + P.V), recomputing the cheap 16x16 score tiles
// instead of holding T/16 accumulator tiles in registers.
// K-order inside a d-step is free, so lane group g owns d = 16x + 4g .. +3 and fetches them with ONE
// 16-byte LDS read per 4 MFMAs (rows padded by 4 floats).
#include <algorithm>
#include <math.h>

#include "msn_common.h"

namespace msn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kFill = -1e7f;  // ref transformer_utils.py:77

struct MAttn {
    const float* q; const float* k; const float* v; const float* o; const float* dout;
    float* out; float* dq; float* dk; float* dv;
    const uint8_t* mask;      // [B][Tk] or null
    float* lse;               // [B][H][Tq][2] = (row max, log sum)
    float* delta;             // [B][H][Tq]
    int64_t ldq, ldk, ldv, ldo, ldd, lddq, lddk, lddv;
    int64_t q_bs, k_bs, v_bs, o_bs, d_bs, dq_bs, dk_bs, dv_bs;
    int B, H, Tq, Tk, hd;
    float scale;
};

// rows [0, T) of src (row stride ld, columns col0..col0+HD-1) -> LDS [TP][HD + 4], zero rows beyond T
template <int HD>
__device__ __forceinline__ void stage(float* dst, const float* __restrict__ src, int64_t ld, int col0, int T, int TP,
                                      int hd) {
    constexpr int LS = HD + 4, Q4 = HD / 4, U = 4;   // U independent 16-byte loads in flight per thread
    const int total = TP * Q4;
    for (int base = threadIdx.x; base < total; base += U * blockDim.x) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * blockDim.x;
            const int r = idx / Q4, c = 4 * (idx % Q4);
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < total && r < T && c < hd) v[u] = *reinterpret_cast<const float4*>(src + (int64_t)r * ld + col0 + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * blockDim.x;
            if (idx < total) *reinterpret_cast<float4*>(dst + (idx / Q4) * LS + 4 * (idx % Q4)) = v[u];
        }
    }
}
// fragments of one 16-row tile held by this wave as the B operand: row = lane & 15, d = 16x + 4g + j
template <int HD>
__device__ __forceinline__ void load_frags(float4 (&f)[HD / 16], const float* __restrict__ src, int64_t ld, int col0,
                                           int row, int T, int g, float mul, int hd) {
    const float* p = src + (int64_t)(row < T ? row : T - 1) * ld + col0;
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const int d0 = 16 * x + 4 * g;
        float4 v = *reinterpret_cast<const float4*>(p + (d0 < hd ? d0 : 0));
        if (row >= T || d0 >= hd) v = make_float4(0.f, 0.f, 0.f, 0.f);
        f[x] = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
    }
}
// acc[r] = T[tile row 4g + r][fixed row c] = sum_d tile[4g + r][d] * fixed[c][d]   (A = LDS tile rows, B = fragments)
template <int HD>
__device__ __forceinline__ f32x4 score16(const float* tile, const float4 (&bf)[HD / 16], int c, int g) {
    constexpr int LS = HD + 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const float4 a = *reinterpret_cast<const float4*>(tile + c * LS + 16 * x + 4 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bf[x].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bf[x].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bf[x].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bf[x].w, acc, 0, 0, 0);
    }
    return acc;
}
// out[t][r'] (row 4g + r' of the fixed tile, column 16t + c) += sum over the tile's 16 rows of
// a[row][fixed col] * tile[row][16t + c];  a[r] belongs to tile row 4g + r (the MFMA k index is g)
template <int HD>
__device__ __forceinline__ void accum16(const f32x4& a, const float* tile, f32x4 (&out)[HD / 16], int c, int g) {
    constexpr int LS = HD + 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float* row = tile + (4 * g + r) * LS + c;
#pragma unroll
        for (int t = 0; t < HD / 16; ++t) out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], row[16 * t], out[t], 0, 0, 0);
    }
}
__device__ __forceinline__ float group_max4(float v) {  // over the 4 lane groups (same lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------ forward
template <int HD>
__global__ __launch_bounds__(1024) void mattn_fwd_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TPk = (p.Tk + 15) / 16 * 16;
    float* Ks = smem;
    float* Vs = smem + (size_t)TPk * LS;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + (size_t)TPk * LS);
    const int b = blockIdx.x / p.H, hh = blockIdx.x % p.H, col0 = hh * p.hd;
    stage<HD>(Ks, p.k + (int64_t)b * p.k_bs, p.ldk, col0, p.Tk, TPk, p.hd);
    stage<HD>(Vs, p.v + (int64_t)b * p.v_bs, p.ldv, col0, p.Tk, TPk, p.hd);
    for (int j = threadIdx.x; j < TPk; j += blockDim.x) Ms[j] = (j < p.Tk && p.mask) ? p.mask[(int64_t)b * p.Tk + j] : 1;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = wave * 16, qrow = q0 + c;
    float4 qf[DT];
    load_frags<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.scale, p.hd);
    const int nkt = TPk / 16;

    float m = -INFINITY;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    // key tiles whose scores fit in registers (4 per tile) next to the output accumulators: sequences up to 256
    // tokens for 16-wide heads (the reference's spectrum tower: 220 tokens), 128 tokens for wider heads
    constexpr int KEEP = HD <= 16 ? 16 : 8;
    if (nkt <= KEEP) {
        // one pass: every score tile is computed once and kept (ViT-S/8 at 64x64: 65 tokens -> 5 tiles)
        f32x4 sc[KEEP];
#pragma unroll
        for (int kt = 0; kt < KEEP; ++kt) {
            if (kt < nkt) {
                sc[kt] = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * g + r;
                    sc[kt][r] = key < p.Tk ? (Ms[key] ? sc[kt][r] : kFill) : -INFINITY;
                    m = fmaxf(m, sc[kt][r]);
                }
            }
        }
        m = group_max4(m);
#pragma unroll
        for (int kt = 0; kt < KEEP; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sc[kt][r] - m);   // exp(-inf) = 0 for the padded keys
                    sc[kt][r] = e;
                    l += e;
                }
                accum16<HD>(sc[kt], Vs + kt * 16 * LS, o, c, g);
            }
        }
    } else {
        // two passes over the key tiles (row maximum, then exponentials + P.V), recomputing the 16x16 score tiles
        for (int kt = 0; kt < nkt; ++kt) {
            const f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                if (key < p.Tk) m = fmaxf(m, Ms[key] ? s[r] : kFill);
            }
        }
        m = group_max4(m);
        for (int kt = 0; kt < nkt; ++kt) {
            f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                const float e = key < p.Tk ? __expf((Ms[key] ? s[r] : kFill) - m) : 0.f;
                s[r] = e;
                l += e;
            }
            accum16<HD>(s, Vs + kt * 16 * LS, o, c, g);
        }
    }
    l = group_sum4(l);
    // o[t][r] belongs to query q0 + 4g + r: fetch that query's normaliser from its column-owner lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lq = __shfl(l, 4 * g + r, 64);
        const int q = q0 + 4 * g + r;
        if (q < p.Tq) {
            float* op = p.out + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + col0 + c;
            const float inv = 1.f / lq;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) op[16 * t] = o[t][r] * inv;
        }
    }
    if (g == 0 && qrow < p.Tq) {
        float* st = p.lse + 2 * (((int64_t)b * p.H + hh) * p.Tq + qrow);
        st[0] = m;
        st[1] = __logf(l);
    }
}

// ------------------------------------------------------------------------------- backward: dQ, delta
template <int HD>
__global__ __launch_bounds__(1024) void mattn_bwd_dq_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TPk = (p.Tk + 15) / 16 * 16;
    float* Ks = smem;
    float* Vs = smem + (size_t)TPk * LS;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + (size_t)TPk * LS);
    const int b = blockIdx.x / p.H, hh = blockIdx.x % p.H, col0 = hh * p.hd;
    stage<HD>(Ks, p.k + (int64_t)b * p.k_bs, p.ldk, col0, p.Tk, TPk, p.hd);
    stage<HD>(Vs, p.v + (int64_t)b * p.v_bs, p.ldv, col0, p.Tk, TPk, p.hd);
    for (int j = threadIdx.x; j < TPk; j += blockDim.x) Ms[j] = (j < p.Tk && p.mask) ? p.mask[(int64_t)b * p.Tk + j] : 1;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = wave * 16, qrow = q0 + c;
    const bool q_ok = qrow < p.Tq;
    float4 qf[DT], df[DT], of[DT];
    load_frags<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.scale, p.hd);
    load_frags<HD>(df, p.dout + (int64_t)b * p.d_bs, p.ldd, col0, qrow, p.Tq, g, 1.f, p.hd);
    load_frags<HD>(of, p.o + (int64_t)b * p.o_bs, p.ldo, col0, qrow, p.Tq, g, 1.f, p.hd);
    float delta = 0.f;
#pragma unroll
    for (int x = 0; x < DT; ++x)
        delta += df[x].x * of[x].x + df[x].y * of[x].y + df[x].z * of[x].z + df[x].w * of[x].w;
    delta = group_sum4(delta);
    const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + (q_ok ? qrow : 0);
    const float lm = q_ok ? p.lse[2 * stat] : 0.f, ll = q_ok ? p.lse[2 * stat + 1] : 0.f;
    if (g == 0 && q_ok) p.delta[stat] = delta;

    f32x4 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = TPk / 16;
    for (int kt = 0; kt < nkt; ++kt) {
        const f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
        const f32x4 dp = score16<HD>(Vs + kt * 16 * LS, df, c, g);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            // a padded key contributes nothing; a masked key has a constant score: no gradient to q / k
            const bool live = q_ok && key < p.Tk && Ms[key];
            ds[r] = live ? __expf((s[r] - lm) - ll) * (dp[r] - delta) : 0.f;
        }
        accum16<HD>(ds, Ks + kt * 16 * LS, dq, c, g);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        if (q < p.Tq) {
            float* op = p.dq + (int64_t)b * p.dq_bs + (int64_t)q * p.lddq + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) op[16 * t] = dq[t][r] * p.scale;
        }
    }
}

// ------------------------------------------------------------------------------- backward: dK, dV
template <int HD>
__global__ __launch_bounds__(1024) void mattn_bwd_dkv_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TPq = (p.Tq + 15) / 16 * 16;
    float* Qs = smem;
    float* Ds = smem + (size_t)TPq * LS;
    float* Lm = Ds + (size_t)TPq * LS;
    float* Ll = Lm + TPq;
    float* Dl = Ll + TPq;
    const int b = blockIdx.x / p.H, hh = blockIdx.x % p.H, col0 = hh * p.hd;
    stage<HD>(Qs, p.q + (int64_t)b * p.q_bs, p.ldq, col0, p.Tq, TPq, p.hd);
    stage<HD>(Ds, p.dout + (int64_t)b * p.d_bs, p.ldd, col0, p.Tq, TPq, p.hd);
    for (int t = threadIdx.x; t < TPq; t += blockDim.x) {
        const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + (t < p.Tq ? t : 0);
        Lm[t] = t < p.Tq ? p.lse[2 * stat] : INFINITY;     // +inf: a padded query row gets p = exp(-inf) = 0
        Ll[t] = t < p.Tq ? p.lse[2 * stat + 1] : 0.f;
        Dl[t] = t < p.Tq ? p.delta[stat] : 0.f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int k0 = wave * 16, krow = k0 + c;
    float4 kf[DT], vf[DT];
    load_frags<HD>(kf, p.k + (int64_t)b * p.k_bs, p.ldk, col0, krow, p.Tk, g, p.scale, p.hd);
    load_frags<HD>(vf, p.v + (int64_t)b * p.v_bs, p.ldv, col0, krow, p.Tk, g, 1.f, p.hd);
    const bool keep = krow < p.Tk && (p.mask ? p.mask[(int64_t)b * p.Tk + krow] != 0 : true);

    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nqt = TPq / 16;
    for (int qt = 0; qt < nqt; ++qt) {
        const f32x4 s = score16<HD>(Qs + qt * 16 * LS, kf, c, g);     // rows = queries, col = this lane's key
        const f32x4 dp = score16<HD>(Ds + qt * 16 * LS, vf, c, g);
        f32x4 pr, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * qt + 4 * g + r;
            const float e = __expf(((keep ? s[r] : kFill) - Lm[q]) - Ll[q]);
            pr[r] = krow < p.Tk ? e : 0.f;
            ds[r] = keep ? e * (dp[r] - Dl[q]) : 0.f;
        }
        accum16<HD>(pr, Ds + qt * 16 * LS, dv, c, g);
        accum16<HD>(ds, Qs + qt * 16 * LS, dk, c, g);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + 4 * g + r;
        if (k < p.Tk) {
            float* ok = p.dk + (int64_t)b * p.dk_bs + (int64_t)k * p.lddk + col0 + c;
            float* ov = p.dv + (int64_t)b * p.dv_bs + (int64_t)k * p.lddv + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) {
                    ok[16 * t] = dk[t][r] * p.scale;
                    ov[16 * t] = dv[t][r];
                }
        }
    }
}

template <typename K>
static int launch_big_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, const MAttn& a) {
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            set_error("attention: cannot reserve %zu bytes of LDS", lds);
            return MSN_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// Is the matrix-core path applicable?  (16-B aligned operands, head width 4/8/12/16/32/48/64, <= 256 tokens.)
static int padded_hd(int hd) { return hd < 16 ? 16 : hd; }
bool mattn_applicable(const MAttn& a) {
    if (!(a.hd % 16 == 0 || (a.hd < 16 && a.hd % 4 == 0)) || a.hd > 64 || a.hd < 4 || a.Tq > 256 || a.Tk > 256 ||
        a.q_bs == 0)
        return false;
    if ((int64_t)a.B * a.H > 0x7fffffffLL) return false;
    const int64_t lds[] = {a.ldq, a.ldk, a.ldv, a.q_bs, a.k_bs, a.v_bs};
    for (int64_t v : lds)
        if (v % 4 != 0) return false;
    const void* ptrs[] = {a.q, a.k, a.v};
    for (const void* ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) & 15) return false;
    return true;
}

#define MSN_MATTN_DISPATCH(KERNEL, ...)                                                    \
    switch (padded_hd(a.hd)) {                                                                      \
        case 16: rc = launch_big_lds(KERNEL<16>, __VA_ARGS__); break;                      \
        case 32: rc = launch_big_lds(KERNEL<32>, __VA_ARGS__); break;                      \
        case 48: rc = launch_big_lds(KERNEL<48>, __VA_ARGS__); break;                      \
        default: rc = launch_big_lds(KERNEL<64>, __VA_ARGS__); break;                      \
    }

int mattn_forward(const MAttn& a, hipStream_t st) {
    const int TPk = (a.Tk + 15) / 16 * 16, nq = (a.Tq + 15) / 16;
    const size_t lds = sizeof(float) * 2 * (size_t)TPk * (padded_hd(a.hd) + 4) + (size_t)TPk;
    int rc;
    MSN_MATTN_DISPATCH(mattn_fwd_kernel, dim3(a.B * a.H), dim3(64 * nq), lds, st, a)
    return rc;
}

int mattn_backward(const MAttn& a, hipStream_t st) {
    const int TPk = (a.Tk + 15) / 16 * 16, TPq = (a.Tq + 15) / 16 * 16;
    int rc;
    const int64_t al[] = {a.ldd, a.d_bs, a.ldo, a.o_bs};
    bool ok = ((reinterpret_cast<uintptr_t>(a.dout) | reinterpret_cast<uintptr_t>(a.o)) & 15) == 0;
    for (int64_t v : al) ok = ok && (v % 4 == 0);
    if (!ok) {
        set_error("attention backward: out / dout must be 16-byte aligned with strides %% 4 == 0");
        return MSN_ERR_SHAPE;
    }
    {
        const size_t lds = sizeof(float) * 2 * (size_t)TPk * (padded_hd(a.hd) + 4) + (size_t)TPk;
        MSN_MATTN_DISPATCH(mattn_bwd_dq_kernel, dim3(a.B * a.H), dim3(64 * (TPq / 16)), lds, st, a)
        if (rc != MSN_OK) return rc;
    }
    {
        const size_t lds = sizeof(float) * (2 * (size_t)TPq * (padded_hd(a.hd) + 4) + 3 * (size_t)TPq);
        MSN_MATTN_DISPATCH(mattn_bwd_dkv_kernel, dim3(a.B * a.H), dim3(64 * (TPk / 16)), lds, st, a)
    }
    return rc;
}

}  // namespace msn
