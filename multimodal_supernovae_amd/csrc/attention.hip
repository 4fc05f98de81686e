// Fused multi-head attention forward / backward in exact fp32, scores never materialised.
//
// Reference: SelfAttention.forward, src/transformer_utils.py:36-89 -- per head softmax(mask(q k^T *
// scale)) v with  scale = 1/sqrt(EMB) (the full embedding width, :63-64), key-padding positions
// REPLACED by -1e7 (:73-77: a finite fill, so a fully padded sample attends uniformly), padded
// queries still produce outputs.  The same kernels serve the 1-query nn.MultiheadAttention
// pooling (:240-246; Tq = 1, no mask, scale = 1/sqrt(head_dim)) and the build-defined ViT blocks.
//
// The reference-native heads are 8 and 16 wide (emb/heads = 64/8, 32/2): MFMA tiles would be mostly
// padding, and the f32 matrix-core rate equals the f32 vector rate anyway, so this is a VALU
// kernel: one lane owns one query row (q, o and the online-softmax state in registers), the
// keys / values of the (batch, head) stream through LDS in 32-KB tiles and are read as wave-wide
// broadcasts (all lanes read the same key row: conflict-free by construction).  The win over the
// reference is the B*h*T*T score / probability tensors that never touch HBM (328 MB per layer for
// the light-curve tower at B = 256).  Backward recomputes the probabilities from the saved
// log-sum-exp: one kernel with a lane per query (dQ, and delta = <dO, O>), one with a lane per key
// (dK, dV) -- no atomics, deterministic.
#include <algorithm>
#include <cstdlib>
#include <math.h>

#include "msn_common.h"
#include "attention_args.h"

namespace msn {

constexpr float kMaskFill = -1e7f;
// Pairs of head-dimension elements: spelling the inner products on 2-vectors makes the compiler emit packed fp32
// FMAs (v_pk_fma_f32: two lanes of arithmetic per instruction slot), which it does not reliably do from scalar code.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int S>
__device__ __forceinline__ void load_row2(f2 (&dst)[S / 2], const float* __restrict__ lds_row) {   // LDS row -> pairs
#pragma unroll
    for (int d = 0; d < S / 2; ++d) dst[d] = f2{lds_row[2 * d], lds_row[2 * d + 1]};
}
template <int S>
__device__ __forceinline__ float dot2(const f2 (&a)[S / 2], const f2 (&b)[S / 2]) {
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int d = 0; d < S / 2; ++d) acc = fma2(a[d], b[d], acc);
    return acc.x + acc.y;
}
template <int S>
__device__ __forceinline__ void axpy2(f2 (&y)[S / 2], float a, const f2 (&x)[S / 2]) {   // y += a * x
    const f2 aa = {a, a};
#pragma unroll
    for (int d = 0; d < S / 2; ++d) y[d] = fma2(aa, x[d], y[d]);
}
constexpr int KB = 8;  // keys per online-softmax micro-step

struct AttnArgs {
    const float* q; const float* k; const float* v;   // [B][T][ld*], head hh at columns hh*s .. hh*s+s-1
    float* o;                                         // [B][Tq][ldo]
    const uint8_t* mask;                              // [B][Tk] or null
    float* lse;                                       // [B][H][Tq][2] = (running max, log of the sum)
    int64_t q_bstride, k_bstride, v_bstride, o_bstride;   // batch strides in elements (q: 0 = shared query)
    int64_t ldq, ldk, ldv, ldo;
    int B, H, Tq, Tk, s;
    int tile_rows;                                    // rows of the streamed operand per LDS tile
    float scale;
    // backward only
    const float* dout; int64_t ldd, d_bstride;
    float* delta;                                     // [B][H][Tq]
    float* dq; float* dk; float* dv;                  // same layout as q / k / v
    int64_t lddq, lddk, lddv, dq_bstride, dk_bstride, dv_bstride;
};

// rows r0..r0+n-1 of a [T][ld] matrix, columns col0..col0+s-1 -> LDS [n][S] (zero padded to S)
// rows n .. npad-1 (the tile is padded to a whole number of inner-loop chunks) are zero-filled
template <int S>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, int64_t ld, int col0, int s,
                                           int r0, int n, bool vec_ok, int npad) {
    constexpr int Q4 = S / 4;
    for (int idx = threadIdx.x; idx < npad * Q4; idx += blockDim.x) {
        const int r = idx / Q4, c = 4 * (idx % Q4);
        const float* p = src + (int64_t)(r0 + r) * ld + col0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= n) {
        } else if (vec_ok && c + 3 < s) v = *reinterpret_cast<const float4*>(p);
        else {
            if (c < s) v.x = p[0];
            if (c + 1 < s) v.y = p[1];
            if (c + 2 < s) v.z = p[2];
            if (c + 3 < s) v.w = p[3];
        }
        *reinterpret_cast<float4*>(dst + r * S + c) = v;
    }
}
template <int S>
__device__ __forceinline__ void load_vec(float (&dst)[S], const float* __restrict__ p, int s, bool on) {
#pragma unroll
    for (int d = 0; d < S; ++d) dst[d] = (on && d < s) ? p[d] : 0.f;
}
// Exact head width + 16-byte aligned rows: S / 4 unconditional 16-byte loads (the caller passes a valid, clamped row
// address) and a select afterwards.  The guarded scalar form above costs one branch and one memory round trip per
// element -- with one wave per workgroup that prologue, not the key loop, set the kernels' time.
template <int S>
__device__ __forceinline__ void load_vec4(float (&dst)[S], const float* __restrict__ p, bool on) {
#pragma unroll
    for (int c = 0; c < S / 4; ++c) {
        const float4 v = reinterpret_cast<const float4*>(p)[c];
        dst[4 * c] = on ? v.x : 0.f;
        dst[4 * c + 1] = on ? v.y : 0.f;
        dst[4 * c + 2] = on ? v.z : 0.f;
        dst[4 * c + 3] = on ? v.w : 0.f;
    }
}
template <int S>
__device__ __forceinline__ void load_row_any(float (&dst)[S], const float* __restrict__ p, int s, bool on, bool vec) {
    if (vec) load_vec4<S>(dst, p, on);
    else load_vec<S>(dst, p, s, on);
}
template <int S>
__device__ __forceinline__ void store_row_any(float* __restrict__ p, const float (&src)[S], int s, bool vec) {
    if (vec) {
#pragma unroll
        for (int c = 0; c < S / 4; ++c)
            reinterpret_cast<float4*>(p)[c] = make_float4(src[4 * c], src[4 * c + 1], src[4 * c + 2], src[4 * c + 3]);
    } else {
#pragma unroll
        for (int d = 0; d < S; ++d)
            if (d < s) p[d] = src[d];
    }
}
__device__ __forceinline__ bool vec4_ok(const float* p, int64_t ld, int64_t bstride, int s) {
    return (s % 4 == 0) && (ld % 4 == 0) && (bstride % 4 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
}

// (batch, head) of a workgroup.  Workgroups go to the 8 XCDs round-robin by their linear id, and a narrow head reads
// only s * 4 bytes (32 for the reference's 8-wide heads) of every 128-byte line of the q|k|v rows: with the natural
// order (head fastest) XCD x serves head x of EVERY sample and each line is fetched from HBM by up to four XCDs
// (rocprofv3 FETCH_SIZE: 3.3-3.7 x the algorithmic bytes on the light-curve tower).  With one workgroup per (b, h)
// and B % 8 == 0 the ids are re-read so that the H heads of a sample are consecutive workgroups of ONE XCD: the first
// toucher fetches the line into that XCD's L2, the other heads hit it.
__device__ __forceinline__ void locate_head(int& b, int& hh) {
    b = blockIdx.z, hh = blockIdx.y;
    if (gridDim.x == 1 && (gridDim.z & 7) == 0) {
        const unsigned id = blockIdx.y + gridDim.y * blockIdx.z;   // linear workgroup id (x is 0)
        const unsigned xcd = id & 7, j = id >> 3;                  // j-th workgroup of this XCD
        b = (int)((j / gridDim.y) * 8 + xcd);
        hh = (int)(j % gridDim.y);
    }
}

// R = query (or key) rows per lane.  Every key row is read from LDS as a wave-wide broadcast, 8 LDS cycles per 16
// bytes whatever the number of distinct addresses -- with one row per lane those 4 reads per key bound the kernel
// (328 M query-key pairs per light-curve layer: 340 us of LDS time per CU against 160 us of vector ALU).  A lane that
// owns R consecutive rows re-uses each broadcast R times; narrow heads fit that in registers (<= 8 floats: R = 4,
// 16 floats: R = 2).
template <int S, int R>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
    // LDS sized by the launch: KTILE = min(4096 / S, keys rounded up to KB) rows each of K and V + one float per key:
    // 0 = live key, -1e7 = padded-out key (masked_fill value), -inf = beyond the sequence (tile padding)
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int KTILE = p.tile_rows;
    float* Ks = attn_smem;
    float* Vs = attn_smem + KTILE * S;
    float* Fs = Vs + KTILE * S;
    int b, hh;
    locate_head(b, hh);
    const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * R;
    const int col0 = hh * p.s;
    const float* kb = p.k + (int64_t)b * p.k_bstride;
    const float* vb = p.v + (int64_t)b * p.v_bstride;
    const bool kvec = vec4_ok(p.k, p.ldk, p.k_bstride, p.s), vvec = vec4_ok(p.v, p.ldv, p.v_bstride, p.s);
    const bool qfast = p.s == S && vec4_ok(p.q, p.ldq, p.q_bstride, p.s);
    const bool ofast = p.s == S && vec4_ok(p.o, p.ldo, p.o_bstride, p.s);

    // scores are kept in log2 units (log2(e) folded into the query scale): v_exp_f32 is 2^x, so the exponentials need
    // no multiply; the masked-fill value is scaled the same way and the stored maximum converted back (exactly -1e7 when
    // every key is masked: one fp32 ulp is 1.0 there)
    constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
    const float qscale = p.scale * kLog2e;
    f2 q[R][S / 2], o[R][S / 2];
    float m[R], l[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool on = i0 + r < p.Tq;
        float qs[S];
        load_row_any<S>(qs, p.q + (int64_t)b * p.q_bstride + (int64_t)(on ? i0 + r : 0) * p.ldq + col0, p.s, on, qfast);
#pragma unroll
        for (int d = 0; d < S / 2; ++d) {
            q[r][d] = f2{qs[2 * d] * qscale, qs[2 * d + 1] * qscale};
            o[r][d] = f2{0.f, 0.f};
        }
        m[r] = -INFINITY, l[r] = 0.f;
    }

    for (int k0 = 0; k0 < p.Tk; k0 += KTILE) {
        const int nt = min(KTILE, p.Tk - k0), ntp = (nt + KB - 1) / KB * KB;
        __syncthreads();
        stage_rows<S>(Ks, kb, p.ldk, col0, p.s, k0, nt, kvec, ntp);
        stage_rows<S>(Vs, vb, p.ldv, col0, p.s, k0, nt, vvec, ntp);
        for (int j = threadIdx.x; j < ntp; j += blockDim.x)
            Fs[j] = j >= nt ? -INFINITY : ((p.mask && !p.mask[(int64_t)b * p.Tk + k0 + j]) ? kMaskFill * kLog2e : 0.f);
        __syncthreads();
        // whole chunks of KB keys, no per-key control flow: the compiler interleaves the KB x R independent dot
        // products; a key beyond the sequence scores -inf (probability exactly 0)
        for (int j0 = 0; j0 < ntp; j0 += KB) {
            float sc[R][KB], mx[R];
#pragma unroll
            for (int r = 0; r < R; ++r) mx[r] = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < KB; ++jj) {
                f2 kv[S / 2];
                load_row2<S>(kv, Ks + (j0 + jj) * S);
                const float fill = Fs[j0 + jj];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float a = dot2<S>(q[r], kv);
                    a = fill == 0.f ? a : fill;
                    sc[r][jj] = a;
                    mx[r] = fmaxf(mx[r], a);
                }
            }
            float mn[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mn[r] = fmaxf(m[r], mx[r]);
                const float alpha = __builtin_amdgcn_exp2f(m[r] - mn[r]);
                l[r] *= alpha;
#pragma unroll
                for (int d = 0; d < S / 2; ++d) o[r][d] *= alpha;
                m[r] = mn[r];
            }
#pragma unroll
            for (int jj = 0; jj < KB; ++jj) {
                f2 vv[S / 2];
                load_row2<S>(vv, Vs + (j0 + jj) * S);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float pj = __builtin_amdgcn_exp2f(sc[r][jj] - mn[r]);
                    l[r] += pj;
                    axpy2<S>(o[r], pj, vv);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = i0 + r;
        if (i < p.Tq) {
            const float inv = 1.f / l[r];
            float* op = p.o + (int64_t)b * p.o_bstride + (int64_t)i * p.ldo + col0;
            float os[S];
#pragma unroll
            for (int d = 0; d < S; ++d) os[d] = (d & 1 ? o[r][d / 2].y : o[r][d / 2].x) * inv;
            store_row_any<S>(op, os, p.s, ofast);
            // (max, log-sum) kept apart: with every key padded the max is -1e7, where one fp32 ulp is 1.0
            float* st = p.lse + 2 * (((int64_t)b * p.H + hh) * p.Tq + i);
            st[0] = m[r] < -1e6f ? kMaskFill : m[r] * kLn2;
            st[1] = __logf(l[r]);
        }
    }
}

// lane = R queries: dQ_i = scale * sum_j dS_ij k_j,  dS_ij = p_ij (dO_i . v_j - delta_i) (0 through masked keys)
template <int S, int R>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
    // LDS sized by the launch: KTILE = min(4096 / S, keys rounded up to 8) rows each of K and V + the mask bytes
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int KTILE = p.tile_rows;
    float* Ks = attn_smem;
    float* Vs = attn_smem + KTILE * S;
    float* Fs = Vs + KTILE * S;          // 1 = live key, 0 = masked or beyond the sequence (no gradient through it)
    int b, hh;
    locate_head(b, hh);
    const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * R;
    const int col0 = hh * p.s;
    const float* kb = p.k + (int64_t)b * p.k_bstride;
    const float* vb = p.v + (int64_t)b * p.v_bstride;
    const bool kvec = vec4_ok(p.k, p.ldk, p.k_bstride, p.s), vvec = vec4_ok(p.v, p.ldv, p.v_bstride, p.s);

    const bool qfast = p.s == S && vec4_ok(p.q, p.ldq, p.q_bstride, p.s);
    const bool dfast = p.s == S && vec4_ok(p.dout, p.ldd, p.d_bstride, p.s);
    const bool ofast = p.s == S && vec4_ok(p.o, p.ldo, p.o_bstride, p.s);
    const bool gfast = p.s == S && vec4_ok(p.dq, p.lddq, p.dq_bstride, p.s);
    f2 q[R][S / 2], dO[R][S / 2], dq[R][S / 2];
    float delta[R], lse_m[R], lse_l[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool on = i0 + r < p.Tq;
        const int ii = on ? i0 + r : 0;
        float qs[S], ds_[S], os[S];
        load_row_any<S>(qs, p.q + (int64_t)b * p.q_bstride + (int64_t)ii * p.ldq + col0, p.s, on, qfast);
        load_row_any<S>(ds_, p.dout + (int64_t)b * p.d_bstride + (int64_t)ii * p.ldd + col0, p.s, on, dfast);
        load_row_any<S>(os, p.o + (int64_t)b * p.o_bstride + (int64_t)ii * p.ldo + col0, p.s, on, ofast);
        delta[r] = 0.f;
#pragma unroll
        for (int d = 0; d < S; ++d) delta[r] = fmaf(ds_[d], os[d], delta[r]);
#pragma unroll
        for (int d = 0; d < S / 2; ++d) {
            q[r][d] = f2{qs[2 * d] * p.scale, qs[2 * d + 1] * p.scale};
            dO[r][d] = f2{ds_[2 * d], ds_[2 * d + 1]};
            dq[r][d] = f2{0.f, 0.f};
        }
        const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + ii;
        // a row beyond Tq keeps p = exp(-inf) = 0
        lse_m[r] = on ? p.lse[2 * stat] : INFINITY;
        lse_l[r] = on ? p.lse[2 * stat + 1] : 0.f;
        if (on) p.delta[stat] = delta[r];
    }

    constexpr int KC = 4;   // keys per unrolled chunk
    for (int k0 = 0; k0 < p.Tk; k0 += KTILE) {
        const int nt = min(KTILE, p.Tk - k0), ntp = (nt + KC - 1) / KC * KC;
        __syncthreads();
        stage_rows<S>(Ks, kb, p.ldk, col0, p.s, k0, nt, kvec, ntp);
        stage_rows<S>(Vs, vb, p.ldv, col0, p.s, k0, nt, vvec, ntp);
        for (int j = threadIdx.x; j < ntp; j += blockDim.x)
            Fs[j] = (j < nt && (!p.mask || p.mask[(int64_t)b * p.Tk + k0 + j])) ? 1.f : 0.f;
        __syncthreads();
        for (int j0 = 0; j0 < ntp; j0 += KC) {
#pragma unroll
            for (int jj = 0; jj < KC; ++jj) {
                f2 kv[S / 2], vv[S / 2];
                load_row2<S>(kv, Ks + (j0 + jj) * S);
                load_row2<S>(vv, Vs + (j0 + jj) * S);
                const float live = Fs[j0 + jj];   // 0 for a masked / padded key: no gradient flows through it
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float a = dot2<S>(q[r], kv), dp = dot2<S>(dO[r], vv);
                    // A probability never exceeds 1, so the exponent is clamped at 0: nothing changes for live keys,
                    // and a masked key of a fully padded sample (row max -1e7, raw score here) cannot overflow
                    // into inf * 0.  The product with `live` keeps the chunk free of control flow.
                    const float ds = live * (__expf(fminf((a - lse_m[r]) - lse_l[r], 0.f)) * (dp - delta[r]));
                    axpy2<S>(dq[r], ds, kv);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = i0 + r;
        if (i < p.Tq) {
            float* out = p.dq + (int64_t)b * p.dq_bstride + (int64_t)i * p.lddq + col0;
            float gs[S];
#pragma unroll
            for (int d = 0; d < S; ++d) gs[d] = (d & 1 ? dq[r][d / 2].y : dq[r][d / 2].x) * p.scale;
            store_row_any<S>(out, gs, p.s, gfast);
        }
    }
}

// lane = R keys: dV_j = sum_i p_ij dO_i ;  dK_j = scale * sum_i dS_ij q_i  (0 for a padded key)
template <int S, int R>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float attn_smem[];
    const int QTILE = p.tile_rows;
    float* Qs = attn_smem;
    float* Ds = attn_smem + QTILE * S;
    float* Lm = Ds + QTILE * S;
    float* Ll = Lm + QTILE;
    float* Dl = Ll + QTILE;
    int b, hh;
    locate_head(b, hh);
    const int j0 = (blockIdx.x * blockDim.x + threadIdx.x) * R;
    const int col0 = hh * p.s;
    const float* qb = p.q + (int64_t)b * p.q_bstride;
    const float* db = p.dout + (int64_t)b * p.d_bstride;
    const bool qvec = vec4_ok(p.q, p.ldq, p.q_bstride, p.s), dvec = vec4_ok(p.dout, p.ldd, p.d_bstride, p.s);

    const bool kfast = p.s == S && vec4_ok(p.k, p.ldk, p.k_bstride, p.s);
    const bool vfast = p.s == S && vec4_ok(p.v, p.ldv, p.v_bstride, p.s);
    const bool gkfast = p.s == S && vec4_ok(p.dk, p.lddk, p.dk_bstride, p.s);
    const bool gvfast = p.s == S && vec4_ok(p.dv, p.lddv, p.dv_bstride, p.s);
    f2 k[R][S / 2], v[R][S / 2], dk[R][S / 2], dv[R][S / 2];
    bool keep[R], on[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        on[r] = j0 + r < p.Tk;
        const int jj = on[r] ? j0 + r : 0;
        float ks[S], vs[S];
        load_row_any<S>(ks, p.k + (int64_t)b * p.k_bstride + (int64_t)jj * p.ldk + col0, p.s, on[r], kfast);
        load_row_any<S>(vs, p.v + (int64_t)b * p.v_bstride + (int64_t)jj * p.ldv + col0, p.s, on[r], vfast);
#pragma unroll
        for (int d = 0; d < S / 2; ++d) {
            k[r][d] = f2{ks[2 * d] * p.scale, ks[2 * d + 1] * p.scale};
            v[r][d] = f2{vs[2 * d], vs[2 * d + 1]};
            dk[r][d] = dv[r][d] = f2{0.f, 0.f};
        }
        keep[r] = on[r] && (p.mask ? p.mask[(int64_t)b * p.Tk + jj] != 0 : true);
    }

    constexpr int QC = 4;   // queries per unrolled chunk
    for (int q0 = 0; q0 < p.Tq; q0 += QTILE) {
        const int nt = min(QTILE, p.Tq - q0), ntp = (nt + QC - 1) / QC * QC;
        __syncthreads();
        stage_rows<S>(Qs, qb, p.ldq, col0, p.s, q0, nt, qvec, ntp);
        stage_rows<S>(Ds, db, p.ldd, col0, p.s, q0, nt, dvec, ntp);
        for (int t = threadIdx.x; t < ntp; t += blockDim.x) {
            const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + q0 + (t < nt ? t : 0);
            Lm[t] = t < nt ? p.lse[2 * stat] : INFINITY;      // +inf: a query beyond the sequence gets p = exp(-inf) = 0
            Ll[t] = t < nt ? p.lse[2 * stat + 1] : 0.f;
            Dl[t] = t < nt ? p.delta[stat] : 0.f;
        }
        __syncthreads();
        for (int t0 = 0; t0 < ntp; t0 += QC) {
#pragma unroll
            for (int tt = 0; tt < QC; ++tt) {
                const int t = t0 + tt;
                f2 qv[S / 2], dvv[S / 2];
                load_row2<S>(qv, Qs + t * S);
                load_row2<S>(dvv, Ds + t * S);
                const float lm = Lm[t], ll = Ll[t], dl = Dl[t];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float a = dot2<S>(qv, k[r]);
                    const float dp = dot2<S>(dvv, v[r]);
                    a = keep[r] ? a : kMaskFill;
                    const float pr = __expf((a - lm) - ll);
                    axpy2<S>(dv[r], pr, dvv);
                    const float ds = keep[r] ? pr * (dp - dl) : 0.f;
                    axpy2<S>(dk[r], ds, qv);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (on[r]) {
            float* ok = p.dk + (int64_t)b * p.dk_bstride + (int64_t)(j0 + r) * p.lddk + col0;
            float* ov = p.dv + (int64_t)b * p.dv_bstride + (int64_t)(j0 + r) * p.lddv + col0;
            float gk[S], gv[S];
#pragma unroll
            for (int d = 0; d < S; ++d) {
                gk[d] = (d & 1 ? dk[r][d / 2].y : dk[r][d / 2].x) * p.scale;
                gv[d] = d & 1 ? dv[r][d / 2].y : dv[r][d / 2].x;
            }
            store_row_any<S>(ok, gk, p.s, gkfast);
            store_row_any<S>(ov, gv, p.s, gvfast);
        }
    }
}

// matrix-core path (attention_mfma.hip)
// (MAttn and the mattn_* / pattn_* entry points: attention_args.h)
static int g_attn_path = 0;  // 0 = automatic (matrix cores for head widths >= 16: narrower heads measured no faster
                             // there, tools/bench_attention.py), 1 = always the vector-ALU kernels, 2 = matrix cores
                             // whenever applicable

// The vector-ALU kernels exist for head widths up to 32.  (Their 64- and 128-wide instantiations kept a row of the head in
// registers per lane: 256 VGPRs + 256 AGPRs and 164 - 1972 bytes of scratch per lane in the backward -- and were only reached by
// widths above 32 that are not a multiple of 4, or by misaligned operands; those are refused now: the caller pads the head to a
// multiple of 4 -- ops.attention_fwd / attention_bwd do -- and the matrix-core kernels take it.)
static int pad_head(int s) {
    for (int c : {4, 8, 16, 32})
        if (s <= c) return c;
    return -1;
}
static unsigned block_for(int t) { return (unsigned)std::min(256, (t + 63) / 64 * 64); }

// rows per lane: 4 for narrow heads once the sequence would fill more than two waves at one row per lane
// ... and only with at least 8192 (batch, head) pairs (measured at T = 200, 8-wide heads: R = 1 is 5-15 % faster from
// 1024 to 4096 pairs, equal at 8192): at the reference's own batch sizes R = 4 leaves most SIMDs without a wave.
static int rows_per_lane(int S, int t, int64_t pairs) {
    if (t <= 128 || pairs < 8192) return 1;
    return S <= 8 ? 4 : (S == 16 ? 2 : 1);
}
#define MSN_ATTN_DISPATCH(KERNEL, S, R, grid, block, lds, st, args)                                   \
    switch (S) {                                                                                 \
        case 4:                                                                                  \
            if (R == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, lds, st, args);            \
            else hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, lds, st, args);                   \
            break;                                                                               \
        case 8:                                                                                  \
            if (R == 4) hipLaunchKernelGGL((KERNEL<8, 4>), grid, block, lds, st, args);            \
            else hipLaunchKernelGGL((KERNEL<8, 1>), grid, block, lds, st, args);                   \
            break;                                                                               \
        case 16:                                                                                 \
            if (R == 2) hipLaunchKernelGGL((KERNEL<16, 2>), grid, block, lds, st, args);           \
            else hipLaunchKernelGGL((KERNEL<16, 1>), grid, block, lds, st, args);                  \
            break;                                                                               \
        default: hipLaunchKernelGGL((KERNEL<32, 1>), grid, block, lds, st, args); break;           \
    }

static int check_attn(const char* who, int B, int H, int Tq, int Tk, int s) {
    MSN_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0 && s > 0, "%s: empty input", who);
    MSN_REQUIRE(s <= 128, "%s: head width %d > 128 unsupported", who, s);
    MSN_REQUIRE(B <= 65535 && H <= 65535, "%s: batch / heads exceed the grid limits", who);
    return MSN_OK;
}

}  // namespace msn

using namespace msn;

extern "C" int msn_attention_fwd(const float* q, int64_t ldq, int64_t q_bstride, const float* k, int64_t ldk,
                                 int64_t k_bstride, const float* v, int64_t ldv, int64_t v_bstride,
                                 const uint8_t* key_mask, int B, int H, int Tq, int Tk, int head_dim, float scale,
                                 float* out, int64_t ldo, int64_t o_bstride, float* lse, msn_stream_t stream) {
    if (int rc = check_attn("msn_attention_fwd", B, H, Tq, Tk, head_dim)) return rc;
    MSN_REQUIRE(q && k && v && out && lse, "msn_attention_fwd: null pointer");
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = out; a.mask = key_mask; a.lse = lse;
    a.q_bstride = q_bstride; a.k_bstride = k_bstride; a.v_bstride = v_bstride; a.o_bstride = o_bstride;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.s = head_dim; a.scale = scale;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_attn_path != 1) {
        MAttn m = {};
        m.q = q; m.k = k; m.v = v; m.out = out; m.mask = key_mask; m.lse = lse;
        m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo;
        m.q_bs = q_bstride; m.k_bs = k_bstride; m.v_bs = v_bstride; m.o_bs = o_bstride;
        m.B = B; m.H = H; m.Tq = Tq; m.Tk = Tk; m.hd = head_dim; m.scale = scale;
        if (mattn_applicable(m) && (head_dim >= 16 || g_attn_path == 2)) return mattn_forward(m, st);
    }
    const int S = pad_head(head_dim);
    if (S < 0) {                                           // wider than 32: matrix cores only (also under mode 1), shared query included
        MAttn m = {};
        m.q = q; m.k = k; m.v = v; m.out = out; m.mask = key_mask; m.lse = lse;
        m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo;
        m.q_bs = q_bstride; m.k_bs = k_bstride; m.v_bs = v_bstride; m.o_bs = o_bstride;
        m.B = B; m.H = H; m.Tq = Tq; m.Tk = Tk; m.hd = head_dim; m.scale = scale;
        MSN_REQUIRE(mattn_applicable(m, true), "msn_attention_fwd: a head width above 32 (%d) must be a multiple of 4 with 16-byte aligned "
                    "rows (pad the heads with zero columns)", head_dim);
        return mattn_forward(m, st);
    }
    const int R = rows_per_lane(S, Tq, (int64_t)B * H);
    const unsigned bs = block_for((int)cdiv(Tq, R));
    const dim3 grid((unsigned)cdiv(Tq, (int64_t)bs * R), H, B), block(bs);
    a.tile_rows = std::min(4096 / S, (Tk + 7) / 8 * 8);
    const size_t lds = sizeof(float) * (2 * (size_t)a.tile_rows * S + (size_t)a.tile_rows);
    MSN_ATTN_DISPATCH(attn_fwd_kernel, S, R, grid, block, lds, st, a)
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_set_attention_path(int mode) {
    MSN_REQUIRE(mode >= 0 && mode <= 2, "msn_set_attention_path: mode must be 0, 1 or 2");
    g_attn_path = mode;
    return MSN_OK;
}

extern "C" int msn_attention_bwd(const float* q, int64_t ldq, int64_t q_bstride, const float* k, int64_t ldk,
                                 int64_t k_bstride, const float* v, int64_t ldv, int64_t v_bstride,
                                 const uint8_t* key_mask, int B, int H, int Tq, int Tk, int head_dim, float scale,
                                 const float* out, int64_t ldo, int64_t o_bstride, const float* lse,
                                 const float* dout, int64_t ldd, int64_t d_bstride, float* delta, float* dq,
                                 int64_t lddq, int64_t dq_bstride, float* dk, int64_t lddk, int64_t dk_bstride,
                                 float* dv, int64_t lddv, int64_t dv_bstride, msn_stream_t stream) {
    if (int rc = check_attn("msn_attention_bwd", B, H, Tq, Tk, head_dim)) return rc;
    MSN_REQUIRE(q && k && v && out && lse && dout && delta && dq && dk && dv, "msn_attention_bwd: null pointer");
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = const_cast<float*>(out); a.mask = key_mask; a.lse = const_cast<float*>(lse);
    a.q_bstride = q_bstride; a.k_bstride = k_bstride; a.v_bstride = v_bstride; a.o_bstride = o_bstride;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.s = head_dim; a.scale = scale;
    a.dout = dout; a.ldd = ldd; a.d_bstride = d_bstride; a.delta = delta;
    a.dq = dq; a.dk = dk; a.dv = dv; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.dq_bstride = dq_bstride; a.dk_bstride = dk_bstride; a.dv_bstride = dv_bstride;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (g_attn_path != 1) {
        MAttn m = {};
        m.q = q; m.k = k; m.v = v; m.o = out; m.dout = dout; m.dq = dq; m.dk = dk; m.dv = dv;
        m.mask = key_mask; m.lse = const_cast<float*>(lse); m.delta = delta;
        m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo; m.ldd = ldd; m.lddq = lddq; m.lddk = lddk; m.lddv = lddv;
        m.q_bs = q_bstride; m.k_bs = k_bstride; m.v_bs = v_bstride; m.o_bs = o_bstride; m.d_bs = d_bstride;
        m.dq_bs = dq_bstride; m.dk_bs = dk_bstride; m.dv_bs = dv_bstride;
        m.B = B; m.H = H; m.Tq = Tq; m.Tk = Tk; m.hd = head_dim; m.scale = scale;
        if (mattn_applicable(m) && (head_dim >= 16 || g_attn_path == 2)) return mattn_backward(m, st);
        if (g_attn_path == 0 && pattn_backward_preferred(m)) return pattn_backward(m, st);
    }
    const int S = pad_head(head_dim);
    if (S < 0) {                                           // wider than 32: matrix cores only (also under mode 1), shared query included
        MAttn m = {};
        m.q = q; m.k = k; m.v = v; m.o = out; m.dout = dout; m.dq = dq; m.dk = dk; m.dv = dv;
        m.mask = key_mask; m.lse = const_cast<float*>(lse); m.delta = delta;
        m.ldq = ldq; m.ldk = ldk; m.ldv = ldv; m.ldo = ldo; m.ldd = ldd; m.lddq = lddq; m.lddk = lddk; m.lddv = lddv;
        m.q_bs = q_bstride; m.k_bs = k_bstride; m.v_bs = v_bstride; m.o_bs = o_bstride; m.d_bs = d_bstride;
        m.dq_bs = dq_bstride; m.dk_bs = dk_bstride; m.dv_bs = dv_bstride;
        m.B = B; m.H = H; m.Tq = Tq; m.Tk = Tk; m.hd = head_dim; m.scale = scale;
        MSN_REQUIRE(mattn_applicable(m, true), "msn_attention_bwd: a head width above 32 (%d) must be a multiple of 4 with 16-byte aligned "
                    "rows (pad the heads with zero columns)", head_dim);
        return mattn_backward(m, st);
    }
    {
        const int R = rows_per_lane(S, Tq, (int64_t)B * H);
        const unsigned bs = block_for((int)cdiv(Tq, R));
        const dim3 grid((unsigned)cdiv(Tq, (int64_t)bs * R), H, B), block(bs);
        a.tile_rows = std::min(4096 / S, (Tk + 7) / 8 * 8);
        const size_t lds = sizeof(float) * (2 * (size_t)a.tile_rows * S + (size_t)a.tile_rows);
        MSN_ATTN_DISPATCH(attn_bwd_dq_kernel, S, R, grid, block, lds, st, a)
        MSN_LAUNCH_CHECK();
    }
    {
        const int R = rows_per_lane(S, Tk, (int64_t)B * H);
        const unsigned bs = block_for((int)cdiv(Tk, R));
        const dim3 grid((unsigned)cdiv(Tk, (int64_t)bs * R), H, B), block(bs);
        a.tile_rows = std::min(4096 / S, (Tq + 7) / 8 * 8);
        const size_t lds = sizeof(float) * (2 * (size_t)a.tile_rows * S + 3 * (size_t)a.tile_rows);
        MSN_ATTN_DISPATCH(attn_bwd_dkv_kernel, S, R, grid, block, lds, st, a)
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}
