// Fused symmetric InfoNCE (CLIP loss) forward / backward -- never materialises the N x N logits.
//
// Reference: src/loss.py:14-38 (clip_loss).  With S = s * E2 . E1^T + b  (s = exp(logit_scale)):
//   loss = 1/(2n) * sum_{i<n} [ LSE_j S_ij + LSE_j S_ji - 2 S_ii ]
//   G_ij = dloss/dS_ij = ( [i<n] softmax_row(S)_ij + [j<n] softmax_col(S)_ij - 2 delta_ij ) / (2n)
//   dE2_i = s * sum_j G_ij E1_j ;  dE1_j = s * sum_i G_ij E2_i ;  dlogit_scale = sum_ij G_ij (S_ij - b)
//
// Sharding (one process per GPU, rows of the global batch split by rank): a rank owns `b` rows of
// each modality and holds the all-gathered (N x D) matrices.  Both directions have the same shape:
// "queries" = the local rows of one modality, "keys" = all rows of the other; direction 0 takes
// queries from E2 (rows of S), direction 1 from E1 (columns of S).  The forward needs only the
// per-query log-sum-exp; the backward recomputes S tiles from E and the (all-gathered) LSE vectors,
// so dE of the local rows is complete on the owning rank and no reduce-scatter is needed.
//
// Kernel shape: a workgroup (4 waves) owns ONE 32-query tile (32 * blockIdx.x), keeps the Q fragments in registers, and
// splits its share of the keys (blockIdx.y = key split) over its waves: wave w sweeps the 32-key tiles w, w + 4, ...
// through its own LDS buffer (no workgroup barrier inside the sweep).  S^T tile = K_tile . Q_tile^T on the f32 matrix
// cores (v_mfma_f32_32x32x2_f32), oriented so that the lane's column is the QUERY: the softmax statistics of a query
// are then lane-local (16 registers = 16 keys per half-wave), combined once at the end with one cross-half shuffle and
// across the four waves through LDS.  The positive S_ii is taken from the score tile that holds it (the same
// accumulation the log-sum-exp sees), not recomputed.  In the backward the G^T tile sitting in the accumulator
// registers is directly the A operand of the second product dQ += G . K (key index on the MFMA k axis), whose B operand
// is read from the same LDS key tile; the four waves' dQ tiles are summed through LDS in a fixed order, so a key split
// publishes ONE partial per query tile (a quarter of the slab traffic of a wave-per-query-tile layout) and with a
// single key split the gradient is final.  Partial results of the key splits go to caller scratch and are merged in
// a fixed order (deterministic; no float atomics, no zero-initialised scratch).
//
// Any embedding width 1 <= D <= 256 (the reference takes any enc_dim, src/models_multimodal.py:101): the tiles are
// DP = 8 / 16 / 32 / 64 / 128 / 256 columns wide and columns D .. DP-1 are zero in LDS and in the Q fragments -- zero
// columns change no inner product.
#include <algorithm>
#include <math.h>

#include "msn_common.h"

namespace msn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int QT = 32;        // queries per workgroup
constexpr int KT = 32;        // keys per tile
constexpr int NW = 4;         // waves per workgroup, each sweeping its own key tiles
constexpr int MODE_SOFTMAX = 0, MODE_SIGMOID = 1;

struct Side {
    const float* Q;      // local rows (queries), [nq][ldq]
    const float* K;      // all rows of the other modality (keys), [nk][ldk]
    const float* lse_q;  // bwd: LSE of the queries' own direction, indexed by GLOBAL row id
    const float* lse_k;  // bwd: LSE of the keys' direction, indexed by global row id
    float* dQ;           // bwd: [nq][ldd]
    int64_t ldq, ldk, ldd;
    int nq, nk;
};

struct NceArgs {
    Side side[2];
    const float* log_scale;  // device scalar (log of the logit scale)
    const float* bias;       // device scalar
    const float* grad_out;   // bwd: device scalar
    int q_offset;            // global row id of local row 0
    int n_diag;              // n = min(N1, N2)
    int D;                   // real embedding width (<= the kernel's DP)
    int ksplit, keys_per_split;
    float* part_m;           // fwd scratch [2][ksplit][maxq]: running max, sum, positive score of a key split
    float* part_l;
    float* part_d;
    float* slab;             // bwd scratch [2][ksplit][maxq][D] (ksplit > 1 only)
    double* scal;            // bwd scratch [qtiles * ksplit][2] : partial dscale, dbias; fwd sigmoid partial loss
    int maxq;
    int mode;
};

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// A wave re-uses its LDS key buffer: LDS operations of one wave execute in order, this only pins the compiler.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float4 load4_guarded(const float* p, int c0, int D, bool vec_ok) {
    if (vec_ok && c0 + 3 < D) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < D) v.x = p[0];
    if (c0 + 1 < D) v.y = p[1];
    if (c0 + 2 < D) v.z = p[2];
    if (c0 + 3 < D) v.w = p[3];
    return v;
}

// One wave stages one 32-key tile (rows k0..k0+31 of K, zero beyond k_end and beyond column D) into ITS LDS buffer,
// row stride DP + 4.
template <int DP>
__device__ __forceinline__ void stage_keys(float* Ks, const float* __restrict__ K, int64_t ldk, int k0, int k_end, int D,
                                           bool vec_ok, int lane) {
    constexpr int KS = DP + 4;
    constexpr int PER_ROW = DP / 4;
#pragma unroll
    for (int j = 0; j < KT * PER_ROW / 64; ++j) {
        const int idx = lane + 64 * j;
        const int r = idx / PER_ROW, q = idx % PER_ROW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + r < k_end) v = load4_guarded(K + (int64_t)(k0 + r) * ldk + 4 * q, 4 * q, D, vec_ok);
        *reinterpret_cast<float4*>(Ks + r * KS + 4 * q) = v;
    }
}

template <int DP>
__device__ __forceinline__ void load_q_frags(float4 (&qf)[DP / 8], const float* __restrict__ Q, int64_t ldq,
                                             int qrow, int h, int D, bool vec_ok) {
    const float* p = Q + (int64_t)qrow * ldq + 4 * h;
#pragma unroll
    for (int ko = 0; ko < DP / 8; ++ko) qf[ko] = load4_guarded(p + 8 * ko, 8 * ko + 4 * h, D, vec_ok);
}

// acc[key][query] = sum_d K[key][d] * Q[query][d] for the staged tile; lane col = query (lane & 31).
template <int DP>
__device__ __forceinline__ f32x16 score_tile(const float* Ks, const float4 (&qf)[DP / 8], int l32, int h) {
    constexpr int KS = DP + 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ko = 0; ko < DP / 8; ++ko) {
        const float4 kf = *reinterpret_cast<const float4*>(Ks + l32 * KS + 8 * ko + 4 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[ko].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[ko].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[ko].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[ko].w, acc, 0, 0, 0);
    }
    return acc;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------------ forward
template <int DP>
__global__ __launch_bounds__(256) void nce_fwd_kernel(const NceArgs p) {
    constexpr int KS = DP + 4;
    __shared__ __attribute__((aligned(16))) float lds[cmax(NW * KT * KS, 3 * NW * QT)];
    __shared__ double dred[NW];
    const Side& sd = p.side[blockIdx.z];
    const int qb0 = blockIdx.x * QT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, h = lane >> 5;
    const bool sig = p.mode == MODE_SIGMOID;
    if (qb0 >= sd.nq) {          // uniform per workgroup: the other direction has more query tiles
        if (sig && threadIdx.x == 0) p.scal[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = 0.0;
        return;
    }
    const int q_local = qb0 + l32;
    const bool q_ok = q_local < sd.nq;
    const int q_glob = p.q_offset + q_local;
    const float scale = __expf(*p.log_scale), bias = *p.bias;
    const bool qvec = (sd.ldq % 4 == 0) && ((reinterpret_cast<uintptr_t>(sd.Q) & 15) == 0);
    const bool kvec = (sd.ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(sd.K) & 15) == 0);

    float4 qf[DP / 8];
    load_q_frags<DP>(qf, sd.Q, sd.ldq, q_ok ? q_local : sd.nq - 1, h, p.D, qvec);
    float* Ks = lds + wave * (KT * KS);

    const int k_begin = blockIdx.y * p.keys_per_split;
    const int k_end = min(sd.nk, k_begin + p.keys_per_split);
    if (sig) {
        // ref src/loss.py:68-83: Z = -(E2 . E1^T) s + b (fp32), then -log sigmoid(-z Z) = softplus(z Z) in fp64,
        // z = +1 on the diagonal and -1 elsewhere.  Only direction 0 accumulates (each (i, j) once).
        double part = 0.0;
        for (int k0 = k_begin + KT * wave; k0 < k_end; k0 += KT * NW) {
            wave_fence();
            stage_keys<DP>(Ks, sd.K, sd.ldk, k0, k_end, p.D, kvec, lane);
            wave_fence();
            const f32x16 acc = score_tile<DP>(Ks, qf, l32, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kj = k0 + row_of(r, h);
                if (q_ok && kj < k_end) {
                    const float Z = -acc[r] * scale + bias;
                    const double u = (kj == q_glob) ? (double)Z : -(double)Z;     // z * Z
                    part += u > 0.0 ? u + log1p(exp(-u)) : log1p(exp(u));         // softplus(u)
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) dred[wave] = part;
        __syncthreads();
        if (threadIdx.x == 0) p.scal[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
        return;
    }
    float m = -INFINITY, l = 0.f, dg = -INFINITY;
    for (int k0 = k_begin + KT * wave; k0 < k_end; k0 += KT * NW) {
        wave_fence();
        stage_keys<DP>(Ks, sd.K, sd.ldk, k0, k_end, p.D, kvec, lane);
        wave_fence();
        const f32x16 acc = score_tile<DP>(Ks, qf, l32, h);
        float s[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = k0 + row_of(r, h);
            s[r] = (kj < k_end) ? acc[r] * scale + bias : -INFINITY;
            tmax = fmaxf(tmax, s[r]);
            if (kj == q_glob) dg = s[r];          // the positive: key = the query's own global row
        }
        const float mn = fmaxf(m, tmax);
        if (mn > -INFINITY) {
            float add = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += __expf(s[r] - mn);
            l = l * __expf(m - mn) + add;
            m = mn;
        }
    }
    // combine the two half-waves (a tile's keys are split between them), then the four waves through LDS
    const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64);
    const float mm = fmaxf(m, m2);
    float ll = 0.f;
    if (mm > -INFINITY) ll = l * __expf(m - mm) + l2 * __expf(m2 - mm);
    dg = fmaxf(dg, __shfl_xor(dg, 32, 64));
    __syncthreads();                              // every wave is done with its key buffer
    float* wm = lds;
    float* wl = lds + NW * QT;
    float* wd = lds + 2 * NW * QT;
    if (h == 0) {
        wm[wave * QT + l32] = mm;
        wl[wave * QT + l32] = ll;
        wd[wave * QT + l32] = dg;
    }
    __syncthreads();
    if (threadIdx.x < QT && qb0 + threadIdx.x < sd.nq) {
        const int t = threadIdx.x;
        float M = -INFINITY, Dg = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            M = fmaxf(M, wm[w * QT + t]);
            Dg = fmaxf(Dg, wd[w * QT + t]);
        }
        float L = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float mw = wm[w * QT + t];
            if (mw > -INFINITY) L += wl[w * QT + t] * __expf(mw - M);
        }
        const int64_t o = ((int64_t)blockIdx.z * p.ksplit + blockIdx.y) * p.maxq + qb0 + t;
        p.part_m[o] = M;
        p.part_l[o] = L;
        p.part_d[o] = Dg;
    }
}

// Sum of n doubles at stride `stride` by ONE wave: lane l adds elements l, l + 64, ... in order, then a fixed xor tree
// (deterministic; 64 loads in flight instead of a chain of n dependent ones).  Every lane gets the result.
__device__ __forceinline__ double wave_sum_strided(const double* __restrict__ v, int n, int stride, int lane) {
    double acc = 0.0;
    for (int k = lane; k < n; k += 64) acc += v[(int64_t)k * stride];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    return acc;
}

// Merge key splits -> LSE per local row and direction; this rank's share of the loss.  One workgroup: a thread owns
// local row i in BOTH directions; the splits' (m, l, positive) triples are fetched eight at a time (independent loads,
// one wait) and folded into a running (M, L) pair, then one block sum.
__global__ __launch_bounds__(1024) void nce_fwd_finish_kernel(const NceArgs p, float* __restrict__ lse_row,
                                                              float* __restrict__ lse_col, float* __restrict__ loss,
                                                              int n_part) {
    __shared__ float red[16];
    if (p.mode == MODE_SIGMOID) {   // loss = sum of the block partials / bs^2 (mean over all bs x bs entries)
        if (threadIdx.x < 64) {
            const double tot = wave_sum_strided(p.scal, n_part, 1, threadIdx.x);
            if (threadIdx.x == 0) *loss = (float)(tot / ((double)p.n_diag * (double)p.n_diag));
        }
        return;
    }
    float* lse_out[2] = {lse_row, lse_col};
    // rows of S come from side[0].Q (E2), columns from side[1].Q (E1); local row i <-> global q_offset + i
    const int nb = min(p.side[0].nq, p.side[1].nq);
    const int nmax = max(p.side[0].nq, p.side[1].nq);
    const int ks = p.ksplit;
    float acc = 0.f;
    for (int i = threadIdx.x; i < nmax; i += blockDim.x) {
        float lse[2] = {0.f, 0.f};
        float diag = -INFINITY;
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            if (i >= p.side[dir].nq) continue;
            const float* pm = p.part_m + (int64_t)dir * ks * p.maxq + i;
            const float* pl = p.part_l + (int64_t)dir * ks * p.maxq + i;
            const float* pd = p.part_d + (int64_t)dir * ks * p.maxq + i;
            float M = -INFINITY, L = 0.f;
            for (int s0 = 0; s0 < ks; s0 += 8) {
                float mv[8], lv[8], dv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {                      // clamped index: unconditional, independent loads
                    const int64_t o = (int64_t)min(s0 + j, ks - 1) * p.maxq;
                    mv[j] = pm[o];
                    lv[j] = pl[o];
                    dv[j] = dir == 0 ? pd[o] : -INFINITY;
                }
                float cm = M;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (s0 + j >= ks) mv[j] = -INFINITY;
                    cm = fmaxf(cm, mv[j]);
                    diag = fmaxf(diag, dv[j]);                     // a clamped duplicate repeats the same value: harmless
                }
                if (cm > -INFINITY) {
                    float Ln = (M > -INFINITY) ? L * __expf(M - cm) : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (mv[j] > -INFINITY) Ln += lv[j] * __expf(mv[j] - cm);
                    L = Ln;
                    M = cm;
                }
            }
            lse[dir] = M + __logf(L);
            lse_out[dir][i] = lse[dir];
        }
        if (i < nb && p.q_offset + i < p.n_diag) acc += lse[0] + lse[1] - 2.f * diag;
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) *loss = tot / (2.f * (float)p.n_diag);
}

// ----------------------------------------------------------------------------------------- backward
template <int DP>
__global__ __launch_bounds__(256) void nce_bwd_kernel(const NceArgs p) {
    constexpr int KS = DP + 4;
    constexpr int DT = (DP + 31) / 32;
    constexpr int DW = DT * 32;                 // columns of the dQ reduction image
    __shared__ __attribute__((aligned(16))) float lds[cmax(NW * KT * KS, NW * QT * DW)];
    __shared__ float lseK[NW][KT];
    __shared__ float red[8];
    const int dir = blockIdx.z;
    const Side& sd = p.side[dir];
    const int qb0 = blockIdx.x * QT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, h = lane >> 5;
    double* scal_out = p.scal + 2 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
    if (qb0 >= sd.nq) {          // uniform per workgroup; direction 0 still owes its (zero) scalar partials
        if (dir == 0 && threadIdx.x == 0) scal_out[0] = scal_out[1] = 0.0;
        return;
    }
    const int q_local = qb0 + l32;
    const bool q_ok = q_local < sd.nq;
    const int q_glob = p.q_offset + q_local;
    const float scale = __expf(*p.log_scale), bias = *p.bias;
    const bool qvec = (sd.ldq % 4 == 0) && ((reinterpret_cast<uintptr_t>(sd.Q) & 15) == 0);
    const bool kvec = (sd.ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(sd.K) & 15) == 0);

    float4 qf[DP / 8];
    load_q_frags<DP>(qf, sd.Q, sd.ldq, q_ok ? q_local : sd.nq - 1, h, p.D, qvec);
    const bool sig = p.mode == MODE_SIGMOID;
    const bool q_in = q_ok && q_glob < p.n_diag;
    const float lq = (q_in && !sig) ? sd.lse_q[q_glob] : 0.f;
    float* Ks = lds + wave * (KT * KS);

    f32x16 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
    float ds = 0.f, db = 0.f;  // partial sum G (S - b), sum G

    const int k_begin = blockIdx.y * p.keys_per_split;
    const int k_end = min(sd.nk, k_begin + p.keys_per_split);
    for (int k0 = k_begin + KT * wave; k0 < k_end; k0 += KT * NW) {
        wave_fence();
        stage_keys<DP>(Ks, sd.K, sd.ldk, k0, k_end, p.D, kvec, lane);
        if (lane < KT) {
            const int kj = k0 + lane;
            lseK[wave][lane] = (!sig && kj < k_end && kj < p.n_diag) ? sd.lse_k[kj] : INFINITY;
        }
        wave_fence();
        f32x16 g = score_tile<DP>(Ks, qf, l32, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = row_of(r, h);
            const int kj = k0 + kr;
            float S, G = 0.f;
            if (sig) {  // dL/dZ = z sigmoid(z Z) (/ bs^2 in the finish), Z = -x s + b
                S = -g[r] * scale + bias;
                if (q_ok && kj < k_end) {
                    const float zz = (kj == q_glob) ? S : -S;
                    const float sg = 1.f / (1.f + __expf(-zz));
                    G = (kj == q_glob) ? sg : -sg;
                }
            } else {
                S = g[r] * scale + bias;
                if (q_ok && kj < k_end) {
                    if (q_in) G += __expf(S - lq);
                    G += __expf(S - lseK[wave][kr]);  // lseK = +inf for keys outside the diagonal range -> 0
                    if (q_in && kj == q_glob) G -= 2.f;
                }
            }
            g[r] = G;
            ds = fmaf(G, S - bias, ds);
            db += G;
        }
        // dQ[query][d] += sum_key G[query][key] * K[key][d]: A = G (lane = query, k = half-wave),
        // B = K[key_r(h)][32 t + lane&31] from LDS (32 lanes sweep one row: conflict-free)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* krow = Ks + row_of(r, h) * KS;
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                const int d = 32 * t + l32;
                const float bv = (DP % 32 == 0 || d < DP) ? krow[d] : 0.f;
                dq[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[r], bv, dq[t], 0, 0, 0);
            }
        }
    }
    // the four waves' partial dQ tiles -> LDS [wave][query][DW] (C layout: col = d on the lane, rows = queries), summed
    // in wave order; with a single key split the sum is the gradient, otherwise this split's slab
    __syncthreads();
    float* R = lds + wave * (QT * DW);
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) R[row_of(r, h) * DW + 32 * t + l32] = dq[t][r];
    __syncthreads();
    const bool final_pass = p.ksplit == 1;
    // softmax: G carries 1/(2n) and dS/dx = +s ; sigmoid: G carries 1/bs^2 and dZ/dx = -s
    const float gsc = sig ? -*p.grad_out / ((float)p.n_diag * (float)p.n_diag) : *p.grad_out / (2.f * (float)p.n_diag);
    const float f = gsc * scale;
    float* slab = p.slab + (((int64_t)dir * p.ksplit + blockIdx.y) * p.maxq) * p.D;
    for (int idx = threadIdx.x; idx < QT * DW; idx += 256) {
        const int q = idx / DW, d = idx % DW;
        if (d >= p.D || qb0 + q >= sd.nq) continue;
        float s = lds[idx];
#pragma unroll
        for (int w = 1; w < NW; ++w) s += lds[w * (QT * DW) + idx];
        if (final_pass) sd.dQ[(int64_t)(qb0 + q) * sd.ldd + d] = f * s;
        else slab[(int64_t)(qb0 + q) * p.D + d] = s;
    }
    // scalar partials (direction 0 covers every (i, j) exactly once)
    if (dir == 0) {
        ds = wave_sum(ds);
        db = wave_sum(db);
        if (lane == 0) {
            red[wave] = ds;
            red[NW + wave] = db;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            scal_out[0] = (double)((red[0] + red[1]) + (red[2] + red[3]));
            scal_out[1] = (double)((red[NW] + red[NW + 1]) + (red[NW + 2] + red[NW + 3]));
        }
    }
}

// dQ = grad_out * s / (2n) * sum_split slab (ksplit > 1) ; dscale/dbias = grad_out / (2n) * sum partials
__global__ void nce_bwd_finish_kernel(const NceArgs p, int n_scal, float* __restrict__ dscal_out) {
    const float g = p.mode == MODE_SIGMOID ? *p.grad_out / ((float)p.n_diag * (float)p.n_diag)
                                           : *p.grad_out / (2.f * (float)p.n_diag);
    const float f = (p.mode == MODE_SIGMOID ? -g : g) * __expf(*p.log_scale);
    const int D = p.D, ks = p.ksplit;
    if (ks > 1) {
        const int64_t slab_stride = (int64_t)p.maxq * D;
        for (int dir = 0; dir < 2; ++dir) {
            const Side& sd = p.side[dir];
            const int64_t total = (int64_t)sd.nq * D;
            const float* base = p.slab + (int64_t)dir * ks * slab_stride;
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
                 i += (int64_t)gridDim.x * blockDim.x) {
                float s = 0.f;
                for (int k0 = 0; k0 < ks; k0 += 8) {              // eight independent loads per wait, summed in split order
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = base[(int64_t)min(k0 + j, ks - 1) * slab_stride + i];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k0 + j < ks) s += v[j];
                }
                sd.dQ[(i / D) * sd.ldd + (i % D)] = f * s;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 128 && dscal_out) {          // wave 0: dscale, wave 1: dbias
        const int which = threadIdx.x >> 6;
        const double s = wave_sum_strided(p.scal + which, n_scal, 2, threadIdx.x & 63);
        if ((threadIdx.x & 63) == 0) dscal_out[which] = g * (float)s;
    }
}

// -------------------------------------------------------------------------- retrieval rank (validation AUC)
// rank[i] = #{ j != i : <E2_i, E1_j> > <E2_i, E1_i> }  -- the position of the true partner in the
// similarity ranking, ref src/utils.py:380-411 (get_ROC_data sorts every row on the host in a Python
// loop).  Same tile machinery: the diagonal score is taken from the SAME MFMA accumulation as the scores
// it is compared with (a first pass of every wave over the tile of the workgroup's own partners), then the key split
// is swept.
template <int DP>
__global__ __launch_bounds__(256) void nce_rank_kernel(const NceArgs p, int* __restrict__ part_cnt) {
    constexpr int KS = DP + 4;
    __shared__ __attribute__((aligned(16))) float lds[NW * KT * KS];
    __shared__ int wc[NW][QT];
    const Side& sd = p.side[0];
    const int qb0 = blockIdx.x * QT;
    if (qb0 >= sd.nq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, h = lane >> 5;
    const int q_local = qb0 + l32;
    const bool q_ok = q_local < sd.nq;
    const bool qvec = (sd.ldq % 4 == 0) && ((reinterpret_cast<uintptr_t>(sd.Q) & 15) == 0);
    const bool kvec = (sd.ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(sd.K) & 15) == 0);
    float4 qf[DP / 8];
    load_q_frags<DP>(qf, sd.Q, sd.ldq, q_ok ? q_local : sd.nq - 1, h, p.D, qvec);
    float* Ks = lds + wave * (KT * KS);

    float diag = -INFINITY;
    {
        stage_keys<DP>(Ks, sd.K, sd.ldk, qb0, sd.nk, p.D, kvec, lane);
        wave_fence();
        const f32x16 acc = score_tile<DP>(Ks, qf, l32, h);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (row_of(r, h) == l32) diag = acc[r];
    }
    diag = fmaxf(diag, __shfl_xor(diag, 32, 64));

    const int k_begin = blockIdx.y * p.keys_per_split;
    const int k_end = min(sd.nk, k_begin + p.keys_per_split);
    int cnt = 0;
    for (int k0 = k_begin + KT * wave; k0 < k_end; k0 += KT * NW) {
        wave_fence();
        stage_keys<DP>(Ks, sd.K, sd.ldk, k0, k_end, p.D, kvec, lane);
        wave_fence();
        const f32x16 acc = score_tile<DP>(Ks, qf, l32, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = k0 + row_of(r, h);
            cnt += (kj < k_end && kj != q_local && acc[r] > diag) ? 1 : 0;
        }
    }
    cnt += __shfl_xor(cnt, 32, 64);
    if (h == 0) wc[wave][l32] = cnt;
    __syncthreads();
    if (threadIdx.x < QT && qb0 + threadIdx.x < sd.nq)
        part_cnt[(int64_t)blockIdx.y * p.maxq + qb0 + threadIdx.x] =
            (wc[0][threadIdx.x] + wc[1][threadIdx.x]) + (wc[2][threadIdx.x] + wc[3][threadIdx.x]);
}
__global__ void nce_rank_finish_kernel(const int* __restrict__ part_cnt, int ksplit, int maxq, int n, int* __restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = 0;
    for (int k = 0; k < ksplit; ++k) s += part_cnt[(int64_t)k * maxq + i];
    rank[i] = s;
}

struct Plan {
    int ksplit, keys_per_split, maxq, qtiles;
    size_t off_m, off_l, off_d, off_slab, off_scal, total;
};

// Enough workgroups to fill the chip's 512 slots (two directions x query tiles x key splits); a key split is a whole
// number of 4-tile rounds so that the four waves of a workgroup carry equal shares.
static Plan make_plan(int b1, int b2, int n1, int n2, int D) {
    Plan pl;
    pl.maxq = std::max(b1, b2);
    const int maxk = std::max(n1, n2);
    pl.qtiles = (int)cdiv(pl.maxq, QT);
    int ks = std::max(1, 512 / (2 * pl.qtiles));
    ks = std::min(ks, (int)cdiv(maxk, KT * NW));
    ks = std::max(std::min(ks, 64), 1);
    pl.keys_per_split = (int)(cdiv(cdiv(maxk, ks), KT * NW) * KT * NW);
    pl.ksplit = (int)cdiv(maxk, pl.keys_per_split);
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    pl.off_m = take(sizeof(float) * 2 * pl.ksplit * (size_t)pl.maxq);
    pl.off_l = take(sizeof(float) * 2 * pl.ksplit * (size_t)pl.maxq);
    pl.off_d = take(sizeof(float) * 2 * pl.ksplit * (size_t)pl.maxq);
    pl.off_slab = take(sizeof(float) * 2 * pl.ksplit * (size_t)pl.maxq * D);
    pl.off_scal = take(sizeof(double) * 2 * (size_t)pl.ksplit * pl.qtiles);
    pl.total = o;
    return pl;
}

static int check_common(const char* who, int b1, int b2, int n1, int n2, int D, int q_offset) {
    MSN_REQUIRE(D >= 1 && D <= 256, "%s: embedding width D=%d unsupported (1 <= D <= 256)", who, D);
    MSN_REQUIRE(b1 > 0 && b2 > 0 && n1 >= b1 && n2 >= b2 && q_offset >= 0, "%s: bad row counts b1=%d b2=%d N1=%d N2=%d",
                who, b1, b2, n1, n2);
    return MSN_OK;
}

// tile width DP = D rounded up to 8 / 16 / 32 / 64 / 128 / 256
#define MSN_NCE_DISPATCH(KERNEL, ...)                                                        \
    if (D <= 8) hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__);                                \
    else if (D <= 16) hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__);                         \
    else if (D <= 32) hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__);                         \
    else if (D <= 64) hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__);                         \
    else if (D <= 128) hipLaunchKernelGGL((KERNEL<128>), __VA_ARGS__);                       \
    else hipLaunchKernelGGL((KERNEL<256>), __VA_ARGS__);

}  // namespace msn

using namespace msn;

extern "C" size_t msn_infonce_workspace_bytes(int b1, int b2, int n1, int n2, int D) {
    if (b1 <= 0 || b2 <= 0 || n1 <= 0 || n2 <= 0 || D <= 0) return 0;
    return make_plan(b1, b2, n1, n2, D).total;
}

static int infonce_fwd_impl(int mode, const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                            const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2, int D,
                            int q_offset, const float* log_scale, const float* bias, float* lse_row, float* lse_col,
                            float* loss, void* ws, size_t ws_bytes, msn_stream_t stream);

extern "C" int msn_infonce_fwd(const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                               const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2,
                               int D, int q_offset, const float* log_scale, const float* bias, float* lse_row,
                               float* lse_col, float* loss, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(lse_row && lse_col, "msn_infonce_fwd: null pointer");
    return infonce_fwd_impl(MODE_SOFTMAX, E1_loc, ld1, b1, E2_loc, ld2, b2, E1_all, ld1a, n1, E2_all, ld2a, n2, D, q_offset,
                            log_scale, bias, lse_row, lse_col, loss, ws, ws_bytes, stream);
}

extern "C" int msn_sigmoid_loss_fwd(const float* E1_loc, int64_t ld1, const float* E2_loc, int64_t ld2, int b,
                                    const float* E1_all, int64_t ld1a, const float* E2_all, int64_t ld2a, int n, int D,
                                    int q_offset, const float* log_scale, const float* bias, float* loss, void* ws,
                                    size_t ws_bytes, msn_stream_t stream) {
    return infonce_fwd_impl(MODE_SIGMOID, E1_loc, ld1, b, E2_loc, ld2, b, E1_all, ld1a, n, E2_all, ld2a, n, D, q_offset,
                            log_scale, bias, nullptr, nullptr, loss, ws, ws_bytes, stream);
}

static int infonce_fwd_impl(int mode, const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                               const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2,
                               int D, int q_offset, const float* log_scale, const float* bias, float* lse_row,
                               float* lse_col, float* loss, void* ws, size_t ws_bytes, msn_stream_t stream) {
    if (int rc = check_common("msn_infonce_fwd", b1, b2, n1, n2, D, q_offset)) return rc;
    MSN_REQUIRE(E1_loc && E2_loc && E1_all && E2_all && log_scale && bias && loss, "msn_infonce_fwd: null pointer");
    MSN_REQUIRE(ld1 >= D && ld2 >= D && ld1a >= D && ld2a >= D, "msn_infonce_fwd: leading dimension < D");
    const Plan pl = make_plan(b1, b2, n1, n2, D);
    MSN_REQUIRE(ws && ws_bytes >= pl.total, "msn_infonce_fwd: workspace %zu < %zu bytes", ws_bytes, pl.total);
    NceArgs a = {};
    a.side[0] = Side{E2_loc, E1_all, nullptr, nullptr, nullptr, ld2, ld1a, 0, b2, n1};
    a.side[1] = Side{E1_loc, E2_all, nullptr, nullptr, nullptr, ld1, ld2a, 0, b1, n2};
    a.log_scale = log_scale; a.bias = bias; a.q_offset = q_offset; a.n_diag = std::min(n1, n2); a.D = D;
    a.ksplit = pl.ksplit; a.keys_per_split = pl.keys_per_split; a.maxq = pl.maxq; a.mode = mode;
    char* w = static_cast<char*>(ws);
    a.part_m = reinterpret_cast<float*>(w + pl.off_m);
    a.part_l = reinterpret_cast<float*>(w + pl.off_l);
    a.part_d = reinterpret_cast<float*>(w + pl.off_d);
    a.scal = reinterpret_cast<double*>(w + pl.off_scal);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the sigmoid loss sums each (i, j) once: direction 0 only
    const dim3 grid(pl.qtiles, pl.ksplit, mode == MODE_SIGMOID ? 1 : 2), block(256);
    MSN_NCE_DISPATCH(nce_fwd_kernel, grid, block, 0, st, a)
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_fwd_finish_kernel, dim3(1), dim3(1024), 0, st, a, lse_row, lse_col, loss,
                       pl.qtiles * pl.ksplit);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

static int infonce_bwd_impl(int mode, const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                            const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2, int D,
                            int q_offset, const float* log_scale, const float* bias, const float* lse_row_all,
                            const float* lse_col_all, const float* grad_out, float* dE1_loc, int64_t ldd1,
                            float* dE2_loc, int64_t ldd2, float* dscale_dbias, void* ws, size_t ws_bytes,
                            msn_stream_t stream);

extern "C" int msn_sigmoid_loss_bwd(const float* E1_loc, int64_t ld1, const float* E2_loc, int64_t ld2, int b,
                                    const float* E1_all, int64_t ld1a, const float* E2_all, int64_t ld2a, int n, int D,
                                    int q_offset, const float* log_scale, const float* bias, const float* grad_out,
                                    float* dE1_loc, int64_t ldd1, float* dE2_loc, int64_t ldd2, float* dscale_dbias,
                                    void* ws, size_t ws_bytes, msn_stream_t stream) {
    return infonce_bwd_impl(MODE_SIGMOID, E1_loc, ld1, b, E2_loc, ld2, b, E1_all, ld1a, n, E2_all, ld2a, n, D, q_offset,
                            log_scale, bias, nullptr, nullptr, grad_out, dE1_loc, ldd1, dE2_loc, ldd2, dscale_dbias, ws,
                            ws_bytes, stream);
}

extern "C" int msn_infonce_bwd(const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                               const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2,
                               int D, int q_offset, const float* log_scale, const float* bias,
                               const float* lse_row_all, const float* lse_col_all, const float* grad_out,
                               float* dE1_loc, int64_t ldd1, float* dE2_loc, int64_t ldd2, float* dscale_dbias,
                               void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(lse_row_all && lse_col_all, "msn_infonce_bwd: null pointer");
    return infonce_bwd_impl(MODE_SOFTMAX, E1_loc, ld1, b1, E2_loc, ld2, b2, E1_all, ld1a, n1, E2_all, ld2a, n2, D, q_offset,
                            log_scale, bias, lse_row_all, lse_col_all, grad_out, dE1_loc, ldd1, dE2_loc, ldd2,
                            dscale_dbias, ws, ws_bytes, stream);
}

static int infonce_bwd_impl(int mode, const float* E1_loc, int64_t ld1, int b1, const float* E2_loc, int64_t ld2, int b2,
                               const float* E1_all, int64_t ld1a, int n1, const float* E2_all, int64_t ld2a, int n2,
                               int D, int q_offset, const float* log_scale, const float* bias,
                               const float* lse_row_all, const float* lse_col_all, const float* grad_out,
                               float* dE1_loc, int64_t ldd1, float* dE2_loc, int64_t ldd2, float* dscale_dbias,
                               void* ws, size_t ws_bytes, msn_stream_t stream) {
    if (int rc = check_common("msn_infonce_bwd", b1, b2, n1, n2, D, q_offset)) return rc;
    MSN_REQUIRE(E1_loc && E2_loc && E1_all && E2_all && log_scale && bias && grad_out && dE1_loc && dE2_loc,
                "msn_infonce_bwd: null pointer");
    MSN_REQUIRE(ld1 >= D && ld2 >= D && ld1a >= D && ld2a >= D && ldd1 >= D && ldd2 >= D,
                "msn_infonce_bwd: leading dimension < D");
    const Plan pl = make_plan(b1, b2, n1, n2, D);
    MSN_REQUIRE(ws && ws_bytes >= pl.total, "msn_infonce_bwd: workspace %zu < %zu bytes", ws_bytes, pl.total);
    NceArgs a = {};
    // direction 0: queries = rows of S (E2), own LSE = row LSE, keys = E1 with the column LSE
    a.side[0] = Side{E2_loc, E1_all, lse_row_all, lse_col_all, dE2_loc, ld2, ld1a, ldd2, b2, n1};
    a.side[1] = Side{E1_loc, E2_all, lse_col_all, lse_row_all, dE1_loc, ld1, ld2a, ldd1, b1, n2};
    a.log_scale = log_scale; a.bias = bias; a.grad_out = grad_out; a.q_offset = q_offset; a.n_diag = std::min(n1, n2);
    a.D = D; a.ksplit = pl.ksplit; a.keys_per_split = pl.keys_per_split; a.maxq = pl.maxq; a.mode = mode;
    char* w = static_cast<char*>(ws);
    a.slab = reinterpret_cast<float*>(w + pl.off_slab);
    a.scal = reinterpret_cast<double*>(w + pl.off_scal);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(pl.qtiles, pl.ksplit, 2), block(256);
    MSN_NCE_DISPATCH(nce_bwd_kernel, grid, block, 0, st, a)
    MSN_LAUNCH_CHECK();
    // key splits > 1: fixed-order sum of the slabs; always: the (dscale, dbias) partials of direction 0
    const int64_t total = (int64_t)pl.maxq * D;
    const int blocks = pl.ksplit > 1 ? (int)std::min<int64_t>(cdiv(total, 256), 1024) : 1;
    hipLaunchKernelGGL(nce_bwd_finish_kernel, dim3(blocks), dim3(256), 0, st, a, pl.ksplit * pl.qtiles, dscale_dbias);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// rank[i] of the true partner (row i of E1) among all rows of E1 for query row i of E2; rows must be unit
// vectors (cosine similarity).  n <= 2^31 / D; workspace = msn_infonce_workspace_bytes(n, n, n, n, D).
extern "C" int msn_retrieval_rank(const float* E1, int64_t ld1, const float* E2, int64_t ld2, int n, int D, int* rank,
                                  void* ws, size_t ws_bytes, msn_stream_t stream) {
    if (int rc = check_common("msn_retrieval_rank", n, n, n, n, D, 0)) return rc;
    MSN_REQUIRE(E1 && E2 && rank && ld1 >= D && ld2 >= D, "msn_retrieval_rank: bad arguments");
    const Plan pl = make_plan(n, n, n, n, D);
    MSN_REQUIRE(ws && ws_bytes >= pl.total, "msn_retrieval_rank: workspace %zu < %zu bytes", ws_bytes, pl.total);
    NceArgs a = {};
    a.side[0] = Side{E2, E1, nullptr, nullptr, nullptr, ld2, ld1, 0, n, n};
    a.D = D; a.ksplit = pl.ksplit; a.keys_per_split = pl.keys_per_split; a.maxq = pl.maxq;
    int* part = reinterpret_cast<int*>(static_cast<char*>(ws) + pl.off_m);   // [ksplit][maxq] ints fit the (m) slab
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(pl.qtiles, pl.ksplit, 1), block(256);
    MSN_NCE_DISPATCH(nce_rank_kernel, grid, block, 0, st, a, part)
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_rank_finish_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, part, pl.ksplit, pl.maxq, n, rank);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
