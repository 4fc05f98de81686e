// fp32-grade GEMMs on the bf16 matrix cores from RESIDENT bf16 planes (v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
// Replaces, for the wide products of the build-defined ViT towers, every nn.Linear product the fp32 kernels of gemm.hip
// multiply (ref src/transformer_utils.py:45-47, 89, 102-106, 251 are the same Linear pattern).  The native fp32 matrix
// instruction runs at 1/16 of the bf16 rate and its kernel family tops out at 0.85 of that peak on long K
// (profiles/r03_gemm_kloop_ablation.txt); here an fp32 operand is held in HBM as NP bf16 PLANES
//
//      x = p0 + p1 + p2,   p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1)      (round to nearest even)
//
// -- an fp32 significand (24 bits) splits EXACTLY into three bf16 significands (8 bits + sign each), and bf16 has the
// fp32 exponent range, so no scaling is needed -- and a product is the sum of the plane products with pa + pb < NP:
//
//      NP = 3:  p0.q0 + p0.q1 + p1.q0 + p0.q2 + p2.q0 + p1.q1     6 bf16 MFMA products, dropped terms <= 2^-26 |a||b|
//      NP = 2:  p0.q0 + p0.q1 + p1.q0                             3 products, dropped terms <= 2^-17 |a||b|  ("bf16x3")
//
// every bf16 x bf16 product is exact in fp32 and the sums accumulate in fp32 inside the MFMA.  The split is done ONCE by
// the producer of a matrix (msn_plane_split, or the epilogue of the product that writes it), not inside the K loop.
//
// Plane matrix format ("blocked planes") of a logical R x C matrix: 32-row x 16-column blocks, block (rb, cb) holds NP
// consecutive 1-KB plane images [32 rows][16 bf16]:
//      byte offset of (r, c, plane) = (((r / 32) * CB + c / 16) * NP + plane) * 1024 + (r % 32) * 32 + (c % 16) * 2,
// CB = 2 ceil(C / 32) (an even number of column blocks: a product then has an even number of K-steps and a K-step's
// register roles alternate without a tail case), rows / columns past R / C are ZERO.  One plane image of one block is
// exactly one LDS-DMA piece (one wave instruction of global_load_lds_dwordx4): a contiguous 1-KB read, whatever K is.
//
//   msn_pgemm_nt :  C[M][N] = epi( A[M][K] . B[N][K]^T + bias )    both operands plane matrices with K on the columns
//                   (forward: activations x weight; dgrad: dY x W^T against the planes of the transposed weight)
//   msn_pgemm_tn :  C[N][K] = sum_m A[m][N]^T . B[m][K]             both operands plane matrices with the reduction on the
//                   ROWS (wgrad dY^T . X), fragments transposed out of LDS by ds_read_b64_tr_b16, reduction split over
//                   workgroups with fixed-order slab sums
//
// NT workgroup = 8 waves (2 x 4), tile 256 x BN (BN = 256 or 128), K-step 16 = one MFMA deep, ring of 3-4 LDS slots of
// (8 + BN / 32) * NP pieces; persistent (one workgroup per CU walks its tiles, the ring runs on across tile boundaries).


#include "pgemm_kernels.h"

// instantiated in pgemm_alt2.hip
namespace msn {
extern template __global__ void pgemm_nt_kernel<2, 128, true, false, 2, 4, false>(const PgemmArgs);
extern template __global__ void pgemm_nt_kernel<2, 128, false, false, 2, 4, false>(const PgemmArgs);
extern template __global__ void pgemm_nt_kernel<2, 128, false, true, 4, 2, false, true>(const PgemmArgs);
extern template __global__ void pgemm_tn_kernel<2, 128, true, false, 2, 4>(const PgemmArgs);
extern template __global__ void pgemm_tn_kernel<2, 128, false, false, 2, 4>(const PgemmArgs);
extern template __global__ void pgemm_tn_kernel<2, 128, true, true, 2, 4, true>(const PgemmArgs);
extern template __global__ void pgemm_tn_kernel<2, 128, false, true, 2, 4, true>(const PgemmArgs);
}  // namespace msn

namespace msn {

// Tail tiles of an NT product (PgemmArgs::tail_*): C tile = epilogue(sum over its K-segments' slabs, in K order).  One thread
// per float4 of a 256 x 128 tile; epilogues NONE / RELU / ADD (the products whose tails are split).
__global__ __launch_bounds__(256) void pgemm_tail_finish_kernel(const PgemmArgs p, int bn) {
    const int tiles_tail = p.tiles_m * p.tiles_n - p.tail_full;
    const int per_tile = BM * bn / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)tiles_tail * per_tile) return;
    const int tt = (int)(i / per_tile), e = (int)(i % per_tile);
    int tm, tn;
    nt_locate(p, p.tail_full + tt, tm, tn);       // the tile's position: the kernel's own map
    const int row = e / (bn / 4), c4 = e % (bn / 4);
    const int64_t m = (int64_t)tm * BM + row;
    const int n = tn * bn + 4 * c4;
    if (m >= p.M || n >= p.N) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < p.tail_segs; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(p.tail_slabs + ((int64_t)(tt * p.tail_segs + g) * BM + row) * bn + 4 * c4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (p.epi == MSN_EPI_ADD) {
        const float4 a = *reinterpret_cast<const float4*>(p.aux + m * p.ldaux + n);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    } else if (p.epi == MSN_EPI_RELU) {
        s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f);
    }
    *reinterpret_cast<float4*>(static_cast<float*>(p.C) + m * p.ldc + n) = s;
}

// C[i] = sum_s slab[s][i] in split order (float4 per thread, eight independent loads per wait)
__global__ void pgemm_slab_sum_kernel(const float* __restrict__ slabs, int splits, int64_t n4, int K4, int64_t ldc4,
                                      float* __restrict__ C) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < splits; k0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float4*>(slabs)[(int64_t)std::min(k0 + j, splits - 1) * n4 + i];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < splits) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        reinterpret_cast<float4*>(C)[(i / K4) * ldc4 + (i % K4)] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 [R][ld] -> blocked planes.  A wave converts two neighbouring blocks of one row block: lane = row (lane >> 1) x
// 16 columns (lane & 1): 64 contiguous bytes in, two 16-byte stores per plane out (each plane image is written whole by
// its 32 lanes: 1-KB bursts).  Column sums of x (bias gradients) ride along when `colpart` is given: [row blocks][C].
// fp16 planes (two: 22 significand bits): x * s = h0 + h1 with s = 2^e chosen from the matrix' largest magnitude m so that
// m s lies in [2^13, 2^14) -- fp16 carries 11 bits per plane only between 2^-14 and 2^16, and the second plane of an element
// is 2^-11 of the first, so elements down to 2^-16 m keep 22 bits and smaller ones an absolute error below 2^-38 m.
__device__ __forceinline__ float f16_plane_scale(unsigned amax_bits, float& inv) {
    const int ex = (int)((amax_bits >> 23) & 0xff) - 127;          // floor(log2 m); -127 for zero / subnormal m
    int e = amax_bits == 0 ? 0 : 13 - ex;
    e = e < -126 ? -126 : (e > 126 ? 126 : e);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
__device__ __forceinline__ void split_f16(float xs, u16& h0, u16& h1) {
    const _Float16 a = (_Float16)xs;                               // round to nearest even
    const _Float16 b = (_Float16)(xs - (float)a);                  // the residual is exact in fp32
    h0 = *reinterpret_cast<const u16*>(&a);
    h1 = *reinterpret_cast<const u16*>(&b);
}
template <int NP, bool F16>
__device__ __forceinline__ void split_any(float x, float s, u16 (&pl)[NP]) {
    if constexpr (F16) {
        static_assert(NP == 2, "fp16 planes come in pairs");
        split_f16(x * s, pl[0], pl[1]);
    } else {
        split_planes<NP>(x, pl);
    }
}

template <int NP, bool F16 = false>
__device__ __forceinline__ void plane_split_body(const float* __restrict__ x, int64_t ld, int64_t R, int C, int RB, int CB,
                                                 unsigned char* __restrict__ out, float* __restrict__ colpart, int64_t wg,
                                                 float s = 1.f) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = wg * 4 + (threadIdx.x >> 6);
    const int CB2 = (CB + 1) / 2;
    const int64_t nw = (int64_t)RB * CB2;
    if (wid >= nw) return;
    const int rb = (int)(wid / CB2), cb = 2 * (int)(wid % CB2) + (lane & 1);
    const int64_t r = (int64_t)rb * 32 + (lane >> 1);
    const int c0 = cb * 16;
    float v[16];
    if (r < R && c0 + 15 < C && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4* s = reinterpret_cast<const float4*>(x + r * ld + c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float4 t = s[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (r < R && c0 + q < C) ? x[r * ld + c0 + q] : 0.f;
    }
    if (cb < CB) {
        u16 pl[16][NP];
#pragma unroll
        for (int q = 0; q < 16; ++q) split_any<NP, F16>(v[q], s, pl[q]);
        unsigned char* dst = out + ((int64_t)rb * CB + cb) * (NP * PBLK) + (lane >> 1) * 32;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            uint4 a, b;
            a.x = pl[0][k] | ((unsigned)pl[1][k] << 16); a.y = pl[2][k] | ((unsigned)pl[3][k] << 16);
            a.z = pl[4][k] | ((unsigned)pl[5][k] << 16); a.w = pl[6][k] | ((unsigned)pl[7][k] << 16);
            b.x = pl[8][k] | ((unsigned)pl[9][k] << 16); b.y = pl[10][k] | ((unsigned)pl[11][k] << 16);
            b.z = pl[12][k] | ((unsigned)pl[13][k] << 16); b.w = pl[14][k] | ((unsigned)pl[15][k] << 16);
            *reinterpret_cast<uint4*>(dst + k * PBLK) = a;
            *reinterpret_cast<uint4*>(dst + k * PBLK + 16) = b;
        }
    }
    if (colpart) {      // sum over the 32 rows of the block: lanes of equal parity (xor 2, 4, .., 32)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float t = v[q];
#pragma unroll
            for (int o = 2; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
            v[q] = t;
        }
        if (lane < 2 && cb < CB) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (c0 + q < C) colpart[(int64_t)rb * C + c0 + q] = v[q];
        }
    }
}

template <int NP>
__global__ __launch_bounds__(256) void plane_split_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int C, int RB, int CB,
                                                          unsigned char* __restrict__ out, float* __restrict__ colpart) {
    plane_split_body<NP>(x, ld, R, C, RB, CB, out, colpart, blockIdx.x);
}

// planes of x^T: out is the plane matrix of the C x R transpose (weights: a few MB per step).  One workgroup per 32 x 32
// tile of x through LDS.
template <int NP, bool F16 = false>
__device__ __forceinline__ void plane_split_t_body(const float* __restrict__ x, int64_t ld, int R, int C, int CBT,
                                                   unsigned char* __restrict__ out, float (&t)[32][33], int bx, int by,
                                                   float s = 1.f) {
    const int r0 = by * 32, c0 = bx * 32;                        // tile of x; the transpose has rows c0.., columns r0..
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) t[i][tx] = (r0 + i < R && c0 + tx < C) ? x[(int64_t)(r0 + i) * ld + c0 + tx] : 0.f;
    __syncthreads();
    // transposed element (row c0 + i, column r0 + j) = t[j][i]; thread -> row i = threadIdx.x / 8, 4 columns j0 = 4 (threadIdx.x % 8)
    const int i = threadIdx.x >> 3, j0 = 4 * (threadIdx.x & 7);
    u16 pl[4][NP];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_any<NP, F16>(t[j0 + q][i], s, pl[q]);
    const int rbT = bx;                                            // row block of the transpose (32 rows = c0 .. c0 + 31)
    const int cbT = (r0 + j0) >> 4;                                // column block of the transpose
    if (cbT < CBT) {
        unsigned char* dst = out + ((int64_t)rbT * CBT + cbT) * (NP * PBLK) + i * 32 + ((r0 + j0) & 15) * 2;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            uint2 o;
            o.x = pl[0][k] | ((unsigned)pl[1][k] << 16);
            o.y = pl[2][k] | ((unsigned)pl[3][k] << 16);
            *reinterpret_cast<uint2*>(dst + k * PBLK) = o;
        }
    }
}
template <int NP>
__global__ __launch_bounds__(256) void plane_split_t_kernel(const float* __restrict__ x, int64_t ld, int R, int C, int CBT,
                                                            unsigned char* __restrict__ out) {
    __shared__ float t[32][33];
    plane_split_t_body<NP>(x, ld, R, C, CBT, out, t, blockIdx.x, blockIdx.y);
}

// ---- fp16 planes: largest magnitude (bits of a non-negative float order like unsigned integers), then the split with its scale
__global__ __launch_bounds__(256) void plane_absmax_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int C, unsigned* __restrict__ out) {
    float m = 0.f;
    if ((C & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const int C4 = C / 4;
        const int64_t n = R * C4;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const float4 v = *reinterpret_cast<const float4*>(x + (i / C4) * ld + 4 * (i % C4));
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        const int64_t n = R * C;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
            m = fmaxf(m, fabsf(x[(i / C) * ld + i % C]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
// scale[0] = 2^-e (what the products multiply their sums with), scale[1] = the bits of the largest magnitude
__global__ __launch_bounds__(256) void plane_split_f16_kernel(const float* __restrict__ x, int64_t ld, int64_t R, int C, int RB, int CB,
                                                              unsigned char* __restrict__ out, float* __restrict__ colpart,
                                                              float* __restrict__ scale) {
    float inv;
    const float s = f16_plane_scale(reinterpret_cast<const unsigned*>(scale)[1], inv);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = inv;
    plane_split_body<2, true>(x, ld, R, C, RB, CB, out, colpart, blockIdx.x, s);
}
__global__ __launch_bounds__(256) void plane_split_t_f16_kernel(const float* __restrict__ x, int64_t ld, int R, int C, int CBT,
                                                                unsigned char* __restrict__ out, float* __restrict__ scale) {
    __shared__ float t[32][33];
    float inv;
    const float s = f16_plane_scale(reinterpret_cast<const unsigned*>(scale)[1], inv);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale[0] = inv;
    plane_split_t_body<2, true>(x, ld, R, C, CBT, out, t, blockIdx.x, blockIdx.y, s);
}

// Several matrices in ONE launch (the weights of every block of a tower, or their transposes: 44 launches of 5-6 us each, every
// one a bubble between two products, become one).  The table travels as the kernel argument; a workgroup finds its matrix by
// its first-workgroup prefix.
constexpr int SPLIT_LIST_MAX = 64;
struct SplitEntry {
    const float* x;
    unsigned char* out;
    int64_t ld;
    int R, C, transposed, first, tiles_x, pad;
};
struct SplitTable {
    int n, pad;
    SplitEntry e[SPLIT_LIST_MAX];
};
template <int NP>
__global__ __launch_bounds__(256) void plane_split_list_kernel(const SplitTable tb) {
    __shared__ float t[32][33];
    int i = 0;
    for (int k = 1; k < tb.n; ++k) i += (int)blockIdx.x >= tb.e[k].first ? 1 : 0;      // entries are in workgroup order
    const SplitEntry& en = tb.e[i];
    const int w = blockIdx.x - en.first;
    if (en.transposed) {
        plane_split_t_body<NP>(en.x, en.ld, en.R, en.C, 2 * ((en.R + 31) / 32), en.out, t, w % en.tiles_x, w / en.tiles_x);
    } else {
        plane_split_body<NP>(en.x, en.ld, en.R, en.C, (en.R + 31) / 32, 2 * ((en.C + 31) / 32), en.out, nullptr, w);
    }
}

// blocked planes -> fp32 (sum of the planes, smallest first): tests and debugging
__global__ void plane_merge_kernel(const unsigned char* __restrict__ in, int NP, int64_t R, int C, int CB, float* __restrict__ y,
                                   int64_t ld) {
    const int64_t n = R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C;
        const int c = (int)(i % C);
        const unsigned char* s = in + ((r >> 5) * CB + (c >> 4)) * ((int64_t)NP * PBLK) + (r & 31) * 32 + (c & 15) * 2;
        float acc = 0.f;
        for (int k = NP - 1; k >= 0; --k) acc += bf2f(*reinterpret_cast<const u16*>(s + k * PBLK));
        y[r * ld + c] = acc;
    }
}

// out[y][n] = sum over the parts k of slice y (parts_per_slice each) of part[k][n]: 64 columns x 4 row groups per workgroup,
// fixed order.  Thousands of parts (one per 32-row block of a split, one per wave row of a GEMM) are summed in two passes:
// 32 slices, then the 32 slice sums (colsum_finish) -- one pass over 2080 parts with 6 workgroups took 32 us.
__global__ __launch_bounds__(256) void pcolsum_finish_kernel(const float* __restrict__ part, int nparts, int parts_per_slice, int N,
                                                             float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    const int k_begin = blockIdx.y * parts_per_slice, k_end = min(nparts, k_begin + parts_per_slice);
    float s = 0.f;
    if (n < N && k_begin < k_end) {
        const int cnt = k_end - k_begin;
        const int mine = (cnt - rg + 3) / 4;
        for (int k0 = 0; k0 < mine; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(int64_t)(k_begin + rg + 4 * std::min(k0 + j, mine - 1)) * N + n];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < mine) s += v[j];
        }
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && n < N) out[(int64_t)blockIdx.y * N + n] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}

// part: [nparts][N] followed by COLSUM_SLICES x N floats of scratch (declared in msn_common.h)
int colsum_finish(float* part, int nparts, int N, float* out, hipStream_t st) {
    const dim3 block(256);
    if (nparts <= 2 * COLSUM_SLICES) {
        hipLaunchKernelGGL(pcolsum_finish_kernel, dim3((unsigned)cdiv(N, 64), 1), block, 0, st, part, nparts, nparts, N, out);
    } else {
        float* tmp = part + (size_t)nparts * N;
        const int per = (int)cdiv(nparts, COLSUM_SLICES);
        hipLaunchKernelGGL(pcolsum_finish_kernel, dim3((unsigned)cdiv(N, 64), COLSUM_SLICES), block, 0, st, part, nparts, per, N, tmp);
        hipLaunchKernelGGL(pcolsum_finish_kernel, dim3((unsigned)cdiv(N, 64), 1), block, 0, st, tmp, COLSUM_SLICES, COLSUM_SLICES, N, out);
    }
    return hipGetLastError() == hipSuccess ? MSN_OK : MSN_ERR_HIP;
}

static bool aligned16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace msn

using namespace msn;

extern "C" size_t msn_colsum_workspace_bytes(int64_t M, int64_t N);       // gemm.hip
extern "C" int msn_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, msn_stream_t stream);

// (Round 6 retired the measurement switches of this file with their decided A/Bs: msn_set_pgemm_variant -- the 2 x 4 wave layout
// and the 16 x 16 x 32 form of the 3-plane NT kernel, both slower or equal: profiles/r04_pgemm_variants.txt, r05_experiments_tried.txt
// item 10 --, msn_set_pgemm_skew -- a start skew of the workgroups: nothing in time, and more than a few K-steps of it destroy the
// sharing of operand panels in L2: r06_experiments_tried.txt item 1 --, msn_set_pgemm_tile_n and msn_set_pgemm_walk.)

extern "C" size_t msn_plane_bytes(int64_t R, int64_t C, int planes) {
    if (R <= 0 || C <= 0 || planes < 1 || planes > 3) return 0;
    return (size_t)cdiv(R, 32) * (size_t)(2 * cdiv(C, 32)) * (size_t)planes * PBLK;
}

extern "C" size_t msn_plane_split_colsum_workspace_bytes(int64_t R, int64_t C) {
    if (R <= 0 || C <= 0) return 0;
    return sizeof(float) * ((size_t)cdiv(R, 32) + COLSUM_SLICES) * (size_t)C;
}

extern "C" int msn_plane_split(const float* x, int64_t ldx, int64_t R, int64_t C, int planes, int transposed, void* out,
                               float* colsum, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(x && out && R > 0 && C > 0 && ldx >= C, "msn_plane_split: bad operand");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_plane_split: planes must be 2 or 3 (got %d)", planes);
    MSN_REQUIRE(aligned16p(out), "msn_plane_split: the plane matrix must be 16-byte aligned");
    MSN_REQUIRE(R < (1ll << 31) && C < (1ll << 31), "msn_plane_split: matrix too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (transposed) {
        MSN_REQUIRE(!colsum, "msn_plane_split: column sums only for the untransposed form");
        const int CBT = 2 * (int)cdiv(R, 32);                       // the transpose is C x R
        const dim3 grid((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32));
        if (planes == 3) hipLaunchKernelGGL(plane_split_t_kernel<3>, grid, dim3(256), 0, st, x, ldx, (int)R, (int)C, CBT, static_cast<unsigned char*>(out));
        else hipLaunchKernelGGL(plane_split_t_kernel<2>, grid, dim3(256), 0, st, x, ldx, (int)R, (int)C, CBT, static_cast<unsigned char*>(out));
        MSN_LAUNCH_CHECK();
        return MSN_OK;
    }
    const int RB = (int)cdiv(R, 32), CB = 2 * (int)cdiv(C, 32);
    float* part = nullptr;
    if (colsum) {
        const size_t need = msn_plane_split_colsum_workspace_bytes(R, C);
        MSN_REQUIRE(ws && ws_bytes >= need, "msn_plane_split: column-sum workspace %zu < %zu bytes", ws_bytes, need);
        part = static_cast<float*>(ws);
    }
    const int64_t nw = (int64_t)RB * ((CB + 1) / 2);
    const dim3 grid((unsigned)cdiv(nw, 4));
    if (planes == 3) hipLaunchKernelGGL(plane_split_kernel<3>, grid, dim3(256), 0, st, x, ldx, R, (int)C, RB, CB, static_cast<unsigned char*>(out), part);
    else hipLaunchKernelGGL(plane_split_kernel<2>, grid, dim3(256), 0, st, x, ldx, R, (int)C, RB, CB, static_cast<unsigned char*>(out), part);
    MSN_LAUNCH_CHECK();
    if (colsum) {
        if (colsum_finish(part, RB, (int)C, colsum, st) != MSN_OK) {
            set_error("msn_plane_split: column-sum launch failed");
            return MSN_ERR_HIP;
        }
    }
    return MSN_OK;
}

extern "C" int msn_plane_split_list(int n, const msn_split_item* items, int planes, msn_stream_t stream) {
    MSN_REQUIRE(items && n > 0, "msn_plane_split_list: empty list");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_plane_split_list: planes must be 2 or 3 (got %d)", planes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += SPLIT_LIST_MAX) {
        SplitTable tb = {};
        tb.n = std::min(n - base, SPLIT_LIST_MAX);
        int64_t wgs = 0;
        for (int k = 0; k < tb.n; ++k) {
            const msn_split_item& it = items[base + k];
            MSN_REQUIRE(it.x && it.out && it.R > 0 && it.C > 0 && it.ldx >= it.C, "msn_plane_split_list: bad operand in item %d", base + k);
            MSN_REQUIRE(aligned16p(it.out), "msn_plane_split_list: the plane matrix of item %d must be 16-byte aligned", base + k);
            MSN_REQUIRE(it.R < (1ll << 31) && it.C < (1ll << 31), "msn_plane_split_list: item %d too large", base + k);
            SplitEntry& e = tb.e[k];
            e.x = it.x, e.out = static_cast<unsigned char*>(it.out), e.ld = it.ldx, e.R = (int)it.R, e.C = (int)it.C;
            e.transposed = it.transposed ? 1 : 0;
            e.first = (int)wgs;
            e.tiles_x = (int)cdiv(it.C, 32);
            const int64_t rb = cdiv(it.R, 32), cb2 = cdiv(it.C, 32);
            wgs += e.transposed ? rb * cb2 : cdiv(rb * cb2, 4);
            MSN_REQUIRE(wgs < (1ll << 31), "msn_plane_split_list: list too large");
        }
        if (planes == 3) hipLaunchKernelGGL(plane_split_list_kernel<3>, dim3((unsigned)wgs), dim3(256), 0, st, tb);
        else hipLaunchKernelGGL(plane_split_list_kernel<2>, dim3((unsigned)wgs), dim3(256), 0, st, tb);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}

extern "C" int msn_plane_split_f16(const float* x, int64_t ldx, int64_t R, int64_t C, int transposed, void* out, float* scale,
                                   int reuse_scale, float* colsum, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(x && out && scale && R > 0 && C > 0 && ldx >= C, "msn_plane_split_f16: bad operand");
    MSN_REQUIRE(aligned16p(out), "msn_plane_split_f16: the plane matrix must be 16-byte aligned");
    MSN_REQUIRE(R < (1ll << 31) && C < (1ll << 31), "msn_plane_split_f16: matrix too large");
    MSN_REQUIRE(!(transposed && colsum), "msn_plane_split_f16: column sums only for the untransposed form");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!reuse_scale) {
        if (hipMemsetAsync(scale + 1, 0, sizeof(float), st) != hipSuccess) {
            set_error("msn_plane_split_f16: memset failed");
            return MSN_ERR_HIP;
        }
        const int64_t n4 = R * ((C + 3) / 4);
        hipLaunchKernelGGL(plane_absmax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4, 256 * 4), 2048)), dim3(256), 0, st, x, ldx, R,
                           (int)C, reinterpret_cast<unsigned*>(scale + 1));
        MSN_LAUNCH_CHECK();
    }
    if (transposed) {
        const int CBT = 2 * (int)cdiv(R, 32);
        hipLaunchKernelGGL(plane_split_t_f16_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32)), dim3(256), 0, st, x, ldx, (int)R,
                           (int)C, CBT, static_cast<unsigned char*>(out), scale);
        MSN_LAUNCH_CHECK();
        return MSN_OK;
    }
    const int RB = (int)cdiv(R, 32), CB = 2 * (int)cdiv(C, 32);
    float* part = nullptr;
    if (colsum) {
        const size_t need = msn_plane_split_colsum_workspace_bytes(R, C);
        MSN_REQUIRE(ws && ws_bytes >= need, "msn_plane_split_f16: column-sum workspace %zu < %zu bytes", ws_bytes, need);
        part = static_cast<float*>(ws);
    }
    const int64_t nw = (int64_t)RB * ((CB + 1) / 2);
    hipLaunchKernelGGL(plane_split_f16_kernel, dim3((unsigned)cdiv(nw, 4)), dim3(256), 0, st, x, ldx, R, (int)C, RB, CB,
                       static_cast<unsigned char*>(out), part, scale);
    MSN_LAUNCH_CHECK();
    if (colsum && colsum_finish(part, RB, (int)C, colsum, st) != MSN_OK) {
        set_error("msn_plane_split_f16: column-sum launch failed");
        return MSN_ERR_HIP;
    }
    return MSN_OK;
}

extern "C" int msn_plane_merge(const void* planes_in, int planes, int64_t R, int64_t C, float* y, int64_t ldy, msn_stream_t stream) {
    MSN_REQUIRE(planes_in && y && R > 0 && C > 0 && ldy >= C && planes >= 1 && planes <= 3, "msn_plane_merge: bad operand");
    const int64_t n = R * C;
    hipLaunchKernelGGL(plane_merge_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const unsigned char*>(planes_in), planes, R, (int)C,
                       2 * (int)cdiv(C, 32), y, ldy);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" size_t msn_pgemm_nt_colsum_workspace_bytes(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return sizeof(float) * (4 * (size_t)cdiv(M, BM) + COLSUM_SLICES) * (size_t)N;          // up to 4 wave rows per tile row
}

// ---- tail split of msn_pgemm_nt: the tiles that do not fill a round of the 256 persistent workgroups
namespace {
struct NtTail { int full, segs, steps; };
static int g_pgemm_tail = 1;
NtTail nt_tail_plan(int64_t M, int N, int K, int c_planes, int epilogue, bool want_colsum) {
    NtTail t = {0, 0, 0};
    if (!g_pgemm_tail || c_planes || want_colsum) return t;
    if (epilogue != MSN_EPI_NONE && epilogue != MSN_EPI_RELU && epilogue != MSN_EPI_ADD) return t;
    const int total = (int)(cdiv(M, BM) * cdiv(N, 128)), G = 256;
    const int nk = 2 * (int)cdiv(K, 32);
    if (total <= G / 2) {
        // fewer tiles than half the chip (the 128 - 256 rows per GPU of a strong-scaling run): EVERY tile is cut, so that the
        // product's K-steps spread over the idle CUs (99 tiles of K = 1536: 70 -> 40 us)
        if (nk < 24) return t;
        int segs = std::min(std::min(G / total, nk / 4), 8);
        if (segs < 2) return t;
        t.steps = 2 * (int)cdiv(nk, 2 * segs);
        t.segs = (int)cdiv(nk, t.steps);
        t.full = 0;
        if (t.segs < 2) t = {0, 0, 0};
        return t;
    }
    if (total <= G) return t;
    const int left = total % G;
    // short reductions do not gain: a unit's slab (128 KB) and the finishing launch cost what the cut saves (K = 384: proj
    // 131 -> 131 us, qkv 360 -> 366, fc1 433 -> 448; K = 1536: 511 -> 472, K = 1152: 399 -> 373)
    if (left == 0 || left > G / 2 || nk < 48) return t;
    int segs = std::min(std::min(G / left, nk / 2), 16);
    if (segs < 2) return t;
    t.steps = 2 * (int)cdiv(nk, 2 * segs);
    t.segs = (int)cdiv(nk, t.steps);
    t.full = total - left;
    if (t.segs < 2) t = {0, 0, 0};
    return t;
}
}  // namespace
extern "C" int msn_set_pgemm_tail_split(int enabled) {
    g_pgemm_tail = enabled ? 1 : 0;
    return MSN_OK;
}

extern "C" size_t msn_pgemm_nt_workspace_bytes(int64_t M, int N, int K, int planes, int c_planes, int epilogue, int want_colsum) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (want_colsum) return std::max(msn_pgemm_nt_colsum_workspace_bytes(M, N), msn_colsum_workspace_bytes(M, N));   // (chunked K: see below)
    const NtTail t = nt_tail_plan(M, N, K, c_planes, epilogue, false);
    const int total = (int)(cdiv(M, BM) * cdiv(N, 128));
    return t.segs ? sizeof(float) * (size_t)(total - t.full) * t.segs * BM * 128 : 0;
}

template <int NP, int BN, bool DUAL, int WM = 2, int WN = 4, bool STAG = false>
static void launch_nt(const PgemmArgs& a, bool outp, int grid, hipStream_t st) {
    if (outp) hipLaunchKernelGGL((pgemm_nt_kernel<NP, BN, true, DUAL, WM, WN, STAG>), dim3((unsigned)grid), dim3(64 * WM * WN), 0, st, a);
    else hipLaunchKernelGGL((pgemm_nt_kernel<NP, BN, false, DUAL, WM, WN, STAG>), dim3((unsigned)grid), dim3(64 * WM * WN), 0, st, a);
}

static int pgemm_nt_impl(int64_t M, int N, int K, int planes, const void* A, const void* B, void* C, int64_t ldc,
                         int c_planes, const float* bias, int epilogue, float* aux, int64_t ldaux, float* colsum_out,
                         void* ws, size_t ws_bytes, msn_stream_t stream, const float* scaleA, const float* scaleB) {
    const bool f16 = scaleA != nullptr;
    MSN_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "msn_pgemm_nt: empty operand");
    MSN_REQUIRE(!f16 || (scaleB && planes == 2 && !c_planes), "msn_pgemm_nt_f16: two fp16 planes per operand, both scales, fp32 result");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_pgemm_nt: planes must be 2 or 3 (got %d)", planes);
    MSN_REQUIRE(N % 4 == 0 && (!c_planes || N % 16 == 0), "msn_pgemm_nt: N = %d must be a multiple of 4 (16 for a plane output)", N);
    MSN_REQUIRE(aligned16p(A) && aligned16p(B) && aligned16p(C) && (!bias || aligned16p(bias)), "msn_pgemm_nt: operands must be 16-byte aligned");
    MSN_REQUIRE(c_planes || (ldc >= N && ldc % 4 == 0), "msn_pgemm_nt: bad output row stride %lld", (long long)ldc);
    MSN_REQUIRE(epilogue >= MSN_EPI_NONE && epilogue <= MSN_EPI_ADD, "msn_pgemm_nt: unknown epilogue %d", epilogue);
    const bool needs_aux = epilogue == MSN_EPI_RELU_BWD || epilogue == MSN_EPI_GELU_BWD || epilogue == MSN_EPI_ADD;
    MSN_REQUIRE(!needs_aux || aux, "msn_pgemm_nt: epilogue %d needs an aux matrix", epilogue);
    MSN_REQUIRE(!aux || (ldaux >= N && ldaux % 4 == 0 && aligned16p(aux)), "msn_pgemm_nt: bad aux matrix");
    MSN_REQUIRE(M < (1ll << 31) * 32, "msn_pgemm_nt: too many rows");
    // the epilogue addresses C / aux through buffer descriptors per tile row with 32-bit lane offsets (pgemm_kernels.h)
    MSN_REQUIRE(N < (1 << 20) && (c_planes || ldc < (1 << 20)) && (!aux || ldaux < (1 << 20)), "msn_pgemm_nt: row strides must be below 2^20 elements");
    // the ring addresses its sources by 32-bit byte offsets inside a 2-GB descriptor: up to 8 row blocks of K / 16 column blocks, `planes` 1-KB images each
    MSN_REQUIRE(K < (1 << 19), "msn_pgemm_nt: K must be below 2^19 (32-bit offsets inside the row blocks of a tile)");
    PgemmArgs a = {};
    a.A = static_cast<const unsigned char*>(A); a.B = static_cast<const unsigned char*>(B); a.C = C;
    a.aux = aux; a.bias = bias; a.ldc = ldc; a.ldaux = ldaux; a.M = M; a.N = N; a.K = K;
    a.rbA = (int)cdiv(M, 32); a.rbB = (int)cdiv(N, 32);
    a.cbA = a.cbB = 2 * (int)cdiv(K, 32);
    a.cbC = 2 * (int)cdiv(N, 32);
    a.epi = epilogue;
    // tile width 128: the 3-plane form carries two accumulator sets (see the kernel), which fill the register file at 256 x 128;
    // the 2-plane form measured faster at 128 than at 256 on every headline shape (profiles/r04_pgemm_variants.txt)
    const int bn = 128;
    a.tiles_m = (int)cdiv(M, BM); a.tiles_n = (int)cdiv(N, bn);
    // super-rows: tile-rows walked together while their A panels (256 rows x K x 2 NP bytes) fit half an L2
    a.super_rows = (int)std::max<int64_t>(1, std::min<int64_t>(8, (2 << 20) / ((int64_t)BM * K * 2 * planes)));
    // fp32 grade: reductions longer than 768 columns are cut into chunks of at most 512 (the partial sums meet in C by fp32 adds)
    a.chunk_steps = 0;
    {
        const int cs = 32;                                             // K-steps per chunk
        if ((planes == 3 || f16) && !c_planes && a.cbA > std::max(cs, 48)) {
            const int nch = (int)cdiv(a.cbA, cs);
            a.chunk_steps = 2 * (int)cdiv(a.cbA, 2 * nch);
        }
    }
    a.scaleA = scaleA, a.scaleB = scaleB;
    const NtTail tail = nt_tail_plan(M, N, K, c_planes, epilogue, colsum_out != nullptr);
    if (tail.segs && ws && ws_bytes >= msn_pgemm_nt_workspace_bytes(M, N, K, planes, c_planes, epilogue, 0) && aligned16p(ws)) {
        a.tail_full = tail.full; a.tail_segs = tail.segs; a.tail_steps = tail.steps;
        a.tail_slabs = static_cast<float*>(ws);
    }
    a.colpart = nullptr;
    // a reduction cut into K chunks meets its partial sums in C: the column sums of the finished values are then taken from C
    // by msn_colsum behind the product (the chunked epilogues carry no column-sum form)
    const bool colsum_after = colsum_out && a.chunk_steps > 0;
    if (colsum_out && !colsum_after) {
        const size_t need = msn_pgemm_nt_colsum_workspace_bytes(M, N);
        MSN_REQUIRE(ws && ws_bytes >= need && aligned16p(ws), "msn_pgemm_nt: column-sum workspace %zu < %zu bytes", ws_bytes, need);
        a.colpart = static_cast<float*>(ws);
    }
    const int total = a.tiles_m * a.tiles_n;
    // one persistent workgroup per CU; a launch whose tiles are ALL cut (tail_full == 0) has one workgroup per unit
    const int grid = a.tail_segs ? std::max(a.tail_full ? 256 : 0, (total - a.tail_full) * a.tail_segs) : std::min(total, 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (planes == 3) {
        a.colsum_rows = 4;
        launch_nt<3, 128, true, 4, 2>(a, c_planes != 0, grid, st);      // 4 x 2 waves of 64 x 64 (the 2 x 4 layout: -7 ... -12 %)
    } else if (f16) {
        a.colsum_rows = 4;
        hipLaunchKernelGGL((pgemm_nt_kernel<2, 128, false, true, 4, 2, false, true>), dim3((unsigned)grid), dim3(512), 0, st, a);
    } else {
        launch_nt<2, 128, false>(a, c_planes != 0, grid, st);
    }
    MSN_LAUNCH_CHECK();
    if (a.tail_segs) {
        const int64_t threads = (int64_t)(total - a.tail_full) * (BM * bn / 4);
        hipLaunchKernelGGL(pgemm_tail_finish_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, a, bn);
        MSN_LAUNCH_CHECK();
    }
    if (colsum_after) return msn_colsum(static_cast<const float*>(C), ldc, M, N, colsum_out, ws, ws_bytes, stream);
    if (colsum_out) {
        if (colsum_finish(a.colpart, (a.colsum_rows ? a.colsum_rows : 2) * a.tiles_m, N, colsum_out, st) != MSN_OK) {
            set_error("msn_pgemm_nt: column-sum launch failed");
            return MSN_ERR_HIP;
        }
    }
    return MSN_OK;
}

extern "C" int msn_pgemm_nt(int64_t M, int N, int K, int planes, const void* A, const void* B, void* C, int64_t ldc,
                            int c_planes, const float* bias, int epilogue, float* aux, int64_t ldaux, float* colsum_out,
                            void* ws, size_t ws_bytes, msn_stream_t stream) {
    return pgemm_nt_impl(M, N, K, planes, A, B, C, ldc, c_planes, bias, epilogue, aux, ldaux, colsum_out, ws, ws_bytes, stream,
                         nullptr, nullptr);
}
extern "C" int msn_pgemm_nt_f16(int64_t M, int N, int K, const void* A, const float* scaleA, const void* B, const float* scaleB,
                                float* C, int64_t ldc, const float* bias, int epilogue, float* aux, int64_t ldaux,
                                float* colsum_out, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(scaleA && scaleB, "msn_pgemm_nt_f16: the operands' scales are missing");
    return pgemm_nt_impl(M, N, K, 2, A, B, C, ldc, 0, bias, epilogue, aux, ldaux, colsum_out, ws, ws_bytes, stream, scaleA, scaleB);
}

// ---- TN host side: orientation, tile width and reduction split
namespace {
struct TnPlan { bool swap; int bq, tiles_p, tiles_q, splits, rb_per_split; };
TnPlan tn_plan(int64_t M, int N, int K, int planes) {
    TnPlan t;
    // the 256-wide side goes to the dimension with fewer wasted columns (ties: N, the non-swapped form)
    auto waste = [](int n, int w) { return (int)(cdiv(n, w) * w - n); };
    const int wn = waste(N, 256), wk = waste(K, 256);
    t.swap = wk < wn;
    const int Pd = t.swap ? K : N, Qd = t.swap ? N : K;
    t.bq = 128;      // (3 planes: two accumulator sets fill the register file at 256 x 128; 2 planes: 128 measured faster)
    t.tiles_p = (int)cdiv(Pd, 256);
    t.tiles_q = (int)cdiv(Qd, t.bq);
    const int tiles = t.tiles_p * t.tiles_q;
    const int rbs = (int)cdiv(M, 32);
    int s = std::max(1, 256 / tiles);                 // enough workgroups for the 256 CUs (one 144-KB workgroup per CU)
    // the kernel addresses a split's row blocks through a buffer descriptor with 32-bit offsets: a split spans < 2 GB of either operand
    const int64_t rb_bytes = 2 * std::max(cdiv(N, 32), cdiv(K, 32)) * (int64_t)planes * 1024;
    s = (int)std::max<int64_t>(s, cdiv(rbs, std::max<int64_t>(1, ((1ll << 31) - 1) / rb_bytes)));
    s = std::min(s, rbs);
    t.rb_per_split = (int)cdiv(rbs, s);
    t.splits = (int)cdiv(rbs, t.rb_per_split);
    return t;
}
}  // namespace

extern "C" size_t msn_pgemm_tn_workspace_bytes(int64_t M, int N, int K, int planes) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan t = tn_plan(M, N, K, planes);
    return t.splits > 1 ? sizeof(float) * (size_t)t.splits * N * K : 0;
}

template <int NP, int BQ, bool DUAL, int WM = 2, int WN = 4, int FOLDN = 1>
static void launch_tn(const PgemmArgs& a, bool swap, int grid, hipStream_t st) {
    if (swap) hipLaunchKernelGGL((pgemm_tn_kernel<NP, BQ, true, DUAL, WM, WN, false, FOLDN>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((pgemm_tn_kernel<NP, BQ, false, DUAL, WM, WN, false, FOLDN>), dim3((unsigned)grid), dim3(512), 0, st, a);
}

static int pgemm_tn_impl(int64_t M, int N, int K, int planes, const void* A, const void* B, float* C, int64_t ldc, void* ws,
                         size_t ws_bytes, msn_stream_t stream, const float* scaleA, const float* scaleB) {
    const bool f16 = scaleA != nullptr;
    MSN_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "msn_pgemm_tn: empty operand");
    MSN_REQUIRE(!f16 || (scaleB && planes == 2), "msn_pgemm_tn_f16: two fp16 planes per operand, both scales");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_pgemm_tn: planes must be 2 or 3 (got %d)", planes);
    MSN_REQUIRE(K % 4 == 0 && ldc >= K && ldc % 4 == 0 && aligned16p(C) && aligned16p(A) && aligned16p(B),
                "msn_pgemm_tn: K and ldc must be multiples of 4, operands 16-byte aligned");
    MSN_REQUIRE(std::max(N, K) < (1 << 19), "msn_pgemm_tn: N and K must be below 2^19 (32-bit offsets inside a row block of planes)");
    const TnPlan t = tn_plan(M, N, K, planes);
    PgemmArgs a = {};
    a.A = static_cast<const unsigned char*>(A); a.B = static_cast<const unsigned char*>(B); a.C = C;
    a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.rbA = a.rbB = (int)cdiv(M, 32);
    a.cbA = 2 * (int)cdiv(N, 32); a.cbB = 2 * (int)cdiv(K, 32);
    a.tiles_m = t.tiles_p; a.tiles_n = t.tiles_q;
    a.splits = t.splits; a.rb_per_split = t.rb_per_split;
    const size_t need = t.splits > 1 ? sizeof(float) * (size_t)t.splits * N * K : 0;
    MSN_REQUIRE(need == 0 || (ws && ws_bytes >= need && aligned16p(ws)), "msn_pgemm_tn: workspace %zu < %zu bytes", ws_bytes, need);
    a.slabs = static_cast<float*>(ws);
    a.scaleA = scaleA, a.scaleB = scaleB;
    const int grid = t.tiles_p * t.tiles_q * t.splits;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // (2 x 4 waves: the 64 x 64 wave tiles that gain 7-12 % on the NT kernel LOSE 3-4 % here -- 374 / 151 / 452 / 454 us
    // against 384 / 157 / 469 / 470 on the four headline weight gradients; that instantiation is no longer built)
    // (three planes: each tile's p0 q0 temporaries run over TWO K-steps before they are folded into the running sums, the four tiles
    // of a wave staggered so that two finish per step -- 351 / 151 / 426 / 426 us against 372 / 160 / 450 / 452 with a fold per step
    // on the four headline weight gradients, and 0.2 - 0.4 x the exact-fp32 kernel's error either way)
    if (planes == 3) launch_tn<3, 128, true, 2, 4, 2>(a, t.swap, grid, st);
    else if (f16 && t.swap) hipLaunchKernelGGL((pgemm_tn_kernel<2, 128, true, true, 2, 4, true>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else if (f16) hipLaunchKernelGGL((pgemm_tn_kernel<2, 128, false, true, 2, 4, true>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else launch_tn<2, 128, false>(a, t.swap, grid, st);
    MSN_LAUNCH_CHECK();
    if (t.splits > 1) {
        const int64_t n4 = (int64_t)N * K / 4;
        hipLaunchKernelGGL(pgemm_slab_sum_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4, 256), 2048)), dim3(256), 0, st,
                           a.slabs, t.splits, n4, K / 4, ldc / 4, C);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}

extern "C" int msn_pgemm_tn(int64_t M, int N, int K, int planes, const void* A, const void* B, float* C, int64_t ldc, void* ws,
                            size_t ws_bytes, msn_stream_t stream) {
    return pgemm_tn_impl(M, N, K, planes, A, B, C, ldc, ws, ws_bytes, stream, nullptr, nullptr);
}
extern "C" int msn_pgemm_tn_f16(int64_t M, int N, int K, const void* A, const float* scaleA, const void* B, const float* scaleB,
                                float* C, int64_t ldc, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(scaleA && scaleB, "msn_pgemm_tn_f16: the operands' scales are missing");
    return pgemm_tn_impl(M, N, K, 2, A, B, C, ldc, ws, ws_bytes, stream, scaleA, scaleB);
}
