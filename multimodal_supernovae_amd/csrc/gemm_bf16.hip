// GEMM on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate) for fp32 tensors.
//
// PLANES = 2 ("split" mode): every fp32 operand element a is split on the fly into two bf16 numbers,
// hi = bf16(a) and lo = bf16(a - hi), and the product is accumulated in fp32 as
//     a.b  ~=  hi_a.hi_b + hi_a.lo_b + lo_a.hi_b           (3 MFMAs per algorithmic MAC tile)
// The dropped terms are <= ~3 * 2^-18 |a||b| (1.1e-5 relative per product; fp32 itself is 6e-8), two
// orders inside the 1e-3 parity bar, while the matrix pipe runs 16/3 = 5.3x faster than in fp32.
// PLANES = 1: operands rounded to bf16, one product (the "bf16 on MFMA" configuration of BASELINE cfg5).
// Inputs and outputs stay fp32 in HBM; the split happens in registers between the global load and the
// LDS write, so no extra pass over memory and the same C-ABI.
//
// Structure = the fp32 kernel's: 128 x BN tile, BK = 32, 4 waves x (2 x 2) 32x32 tiles, LDS double
// buffer of bf16 planes [row][k] (row stride 40 bf16 = 80 B: the 16-B fragment reads of 16 consecutive
// rows hit 16 different bank quads), next K-step prefetched into registers as raw fp32.  A lane's
// MFMA fragment is 8 consecutive k of one row = ONE ds_read_b128.  K-contiguous operands are loaded
// with 16-B global loads; K-major operands ([k][rows] in memory: dgrad weights, both wgrad operands)
// are loaded with one row per lane (coalesced 4-B loads along rows, 4..16 consecutive k per thread)
// so that the transposition to [row][k] costs nothing but the register->LDS write.
#include <algorithm>

#include "gemm_common.h"

namespace msn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BBK = 32;   // K-step
constexpr int BRS = 40;   // LDS row stride in bf16 elements

template <int ROWS, bool KMAJOR, int PLANES>
struct BTile {
    static constexpr int kPlane = ROWS * BRS;            // bf16 elements per plane
    static constexpr int kElems = PLANES * kPlane;
    static constexpr int kRegs = ROWS / 8;               // fp32 values held per thread per K-step
    static constexpr int kVec = kRegs / 4;               // float4 loads (K-contiguous operand)
    static constexpr int kPerThreadK = kRegs;            // consecutive k per thread (K-major operand)

    // per-thread source pointers, computed once (rows / columns clamped into the matrix: whatever a
    // clamped lane loads only feeds outputs that are never stored)
    __device__ static __forceinline__ void init_ptrs(const float* (&ptr)[kVec > 0 ? kVec : 1], const float*& kptr,
                                                     const float* __restrict__ g, int64_t ld, int64_t row0,
                                                     int64_t nrows, int64_t k_begin) {
        const int t = threadIdx.x;
        if (KMAJOR) {
            int64_t r = row0 + (t % ROWS);
            r = r < nrows ? r : nrows - 1;
            kptr = g + (k_begin + (int64_t)(t / ROWS) * kPerThreadK) * ld + r;
        } else {
#pragma unroll
            for (int i = 0; i < kVec; ++i) {
                const int idx = t + 256 * i;
                int64_t r = row0 + idx / 8;
                r = r < nrows ? r : nrows - 1;
                ptr[i] = g + r * ld + k_begin + 4 * (idx % 8);
            }
        }
    }
    // raw fp32 of K-step `kt` -> registers; `klim` = number of valid k in this step (BBK when full)
    __device__ static __forceinline__ void fetch(float (&reg)[kRegs], const float* const (&ptr)[kVec > 0 ? kVec : 1],
                                                 const float* kptr, int64_t ld, int kt, int klim) {
        const int t = threadIdx.x;
        if (KMAJOR) {
            const int kb = (t / ROWS) * kPerThreadK;
            const float* p = kptr + (int64_t)kt * BBK * ld;
            const float* safe = kptr - (int64_t)kb * ld;          // (k_begin, row): always inside the matrix
#pragma unroll
            for (int j = 0; j < kPerThreadK; ++j) {
                const bool in = kb + j < klim;
                reg[j] = *(in ? p + (int64_t)j * ld : safe);        // masked later, in stash()
            }
        } else {
#pragma unroll
            for (int i = 0; i < kVec; ++i) {
                const int k = 4 * ((t + 256 * i) % 8);
                const bool in = k < klim;                       // K % 4 == 0: a vector is all in or all out
                const float4 v = *reinterpret_cast<const float4*>(ptr[i] + (in ? (int64_t)kt * BBK : -(int64_t)k));
                reg[4 * i + 0] = v.x;                                // masked later, in stash(): a select here
                reg[4 * i + 1] = v.y;                                // would make the loads be waited for before
                reg[4 * i + 2] = v.z;                                // the MFMA block instead of after it
                reg[4 * i + 3] = v.w;
            }
        }
    }
    // zero the k beyond `klim`, split to bf16 plane(s) and write [row][k]
    __device__ static __forceinline__ void stash(const float (&reg)[kRegs], __bf16* lds, int klim) {
        const int t = threadIdx.x;
        if (KMAJOR) {
            const int kb = (t / ROWS) * kPerThreadK;
            __bf16* dst = lds + (t % ROWS) * BRS + kb;
#pragma unroll
            for (int j0 = 0; j0 < kPerThreadK; j0 += 4) {
                bf16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = kb + j0 + j < klim ? reg[j0 + j] : 0.f;
                    hi[j] = (__bf16)a;
                    if (PLANES == 2) lo[j] = (__bf16)(a - (float)hi[j]);
                }
                *reinterpret_cast<bf16x4*>(dst + j0) = hi;
                if (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kPlane + j0) = lo;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kVec; ++i) {
                const int idx = t + 256 * i;
                __bf16* dst = lds + (idx / 8) * BRS + 4 * (idx % 8);
                const bool in = 4 * (idx % 8) < klim;
                bf16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = in ? reg[4 * i + j] : 0.f;
                    hi[j] = (__bf16)a;
                    if (PLANES == 2) lo[j] = (__bf16)(a - (float)hi[j]);
                }
                *reinterpret_cast<bf16x4*>(dst) = hi;
                if (PLANES == 2) *reinterpret_cast<bf16x4*>(dst + kPlane) = lo;
            }
        }
    }
    // MFMA fragment: 8 consecutive k (k-step s, lane half h) of one row
    __device__ static __forceinline__ bf16x8 frag(const __bf16* plane, int row, int s, int h) {
        return *reinterpret_cast<const bf16x8*>(plane + row * BRS + 16 * s + 8 * h);
    }
};

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int PLANES>
__global__ __launch_bounds__(256) void bgemm_kernel(const GemmArgs p) {
    using TA = BTile<BM, AKM, PLANES>;
    using TB = BTile<BN, BKM, PLANES>;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) __bf16 smem[2 * (TA::kElems + TB::kElems)];
    auto a_buf = [&](int i) { return smem + i * TA::kElems; };
    auto b_buf = [&](int i) { return smem + 2 * TA::kElems + i * TB::kElems; };

    const TileCoord tc = locate_tile(p);
    const int logical = tc.logical;
    const int tile_m = logical / p.tiles_n, tile_n = logical % p.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int split = tc.split;
    const int64_t k_begin = tc.k_begin, k_end = tc.k_end;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* pa[TA::kVec > 0 ? TA::kVec : 1];
    const float* pb[TB::kVec > 0 ? TB::kVec : 1];
    const float* ka = nullptr;
    const float* kb = nullptr;
    TA::init_ptrs(pa, ka, p.A, p.lda, m0, p.M, k_begin);
    TB::init_ptrs(pb, kb, p.B, p.ldb, n0, p.N, k_begin);
    float ra[TA::kRegs], rb[TB::kRegs];
    const int nkt = (int)((k_end - k_begin + BBK - 1) / BBK);
    auto klim = [&](int kt) { return (int)min<int64_t>(BBK, k_end - k_begin - (int64_t)kt * BBK); };
    if (nkt > 0) {
        TA::fetch(ra, pa, ka, p.lda, 0, klim(0));
        TB::fetch(rb, pb, kb, p.ldb, 0, klim(0));
        TA::stash(ra, a_buf(0), klim(0));
        TB::stash(rb, b_buf(0), klim(0));
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
        if (more) {
            TA::fetch(ra, pa, ka, p.lda, kt + 1, klim(kt + 1));
            TB::fetch(rb, pb, kb, p.ldb, kt + 1, klim(kt + 1));
        }
        const __bf16* as = a_buf(cur);
        const __bf16* bs = b_buf(cur);
#pragma unroll
        for (int s = 0; s < BBK / 16; ++s) {
            bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = TA::frag(as, wm0 + 32 * i + l32, s, h);
                if (PLANES == 2) al[i] = TA::frag(as + TA::kPlane, wm0 + 32 * i + l32, s, h);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = TB::frag(bs, wn0 + 32 * j + l32, s, h);
                if (PLANES == 2) bl[j] = TB::frag(bs + TB::kPlane, wn0 + 32 * j + l32, s, h);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (PLANES == 2) {   // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (more) {
            TA::stash(ra, a_buf(cur ^ 1), klim(kt + 1));
            TB::stash(rb, b_buf(cur ^ 1), klim(kt + 1));
        }
        __syncthreads();
    }
    if (BN == 128 && tc.tail_slab >= 0)   // tails are only planned for 128 x 128 tiles
        finish_tail<TM, TN>(acc, p, tc, wave, 4, lane, m0, n0, wm0, wn0, reinterpret_cast<unsigned*>(smem));
    else gemm_epilogue<TM, TN>(acc, p, m0, n0, wm0, wn0, l32, h, split);
}

template <int BM, int BN, int WM, int WN, int PLANES>
static int launch_b(const GemmArgs& a, int opA, int opB, hipStream_t st) {
    const dim3 grid(gemm_grid(a)), block(256);
    if (opA == MSN_OP_N && opB == MSN_OP_T) hipLaunchKernelGGL((bgemm_kernel<BM, BN, WM, WN, false, false, PLANES>), grid, block, 0, st, a);
    else if (opA == MSN_OP_N && opB == MSN_OP_N) hipLaunchKernelGGL((bgemm_kernel<BM, BN, WM, WN, false, true, PLANES>), grid, block, 0, st, a);
    else if (opA == MSN_OP_T && opB == MSN_OP_N) hipLaunchKernelGGL((bgemm_kernel<BM, BN, WM, WN, true, true, PLANES>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bgemm_kernel<BM, BN, WM, WN, true, false, PLANES>), grid, block, 0, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

int launch_bgemm(const GemmArgs& a, int opA, int opB, int planes, int bm, int bn, hipStream_t st) {
    (void)bm;
    if (planes == 2) {
        if (bn == 128) return launch_b<128, 128, 64, 64, 2>(a, opA, opB, st);
        if (bn == 64) return launch_b<128, 64, 64, 32, 2>(a, opA, opB, st);
        return launch_b<128, 32, 32, 32, 2>(a, opA, opB, st);
    }
    if (bn == 128) return launch_b<128, 128, 64, 64, 1>(a, opA, opB, st);
    if (bn == 64) return launch_b<128, 64, 64, 32, 1>(a, opA, opB, st);
    return launch_b<128, 32, 32, 32, 1>(a, opA, opB, st);
}

}  // namespace msn
