// Register-staged fp32 GEMM kernels (msn_set_gemm_variant(0), and every shape the LDS-DMA kernels do not take); a
// translation unit of its own so that it compiles beside gemm.hip.  See gemm.hip for the tiling.
#include <algorithm>
#include <type_traits>

#include "gemm_common.h"

namespace msn {

// Load 4 consecutive elements (r, c..c+3) of a stored row-major matrix from an address CLAMPED into
// the matrix; `ok` bit j says whether element j is really inside [nr, nc).  Branch-free and with the
// zero-masking deferred to the LDS write (OperandTile::stash), so that a K-step's loads issue back to
// back, stay in flight across the MFMA block of the previous K-step, and are waited for only when
// they are written to LDS.  (A guarded `if (in range) load`, or even a wave-uniform runtime
// `if (aligned)`, makes hipcc wait vmcnt(0) after every load; masking right after the load makes it
// wait before the MFMAs.)  VEC is chosen on the host: base 16-B aligned, ld % 4 == 0 and the
// contiguous extent % 4 == 0, so a vector never straddles nc; otherwise the scalar instantiation runs.
template <bool VEC>
__device__ __forceinline__ float4 load4_clamped(const float* __restrict__ base, int64_t r, int64_t c, int64_t ld,
                                                int64_t nr, int64_t nc, unsigned& ok) {
    const bool in_r = r < nr;
    const float* row = base + (in_r ? r : nr - 1) * ld;
    float4 v;
    if (VEC) {
        const bool in = in_r && c < nc;
        v = *reinterpret_cast<const float4*>(row + (c < nc ? c : 0));
        ok = in ? 0xFu : 0u;
    } else {
        const int64_t last = nc - 1;
        v.x = row[c < nc ? c : last];
        v.y = row[c + 1 < nc ? c + 1 : last];
        v.z = row[c + 2 < nc ? c + 2 : last];
        v.w = row[c + 3 < nc ? c + 3 : last];
        ok = in_r ? ((c < nc ? 1u : 0u) | (c + 1 < nc ? 2u : 0u) | (c + 2 < nc ? 4u : 0u) | (c + 3 < nc ? 8u : 0u)) : 0u;
    }
    return v;
}

template <int ROWS, bool KMAJOR>
struct OperandTile {
    // K-contiguous: [ROWS][BK + KPAD];  K-major: [BK][ROWS + KPAD]
    static constexpr int kStride = KMAJOR ? (ROWS + KPAD) : (BK + KPAD);
    static constexpr int kFloats = KMAJOR ? BK * (ROWS + KPAD) : ROWS * (BK + KPAD);
    static constexpr int kLoads = ROWS / 32;  // float4 per thread per K-step (256 threads)

    // row0: first row (M or N index) of this tile; k0: first k.  `nrows` = M or N, `nk` = K limit.
    // ok: 4 validity bits per load, consumed by stash().
    template <bool VEC>
    __device__ static __forceinline__ void fetch(float4 (&reg)[kLoads], unsigned& ok, const float* __restrict__ g,
                                                 int64_t ld, int64_t row0, int64_t nrows, int64_t k0, int64_t nk) {
        const int t = threadIdx.x;
        ok = 0u;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int idx = t + 256 * i;
            unsigned m;
            if (KMAJOR) {  // stored [K][rows]: 4 consecutive rows of one k
                const int k = idx / (ROWS / 4), q = idx % (ROWS / 4);
                reg[i] = load4_clamped<VEC>(g, k0 + k, row0 + 4 * q, ld, nk, nrows, m);
            } else {  // stored [rows][K]: 4 consecutive k of one row
                const int r = idx / (BK / 4), q = idx % (BK / 4);
                reg[i] = load4_clamped<VEC>(g, row0 + r, k0 + 4 * q, ld, nrows, nk, m);
            }
            ok |= m << (4 * i);
        }
    }
    __device__ static __forceinline__ void stash(const float4 (&reg)[kLoads], unsigned ok, float* lds) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int idx = t + 256 * i;
            const unsigned m = ok >> (4 * i);
            float4 v = reg[i];
            v.x = (m & 1u) ? v.x : 0.f;
            v.y = (m & 2u) ? v.y : 0.f;
            v.z = (m & 4u) ? v.z : 0.f;
            v.w = (m & 8u) ? v.w : 0.f;
            if (KMAJOR) {
                const int k = idx / (ROWS / 4), q = idx % (ROWS / 4);
                *reinterpret_cast<float4*>(lds + k * kStride + 4 * q) = v;
            } else {
                const int r = idx / (BK / 4), q = idx % (BK / 4);
                *reinterpret_cast<float4*>(lds + r * kStride + 4 * q) = v;
            }
        }
    }
    // ---- fast path (VEC kernels, K-steps that lie completely inside [k_begin, k_end)) ----------------
    // Per-thread source pointers are computed ONCE: rows / columns beyond the matrix are clamped onto
    // valid ones (whatever they load only feeds output elements that are never stored), so a full
    // K-step needs no masks and no address arithmetic beyond one add.
    __device__ static __forceinline__ void init_ptrs(const float* (&ptr)[kLoads], const float* __restrict__ g,
                                                     int64_t ld, int64_t row0, int64_t nrows, int64_t k_begin) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int idx = t + 256 * i;
            if (KMAJOR) {
                const int k = idx / (ROWS / 4), q = idx % (ROWS / 4);
                int64_t c = row0 + 4 * q;
                c = c + 3 < nrows ? c : nrows - 4;
                ptr[i] = g + (k_begin + k) * ld + c;
            } else {
                const int r = idx / (BK / 4), q = idx % (BK / 4);
                int64_t rr = row0 + r;
                rr = rr < nrows ? rr : nrows - 1;
                ptr[i] = g + rr * ld + k_begin + 4 * q;
            }
        }
    }
    __device__ static __forceinline__ void fetch_fast(float4 (&reg)[kLoads], const float* const (&ptr)[kLoads],
                                                      int64_t step_offset) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) reg[i] = *reinterpret_cast<const float4*>(ptr[i] + step_offset);
    }
    __device__ static __forceinline__ void stash_fast(const float4 (&reg)[kLoads], float* lds) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) {
            const int idx = t + 256 * i;
            if (KMAJOR) {
                const int k = idx / (ROWS / 4), q = idx % (ROWS / 4);
                *reinterpret_cast<float4*>(lds + k * kStride + 4 * q) = reg[i];
            } else {
                const int r = idx / (BK / 4), q = idx % (BK / 4);
                *reinterpret_cast<float4*>(lds + r * kStride + 4 * q) = reg[i];
            }
        }
    }
    // Fragment for one 32-row MFMA slab and one k-octet `ko`: f.{x,y,z,w} = element k = 8ko+4h+{0..3}.
    __device__ static __forceinline__ float4 frag(const float* lds, int row_in_tile, int ko, int h) {
        if (KMAJOR) {
            const float* p = lds + (8 * ko + 4 * h) * kStride + row_in_tile;
            return make_float4(p[0], p[kStride], p[2 * kStride], p[3 * kStride]);
        } else {
            return *reinterpret_cast<const float4*>(lds + row_in_tile * kStride + 8 * ko + 4 * h);
        }
    }
};

template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, bool VEC>
__global__ __launch_bounds__(256) void sgemm_kernel(const GemmArgs p) {
    using TA = OperandTile<BM, AKM>;
    using TB = OperandTile<BN, BKM>;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float smem[2 * (TA::kFloats + TB::kFloats)];
    auto a_buf = [&](int i) { return smem + i * TA::kFloats; };
    auto b_buf = [&](int i) { return smem + 2 * TA::kFloats + i * TB::kFloats; };

    const TileCoord tc = locate_tile(p);
    const int logical = tc.logical;
    const int tile_m = logical / p.tiles_n, tile_n = logical % p.tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;
    const int split = tc.split;
    const int64_t k_begin = tc.k_begin, k_end = tc.k_end;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[TA::kLoads], rb[TB::kLoads];
    unsigned oka = 0u, okb = 0u;
    const int nkt = (int)((k_end - k_begin + BK - 1) / BK);
    const int nfull = VEC ? (int)((k_end - k_begin) / BK) : 0;   // K-steps that need no bounds handling
    const float* pa[TA::kLoads];
    const float* pb[TB::kLoads];
    if (VEC) {
        // VEC guarantees M (or K) % 4 == 0 etc., but a K-major operand narrower than one vector cannot be clamped
        TA::init_ptrs(pa, p.A, p.lda, m0, p.M, k_begin);
        TB::init_ptrs(pb, p.B, p.ldb, n0, p.N, k_begin);
    }
    const int64_t a_step = AKM ? (int64_t)BK * p.lda : BK, b_step = BKM ? (int64_t)BK * p.ldb : BK;
    auto fetch_step = [&](int kt) {   // global -> registers for K-step kt
        if (kt < nfull) {
            TA::fetch_fast(ra, pa, kt * a_step);
            TB::fetch_fast(rb, pb, kt * b_step);
        } else {
            const int64_t k0 = k_begin + (int64_t)kt * BK;
            TA::template fetch<VEC>(ra, oka, p.A, p.lda, m0, p.M, k0, k_end);
            TB::template fetch<VEC>(rb, okb, p.B, p.ldb, n0, p.N, k0, k_end);
        }
    };
    auto stash_step = [&](int kt, int buf) {   // registers -> LDS buffer `buf`
        if (kt < nfull) {
            TA::stash_fast(ra, a_buf(buf));
            TB::stash_fast(rb, b_buf(buf));
        } else {
            TA::stash(ra, oka, a_buf(buf));
            TB::stash(rb, okb, b_buf(buf));
        }
    };
    if (nkt > 0) {
        fetch_step(0);
        stash_step(0, 0);
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nkt;
#ifndef MSN_ABL_NOFETCH
        if (more) fetch_step(kt + 1);   // prefetch the next K-step into registers while this one is multiplied
#endif
        const float* as = a_buf(cur);
        const float* bs = b_buf(cur);
#ifdef MSN_ABL_NOFRAG
        float4 fa[TM], fb[TN];
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = TA::frag(as, wm0 + 32 * i + l32, 0, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = TB::frag(bs, wn0 + 32 * j + l32, 0, h);
        }
#endif
#pragma unroll
        for (int ko = 0; ko < BK / 8; ++ko) {
#ifndef MSN_ABL_NOFRAG
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = TA::frag(as, wm0 + 32 * i + l32, ko, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = TB::frag(bs, wn0 + 32 * j + l32, ko, h);
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
#ifndef MSN_ABL_NOSTASH
        if (more) stash_step(kt + 1, cur ^ 1);
#endif
#ifndef MSN_ABL_NOBAR
        __syncthreads();
#endif
    }

    if (BN == 128 && tc.tail_slab >= 0)   // tails are only planned for 128 x 128 tiles
        finish_tail<TM, TN>(acc, p, tc, wave, 4, lane, m0, n0, wm0, wn0, reinterpret_cast<unsigned*>(smem));
    else gemm_epilogue<TM, TN>(acc, p, m0, n0, wm0, wn0, l32, h, split);
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const GemmArgs& a, int opA, int opB, hipStream_t st) {
    const dim3 grid(gemm_grid(a)), block(256);
    // 16-byte operand loads need: base aligned, ld % 4 == 0, contiguous extent % 4 == 0 (K for a
    // K-contiguous operand, M / N for a K-major one)
    const int64_t a_ext = opA == MSN_OP_T ? a.M : a.K, b_ext = opB == MSN_OP_N ? a.N : a.K;
    const bool vec = (a.lda % 4 == 0) && (a_ext % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0) &&
                     (a.ldb % 4 == 0) && (b_ext % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0) &&
                     a_ext >= 4 && b_ext >= 4;
#define MSN_GEMM_GO(AKM, BKM)                                                                               \
    {                                                                                                       \
        if (vec) hipLaunchKernelGGL((sgemm_kernel<BM, BN, WM, WN, AKM, BKM, true>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((sgemm_kernel<BM, BN, WM, WN, AKM, BKM, false>), grid, block, 0, st, a);    \
    }
    if (opA == MSN_OP_N && opB == MSN_OP_T) MSN_GEMM_GO(false, false)
    else if (opA == MSN_OP_N && opB == MSN_OP_N) MSN_GEMM_GO(false, true)
    else if (opA == MSN_OP_T && opB == MSN_OP_N) MSN_GEMM_GO(true, true)
    else MSN_GEMM_GO(true, false)
#undef MSN_GEMM_GO
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

int launch_cfg_tile(const GemmArgs& a, int bn, int opA, int opB, hipStream_t st) {
    if (bn == 128) return launch_cfg<128, 128, 64, 64>(a, opA, opB, st);
    if (bn == 64) return launch_cfg<128, 64, 64, 32>(a, opA, opB, st);
    return launch_cfg<128, 32, 32, 32>(a, opA, opB, st);
}

}  // namespace msn
