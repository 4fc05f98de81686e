// Loader-wave fp32 GEMM (msn_set_gemm_variant(4)); a translation unit of its own so that it compiles beside gemm.hip.
#include <algorithm>
#include <type_traits>

#include "gemm_common.h"

namespace msn {

// ------------------------------------------------------------------------------------------------
// Loader-wave variant: persistent workgroups of SIX waves.  Waves 0-3 multiply exactly as above; waves 4 and 5 do
// nothing but issue the LDS-DMA loads of the ring (A pieces / B pieces), for the workgroup's whole sequence of
// (tile, K-step) items, running ahead across tile boundaries.  Why a wave of its own: epilogue stores and LDS-DMA loads of one wave share vmcnt and may
// complete out of order with respect to each other, so a wave that does both has to drain its stores before it can
// know that the next tile's first K-step has landed -- the prologue comes back as a store drain (round 1 measured a
// persistent kernel built that way equal to the one-tile launch).  Here the multiplying waves never wait on vmcnt for
// operands: their stores of tile j drain under the K-steps of tile j + 1, whose data the loader has already
// brought in.  One raw s_barrier per K-step (all five waves), placed as above: the multiplying waves arrive when
// their fragment reads of the step have returned (its slot is free), the loader when the NEXT step has landed.
// Work items (tiles, split-K slabs, tail slabs) are walked w = blockIdx.x, + gridDim.x, ... (a multiple of 8: the
// workgroup stays on its XCD's run of tiles).  Tail slabs are summed by the finishing launch.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int DBK, int STAGES, bool CSUM = false>
__global__ __launch_bounds__(64 * ((BM / WM) * (BN / WN) + 2), 3) void sgemm_pw_kernel(const GemmArgs p) {
    static_assert(!CSUM || AKM, "the fused column sums are those of a K-major A operand (wgrad: A = dY)");
    using TA = DmaTile<BM, AKM, DBK>;
    using TB = DmaTile<BN, BKM, DBK>;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN, NWAVES = (BM / WM) * (BN / WN);
    constexpr int STAGE = TA::kFloats + TB::kFloats;
    constexpr int PAT = TA::kPieces, PBT = TB::kPieces;   // LDS-DMA pieces per K-step: A by loader wave 0, B by loader wave 1
    static_assert((STAGES - 1) * PAT <= 63 && (STAGES - 1) * PBT <= 63, "vmcnt is 6 bits");
    constexpr int NKO = DBK / 8;
    __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE];
    const unsigned smem_addr = (unsigned)(uintptr_t)(lptr_t*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwork = work_items(p);
    if (wave >= NWAVES) {   // ------------------------------------------------------------------ the two loader waves
        __builtin_amdgcn_s_setprio(3);   // their few instructions go ahead of the multiplying waves on the same SIMD
        const bool is_b = wave > NWAVES;
        constexpr int PT = PAT > PBT ? PAT : PBT;
        const int np = is_b ? PBT : PAT;
        const int64_t step = is_b ? (BKM ? (int64_t)DBK * p.ldb : DBK) : (AKM ? (int64_t)DBK * p.lda : DBK);
        int total = 0;      // K-steps of this workgroup's items
        for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
            const TileCoord t = locate_tile(p, w);
            total += (int)((t.k_end - t.k_begin) / DBK);
        }
        const float* src[PT];
        int w_i = blockIdx.x, kt_i = 0, nkt_i = 0, issued = 0;
        auto setup = [&]() {
            const TileCoord t = locate_tile(p, w_i);
            const int64_t m0 = (int64_t)(t.logical / p.tiles_n) * BM, n0 = (int64_t)(t.logical % p.tiles_n) * BN;
            if (is_b) {
#pragma unroll
                for (int q = 0; q < PBT; ++q) src[q] = TB::src(p.B, p.ldb, n0, p.N, t.k_begin, q, lane);
            } else {
#pragma unroll
                for (int q = 0; q < PAT; ++q) src[q] = TA::src(p.A, p.lda, m0, p.M, t.k_begin, q, lane);
            }
            nkt_i = (int)((t.k_end - t.k_begin) / DBK);
            kt_i = 0;
        };
        auto issue_next = [&]() {   // the next K-step of the item sequence into ring slot issued % STAGES
            if (issued >= total) return;
            while (kt_i == nkt_i) {   // next item (one without K-steps is skipped by both sides)
                w_i += gridDim.x;
                setup();
            }
            float* base = smem + (issued % STAGES) * STAGE + (is_b ? TA::kFloats : 0);
#pragma unroll
            for (int q = 0; q < PT; ++q)
                if (q < np) {
                    __builtin_amdgcn_global_load_lds((gptr_t*)src[q], (lptr_t*)(base + q * 256), 16, 0, 0);
                    src[q] += step;
                }
            ++kt_i, ++issued;
        };
        if (total > 0) setup();
#pragma unroll
        for (int s = 0; s < STAGES; ++s) issue_next();
        // `n` younger K-steps may stay in flight (n <= STAGES - 1; immediate operands only: PAT == PBT is the common case)
        auto wait_younger = [&](int n) {
            if (PAT == PBT && STAGES > 2 && n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PAT <= 63 ? 2 * PAT : 0) : "memory");
            else if (PAT == PBT && n >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PAT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        wait_younger(issued - 1);                  // K-step 0 has landed
        __builtin_amdgcn_s_barrier();
        for (int g = 0; g < total; ++g) {
            if (g + 1 < total) wait_younger(issued - (g + 2));   // K-step g + 1 has landed
            __builtin_amdgcn_s_barrier();                          // ... and slot g % STAGES has been read by every wave
            issue_next();
        }
        return;
    }

    // ------------------------------------------------------------------------------ the multiplying waves
    const int h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    constexpr int NREADS = TM * TA::kReads + TN * TB::kReads;
    static_assert(NKO % 2 == 0, "k-octets are processed in pairs");
    typename TA::Frag fa[2][TM];
    typename TB::Frag fb[2][TN];
    f32x16 acc[TM][TN];
    float csum[TM];
    auto request = [&](auto set, int slot, int ko) {
        constexpr int S = decltype(set)::value;
        const unsigned as = smem_addr + 4u * (slot * STAGE);
        const unsigned bs = as + 4u * TA::kFloats;
#pragma unroll
        for (int i = 0; i < TM; ++i) TA::frag_issue(fa[S][i], as, wm0 + 32 * i + l32, ko, h);
#pragma unroll
        for (int j = 0; j < TN; ++j) TB::frag_issue(fb[S][j], bs, wn0 + 32 * j + l32, ko, h);
    };
    auto multiply = [&](auto set, bool younger_in_flight) {
        constexpr int S = decltype(set)::value;
        if (younger_in_flight) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NREADS) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CSUM) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                csum[i] += (TA::template get<0>(fa[S][i]) + TA::template get<1>(fa[S][i])) +
                           (TA::template get<2>(fa[S][i]) + TA::template get<3>(fa[S][i]));
        }
#define MSN_MFMA_SWEEP(C)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(TA::template get<C>(fa[S][i]), TB::template get<C>(fb[S][j]), \
                                                         acc[i][j], 0, 0, 0);
        MSN_MFMA_SWEEP(0) MSN_MFMA_SWEEP(1) MSN_MFMA_SWEEP(2) MSN_MFMA_SWEEP(3)
#undef MSN_MFMA_SWEEP
        __builtin_amdgcn_sched_barrier(0);
    };
    int slot = 0;
    __builtin_amdgcn_s_barrier();                  // K-step 0 of the first item has landed
    for (int w = blockIdx.x; w < nwork; w += gridDim.x) {
        const TileCoord tc = locate_tile(p, w);
        const int logical = tc.logical;
        const int64_t m0 = (int64_t)(logical / p.tiles_n) * BM, n0 = (int64_t)(logical % p.tiles_n) * BN;
        const int nkt = (int)((tc.k_end - tc.k_begin) / DBK);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            csum[i] = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        if (nkt > 0) request(S0{}, slot, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int next_slot = slot + 1 == STAGES ? 0 : slot + 1;
#pragma unroll
            for (int ko = 0; ko < NKO; ko += 2) {
                request(S1{}, slot, ko + 1);
                multiply(S0{}, true);
                if (ko + 2 < NKO) {
                    request(S0{}, slot, ko + 2);
                    multiply(S1{}, true);
                } else {
                    const bool has_next = kt + 1 < nkt;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all fragment reads of this K-step are in
                    __builtin_amdgcn_s_barrier();                         // next K-step landed (loader), this slot is free
                    if (has_next) request(S0{}, next_slot, 0);
                    multiply(S1{}, has_next);
                }
            }
            slot = next_slot;
        }
        if constexpr (CSUM) {
            if (wn0 == 0 && logical % p.tiles_n == 0) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float v = csum[i] + __shfl_xor(csum[i], 32, 64);
                    const int64_t row = m0 + wm0 + 32 * i + l32;
                    if (h == 0 && row < p.M) p.colsum[(int64_t)tc.split * p.M + row] = v;
                }
            }
        }
        if (BN == 128 && tc.tail_slab >= 0) dump_tail<TM, TN>(acc, p, tc.tail_slab, wave, NWAVES, lane);
        else gemm_epilogue<TM, TN>(acc, p, m0, n0, wm0, wn0, l32, h, tc.split);
    }
}

template <int BM, int BN, int WM, int WN, int DBK, int STAGES>
static int launch_pw(const GemmArgs& a, int opA, int opB, hipStream_t st) {
    const unsigned items = gemm_grid(a);
    const dim3 grid(items < 512u ? items : 512u), block(64 * ((BM / WM) * (BN / WN) + 2));   // two workgroups per CU
    if (opA == MSN_OP_N && opB == MSN_OP_T) hipLaunchKernelGGL((sgemm_pw_kernel<BM, BN, WM, WN, false, false, DBK, STAGES>), grid, block, 0, st, a);
    else if (opA == MSN_OP_N && opB == MSN_OP_N) hipLaunchKernelGGL((sgemm_pw_kernel<BM, BN, WM, WN, false, true, DBK, STAGES>), grid, block, 0, st, a);
    else if (opA == MSN_OP_T && opB == MSN_OP_N && a.colsum) hipLaunchKernelGGL((sgemm_pw_kernel<BM, BN, WM, WN, true, true, DBK, STAGES, true>), grid, block, 0, st, a);
    else if (opA == MSN_OP_T && opB == MSN_OP_N) hipLaunchKernelGGL((sgemm_pw_kernel<BM, BN, WM, WN, true, true, DBK, STAGES>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((sgemm_pw_kernel<BM, BN, WM, WN, true, false, DBK, STAGES>), grid, block, 0, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

int launch_pw_128(const GemmArgs& a, int opA, int opB, hipStream_t st) { return launch_pw<128, 128, 64, 64, 32, 2>(a, opA, opB, st); }

}  // namespace msn
