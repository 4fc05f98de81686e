// fp32 GEMM on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
//
// Tiling (per 256-thread workgroup = 4 waves): BM x BN output tile, BK = 32 deep K-steps, operand
// tiles double-buffered in LDS, next tile prefetched global->registers while the current one is
// multiplied.  One MFMA consumes a 32x2 A slab and a 2x32 B slab: lane l supplies A[l&31][l>>5]
// and B[l>>5][l&31].  K order inside a K-step is free (it is a sum), so the lane half h = l>>5
// owns k = 4h..4h+3 of every 8-deep "octet" and reads them with ONE 16-byte LDS read when the
// operand is stored K-contiguous ([row][k], rows padded by 4 floats -> conflict-free b128 reads);
// a K-major operand ([k][col]) is read with four 4-byte reads whose 32 lanes sweep one row.
//
// Workgroup -> tile map is XCD-aware: the 8 XCDs have private L2s and consecutive workgroup ids
// are dealt round-robin over them, so ids are remapped to give each XCD a contiguous run of
// tiles that walk N fastest (neighbours share the A row-panel in that XCD's L2).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "gemm_common.h"

namespace msn {

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (the default whenever it applies): operand tiles go HBM -> LDS directly with
// global_load_lds_dwordx4 (no staging registers, no VALU, no ds_write), through a STAGES-deep ring so the
// loads of K-step kt + STAGES - 1 are in flight while kt is multiplied -- the register-staged kernel above
// can only run one K-step ahead (a second register set costs the second wave per SIMD), and one K-step of
// MFMA time is shorter than an HBM round trip under load.  An LDS-DMA instruction writes 64 lanes x 16 B =
// 1 KB contiguously, so the images are unpadded; bank conflicts of the 16-byte fragment reads of a
// K-contiguous image ([row][32 floats], 128-B rows) are broken by an XOR swizzle applied on the per-lane
// SOURCE address and again on the read: chunk c of row r lives at position c ^ (r & 7).  A K-major image
// ([k][rows]) is read 4 bytes per lane along a row: conflict-free as it is.
// Waits are counted (each wave leaves the youngest stage in flight) and the barrier is a raw s_barrier: a
// __syncthreads() would drain the DMA queue.  Needs: 16-B aligned operands, ld % 4 == 0, extents % 4 == 0,
// and K-ranges that are whole K-steps (K % 32 == 0); anything else takes the register-staged kernel.
// CONV = 1: the A operand (K-contiguous rows) is an implicit column matrix (convolution forward / dgrad);
// CONV = 2: the B operand (K-major) is (convolution wgrad).  See ConvGather.
template <int BM, int BN, int WM, int WN, bool AKM, bool BKM, int DBK, int STAGES, bool CSUM = false, int CONV = 0>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN), (BM / WM) * (BN / WN) == 4 ? 2 : 1) void sgemm_dma_kernel(const GemmArgs p) {
    static_assert(!CSUM || AKM, "the fused column sums are those of a K-major A operand (wgrad: A = dY)");
    static_assert(CONV == 0 || (CONV == 1 && !AKM) || (CONV == 2 && BKM), "implicit operand: A rows (1) or K-major B (2)");
    using TA = DmaTile<BM, AKM, DBK>;
    using TB = DmaTile<BN, BKM, DBK>;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN, NWAVES = (BM / WM) * (BN / WN);
    constexpr int STAGE = TA::kFloats + TB::kFloats;
    constexpr int PA = TA::kPieces / NWAVES, PB = TB::kPieces / NWAVES;   // pieces per wave per K-step
    static_assert(TA::kPieces % NWAVES == 0 && TB::kPieces % NWAVES == 0 && PA > 0 && PB > 0,
                  "LDS-DMA pieces must divide over the waves");
    constexpr int INFLIGHT = (STAGES - 2) * (PA + PB);                    // youngest stages left in flight
    constexpr int NKO = DBK / 8;
    __shared__ __attribute__((aligned(16))) float smem[STAGES * STAGE];
    const unsigned smem_addr = (unsigned)(uintptr_t)(lptr_t*)smem;   // LDS byte address of the ring

#ifdef MSN_TIMELINE
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const TileCoord tc = locate_tile(p);
    const int logical = tc.logical;
    const int64_t m0 = (int64_t)(logical / p.tiles_n) * BM, n0 = (int64_t)(logical % p.tiles_n) * BN;
    const int split = tc.split;
    const int64_t k_begin = tc.k_begin, k_end = tc.k_end;
    const int nkt = (int)((k_end - k_begin) / DBK);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

    // this lane's source pointers for K-step 0 (they advance by a constant per K-step)
    const float* sa[PA];
    const float* sb[PB];
    if constexpr (CONV != 1) {
#pragma unroll
        for (int i = 0; i < PA; ++i) sa[i] = TA::src(p.A, p.lda, m0, p.M, k_begin, wave * PA + i, lane);
    }
    if constexpr (CONV != 2) {
#pragma unroll
        for (int i = 0; i < PB; ++i) sb[i] = TB::src(p.B, p.ldb, n0, p.N, k_begin, wave * PB + i, lane);
    }
    const int64_t a_step = AKM ? (int64_t)DBK * p.lda : DBK, b_step = BKM ? (int64_t)DBK * p.ldb : DBK;
    // ---- implicit operand (ConvGather) -------------------------------------------------------------------------
    // CONV 1: a lane serves the same PA rows (pixels) in every K-step; a K-step lies inside one tap (SC % DBK == 0), so
    // the tap and the channel offset of the next K-step to issue are wave-uniform running values.
    [[maybe_unused]] int gy[PA], gx[PA];           // source row / column of tap (0, 0) for this lane's rows
    [[maybe_unused]] int64_t gb[PA];               // batch offset + this lane's (swizzled) chunk inside the K-step
    [[maybe_unused]] int tky = 0, tkx = 0, tc0 = 0;
    if constexpr (CONV == 1) {
        const ConvGather& g = p.cg;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = (wave * PA + i) * (64 / TA::CPR) + lane / TA::CPR, pos = lane % TA::CPR;
            const int64_t m = m0 + r;
            const int mm = m < p.M ? (int)m : 0;
            const int b = mm / (g.RH * g.RW), rem = mm - b * (g.RH * g.RW), ry = rem / g.RW, rx = rem - ry * g.RW;
            gy[i] = m < p.M ? ry * g.sy + g.y0 : -(1 << 24);   // rows beyond M: never inside the image
            gx[i] = rx * g.sx + g.x0;
            gb[i] = (int64_t)b * g.SH * g.SW * g.SC + 4 * (pos ^ TA::swz(r));
        }
        const int tap = (int)(k_begin / g.SC);
        tc0 = (int)(k_begin - (int64_t)tap * g.SC);
        tky = tap / g.kw, tkx = tap - tky * g.kw;
    }
    // CONV 2: a lane serves the same 4 columns (one tap, 4 channels) in every K-step; the k index walks the pixels.
    [[maybe_unused]] int wky = 0, wkx = 0, wcc = 0;
    [[maybe_unused]] bool wn_ok = false;
    if constexpr (CONV == 2) {
        const ConvGather& g = p.cg;
        constexpr int LPR = BN / 4;
        const int64_t n = n0 + 4 * (lane % LPR);
        wn_ok = n < p.N;
        const int nn = wn_ok ? (int)n : 0, tap = nn / g.SC;
        wcc = nn - tap * g.SC;
        wky = tap / g.kw, wkx = tap - wky * g.kw;
    }
    auto issue = [&](int kt) {   // LDS-DMA of K-step kt into ring slot kt % STAGES
#ifdef MSN_ABL_NODMA
        return;                  // diagnostic build (tools/microbench/build_ablate.sh): no operand traffic at all
#endif
        float* base = smem + (kt % STAGES) * STAGE;
        if constexpr (CONV == 1) {
            const ConvGather& g = p.cg;
            const int dy = g.dir * tky, dx = g.dir * tkx;
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int sy = gy[i] + dy, sx = gx[i] + dx;
                const bool ok = (unsigned)sy < (unsigned)g.SH && (unsigned)sx < (unsigned)g.SW;
                const float* s = ok ? g.src + gb[i] + ((int64_t)sy * g.SW + sx) * g.SC + tc0 : g.zero;
                __builtin_amdgcn_global_load_lds((gptr_t*)s, (lptr_t*)(base + (wave * PA + i) * 256), 16, 0, 0);
            }
            tc0 += DBK;                                   // K-steps are issued in order
            if (tc0 == g.SC) {
                tc0 = 0;
                if (++tkx == g.kw) tkx = 0, ++tky;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(sa[i] + kt * a_step), (lptr_t*)(base + (wave * PA + i) * 256), 16, 0, 0);
        }
        if constexpr (CONV == 2) {
            const ConvGather& g = p.cg;
            constexpr int LPR = BN / 4;
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const unsigned m = (unsigned)(k_begin + (int64_t)kt * DBK) + (unsigned)((wave * PB + i) * (64 / LPR) + lane / LPR);
                const unsigned t = (unsigned)(((unsigned long long)m * g.rw_magic) >> 36), rx = m - t * (unsigned)g.RW;
                const unsigned b = (unsigned)(((unsigned long long)t * g.rh_magic) >> 36), ry = t - b * (unsigned)g.RH;
                const int sy = (int)ry * g.sy + g.y0 + wky, sx = (int)rx * g.sx + g.x0 + wkx;
                const bool ok = wn_ok && (unsigned)sy < (unsigned)g.SH && (unsigned)sx < (unsigned)g.SW;
                const float* s = ok ? g.src + (((int64_t)b * g.SH + sy) * g.SW + sx) * g.SC + wcc : g.zero;
                __builtin_amdgcn_global_load_lds((gptr_t*)s, (lptr_t*)(base + TA::kFloats + (wave * PB + i) * 256), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PB; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t*)(sb[i] + kt * b_step),
                                                 (lptr_t*)(base + TA::kFloats + (wave * PB + i) * 256), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#ifdef MSN_ACC_AGPR
    { float agpr_hint = 0.f; asm volatile("" : "+a"(agpr_hint)); }
#endif

    // Ring protocol: one barrier per K-step, placed before the MFMAs of the step's LAST k-octet.  At that point
    // every wave has received all its fragment reads of this step (lgkmcnt(0)), so after the barrier
    //   * the slot just read is free: the DMA of K-step kt + STAGES is issued into it (write-after-read safe);
    //   * the DMA pieces of K-step kt + 1 have landed for every wave (each waited for its own: vmcnt), so the first
    //     fragments of kt + 1 are requested right away and their LDS latency hides under the last octet's MFMAs
    //     -- no wave leaves a barrier with nothing to multiply.
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (s < nkt) issue(s);
    // K-step 0 must have landed (for every wave: each waits for its own pieces, then the barrier)
    if (nkt >= STAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * (PA + PB)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef MSN_TIMELINE
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
#endif

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    constexpr int NREADS = TM * TA::kReads + TN * TB::kReads;
    static_assert(NKO % 2 == 0, "k-octets are processed in pairs");
    // fragments of the next k-octet are requested before the MFMAs of the current one; the counted wait retires
    // exactly the reads of the current octet (LDS returns in order) and leaves the younger ones in flight
    typename TA::Frag fa[2][TM];
    typename TB::Frag fb[2][TN];
    // CSUM (wgrad with bias gradient): the A fragments this wave multiplies are dY[k][row]; their running sum over
    // k is the bias gradient of `row` -- a few v_add per 16 MFMAs instead of a separate pass over dY
    float csum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) csum[i] = 0.f;
#ifdef MSN_ABL_NOFRAG
    for (int s = 0; s < 2; ++s) {
        for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(fa[s][i]));
        for (int j = 0; j < TN; ++j) asm volatile("" : "=v"(fb[s][j]));
    }
#endif
    auto request = [&](auto set, int slot, int ko) {
        constexpr int S = decltype(set)::value;
#ifdef MSN_ABL_NOFRAG
        return;                  // diagnostic build: no fragment reads (the MFMAs multiply whatever the registers hold)
#endif
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
        // consecutive MFMAs go to different accumulators (no back-to-back dependent issue)
#define MSN_MFMA_SWEEP(C)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(TA::template get<C>(fa[S][i]), TB::template get<C>(fb[S][j]), \
                                                         acc[i][j], 0, 0, 0);
        MSN_MFMA_SWEEP(0) MSN_MFMA_SWEEP(1) MSN_MFMA_SWEEP(2) MSN_MFMA_SWEEP(3)
#undef MSN_MFMA_SWEEP
        __builtin_amdgcn_sched_barrier(0);
    };
#ifdef MSN_TIMELINE
    unsigned long long w_lds = 0, w_vm = 0, w_bar = 0;
    const unsigned long long tl0 = __builtin_readcyclecounter();
#endif
    int slot = 0;
    if (nkt > 0) request(S0{}, 0, 0);
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
#if MSN_TIMELINE >= 2
                const unsigned long long ta = __builtin_readcyclecounter();
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all fragment reads of this K-step are in
#if MSN_TIMELINE >= 2
                const unsigned long long tb = __builtin_readcyclecounter();
#endif
                if (kt + STAGES - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if MSN_TIMELINE >= 2
                const unsigned long long tcc = __builtin_readcyclecounter();
#endif
#ifndef MSN_ABL_NOBAR
                __builtin_amdgcn_s_barrier();
#endif
#if MSN_TIMELINE >= 2
                const unsigned long long td = __builtin_readcyclecounter();
                w_lds += tb - ta, w_vm += tcc - tb, w_bar += td - tcc;
#endif
                if (kt + STAGES < nkt) issue(kt + STAGES);           // into `slot`
                if (has_next) request(S0{}, next_slot, 0);
                multiply(S1{}, has_next);
            }
        }
        slot = next_slot;
    }
#ifdef MSN_TIMELINE
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl2 = __builtin_readcyclecounter();
#endif
    if constexpr (CSUM) {
        if (wn0 == 0 && logical % p.tiles_n == 0) {   // one wave column of the first tile column owns each row
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float v = csum[i] + __shfl_xor(csum[i], 32, 64);   // lane halves hold k = 4h .. 4h + 3
                const int64_t row = m0 + wm0 + 32 * i + l32;
                if (h == 0 && row < p.M) p.colsum[(int64_t)split * p.M + row] = v;
            }
        }
    }
    if (BN == 128 && tc.tail_slab >= 0)   // tails are only planned for 128 x 128 tiles
        finish_tail<TM, TN>(acc, p, tc, wave, NWAVES, lane, m0, n0, wm0, wn0, reinterpret_cast<unsigned*>(smem));
    else gemm_epilogue<TM, TN>(acc, p, m0, n0, wm0, wn0, l32, h, split);
#ifdef MSN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
    if (p.dbg && threadIdx.x == 0) {
        unsigned long long* d = p.dbg + (size_t)blockIdx.x * 10;
        d[0] = t0, d[1] = t1, d[2] = t2, d[3] = t3;
        d[4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));      // HW_ID
        d[5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));     // XCC_ID
        d[6] = w_lds, d[7] = w_vm, d[8] = w_bar, d[9] = tl2 - tl0;   // shader-clock cycles inside the K loop
    }
#endif
}

// Sum of the K-slabs of one tail tile (slab order: deterministic) + the epilogue; same wave / register layout as
// the kernel that wrote the slabs.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * (BM / WM) * (BN / WN)) void tail_finish_kernel(const GemmArgs p) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN, NWAVES = (BM / WM) * (BN / WN);
    const int logical = p.tiles_m * p.tiles_n - p.tail_tiles + (int)blockIdx.x;
    const int64_t m0 = (int64_t)(logical / p.tiles_n) * BM, n0 = (int64_t)(logical % p.tiles_n) * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.tail_splits; ++s) {
        const float* base = p.tail_partial +
                            (((int64_t)blockIdx.x * p.tail_splits + s) * NWAVES + wave) * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += base[((i * TN + j) * 16 + r) * 64];
    }
    gemm_epilogue<TM, TN>(acc, p, m0, n0, wm0, wn0, l32, h, 0);
}

int launch_tail_finish(const GemmArgs& a, int bn, int waves, hipStream_t st) {
    const dim3 grid(a.tail_tiles);
    if (bn == 128 && waves == 8) hipLaunchKernelGGL((tail_finish_kernel<128, 128, 64, 32>), grid, dim3(512), 0, st, a);
    else if (bn == 128) hipLaunchKernelGGL((tail_finish_kernel<128, 128, 64, 64>), grid, dim3(256), 0, st, a);
    else return MSN_OK;   // tails are only planned for 128 x 128 tiles
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// C[m][n] = sum_s partial[s][m][n]   (split-K finish; fixed order -> deterministic).  A block covers 64 x V
// consecutive outputs (V = 4: one 16-byte load per slab and lane when N and ldc allow, else V = 1) with G thread
// groups that each sum every G-th slab (four independent partial sums in flight), combined through LDS.  G = 16 for
// many slabs of a small output (a 32 x 32 weight gradient cut into 512 K-slabs would otherwise be 4 workgroups
// walking 128 dependent loads each), G = 4 otherwise.
// The last `cs_blocks` workgroups of the grid (fused bias gradient, msn_wgrad_bias) sum the [splits][M] column-sum
// slabs instead: 64 rows per workgroup, the G groups sharing the slabs the same way.
template <int V, int G>
__global__ __launch_bounds__(64 * G) void splitk_reduce_kernel(const float* __restrict__ partial, float* __restrict__ C,
                                                               int64_t M, int64_t N, int64_t ldc, int splits,
                                                               const float* __restrict__ cs_partial,
                                                               float* __restrict__ cs_out, int cs_blocks) {
    __shared__ float red[G][64 * V];
    const int main_blocks = (int)gridDim.x - cs_blocks;
    if ((int)blockIdx.x >= main_blocks) {   // 64 rows of the column-sum slabs, the G groups sharing the slabs
        const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int64_t row = (int64_t)((int)blockIdx.x - main_blocks) * 64 + cl;
        float s0 = 0.f, s1 = 0.f;
        if (row < M) {
            int k = g;
            for (; k + G < splits; k += 2 * G) {
                s0 += cs_partial[(int64_t)k * M + row];
                s1 += cs_partial[(int64_t)(k + G) * M + row];
            }
            if (k < splits) s0 += cs_partial[(int64_t)k * M + row];
        }
        red[g][cl] = s0 + s1;
        __syncthreads();
        if (g == 0 && row < M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < G; ++q) t += red[q][cl];
            cs_out[row] = t;
        }
        return;
    }
    const int64_t total = M * N;
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64 * V; base < total; base += (int64_t)main_blocks * 64 * V) {
        const int64_t i = base + (int64_t)cl * V;
        float s[4][V];
#pragma unroll
        for (int v = 0; v < V; ++v) s[0][v] = s[1][v] = s[2][v] = s[3][v] = 0.f;
        if (i < total) {
            auto add = [&](float (&acc)[V], int k) {
                if (V == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(partial + (int64_t)k * total + i);
                    acc[0] += t.x, acc[1 % V] += t.y, acc[2 % V] += t.z, acc[3 % V] += t.w;
                } else {
                    acc[0] += partial[(int64_t)k * total + i];
                }
            };
            int k = g;
            for (; k + 3 * G < splits; k += 4 * G) {   // four independent loads in flight
                add(s[0], k);
                add(s[1], k + G);
                add(s[2], k + 2 * G);
                add(s[3], k + 3 * G);
            }
            for (; k < splits; k += G) add(s[0], k);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) red[g][cl * V + v] = (s[0][v] + s[1][v]) + (s[2][v] + s[3][v]);
        __syncthreads();
        if (g == 0 && i < total) {
            float t[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (G == 4) {
                    t[v] = (red[0][cl * V + v] + red[1][cl * V + v]) + (red[2 % G][cl * V + v] + red[3 % G][cl * V + v]);
                } else {
                    t[v] = 0.f;
#pragma unroll
                    for (int q = 0; q < G; ++q) t[v] += red[q][cl * V + v];
                }
            }
            float* out = C + (i / N) * ldc + (i % N);
            if (V == 4) *reinterpret_cast<float4*>(out) = make_float4(t[0], t[1 % V], t[2 % V], t[3 % V]);
            else out[0] = t[0];
        }
        __syncthreads();
    }
}

// Column sums, two deterministic stages.  Stage 1: a 256-thread block = 4 row groups x 64 columns
// streams its share of the rows (coalesced 256-B row segments, 4 rows in flight per thread) and
// reduces the row groups through LDS -> part[block][N].  Stage 2: the same shape sums the slabs.
constexpr int CS_COLS = 64, CS_RG = 4, CS_MAX_BLOCKS = 1024;   // 128 blocks left half of the CUs without a workgroup
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int64_t ldx, int64_t M,
                                                             int64_t N, float* __restrict__ part) {
    __shared__ float red[CS_RG][CS_COLS];
    const int cl = threadIdx.x % CS_COLS, rg = threadIdx.x / CS_COLS;
    const int64_t n = (int64_t)blockIdx.y * CS_COLS + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        const int64_t stride = (int64_t)gridDim.x * CS_RG;
        int64_t r = (int64_t)blockIdx.x * CS_RG + rg;
        for (; r + 3 * stride < M; r += 4 * stride) {
            s0 += X[r * ldx + n];
            s1 += X[(r + stride) * ldx + n];
            s2 += X[(r + 2 * stride) * ldx + n];
            s3 += X[(r + 3 * stride) * ldx + n];
        }
        for (; r < M; r += stride) s0 += X[r * ldx + n];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && n < N) part[(int64_t)blockIdx.x * N + n] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int64_t slabs, int64_t N,
                                                           float* __restrict__ out) {
    __shared__ float red[CS_RG][CS_COLS];
    const int cl = threadIdx.x % CS_COLS, rg = threadIdx.x / CS_COLS;
    const int64_t n = (int64_t)blockIdx.x * CS_COLS + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n < N) {
        int64_t k = rg;
        for (; k + 3 * CS_RG < slabs; k += 4 * CS_RG) {   // four slabs in flight per thread
            s0 += part[k * N + n];
            s1 += part[(k + CS_RG) * N + n];
            s2 += part[(k + 2 * CS_RG) * N + n];
            s3 += part[(k + 3 * CS_RG) * N + n];
        }
        for (; k < slabs; k += CS_RG) s0 += part[k * N + n];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && n < N) out[n] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
static int colsum_blocks(int64_t M) { return (int)std::min<int64_t>(cdiv(M, 16 * CS_RG), CS_MAX_BLOCKS); }

// (msn_set_gemm_lds_pad -- unused dynamic LDS that kept a second workgroup of the 128 x 128 kernel off the CU for the sake of the side
// streams -- went in round 6: the image tower slows by 9 % for 0.8 ms more of the light-curve tower hidden, DESIGN section 4 "Streams".)
template <typename K>
static void launch_padded(K kernel, dim3 grid, dim3 block, size_t pad, hipStream_t st, const GemmArgs& a) {
    hipLaunchKernelGGL(kernel, grid, block, pad, st, a);
}
template <int BM, int BN, int WM, int WN, int DBK, int STAGES>
static int launch_dma(const GemmArgs& a, int opA, int opB, hipStream_t st) {
    const dim3 grid(gemm_grid(a)), block(64 * (BM / WM) * (BN / WN));
    const size_t pad = 0;
    if (opA == MSN_OP_N && opB == MSN_OP_T) launch_padded(sgemm_dma_kernel<BM, BN, WM, WN, false, false, DBK, STAGES>, grid, block, pad, st, a);
    else if (opA == MSN_OP_N && opB == MSN_OP_N) launch_padded(sgemm_dma_kernel<BM, BN, WM, WN, false, true, DBK, STAGES>, grid, block, pad, st, a);
    else if (opA == MSN_OP_T && opB == MSN_OP_N && a.colsum) launch_padded(sgemm_dma_kernel<BM, BN, WM, WN, true, true, DBK, STAGES, true>, grid, block, pad, st, a);
    else if (opA == MSN_OP_T && opB == MSN_OP_N) launch_padded(sgemm_dma_kernel<BM, BN, WM, WN, true, true, DBK, STAGES>, grid, block, pad, st, a);
    else launch_padded(sgemm_dma_kernel<BM, BN, WM, WN, true, false, DBK, STAGES>, grid, block, pad, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// gemm_pw.hip: the loader-wave kernel (msn_set_gemm_variant(4)), 128 x 128 tiles
int launch_pw_128(const GemmArgs& a, int opA, int opB, hipStream_t st);

// conv = 1: A implicit (forward: B = weights [N][K], opB = T; dgrad: B = weights [K][N], opB = N);
// conv = 2: B implicit, A = dY K-major (wgrad, with or without the fused bias gradient)
template <int BM, int BN, int WM, int WN, int DBK, int STAGES>
static int launch_dma_conv(const GemmArgs& a, int conv, int opB, hipStream_t st) {
    const dim3 grid(gemm_grid(a)), block(64 * (BM / WM) * (BN / WN));
    if (conv == 1 && opB == MSN_OP_T) hipLaunchKernelGGL((sgemm_dma_kernel<BM, BN, WM, WN, false, false, DBK, STAGES, false, 1>), grid, block, 0, st, a);
    else if (conv == 1) hipLaunchKernelGGL((sgemm_dma_kernel<BM, BN, WM, WN, false, true, DBK, STAGES, false, 1>), grid, block, 0, st, a);
    else if (a.colsum) hipLaunchKernelGGL((sgemm_dma_kernel<BM, BN, WM, WN, true, true, DBK, STAGES, true, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((sgemm_dma_kernel<BM, BN, WM, WN, true, true, DBK, STAGES, false, 2>), grid, block, 0, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

static int g_gemm_variant = 3;   // 0 = register-staged kernels; LDS-DMA: 1 = 8 waves, ring 3 x BK 32; 2 = 8 waves, ring 2 x BK 64;
                                 // 3 (default) = 4 waves, ring 2 x BK 32, two workgroups per CU

// gemm_reg.hip: the register-staged kernels (every shape; msn_set_gemm_variant(0)) for the three tile widths
int launch_cfg_tile(const GemmArgs& a, int bn, int opA, int opB, hipStream_t st);

static int g_gemm_tail = 1;      // cut the partly filled last round of tiles into K-slabs (msn_set_gemm_tail_split):
                                 // 1 = summed by the last workgroup to arrive, 2 = by a finishing launch, 0 = no slabs

// Arrival counters of the in-kernel tail finish.  Launches on ONE stream run one after the other and every launch
// leaves its counters at zero, so a stream needs one slice of counters; streams that overlap (the towers of a
// step run on their own streams) get a slice each.  Module memory: nothing is allocated, the first use looks the
// address up.  More streams than slices: those launches fall back to the finishing launch.
// A __device__ symbol has one address PER GPU and stream handles are per GPU too: the tables below are kept per device
// (hipGetDevice() of the calling thread, the device the launch goes to), so a process that drives several GPUs never
// hands a kernel another GPU's counters or zero page.
constexpr int kTailSlices = 32, kTailSliceTiles = 512;     // a tail has fewer than 512 tiles (tiles % 512)
constexpr int kMaxDevices = 64;
__device__ unsigned g_tail_counters[kTailSlices * kTailSliceTiles];
static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}
static unsigned* tail_counter_slice(hipStream_t st) {
    struct PerDevice {
        hipStream_t owner[kTailSlices];
        int used = 0;
        unsigned* base = nullptr;
    };
    static std::mutex mu;
    static PerDevice table[kMaxDevices];
    const int dev = current_device();
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    PerDevice& d = table[dev];
    if (!d.base) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_tail_counters)) != hipSuccess) return nullptr;
        d.base = static_cast<unsigned*>(sym);
    }
    for (int i = 0; i < d.used; ++i)
        if (d.owner[i] == st) return d.base + (size_t)i * kTailSliceTiles;
    if (d.used == kTailSlices) return nullptr;
    d.owner[d.used] = st;
    return d.base + (size_t)(d.used++) * kTailSliceTiles;
}
unsigned* gemm_counter_slice(hipStream_t st) { return tail_counter_slice(st); }
static int g_gemm_streamk = 1024, g_gemm_streamk_min_k = 1024;   // msn_sgemm: opA = N products of at most this many 128 x 128 tiles and at
                                                                 // least this K take the work-list kernel (0 tiles = never); set by
                                                                 // msn_set_gemm_list (gemm_set_single_rule)
static int g_gemm_bn = 0;        // measurement switch (msn_set_gemm_tile_n): 0 = planned, 64 / 128 = forced tile width for N > 64

// Launch geometry of one product.
//  * tile shape by N;
//  * split-K (opA = T, i.e. wgrad: small M x N, huge K): the chip holds 512 workgroups at a time (256 CUs x 2) and
//    equal-length workgroups run in rounds, so the split count is the one whose tiles x splits comes closest under a
//    whole number of rounds (36 tiles: 14 splits = 504 workgroups in one round, where 21 splits = 756 would leave
//    half the chip idle in the second);
//  * tail: without split-K, the tiles % 512 tiles of the last, partly filled round are cut into K-slabs
//    (1560 tiles = 3 rounds + 24 tiles: those 24 would occupy 24 CUs for a whole fourth round).
struct Plan {
    int bm, bn, splits, kps, tail_tiles, tail_splits, tail_kps;
    size_t ws_bytes(int64_t M, int64_t N) const {
        if (splits > 1) return sizeof(float) * (size_t)splits * (size_t)M * (size_t)N;
        return sizeof(float) * (size_t)tail_tiles * (size_t)tail_splits * (size_t)bm * (size_t)bn;
    }
};

static Plan plan(int64_t M, int64_t N, int64_t K, int opA, bool allow_tail = true, bool small_tail = false) {
    Plan p;
    p.bm = 128;
    p.bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    // Under-filled launches (strong scaling: 128 - 512 rows per GPU): up to ~1.5 rounds of 128 x 128 tiles, twice as many
    // 128 x 64 tiles fill the 512 workgroup slots better than K-slabs of the wide tiles + a finishing pass
    // (tools/bench_gemm_small.py, M = 8320: proj 42.3 -> 33.1 us, qkv 83.2 -> 77.8, ff1 110.4 -> 102.1; M = 33280 proj
    // 104.2 -> 97.4; from ~1000 tiles on the wide tile wins again)
    // (small_tail: the implicit convolutions, which the work-list kernel does not take -- a launch of fewer than 128 wide tiles with a long
    // K keeps them and cuts EVERY tile into K-slabs instead, see the tail split below: ResNet-18's last stages at 256 rows per GPU are
    // 32 / 64 tiles of 144 / 72 K-steps)
    const bool cut_all = small_tail && opA == MSN_OP_N && g_gemm_bn == 0 && g_gemm_tail && allow_tail && N >= 128 &&
                         cdiv(M, 128) * cdiv(N, 128) < 128 && cdiv(K, BK) >= 32;
    if (N > 64 && N % 64 == 0 && opA == MSN_OP_N && g_gemm_bn == 0 && cdiv(M, 128) * cdiv(N, 128) <= 800 && !cut_all) p.bn = 64;
    // A width whose last 128-wide tile would be at most half full (N = 96, 192: the stacked q|k|v projections of the emb-32 /
    // emb-64 towers) wastes a quarter and more of the MFMAs on columns that do not exist: 64-wide tiles (measured on
    // 200k-row products: N = 192 80.5 -> 69.8 us, N = 96 46.2 -> 38.8 us)
    if (N > 64 && N % 128 != 0 && N % 128 <= 64 && opA == MSN_OP_N && g_gemm_bn == 0) p.bn = 64;
    // Weight gradients with a small output (dW of the e x e projection: 384 x 384 = 9 tiles of 128 x 128) are all split-K: twice as
    // many 128 x 64 tiles need half the K-slabs for the same 512 workgroups -- half the slab bytes, K loops twice as long per prologue
    // (tools/microbench/wgrad_tiles.py: 384 x 384 over 16 640 rows 66.1 -> 53.1 us, 8320: 38.7 -> 35.8, 66 560: 182 -> 175;
    // 1152 x 384 gains up to 16 640 rows (94 -> 80, 146 -> 138 us) and loses from 66 560 on; 36 tiles and more: no gain)
    if (opA == MSN_OP_T && M > 64 && N > 64 && N % 64 == 0 && g_gemm_bn == 0) {          // (few rows: the 32- / 64-row tiles below)
        const int64_t t128 = cdiv(M, 128) * cdiv(N, 128);
        if (t128 <= 9 || (t128 <= 27 && K <= 20000)) p.bn = 64;
    }
    if (N > 64 && g_gemm_bn != 0) p.bn = g_gemm_bn;
    // Weight gradients with few rows (dW of a Linear whose OUTPUT is 32 / 64 wide: the reference towers' ff2): a 128-row
    // tile would multiply 4x / 2x rows that do not exist -- measured compute-bound on them (M = 32, N = 128, K = 225 280:
    // 82 us against 18 us of HBM time)
    if (opA == MSN_OP_T && p.bn == 128 && M <= 64 && g_gemm_bn == 0) p.bm = M <= 32 ? 32 : 64;
    // ... and with a narrow OUTPUT as well (dW of the emb-32 / emb-64 unifyheads: 32 x 32, 64 x 64): 64-row tiles of 64 / 32 columns
    if (opA == MSN_OP_T && p.bn <= 64 && M <= 64 && g_gemm_bn == 0) p.bm = 64;
    // ... or whose last 128-row tile would be at most half full (dW of the emb-64 q|k|v projection: 192 x 64 = three 64 x 64 tiles)
    if (opA == MSN_OP_T && p.bn <= 64 && M > 64 && M % 128 != 0 && M % 128 <= 64 && g_gemm_bn == 0) p.bm = 64;
    // Forward / dgrad products with one or two K-steps (K <= 64: ff1 forward and ff2 dgrad of the emb-32 / emb-64 towers): a
    // workgroup's life is prologue + epilogue more than K loop (per-launch accounting: 25-30 % of the CU time with nobody in
    // a K loop), so 64-row tiles -- 48 KB of LDS, three workgroups per CU -- overlap more of it (N = 256, K = 64 over 204 800
    // rows: forward 83.7 -> 78.2 us, dgrad + ReLU' 122.2 -> 109.8; 32-row tiles lose again)
    if (opA == MSN_OP_N && K <= 64 && p.bn >= 64 && g_gemm_bn == 0) p.bm = 64;
    p.tail_tiles = 0, p.tail_splits = 1, p.tail_kps = 0;
    const int64_t tiles = cdiv(M, p.bm) * cdiv(N, p.bn);
    const int64_t ksteps = cdiv(K, BK);
    int s = 1;
    if (opA == MSN_OP_T && tiles < 512) {
        const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(ksteps / 4, 512), 1024 / tiles));
        double best = 0.0;
        for (int64_t c = 1; c <= smax; ++c) {
            const int64_t wg = tiles * c, rounds = cdiv(wg, 512);
            const double fill = (double)wg / (double)(rounds * 512);
            if (fill > best + 1e-9) best = fill, s = (int)c;
        }
    }
    int64_t per = cdiv(cdiv(K, s), BK) * BK;
    if (per < BK) per = BK;
    s = (int)cdiv(K, per);
    if (s < 1) s = 1;
    p.splits = s;
    p.kps = (int)per;
    const int64_t r = tiles % 512;
    if (g_gemm_tail && allow_tail && s == 1 && p.bn == 128 && p.bm == 128 && r > 0 && (tiles >= 128 || cut_all) && tiles < (1ll << 30)) {
        // time of the last round in units of one whole tile: ceil(r c / 512) rounds of 1 / c tile each (+ a small
        // charge per slab for its prologue, the slab store and the finishing pass)
        const int64_t cmax = std::min<int64_t>(std::min<int64_t>(ksteps / 4, 16), 1024 / r);
        double best = 1.0;
        int c_best = 1;
        for (int64_t c = 2; c <= cmax; ++c) {
            const double t = (double)cdiv(r * c, 512) / (double)c + 0.02 * (double)c;
            if (t < best - 1e-9) best = t, c_best = (int)c;
        }
        if (c_best > 1 && best <= 0.85) {
            const int64_t tper = cdiv(cdiv(K, c_best), 2 * BK) * 2 * BK;   // whole K-steps of every kernel family
            p.tail_tiles = (int)r;
            p.tail_kps = (int)tper;
            p.tail_splits = (int)cdiv(K, tper);
            if (p.tail_splits < 2) p.tail_tiles = 0, p.tail_splits = 1;
        }
    }
    return p;
}

int launch_tail_finish(const GemmArgs& a, int bn, int waves, hipStream_t st);

}  // namespace msn

using namespace msn;

extern "C" size_t msn_colsum_workspace_bytes(int64_t M, int64_t N);
extern "C" int msn_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes,
                          msn_stream_t stream);

#ifdef MSN_TIMELINE
static unsigned long long* g_timeline = nullptr;
extern "C" int msn_debug_timeline(unsigned long long* buf) {
    g_timeline = buf;
    return MSN_OK;
}
#endif

// msn_sgemm's own use of the work-list kernel: an under-filled forward / dgrad product (shape test only; operand
// alignment is checked at the call)
static bool streamk_shape(int opA, int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = cdiv(M, 128) * cdiv(N, 128);
    // under-filled = the last round of 512 workgroup slots is at most ~90 % full (a whole number of rounds is what the flat
    // launch does best: 4096^3 = 1024 tiles runs at 134 TFLOP/s flat)
    return g_gemm_streamk > 0 && opA == MSN_OP_N && M > 64 && N > 64 && K % BK == 0 && K >= g_gemm_streamk_min_k && g_gemm_bn == 0 &&
           tiles <= g_gemm_streamk && tiles % 512 != 0 && tiles % 512 <= 460 && tiles * (K / BK) >= 64;
}

extern "C" size_t msn_sgemm_workspace_bytes(int opA, int opB, int64_t M, int64_t N, int64_t K) {
    (void)opB;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const size_t own = plan(M, N, K, opA).ws_bytes(M, N);
    return streamk_shape(opA, M, N, K) ? std::max(own, gemm_list_ws_bytes()) : own;
}

// The launch behind msn_sgemm and msn_wgrad_bias.  `colsum_out` (opA = T only): also produce out[m] = sum_k A[k][m]
// inside the product kernel when the LDS-DMA kernels take the shape; *colsum_done tells the caller whether it was.
static int sgemm_impl(int opA, int opB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* bias, int epilogue, float* aux, int64_t ldaux,
                      int precision, void* ws, size_t ws_bytes, msn_stream_t stream, float* colsum_out,
                      bool* colsum_done, const ConvGather* conv_gather = nullptr, int conv = 0) {
    MSN_REQUIRE(M >= 0 && N >= 0 && K >= 0, "msn_sgemm: negative size M=%lld N=%lld K=%lld", (long long)M,
                (long long)N, (long long)K);
    if (M == 0 || N == 0) return MSN_OK;
    MSN_REQUIRE(K > 0, "msn_sgemm: K must be positive");
    MSN_REQUIRE(A && B && C, "msn_sgemm: null operand");
    MSN_REQUIRE((opA == MSN_OP_N || opA == MSN_OP_T) && (opB == MSN_OP_N || opB == MSN_OP_T),
                "msn_sgemm: bad op flags %d %d", opA, opB);
    MSN_REQUIRE(lda >= (opA == MSN_OP_N ? K : M), "msn_sgemm: lda %lld too small", (long long)lda);
    MSN_REQUIRE(ldb >= (opB == MSN_OP_N ? N : K), "msn_sgemm: ldb %lld too small", (long long)ldb);
    MSN_REQUIRE(ldc >= N, "msn_sgemm: ldc %lld < N", (long long)ldc);
    MSN_REQUIRE(epilogue >= MSN_EPI_NONE && epilogue <= MSN_EPI_ADD, "msn_sgemm: bad epilogue %d", epilogue);
    MSN_REQUIRE(precision >= MSN_PREC_F32 && precision <= MSN_PREC_BF16, "msn_sgemm: bad precision %d", precision);
    const bool needs_aux = epilogue == MSN_EPI_RELU_BWD || epilogue == MSN_EPI_GELU_BWD || epilogue == MSN_EPI_ADD;
    MSN_REQUIRE(!needs_aux || (aux && ldaux >= N), "msn_sgemm: epilogue %d needs aux with ldaux >= N", epilogue);
    MSN_REQUIRE(!(epilogue == MSN_EPI_GELU && aux) || ldaux >= N, "msn_sgemm: ldaux < N");

    if (conv == 0 && colsum_out == nullptr && precision == MSN_PREC_F32 && g_gemm_variant == 3 && streamk_shape(opA, M, N, K)) {
        const msn_gemm_desc d{opA, opB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, aux, ldaux, nullptr};
        // (a stream beyond the 32 counter slices of its device keeps the flat launch below: the work-list kernel needs one)
        if (gemm_list_takes(1, &d) && ws && ws_bytes >= gemm_list_ws_bytes() && gemm_counter_slice(static_cast<hipStream_t>(stream)))
            return gemm_list_launch(1, &d, ws, ws_bytes, static_cast<hipStream_t>(stream));
    }
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.aux = aux;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux;
    a.epilogue = epilogue;
    const Plan pl = plan(M, N, K, opA, colsum_out == nullptr, conv != 0);
    // (short tiles: M <= bm, one row of tiles either way; of the implicit convolutions only the weight gradient of a 64-channel layer
    // -- 64 x (taps x C) over all output pixels: ResNet-18's first stage -- has a 64-row kernel, the others keep 128 rows)
    int bm = conv != 0 ? ((conv == 2 && pl.bm == 64 && pl.bn == 128) ? 64 : 128) : pl.bm;
    const int bn = pl.bn, splits = pl.splits, kps = pl.kps;
    {   // the 32- / 64-row tiles exist among the LDS-DMA kernels only: a product those do not take (K % 32 != 0, unaligned
        // operands, msn_set_gemm_variant(0)) runs on the 128-row register-staged kernels -- with the 64-row plan it would launch
        // twice the workgroups, every second one multiplying rows that do not exist (round-3 advisor finding)
        const int64_t a_ext0 = opA == MSN_OP_T ? M : K, b_ext0 = opB == MSN_OP_N ? N : K;
        const bool dma_pre = (g_gemm_variant != 0 || conv != 0) && K % BK == 0 && (lda % 4 == 0) && (ldb % 4 == 0) && (a_ext0 % 4 == 0) &&
                             (b_ext0 % 4 == 0) && a_ext0 >= 4 && b_ext0 >= 4 &&
                             ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
        if (!dma_pre && conv == 0 && bm != 128) bm = 128;
    }
    a.colsum = nullptr;
    a.tiles_m = (int)cdiv(M, bm);
    a.tiles_n = (int)cdiv(N, bn);
    a.splits = splits;
    a.k_per_split = kps;
    a.partial = nullptr;
    a.tail_tiles = pl.tail_tiles, a.tail_splits = pl.tail_splits, a.tail_kps = pl.tail_kps;
    a.tail_partial = nullptr;
    a.tail_counter = nullptr;
    a.cg = conv_gather ? *conv_gather : ConvGather{};
#ifdef MSN_TIMELINE
    a.dbg = g_timeline;
#endif
    if (a.tail_tiles > 0) {
        const size_t need = pl.ws_bytes(M, N);
        MSN_REQUIRE(ws && ws_bytes >= need, "msn_sgemm: workspace %zu < %zu bytes", ws_bytes, need);
        a.tail_partial = static_cast<float*>(ws);
        if (g_gemm_tail == 1 && a.tail_tiles <= kTailSliceTiles && !(g_gemm_variant == 4 && conv == 0))
            a.tail_counter = tail_counter_slice(static_cast<hipStream_t>(stream));
    }
    if (splits > 1) {
        MSN_REQUIRE(epilogue == MSN_EPI_NONE && bias == nullptr,
                    "msn_sgemm: split-K (opA=T) supports only the plain epilogue without bias");
        const size_t need = sizeof(float) * (size_t)splits * (size_t)M * (size_t)N;
        MSN_REQUIRE(ws && ws_bytes >= need, "msn_sgemm: workspace %zu < %zu bytes", ws_bytes, need);
        a.partial = static_cast<float*>(ws);
    }
    MSN_REQUIRE((int64_t)a.tiles_m * a.tiles_n * a.splits + (int64_t)a.tail_tiles * a.tail_splits < (1ll << 31),
                "msn_sgemm: too many tiles");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    // bf16 matrix-core paths need 16-byte loads on the K-contiguous operands; otherwise stay on fp32
    bool bf16_ok = precision != MSN_PREC_F32;
    if (bf16_ok && opA == MSN_OP_N)
        bf16_ok = (lda % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    if (bf16_ok && opB == MSN_OP_T)
        bf16_ok = (ldb % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    // LDS-DMA kernels: every operand 16-B aligned with ld % 4 == 0 and extents % 4 == 0, whole K-steps only
    const int64_t a_ext = opA == MSN_OP_T ? M : K, b_ext = opB == MSN_OP_N ? N : K;
    const bool dma_ok = (g_gemm_variant != 0 || conv != 0) && bn >= 32 && K % BK == 0 && kps % BK == 0 && (lda % 4 == 0) && (ldb % 4 == 0) &&
                        (a_ext % 4 == 0) && (b_ext % 4 == 0) && a_ext >= 4 && b_ext >= 4 &&
                        ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
    const bool fuse_colsum = colsum_out && !bf16_ok && dma_ok && opA == MSN_OP_T && opB == MSN_OP_N;
    if (fuse_colsum) {   // slabs of partial sums behind the product's own split-K slabs, or the result itself
        if (splits > 1) {
            const size_t need = pl.ws_bytes(M, N) + sizeof(float) * (size_t)splits * (size_t)M;
            MSN_REQUIRE(ws && ws_bytes >= need, "msn_wgrad_bias: workspace %zu < %zu bytes", ws_bytes, need);
            a.colsum = a.partial + (size_t)splits * (size_t)M * (size_t)N;
        } else {
            a.colsum = colsum_out;
        }
    }
    if (colsum_done) *colsum_done = fuse_colsum;
    int waves = 4;   // per workgroup of the kernel chosen (the tail pass mirrors its register layout)
    if (bm != 128 && M <= 64) bf16_ok = false;   // the short tiles of products with few rows exist for the fp32 kernels only (tiny
                                                 // products: exact fp32 costs nothing); a tall product keeps the precision it was asked for
    if (bf16_ok && bm != 128) bm = 128, a.tiles_m = (int)cdiv(M, bm);
    if (conv != 0) {   // implicit-GEMM convolution: the 4-wave LDS-DMA kernels only
        MSN_REQUIRE(dma_ok && !bf16_ok && bn >= 64, "implicit convolution: shape not taken by the LDS-DMA kernels");
        if (bm == 64) rc = launch_dma_conv<64, 128, 32, 64, 32, 2>(a, conv, opB, st);
        else if (bn == 128) rc = launch_dma_conv<128, 128, 64, 64, 32, 2>(a, conv, opB, st);
        else rc = launch_dma_conv<128, 64, 64, 32, 32, 3>(a, conv, opB, st);
    } else if (bf16_ok) rc = launch_bgemm(a, opA, opB, precision == MSN_PREC_BF16X3 ? 2 : 1, bm, bn, st);
    else if (bm == 64 && bn == 64 && dma_ok) rc = launch_dma<64, 64, 32, 32, 32, 3>(a, opA, opB, st);
    else if (bm == 64 && bn == 32 && dma_ok) rc = launch_dma<64, 32, 32, 32, 32, 3>(a, opA, opB, st);
    else if (bm == 32 && dma_ok) rc = launch_dma<32, 128, 32, 32, 32, 2>(a, opA, opB, st);
    else if (bm == 64 && dma_ok) rc = launch_dma<64, 128, 32, 64, 32, 2>(a, opA, opB, st);   // (not dma_ok: the 128-row kernels below; M <= 64 is one row of tiles either way)
    else if (dma_ok && bn == 128 && K % 64 == 0 && kps % 64 == 0 && g_gemm_variant == 2)
        rc = launch_dma<128, 128, 64, 32, 64, 2>(a, opA, opB, st), waves = 8;
    else if (dma_ok && bn == 128 && g_gemm_variant == 4) rc = launch_pw_128(a, opA, opB, st);
    else if (dma_ok && bn == 128 && g_gemm_variant == 3) rc = launch_dma<128, 128, 64, 64, 32, 2>(a, opA, opB, st);
    else if (dma_ok && bn == 128 && g_gemm_variant < 3) rc = launch_dma<128, 128, 64, 32, 32, 3>(a, opA, opB, st), waves = 8;
    else if (dma_ok && bn == 64) rc = launch_dma<128, 64, 64, 32, 32, 3>(a, opA, opB, st);
    else if (dma_ok && bn == 32) rc = launch_dma<128, 32, 32, 32, 32, 2>(a, opA, opB, st);
    else rc = launch_cfg_tile(a, bn, opA, opB, st);
    if (rc != MSN_OK) return rc;
    if (a.tail_tiles > 0 && a.tail_counter == nullptr) {
        rc = launch_tail_finish(a, bn, waves, st);
        if (rc != MSN_OK) return rc;
    }
    if (splits > 1) {
        const int64_t total = M * N;
        // 16-byte path: a group of 4 outputs never straddles a row (N % 4 == 0) and C rows stay 16-byte aligned
        const bool v4 = N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
        const int cs_blocks = fuse_colsum ? (int)cdiv(M, 64) : 0;   // the column-sum slabs ride in the same launch
        const bool deep = splits >= 64;   // many slabs: 16 thread groups share them
        const int blocks = (int)std::min<int64_t>(cdiv(total, v4 ? 256 : 64), 4096);
        const dim3 grid(blocks + cs_blocks);
        if (v4 && deep)
            hipLaunchKernelGGL((splitk_reduce_kernel<4, 16>), grid, dim3(1024), 0, st, a.partial, C, M, N, ldc, splits,
                               a.colsum, colsum_out, cs_blocks);
        else if (v4)
            hipLaunchKernelGGL((splitk_reduce_kernel<4, 4>), grid, dim3(256), 0, st, a.partial, C, M, N, ldc, splits,
                               a.colsum, colsum_out, cs_blocks);
        else if (deep)
            hipLaunchKernelGGL((splitk_reduce_kernel<1, 16>), grid, dim3(1024), 0, st, a.partial, C, M, N, ldc, splits,
                               a.colsum, colsum_out, cs_blocks);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<1, 4>), grid, dim3(256), 0, st, a.partial, C, M, N, ldc, splits,
                               a.colsum, colsum_out, cs_blocks);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}

extern "C" int msn_sgemm(int opA, int opB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                         const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int epilogue,
                         float* aux, int64_t ldaux, int precision, void* ws, size_t ws_bytes, msn_stream_t stream) {
    return sgemm_impl(opA, opB, M, N, K, A, lda, B, ldb, C, ldc, bias, epilogue, aux, ldaux, precision, ws, ws_bytes,
                      stream, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on channels-last tensors (ConvGather): no column matrix in memory.
__device__ float g_conv_zero_page[64];
static const float* conv_zero_page() {      // per device, like the tail counters
    static const float* page[kMaxDevices] = {};
    static std::mutex mu;
    const int dev = current_device();
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!page[dev]) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_conv_zero_page)) != hipSuccess) return nullptr;
        page[dev] = static_cast<const float*>(sym);
    }
    return page[dev];
}
// n / d == (n * div_magic(d)) >> 36 for every n with n * d < 2^36 (the kernels split pixel indices this way)
static unsigned long long div_magic(int d) { return (1ull << 36) / (unsigned long long)d + 1ull; }
// image sizes the gather arithmetic covers: fewer than 2^27 pixels on either side of the convolution and every index x divisor pair
// inside the multiply-shift division's exact range (512 x 512 images at any batch up to 512; a 1 x 1024 series at batch 1024 ...)
static bool conv_size_ok(int B, int H, int W, int OH, int OW) {
    const int64_t n = std::max((int64_t)B * H * W, (int64_t)B * OH * OW);
    const int64_t d = std::max(std::max(H, W), std::max(OH, OW));
    return H <= 8192 && W <= 8192 && n < (1ll << 27) && n * d < (1ll << 36);
}
struct ConvShape { int B, H, W, C, Cout, kh, kw, sh, sw, ph, pw, OH, OW; };
static int conv_shape(const char* who, int B, int H, int W, int C, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                      ConvShape* cs) {
    MSN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && Cout > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
                "%s: bad geometry", who);
    const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
    MSN_REQUIRE(OH > 0 && OW > 0, "%s: kernel larger than the padded input", who);
    MSN_REQUIRE(conv_size_ok(B, H, W, OH, OW),
                "%s: implicit convolution takes fewer than 2^27 pixels per launch with pixels x longest side < 2^36", who);
    *cs = ConvShape{B, H, W, C, Cout, kh, kw, sh, sw, ph, pw, OH, OW};
    return MSN_OK;
}
// which convolutions the implicit kernels take (the caller falls back to im2col + msn_sgemm otherwise)
extern "C" int msn_conv2d_implicit_ok(int B, int H, int W, int C, int Cout, int kh, int kw, int sh, int sw, int ph, int pw) {
    if (B <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || ph < 0 || pw < 0) return 0;
    const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
    if (OH <= 0 || OW <= 0 || !conv_size_ok(B, H, W, OH, OW)) return 0;
    // a K-step of 32 lies inside one tap (C % 32 == 0; dgrad: C_out % 32 == 0); tiles at least 64 wide (C_out, C > 32);
    // wgrad walks whole K-steps of output pixels
    return C % 32 == 0 && Cout % 32 == 0 && C > 32 && Cout > 32 && ((int64_t)B * OH * OW) % 32 == 0;
}
extern "C" size_t msn_conv2d_workspace_bytes(int B, int H, int W, int C, int Cout, int kh, int kw, int sh, int sw, int ph,
                                             int pw) {
    if (!msn_conv2d_implicit_ok(B, H, W, C, Cout, kh, kw, sh, sw, ph, pw)) return 0;
    const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
    const int64_t Mo = (int64_t)B * OH * OW, Mi = (int64_t)B * H * W, Kc = (int64_t)kh * kw * C;
    const Plan f = plan(Mo, Cout, Kc, MSN_OP_N, true, true), d = plan(Mi, C, (int64_t)kh * kw * Cout, MSN_OP_N, true, true), w = plan(Cout, Kc, Mo, MSN_OP_T, false);
    const size_t wg = w.ws_bytes(Cout, Kc) + sizeof(float) * (size_t)w.splits * (size_t)Cout;
    return std::max(std::max(f.ws_bytes(Mo, Cout), d.ws_bytes(Mi, C)), wg);
}
extern "C" int msn_conv2d_fwd(const float* x, int B, int H, int W, int C, const float* w_tap, int Cout, int kh, int kw,
                              int sh, int sw, int ph, int pw, const float* bias, int epilogue, float* y, void* ws,
                              size_t ws_bytes, msn_stream_t stream) {
    ConvShape c;
    if (int rc = conv_shape("msn_conv2d_fwd", B, H, W, C, Cout, kh, kw, sh, sw, ph, pw, &c)) return rc;
    MSN_REQUIRE(x && w_tap && y, "msn_conv2d_fwd: null pointer");
    MSN_REQUIRE(msn_conv2d_implicit_ok(B, H, W, C, Cout, kh, kw, sh, sw, ph, pw), "msn_conv2d_fwd: shape not taken (msn_conv2d_implicit_ok)");
    MSN_REQUIRE(epilogue == MSN_EPI_NONE || epilogue == MSN_EPI_RELU, "msn_conv2d_fwd: epilogue none or ReLU");
    const float* zero = conv_zero_page();
    MSN_REQUIRE(zero, "msn_conv2d_fwd: no zero page");
    ConvGather g{x, zero, H, W, C, c.OH, c.OW, sh, sw, -ph, -pw, 1, kw, div_magic(c.OW), div_magic(c.OH)};
    const int64_t M = (int64_t)B * c.OH * c.OW, K = (int64_t)kh * kw * C;
    return sgemm_impl(MSN_OP_N, MSN_OP_T, M, Cout, K, x, K, w_tap, K, y, Cout, bias, epilogue, nullptr, 0, MSN_PREC_F32, ws,
                      ws_bytes, stream, nullptr, nullptr, &g, 1);
}
// dX of a stride-1 convolution: rows = input pixels, taps mirrored, weights as [(tap, c_out)][c_in]
// (msn_conv_weight_relayout, to_tap = 2)
extern "C" int msn_conv2d_dgrad(const float* dy, int B, int H, int W, int C, const float* w_tco, int Cout, int kh, int kw,
                                int ph, int pw, float* dx, void* ws, size_t ws_bytes, msn_stream_t stream) {
    ConvShape c;
    if (int rc = conv_shape("msn_conv2d_dgrad", B, H, W, C, Cout, kh, kw, 1, 1, ph, pw, &c)) return rc;
    MSN_REQUIRE(dy && w_tco && dx, "msn_conv2d_dgrad: null pointer");
    MSN_REQUIRE(msn_conv2d_implicit_ok(B, H, W, C, Cout, kh, kw, 1, 1, ph, pw), "msn_conv2d_dgrad: shape not taken (msn_conv2d_implicit_ok)");
    const float* zero = conv_zero_page();
    MSN_REQUIRE(zero, "msn_conv2d_dgrad: no zero page");
    ConvGather g{dy, zero, c.OH, c.OW, Cout, H, W, 1, 1, ph, pw, -1, kw, div_magic(W), div_magic(H)};
    const int64_t M = (int64_t)B * H * W, K = (int64_t)kh * kw * Cout;
    return sgemm_impl(MSN_OP_N, MSN_OP_N, M, C, K, dy, K, w_tco, C, dx, C, nullptr, MSN_EPI_NONE, nullptr, 0, MSN_PREC_F32, ws,
                      ws_bytes, stream, nullptr, nullptr, &g, 1);
}
// dW[c_out][(tap, c_in)] = sum over output pixels of dY[pixel][c_out] * x[pixel shifted by the tap][c_in]; dbias (or null)
// = column sums of dY, inside the same launch
extern "C" int msn_conv2d_wgrad(const float* dy, const float* x, int B, int H, int W, int C, int Cout, int kh, int kw, int sh,
                                int sw, int ph, int pw, float* dw_tap, float* dbias, void* ws, size_t ws_bytes,
                                msn_stream_t stream) {
    ConvShape c;
    if (int rc = conv_shape("msn_conv2d_wgrad", B, H, W, C, Cout, kh, kw, sh, sw, ph, pw, &c)) return rc;
    MSN_REQUIRE(dy && x && dw_tap, "msn_conv2d_wgrad: null pointer");
    MSN_REQUIRE(msn_conv2d_implicit_ok(B, H, W, C, Cout, kh, kw, sh, sw, ph, pw), "msn_conv2d_wgrad: shape not taken (msn_conv2d_implicit_ok)");
    const float* zero = conv_zero_page();
    MSN_REQUIRE(zero, "msn_conv2d_wgrad: no zero page");
    ConvGather g{x, zero, H, W, C, c.OH, c.OW, sh, sw, -ph, -pw, 1, kw, div_magic(c.OW), div_magic(c.OH)};
    const int64_t K = (int64_t)B * c.OH * c.OW, N = (int64_t)kh * kw * C;
    bool done = false;
    const int rc = sgemm_impl(MSN_OP_T, MSN_OP_N, Cout, N, K, dy, Cout, x, N, dw_tap, N, nullptr, MSN_EPI_NONE, nullptr, 0,
                              MSN_PREC_F32, ws, ws_bytes, stream, dbias, &done, &g, 2);
    if (rc != MSN_OK) return rc;
    MSN_REQUIRE(dbias == nullptr || done, "msn_conv2d_wgrad: bias gradient not produced");
    return MSN_OK;
}

extern "C" size_t msn_wgrad_bias_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const Plan pl = plan(M, N, K, MSN_OP_T, false);
    const size_t fused = pl.ws_bytes(M, N) + sizeof(float) * (size_t)pl.splits * (size_t)M;
    const size_t separate = std::max(pl.ws_bytes(M, N), msn_colsum_workspace_bytes(K, M));
    return std::max(fused, separate);
}

extern "C" int msn_wgrad_bias(int64_t M, int64_t N, int64_t K, const float* dY, int64_t lddy, const float* X, int64_t ldx,
                              float* dW, int64_t lddw, float* db, int precision, void* ws, size_t ws_bytes,
                              msn_stream_t stream) {
    MSN_REQUIRE(db, "msn_wgrad_bias: null db");
    bool fused = false;
    const int rc = sgemm_impl(MSN_OP_T, MSN_OP_N, M, N, K, dY, lddy, X, ldx, dW, lddw, nullptr, MSN_EPI_NONE, nullptr, 0,
                              precision, ws, ws_bytes, stream, db, &fused);
    if (rc != MSN_OK || fused || M == 0 || N == 0) return rc;
    return msn_colsum(dY, lddy, K, M, db, ws, ws_bytes, stream);   // shapes / precisions the fused kernel does not take
}

namespace msn {
// msn_set_gemm_list's rule for SINGLE products: 0 = msn_sgemm never takes the work-list kernel, 1 = the planner's rule (products of at
// most 1024 tiles with K >= 1024 whose last round of workgroups is under-filled), 2 = every product the kernel can take (tests)
void gemm_set_single_rule(int rule) {
    g_gemm_streamk = rule == 0 ? 0 : (rule == 2 ? (1 << 20) : 1024);
    g_gemm_streamk_min_k = rule == 2 ? 32 : 1024;
}
}  // namespace msn

extern "C" int msn_reset_gemm_counters(msn_stream_t stream) {
    // only the slice this stream owns: the other slices belong to streams whose tail-split / work-list kernels may be in
    // flight (the towers of a step run concurrently), and zeroing their counters under them would leave a cut tile with no
    // finisher or with two.  A stream without a slice has nothing to reset.
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return MSN_OK;   // never record it into a graph
    unsigned* slice = tail_counter_slice(st);
    if (!slice) return MSN_OK;
    if (hipMemsetAsync(slice, 0, sizeof(unsigned) * kTailSliceTiles, st) != hipSuccess) {
        set_error("msn_reset_gemm_counters: %s", hipGetErrorString(hipGetLastError()));
        return MSN_ERR_HIP;
    }
    return MSN_OK;
}

extern "C" int msn_set_gemm_tile_n(int bn) {
    MSN_REQUIRE(bn == 0 || bn == 64 || bn == 128, "msn_set_gemm_tile_n: 0 (planned), 64 or 128");
    g_gemm_bn = bn;
    return MSN_OK;
}

extern "C" int msn_set_gemm_tail_split(int enabled) {
    MSN_REQUIRE(enabled >= 0 && enabled <= 2, "msn_set_gemm_tail_split: 0 (off), 1 (in-kernel finish) or 2 (finishing launch)");
    g_gemm_tail = enabled;
    return MSN_OK;
}

extern "C" int msn_set_gemm_variant(int mode) {
    MSN_REQUIRE(mode >= 0 && mode <= 4, "msn_set_gemm_variant: mode must be 0 (register-staged), 1..3 (LDS-DMA rings) or 4 (loader wave)");
    g_gemm_variant = mode;
    return MSN_OK;
}

extern "C" size_t msn_colsum_workspace_bytes(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 0;
    return sizeof(float) * (size_t)colsum_blocks(M) * (size_t)N;
}

extern "C" int msn_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, void* ws,
                          size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(M > 0 && N > 0 && X && out && ldx >= N, "msn_colsum: bad arguments");
    const int slabs = colsum_blocks(M);
    MSN_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)slabs * (size_t)N, "msn_colsum: workspace too small");
    MSN_REQUIRE(cdiv(N, CS_COLS) <= 65535, "msn_colsum: too many columns");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const unsigned by = (unsigned)cdiv(N, CS_COLS);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(slabs, by), dim3(256), 0, st, X, ldx, M, N, part);
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final_kernel, dim3(by), dim3(256), 0, st, part, (int64_t)slabs, N, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
