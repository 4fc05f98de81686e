// Work-list fp32 GEMM: ONE persistent launch for a short list of products, every workgroup an equal share of the K-steps.
//
// Why (profiles/r03_gemm_launch_accounting.txt): at 128 - 512 rows per GPU a product has fewer tiles than the chip has
// workgroup slots (512 = 256 CUs x 2) or a fractional number of rounds -- 15-25 % of the CU time of a launch has no
// workgroup resident, another 12-30 % only one -- and the backward pass issues two independent products per layer input
// (dX = dY . W and dW = dY^T . X) as two such launches plus a split-K sum.  Here the (tile, K-step) pairs of all products
// of the list form one sequence of `total` unit steps (a 128 x 128 x 32 slab of MFMAs each); workgroup r of G owns
// the contiguous range [total r / G, total (r + 1) / G) and walks it: whole tiles get their epilogue directly, a tile cut by a
// range boundary is finished by whichever of its contributors arrives LAST (arrival counter; slabs of raw accumulators
// summed in k order, so the result does not depend on timing).  A weight gradient's K = rows-of-the-batch dimension is
// covered by as many ranges as its length needs: split-K without a split count and without a reduction launch.
// Ranges are dealt to the XCDs in contiguous runs (xcd_remap), so neighbours in the sequence -- tiles of one row panel,
// or K-chunks of one tile -- share that XCD's L2.
// Hybrid: cutting costs (a 64-KB slab store per contributor + a serial re-read by the finisher), so only what must be cut
// is: of a product with T tiles, floor(T / G) G tiles are dealt WHOLE, floor(T / G) per workgroup (tile j G + r to range r);
// the remaining T mod G tiles of every product -- all tiles of a weight gradient -- form the sequence that is cut into equal
// ranges.  The cut part runs first, so its slabs are on their way while everybody still has whole tiles to multiply.
//
// Same tile machinery as sgemm_dma_kernel<128, 128, 64, 64, .., 32, 2> (gemm.hip): LDS-DMA ring of 2 x 32-deep K-steps,
// swizzled unpadded images, inline-asm fragment reads with counted lgkmcnt, one raw s_barrier per K-step, epilogues of
// gemm_common.h.  The ring restarts at every segment (tile or part of a tile) of a range.
#include <algorithm>

#include "gemm_common.h"

namespace msn {

constexpr int LT = 128;                        // tile edge (both directions)
constexpr int LW = 64;                         // wave tile edge
constexpr int LTM = LW / 32, LTN = LW / 32;    // 32 x 32 accumulator tiles per wave
constexpr int LNW = 4;                         // waves per workgroup
constexpr int LSTAGES = 2;
constexpr int kAccFloats = LNW * LTM * LTN * 16 * 64;     // one tile's accumulators in register order
constexpr int kSlabFloats = kAccFloats + LT;             // + the tile rows' partial column sums (wgrad + bias gradient)
constexpr int kListMax = 3;

struct ListItem {
    GemmArgs g;          // A, B, C, bias, aux, sizes, epilogue; splits = 1, tiles_m / tiles_n of 128 x 128 tiles
    int kind;            // 0: A [M][K], B [N][K]   1: A [M][K], B [K][N]   2: A [K][M], B [K][N]   3: kind 2 + column sums of A
    int ksteps;          // K / 32
    int whole;           // tiles dealt whole to every workgroup: tile j G + r, j < whole
    int tile0;           // = whole * G: first tile of the part that is cut into ranges
    long long first;     // position of tile0's first unit step in the launch's cut sequence
};
struct ListArgs {
    ListItem it[kListMax];
    int n, G;            // G workgroups
    int Gt, stride;      // the cut sequence is shared by Gt <= G / stride of them (r % stride == 0, r / stride < Gt): short sequences
                         // are not cut into pieces of less than ~3 K-steps, and no range is empty
    long long total;     // unit steps of the cut sequence
    float* slabs;        // [Gt][2][kSlabFloats]: a range's first and last segment, when they are parts of tiles
    unsigned* counters;  // [Gt] arrival counters, indexed by a cut tile's first contributing range; zero between launches
};

__device__ __forceinline__ long long range_lo(const ListArgs& L, int r) { return L.total * r / L.Gt; }
// the range that owns unit step x: max r with total r / Gt <= x
__device__ __forceinline__ int range_of(const ListArgs& L, long long x) { return (int)(((x + 1) * L.Gt - 1) / L.total); }

// K loop of one segment: acc += A[m0.., k_begin ..] . B[.., n0..] over nkt K-steps (the loop of sgemm_dma_kernel, CONV = 0).
template <bool AKM, bool BKM, bool CSUM>
__device__ __forceinline__ void list_kloop(const GemmArgs& p, const int64_t m0, const int64_t n0, const int64_t k_begin,
                                           const int nkt, float* smem, f32x16 (&acc)[LTM][LTN], float (&csum)[LTM]) {
    using TA = DmaTile<LT, AKM, BK>;
    using TB = DmaTile<LT, BKM, BK>;
    constexpr int STAGE = TA::kFloats + TB::kFloats;
    constexpr int PA = TA::kPieces / LNW, PB = TB::kPieces / LNW;
    constexpr int NKO = BK / 8;
    constexpr int NREADS = LTM * TA::kReads + LTN * TB::kReads;
    const unsigned smem_addr = (unsigned)(uintptr_t)(lptr_t*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / 2) * LW, wn0 = (wave % 2) * LW;

    const float* sa[PA];
    const float* sb[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) sa[i] = TA::src(p.A, p.lda, m0, p.M, k_begin, wave * PA + i, lane);
#pragma unroll
    for (int i = 0; i < PB; ++i) sb[i] = TB::src(p.B, p.ldb, n0, p.N, k_begin, wave * PB + i, lane);
    const int64_t a_step = AKM ? (int64_t)BK * p.lda : BK, b_step = BKM ? (int64_t)BK * p.ldb : BK;
    auto issue = [&](int kt) {
        float* base = smem + (kt % LSTAGES) * STAGE;
#pragma unroll
        for (int i = 0; i < PA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(sa[i] + kt * a_step), (lptr_t*)(base + (wave * PA + i) * 256), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(sb[i] + kt * b_step),
                                             (lptr_t*)(base + TA::kFloats + (wave * PB + i) * 256), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < LSTAGES; ++s)
        if (s < nkt) issue(s);
    if (nkt >= LSTAGES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LSTAGES - 1) * (PA + PB)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    typename TA::Frag fa[2][LTM];
    typename TB::Frag fb[2][LTN];
    auto request = [&](auto set, int slot, int ko) {
        constexpr int S = decltype(set)::value;
        const unsigned as = smem_addr + 4u * (slot * STAGE);
        const unsigned bs = as + 4u * TA::kFloats;
#pragma unroll
        for (int i = 0; i < LTM; ++i) TA::frag_issue(fa[S][i], as, wm0 + 32 * i + l32, ko, h);
#pragma unroll
        for (int j = 0; j < LTN; ++j) TB::frag_issue(fb[S][j], bs, wn0 + 32 * j + l32, ko, h);
    };
    auto multiply = [&](auto set, bool younger_in_flight) {
        constexpr int S = decltype(set)::value;
        if (younger_in_flight) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NREADS) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (CSUM) {
#pragma unroll
            for (int i = 0; i < LTM; ++i)
                csum[i] += (TA::template get<0>(fa[S][i]) + TA::template get<1>(fa[S][i])) +
                           (TA::template get<2>(fa[S][i]) + TA::template get<3>(fa[S][i]));
        }
#define MSN_MFMA_SWEEP(C)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < LTM; ++i) _Pragma("unroll") for (int j = 0; j < LTN; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(TA::template get<C>(fa[S][i]), TB::template get<C>(fb[S][j]), \
                                                         acc[i][j], 0, 0, 0);
        MSN_MFMA_SWEEP(0) MSN_MFMA_SWEEP(1) MSN_MFMA_SWEEP(2) MSN_MFMA_SWEEP(3)
#undef MSN_MFMA_SWEEP
        __builtin_amdgcn_sched_barrier(0);
    };
    int slot = 0;
    if (nkt > 0) request(S0{}, 0, 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int next_slot = slot + 1 == LSTAGES ? 0 : slot + 1;
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
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // 2-slot ring: the next K-step's pieces have landed
                __builtin_amdgcn_s_barrier();
                if (kt + LSTAGES < nkt) issue(kt + LSTAGES);         // into `slot`
                if (has_next) request(S0{}, next_slot, 0);
                multiply(S1{}, has_next);
            }
        }
        slot = next_slot;
    }
}

// One segment of a range: K-steps [ks, ke) of tile `tile` of product I.  `r` = this workgroup's range, `lo` its first
// unit step, `pos` the segment's first unit step.
template <bool AKM, bool BKM, bool CSUM>
__device__ __forceinline__ void list_segment(const ListArgs& L, const ListItem& I, const int tile, const int ks, const int ke,
                                             const int r, const long long lo, const long long pos, float* smem) {
    const GemmArgs& p = I.g;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int64_t m0 = (int64_t)tm * LT, n0 = (int64_t)tn * LT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l32 = lane & 31;
    const int wm0 = (wave / 2) * LW, wn0 = (wave % 2) * LW;
    f32x16 acc[LTM][LTN];
    float csum[LTM];
#pragma unroll
    for (int i = 0; i < LTM; ++i) {
        csum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < LTN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    }
    list_kloop<AKM, BKM, CSUM>(p, m0, n0, (int64_t)ks * BK, ke - ks, smem, acc, csum);
    // column sums of A (the bias gradient of a wgrad): one wave column of the first tile column owns each row
    const bool cs_owner = CSUM && wn0 == 0 && tn == 0;
    float cs[LTM];
#pragma unroll
    for (int i = 0; i < LTM; ++i) cs[i] = CSUM ? csum[i] + __shfl_xor(csum[i], 32, 64) : 0.f;   // lane halves hold k = 4h .. 4h + 3
    if (ks == 0 && ke == I.ksteps) {     // a whole tile
        if (cs_owner && h == 0) {
#pragma unroll
            for (int i = 0; i < LTM; ++i) {
                const int64_t row = m0 + wm0 + 32 * i + l32;
                if (row < p.M) p.colsum[row] = cs[i];
            }
        }
        gemm_epilogue<LTM, LTN>(acc, p, m0, n0, wm0, wn0, l32, h, 0);
        return;
    }
    // ---- part of a tile: publish the raw accumulators; the contributor that arrives last sums all parts in k order.
    // Visibility across XCDs as in finish_tail (gemm_common.h): slab words are relaxed device-scope atomics, every
    // thread waits for its own stores, then the workgroup barrier, then the arrival is counted.
    {
        float* slab = L.slabs + ((int64_t)r * 2 + (pos == lo ? 0 : 1)) * kSlabFloats;
        float* base = slab + wave * (LTM * LTN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < LTM; ++i)
#pragma unroll
            for (int j = 0; j < LTN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    __hip_atomic_store(base + ((i * LTN + j) * 16 + q) * 64, acc[i][j][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cs_owner && h == 0) {
#pragma unroll
            for (int i = 0; i < LTM; ++i)
                __hip_atomic_store(slab + kAccFloats + wm0 + 32 * i + l32, cs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const long long a = I.first + (long long)(tile - I.tile0) * I.ksteps;   // the tile's unit steps are [a, a + ksteps)
    const int r_first = range_of(L, a), r_last = range_of(L, a + I.ksteps - 1);
    unsigned* flag = reinterpret_cast<unsigned*>(smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(L.counters + r_first, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = prev + 1u == (unsigned)(r_last - r_first + 1) ? 1u : 0u;
        if (last) __hip_atomic_store(L.counters + r_first, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *flag = last;
    }
    __syncthreads();
    const unsigned last = *flag;
    __syncthreads();                     // the flag word belongs to the next segment's ring again
    if (last == 0u) return;
#pragma unroll
    for (int i = 0; i < LTM; ++i) {
        cs[i] = 0.f;
#pragma unroll
        for (int j = 0; j < LTN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    }
    for (int rr = r_first; rr <= r_last; ++rr) {
        // range rr's part of this tile is its FIRST segment when the range starts inside the tile, else its last
        const float* slab = L.slabs + ((int64_t)rr * 2 + (range_lo(L, rr) >= a ? 0 : 1)) * kSlabFloats;
        const float* base = slab + wave * (LTM * LTN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < LTM; ++i)
#pragma unroll
            for (int j = 0; j < LTN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    acc[i][j][q] += __hip_atomic_load(base + ((i * LTN + j) * 16 + q) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cs_owner && h == 0) {
#pragma unroll
            for (int i = 0; i < LTM; ++i)
                cs[i] += __hip_atomic_load(slab + kAccFloats + wm0 + 32 * i + l32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (cs_owner && h == 0) {
#pragma unroll
        for (int i = 0; i < LTM; ++i) {
            const int64_t row = m0 + wm0 + 32 * i + l32;
            if (row < p.M) p.colsum[row] = cs[i];
        }
    }
    gemm_epilogue<LTM, LTN>(acc, p, m0, n0, wm0, wn0, l32, h, 0);
}

// The list is indexed at run time (which product a range is in).  Indexing the by-value argument would make hipcc copy all
// of it to scratch (832 bytes per lane); the kernel reads it where it already lies -- the kernarg segment, whose first
// bytes are this kernel's only explicit argument -- through a constant-address-space pointer: scalar loads, no copy.
typedef __attribute__((address_space(4))) const void kernarg_ptr_t;
__global__ __launch_bounds__(64 * LNW, 2) void sgemm_list_kernel(const ListArgs list_by_value) {
    __shared__ __attribute__((aligned(16))) float smem[LSTAGES * 2 * LT * BK];
    (void)list_by_value;
    const ListArgs& L = *(const ListArgs*)(kernarg_ptr_t*)__builtin_amdgcn_kernarg_segment_ptr();
    const int r = xcd_remap((int)blockIdx.x, L.G);
#define MSN_LIST_SEGMENT(I, tile, ks, ke, rt, lo, pos)                                                              \
    switch ((I).kind) {                                                                                             \
        case 0: list_segment<false, false, false>(L, I, tile, ks, ke, rt, lo, pos, smem); break;                     \
        case 1: list_segment<false, true, false>(L, I, tile, ks, ke, rt, lo, pos, smem); break;                      \
        case 2: list_segment<true, true, false>(L, I, tile, ks, ke, rt, lo, pos, smem); break;                       \
        default: list_segment<true, true, true>(L, I, tile, ks, ke, rt, lo, pos, smem); break;                       \
    }
    // ---- the cut sequence (ranges of equal length over the tiles that are not dealt whole)
    if (L.total > 0 && r % L.stride == 0 && r / L.stride < L.Gt) {
        const int rt = r / L.stride;
        const long long lo = range_lo(L, rt), end = range_lo(L, rt + 1);
        long long pos = lo;
        while (pos < end) {
            int pi = 0;
            if (L.n > 1 && pos >= L.it[1].first) pi = 1;
            if (L.n > 2 && pos >= L.it[2].first) pi = 2;
            const ListItem& I = L.it[pi];
            const long long local = pos - I.first;
            const int t = (int)(local / I.ksteps);
            const int ks = (int)(local - (long long)t * I.ksteps);
            const int ke = (int)std::min<long long>(I.ksteps, ks + (end - pos));
            MSN_LIST_SEGMENT(I, I.tile0 + t, ks, ke, rt, lo, pos)
            pos += ke - ks;
        }
    }
    // ---- whole tiles: the same number for every workgroup
    for (int pi = 0; pi < L.n; ++pi) {
        const ListItem& I = L.it[pi];
        for (int j = 0; j < I.whole; ++j) MSN_LIST_SEGMENT(I, j * L.G + r, 0, I.ksteps, 0, 0ll, 0ll)
    }
#undef MSN_LIST_SEGMENT
}

// which single products / lists the kernel takes: plain fp32, 128-wide tiles, the LDS-DMA conditions of sgemm_impl
static bool list_takes(const msn_gemm_desc& d) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.B || !d.C) return false;
    int kind;
    if (d.opA == MSN_OP_N && d.opB == MSN_OP_T) kind = 0;
    else if (d.opA == MSN_OP_N && d.opB == MSN_OP_N) kind = 1;
    else if (d.opA == MSN_OP_T && d.opB == MSN_OP_N) kind = 2;
    else return false;
    if (d.colsum && kind != 2) return false;
    if (kind == 2 && (d.bias || d.epilogue != MSN_EPI_NONE)) return false;
    if (d.N <= 64 || d.M <= 64) return false;                      // narrow / short products keep their own tiles (gemm.hip plan)
    if (d.K % BK != 0 || d.lda % 4 != 0 || d.ldb % 4 != 0) return false;
    const int64_t a_ext = d.opA == MSN_OP_T ? d.M : d.K, b_ext = d.opB == MSN_OP_N ? d.N : d.K;
    if (a_ext % 4 != 0 || b_ext % 4 != 0 || a_ext < 4 || b_ext < 4) return false;
    if (((reinterpret_cast<uintptr_t>(d.A) | reinterpret_cast<uintptr_t>(d.B)) & 15) != 0) return false;
    if (d.lda < (d.opA == MSN_OP_N ? d.K : d.M) || d.ldb < (d.opB == MSN_OP_N ? d.N : d.K) || d.ldc < d.N) return false;
    if (d.epilogue < MSN_EPI_NONE || d.epilogue > MSN_EPI_ADD) return false;
    const bool needs_aux = d.epilogue == MSN_EPI_RELU_BWD || d.epilogue == MSN_EPI_GELU_BWD || d.epilogue == MSN_EPI_ADD;
    if (needs_aux && (!d.aux || d.ldaux < d.N)) return false;
    if (d.epilogue == MSN_EPI_GELU && d.aux && d.ldaux < d.N) return false;
    if (cdiv(d.M, LT) * cdiv(d.N, LT) >= (1ll << 24) || d.K / BK >= (1ll << 24)) return false;
    return true;
}

// Workgroups of a launch: the 512 resident ones (256 CUs x 2) once there is work for them; tiny lists get fewer, so that a
// range is not shorter than ~4 unit steps.
static int list_groups(long long all_steps) {
    long long g = std::min<long long>(512, std::max<long long>(8, all_steps / 4));
    g = g / 8 * 8;
    return (int)std::max<long long>(8, g);
}

unsigned* gemm_counter_slice(hipStream_t st);   // gemm.hip: one zeroed slice of 512 arrival counters per (device, stream)

bool gemm_list_takes(int n, const msn_gemm_desc* d) {
    if (n < 1 || n > kListMax || !d) return false;
    for (int i = 0; i < n; ++i)
        if (!list_takes(d[i])) return false;
    return true;
}
size_t gemm_list_ws_bytes() { return sizeof(float) * (size_t)512 * 2 * (size_t)kSlabFloats; }

int gemm_list_launch(int n, const msn_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st) {
    ListArgs L;
    long long all_steps = 0;
    for (int i = 0; i < n; ++i) all_steps += cdiv(d[i].M, LT) * cdiv(d[i].N, LT) * (d[i].K / BK);
    L.n = n, L.G = list_groups(all_steps);
    long long total = 0;
    for (int i = 0; i < n; ++i) {
        ListItem& I = L.it[i];
        GemmArgs& a = I.g;
        a = GemmArgs{};
        a.A = d[i].A, a.B = d[i].B, a.C = d[i].C, a.bias = d[i].bias, a.aux = d[i].aux;
        a.M = d[i].M, a.N = d[i].N, a.K = d[i].K, a.lda = d[i].lda, a.ldb = d[i].ldb, a.ldc = d[i].ldc, a.ldaux = d[i].ldaux;
        a.epilogue = d[i].epilogue;
        a.tiles_m = (int)cdiv(a.M, LT), a.tiles_n = (int)cdiv(a.N, LT);
        a.splits = 1, a.k_per_split = (int)std::min<int64_t>(a.K, 1 << 30), a.partial = nullptr;
        a.tail_tiles = 0, a.tail_splits = 1, a.tail_kps = 0, a.tail_partial = nullptr, a.tail_counter = nullptr;
        a.colsum = d[i].colsum;
        I.kind = d[i].opA == MSN_OP_T ? (d[i].colsum ? 3 : 2) : (d[i].opB == MSN_OP_T ? 0 : 1);
        I.ksteps = (int)(a.K / BK);
        const long long tiles = (long long)a.tiles_m * a.tiles_n;
        I.whole = (int)(tiles / L.G);
        I.tile0 = I.whole * L.G;
        I.first = total;
        total += (tiles - I.tile0) * I.ksteps;
    }
    for (int i = n; i < kListMax; ++i) L.it[i] = L.it[0], L.it[i].first = total, L.it[i].whole = 0;
    L.total = total;
    // the cut sequence is shared by Gt = G / stride workgroups: pieces of at least ~3 unit steps (G is a multiple of 8)
    // (every range must own at least one unit step: a tile's contributors are counted as r_last - r_first + 1)
    L.stride = 1;
    while (L.G / L.stride > 8 && (L.G / L.stride) % 2 == 0 && total < 3ll * (L.G / L.stride)) L.stride *= 2;
    L.Gt = (int)std::max<long long>(1, std::min<long long>(L.G / L.stride, total / 3));
    MSN_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)L.Gt * 2 * (size_t)kSlabFloats, "msn_sgemm_list: workspace %zu too small", ws_bytes);
    L.slabs = static_cast<float*>(ws);
    L.counters = gemm_counter_slice(st);
    MSN_REQUIRE(L.counters, "msn_sgemm_list: no arrival-counter slice left for this stream (more than 32 streams in use)");
    hipLaunchKernelGGL(sgemm_list_kernel, dim3(L.G), dim3(64 * LNW), 0, st, L);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

}  // namespace msn

using namespace msn;

extern "C" size_t msn_sgemm_list_workspace_bytes(int n, const msn_gemm_desc* d) {
    size_t need = 0;
    if (!d || n < 1) return 0;
    if (gemm_list_takes(n, d)) need = gemm_list_ws_bytes();
    for (int i = 0; i < n; ++i) {      // the one-by-one path (msn_set_gemm_list(0), or a product the list kernel does not take)
        if (d[i].M <= 0 || d[i].N <= 0 || d[i].K <= 0) continue;
        need = std::max(need, d[i].colsum ? msn_wgrad_bias_workspace_bytes(d[i].M, d[i].N, d[i].K)
                                          : msn_sgemm_workspace_bytes(d[i].opA, d[i].opB, d[i].M, d[i].N, d[i].K));
    }
    return need;
}

static int g_gemm_list = 1;   // 0: msn_sgemm_list issues its products one by one (measurements, bit-comparison of the two paths)
namespace msn { void gemm_set_single_rule(int rule); }
extern "C" int msn_set_gemm_list(int mode) {
    MSN_REQUIRE(mode >= 0 && mode <= 3, "msn_set_gemm_list: 0 = no work-list launches at all, 1 = default, 2 = lists only (single products never take "
                "the kernel), 3 = lists + every single product the kernel can take");
    g_gemm_list = mode != 0;
    msn::gemm_set_single_rule(mode == 1 ? 1 : (mode == 3 ? 2 : 0));
    return MSN_OK;
}

extern "C" int msn_sgemm_list(int n, const msn_gemm_desc* d, int precision, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(d && n >= 1 && n <= kListMax, "msn_sgemm_list: 1 .. %d products", kListMax);
    // (slices of arrival counters are keyed by stream handle and never released -- 32 per device; a long-lived process that has
    // used more streams than that gets the one-by-one path on the later ones instead of an error)
    if (g_gemm_list && precision == MSN_PREC_F32 && gemm_list_takes(n, d) && gemm_counter_slice(static_cast<hipStream_t>(stream)))
        return gemm_list_launch(n, d, ws, ws_bytes, static_cast<hipStream_t>(stream));
    for (int i = 0; i < n; ++i) {
        int rc;
        if (d[i].colsum)
            rc = msn_wgrad_bias(d[i].M, d[i].N, d[i].K, d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, d[i].colsum, precision, ws,
                                ws_bytes, stream);
        else
            rc = msn_sgemm(d[i].opA, d[i].opB, d[i].M, d[i].N, d[i].K, d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, d[i].bias,
                           d[i].epilogue, d[i].aux, d[i].ldaux, precision, ws, ws_bytes, stream);
        if (rc != MSN_OK) return rc;
    }
    return MSN_OK;
}
