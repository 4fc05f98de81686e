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
#include <algorithm>
#include <type_traits>

#include "msn_common.h"

namespace msn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int PBLK = 1024;          // bytes of one plane image of one 32 x 16 block
constexpr int BM = 256;             // tile rows

struct PgemmArgs {
    const unsigned char* A;         // NT: planes of the M x K operand.   TN: planes of dY (rows = reduction, N columns)
    const unsigned char* B;         // NT: planes of the N x K operand.   TN: planes of X  (rows = reduction, K columns)
    void* C;                        // fp32 matrix (ldc) or plane matrix (cbC column blocks)
    float* aux;
    const float* bias;
    int64_t ldc, ldaux;
    int64_t M;                      // NT: rows of A / C.  TN: reduction length
    int N, K;
    int rbA, rbB;                   // NT: row blocks of A / B.  TN: row blocks (both operands share the rows)
    int cbA, cbB;                   // column blocks of A / B (NT: both = K-steps)
    int cbC;                        // column blocks of a plane output
    int tiles_m, tiles_n, super_rows, epi;
    int colsum_rows;                // NT: column-sum partial rows per tile row (= WM of the kernel's wave layout)
    int chunk_steps;                // NT: K-steps per chunk (0 = the whole reduction in one accumulation)
    int tail_full;                  // NT: tiles [0, tail_full) are dealt whole; [tail_full, total) are the TAIL tiles, each cut into
    int tail_segs, tail_steps;      //     tail_segs K-segments of tail_steps K-steps (units): unit u -> workgroup u, raw sums -> tail_slabs[u]
    float* tail_slabs;              //     (0 segments: no tail split)
    int skew;                       // NT: start delay unit in shader cycles (0 = none): workgroup b waits ((b >> 3) & 3) * skew
    float* colpart;                 // NT (nullable): [2 * tiles_m][N] column sums of the values written to C
    int splits;                     // TN: reduction split
    int rb_per_split;               // TN: row blocks per split
    float* slabs;                   // TN: [splits][N][K] partials (splits > 1)
    const float* scaleA;            // fp16 planes: the operands hold x * 2^e; scaleA[0] * scaleB[0] = 2^-(eA + eB) multiplies every sum
    const float* scaleB;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int OFF>
__device__ __forceinline__ void ds_read128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_vm(int n) {        // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

__device__ __forceinline__ u16 f2bf(float f) {          // round to nearest even; NaN stays NaN (plain cast)
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const u16*>(&b);
}
__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }

// x -> NP planes (u16 each).  Each residual x - p0 (- p1) is exact in fp32.
template <int NP>
__device__ __forceinline__ void split_planes(float x, u16 (&pl)[NP]) {
    float r = x;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        pl[k] = f2bf(r);
        r -= bf2f(pl[k]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NT kernel.  LDS slot = [A: 8 row blocks][NP planes][1 KB] then [B: BN / 32 row blocks][NP][1 KB].  Inside a plane image
// the 16-byte half h of row r sits at chunk 2 r + (h ^ swz(r)), swz(r) = (r >> 3) & 1 (applied on the DMA source address):
// a ds_read_b128 of the 32x32x16 operand (lane l: row l & 31, half l >> 5) is served in 16-lane groups {0-3, 12-15,
// 20-27} / {4-11, 16-19, 28-31}; row r covers banks 8 r + 4 (h ^ swz) .. + 3 (mod 64): with the swizzle every group touches
// 16 distinct 4-bank quads -> conflict-free.
//
// DUAL (the fp32-grade form, NP = 3, BN = 128): TWO accumulator sets.  The bf16 MFMA adds its 16-term dot product into the
// accumulator with a truncating (biased) rounding -- unlike the fp32-input MFMA's round-to-nearest -- so every MFMA into a
// LARGE accumulator costs a systematic ~1/18 ulp and the error of one accumulator grows like K^1.5 (measured: 5.4 x the
// native kernel's at K = 1536 with all six products in one accumulator).  The p0.q0 products alone go to `acc`; the five
// small products (<= 2^-8 of it) go to `acc2`, whose roundings are 2^-8 smaller; the epilogue adds the two in fp32.  With
// that the error is 0.3 x the native kernel's up to K = 768; longer reductions are cut into K chunks of <= 512 columns whose
// partial sums meet in C by fp32 adds (host: chunk_steps).  (The TN kernel instead folds the p0.q0 product in by VALU adds:
// there the reduction is thousands of rows long.  The same fold on the NT kernel costs 11 % at K = 384 and leaves the
// maximum error at K >= 1152 where chunks put it -- profiles/r04_pgemm_accuracy.txt.)
// Wave layout WM x WN (rows x columns of the tile): 2 x 4 = eight waves of 128 x (BN / 4); 4 x 2 = eight waves of 64 x (BN / 2)
// (12 instead of 15 fragment reads per 24 MFMAs on BN = 128: +7-12 %, the 3-plane default; four waves of 128 x 64 with the
// whole register file each were 15-25 % slower).  STAG: the two waves of a SIMD issue their LDS-DMA pieces in different halves of a K-step
// (waves 0-3 behind the first products, waves 4-7 behind the last ones) instead of both stalling in the same gaps.
template <int NP, int BN, bool OUTP, bool DUAL, int WM = 2, int WN = 4, bool STAG = false, bool F16 = false, bool S16 = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void pgemm_nt_kernel(const PgemmArgs p) {
    static_assert(!F16 || !OUTP, "fp16 planes: fp32 results only");
    static_assert(!S16 || (NP == 3 && DUAL && !F16 && !STAG && WM == 4 && WN == 2 && BN == 128), "the 16 x 16 x 32 form: three bf16 planes, two accumulator sets, 4 x 2 waves");
    constexpr int NW = WM * WN;
    constexpr int ARB = BM / 32, BRB = BN / 32, AP = ARB * NP, BP = BRB * NP, PIECES = AP + BP;
    constexpr int SLOT = PIECES * PBLK;
    constexpr int STG = 6 * 1024;                    // epilogue staging per wave (one 32 x 32 tile: 4 KB fp32 / 6 plane images)
    constexpr int NSLOT = (4 * SLOT + NW * STG <= 160 * 1024) ? 4 : 3;
    static_assert(NSLOT * SLOT + NW * STG <= 160 * 1024, "LDS: ring + staging");
    constexpr int MAXQ = (PIECES + NW - 1) / NW;     // pieces per wave and K-step (the last one only for the low waves)
    constexpr int MT = 8 / WM, NT = BN / (32 * WN);  // 32 x 32 MFMA tiles per wave: rows x columns
    constexpr int NG = S16 ? 3 : NP * (NP + 1) / 2;  // MFMA groups per K-step (S16: three plane-PAIR products)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSLOT * SLOT + NW * STG];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int G = gridDim.x;
    const int total = p.tiles_m * p.tiles_n;
    const int nk = p.cbA;                            // K-steps per tile

    // tile order: ids are dealt round-robin to the 8 XCDs; each XCD gets a contiguous run of tiles walked in super-rows
    // (SR tile-rows x all tile columns, row-fastest) so that its workgroups share A row panels and the weight in one L2
    auto locate = [&](int t, int& tm_, int& tn_) {
        const int q = total / 8, r = total % 8, x = t % 8, i = t / 8;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        const int SR = p.super_rows;
        const int sr = bid / (SR * p.tiles_n), j = bid % (SR * p.tiles_n);
        const int rows_sr = min(SR, p.tiles_m - sr * SR);
        tm_ = sr * SR + j % rows_sr;
        tn_ = j / rows_sr;
    };
    // idx-th SEGMENT of this workgroup: its whole tiles first (every K-step), then -- tail split -- at most one unit: a K range
    // of one of the tiles that do not fill a round of the G workgroups (the host cuts each of them into tail_segs ranges so that
    // the units spread over the chip instead of a few workgroups multiplying a whole extra tile while the others wait)
    const int n_whole = p.tail_segs ? p.tail_full : total;
    auto seg_of = [&](int idx, int& tm_, int& tn_, int& kb_, int& ke_) -> bool {
        const int t = blockIdx.x + idx * G;
        if (t < n_whole) {
            locate(t, tm_, tn_);
            kb_ = 0, ke_ = nk;
            return true;
        }
        const int first_tail_idx = n_whole / G;                       // n_whole is a multiple of G when the tail is split
        if (p.tail_segs && idx == first_tail_idx && (int)blockIdx.x < (total - n_whole) * p.tail_segs) {
            const int u = blockIdx.x;
            locate(n_whole + u / p.tail_segs, tm_, tn_);
            kb_ = (u % p.tail_segs) * p.tail_steps;
            ke_ = min(nk, kb_ + p.tail_steps);
            return kb_ < ke_;
        }
        return false;
    };

    // ---- producer: the LDS-DMA stream.  Global K-step g (over all tiles of this workgroup) lives in slot g % NSLOT.
    // (S16: no swizzle -- 16-row fragments of rows r, r + 8 take opposite halves in one ds_read_b128 lane group: conflict-free as laid)
    const int lane_src = S16 ? lane * 16 : (((lane >> 1) * 2) + ((lane & 1) ^ ((lane >> 4) & 1))) * 16;
    const unsigned char* pbase[MAXQ];                // source of piece q of the producer's tile at K-step 0 (wave-uniform)
    int p_idx = 0, p_k = 0, p_ke = 0, p_slot = 0, p_tm = 0, p_tn = 0;
    bool p_live = seg_of(0, p_tm, p_tn, p_k, p_ke);
    auto set_bases = [&]() {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int id = wave + NW * q;
            if (id < AP) {
                const int rb = min(p_tm * ARB + id / NP, p.rbA - 1);            // rows past the edge: clamped, never stored
                pbase[q] = p.A + ((int64_t)rb * nk * NP + id % NP) * PBLK;
            } else {
                const int id2 = min(id, PIECES - 1) - AP;
                const int rb = min(p_tn * BRB + id2 / NP, p.rbB - 1);
                pbase[q] = p.B + ((int64_t)rb * nk * NP + id2 % NP) * PBLK;
            }
        }
    };
    if (p_live) set_bases();
    auto issue_q = [&](int q) {                      // piece q of the producer's current K-step
#ifdef MSN_ABL_PG_NODMA                      // diagnostic build (tools/microbench/build_ablate.sh): no operand traffic at all
        return;
#endif
        const int id = wave + NW * q;
        if (id < PIECES)
            __builtin_amdgcn_global_load_lds((gptr_t*)(pbase[q] + (int64_t)p_k * (NP * PBLK) + lane_src),
                                             (lptr_t*)(lds + p_slot * SLOT + id * PBLK), 16, 0, 0);
    };
    // gap c of NG (behind the c-th plane product of a K-step): pieces [c MAXQ / NG, (c + 1) MAXQ / NG); STAG: the low waves
    // issue theirs in the gaps 0 .. HG - 1, the high waves in HG .. NG - 2 (none behind the barrier: the counted wait stays
    // the same for both halves)
    constexpr int HG = (NG - 1) / 2;
    const bool hi_half = wave >= NW / 2;
    auto issue_chunk = [&](auto c_) {
        constexpr int c = decltype(c_)::value;
        if (p_live) {
            if constexpr (STAG) {
                if (c < HG && !hi_half) {
#pragma unroll
                    for (int q = c * MAXQ / HG; q < (c + 1) * MAXQ / HG; ++q) issue_q(q);
                } else if (c >= HG && c < NG - 1 && hi_half) {
#pragma unroll
                    for (int q = (c - HG) * MAXQ / (NG - 1 - HG); q < (c - HG + 1) * MAXQ / (NG - 1 - HG); ++q) issue_q(q);
                }
            } else {
#pragma unroll
                for (int q = c * MAXQ / NG; q < (c + 1) * MAXQ / NG; ++q) issue_q(q);
            }
        }
    };
    auto advance = [&]() {                           // producer -> next global K-step
        if (!p_live) return;
        p_slot = (p_slot + 1 == NSLOT) ? 0 : p_slot + 1;
        if (++p_k == p_ke) {
            p_live = seg_of(++p_idx, p_tm, p_tn, p_k, p_ke);
            if (p_live) set_bases();
        }
    };
    // pieces of this wave issued by the chunks 0 .. NG - 2 of one K-step (all waves own every piece q < MAXQ - 1)
    constexpr int Q_BEFORE = STAG ? MAXQ : (NG - 1) * MAXQ / NG;
    const int ppw = (wave + NW * (MAXQ - 1) < PIECES) ? MAXQ : MAXQ - 1;      // pieces of this wave per K-step

    // ---- fragments
    const int frow = lane & 31, fhalf = lane >> 5;
    const unsigned frag_lane = (unsigned)(frow * 32 + 16 * (fhalf ^ ((frow >> 3) & 1)));
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)lds;
    const unsigned fragA = lds0 + frag_lane + (unsigned)(wm * MT * NP * PBLK);                   // + (i * NP + plane) KB
    const unsigned fragB = lds0 + frag_lane + (unsigned)((AP + wn * NT * NP) * PBLK);           // + (j * NP + plane) KB

    f32x16 acc[MT][NT], acc2[DUAL ? MT : 1][DUAL ? NT : 1];
    bf16x8 fa[2][MT], fb0[2][NT], fbh[NP > 1 ? NP - 1 : 1][NT];
    // ---- S16: v_mfma_f32_16x16x32_bf16 (holds 2.0 GHz under load where the 32 x 32 x 16 shape holds 1.75: profiles/
    // r04_mfma_shape_clock.txt).  The 32 k of one instruction are the SAME 16 columns of TWO PLANES: lane group kg = lane >> 4
    // supplies plane (kg >> 1 ? second : first), 16-byte half kg & 1 of row lane & 15 -- so [p0 | p1].[q0 | q1] = p0q0 + p1q1 (first
    // accumulator set), [p0 | p1].[q1 | q0] = p0q1 + p1q0 and [p0 | p2].[q2 | q0] = p0q2 + p2q0 (second set): three 16-cycle
    // instructions per 16 x 16 tile and K-step.  A 32 x 32 accumulator tile acc[i][j] is four 16 x 16 sub-tiles: chunk b =
    // 2 (row half) + (column half); the lane holds row 16 (b >> 1) + (lane & 15), columns 16 (b & 1) + 4 (lane >> 4) + r.
    const int r16 = lane & 15, kg = lane >> 4;
    const unsigned s16_lane = (unsigned)(r16 * 32 + (kg & 1) * 16);
    const unsigned s16A = lds0 + s16_lane + (unsigned)(wm * MT * NP * PBLK);
    const unsigned s16B = lds0 + s16_lane + (unsigned)((AP + wn * NT * NP) * PBLK);
    const unsigned fA01 = s16A + (unsigned)((kg >> 1) * PBLK), fA02 = s16A + (unsigned)((kg >> 1) * 2 * PBLK);
    const unsigned fB01 = s16B + (unsigned)((kg >> 1) * PBLK), fB10 = s16B + (unsigned)((1 - (kg >> 1)) * PBLK);
    const unsigned fB20 = s16B + (unsigned)((1 - (kg >> 1)) * 2 * PBLK);
    bf16x8 fr[2][2][S16 ? 4 : 1];                    // [set][A-type | B-type][16-row sub-tile]
    auto req16 = [&](bf16x8 (&dst)[S16 ? 4 : 1], unsigned base) {
        if constexpr (S16) {
            static_for<0, 4>([&](auto t_) {
                constexpr int t = decltype(t_)::value;
                ds_read128<(t >> 1) * NP * PBLK + (t & 1) * 512>(dst[t], base);
            });
        }
    };

    auto req_a = [&](bf16x8 (&dst)[MT], unsigned slot_off, auto pl_) {
        constexpr int pl = decltype(pl_)::value;
#ifdef MSN_ABL_PG_NOFRAG                     // diagnostic build: no fragment reads (the MFMAs multiply whatever the registers hold)
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "=v"(dst[i]));
        return;
#endif
        static_for<0, MT>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            ds_read128<(i * NP + pl) * PBLK>(dst[i], fragA + slot_off);
        });
    };
    auto req_b = [&](bf16x8 (&dst)[NT], unsigned slot_off, auto pl_) {
        constexpr int pl = decltype(pl_)::value;
#ifdef MSN_ABL_PG_NOFRAG
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "=v"(dst[j]));
        return;
#endif
        static_for<0, NT>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            ds_read128<(j * NP + pl) * PBLK>(dst[j], fragB + slot_off);
        });
    };
    // D^T tile = B_frag . A_frag^T: lane gets row m = lane & 31 of the tile and columns n = 8 b + 4 (lane >> 5) + r
    auto mult = [&](const bf16x8 (&a)[MT], const bf16x8 (&b)[NT], auto small_) {
        constexpr bool SMALL = decltype(small_)::value && DUAL;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if constexpr (F16) {
                    const f16x8 hb = __builtin_bit_cast(f16x8, b[j]), ha = __builtin_bit_cast(f16x8, a[i]);
                    if constexpr (SMALL) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc2[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[i][j], 0, 0, 0);
                } else {
                    if constexpr (SMALL) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc2[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mult16 = [&](const bf16x8 (&a)[S16 ? 4 : 1], const bf16x8 (&b)[S16 ? 4 : 1], auto small_) {
        constexpr bool SMALL = decltype(small_)::value;
        if constexpr (S16) {
            static_for<0, 16>([&](auto u_) {
                constexpr int u = decltype(u_)::value, ti = u >> 2, tj = u & 3, i = ti >> 1, j = tj >> 1, bch = 2 * (ti & 1) + (tj & 1);
                f32x16& dstv = SMALL ? acc2[i][j] : acc[i][j];
                f32x4 c = {dstv[4 * bch], dstv[4 * bch + 1], dstv[4 * bch + 2], dstv[4 * bch + 3]};
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[tj], a[ti], c, 0, 0, 0);
                dstv[4 * bch] = c[0], dstv[4 * bch + 1] = c[1], dstv[4 * bch + 2] = c[2], dstv[4 * bch + 3] = c[3];
            });
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using BIG = std::integral_constant<bool, false>;
    using SML = std::integral_constant<bool, true>;
    // request the first fragments of a K-step (A plane 0 and every B plane) -- order matters for the counted waits
    auto req_first = [&](bf16x8 (&a0)[MT], bf16x8 (&b0)[NT], unsigned slot_off) {
        req_a(a0, slot_off, std::integral_constant<int, 0>{});
        req_b(b0, slot_off, std::integral_constant<int, 0>{});
        static_for<1, NP>([&](auto pb_) {
            constexpr int pb = decltype(pb_)::value;
            req_b(fbh[pb - 1], slot_off, pb_);
        });
    };

    // req_first at a tile boundary (behind an epilogue, before the first tile): the registers have three defining sites that
    // meet at the K loop's head, and where the register allocator does not coalesce them it copies them -- a copy of a
    // fragment that has not ARRIVED yet (the compiler believes the asm that requested it defined it) is a stale fragment: one
    // K-step of two MFMA tiles came out wrong in a few tiles per launch of the plane-output kernel.  So here the reads are
    // waited for, and every register is re-defined behind the wait (an empty asm) so that any copy sits behind it too.
    auto req_first_settled = [&](unsigned slot_off) {
        if constexpr (S16) return;                   // (this form requests every fragment inside its step)
        req_first(fa[0], fb0[0], slot_off);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(fa[0][i]));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fb0[0][j]));
#pragma unroll
        for (int k = 0; k < NP - 1; ++k)
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(fbh[k][j]));
        __builtin_amdgcn_sched_barrier(0);
    };

    int c_slot = 0;                                  // slot of the consumer's current K-step
    // One K-step; PAR = parity of the step inside its tile (a tile has an EVEN number of K-steps: the plane format pads K).
    // On entry outstanding LDS reads, oldest first: A0 (MT), B0 (NT), B1 .. B_{NP-1} (NT each).  The first fragments of the
    // next K-step are requested before this step's last product; behind the last step of a tile (or of a K chunk) the
    // epilogue requests them once it is done -- the ring of K-steps itself runs on across tiles.
    auto step = [&](auto par_, bool last = false) {
        constexpr int PAR = decltype(par_)::value;
        if constexpr (S16) {
            // Every fragment is requested AND consumed inside its step: fr[0][0] = [p0 | p1] of A, fr[0][1] = [q0 | q1] of B,
            // fr[1][0] = [p0 | p2], fr[1][1] = [q2 | q0]; [q1 | q0] is [q0 | q1] with the lane halves exchanged (two
            // v_permlane32_swap per register pair, in place) instead of a fifth read: 16 ds_read_b128 per step (12 in the
            // 32 x 32 x 16 form).  NOT carried over from that form: the next step's first fragments requested behind the barrier.
            // With 16 fragment registers more than that form holds, the allocator (256 VGPRs) copies or spills the in-flight
            // destinations at the step's merges -- multiplying fragments that have not arrived (tools/check_fragment_waits.py
            // finds such copies in the ISA; the integer tests found them first).  What this costs is most of what the shape gains.
            const unsigned cur = (unsigned)(c_slot * SLOT);
            const int n_slot = (c_slot + 1 == NSLOT) ? 0 : c_slot + 1;
            const bool p_was_live = p_live;
            (void)last;
            req16(fr[0][0], fA01 + cur);
            req16(fr[0][1], fB01 + cur);
            req16(fr[1][1], fB20 + cur);
            wait_lgkm<4>();
            mult16(fr[0][0], fr[0][1], BIG{});           // p0q0 + p1q1
            issue_chunk(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            req16(fr[1][0], fA02 + cur);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                u32x4 w = __builtin_bit_cast(u32x4, fr[0][1][t]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u32x2 r = __builtin_amdgcn_permlane32_swap(w[2 * h], w[2 * h + 1], false, false);
                    const u32x2 q = __builtin_amdgcn_permlane32_swap(r[1], r[0], false, false);
                    w[2 * h] = q[0], w[2 * h + 1] = q[1];
                }
                fr[0][1][t] = __builtin_bit_cast(bf16x8, w);
            }
            __builtin_amdgcn_sched_barrier(0);
            mult16(fr[0][0], fr[0][1], SML{});           // p0q1 + p1q0
            issue_chunk(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            wait_lgkm<0>();
            wait_vm(p_was_live ? (NSLOT - 3) * ppw + Q_BEFORE : 0);
#ifndef MSN_ABL_PG_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
            mult16(fr[1][0], fr[1][1], SML{});           // p0q2 + p2q0
            issue_chunk(std::integral_constant<int, NG - 1>{});
            __builtin_amdgcn_sched_barrier(0);
            advance();
            c_slot = n_slot;
            return;
        }
        constexpr int QA = (NP & 1) ? PAR : 0;       // fa buffer of plane 0; plane pa sits in fa[(QA + pa) & 1]
        const unsigned cur = (unsigned)(c_slot * SLOT);
        const int n_slot = (c_slot + 1 == NSLOT) ? 0 : c_slot + 1;
        const unsigned nxt = (unsigned)(n_slot * SLOT);
        const bool p_was_live = p_live;
        req_a(fa[QA ^ 1], cur, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        // plane 0 of A against every plane of B
        static_for<0, NP>([&](auto pb_) {
            constexpr int pb = decltype(pb_)::value;
            wait_lgkm<NT*(NP - 1 - pb) + MT>();
            if constexpr (pb == 0) mult(fa[QA], fb0[PAR], BIG{});
            else mult(fa[QA], fbh[pb - 1], SML{});
            issue_chunk(std::integral_constant<int, pb>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // planes 1 .. NP - 2 of A
        static_for<1, NP - 1>([&](auto pa_) {
            constexpr int pa = decltype(pa_)::value;
            constexpr int g0 = pa * NP - pa * (pa - 1) / 2;         // products before plane pa
            req_a(fa[(QA + pa + 1) & 1], cur, std::integral_constant<int, pa + 1>{});
            wait_lgkm<MT>();
            static_for<0, NP - pa>([&](auto pb_) {
                constexpr int pb = decltype(pb_)::value;
                if constexpr (pb == 0) mult(fa[(QA + pa) & 1], fb0[PAR], SML{});
                else mult(fa[(QA + pa) & 1], fbh[pb - 1], SML{});
                issue_chunk(std::integral_constant<int, g0 + pb>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // every fragment of this K-step is in registers -> its slot may be refilled; the next K-step must have landed
        wait_lgkm<0>();
        wait_vm(p_was_live ? (NSLOT - 3) * ppw + (STAG ? ppw : Q_BEFORE) : 0);
#ifndef MSN_ABL_PG_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        // (not behind a tile's or a K chunk's last step: the epilogue wants the registers -- it requests them when it is done)
        if (!last) req_first(fa[(QA + NP) & 1], fb0[PAR ^ 1], nxt);
        __builtin_amdgcn_sched_barrier(0);
        mult(fa[(QA + NP - 1) & 1], fb0[PAR], SML{});
        issue_chunk(std::integral_constant<int, NG - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        advance();
        c_slot = n_slot;
    };

    // ---- epilogue: lane holds, per (i, j) tile, row m = 32 i + (lane & 31) and columns n = 32 j + 8 b + 4 (lane >> 5) + r
    // ---- epilogue.  The accumulator layout (lane = one row, 4 consecutive columns) is the wrong shape for memory: a 16-byte
    // store per lane touches 32 different 128-byte lines per wave instruction and the address coalescer, not HBM, bounds the
    // epilogue (64 such instructions per wave for a GELU tile: +320 us on a 470-us product).  Every 32 x 32 MFMA tile therefore
    // goes through a per-wave LDS staging area (6 KB behind the ring) and leaves -- or enters, for the matrices an epilogue
    // reads -- in memory order: fp32 as 8 rows x 128 bytes per instruction, planes as whole 1-KB block images.
    // MODE 0: the tile's result.  MODE 1: the tile's result when earlier K chunks left a partial sum in C (added first).
    // MODE 2 / 3: a K chunk's partial sum -> C (first chunk) / C += (later chunks); no bias, no epilogue (fp32 outputs only).
    const unsigned stg = lds0 + (unsigned)(NSLOT * SLOT + wave * STG);
    // fp32 image of a tile: [32 rows][128 B], 16-byte chunk c of row r at position c ^ (r & 7) (conflict-free both ways)
    const unsigned st_acc = stg + (unsigned)(frow * 128);                       // + ((2 b + fhalf) ^ (frow & 7)) * 16
    // chunk b of this lane's 16 accumulator values of a 32 x 32 tile: its row and its 16-byte column chunk (4 columns) in the tile
    auto erow = [&](int b) { return S16 ? 16 * (b >> 1) + r16 : frow; };
    auto ecolq = [&](int b) { return S16 ? 4 * (b & 1) + kg : 2 * b + fhalf; };
    auto acc_addr = [&](int b) { return stg + (unsigned)(erow(b) * 128 + ((ecolq(b) ^ (erow(b) & 7)) * 16)); };
    const int srow = lane >> 3, schunk = lane & 7;                              // memory order: pass q -> row 8 q + srow
    const unsigned st_mem = stg + (unsigned)(srow * 128 + ((schunk ^ srow) * 16));   // + q * 1024   ((8 q + srow) & 7 == srow)
    // (s_nop behind every ds_write_b128: a VALU write to the data registers of a DS store of more than 8 bytes needs a wait
    // state the compiler's hazard recognizer would insert -- it cannot see the store inside the asm; without it a few lanes
    // of a tile came out wrong once in a few launches)
    auto lds_w128 = [&](unsigned addr, const float (&v)[4]) {
        asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"(addr), "v"(*reinterpret_cast<const f32x4*>(v)) : "memory");
    };
    auto lds_r128 = [&](f32x4& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); };
    // a 32 x 32 fp32 tile of a matrix the epilogue READS (residual / saved activation derivative / K-chunk partial):
    // tile_fetch issues its four 16-byte loads per lane in memory order (for EVERY tile of the wave before the first is used:
    // one exposed memory latency per output tile instead of one per MFMA tile), tile_take turns one through the staging
    // area into the accumulator layout a[4 b + r]
    auto tile_fetch = [&](const float* src, int64_t ld, int64_t m0, int n0, f32x4 (&t)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t m = m0 + 8 * q + srow;
            const int n = n0 + 4 * schunk;
            t[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < p.M && n < p.N) t[q] = *reinterpret_cast<const f32x4*>(src + m * ld + n);
        }
    };
    auto tile_take = [&](const f32x4 (&t)[4], float (&a)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_write_b128 %0, %1\n\ts_nop 1" ::"v"(st_mem + q * 1024), "v"(t[q]) : "memory");
        f32x4 r[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) lds_r128(r[b], acc_addr(b));
        // the wait names the registers it guards: the compiler may otherwise move a copy of them above it (it believes the
        // asm that issued the read has already defined them)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * b + e] = r[b][e];
    };
    // accumulator layout -> rows m0.., columns n0.. of `dst` in memory order: stage_chunk(b, 4 values) four times, then flush
    auto stage_chunk = [&](int b, const float (&w)[4]) { lds_w128(acc_addr(b), w); };
    auto tile_flush = [&](float* dst, int64_t ld, int64_t m0, int n0) {
        f32x4 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_r128(r[q], st_mem + q * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t m = m0 + 8 * q + srow;
            const int n = n0 + 4 * schunk;
            if (m < p.M && n < p.N) *reinterpret_cast<f32x4*>(dst + m * ld + n) = r[q];
        }
    };
    auto slab_flush = [&](float* slab, int r0, int c0) {            // the staged tile -> rows r0.., columns c0.. of a 256 x BN slab
        f32x4 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_r128(r[q], st_mem + q * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(slab + (r0 + 8 * q + srow) * BN + c0 + 4 * schunk) = r[q];
    };
    auto store_tile_epi = [&](int tm, int tn, auto epi_, auto mode_) {
        constexpr int epi = decltype(epi_)::value;
        constexpr int MODE = decltype(mode_)::value & 3;
        constexpr bool CS = (decltype(mode_)::value & 4) != 0;    // column sums of the stored values -> p.colpart
        constexpr bool SLAB = (decltype(mode_)::value & 8) != 0;  // a tail unit's raw sums -> its slab (256 x BN, row-major)
        constexpr bool PARTIAL = MODE >= 2, ADDC = (MODE == 1 || MODE == 3) && !OUTP;
        constexpr bool AUXIN = !PARTIAL && (epi == MSN_EPI_RELU_BWD || epi == MSN_EPI_GELU_BWD || epi == MSN_EPI_ADD);
        // everything the epilogue reads, requested up front: the K-chunk partial if there is one, else the aux matrix
        if constexpr (DUAL) {      // the two accumulator sets meet here (one round-to-nearest add); acc2's registers are free from now on
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] += acc2[i][j];
        }
        if constexpr (F16) {       // fp16 planes hold x * 2^e: one exact power of two puts the sums back (before bias / epilogue / partial sums)
            const float os = p.scaleA[0] * p.scaleB[0];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] *= os;
        }
        constexpr bool PRE = ADDC || AUXIN;
        constexpr int PW = (DUAL || MT * NT < 2) ? MT * NT : 2;          // tiles requested ahead (16 registers each)
        f32x4 pre[PRE ? PW : 1][4];
        const float* pre_src = ADDC ? static_cast<const float*>(p.C) : p.aux;
        const int64_t pre_ld = ADDC ? p.ldc : p.ldaux;
        auto pre_fetch = [&](auto t_) {                            // tile t = j * MT + i of this wave -> pre[t % PW]
            constexpr int t = decltype(t_)::value;
            if constexpr (PRE && t < MT * NT)
                tile_fetch(pre_src, pre_ld, (int64_t)tm * BM + wm * (32 * MT) + 32 * (t % MT), tn * BN + wn * (32 * NT) + 32 * (t / MT),
                           pre[t % PW]);
        };
        static_for<0, PW>([&](auto t_) { pre_fetch(t_); });
        static_for<0, NT>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            const int n0 = tn * BN + wn * (32 * NT) + 32 * j;
            float cs[CS ? 16 : 1];                                 // column sums over this wave's rows, accumulator layout
#pragma unroll
            for (int e = 0; e < (CS ? 16 : 1); ++e) cs[e] = 0.f;
            static_for<0, MT>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                constexpr int t = j * MT + i;
                const int64_t m0 = (int64_t)tm * BM + wm * (32 * MT) + 32 * i;
                const int rbk = tm * ARB + wm * MT + i;           // row block of a plane output (it has the rows of A)
                const bool live = SLAB || (n0 < p.N && (OUTP ? rbk < p.rbA : m0 < p.M));      // (wave-uniform)
                if (!live) {
                    pre_fetch(std::integral_constant<int, t + PW>{});
                    return;
                }
                const bool row_ok = m0 + frow < p.M;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = acc[i][j][e];
                if constexpr (ADDC) {
                    float c0[16];
                    tile_take(pre[t % PW], c0);
                    pre_fetch(std::integral_constant<int, t + PW>{});
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += c0[e];
                }
                if (!PARTIAL && p.bias) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int n = n0 + 4 * ecolq(b);
                        if (n < p.N) {
                            const float4 t = *reinterpret_cast<const float4*>(p.bias + n);
                            v[4 * b] += t.x; v[4 * b + 1] += t.y; v[4 * b + 2] += t.z; v[4 * b + 3] += t.w;
                        }
                    }
                }
                if constexpr (!PARTIAL && epi == MSN_EPI_GELU) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float dg[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) gelu_both(v[4 * b + r], v[4 * b + r], dg[r]);
                        if (p.aux) stage_chunk(b, dg);
                    }
                    if (p.aux) tile_flush(p.aux, p.ldaux, m0, n0);
                } else if constexpr (!PARTIAL && epi == MSN_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if constexpr (!PARTIAL && epi != MSN_EPI_NONE) {
                    float a[16];
                    if constexpr (ADDC) {                              // (both a partial and an aux matrix: the aux tile comes now)
                        f32x4 t[4];
                        tile_fetch(p.aux, p.ldaux, m0, n0, t);
                        tile_take(t, a);
                    } else {
                        tile_take(pre[t % PW], a);
                        pre_fetch(std::integral_constant<int, t + PW>{});
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        if constexpr (epi == MSN_EPI_GELU_BWD) v[e] *= a[e];
                        else if constexpr (epi == MSN_EPI_RELU_BWD) v[e] = a[e] > 0.f ? v[e] : 0.f;
                        else v[e] += a[e];                         // MSN_EPI_ADD
                    }
                }
                if constexpr (S16) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (m0 + erow(e >> 2) >= p.M) v[e] = 0.f;
                } else if (!row_ok) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = 0.f;       // padding rows of a plane output are zero
                }
                if constexpr (OUTP) {
                    // planes -> the 2 * NP block images of this tile ([32 rows][16 bf16] each, as they lie in HBM) -> 1-KB stores
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        u16 pl[4][NP];
#pragma unroll
                        for (int r = 0; r < 4; ++r) split_planes<NP>(v[4 * b + r], pl[r]);
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            uint2 o;
                            o.x = pl[0][k] | ((unsigned)pl[1][k] << 16);
                            o.y = pl[2][k] | ((unsigned)pl[3][k] << 16);
                            asm volatile("ds_write_b64 %0, %1" ::"v"(stg + (unsigned)((((ecolq(b) >> 2) * NP + k) * PBLK) + erow(b) * 32 + (ecolq(b) & 3) * 8)), "v"(o) : "memory");
                        }
                    }
                    f32x4 img[2 * NP];
#pragma unroll
                    for (int q = 0; q < 2 * NP; ++q) lds_r128(img[q], stg + (unsigned)(q * PBLK + lane * 16));
                    if constexpr (NP == 3)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(img[0]), "+v"(img[1]), "+v"(img[2]), "+v"(img[3]), "+v"(img[4]), "+v"(img[5])::"memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(img[0]), "+v"(img[1]), "+v"(img[2]), "+v"(img[3])::"memory");
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned char* blk = static_cast<unsigned char*>(p.C) + ((int64_t)rbk * p.cbC + (n0 >> 4)) * (NP * PBLK) + lane * 16;
#pragma unroll
                    for (int q = 0; q < 2 * NP; ++q)
                        if (n0 + 16 * (q / NP) < p.N) *reinterpret_cast<f32x4*>(blk + q * PBLK) = img[q];
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float w[4] = {v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]};
                        stage_chunk(b, w);
                    }
                    if constexpr (SLAB)
                        slab_flush(p.tail_slabs + (int64_t)blockIdx.x * (BM * BN), wm * (32 * MT) + 32 * i, wn * (32 * NT) + 32 * j);
                    else
                        tile_flush(static_cast<float*>(p.C), p.ldc, m0, n0);
                }
                if constexpr (CS) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) cs[e] += v[e];
                }
            });
            if constexpr (CS && S16) {   // chunks b and b ^ 2 hold the two row halves of the same columns; 16 lanes per row half
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float t = cs[e] + cs[e + 8];
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o, 64);
                    cs[e] = t;
                }
                if (r16 == 0) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int n = n0 + 4 * ecolq(b);
                        if (n < p.N)
                            *reinterpret_cast<float4*>(p.colpart + (int64_t)(WM * tm + wm) * p.N + n) =
                                make_float4(cs[4 * b], cs[4 * b + 1], cs[4 * b + 2], cs[4 * b + 3]);
                    }
                }
            } else if constexpr (CS) {   // the 32 lanes sharing lane >> 5 hold the 32 rows of every row tile: xor tree, one lane writes
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float t = cs[e];
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) t += __shfl_xor(t, o, 64);
                    cs[e] = t;
                }
                if (frow == 0) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int n = n0 + 8 * b + 4 * fhalf;
                        if (n < p.N)
                            *reinterpret_cast<float4*>(p.colpart + (int64_t)(WM * tm + wm) * p.N + n) =
                                make_float4(cs[4 * b], cs[4 * b + 1], cs[4 * b + 2], cs[4 * b + 3]);
                    }
                }
            }
        });
    };
    auto store_tile = [&](int tm, int tn, auto mode_) {
        switch (p.epi) {
            case MSN_EPI_RELU: store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_RELU>{}, mode_); break;
            case MSN_EPI_GELU: store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_GELU>{}, mode_); break;
            case MSN_EPI_RELU_BWD: store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_RELU_BWD>{}, mode_); break;
            case MSN_EPI_GELU_BWD: store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_GELU_BWD>{}, mode_); break;
            case MSN_EPI_ADD: store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_ADD>{}, mode_); break;
            default: store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_NONE>{}, mode_); break;
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[i][j][e] = 0.f;
                    if constexpr (DUAL) acc2[i][j][e] = 0.f;
                }
    };

    // ---- start skew: equal tiles keep the workgroups of a launch in lockstep -- all in their K loops (HBM idle), then all in
    // their epilogues (matrix cores idle, HBM write-bound).  Four start phases a quarter tile apart spread the epilogues'
    // stores over the others' K loops; workgroups with one tile fewer than the rest have the slack for it.
    if (p.skew > 0) {
        const int phase = (blockIdx.x >> 3) & 3;
        const int mine = (total - (int)blockIdx.x + G - 1) / G, most = (total + G - 1) / G;
        if (phase && (mine < most || p.skew < 0x40000000)) {
            const long long until = (long long)__builtin_readcyclecounter() + (long long)phase * (p.skew & 0x3fffffff);
            while ((long long)__builtin_readcyclecounter() < until) __builtin_amdgcn_s_sleep(32);
        }
    }
    // ---- prologue: K-steps 0 .. NSLOT - 2 of the stream
#pragma unroll
    for (int s = 0; s < NSLOT - 1; ++s) {
        if (p_live) {
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) issue_q(q);
        }
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    req_first_settled(0u);

    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    int tm, tn;
    // K chunks (fp32 outputs of the DUAL form): the bf16 MFMA's biased accumulate costs ~K^1.5, so a long reduction is cut into
    // chunks of p.chunk_steps K-steps whose partial sums meet in C by round-to-nearest fp32 adds (the lane that wrote a partial
    // is the lane that reads it back: same wave, same address, program order)
    const int csteps = (!OUTP && p.chunk_steps > 0) ? p.chunk_steps : nk;
    int kb, ke;
    for (int idx = 0; seg_of(idx, tm, tn, kb, ke); ++idx) {
        zero_acc();
        if (ke - kb < nk) {                          // a tail unit: raw sums of its K range -> its slab (finishing launch: host)
            for (int k = kb; k < ke; k += 2) {
                step(T0{});
                step(T1{}, k + 2 >= ke);
            }
            wait_lgkm<0>();
            store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_NONE>{}, std::integral_constant<int, 2 + 8>{});
            req_first_settled((unsigned)(c_slot * SLOT));
            continue;
        }
        int k0 = 0;
        for (; k0 + csteps < nk; k0 += csteps) {
            for (int k = 0; k < csteps; k += 2) {
                step(T0{});
                step(T1{}, k + 2 >= csteps);
            }
            wait_lgkm<0>();
            if (k0 == 0) store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_NONE>{}, std::integral_constant<int, 2>{});
            else store_tile_epi(tm, tn, std::integral_constant<int, MSN_EPI_NONE>{}, std::integral_constant<int, 3>{});
            zero_acc();
            req_first_settled((unsigned)(c_slot * SLOT));
        }
        for (int k = k0; k < nk; k += 2) {           // nk is even (plane format)
            step(T0{});
            step(T1{}, k + 2 >= nk);
        }
        wait_lgkm<0>();
        if (k0 != 0) store_tile(tm, tn, std::integral_constant<int, 1>{});
        else if (p.colpart) store_tile(tm, tn, std::integral_constant<int, 4>{});
        else store_tile(tm, tn, std::integral_constant<int, 0>{});
        // the next tile's first fragments (past the workgroup's last tile: whatever the slot holds, into registers nobody uses)
        req_first_settled((unsigned)(c_slot * SLOT));
    }
}

// Tail tiles of an NT product (PgemmArgs::tail_*): C tile = epilogue(sum over its K-segments' slabs, in K order).  One thread
// per float4 of a 256 x 128 tile; epilogues NONE / RELU / ADD (the products whose tails are split).
__global__ __launch_bounds__(256) void pgemm_tail_finish_kernel(const PgemmArgs p, int bn) {
    const int tiles_tail = p.tiles_m * p.tiles_n - p.tail_full;
    const int per_tile = BM * bn / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)tiles_tail * per_tile) return;
    const int tt = (int)(i / per_tile), e = (int)(i % per_tile);
    // the tile's position: the same XCD-aware map as the kernel's
    const int total = p.tiles_m * p.tiles_n, t = p.tail_full + tt;
    const int q = total / 8, r = total % 8, x = t % 8, ii = t / 8;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + ii;
    const int SR = p.super_rows;
    const int sr = bid / (SR * p.tiles_n), j = bid % (SR * p.tiles_n);
    const int rows_sr = min(SR, p.tiles_m - sr * SR);
    const int tm = sr * SR + j % rows_sr, tn = j / rows_sr;
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

// ---------------------------------------------------------------------------------------------------------------
// TN kernel (weight gradients): T[p][q] = sum_m P[m][p] . Q[m][q], both operands plane matrices with the reduction index m
// on the ROWS.  SWAP = false: P = dY (p = n), Q = X (q = k);  SWAP = true: P = X (p = k), Q = dY (q = n) -- the 256-wide side
// of the tile goes to whichever of N, K it divides better; the MFMA operand order is chosen so that a lane always ends up
// with four consecutive k of one n (16-byte stores into C[n][k]).
// K-step = 16 reduction rows = half a row block.  LDS slot = [P: 16 column blocks][NP][512 B] then [Q: BQ / 16][NP][512 B];
// a 512-byte half-block image is [16 rows of m][32 B], its four row quads (4 rows = 128 B) stored at quad position
// Q ^ (column block & 1).  The MFMA wants, per lane, 8 consecutive m of one column: two ds_read_b64_tr_b16 (each hands a
// 4-row x 16-column block, column-major, to a 16-lane group: lane 4 q + p of the group supplies the address of row q,
// bytes 8 p .. 8 p + 7; lane i receives column i).  Banking is per 32 lanes = the two column blocks of a 32-wide MFMA tile:
// they read the same row quad, which the swizzle puts into opposite 128-byte halves of the bank row -> conflict-free.
template <int OFF>
__device__ __forceinline__ void ds_read_tr_o(bf16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

template <int NP, int BQ, bool SWAP, bool DUAL, int WM = 2, int WN = 4, bool F16 = false>
__global__ __launch_bounds__(512, 2) void pgemm_tn_kernel(const PgemmArgs p) {
    static_assert(WM * WN == 8, "eight waves");
    constexpr int PHB = 16 * NP, QHB = (BQ / 16) * NP;       // half-block images per K-step: P side, Q side
    constexpr int PIECES = (PHB + QHB) / 2;                  // 1-KB pieces (two half-block images each)
    constexpr int PP = PHB / 2;
    constexpr int SLOT = PIECES * PBLK;
    constexpr int NSLOT = (4 * SLOT <= 160 * 1024) ? 4 : 3;
    constexpr int MAXQ = (PIECES + 7) / 8;
    constexpr int MT = 8 / WM, NT = BQ / (32 * WN);       // 32 x 32 MFMA tiles per wave: P side x Q side
    constexpr int NG = NP * (NP + 1) / 2;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSLOT * SLOT];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles = p.tiles_m * p.tiles_n;                 // tiles over (P side, Q side)
    // workgroup ids go round-robin to the 8 XCDs: give each XCD a contiguous run of (split, tile) pairs, i.e. the tiles of
    // one reduction range, so that its panels are shared in ONE L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int split = bid / tiles, tile = bid % tiles;
    const int tp = tile / p.tiles_n, tq = tile % p.tiles_n;
    const unsigned char* Pm = SWAP ? p.B : p.A;
    const unsigned char* Qm = SWAP ? p.A : p.B;
    const int cbP = SWAP ? p.cbB : p.cbA, cbQ = SWAP ? p.cbA : p.cbB;    // column blocks of the plane matrices
    const int rb0 = split * p.rb_per_split;
    const int rb1 = min(p.rbA, rb0 + p.rb_per_split);
    const int nk = 2 * (rb1 - rb0);                          // K-steps (>= 2)

    // ---- LDS-DMA: piece id = two consecutive half-block images hb = 2 id + (lane >> 5); image hb of the P side is column
    // block hb / NP, plane hb % NP: in the plane matrix its 1-KB block is block (cb0 * NP + hb) of the row block.
    const int sub = lane >> 5, j = lane & 31;
    const unsigned char* pbase[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int id = min(wave + 8 * q, PIECES - 1);
        const bool isP = id < PP;
        const int hb = 2 * (isP ? id : id - PP) + sub;
        const int cbp = hb / NP;                                            // column block inside the tile
        const int cb0 = isP ? tp * 16 : tq * (BQ / 16);
        const int cbs = isP ? cbP : cbQ;
        const int hbc = min(hb, (cbs - cb0) * NP - 1);                      // past the matrix: clamped, never stored
        const int rpos = j >> 1;                                            // row position in the image
        const int rsrc = (rpos & 3) + 4 * ((rpos >> 2) ^ (cbp & 1));        // source row of that position
        pbase[q] = (isP ? Pm : Qm) + ((int64_t)rb0 * cbs + cb0) * (NP * PBLK) + (int64_t)hbc * PBLK + rsrc * 32 + (j & 1) * 16;
    }
    const int64_t rbstepP = (int64_t)cbP * (NP * PBLK), rbstepQ = (int64_t)cbQ * (NP * PBLK);
    int p_k = 0, p_slot = 0;
    auto issue_q = [&](int q) {
        const int id = wave + 8 * q;
        if (id < PIECES) {
            const int64_t off = (int64_t)(p_k >> 1) * (id < PP ? rbstepP : rbstepQ) + (p_k & 1) * 512;
            __builtin_amdgcn_global_load_lds((gptr_t*)(pbase[q] + off), (lptr_t*)(lds + p_slot * SLOT + id * PBLK), 16, 0, 0);
        }
    };
    auto issue_chunk = [&](auto c_) {
        constexpr int c = decltype(c_)::value;
        if (p_k < nk) {
#pragma unroll
            for (int q = c * MAXQ / NG; q < (c + 1) * MAXQ / NG; ++q) issue_q(q);
        }
    };
    auto advance = [&]() {
        if (p_k < nk) {
            ++p_k;
            p_slot = (p_slot + 1 == NSLOT) ? 0 : p_slot + 1;
        }
    };
    constexpr int Q_BEFORE = (NG - 1) * MAXQ / NG;
    const int ppw = (wave + 8 * (MAXQ - 1) < PIECES) ? MAXQ : MAXQ - 1;

    // ---- fragments: lane -> 16-lane group g16 = (lane >> 4) & 1 (column block of the 32-wide tile), m-octet lane >> 5;
    // inside the group lane 4 q + pp supplies row q, bytes 8 pp.  Read t = 0, 1 of a fragment takes row quad 2 (lane >> 5) + t.
    const int g16 = (lane >> 4) & 1, moct = lane >> 5, rq = (lane >> 2) & 3, pp = lane & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)lds;
    // address of read t: image(cb pair * 2 + g16, plane) * 512 + ((2 moct + t) ^ g16) * 128 + rq * 32 + pp * 8
    const unsigned tr0 = lds0 + (unsigned)(g16 * NP * 512 + ((2 * moct) ^ g16) * 128 + rq * 32 + pp * 8);
    const unsigned tr1 = lds0 + (unsigned)(g16 * NP * 512 + ((2 * moct + 1) ^ g16) * 128 + rq * 32 + pp * 8);
    const unsigned fragP0 = tr0 + (unsigned)(wm * 2 * MT * NP * 512), fragP1 = tr1 + (unsigned)(wm * 2 * MT * NP * 512);      // + (2 i NP + plane) * 512
    const unsigned fragQ0 = tr0 + (unsigned)(PHB * 512 + wn * 2 * NT * NP * 512), fragQ1 = tr1 + (unsigned)(PHB * 512 + wn * 2 * NT * NP * 512);

    f32x16 acc[MT][NT], acc2[DUAL ? MT : 1][DUAL ? NT : 1];        // DUAL: see pgemm_nt_kernel
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[i][jj][e] = 0.f;
                if constexpr (DUAL) acc2[i][jj][e] = 0.f;
            }
    bf16x4 fa[2][MT][2], fb0[2][NT][2], fbh[NP > 1 ? NP - 1 : 1][NT][2];

    auto req_a = [&](bf16x4 (&dst)[MT][2], unsigned slot_off, auto pl_) {
        constexpr int pl = decltype(pl_)::value;
        static_for<0, MT>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            ds_read_tr_o<(2 * i * NP + pl) * 512>(dst[i][0], fragP0 + slot_off);
            ds_read_tr_o<(2 * i * NP + pl) * 512>(dst[i][1], fragP1 + slot_off);
        });
    };
    auto req_b = [&](bf16x4 (&dst)[NT][2], unsigned slot_off, auto pl_) {
        constexpr int pl = decltype(pl_)::value;
        static_for<0, NT>([&](auto j_) {
            constexpr int jj = decltype(j_)::value;
            ds_read_tr_o<(2 * jj * NP + pl) * 512>(dst[jj][0], fragQ0 + slot_off);
            ds_read_tr_o<(2 * jj * NP + pl) * 512>(dst[jj][1], fragQ1 + slot_off);
        });
    };
    auto mfma1 = [&](const bf16x8& av, const bf16x8& bv, const f32x16& c) -> f32x16 {
        if constexpr (F16) {
            const f16x8 ha = __builtin_bit_cast(f16x8, av), hb = __builtin_bit_cast(f16x8, bv);
            if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, c, 0, 0, 0);
        } else {
            if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, c, 0, 0, 0);
        }
    };
    // DUAL: zero-accumulator MFMA + round-to-nearest fold of the p0.q0 product (see pgemm_nt_kernel)
    f32x16 tbig[2];
    auto fold = [&](auto v_) {
        constexpr int v = decltype(v_)::value;
        acc[v / NT][v % NT] += tbig[v & 1];
        asm volatile("" : "+v"(acc[v / NT][v % NT]));
    };
    auto mult_big = [&](const bf16x4 (&a)[MT][2], const bf16x4 (&b)[NT][2]) {
        static_for<0, MT * NT>([&](auto u_) {
            constexpr int u = decltype(u_)::value;
            if constexpr (u >= 2) fold(std::integral_constant<int, u - 2>{});
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            const bf16x8 av = __builtin_shufflevector(a[u / NT][0], a[u / NT][1], 0, 1, 2, 3, 4, 5, 6, 7);
            const bf16x8 bv = __builtin_shufflevector(b[u % NT][0], b[u % NT][1], 0, 1, 2, 3, 4, 5, 6, 7);
            tbig[u & 1] = mfma1(av, bv, z);
            asm volatile("" : "+v"(tbig[u & 1]));
        });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fold_pending = [&]() {
        fold(std::integral_constant<int, MT * NT - 2>{});
        fold(std::integral_constant<int, MT * NT - 1>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mult = [&](const bf16x4 (&a)[MT][2], const bf16x4 (&b)[NT][2], auto small_) {
        constexpr bool SMALL = decltype(small_)::value && DUAL;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
                const bf16x8 av = __builtin_shufflevector(a[i][0], a[i][1], 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8 bv = __builtin_shufflevector(b[jj][0], b[jj][1], 0, 1, 2, 3, 4, 5, 6, 7);
                if constexpr (SMALL) acc2[i][jj] = mfma1(av, bv, acc2[i][jj]);
                else acc[i][jj] = mfma1(av, bv, acc[i][jj]);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    using BIG = std::integral_constant<bool, false>;
    using SML = std::integral_constant<bool, true>;

    int c_slot = 0;
    // One K-step.  On entry outstanding LDS reads, oldest first: P plane 0 (2 MT), Q plane 0 (2 NT); the other Q planes and
    // the later P planes are requested one product ahead (lgkmcnt counts at most 15 reads).
    auto step = [&](auto par_) {
        constexpr int PAR = decltype(par_)::value;
        constexpr int QA = (NP & 1) ? PAR : 0;
        const unsigned cur = (unsigned)(c_slot * SLOT);
        const int n_slot = (c_slot + 1 == NSLOT) ? 0 : c_slot + 1;
        const unsigned nxt = (unsigned)(n_slot * SLOT);
        const bool p_was_live = p_k < nk;
        static_for<0, NP>([&](auto pb_) {
            constexpr int pb = decltype(pb_)::value;
            if constexpr (pb + 1 < NP) {
                if constexpr (pb == 0 && 2 * MT + 4 * NT > 15) wait_lgkm<15 - 2 * NT>();    // never more than 15 reads in flight
                req_b(fbh[pb], cur, std::integral_constant<int, pb + 1>{});
                wait_lgkm<2 * NT>();
            } else {
                req_a(fa[QA ^ 1], cur, std::integral_constant<int, 1>{});
                wait_lgkm<2 * MT>();
            }
            if constexpr (pb == 0) {
                if constexpr (DUAL) mult_big(fa[QA], fb0[PAR]);
                else mult(fa[QA], fb0[PAR], BIG{});
            } else {
                mult(fa[QA], fbh[pb - 1], SML{});
                if constexpr (DUAL && pb == 1) fold_pending();
            }
            issue_chunk(std::integral_constant<int, pb>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<1, NP - 1>([&](auto pa_) {
            constexpr int pa = decltype(pa_)::value;
            constexpr int g0 = pa * NP - pa * (pa - 1) / 2;
            req_a(fa[(QA + pa + 1) & 1], cur, std::integral_constant<int, pa + 1>{});
            wait_lgkm<2 * MT>();
            static_for<0, NP - pa>([&](auto pb_) {
                constexpr int pb = decltype(pb_)::value;
                if constexpr (pb == 0) mult(fa[(QA + pa) & 1], fb0[PAR], SML{});
                else mult(fa[(QA + pa) & 1], fbh[pb - 1], SML{});
                issue_chunk(std::integral_constant<int, g0 + pb>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        wait_lgkm<0>();
        wait_vm(p_was_live ? (NSLOT - 3) * ppw + Q_BEFORE : 0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        req_a(fa[(QA + NP) & 1], nxt, std::integral_constant<int, 0>{});
        req_b(fb0[PAR ^ 1], nxt, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mult(fa[(QA + NP - 1) & 1], fb0[PAR], SML{});
        issue_chunk(std::integral_constant<int, NG - 1>{});
        __builtin_amdgcn_sched_barrier(0);
        advance();
        c_slot = n_slot;
    };

    // ---- prologue
#pragma unroll
    for (int s_ = 0; s_ < NSLOT - 1; ++s_) {
        if (p_k < nk) {
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) issue_q(q);
        }
        advance();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    req_a(fa[0], 0u, std::integral_constant<int, 0>{});
    req_b(fb0[0], 0u, std::integral_constant<int, 0>{});
    for (int k = 0; k < nk; k += 2) {
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
    }
    wait_lgkm<0>();

    // ---- epilogue: C[n][k] (or this split's slab)
    float* out = p.splits > 1 ? p.slabs + (int64_t)split * p.N * p.K : static_cast<float*>(p.C);
    const int64_t ldo = p.splits > 1 ? p.K : p.ldc;
    const int l31 = lane & 31, h4 = 4 * (lane >> 5);
    float os = 1.f;
    if constexpr (F16) os = p.scaleA[0] * p.scaleB[0];           // see pgemm_nt_kernel
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int n, k;
                if constexpr (SWAP) {     // D[p][q]: lane holds q = l31 (n), p = 8 b + h4 + r (k)
                    k = tp * 256 + wm * (32 * MT) + 32 * i + 8 * b + h4;
                    n = tq * BQ + wn * (32 * NT) + 32 * jj + l31;
                } else {                  // D'[q][p]: lane holds p = l31 (n), q = 8 b + h4 + r (k)
                    n = tp * 256 + wm * (32 * MT) + 32 * i + l31;
                    k = tq * BQ + wn * (32 * NT) + 32 * jj + 8 * b + h4;
                }
                if (n >= p.N || k >= p.K) continue;                  // K % 4 == 0 (host)
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[i][jj][4 * b + r];
                    if constexpr (DUAL) v[r] += acc2[i][jj][4 * b + r];
                    if constexpr (F16) v[r] *= os;
                }
                *reinterpret_cast<float4*>(out + (int64_t)n * ldo + k) = make_float4(v[0], v[1], v[2], v[3]);
            }
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

static int g_pgemm_bn = 0;          // 0 = planned, 128 / 256 forced (measurements)
static int g_pgemm_skew = 0;        // NT start skew (shader cycles per phase; experiments)
extern "C" int msn_set_pgemm_skew(int cycles) {
    g_pgemm_skew = cycles;
    return MSN_OK;
}
static int g_pgemm_chunk = 0;       // K-steps per chunk of the 3-plane NT kernel (0 = whole reduction; experiments)
static int g_pgemm_variant = 1;     // 3-plane NT kernel: 0 = 2 x 4 waves, 1 = 4 x 2 (default), 2 = 4 x 2 on v_mfma_f32_16x16x32_bf16 (plane pairs along k)
extern "C" int msn_set_pgemm_variant(int v) {
    MSN_REQUIRE(v >= 0 && v % 1000 <= 2 && v / 1000 <= 1000, "msn_set_pgemm_variant: 0, 1 or 2 (+ 1000 * K-steps per chunk)");
    g_pgemm_chunk = v / 1000;
    v %= 1000;
    g_pgemm_variant = v;
    return MSN_OK;
}
extern "C" int msn_set_pgemm_tile_n(int bn) {
    MSN_REQUIRE(bn == 0 || bn == 128, "msn_set_pgemm_tile_n: 0 or 128 (the 256-wide tiles were slower on every shape and are no longer built)");
    g_pgemm_bn = bn;
    return MSN_OK;
}

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
        const int cs = g_pgemm_chunk > 0 ? g_pgemm_chunk : 32;         // K-steps per chunk (msn_set_pgemm_variant: experiments)
        if ((planes == 3 || f16) && !c_planes && a.cbA > std::max(cs, 48)) {
            const int nch = (int)cdiv(a.cbA, cs);
            a.chunk_steps = 2 * (int)cdiv(a.cbA, 2 * nch);
        }
    }
    a.skew = g_pgemm_skew;
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
        a.colsum_rows = g_pgemm_variant == 0 ? 2 : 4;
        // (the staggered-DMA instantiations -- variants 3 / 4 of the measurements in profiles/r04_pgemm_variants.txt -- are not
        // built any more: nothing in the step, +9 % on one shape in isolation; STAG stays in the kernel source)
        if (g_pgemm_variant == 0) launch_nt<3, 128, true>(a, c_planes != 0, grid, st);
        else if (g_pgemm_variant == 2 && c_planes) hipLaunchKernelGGL((pgemm_nt_kernel<3, 128, true, true, 4, 2, false, false, true>), dim3((unsigned)grid), dim3(512), 0, st, a);
        else if (g_pgemm_variant == 2) hipLaunchKernelGGL((pgemm_nt_kernel<3, 128, false, true, 4, 2, false, false, true>), dim3((unsigned)grid), dim3(512), 0, st, a);
        else launch_nt<3, 128, true, 4, 2>(a, c_planes != 0, grid, st);
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

template <int NP, int BQ, bool DUAL, int WM = 2, int WN = 4>
static void launch_tn(const PgemmArgs& a, bool swap, int grid, hipStream_t st) {
    if (swap) hipLaunchKernelGGL((pgemm_tn_kernel<NP, BQ, true, DUAL, WM, WN>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((pgemm_tn_kernel<NP, BQ, false, DUAL, WM, WN>), dim3((unsigned)grid), dim3(512), 0, st, a);
}

static int pgemm_tn_impl(int64_t M, int N, int K, int planes, const void* A, const void* B, float* C, int64_t ldc, void* ws,
                         size_t ws_bytes, msn_stream_t stream, const float* scaleA, const float* scaleB) {
    const bool f16 = scaleA != nullptr;
    MSN_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "msn_pgemm_tn: empty operand");
    MSN_REQUIRE(!f16 || (scaleB && planes == 2), "msn_pgemm_tn_f16: two fp16 planes per operand, both scales");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_pgemm_tn: planes must be 2 or 3 (got %d)", planes);
    MSN_REQUIRE(K % 4 == 0 && ldc >= K && ldc % 4 == 0 && aligned16p(C) && aligned16p(A) && aligned16p(B),
                "msn_pgemm_tn: K and ldc must be multiples of 4, operands 16-byte aligned");
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
    if (planes == 3) launch_tn<3, 128, true>(a, t.swap, grid, st);
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
