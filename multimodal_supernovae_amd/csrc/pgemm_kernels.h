// The two hand-scheduled plane GEMM kernels (templates) of csrc/pgemm.hip, in a header so that their instantiations compile in
// two translation units side by side (pgemm.hip: the default forms; pgemm_alt2.hip:
// the 2-plane and the fp16 forms) -- one unit took 2.5 minutes.  See pgemm.hip for the format and the entry points.
#pragma once
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

// Cache policy of the NT epilogue's FINAL stores (buffer aux bits; diagnostic builds: 2 = nt).  Measured in round 6 (profiles/
// r06_experiments_tried.txt, item 2): nt on the final stores is -3 % on the K = 384 launches in isolation (+1 % when K-chunk partial
// sums, which are re-read within microseconds, go out nt as well) and NOTHING on the training step (64.60 vs 64.59 ms): the next
// kernel reads what this one wrote.  Default policy kept.
#ifndef MSN_PG_ST_AUX
#define MSN_PG_ST_AUX 0
#endif
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

template <int OFF>
__device__ __forceinline__ void ds_write64_o(unsigned addr, uint2 v) {
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// x -> NP planes (u16 each).  Each residual x - p0 (- p1) is exact in fp32.
// two values -> NP words (low half: a's plane, high half: b's), the same roundings as split_planes: one v_cvt_pk_bf16_f32 per
// plane and pair, the two planes' values back out of the word by a shift and a mask (11 instructions per pair for three planes;
// split_planes element by element and the packing afterwards cost 14 per ELEMENT in the plane-output epilogue)
template <int NP>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&pk)[NP]) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const f32x2_t ab = {a, b};
        pk[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bf16x2_t));
        if (k + 1 < NP) {
            a -= __uint_as_float(pk[k] << 16);                    // exact
            b -= __uint_as_float(pk[k] & 0xffff0000u);
        }
    }
}
template <int NP>
__device__ __forceinline__ void split_planes(float x, u16 (&pl)[NP]) {
    float r = x;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        pl[k] = f2bf(r);
        r -= bf2f(pl[k]);
    }
}

// The NT kernel's tile walk (also used by its tail-finishing launch): tile ids are dealt round-robin to the 8 XCDs; each XCD gets a
// contiguous run of the walk, which covers the tile grid in super-rows (SR tile-rows x every tile column, row-fastest) so that
// the workgroups of an XCD share A row panels and the weight planes in one L2.
// (Round 6 tried COLUMN GROUPS on top -- the walk covering one group of tile columns after the other, so that an XCD re-reads only
//  its group's weight planes; commit 03e790c has the code and its test -- and measured nothing: a round of an XCD's 32 workgroups
//  fetches >= 4.7 MB of A panels + weight planes whatever its shape, more than the 4-MB L2, so nothing survives from one round to
//  the next; profiles/r06_experiments_tried.txt, item 1.)
__device__ __forceinline__ void nt_locate(const PgemmArgs& p, int t, int& tm_, int& tn_) {
    const int total = p.tiles_m * p.tiles_n;
    const int q = total / 8, r = total % 8, x = t % 8, i = t / 8;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    const int SR = p.super_rows;
    const int sr = bid / (SR * p.tiles_n), j = bid % (SR * p.tiles_n);
    const int rows_sr = min(SR, p.tiles_m - sr * SR);
    tm_ = sr * SR + j % rows_sr;
    tn_ = j / rows_sr;
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
template <int NP, int BN, bool OUTP, bool DUAL, int WM = 2, int WN = 4, bool STAG = false, bool F16 = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) / 4) void pgemm_nt_kernel(const PgemmArgs p) {
    static_assert(!F16 || !OUTP, "fp16 planes: fp32 results only");
    constexpr int NW = WM * WN;
    constexpr int ARB = BM / 32, BRB = BN / 32, AP = ARB * NP, BP = BRB * NP, PIECES = AP + BP;
    constexpr int SLOT = PIECES * PBLK;
    constexpr int STG = 6 * 1024;                    // epilogue staging per wave (one 32 x 32 tile: 4 KB fp32 / 6 plane images)
    constexpr int NSLOT = (4 * SLOT + NW * STG <= 160 * 1024) ? 4 : 3;
    static_assert(NSLOT * SLOT + NW * STG <= 160 * 1024, "LDS: ring + staging");
    constexpr int MAXQ = (PIECES + NW - 1) / NW;     // pieces per wave and K-step (the last one only for the low waves)
    constexpr int MT = 8 / WM, NT = BN / (32 * WN);  // 32 x 32 MFMA tiles per wave: rows x columns
    constexpr int NG = NP * (NP + 1) / 2;            // MFMA groups (plane products) per K-step
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSLOT * SLOT + NW * STG];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int G = gridDim.x;
    const int total = p.tiles_m * p.tiles_n;
    const int nk = p.cbA;                            // K-steps per tile

    // tile order: nt_locate above
    auto locate = [&](int t, int& tm_, int& tn_) { nt_locate(p, t, tm_, tn_); };
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
    // The sources go through BUFFER descriptors: one per operand, in scalar registers, based at the first row block of the
    // producer's tile; the lane's 16 bytes inside a 1-KB block are the vector offset (constant for the kernel), the piece and the
    // K-step a scalar offset.  (With a 64-bit pointer per piece every DMA instruction cost a v_lshl_add_u64 -- seven vector
    // instructions per K-step between the MFMAs, which one SIMD issues instead of, not beside, them.)
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    // (the descriptor's inputs pass through v_readfirstlane: they ARE wave-uniform, but hipcc does not prove it for values that
    //  went through the tile walk, keeps the descriptor in vector registers and wraps EVERY buffer instruction in a waterfall
    //  loop -- 1 600 v_readfirstlane in the listing of the first attempt)
    auto uniform_rsrc = [&](const void* base, int64_t bytes) -> rsrc_t {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)bytes);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, nb, 0x00020000);
    };
    const int lane_src = (((lane >> 1) * 2) + ((lane & 1) ^ ((lane >> 4) & 1))) * 16;
    // (kept as two scalar halves and made into a descriptor by the builtin AT the instruction, the scalar offset through a local:
    //  with `poff[q] + ...` written as the builtin's argument the host pass of hipcc 7.2 drops the kernel's explicit instantiations
    //  without a diagnostic -- no stub and no fat binary in the object, an undefined symbol when the library is loaded)
    unsigned srcA_lo = 0, srcA_hi = 0, srcB_lo = 0, srcB_hi = 0;
    auto halves = [&](const unsigned char* base, unsigned& lo, unsigned& hi) {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    };
    int poff[MAXQ];                                  // byte offset of piece q at K-step 0 behind its descriptor's base (wave-uniform)
    int p_idx = 0, p_k = 0, p_ke = 0, p_slot = 0, p_tm = 0, p_tn = 0;
    bool p_live = seg_of(0, p_tm, p_tn, p_k, p_ke);
    auto set_bases = [&]() {
        const int rbA0 = min(p_tm * ARB, p.rbA - 1), rbB0 = min(p_tn * BRB, p.rbB - 1);
        halves(p.A + (int64_t)rbA0 * nk * (NP * PBLK), srcA_lo, srcA_hi);
        halves(p.B + (int64_t)rbB0 * nk * (NP * PBLK), srcB_lo, srcB_hi);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int id = wave + NW * q;
            if (id < AP) {
                const int rb = min(p_tm * ARB + id / NP, p.rbA - 1);            // rows past the edge: clamped, never stored
                poff[q] = __builtin_amdgcn_readfirstlane(((rb - rbA0) * nk * NP + id % NP) * PBLK);
            } else {
                const int id2 = min(id, PIECES - 1) - AP;
                const int rb = min(p_tn * BRB + id2 / NP, p.rbB - 1);
                poff[q] = __builtin_amdgcn_readfirstlane(((rb - rbB0) * nk * NP + id2 % NP) * PBLK);
            }
        }
    };
    if (p_live) set_bases();
    auto issue_q = [&](int q) {                      // piece q of the producer's current K-step
#ifdef MSN_ABL_PG_NODMA                      // diagnostic build (tools/microbench/build_ablate.sh): no operand traffic at all
        return;
#endif
        const int id = wave + NW * q;
        const bool fromA = (AP % NW == 0) ? (q < AP / NW) : (id < AP);       // (a compile-time choice in the unrolled issue loops)
#ifdef MSN_ABL_PG_NODMA_A                    // diagnostic build: the A operand's pieces are not fetched (upper bound of an A operand
        if (fromA) return;                   // that bypasses LDS; the counted waits become optimistic)
#endif
#ifdef MSN_ABL_PG_SAMEK                      // diagnostic build: every K-step fetches the tile's FIRST one (cache hits: what the ring's
        const int soff = poff[q];            // sources cost beyond their instructions and LDS writes; results are garbage)
#else
        const int soff = poff[q] + p_k * (NP * PBLK);
#endif
        const uint64_t src = ((uint64_t)(fromA ? srcA_hi : srcB_hi) << 32) | (fromA ? srcA_lo : srcB_lo);
#ifndef MSN_PG_A_AUX                         // diagnostic builds: cache policy of the A operand's pieces (2 = nt: streamed, first to leave the L2)
#define MSN_PG_A_AUX 0
#endif
        if (id < PIECES) {
            if (fromA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(src), (short)0, 0x7fffffff, 0x00020000),
                                                         (lptr_t*)(lds + p_slot * SLOT + id * PBLK), 16, lane_src, soff, 0, MSN_PG_A_AUX);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(src), (short)0, 0x7fffffff, 0x00020000),
                                                         (lptr_t*)(lds + p_slot * SLOT + id * PBLK), 16, lane_src, soff, 0, 0);
        }
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
    // (A 16 x 16 x 32 form of this loop -- the 32 k of one instruction = the same 16 columns of TWO planes, three instructions per
    //  16 x 16 tile and K-step -- was built in round 4 and rebuilt with its cross-step prefetch in round 5: all three loop forms
    //  within 1.5 % of each other, profiles/r05_experiments_tried.txt item 10; removed in round 6 with its switch.)
    auto req_a = [&](bf16x8 (&dst)[MT], unsigned slot_off, auto pl_) {
        constexpr int pl = decltype(pl_)::value;
#if defined(MSN_ABL_PG_NOFRAG) || defined(MSN_ABL_PG_NOFRAG_A)   // diagnostic build: no fragment reads (the MFMAs multiply whatever
#pragma unroll                                                  // the registers hold); _A: none of the A operand only
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
    // The epilogue's lane constants (staging addresses, row / chunk of the lane in the two layouts) are derived from an OPAQUE copy
    // of the lane id at the start of every tile's epilogue (epi_lane): computed once at kernel start they were ~30 registers that
    // live across the K loop -- where the two accumulator sets and the fragments fill the file -- and the allocator spilled them
    // (scratch stores in the prologue, reloads in every epilogue form).  Re-deriving them costs ~20 vector instructions per tile.
    int eln = lane;
    int frow_e = 0, fhalf_e = 0, srow = 0, schunk = 0;
    unsigned stg = 0, st_mem = 0;
    auto epi_lane = [&]() {
        asm volatile("" : "+v"(eln));
        frow_e = eln & 31, fhalf_e = eln >> 5;
        srow = eln >> 3, schunk = eln & 7;                                          // memory order: pass q -> row 8 q + srow
        stg = lds0 + (unsigned)(NSLOT * SLOT + wave * STG);
        // fp32 image of a tile: [32 rows][128 B], 16-byte chunk c of row r at position c ^ (r & 7) (conflict-free both ways)
        st_mem = stg + (unsigned)(srow * 128 + ((schunk ^ srow) * 16));              // + q * 1024   ((8 q + srow) & 7 == srow)
    };
    // chunk b of this lane's 16 accumulator values of a 32 x 32 tile: its row and its 16-byte column chunk (4 columns) in the tile
    auto erow = [&](int) { return frow_e; };
    auto ecolq = [&](int b) { return 2 * b + fhalf_e; };
    auto acc_addr = [&](int b) { return stg + (unsigned)(erow(b) * 128 + ((ecolq(b) ^ (erow(b) & 7)) * 16)); };
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
    // Global accesses of the epilogue go through BUFFER instructions: one descriptor per matrix and tile row (scalar registers),
    // a 32-bit byte offset per lane.  With 64-bit per-lane pointers the address arithmetic of the six epilogue forms held ~60
    // registers in 2-register pairs and the allocator spilled them around the K loop (352 bytes of scratch per lane).  The
    // descriptor covers exactly the tile row's rows that exist: rows past M read zeros / are not stored without a compare, and a
    // lane whose columns lie past N is given an offset beyond every descriptor (row strides < 2^20 elements: host).
    constexpr unsigned OOB = 0x40000000u;
    auto rows_rsrc = [&](const float* base, int64_t ld, int tm_) -> rsrc_t {      // fp32 matrix, the 256 rows of tile row tm_
        const int64_t rows = min<int64_t>(max<int64_t>(p.M - (int64_t)tm_ * BM, 0), BM);
        return uniform_rsrc(base + (int64_t)tm_ * BM * ld, rows * ld * 4);
    };
    // r0 = first row of the 32 x 32 tile inside the tile row, n0 = its first column
    auto tile_fetch = [&](rsrc_t rs, int64_t ld, int r0, int n0, f32x4 (&t)[4]) {
        const int n = n0 + 4 * schunk;
        const unsigned lo = n < p.N ? (unsigned)((srow * (int)ld + n) * 4) : OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            t[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lo + (unsigned)((r0 + 8 * q) * (int)ld * 4)), 0, 0));
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
    auto tile_flush = [&](rsrc_t rs, int64_t ld, int r0, int n0, auto final_) {
        constexpr int AUX = decltype(final_)::value ? MSN_PG_ST_AUX : 0;
        f32x4 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_r128(r[q], st_mem + q * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
        const int n = n0 + 4 * schunk;
        const unsigned lo = n < p.N ? (unsigned)((srow * (int)ld + n) * 4) : OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[q]), rs, (int)(lo + (unsigned)((r0 + 8 * q) * (int)ld * 4)), 0, AUX);
    };
    auto slab_flush = [&](rsrc_t rs, int r0, int c0) {              // the staged tile -> rows r0.., columns c0.. of a 256 x BN slab
        f32x4 r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_r128(r[q], st_mem + q * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3])::"memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r[q]), rs, ((r0 + 8 * q + srow) * BN + c0 + 4 * schunk) * 4, 0, 0);
    };
    auto store_tile_epi = [&](int tm, int tn, auto epi_, auto mode_) {
        constexpr int epi = decltype(epi_)::value;
        constexpr int MODE = decltype(mode_)::value & 3;
        constexpr bool CS = (decltype(mode_)::value & 4) != 0;    // column sums of the stored values -> p.colpart
        constexpr bool SLAB = (decltype(mode_)::value & 8) != 0;  // a tail unit's raw sums -> its slab (256 x BN, row-major)
        constexpr bool PARTIAL = MODE >= 2, ADDC = (MODE == 1 || MODE == 3) && !OUTP;
        constexpr bool AUXIN = !PARTIAL && (epi == MSN_EPI_RELU_BWD || epi == MSN_EPI_GELU_BWD || epi == MSN_EPI_ADD);
        epi_lane();
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
        // tiles requested ahead (16 registers each): all four of the wave where the merged accumulator set has freed the room;
        // a plane output also holds six block images per tile (24 registers) -- two ahead, or the epilogue itself spills
        constexpr int PW = (MT * NT < 2) ? MT * NT : ((DUAL && !OUTP) ? MT * NT : 1);
        f32x4 pre[PRE ? PW : 1][4];
        // descriptors of this tile row (scalar registers): C as an fp32 matrix or as planes, the aux matrix, bias, slab, column sums
        const rsrc_t rsC = OUTP ? uniform_rsrc(static_cast<unsigned char*>(p.C) + (int64_t)tm * ARB * p.cbC * (NP * PBLK),
                                               min(max(p.rbA - tm * ARB, 0), ARB) * (int64_t)p.cbC * (NP * PBLK))
                                : rows_rsrc(static_cast<const float*>(p.C), p.ldc, tm);
        const rsrc_t rsAux = rows_rsrc(p.aux, p.ldaux, tm);
        const rsrc_t rsBias = uniform_rsrc(p.bias, p.bias ? p.N * 4 : 0);
        const rsrc_t pre_rs = ADDC ? rsC : rsAux;
        const int64_t pre_ld = ADDC ? p.ldc : p.ldaux;
        auto pre_fetch = [&](auto t_) {                            // tile t = j * MT + i of this wave -> pre[t % PW]
            constexpr int t = decltype(t_)::value;
            if constexpr (PRE && t < MT * NT)
                tile_fetch(pre_rs, pre_ld, wm * (32 * MT) + 32 * (t % MT), tn * BN + wn * (32 * NT) + 32 * (t / MT), pre[t % PW]);
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
                const bool row_ok = m0 + frow_e < p.M;
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
                    for (int b = 0; b < 4; ++b) {                   // (columns past N: beyond the descriptor, zeros)
                        const f32x4 t = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, (n0 + 4 * ecolq(b)) * 4, 0, 0));
                        v[4 * b] += t[0]; v[4 * b + 1] += t[1]; v[4 * b + 2] += t[2]; v[4 * b + 3] += t[3];
                    }
                }
                if constexpr (!PARTIAL && epi == MSN_EPI_GELU) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        float dg[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#ifdef MSN_ABL_PG_NOGELU                     // diagnostic build (plane-output ablations, profiles/r06_pgemm_planeout_ablation.txt): no GELU arithmetic
                            dg[r] = v[4 * b + r];
#else
                            gelu_both(v[4 * b + r], v[4 * b + r], dg[r]);
#endif
                        }
#ifndef MSN_ABL_PG_NOAUXST                   // diagnostic build: the saved GELU' matrix is neither staged nor stored
                        if (p.aux) stage_chunk(b, dg);
#endif
                    }
#ifndef MSN_ABL_PG_NOAUXST
                    if (p.aux) tile_flush(rsAux, p.ldaux, wm * (32 * MT) + 32 * i, n0, std::true_type{});
#endif
                } else if constexpr (!PARTIAL && epi == MSN_EPI_RELU) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if constexpr (!PARTIAL && epi != MSN_EPI_NONE) {
                    float a[16];
                    if constexpr (ADDC) {                              // (both a partial and an aux matrix: the aux tile comes now)
                        f32x4 t[4];
                        tile_fetch(rsAux, p.ldaux, wm * (32 * MT) + 32 * i, n0, t);
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
                if (!row_ok) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] = 0.f;       // padding rows of a plane output are zero
                }
                if constexpr (OUTP) {
                    // planes -> the 2 * NP block images of this tile ([32 rows][16 bf16] each, as they lie in HBM) -> 1-KB stores
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        unsigned lo[NP], hi[NP];
#ifdef MSN_ABL_PG_NOSPLIT                    // diagnostic build: no split arithmetic (the raw words go out as "planes")
#pragma unroll
                        for (int k = 0; k < NP; ++k) lo[k] = __float_as_uint(v[4 * b + (k & 1)]), hi[k] = __float_as_uint(v[4 * b + 2 + (k & 1)]);
#else
                        split_pair<NP>(v[4 * b], v[4 * b + 1], lo);
                        split_pair<NP>(v[4 * b + 2], v[4 * b + 3], hi);
#endif
                        unsigned at = stg + (unsigned)(((ecolq(b) >> 2) * NP) * PBLK + erow(b) * 32 + (ecolq(b) & 3) * 8);
                        static_for<0, NP>([&](auto k_) {
                            constexpr int k = decltype(k_)::value;
                            uint2 o;
                            o.x = lo[k];
                            o.y = hi[k];
                            ds_write64_o<k * PBLK>(at, o);
                        });
                    }
                    f32x4 img[2 * NP];
#pragma unroll
                    for (int q = 0; q < 2 * NP; ++q) lds_r128(img[q], stg + (unsigned)(q * PBLK + eln * 16));
                    if constexpr (NP == 3)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(img[0]), "+v"(img[1]), "+v"(img[2]), "+v"(img[3]), "+v"(img[4]), "+v"(img[5])::"memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(img[0]), "+v"(img[1]), "+v"(img[2]), "+v"(img[3])::"memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const int blk = ((wm * MT + i) * p.cbC + (n0 >> 4)) * (NP * PBLK) + eln * 16;      // inside the tile row's 8 row blocks
#ifndef MSN_ABL_PG_NOPSTORE                  // diagnostic build: the plane images are built but not stored
#pragma unroll
                    for (int q = 0; q < 2 * NP; ++q)
                        if (n0 + 16 * (q / NP) < 16 * p.cbC)      // incl. the zero padding block of N % 32 == 16
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, img[q]), rsC, blk + q * PBLK, 0, MSN_PG_ST_AUX);
#else
                    asm volatile("" ::"v"(img[0]), "v"(img[1]), "v"(img[2]), "v"(img[3]));
                    (void)blk;
#endif
                } else {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const float w[4] = {v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]};
                        stage_chunk(b, w);
                    }
                    if constexpr (SLAB)
                        slab_flush(uniform_rsrc(p.tail_slabs + (int64_t)blockIdx.x * (BM * BN), BM * BN * 4), wm * (32 * MT) + 32 * i,
                                   wn * (32 * NT) + 32 * j);
                    else
                        tile_flush(rsC, p.ldc, wm * (32 * MT) + 32 * i, n0, std::integral_constant<bool, !PARTIAL>{});
                }
                if constexpr (CS) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) cs[e] += v[e];
                }
            });
            if constexpr (CS) {   // the 32 lanes sharing lane >> 5 hold the 32 rows of every row tile: xor tree, one lane writes
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float t = cs[e];
#pragma unroll
                    for (int o = 1; o < 32; o <<= 1) t += __shfl_xor(t, o, 64);
                    cs[e] = t;
                }
                if (frow_e == 0) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int n = n0 + 8 * b + 4 * fhalf_e;
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

template <int NP, int BQ, bool SWAP, bool DUAL, int WM = 2, int WN = 4, bool F16 = false, int FOLDN = 1>
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
    // Sources through BUFFER descriptors (scalar registers: one per operand, based at this split's first row block and this
    // tile's first column block) + a 32-bit byte offset per lane + a scalar offset per K-step: no 64-bit pointers in vector
    // registers (the five pointer pairs of the older form were what the register file could not hold beside the temporaries
    // of FOLDN = 2; the descriptors alone are worth 4 % of the kernel).
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    auto uniform_rsrc = [&](const unsigned char* base, int64_t bytes) -> rsrc_t {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)min<int64_t>(bytes, 0x7fffffff));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, nb, 0x00020000);
    };
    const int64_t rbstepP = (int64_t)cbP * (NP * PBLK), rbstepQ = (int64_t)cbQ * (NP * PBLK);
    const rsrc_t rsP = uniform_rsrc(Pm + ((int64_t)rb0 * cbP + tp * 16) * (NP * PBLK), (int64_t)(rb1 - rb0) * rbstepP);
    const rsrc_t rsQ = uniform_rsrc(Qm + ((int64_t)rb0 * cbQ + tq * (BQ / 16)) * (NP * PBLK), (int64_t)(rb1 - rb0) * rbstepQ);
    unsigned plane_off[MAXQ];                        // the lane's byte offset of piece q inside a row block
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
        plane_off[q] = (unsigned)(hbc * PBLK + rsrc * 32 + (j & 1) * 16);
    }
    int p_k = 0, p_slot = 0;
    auto issue_q = [&](int q) {
        const int id = wave + 8 * q;
        if (id < PIECES) {
            const int soff = (int)((p_k >> 1) * (id < PP ? rbstepP : rbstepQ)) + (p_k & 1) * 512;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(id < PP ? rsP : rsQ, (lptr_t*)(lds + p_slot * SLOT + id * PBLK), 16, (int)plane_off[q], soff, 0, 0);
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
    // DUAL: zero-accumulator MFMA + round-to-nearest fold of the p0.q0 product (see pgemm_nt_kernel).  FOLDN = 1: every
    // K-step's product is folded (64 VALU adds per tile and step, which the matrix pipe waits for: one SIMD issues either).
    // FOLDN = 2: a temporary collects the p0.q0 products of TWO K-steps before its fold (the second MFMA truncates at one ulp
    // of a 32-row partial sum, far below the running sum's), half the adds; the tiles are staggered, see mult_big.
    static_assert(FOLDN == 1 || FOLDN == 2, "a fold per K-step or per two (four do not fit the register file: profiles/r05_experiments_tried.txt)");
    constexpr int NTB = FOLDN > 1 ? MT * NT : 2;
    f32x16 tbig[NTB];
    if constexpr (FOLDN > 1) {
#pragma unroll
        for (int u = 0; u < NTB; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) tbig[u][e] = 0.f;
    }
    auto fold = [&](auto v_) {
        constexpr int v = decltype(v_)::value;
        acc[v / NT][v % NT] += tbig[v % NTB];
        asm volatile("" : "+v"(acc[v / NT][v % NT]));
    };
    auto big1 = [&](const bf16x4 (&a)[MT][2], const bf16x4 (&b)[NT][2], auto u_, auto chain_) {
        constexpr int u = decltype(u_)::value;
        constexpr bool CHAIN = decltype(chain_)::value;
        const bf16x8 av = __builtin_shufflevector(a[u / NT][0], a[u / NT][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 bv = __builtin_shufflevector(b[u % NT][0], b[u % NT][1], 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (CHAIN) {
            tbig[u % NTB] = mfma1(av, bv, tbig[u % NTB]);
        } else {
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            tbig[u % NTB] = mfma1(av, bv, z);
        }
        asm volatile("" : "+v"(tbig[u % NTB]));
    };
    using CH = std::integral_constant<bool, true>;
    using ZR = std::integral_constant<bool, false>;
    auto mult_big = [&](const bf16x4 (&a)[MT][2], const bf16x4 (&b)[NT][2], auto par_) {
        constexpr int PAR = decltype(par_)::value;
        if constexpr (FOLDN == 1) {
            static_for<0, MT * NT>([&](auto u_) {
                constexpr int u = decltype(u_)::value;
                if constexpr (u >= 2) fold(std::integral_constant<int, u - 2>{});
                big1(a, b, u_, ZR{});
            });
        } else {
            // two tiles FINISH a chain of two K-steps here (odd tiles in even steps, even tiles in odd steps) and are folded into
            // the running sums while the other two START theirs from zero: only two temporaries live across a step boundary
            static_assert(MT * NT == 4, "four tiles per wave");
            constexpr int f0 = PAR == 0 ? 1 : 0, f1 = f0 + 2, s0 = 1 - f0, s1 = s0 + 2;
            big1(a, b, std::integral_constant<int, f0>{}, CH{});
            big1(a, b, std::integral_constant<int, f1>{}, CH{});
            big1(a, b, std::integral_constant<int, s0>{}, ZR{});
            fold(std::integral_constant<int, f0>{});
            big1(a, b, std::integral_constant<int, s1>{}, ZR{});
            fold(std::integral_constant<int, f1>{});
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fold_pending = [&]() {                      // FOLDN = 1: the last two tiles of the step; FOLDN = 2 folds inside mult_big
        if constexpr (FOLDN == 1) {
            fold(std::integral_constant<int, MT * NT - 2>{});
            fold(std::integral_constant<int, MT * NT - 1>{});
        }
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
                if constexpr (DUAL) mult_big(fa[QA], fb0[PAR], par_);
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
    if constexpr (DUAL && FOLDN > 1) {               // the odd tiles started a chain in the last step: one product each, fold it
        fold(std::integral_constant<int, 1>{});
        fold(std::integral_constant<int, 3>{});
    }

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

}  // namespace msn
