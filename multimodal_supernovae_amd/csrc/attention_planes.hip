// fp32-grade attention for NARROW heads (head_dim <= 16: the reference's spectrum transformer, emb 32 / 2 heads, and -- padded --
// its light-curve transformer, emb 64 / 8 heads) on the bf16 matrix cores; replaces SelfAttention.forward / backward of
// ref src/transformer_utils.py:36-89 for sequences of more than 128 tokens (1024-bin spectra, 220 / 200-step series).
//
// The exact-fp32 kernels of attention_mfma.hip run these shapes on v_mfma_f32_16x16x4_f32, a pipe 16 x slower than the bf16
// one.  Here every operand is held as THREE bf16 planes (x = x0 + x1 + x2 exactly: an fp32 significand is three 8-bit
// pieces; the plane GEMMs of pgemm.hip use the same format) and a product is the six plane products with pa + pb < 3
// (dropped terms <= 2^-26 relative); every bf16 x bf16 product is exact in fp32.  v_mfma_f32_16x16x32_bf16 throughout:
//
//   * products over the HEAD dimension (scores S = Q K^T, dP = dO V^T; 16 terms): the 32 k of one instruction are the same 16
//     columns of TWO planes side by side -- lane group g = lane >> 4 supplies plane (g >> 1), columns 8 (g & 1) .. + 7 -- so
//     [a0 | a1].[b1 | b0] = a0 b1 + a1 b0, then [a0 | a2].[b2 | b0] += a0 b2 + a2 b0, then [a0 | a1].[b0 | b1] += a0 b0 + a1 b1:
//     three chained instructions per 16 x 16 tile, small terms first;
//   * products over TOKENS (O = P V, dQ = dS K, dK = dS^T Q, dV = P^T dO): the 32 k are 32 tokens of one plane; the
//     probabilities / score gradients are split into planes IN REGISTERS (truncation split: x0 = x & 0xffff0000, x1 likewise
//     of x - x0, x2 = the rest; v_perm_b32 packs the high halves: 5.5 vector instructions per value) and are directly the B
//     operand, because every first product is oriented with the softmax axis on the registers:
//       forward / dQ kernel: S^T[key][query] = K . Q^T  -> lane (c, g) holds keys 4 g + r of query c: statistics lane-local;
//                            O^T[d][query] += V^T . P^T (A = V^T through ds_read_b64_tr_b16, B = P^T from the accumulators)
//       dK,dV kernel:        S[query][key] = Q . K^T   -> lane (c, g) holds queries 4 g + r of key c;
//                            dV^T[d][key] += dO^T . P, dK^T[d][key] += Q^T . dS
//     p0 . v0 of every 32-token block is formed in a zero accumulator and added to the running sum by the vector ALU, the five
//     small products accumulate in a second accumulator (the bf16 MFMA adds into its accumulator with a truncating
//     rounding: pgemm_kernels.h), added once at the end.
//
// LDS images: a 32-row block of a [rows][16] operand is three 1-KB plane images [32 rows][16 bf16] -- row reads
// (ds_read_b128: lane (c, g) takes row c, half g & 1 of plane g >> 1) and transposed reads (ds_read_b64_tr_b16: four rows x 16
// columns per 16-lane group) of the same image are both conflict-free.  The fp32 rows are split into planes as they are
// staged (global fp32 -> registers -> three 8-byte LDS stores per 4 columns).
// Masked keys score -1e7 (ref :77) through ONE v_min per score against a per-key cap (+inf live, -1e7 masked, -inf beyond the
// sequence); the gradient of a masked score is zero (masked_fill), a 0 / 1 factor per key.
// Row statistics (max, log-sum) have the layout AND the values of attention_mfma.hip (either family's backward may follow either
// forward): exponentials are exp2((s - m) log2 e) -- the difference first, so that a row of equal scores (every key masked at
// -1e7) gives exactly 1 -- and exp2(fma(s - m, log2 e, -log2 l)) in the backward.
#include <algorithm>
#include <math.h>

#include <type_traits>

#include "attention_args.h"

namespace msn {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr float kFillP = -1e7f;                 // ref transformer_utils.py:77
constexpr float kLog2e = 1.4426950408889634f;
constexpr int PB = 1024;                        // one plane image of a 32-row block: [32 rows][16 bf16]
constexpr int BLK = 3 * PB;                     // a 32-row block: three planes
constexpr int CH = 256;                         // streamed rows per LDS chunk
constexpr int NBK = CH / 32;
constexpr int UMAX = 2;                         // staging: 16-byte pieces per thread and image in flight

__device__ __forceinline__ float trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
// (high half of b) << 16 | (high half of a): two bf16 (truncated) in fragment order
__device__ __forceinline__ unsigned hi_pack(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// a, b -> their three planes, packed pairwise; exact: x = x0 + x1 + x2 (each difference is exact in fp32)
struct Pair3 {
    unsigned p0, p1, p2;
};
// (Written on 2-vectors -- one v_pk_add_f32 per level and pair instead of two v_sub_f32, 9 instead of 11 instructions per pair --
//  every kernel of this file got SLOWER by 3 - 5 %: r06 log, item 6.  A packed fp32 add is two issue cycles, and its operands
//  want aligned register pairs.)
__device__ __forceinline__ Pair3 split2(float a, float b) {
    const float ra = a - trunc16(a), rb = b - trunc16(b);
    const float sa = ra - trunc16(ra), sb = rb - trunc16(rb);
    return Pair3{hi_pack(a, b), hi_pack(ra, rb), hi_pack(sa, sb)};
}
struct Planes8 {          // eight values of one lane as three fragments
    u32x4 p0, p1, p2;
};
__device__ __forceinline__ Planes8 split8(const float (&v)[8]) {
    Planes8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const Pair3 t = split2(v[2 * j], v[2 * j + 1]);
        o.p0[j] = t.p0, o.p1[j] = t.p1, o.p2[j] = t.p2;
    }
    return o;
}
// min(a, b) as ONE instruction: the median of (a, b, -inf).  (fminf also canonicalises both operands -- two more instructions
// per score; an inline-asm v_min_f32 is not an option: hipcc pads no MFMA -> VALU wait states in front of an asm statement,
// and it read the score accumulators before the matrix core had written them.)
__device__ __forceinline__ float vmin(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// reductions over the four lane groups (lanes c, c + 16, c + 32, c + 48)
#ifdef MSN_PATTN_SWAP_REDUCE      // two register swaps, no LDS traffic
__device__ __forceinline__ float group_max(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float group_sum(float v) {
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float pair_sum16(float v) {     // v(group g) + v(group g ^ 1)
    const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#else
__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float pair_sum16(float v) { return v + __shfl_xor(v, 16, 64); }
#endif

// ---- the three "head-dimension" fragments of one 16-row tile held in registers (the B operand of S / dP): row = lane & 15,
// columns 8 (g & 1) .. + 7 of planes chosen by g >> 1:  f01 = [x0 | x1], f10 = [x1 | x0], f20 = [x2 | x0]
struct HeadFrags {
    bf16x8 f01, f10, f20;
};
// the lane's eight columns of row `row` (clamped; zeros beyond the matrix / the head), times mul
__device__ __forceinline__ void load_row8(float (&v)[8], const float* __restrict__ src, int64_t ld, int col0, int row, int T,
                                          int g, int hd, float mul) {
    const float* p = src + (int64_t)(row < T ? row : T - 1) * ld + col0;
    const int d0 = 8 * (g & 1);
    const float4 a = *reinterpret_cast<const float4*>(p + (d0 < hd ? d0 : 0));
    const float4 b = *reinterpret_cast<const float4*>(p + (d0 + 4 < hd ? d0 + 4 : 0));
    const bool oka = row < T && d0 < hd, okb = row < T && d0 + 4 < hd;
    v[0] = oka ? a.x * mul : 0.f, v[1] = oka ? a.y * mul : 0.f, v[2] = oka ? a.z * mul : 0.f, v[3] = oka ? a.w * mul : 0.f;
    v[4] = okb ? b.x * mul : 0.f, v[5] = okb ? b.y * mul : 0.f, v[6] = okb ? b.z * mul : 0.f, v[7] = okb ? b.w * mul : 0.f;
}
__device__ __forceinline__ HeadFrags head_frags(const float (&v)[8], int g) {
    const Planes8 s = split8(v);
    const bool lo = g < 2;
    HeadFrags f;
    f.f01 = as_frag(lo ? s.p0 : s.p1);
    f.f10 = as_frag(lo ? s.p1 : s.p0);
    f.f20 = as_frag(lo ? s.p2 : s.p0);
    return f;
}

// ---- heads up to 8 wide: the 32 k of one instruction hold FOUR 8-column plane blocks, so the four small products share ONE instruction
//   [a0 | a1 | a0 | a2] . [b1 | b0 | b2 | b0] = a0 b1 + a1 b0 + a0 b2 + a2 b0      then      [a0 | a1 | . | .] . [b0 | b1 | 0 | 0] += a0 b0 + a1 b1
// -- two instead of three per 16 x 16 tile (the round-5 verdict's item 5).  Lane group g supplies columns 0 .. 7 of plane {0, 1, 0, 2}[g]
// (small) / {0, 1, 0, 0}[g] (big: groups 2 - 3 are multiplied by zeros) of the A operand, {1, 0, 2, 0}[g] / {0, 1, -, -}[g] of the B operand.
// The same struct carries them: f10 = the small-product operand, f01 = the big one.
__device__ __forceinline__ void load_row8_narrow(float (&v)[8], const float* __restrict__ src, int64_t ld, int col0, int row, int T, int hd, float mul) {
    const float* p = src + (int64_t)(row < T ? row : T - 1) * ld + col0;
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + (4 < hd ? 4 : 0));
    const bool oka = row < T, okb = row < T && 4 < hd;
    v[0] = oka ? a.x * mul : 0.f, v[1] = oka ? a.y * mul : 0.f, v[2] = oka ? a.z * mul : 0.f, v[3] = oka ? a.w * mul : 0.f;
    v[4] = okb ? b.x * mul : 0.f, v[5] = okb ? b.y * mul : 0.f, v[6] = okb ? b.z * mul : 0.f, v[7] = okb ? b.w * mul : 0.f;
}
__device__ __forceinline__ HeadFrags head_frags_narrow(const float (&v)[8], int g) {
    const Planes8 s = split8(v);
    const u32x4 zero = {0u, 0u, 0u, 0u};
    HeadFrags f;
    f.f10 = as_frag(g == 0 ? s.p1 : g == 2 ? s.p2 : s.p0);          // [b1 | b0 | b2 | b0]
    f.f01 = as_frag(g == 0 ? s.p0 : g == 1 ? s.p1 : zero);          // [b0 | b1 | 0 | 0]
    f.f20 = f.f01;
    return f;
}
// row fragments of the 16-row tile t of a block, columns 0 .. 7: a_small = [x0 | x1 | x0 | x2], a_big = [x0 | x1 | x0 | x0]
__device__ __forceinline__ void row_frags_narrow(const unsigned char* blk, int t, int c, int g, bf16x8& a_small, bf16x8& a_big) {
    const unsigned char* p = blk + t * 512 + c * 32;
    a_small = *reinterpret_cast<const bf16x8*>(p + (g == 1 ? 1 : g == 3 ? 2 : 0) * PB);
    a_big = *reinterpret_cast<const bf16x8*>(p + (g == 1 ? PB : 0));
}
__device__ __forceinline__ f32x4 head_product_narrow(const bf16x8& a_small, const bf16x8& a_big, const HeadFrags& b) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = mma(a_small, b.f10, s);
    return mma(a_big, b.f01, s);
}

// ---- LDS images ------------------------------------------------------------------------------------------------------------
// A chunk's rows of two matrices on their way into LDS: rows [0, nt) (row stride ld0 / ld1, columns col0 .. col0 + hd - 1) ->
// the plane images img0 / img1 (blocks of 32 rows; rows beyond nt up to the next multiple of 32 and columns beyond hd are
// zeros).  load() REQUESTS the first UMAX 16-byte pieces per thread and image into registers -- issued right after the barrier
// that publishes the previous chunk, so they are in flight while that chunk is multiplied -- commit() splits them into planes
// and writes them (what a small workgroup could not request up front goes through the loop at the end).
struct StagePair {
    float4 a[UMAX], b[UMAX];
    const float* s0; const float* s1;
    int64_t ld0, ld1;
    int col0, nt, hd;
    __device__ __forceinline__ void fetch(float4& x, float4& y, int idx) const {
        const int total = ((nt + 31) & ~31) * 4;
        const int r = idx >> 2, cq = 4 * (idx & 3);
        const bool ok = idx < total && r < nt && cq < hd;
        x = *reinterpret_cast<const float4*>(s0 + (int64_t)(ok ? r : 0) * ld0 + col0 + (ok ? cq : 0));
        y = *reinterpret_cast<const float4*>(s1 + (int64_t)(ok ? r : 0) * ld1 + col0 + (ok ? cq : 0));
    }
    __device__ __forceinline__ void put(unsigned char* img0, unsigned char* img1, float4 x, float4 y, int idx, float mul0) const {
        const int total = ((nt + 31) & ~31) * 4;
        if (idx >= total) return;
        const int r = idx >> 2, q = idx & 3;
        if (r >= nt || 4 * q >= hd) x = y = make_float4(0.f, 0.f, 0.f, 0.f);
        x.x *= mul0, x.y *= mul0, x.z *= mul0, x.w *= mul0;
        const int off = (r >> 5) * BLK + (r & 31) * 32 + q * 8;
        const Pair3 xa = split2(x.x, x.y), xb = split2(x.z, x.w), ya = split2(y.x, y.y), yb = split2(y.z, y.w);
        *reinterpret_cast<u32x2*>(img0 + off) = u32x2{xa.p0, xb.p0};
        *reinterpret_cast<u32x2*>(img0 + off + PB) = u32x2{xa.p1, xb.p1};
        *reinterpret_cast<u32x2*>(img0 + off + 2 * PB) = u32x2{xa.p2, xb.p2};
        *reinterpret_cast<u32x2*>(img1 + off) = u32x2{ya.p0, yb.p0};
        *reinterpret_cast<u32x2*>(img1 + off + PB) = u32x2{ya.p1, yb.p1};
        *reinterpret_cast<u32x2*>(img1 + off + 2 * PB) = u32x2{ya.p2, yb.p2};
    }
    // (tid: the thread's index, handed in as an OPAQUE copy per chunk by the callers -- opaque_tid() -- so that the piece indices
    //  derived from it are recomputed per chunk instead of living across the block loop: they were what the allocator spilled)
    __device__ __forceinline__ void load(const float* src0, const float* src1, int64_t l0, int64_t l1, int col, int rows, int width, int tid) {
        s0 = src0, s1 = src1, ld0 = l0, ld1 = l1, col0 = col, nt = rows, hd = width;
#pragma unroll
        for (int u = 0; u < UMAX; ++u) fetch(a[u], b[u], tid + u * (int)blockDim.x);
    }
    __device__ __forceinline__ void commit(unsigned char* img0, unsigned char* img1, int tid, float mul0 = 1.f) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) put(img0, img1, a[u], b[u], tid + u * (int)blockDim.x, mul0);
        const int total = ((nt + 31) & ~31) * 4;
        for (int idx = tid + UMAX * (int)blockDim.x; idx < total; idx += (int)blockDim.x) {   // small workgroups only
            float4 x, y;
            fetch(x, y, idx);
            put(img0, img1, x, y, idx, mul0);
        }
    }
};
__device__ __forceinline__ int opaque_tid() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
// row fragments (A operand of S / dP) of the 16-row tile t of a block: a01 = [x0 | x1], a02 = [x0 | x2]
__device__ __forceinline__ void row_frags(const unsigned char* blk, int t, int c, int g, bf16x8& a01, bf16x8& a02) {
    const unsigned char* p = blk + t * 512 + c * 32 + (g & 1) * 16;
    a01 = *reinterpret_cast<const bf16x8*>(p + (g >> 1) * PB);
    a02 = *reinterpret_cast<const bf16x8*>(p + (g >> 1) * 2 * PB);
}
// transposed fragment (A operand of the token products) of plane pl of a block: lane (c, g) gets column c of rows
// 4 g .. 4 g + 3 (elements 0 - 3) and 16 + 4 g .. + 3 (elements 4 - 7) -- the k order in which the accumulators hold P / dS
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* blk, int pl, int c, int g) {
    const unsigned char* p = blk + pl * PB + (4 * g + (c >> 2)) * 32 + (c & 3) * 8;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * 32));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// 16 x 16 tile over the head dimension, small terms first:  a0 b1 + a1 b0, + a0 b2 + a2 b0, + a0 b0 + a1 b1
__device__ __forceinline__ f32x4 head_product(const bf16x8& a01, const bf16x8& a02, const HeadFrags& b) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = mma(a01, b.f10, s);
    s = mma(a02, b.f20, s);
    return mma(a01, b.f01, s);
}
// acc (+)= X^T . W over 32 tokens: xt[pl] = transposed fragments of X's planes, w = W's planes from the accumulators
__device__ __forceinline__ void token_product(const bf16x8 (&xt)[3], const Planes8& w, f32x4& big, f32x4& small) {
    const bf16x8 w0 = as_frag(w.p0), w1 = as_frag(w.p1), w2 = as_frag(w.p2);
    // x0 . w0 into a ZERO accumulator, folded in by the vector ALU (round to nearest): the bf16 MFMA truncates when it adds
    // into its accumulator, and a running sum over 1024 tokens (32 chained instructions) collected that as a bias -- dV of a
    // peaked softmax came out at 1.5 - 1.6 x the exact-fp32 kernels' maximum error; the small products' sum is 2^-8 of it
    big += mma(xt[0], w0, f32x4{0.f, 0.f, 0.f, 0.f});
    small = mma(xt[2], w0, small);
    small = mma(xt[0], w2, small);
    small = mma(xt[1], w1, small);
    small = mma(xt[1], w0, small);
    small = mma(xt[0], w1, small);
}

// per-key caps of a chunk: +inf live, `fill` masked out (-1e7: ref :77; the dQ kernel passes -inf: the gradient of a masked
// score is zero -- masked_fill -- also in a row whose keys are ALL masked, where its probability is not), -inf beyond the sequence
// Returns whether this thread staged a key that is not live (the chunk's blocks then take the form with the v_min).
__device__ __forceinline__ int stage_caps(float* cap, float fill, const uint8_t* mask, int64_t moff, int nt) {
    const int rows = (nt + 31) & ~31;
    int special = 0;
    for (int j = threadIdx.x; j < rows; j += blockDim.x) {
        const bool in = j < nt;
        const bool on = in && (!mask || mask[moff + j] != 0);
        cap[j] = on ? INFINITY : (in ? fill : -INFINITY);
        special |= on ? 0 : 1;
    }
    return special;
}

// ------------------------------------------------------------------------------------------ forward
// Workgroup = (sample, head, block of 128 QT queries); wave w owns QT tiles of 16 queries; K / V stream through LDS in chunks.
// (MSN_ABL_PATTN diagnostic builds: p.tail carries ablation bits -- 1 = stage the first chunk only, 2 = no products)
template <int QT>
__global__ __launch_bounds__(512, 4) void pattn_fwd_kernel(const MAttn p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* Ki = smem;
    unsigned char* Vi = smem + NBK * BLK;
    float* Cap = reinterpret_cast<float*>(smem + 2 * NBK * BLK);
    const int NB = (p.Tq + 128 * QT - 1) / (128 * QT);
    int b, hh, blk;
    locate_block(p, NB, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = blk * 128 * QT + wave * 16 * QT;
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    StagePair sp;
    HeadFrags qf[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        float v[8];
        load_row8(v, p.q + (int64_t)b * p.q_bs, p.ldq, col0, q0 + 16 * u + c, p.Tq, g, p.hd, p.scale);
        qf[u] = head_frags(v, g);
    }
    float m[QT], l[QT];
    f32x4 ob[QT], os[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        m[u] = -INFINITY, l[u] = 0.f;
        ob[u] = os[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int k0 = 0; k0 < p.Tk; k0 += CH) {
        const int nt = min(CH, p.Tk - k0), nblk = (nt + 31) >> 5;
        __syncthreads();                                  // every wave is done with the previous chunk
#ifdef MSN_ABL_PATTN
        if (!(p.tail & 1) || k0 == 0)
#endif
        {
            const int tid = opaque_tid();
            sp.load(ksrc + (int64_t)k0 * p.ldk, vsrc + (int64_t)k0 * p.ldv, p.ldk, p.ldv, col0, nt, p.hd, tid);
            sp.commit(Ki, Vi, tid);
        }
        const int special = __syncthreads_or(stage_caps(Cap, kFillP, p.mask, (int64_t)b * p.Tk + k0, nt));
#ifdef MSN_ABL_PATTN
        if (p.tail & 2) continue;
#endif
        // MASKED: the chunk holds a masked-out or padding key -> one v_min per score against the per-key caps
        auto blocks = [&](auto masked_) {
        constexpr bool MASKED = decltype(masked_)::value;
        for (int kb = 0; kb < nblk; ++kb) {
            const unsigned char* kblk = Ki + kb * BLK;
            const unsigned char* vblk = Vi + kb * BLK;
            float s[QT][8], mbs[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) mbs[u] = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t) {                 // (fragments of one tile at a time: they serve every query tile of the wave)
                bf16x8 a01, a02;
                row_frags(kblk, t, c, g, a01, a02);
                const f32x4 cap = *reinterpret_cast<const f32x4*>(Cap + kb * 32 + t * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < QT; ++u) {
                    const f32x4 st = head_product(a01, a02, qf[u]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        s[u][4 * t + r] = MASKED ? vmin(st[r], cap[r]) : st[r];
                        mbs[u] = fmaxf(mbs[u], s[u][4 * t + r]);
                    }
                }
            }
            bf16x8 vt[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) vt[pl] = tr_frag(vblk, pl, c, g);
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                float mb = mbs[u];
                mb = group_max(mb);
                if (__any(mb > m[u])) {                   // (wave-uniform; rare once the running maximum has settled)
                    const float mn = fmaxf(m[u], mb);     // finite: a block holds a key of the sequence, masked ones score -1e7
                    const float alpha = __builtin_amdgcn_exp2f((m[u] - mn) * kLog2e);     // exp2(-inf) = 0 on the first block
                    m[u] = mn;
                    l[u] *= alpha;
                    ob[u] *= alpha;
                    os[u] *= alpha;
                }
                float e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {                 // (s - m first: exact where it matters -- a row of equal scores,
                    e[j] = __builtin_amdgcn_exp2f((s[u][j] - m[u]) * kLog2e);   //  e.g. all keys masked at -1e7, must give e = 1)
                    l[u] += e[j];
                }
                const Planes8 pp = split8(e);
                token_product(vt, pp, ob[u], os[u]);
            }
        }
        };
        if (special) blocks(std::true_type{});
        else blocks(std::false_type{});
    }
    // (the lane's place in the output, from an opaque copy of the thread index: kept from the kernel's start it was spilled)
    const int te = opaque_tid(), ce = te & 15, ge = (te >> 4) & 3, q0e = blk * 128 * QT + (te >> 6) * 16 * QT;
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const float lt = group_sum(l[u]);
        const int q = q0e + 16 * u + ce;
        const int g = ge;
        if (q < p.Tq) {
            const float inv = 1.f / lt;
            if (4 * g < p.hd) {                           // O^T[d = 4 g + r][query c]: one 16-byte store per lane
                const f32x4 o = (ob[u] + os[u]) * inv;
                *reinterpret_cast<f32x4*>(p.out + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + col0 + 4 * g) = o;
            }
            if (g == 0) {
                float* st = p.lse + 2 * (((int64_t)b * p.H + hh) * p.Tq + q);
                st[0] = m[u];
                st[1] = __logf(lt);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dQ (and delta)
// (the two-kernel backward: cross attention, few (sample, head) pairs, and msn_set_attention_planes(3); self-attention over many
//  pairs takes pattn_bwd_fused_kernel below)
constexpr int NT = 1;                           // 16-row tiles per wave (two measured slower in both backward kernels: r05 log, item 3)
__global__ __launch_bounds__(512, 4) void pattn_bwd_dq_kernel(const MAttn p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* Ki = smem;
    unsigned char* Vi = smem + NBK * BLK;
    float* Cap = reinterpret_cast<float*>(smem + 2 * NBK * BLK);
    const int NB = (p.Tq + 128 * NT - 1) / (128 * NT);
    int b, hh, blk;
    locate_block(p, NB, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = blk * 128 * NT + wave * 16 * NT;
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    StagePair sp;
    HeadFrags qf[NT], df[NT];
    float delta[NT], mq[NT], ll2[NT];
    f32x4 qb[NT], qs[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int qrow = q0 + 16 * u + c;
        const bool q_ok = qrow < p.Tq;
        float qv[8], dv8[8], ov[8];
        load_row8(qv, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.hd, p.scale);
        load_row8(dv8, p.dout + (int64_t)b * p.d_bs, p.ldd, col0, qrow, p.Tq, g, p.hd, 1.f);
        load_row8(ov, p.o + (int64_t)b * p.o_bs, p.ldo, col0, qrow, p.Tq, g, p.hd, 1.f);
        const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + (q_ok ? qrow : 0);
        const float lm = p.lse[2 * stat], ll = p.lse[2 * stat + 1];
        qf[u] = head_frags(qv, g), df[u] = head_frags(dv8, g);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d = fmaf(dv8[j], ov[j], d);
        d = pair_sum16(d);                                // the two column halves (groups g, g ^ 1)
        if (g == 0 && q_ok) p.delta[stat] = d;
        delta[u] = d;
        mq[u] = q_ok ? lm : INFINITY;                     // a query beyond the sequence: p = exp2(-inf) = 0
        ll2[u] = ll * kLog2e;
        qb[u] = qs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int k0 = 0; k0 < p.Tk; k0 += CH) {
        const int nt = min(CH, p.Tk - k0), nblk = (nt + 31) >> 5;
        __syncthreads();
#ifdef MSN_ABL_PATTN
        if (!(p.tail & 1) || k0 == 0)
#endif
        {
            const int tid = opaque_tid();
            sp.load(ksrc + (int64_t)k0 * p.ldk, vsrc + (int64_t)k0 * p.ldv, p.ldk, p.ldv, col0, nt, p.hd, tid);
            sp.commit(Ki, Vi, tid);
        }
        const int special = __syncthreads_or(stage_caps(Cap, -INFINITY, p.mask, (int64_t)b * p.Tk + k0, nt));
#ifdef MSN_ABL_PATTN
        if (p.tail & 2) continue;
#endif
        auto blocks = [&](auto masked_) {
        constexpr bool MASKED = decltype(masked_)::value;
        for (int kb = 0; kb < nblk; ++kb) {
            const unsigned char* kblk = Ki + kb * BLK;
            const unsigned char* vblk = Vi + kb * BLK;
            float ds[NT][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 a01, a02, v01, v02;
                row_frags(kblk, t, c, g, a01, a02);
                row_frags(vblk, t, c, g, v01, v02);
                const f32x4 cap = *reinterpret_cast<const f32x4*>(Cap + kb * 32 + t * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const f32x4 st = head_product(a01, a02, qf[u]);
                    const f32x4 dp = head_product(v01, v02, df[u]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = MASKED ? vmin(st[r], cap[r]) : st[r];     // masked / padded key: -inf -> e = 0 -> ds = 0
                        const float e = __builtin_amdgcn_exp2f(fmaf(sc - mq[u], kLog2e, -ll2[u]));
                        ds[u][4 * t + r] = e * (dp[r] - delta[u]);
                    }
                }
            }
            bf16x8 kt[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) kt[pl] = tr_frag(kblk, pl, c, g);
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const Planes8 dsp = split8(ds[u]);
                token_product(kt, dsp, qb[u], qs[u]);
            }
        }
        };
        if (special) blocks(std::true_type{});
        else blocks(std::false_type{});
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int qrow = q0 + 16 * u + c;
        if (qrow < p.Tq && 4 * g < p.hd) {
            const f32x4 o = (qb[u] + qs[u]) * p.scale;
            *reinterpret_cast<f32x4*>(p.dq + (int64_t)b * p.dq_bs + (int64_t)qrow * p.lddq + col0 + 4 * g) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
__global__ __launch_bounds__(512, 4) void pattn_bwd_dkv_kernel(const MAttn p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* Qi = smem;
    unsigned char* Di = smem + NBK * BLK;
    float* Ml = reinterpret_cast<float*>(smem + 2 * NBK * BLK);
    float* Ll = Ml + CH;
    float* Dl = Ll + CH;
    const int NB = (p.Tk + 128 * NT - 1) / (128 * NT);
    int b, hh, blk;
    locate_block(p, NB, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int k0w = blk * 128 * NT + wave * 16 * NT;
    const float* qsrc = p.q + (int64_t)b * p.q_bs;
    const float* dsrc = p.dout + (int64_t)b * p.d_bs;
    StagePair sp;
    // (the softmax scale multiplies Q here too, as in the forward and the dQ kernel: the recomputed scores are then the
    //  forward's scores bit for bit -- with the scale on K, as the exact-fp32 kernels place it, s - m of a row's largest score
    //  is a rounding difference instead of zero and a peaked softmax shows it in dV)
    HeadFrags kf[NT], vf[NT];
    float cap[NT], liv[NT];
    f32x4 kb_[NT], ks[NT], vb[NT], vs[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int krow = k0w + 16 * u + c;
        const bool in_seq = krow < p.Tk;
        float kv[8], vv[8];
        load_row8(kv, p.k + (int64_t)b * p.k_bs, p.ldk, col0, krow, p.Tk, g, p.hd, 1.f);
        load_row8(vv, p.v + (int64_t)b * p.v_bs, p.ldv, col0, krow, p.Tk, g, p.hd, 1.f);
        uint8_t mk = 1;
        if (p.mask) mk = p.mask[(int64_t)b * p.Tk + (in_seq ? krow : 0)];
        const bool keep = in_seq && mk != 0;
        cap[u] = keep ? INFINITY : (in_seq ? kFillP : -INFINITY);
        liv[u] = keep ? 1.f : 0.f;
        kf[u] = head_frags(kv, g), vf[u] = head_frags(vv, g);
        kb_[u] = ks[u] = vb[u] = vs[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int special = 0;                                      // this wave owns a key that is not live: the form with the v_min
#pragma unroll
    for (int u = 0; u < NT; ++u) special |= liv[u] == 0.f ? 1 : 0;
    special = __any(special);
    for (int i0 = 0; i0 < p.Tq; i0 += CH) {
        const int nt = min(CH, p.Tq - i0), nblk = (nt + 31) >> 5;
        __syncthreads();
#ifdef MSN_ABL_PATTN
        if (!(p.tail & 1) || i0 == 0)
#endif
        {
            const int tid = opaque_tid();
            sp.load(qsrc + (int64_t)i0 * p.ldq, dsrc + (int64_t)i0 * p.ldd, p.ldq, p.ldd, col0, nt, p.hd, tid);
            sp.commit(Qi, Di, tid, p.scale);
        }
        for (int j = threadIdx.x; j < ((nt + 31) & ~31); j += blockDim.x) {
            const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + i0 + (j < nt ? j : 0);
            Ml[j] = j < nt ? p.lse[2 * stat] : INFINITY;              // +inf: a padded query row gets p = exp2(-inf) = 0
            Ll[j] = j < nt ? p.lse[2 * stat + 1] * kLog2e : 0.f;
            Dl[j] = j < nt ? p.delta[stat] : 0.f;
        }
        __syncthreads();
#ifdef MSN_ABL_PATTN
        if (p.tail & 2) continue;
#endif
        auto blocks = [&](auto masked_) {
        constexpr bool MASKED = decltype(masked_)::value;
        for (int qb = 0; qb < nblk; ++qb) {
            const unsigned char* qblk = Qi + qb * BLK;
            const unsigned char* dblk = Di + qb * BLK;
            float pr[NT][8], ds[NT][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 a01, a02, d01, d02;
                row_frags(qblk, t, c, g, a01, a02);
                row_frags(dblk, t, c, g, d01, d02);
                const f32x4 ml = *reinterpret_cast<const f32x4*>(Ml + qb * 32 + t * 16 + 4 * g);
                const f32x4 l2 = *reinterpret_cast<const f32x4*>(Ll + qb * 32 + t * 16 + 4 * g);
                const f32x4 dl = *reinterpret_cast<const f32x4*>(Dl + qb * 32 + t * 16 + 4 * g);
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const f32x4 st = head_product(a01, a02, kf[u]);            // rows = queries 4 g + r, column = this lane's key
                    const f32x4 dp = head_product(d01, d02, vf[u]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = MASKED ? vmin(st[r], cap[u]) : st[r];
                        const float e = __builtin_amdgcn_exp2f(fmaf(sc - ml[r], kLog2e, -l2[r]));
                        pr[u][4 * t + r] = e;
                        ds[u][4 * t + r] = e * (dp[r] - dl[r]);         // (x 0 for a masked key: once, on the finished column)
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                {
                    bf16x8 dt[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) dt[pl] = tr_frag(dblk, pl, c, g);
                    const Planes8 pp = split8(pr[u]);
                    token_product(dt, pp, vb[u], vs[u]);
                }
                {
                    bf16x8 qt[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) qt[pl] = tr_frag(qblk, pl, c, g);
                    const Planes8 dsp = split8(ds[u]);
                    token_product(qt, dsp, kb_[u], ks[u]);
                }
            }
        }
        };
        if (special) blocks(std::true_type{});
        else blocks(std::false_type{});
    }
#pragma unroll
    for (int u = 0; u < NT; ++u) {
        const int krow = k0w + 16 * u + c;
        if (krow < p.Tk && 4 * g < p.hd) {
            const f32x4 dk = (kb_[u] + ks[u]) * liv[u], dv = vb[u] + vs[u];     // (Q carried the scale) liv: a masked key's score gradients are zero
            *reinterpret_cast<f32x4*>(p.dk + (int64_t)b * p.dk_bs + (int64_t)krow * p.lddk + col0 + 4 * g) = dk;
            *reinterpret_cast<f32x4*>(p.dv + (int64_t)b * p.dv_bs + (int64_t)krow * p.lddv + col0 + 4 * g) = dv;
        }
    }
}

// ------------------------------------------------------------------------------------------ backward in ONE pass: dQ, dK, dV
// S, dP, the exponentials and the plane split of dS were computed twice -- once with the queries on the lanes (dQ kernel), once
// with the keys (dK,dV kernel) -- because a token product needs its reduction index on the REGISTERS of both operands: ~30 % of the
// backward's instructions.  Here every score tile is computed ONCE, in the dK,dV orientation (S[query][key] = Q . K^T, lane = key):
//   * dV^T += dO^T . P and dK^T += Q^T . dS as in pattn_bwd_dkv_kernel;
//   * the planes of dS -- already split for the dK product -- go to a patch of LDS as dS^T[key][query] (8-byte stores) and come back
//     through ds_read_b64_tr_b16 with the KEYS on the registers: the B operand of dQ^T[d][query] += K^T[d][key] . dS^T[key][query].
//     A pair of waves (32 keys = one instruction's k) shares a patch; wave 2 j + i multiplies query tile i of the 32-query block
//     against the pair's keys (K^T fragments: transposed reads of a plane image of the key block's 128 keys, staged once per key
//     block -- held in registers they were the twelve that made the allocator spill accumulators inside the block loop), so each
//     wave adds 6 instructions per block, none of them wasted;
//   * the four pairs' shares of a block's dQ (32 queries x 16 columns) meet in LDS and are summed in pair order by the 512 threads,
//     one output element each;
//   * a workgroup owns a (sample, head) and walks ALL its key blocks of 128, so the sum over key blocks is that thread's own
//     read-modify-write of dq in memory (first key block: a plain store; L2-resident in between, 64 KB per (sample, head) at 1024
//     tokens; the same thread, hence the same CU, wrote the 16 bytes it reads; streaming accesses that do not linger in the CU's
//     L1) -- no slabs, no second launch, a fixed order.
// delta = rowsum(dO o O) is formed by the first key block from the dO pieces it stages (+ the matching pieces of O) and kept in p.delta
// for the others (a 16 - 25 us launch of its own at first: 1.6 % of the Maven step).
// Two barriers per 32-query block (patch written | shares written); 69.5 KB of LDS: two workgroups per CU.
constexpr int CHB = 128;                        // streamed query rows per chunk
constexpr int NBKB = CHB / 32;
constexpr int XP = 64;                          // bytes per key row of a patch plane: 32 queries (bf16) = eight 8-byte chunks; chunk j of
                                                // row r sits at position j ^ xswz(r) -- the 8-byte stores of a block (16 rows x 2
                                                // chunks per half wave) and the transposed reads (8 rows x 4 chunks) both conflict-free
constexpr int PATCH = 32 * XP;                  // one plane of a pair's patch: [32 keys][32 queries]
__device__ __forceinline__ int xswz(int row) { return ((row >> 2) & 1) << 2 | ((row >> 3) & 1) << 1; }
constexpr int FUSED_LDS = 2 * NBKB * BLK + 3 * CHB * 4 + 4 * 3 * PATCH + 4 * 32 * 16 * 4 + 4 * BLK;

template <bool NARROW>      // heads up to 8 wide: two head-dimension instructions per score tile instead of three (see head_frags_narrow)
__global__ __launch_bounds__(512, 4) void pattn_bwd_fused_kernel(const MAttn p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* Qi = smem;
    unsigned char* Di = smem + NBKB * BLK;
    float* Ml = reinterpret_cast<float*>(smem + 2 * NBKB * BLK);
    float* Ll = Ml + CHB;
    float* Dl = Ll + CHB;
    unsigned char* Xp = reinterpret_cast<unsigned char*>(Dl + CHB);       // [4 pairs][3 planes][32 keys][XP]
    float* Ex = reinterpret_cast<float*>(Xp + 4 * 3 * PATCH);             // [4 pairs][32 queries][16]
    unsigned char* Kimg = reinterpret_cast<unsigned char*>(Ex + 4 * 32 * 16);   // the key block's K as plane images: four 32-key blocks
    int b, hh, blk;
    locate_block(p, 1, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int pair = wave >> 1, qtile = wave & 1;
    const float* qsrc = p.q + (int64_t)b * p.q_bs;
    const float* dsrc = p.dout + (int64_t)b * p.d_bs;
    const float* kbase = p.k + (int64_t)b * p.k_bs;
    const float* vbase = p.v + (int64_t)b * p.v_bs;
    float* dqb = p.dq + (int64_t)b * p.dq_bs + col0;
    StagePair sp;
    // this lane's 8-byte slots in its pair's patch (dS^T[key = 16 (wave & 1) + c][queries 4 g .. + 3 | 16 + 4 g .. + 3]) and the
    // chunks it reads back (keys 4 g + (c >> 2) | + 16, queries 16 qtile + 4 (c & 3) .. + 3)
    // (32-bit offsets, not pointers: a generic pointer per lane is a register pair the allocator would rather spill)
    const int xw_off = pair * 3 * PATCH + (16 * qtile + c) * XP + 8 * (g ^ xswz(16 * qtile + c));      // + 32: chunk 4 + g
    const int xr_off = pair * 3 * PATCH + (4 * g + (c >> 2)) * XP + 8 * ((4 * qtile + (c & 3)) ^ xswz(4 * g + (c >> 2)));   // + 16 rows: same swizzle
    for (int kb0 = 0; kb0 < p.Tk; kb0 += 128) {
        // ---- this wave's 16 keys (B operands of S / dP); the key block's K as a plane image (A operand of the dQ product)
        HeadFrags kf, vf;
        float cap, liv;
        {
            // (lane constants of this prologue from an opaque copy of the thread index: hoisted out of the key-block loop as 64-bit
            //  offsets they were spilled)
            const int tid = opaque_tid(), pc = tid & 15, pg = (tid >> 4) & 3;
            const int krow = kb0 + (tid >> 6) * 16 + pc;
            const bool in_seq = krow < p.Tk;
            float kv[8], vv[8];
            if constexpr (NARROW) {
                load_row8_narrow(kv, kbase, p.ldk, col0, krow, p.Tk, p.hd, 1.f);
                load_row8_narrow(vv, vbase, p.ldv, col0, krow, p.Tk, p.hd, 1.f);
            } else {
                load_row8(kv, kbase, p.ldk, col0, krow, p.Tk, pg, p.hd, 1.f);
                load_row8(vv, vbase, p.ldv, col0, krow, p.Tk, pg, p.hd, 1.f);
            }
            uint8_t mk = 1;
            if (p.mask) mk = p.mask[(int64_t)b * p.Tk + (in_seq ? krow : 0)];
            const bool keep = in_seq && mk != 0;
            cap = keep ? INFINITY : (in_seq ? kFillP : -INFINITY);
            liv = keep ? 1.f : 0.f;
            if constexpr (NARROW) kf = head_frags_narrow(kv, pg), vf = head_frags_narrow(vv, pg);
            else kf = head_frags(kv, pg), vf = head_frags(vv, pg);
            // 128 rows x four 16-byte pieces = one piece per thread (the previous key block's last reads of the image lie before the
            // barrier that closed its last query block)
            const int r = tid >> 2, q4 = tid & 3;
            const bool ok = kb0 + r < p.Tk && 4 * q4 < p.hd;
            float4 x = *reinterpret_cast<const float4*>(kbase + (int64_t)(ok ? kb0 + r : 0) * p.ldk + col0 + (ok ? 4 * q4 : 0));
            if (!ok) x = make_float4(0.f, 0.f, 0.f, 0.f);
            const Pair3 xa = split2(x.x, x.y), xb = split2(x.z, x.w);
            unsigned char* dst = Kimg + (r >> 5) * BLK + (r & 31) * 32 + q4 * 8;
            *reinterpret_cast<u32x2*>(dst) = u32x2{xa.p0, xb.p0};
            *reinterpret_cast<u32x2*>(dst + PB) = u32x2{xa.p1, xb.p1};
            *reinterpret_cast<u32x2*>(dst + 2 * PB) = u32x2{xa.p2, xb.p2};
        }
        f32x4 kbg = {0.f, 0.f, 0.f, 0.f}, ksm = kbg, vbg = kbg, vsm = kbg;
        const int special = __any(liv == 0.f ? 1 : 0);    // this wave owns a key that is not live: the form with the v_min
        // A wave whose 16 keys ALL lie past the sequence (the last key block of a 200- or 220-token series: 3 or 2 of its 8 waves) skips
        // the score work -- its rows of the pair's dS^T patch are zeroed once here and never rewritten -- and, when its partner is past
        // the sequence too, the pair's dQ product (its shares zeroed once): the kernel is bound by the instructions a SIMD issues, so
        // what these waves do not issue the waves beside them do sooner.  (Zeroed behind the first barrier of the key block: the previous
        // key block's last reads of the patch lie before its last barrier, those of the shares behind it.)
        const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const bool dead = kb0 + 16 * wave_u >= p.Tk, pair_dead = kb0 + 32 * (wave_u >> 1) >= p.Tk;
        for (int i0 = 0; i0 < p.Tq; i0 += CHB) {
            const int nt = min(CHB, p.Tq - i0), nblk = (nt + 31) >> 5;
            __syncthreads();                              // every wave is done with the previous chunk's images (and shares)
            if (i0 == 0) {
                if (dead) {
                    unsigned char* xw = Xp + xw_off;
                    unsigned char* xw2 = Xp + (xw_off ^ 32);
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        *reinterpret_cast<u32x2*>(xw + pl * PATCH) = u32x2{0u, 0u};
                        *reinterpret_cast<u32x2*>(xw2 + pl * PATCH) = u32x2{0u, 0u};
                    }
                }
                if (pair_dead) {
                    const int te = opaque_tid(), ec = te & 15, eg = (te >> 4) & 3, epair = te >> 7, eqt = (te >> 6) & 1;
                    *reinterpret_cast<f32x4*>(Ex + ((epair * 32 + 16 * eqt + ec) * 16 + 4 * (eg ^ ((ec >> 2) & 3)))) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            {
                const int tid = opaque_tid();
                sp.load(qsrc + (int64_t)i0 * p.ldq, dsrc + (int64_t)i0 * p.ldd, p.ldq, p.ldd, col0, nt, p.hd, tid);
                if (kb0 == 0) {
                    // delta = rowsum(dO o O) of the chunk's rows, once per (sample, head): the FIRST key block forms it from the dO
                    // pieces it is staging anyway (a 128-row chunk = one 4-column piece per thread: sp.b[0]) and the matching pieces
                    // of O, sums a row's four pieces over four neighbouring lanes and leaves it in p.delta for the later key blocks
                    // (same workgroup: the barriers in between order the store and their loads) -- a launch of its own before
                    const int r = tid >> 2, cq = 4 * (tid & 3);
                    const bool ok = r < nt && cq < p.hd;
                    const float4 o = *reinterpret_cast<const float4*>(p.o + (int64_t)b * p.o_bs + (int64_t)(i0 + (ok ? r : 0)) * p.ldo + col0 + (ok ? cq : 0));
                    const float4 d = sp.b[0];
                    float part = ok ? fmaf(d.x, o.x, fmaf(d.y, o.y, fmaf(d.z, o.z, d.w * o.w))) : 0.f;
                    part += __shfl_xor(part, 1, 64);
                    part += __shfl_xor(part, 2, 64);
                    if ((tid & 3) == 0) {
                        Dl[r] = part;                                        // (rows beyond nt: 0)
                        if (r < nt) p.delta[((int64_t)b * p.H + hh) * p.Tq + i0 + r] = part;
                    }
                }
                sp.commit(Qi, Di, tid, p.scale);
                if (tid < CHB) {
                    const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + i0 + (tid < nt ? tid : 0);
                    Ml[tid] = tid < nt ? p.lse[2 * stat] : INFINITY;          // +inf: a padded query row gets p = exp2(-inf) = 0
                    Ll[tid] = tid < nt ? p.lse[2 * stat + 1] * kLog2e : 0.f;
                    if (kb0 > 0) Dl[tid] = tid < nt ? p.delta[stat] : 0.f;
                }
            }
            __syncthreads();
            auto blocks = [&](auto masked_) {
            constexpr bool MASKED = decltype(masked_)::value;
            for (int qb = 0; qb < nblk; ++qb) {
                const unsigned char* qblk = Qi + qb * BLK;
                const unsigned char* dblk = Di + qb * BLK;
                if (!dead) {
                float pr[8], ds[8];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bf16x8 a01, a02, d01, d02;
                    if constexpr (NARROW) {
                        row_frags_narrow(qblk, t, c, g, a01, a02);
                        row_frags_narrow(dblk, t, c, g, d01, d02);
                    } else {
                        row_frags(qblk, t, c, g, a01, a02);
                        row_frags(dblk, t, c, g, d01, d02);
                    }
                    const f32x4 ml = *reinterpret_cast<const f32x4*>(Ml + qb * 32 + t * 16 + 4 * g);
                    const f32x4 l2 = *reinterpret_cast<const f32x4*>(Ll + qb * 32 + t * 16 + 4 * g);
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(Dl + qb * 32 + t * 16 + 4 * g);
                    // rows = queries 4 g + r, column = this lane's key
                    const f32x4 st = NARROW ? head_product_narrow(a01, a02, kf) : head_product(a01, a02, kf);
                    const f32x4 dp = NARROW ? head_product_narrow(d01, d02, vf) : head_product(d01, d02, vf);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sc = MASKED ? vmin(st[r], cap) : st[r];
                        const float e = __builtin_amdgcn_exp2f(fmaf(sc - ml[r], kLog2e, -l2[r]));
                        pr[4 * t + r] = e;
                        const float dsv = e * (dp[r] - dl[r]);
                        ds[4 * t + r] = MASKED ? dsv * liv : dsv;             // a masked key's score gradients are zero (masked_fill)
                    }
                }
                {
                    bf16x8 dt[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) dt[pl] = tr_frag(dblk, pl, c, g);
                    const Planes8 pp = split8(pr);
                    token_product(dt, pp, vbg, vsm);
                }
                {
                    bf16x8 qt[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) qt[pl] = tr_frag(qblk, pl, c, g);
                    const Planes8 dsp = split8(ds);
                    token_product(qt, dsp, kbg, ksm);
                    // dS^T[key][query] into the pair's patch: p?[0 .. 1] = queries 4 g .. + 3, p?[2 .. 3] = queries 16 + 4 g .. + 3
                    unsigned char* xw = Xp + xw_off;
                    unsigned char* xw2 = Xp + (xw_off ^ 32);      // chunk (4 + g) ^ swizzle = (g ^ swizzle) ^ 4
                    *reinterpret_cast<u32x2*>(xw) = u32x2{dsp.p0[0], dsp.p0[1]};
                    *reinterpret_cast<u32x2*>(xw2) = u32x2{dsp.p0[2], dsp.p0[3]};
                    *reinterpret_cast<u32x2*>(xw + PATCH) = u32x2{dsp.p1[0], dsp.p1[1]};
                    *reinterpret_cast<u32x2*>(xw2 + PATCH) = u32x2{dsp.p1[2], dsp.p1[3]};
                    *reinterpret_cast<u32x2*>(xw + 2 * PATCH) = u32x2{dsp.p2[0], dsp.p2[1]};
                    *reinterpret_cast<u32x2*>(xw2 + 2 * PATCH) = u32x2{dsp.p2[2], dsp.p2[3]};
                }
                }       // !dead
#ifdef MSN_ABL_PATTN                         // diagnostic builds: bit 1 = no barriers in the block loop, bit 0 = no dq traffic (wrong results)
                if (!(p.tail & 2))
#endif
                __syncthreads();                          // the pair's patch is complete
                // the four elements of dq this thread owns in the block (threads 0 .. 127: query t >> 2, columns 4 (t & 3) .. + 3 -- 16-byte
                // accesses: one dword per thread over all 512 made the read-modify-write 12 % of the kernel): their value so far, requested
                // here (not at the head of the block: four registers across its widest part) and in flight under the dQ product
                const int te = opaque_tid();
                const int oq = i0 + 32 * qb + (te >> 2), od = 4 * (te & 3);
                const bool o_ok = te < 128 && oq < p.Tq && od < p.hd;
                float* optr = dqb + (int64_t)(o_ok ? oq : 0) * p.lddq + (o_ok ? od : 0);
                f32x4 old = {0.f, 0.f, 0.f, 0.f};
#ifdef MSN_ABL_PATTN
                if (!(p.tail & 1))
#endif
                if (kb0 > 0 && o_ok) old = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(optr));
                if (!pair_dead) {
                    const unsigned char* xr = Xp + xr_off;
                    Planes8 w;
                    u32x4* wp[3] = {&w.p0, &w.p1, &w.p2};
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xr + pl * PATCH));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(xr + pl * PATCH + 16 * XP));
                        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        *wp[pl] = __builtin_bit_cast(u32x4, v);
                    }
                    const int ec = te & 15, eg = (te >> 4) & 3, epair = te >> 7, eqt = (te >> 6) & 1;   // (lane constants per block: see `te`)
                    bf16x8 kt[3];                         // K^T[d = c][the pair's keys, in the k order of the patch reads]
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) kt[pl] = tr_frag(Kimg + epair * BLK, pl, ec, eg);
                    f32x4 qbg = {0.f, 0.f, 0.f, 0.f}, qsm = qbg;
                    token_product(kt, w, qbg, qsm);       // dQ^T[d = 4 g + r][query 16 qtile + c] over the pair's 32 keys
                    // (16-byte column group j of query row q at position j ^ ((q >> 2) & 3): stores and the reduction's reads conflict-free)
                    *reinterpret_cast<f32x4*>(Ex + ((epair * 32 + 16 * eqt + ec) * 16 + 4 * (eg ^ ((ec >> 2) & 3)))) = qbg + qsm;
                }
#ifdef MSN_ABL_PATTN
                if (!(p.tail & 2))
#endif
                __syncthreads();                          // the four pairs' shares are in place
#ifdef MSN_ABL_PATTN
                if (!(p.tail & 1))
#endif
                if (te < 128) {
                    const f32x4* ex = reinterpret_cast<const f32x4*>(Ex) + (te ^ ((te >> 4) & 3));   // [pair][query][4 column groups, swizzled]
                    const f32x4 sum = ((ex[0] + ex[128]) + ex[256]) + ex[384];
                    if (o_ok) __builtin_nontemporal_store(sum * p.scale + old, reinterpret_cast<f32x4*>(optr));
                }
            }
            };
            if (special) blocks(std::true_type{});
            else blocks(std::false_type{});
        }
        {
            const int te = opaque_tid(), ce = te & 15, ge = (te >> 4) & 3;
            const int krow = kb0 + (te >> 6) * 16 + ce;
            if (krow < p.Tk && 4 * ge < p.hd) {
                const f32x4 dk = (kbg + ksm) * liv, dv = vbg + vsm;             // (Q carried the scale)
                *reinterpret_cast<f32x4*>(p.dk + (int64_t)b * p.dk_bs + (int64_t)krow * p.lddk + col0 + 4 * ge) = dk;
                *reinterpret_cast<f32x4*>(p.dv + (int64_t)b * p.dv_bs + (int64_t)krow * p.lddv + col0 + 4 * ge) = dv;
            }
        }
    }
}

template <typename K>
int launch(K kernel, unsigned grid, unsigned block, size_t lds, hipStream_t st, const MAttn& a) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        set_error("plane attention: cannot reserve %zu bytes of LDS", lds);
        return MSN_ERR_HIP;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

int g_planes_on = 1;
int g_fused_bwd = 1;    // backward in one pass where it applies (0: always the dQ kernel followed by the dK,dV kernel; 2: always one pass)
#ifdef MSN_ABL_PATTN
int g_ablate = 0;
#endif

}  // namespace

// heads up to 16 wide (a multiple of 4), every row 16-byte aligned
bool pattn_applicable(const MAttn& a) {
    if (!g_planes_on || a.hd % 4 != 0 || a.hd > 16 || a.hd < 4 || a.Tq > 65535 || a.Tk > 65535 || a.q_bs == 0) return false;
    if ((int64_t)a.B * a.H * ((std::max(a.Tq, a.Tk) + 127) / 128) > 0x7fffffffLL) return false;
    const int64_t lds[] = {a.ldq, a.ldk, a.ldv, a.q_bs, a.k_bs, a.v_bs, a.hd * (int64_t)a.H};
    for (int64_t v : lds)
        if (v % 4 != 0) return false;
    const void* ptrs[] = {a.q, a.k, a.v};
    for (const void* ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) & 15) return false;
    return true;
}
bool pattn_forward_aligned(const MAttn& a) {
    return (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && a.ldo % 4 == 0 && a.o_bs % 4 == 0;
}
bool pattn_backward_aligned(const MAttn& a) {
    const int64_t al[] = {a.ldd, a.d_bs, a.ldo, a.o_bs, a.lddq, a.lddk, a.lddv, a.dq_bs, a.dk_bs, a.dv_bs};
    bool ok = ((reinterpret_cast<uintptr_t>(a.dout) | reinterpret_cast<uintptr_t>(a.o) | reinterpret_cast<uintptr_t>(a.dq) |
                reinterpret_cast<uintptr_t>(a.dk) | reinterpret_cast<uintptr_t>(a.dv)) & 15) == 0;
    for (int64_t v : al) ok = ok && (v % 4 == 0);
    return ok;
}

// threads of a workgroup whose waves own `rows_per_wave` rows each (at most eight waves)
static unsigned block_threads(int T, int rows_per_wave) {
    return 64u * (unsigned)std::min(8, (T + rows_per_wave - 1) / rows_per_wave);
}

int pattn_forward(const MAttn& a0, hipStream_t st) {
    if (!pattn_forward_aligned(a0)) {
        set_error("plane attention forward: out must be 16-byte aligned with strides %% 4 == 0");
        return MSN_ERR_SHAPE;
    }
    MAttn a = a0;
    a.tail = 0;
#ifdef MSN_ABL_PATTN
    a.tail = g_ablate;
#endif
    const size_t lds = 2 * NBK * BLK + sizeof(float) * CH;
    if (a.Tq > 128)       // two query tiles per wave: the K / V fragments of a block serve both (1185 -> 1112 us at 1024 tokens)
        return launch(pattn_fwd_kernel<2>, (unsigned)(a.B * a.H * ((a.Tq + 255) / 256)), block_threads(a.Tq, 32), lds, st, a);
    return launch(pattn_fwd_kernel<1>, (unsigned)(a.B * a.H * ((a.Tq + 127) / 128)), block_threads(a.Tq, 16), lds, st, a);
}

// The one-pass backward holds a (sample, head) in ONE workgroup: it needs enough pairs to fill the chip (two workgroups per CU)
// and the dq accumulation in memory (a stride of whole floats); everything else takes the two kernels.
static bool fused_backward_applies(const MAttn& a) { return g_fused_bwd == 2 || (g_fused_bwd && (int64_t)a.B * a.H >= 256); }

// Heads NARROWER than 16 (the light-curve tower's 8-wide heads) run their forward on the vector-ALU kernels -- the plane forward
// ties with them at best (304 vs 242 us at 1024 x 8 heads x 200 tokens) -- but their backward is faster in one pass on the planes
// (676 vs 753 us): the row statistics are the same (max, log-sum) pairs in every family.
bool pattn_backward_preferred(const MAttn& a) {
    return a.hd < 16 && (a.Tq > 128 || a.Tk > 128) && pattn_applicable(a) && pattn_backward_aligned(a) && fused_backward_applies(a);
}

int pattn_backward(const MAttn& a0, hipStream_t st) {
    if (!pattn_backward_aligned(a0)) {
        set_error("plane attention backward: out / dout / dq / dk / dv must be 16-byte aligned with strides %% 4 == 0");
        return MSN_ERR_SHAPE;
    }
    MAttn a = a0;
    a.tail = 0;
#ifdef MSN_ABL_PATTN
    a.tail = g_ablate;
#endif
    if (fused_backward_applies(a)) {
#ifndef MSN_PATTN_NO_NARROW
        if (a.hd <= 8) return launch(pattn_bwd_fused_kernel<true>, (unsigned)(a.B * a.H), 512u, (size_t)FUSED_LDS, st, a);
#endif
        return launch(pattn_bwd_fused_kernel<false>, (unsigned)(a.B * a.H), 512u, (size_t)FUSED_LDS, st, a);
    }
    {
        const size_t lds = 2 * NBK * BLK + sizeof(float) * CH;
        if (int rc = launch(pattn_bwd_dq_kernel, (unsigned)(a.B * a.H * ((a.Tq + 127) / 128)), block_threads(a.Tq, 16), lds, st, a)) return rc;
    }
    const size_t lds = 2 * NBK * BLK + sizeof(float) * 3 * CH;
    return launch(pattn_bwd_dkv_kernel, (unsigned)(a.B * a.H * ((a.Tk + 127) / 128)), block_threads(a.Tk, 16), lds, st, a);
}

}  // namespace msn

using namespace msn;

extern "C" int msn_set_attention_planes(int mode) {
#ifdef MSN_ABL_PATTN
    MSN_REQUIRE(mode >= 0 && mode < 64, "msn_set_attention_planes (diagnostic build): bits 4 - 5 = ablations");
    g_ablate = (mode >> 4) & 3;
    mode &= 3;
#endif
    MSN_REQUIRE(mode == 0 || mode == 1 || mode == 3 || mode == 5, "msn_set_attention_planes: 0 = off (the exact-fp32 matrix-core kernels), "
                "1 = on, 3 = on with the two-kernel backward (dQ, then dK,dV) everywhere, 5 = on with the one-pass backward everywhere");
    g_planes_on = mode & 1;
    g_fused_bwd = (mode & 2) ? 0 : ((mode & 4) ? 2 : 1);
    return MSN_OK;
}
