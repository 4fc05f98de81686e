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
    __device__ __forceinline__ void load(const float* src0, const float* src1, int64_t l0, int64_t l1, int col, int rows, int width) {
        s0 = src0, s1 = src1, ld0 = l0, ld1 = l1, col0 = col, nt = rows, hd = width;
#pragma unroll
        for (int u = 0; u < UMAX; ++u) fetch(a[u], b[u], (int)threadIdx.x + u * (int)blockDim.x);
    }
    __device__ __forceinline__ void commit(unsigned char* img0, unsigned char* img1, float mul0 = 1.f) {
#pragma unroll
        for (int u = 0; u < UMAX; ++u) put(img0, img1, a[u], b[u], (int)threadIdx.x + u * (int)blockDim.x, mul0);
        const int total = ((nt + 31) & ~31) * 4;
        for (int idx = (int)threadIdx.x + UMAX * (int)blockDim.x; idx < total; idx += (int)blockDim.x) {   // small workgroups only
            float4 x, y;
            fetch(x, y, idx);
            put(img0, img1, x, y, idx, mul0);
        }
    }
};
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
// p.tail carries diagnostic bits (tools/bench_attention_planes.py --ablate): 1 = stage the first chunk only, 2 = no products.
template <int QT, bool PF>
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
    if constexpr (PF) sp.load(ksrc, vsrc, p.ldk, p.ldv, col0, min(CH, p.Tk), p.hd);
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
        if (!(p.tail & 1) || k0 == 0) {
            if constexpr (!PF) sp.load(ksrc + (int64_t)k0 * p.ldk, vsrc + (int64_t)k0 * p.ldv, p.ldk, p.ldv, col0, nt, p.hd);
            sp.commit(Ki, Vi);
        }
        const int special = __syncthreads_or(stage_caps(Cap, kFillP, p.mask, (int64_t)b * p.Tk + k0, nt));
        if (PF && k0 + CH < p.Tk && !(p.tail & 1))        // the next chunk's rows: in flight while this one is multiplied
            sp.load(ksrc + (int64_t)(k0 + CH) * p.ldk, vsrc + (int64_t)(k0 + CH) * p.ldv, p.ldk, p.ldv, col0, min(CH, p.Tk - k0 - CH), p.hd);
        if (p.tail & 2) continue;
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
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const float lt = group_sum(l[u]);
        const int q = q0 + 16 * u + c;
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
// NT tiles of 16 queries per wave (the K / V fragments of a block serve all of them)
template <int NT, bool PF>
__global__ __launch_bounds__(512, NT == 1 ? 4 : 2) void pattn_bwd_dq_kernel(const MAttn p) {
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
    if constexpr (PF) sp.load(ksrc, vsrc, p.ldk, p.ldv, col0, min(CH, p.Tk), p.hd);
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
        if (!(p.tail & 1) || k0 == 0) {
            if constexpr (!PF) sp.load(ksrc + (int64_t)k0 * p.ldk, vsrc + (int64_t)k0 * p.ldv, p.ldk, p.ldv, col0, nt, p.hd);
            sp.commit(Ki, Vi);
        }
        const int special = __syncthreads_or(stage_caps(Cap, -INFINITY, p.mask, (int64_t)b * p.Tk + k0, nt));
        if (PF && k0 + CH < p.Tk && !(p.tail & 1))
            sp.load(ksrc + (int64_t)(k0 + CH) * p.ldk, vsrc + (int64_t)(k0 + CH) * p.ldv, p.ldk, p.ldv, col0, min(CH, p.Tk - k0 - CH), p.hd);
        if (p.tail & 2) continue;
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
template <int NT, bool PF>
__global__ __launch_bounds__(512, NT == 1 ? 4 : 2) void pattn_bwd_dkv_kernel(const MAttn p) {
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
    if constexpr (PF) sp.load(qsrc, dsrc, p.ldq, p.ldd, col0, min(CH, p.Tq), p.hd);
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
        if (!(p.tail & 1) || i0 == 0) {
            if constexpr (!PF) sp.load(qsrc + (int64_t)i0 * p.ldq, dsrc + (int64_t)i0 * p.ldd, p.ldq, p.ldd, col0, nt, p.hd);
            sp.commit(Qi, Di, p.scale);
        }
        for (int j = threadIdx.x; j < ((nt + 31) & ~31); j += blockDim.x) {
            const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + i0 + (j < nt ? j : 0);
            Ml[j] = j < nt ? p.lse[2 * stat] : INFINITY;              // +inf: a padded query row gets p = exp2(-inf) = 0
            Ll[j] = j < nt ? p.lse[2 * stat + 1] * kLog2e : 0.f;
            Dl[j] = j < nt ? p.delta[stat] : 0.f;
        }
        __syncthreads();
        if (PF && i0 + CH < p.Tq && !(p.tail & 1))
            sp.load(qsrc + (int64_t)(i0 + CH) * p.ldq, dsrc + (int64_t)(i0 + CH) * p.ldd, p.ldq, p.ldd, col0, min(CH, p.Tq - i0 - CH), p.hd);
        if (p.tail & 2) continue;
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
int g_fwd_qt = 2;
int g_bwd_nt = 1;
int g_ablate = 0;
int g_prefetch = 0;     // the next chunk's rows requested into registers under the current chunk's products: measured no faster (16 + registers)

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
    a.tail = g_ablate;
    const size_t lds = 2 * NBK * BLK + sizeof(float) * CH;
    if (g_fwd_qt == 2 && a.Tq > 128) {
        const unsigned grid = (unsigned)(a.B * a.H * ((a.Tq + 255) / 256)), bt = block_threads(a.Tq, 32);
        return g_prefetch ? launch(pattn_fwd_kernel<2, true>, grid, bt, lds, st, a) : launch(pattn_fwd_kernel<2, false>, grid, bt, lds, st, a);
    }
    const unsigned grid = (unsigned)(a.B * a.H * ((a.Tq + 127) / 128)), bt = block_threads(a.Tq, 16);
    return g_prefetch ? launch(pattn_fwd_kernel<1, true>, grid, bt, lds, st, a) : launch(pattn_fwd_kernel<1, false>, grid, bt, lds, st, a);
}

int pattn_backward(const MAttn& a0, hipStream_t st) {
    if (!pattn_backward_aligned(a0)) {
        set_error("plane attention backward: out / dout / dq / dk / dv must be 16-byte aligned with strides %% 4 == 0");
        return MSN_ERR_SHAPE;
    }
    MAttn a = a0;
    a.tail = g_ablate;
    {
        const size_t lds = 2 * NBK * BLK + sizeof(float) * CH;
        int rc;
        if (g_bwd_nt == 2 && a.Tq > 128) {
            const unsigned grid = (unsigned)(a.B * a.H * ((a.Tq + 255) / 256)), bt = block_threads(a.Tq, 32);
            rc = g_prefetch ? launch(pattn_bwd_dq_kernel<2, true>, grid, bt, lds, st, a) : launch(pattn_bwd_dq_kernel<2, false>, grid, bt, lds, st, a);
        } else {
            const unsigned grid = (unsigned)(a.B * a.H * ((a.Tq + 127) / 128)), bt = block_threads(a.Tq, 16);
            rc = g_prefetch ? launch(pattn_bwd_dq_kernel<1, true>, grid, bt, lds, st, a) : launch(pattn_bwd_dq_kernel<1, false>, grid, bt, lds, st, a);
        }
        if (rc) return rc;
    }
    const size_t lds = 2 * NBK * BLK + sizeof(float) * 3 * CH;
    if (g_bwd_nt == 2 && a.Tk > 128) {
        const unsigned grid = (unsigned)(a.B * a.H * ((a.Tk + 255) / 256)), bt = block_threads(a.Tk, 32);
        return g_prefetch ? launch(pattn_bwd_dkv_kernel<2, true>, grid, bt, lds, st, a) : launch(pattn_bwd_dkv_kernel<2, false>, grid, bt, lds, st, a);
    }
    const unsigned grid = (unsigned)(a.B * a.H * ((a.Tk + 127) / 128)), bt = block_threads(a.Tk, 16);
    return g_prefetch ? launch(pattn_bwd_dkv_kernel<1, true>, grid, bt, lds, st, a) : launch(pattn_bwd_dkv_kernel<1, false>, grid, bt, lds, st, a);
}

}  // namespace msn

using namespace msn;

extern "C" int msn_set_attention_planes(int mode) {
    MSN_REQUIRE(mode >= 0 && mode < 64, "msn_set_attention_planes: bit 0 on / off, bit 1 one query tile per wave in the forward, "
                "bit 2 two tiles per wave in the backward, bit 3 register prefetch of the next chunk, bits 4 - 5 diagnostic ablations");
    g_planes_on = mode & 1;
    g_fwd_qt = (mode & 2) ? 1 : 2;
    g_bwd_nt = (mode & 4) ? 2 : 1;
    g_prefetch = (mode & 8) ? 1 : 0;
    g_ablate = (mode >> 4) & 3;
    return MSN_OK;
}
