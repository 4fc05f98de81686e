// The feed-forward half of the reference's TransformerBlock -- x + Linear(4e -> e)(ReLU(Linear(e -> 4e)(x))), ref
// src/transformer_utils.py:102-106, 114 -- for the NARROW towers (emb 32: the spectrum transformer, 13 layers over 220 / 1024
// tokens) as ONE kernel per direction whose hidden activations never leave the chip.
//
// Unfused, the 4e-wide hidden matrix crosses HBM seven times per layer and step (written by ff.0, read by ff.2; read by the
// weight gradient of ff.2 and as the ReLU mask of its input gradient; the hidden gradient written, read by ff.0's weight and
// input gradients): 3.8 GB per layer at 1024 x 1024 tokens, ~0.92 ms of skinny fp32 GEMMs at half of their HBM roofline
// (profiles/r05_bench_convmixer_lc_sp_b1024_kernel_stats.csv).  Here the forward reads x and writes z; the backward reads x and
// dz, RECOMPUTES the hidden tile (same instructions on the same operands: the same bits, so the ReLU mask needs no storage at
// all) and writes dx plus per-workgroup partials of the four parameter gradients.
//
// Arithmetic: fp32 grade on the bf16 matrix cores, as the plane GEMMs (pgemm.hip) and the plane attention: every operand is three
// bf16 planes (x = x0 + x1 + x2 exactly), a product is the six plane products with pa + pb < 3, a0.b0 goes to its own
// accumulator (a zero accumulator folded in by the vector ALU where sums run on: the bf16 MFMA adds with a truncating rounding).
// The recomputation that loses at fp32-MFMA rates (r03: 470 vs 358 us per layer) costs a third of that on this pipe.
//
// Both weights stay in LDS for the life of a workgroup as the plane matrices msn_plane_split writes (32-row x 16-column blocks of
// three 1-KB images): W1 [4e][e] and W2^T [4e][e], 24 KB each at e = 32.  Row reads (ds_read_b128) of an image give the operand
// with the e-dimension on k, transposed reads (ds_read_b64_tr_b16) of the SAME image the operand with the hidden dimension on k.
//   forward:  a wave owns 32 tokens; H^T[j][n] = W1 . x^T lands with the hidden index on the REGISTERS of the lane that owns
//             token n, which is directly the B operand of out^T[i][n] += W2[i][j] . H^T[j][n] (no LDS round trip, no barrier).
//   backward: a wave owns a 32-wide hidden chunk (4 of them) of a 32-token group; two groups per workgroup.  H^T, dH^T = W2^T . dz^T
//             and dpre = dH o (H > 0) as above; the chunk's share of dx^T = W1^T . dpre^T meets the other chunks' in LDS (summed
//             in chunk order, + dz: the residual branch); the weight gradients reduce over TOKENS, so relu(H) and dpre go through
//             a 6-KB patch of LDS as [token][hidden] images and come back transposed, against transposed reads of the group's
//             staged x / dz images; each wave keeps its chunk's dW1 / dW2^T (32 x 32 each) in registers for the whole launch.
//   Partials of dW1, dW2, dc1, dc2 per (workgroup, group) are summed in a fixed order by a finishing launch.
#include <algorithm>

#include "msn_common.h"

namespace msn {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int PB = 1024;                        // one plane image of a block: [32 rows][16 bf16]
constexpr int BLK = 3 * PB;                     // a 32 x 16 block: three planes
constexpr int E = 32;                           // embedding width this file is built for
constexpr int HID = 4 * E;                      // ref transformer_utils.py:124 (ff_hidden_mult = 4)
constexpr int WBYTES = (HID / 32) * (E / 16) * BLK;      // a [HID][E] plane matrix: 24 KB
constexpr int NCH = HID / 32;                   // hidden chunks of 32

__device__ __forceinline__ float trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ unsigned hi_pack(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
struct Pair3 { unsigned p0, p1, p2; };
// a, b -> their three planes, packed pairwise (truncation split: every residual is exact in fp32)
__device__ __forceinline__ Pair3 split2(float a, float b) {
    const float ra = a - trunc16(a), rb = b - trunc16(b);
    const float sa = ra - trunc16(ra), sb = rb - trunc16(rb);
    return Pair3{hi_pack(a, b), hi_pack(ra, rb), hi_pack(sa, sb)};
}
struct Planes8 { u32x4 p0, p1, p2; };           // eight values of one lane as three fragments
__device__ __forceinline__ Planes8 split8(const float (&v)[8]) {
    Planes8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const Pair3 t = split2(v[2 * j], v[2 * j + 1]);
        o.p0[j] = t.p0, o.p1[j] = t.p1, o.p2[j] = t.p2;
    }
    return o;
}
__device__ __forceinline__ bf16x8 as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x4 mma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// D (+)= A . B^T over one instruction's 32 k from the operands' planes: a0.b0 into `big` through a zero accumulator (round to
// nearest when it is folded in), the five small products chained in `small`
__device__ __forceinline__ void plane_product(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4& big, f32x4& small) {
    big += mma(a[0], b[0], f32x4{0.f, 0.f, 0.f, 0.f});
    small = mma(a[2], b[0], small);
    small = mma(a[0], b[2], small);
    small = mma(a[1], b[1], small);
    small = mma(a[1], b[0], small);
    small = mma(a[0], b[1], small);
}
// Row fragments of a [rows][32 columns] plane matrix in LDS (row block `rb` = two column blocks): rows 16 t + c, columns 8 g .. + 7
// (k = the 32 columns): lane group g >> 1 picks the column block
__device__ __forceinline__ void row_frags32(const unsigned char* mat, int rb, int t, int c, int g, bf16x8 (&f)[3]) {
    const unsigned char* p = mat + (rb * 2 + (g >> 1)) * BLK + (16 * t + c) * 32 + (g & 1) * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(p + pl * PB);
}
// Transposed fragments of ONE block (32 rows x 16 columns): lane (c, g) gets column c of rows 4 g .. + 3 (elements 0 - 3) and
// 16 + 4 g .. + 3 (elements 4 - 7): k = the block's 32 rows, in the order in which accumulators hold two 16-row tiles
__device__ __forceinline__ void tr_frags(const unsigned char* blk, int c, int g, bf16x8 (&f)[3]) {
    const unsigned char* p = blk + (4 * g + (c >> 2)) * 32 + (c & 3) * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + pl * PB));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + pl * PB + 16 * 32));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        f[pl] = __builtin_bit_cast(bf16x8, v);
    }
}
__device__ __forceinline__ void planes_to_frags(const Planes8& p, bf16x8 (&f)[3]) {
    f[0] = as_frag(p.p0), f[1] = as_frag(p.p1), f[2] = as_frag(p.p2);
}
__device__ __forceinline__ int opaque_tid() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

struct FfnArgs {
    const float* x; int64_t ldx;                // [M][E]: the block's LayerNorm-1 output (ff input and residual)
    const float* dz; int64_t lddz;              // backward: gradient of z = ff(x) + x
    const unsigned char* w1p;                   // planes of W1  [HID][E]   (msn_plane_split)
    const unsigned char* w2tp;                  // planes of W2^T [HID][E]  (msn_plane_split, transposed)
    const float* c1; const float* c2;           // biases [HID], [E]
    float* z; int64_t ldz;                      // forward: ff(x) + x
    float* dx; int64_t lddx;                    // backward: dz + dpre . W1
    float* part;                                // backward: [slabs][PART] partial parameter gradients
    int64_t M;
};
constexpr int PART = 2 * HID * E + HID + E;     // dW1 [HID][E] | dW2^T [HID][E] | dc1 [HID] | dc2 [E]

// copy a [HID][E] plane matrix (global, 16-byte aligned) into LDS verbatim
__device__ __forceinline__ void load_weights(unsigned char* dst, const unsigned char* src) {
    for (int off = threadIdx.x * 16; off < WBYTES; off += blockDim.x * 16)
        *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(src + off);
}

// ------------------------------------------------------------------------------------------ forward
constexpr int FWD_LDS = 2 * WBYTES + (HID + E) * 4;
__global__ __launch_bounds__(512, 4) void ffn_fwd_kernel(const FfnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* W1 = smem;
    unsigned char* W2T = smem + WBYTES;
    float* C1 = reinterpret_cast<float*>(smem + 2 * WBYTES);
    float* C2 = C1 + HID;
    load_weights(W1, p.w1p);
    load_weights(W2T, p.w2tp);
    for (int i = threadIdx.x; i < HID + E; i += blockDim.x) C1[i] = i < HID ? p.c1[i] : p.c2[i - HID];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int64_t ntile = (p.M + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntile; tile += (int64_t)gridDim.x * 8) {
        const int64_t tok0 = tile * 32;
        // x rows of the wave's two 16-token tiles as operand fragments (token on the lane, e-dimension on k)
        bf16x8 xf[2][3];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t row = tok0 + 16 * nt + c;
            const bool ok = row < p.M;
            const float* src = p.x + (ok ? row : 0) * p.ldx + 8 * g;
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            const float v[8] = {ok ? a.x : 0.f, ok ? a.y : 0.f, ok ? a.z : 0.f, ok ? a.w : 0.f,
                                ok ? b.x : 0.f, ok ? b.y : 0.f, ok ? b.z : 0.f, ok ? b.w : 0.f};
            planes_to_frags(split8(v), xf[nt]);
        }
        f32x4 ob[2][2], os[2][2];               // out^T[i = 16 mt + 4 g + r][token c of tile nt]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) ob[mt][nt] = os[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int jc = 0; jc < NCH; ++jc) {
            float h[2][8];                      // relu(H)[token][hidden 32 jc + (4 g + r | 16 + 4 g + r)]
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                bf16x8 wf[3];
                row_frags32(W1, jc, jt, c, g, wf);
                const f32x4 bias = *reinterpret_cast<const f32x4*>(C1 + 32 * jc + 16 * jt + 4 * g);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 big = {0.f, 0.f, 0.f, 0.f}, small = big;
                    plane_product(wf, xf[nt], big, small);
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[nt][4 * jt + r] = fmaxf(big[r] + small[r] + bias[r], 0.f);
                }
            }
            bf16x8 hf[2][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) planes_to_frags(split8(h[nt]), hf[nt]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x8 w2f[3];                  // W2[i = 16 mt + c][hidden of the chunk, in the accumulators' order]
                tr_frags(W2T + (jc * 2 + mt) * BLK, c, g, w2f);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) plane_product(w2f, hf[nt], ob[mt][nt], os[mt][nt]);
            }
        }
        // z[n][i] = out + c2[i] + x[n][i]: 16 bytes per lane and (mt, nt)
        const int te = opaque_tid(), ce = te & 15, ge = (te >> 4) & 3;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t row = tok0 + 16 * nt + ce;
            if (row < p.M) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int i = 16 * mt + 4 * ge;
                    const f32x4 res = *reinterpret_cast<const f32x4*>(p.x + row * p.ldx + i);
                    const f32x4 b2 = *reinterpret_cast<const f32x4*>(C2 + i);
                    *reinterpret_cast<f32x4*>(p.z + row * p.ldz + i) = (ob[mt][nt] + os[mt][nt]) + b2 + res;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ backward
// LDS: W1 | W2^T | c1 | per group: dz image, x image (6 KB each) | per wave: a 6-KB patch | per group: dx shares of the 4 chunks
constexpr int IMG = 2 * BLK;                    // [32 tokens][32 columns]: two column blocks
constexpr int BWD_LDS = 2 * WBYTES + HID * 4 + 2 * 2 * IMG + 8 * IMG + 2 * 4 * 32 * E * 4;
__global__ __launch_bounds__(512, 2) void ffn_bwd_kernel(const FfnArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* W1 = smem;
    unsigned char* W2T = smem + WBYTES;
    float* C1 = reinterpret_cast<float*>(smem + 2 * WBYTES);
    unsigned char* Img = reinterpret_cast<unsigned char*>(C1 + HID);          // [group][dz | x][IMG]
    unsigned char* Patch = Img + 2 * 2 * IMG;                                 // [wave][IMG]
    float* Ex = reinterpret_cast<float*>(Patch + 8 * IMG);                    // [group][chunk][32 tokens][E]
    load_weights(W1, p.w1p);
    load_weights(W2T, p.w2tp);
    for (int i = threadIdx.x; i < HID; i += blockDim.x) C1[i] = p.c1[i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int jc = wave & 3, grp = wave >> 2;
    unsigned char* DZ = Img + grp * 2 * IMG;
    unsigned char* XI = DZ + IMG;
    unsigned char* patch = Patch + wave * IMG;
    float* ex = Ex + (grp * 4 + jc) * 32 * E;
    // persistent: the wave's chunk of dW1[j][i] and dW2^T[j][i] (j = 32 jc + 16 jt + 4 g + r, i = 16 it + c) and of dc1
    f32x4 w1b[2][2], w1s[2][2], w2b[2][2], w2s[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) w1b[a][b] = w1s[a][b] = w2b[a][b] = w2s[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dc1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                // this lane's tokens' share of dc1[32 jc + (4 g + r | 16 + 4 g + r)]
    f32x4 dc2 = {0.f, 0.f, 0.f, 0.f};                                       // the reducing thread's columns of dc2 (fixed per thread)
    const int64_t nit = (p.M + 63) >> 6;
    // this thread's piece of the staging: row (t & 255) >> 3 of its group, columns 4 (t & 7) .. + 3 of dz and of x.  The rows of the
    // NEXT iteration are requested right after this iteration's are staged (the loads' 1 - 2 us would otherwise stand between two
    // barriers of the only workgroup on the CU), and the dz piece is also this thread's residual term in the dx reduction below
    float4 d_nx = make_float4(0.f, 0.f, 0.f, 0.f), x_nx = d_nx;
    auto request = [&](int64_t it_) {
        const int t = opaque_tid();
        const int64_t row = it_ * 64 + (t >> 3);
        const bool ok = it_ < nit && row < p.M;
        d_nx = *reinterpret_cast<const float4*>(p.dz + (ok ? row : 0) * p.lddz + 4 * (t & 7));
        x_nx = *reinterpret_cast<const float4*>(p.x + (ok ? row : 0) * p.ldx + 4 * (t & 7));
        if (!ok) d_nx = x_nx = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    request(blockIdx.x);
    for (int64_t it = blockIdx.x; it < nit; it += gridDim.x) {
        const float4 d = d_nx, x = x_nx;
        __syncthreads();                        // the previous iteration's reads of the images / shares are done (first: weights in place)
        {   // stage the group's 32 rows of dz and x as plane images: 32 rows x 8 pieces of 4 columns = one piece per thread of the group
            const int t = opaque_tid() & 255, r = t >> 3, q = t & 7;
            const int off = (q >> 2) * BLK + r * 32 + (q & 3) * 8;
            const Pair3 da = split2(d.x, d.y), db = split2(d.z, d.w), xa = split2(x.x, x.y), xb = split2(x.z, x.w);
            *reinterpret_cast<u32x2*>(DZ + off) = u32x2{da.p0, db.p0};
            *reinterpret_cast<u32x2*>(DZ + off + PB) = u32x2{da.p1, db.p1};
            *reinterpret_cast<u32x2*>(DZ + off + 2 * PB) = u32x2{da.p2, db.p2};
            *reinterpret_cast<u32x2*>(XI + off) = u32x2{xa.p0, xb.p0};
            *reinterpret_cast<u32x2*>(XI + off + PB) = u32x2{xa.p1, xb.p1};
            *reinterpret_cast<u32x2*>(XI + off + 2 * PB) = u32x2{xa.p2, xb.p2};
        }
        request(it + gridDim.x);
        __syncthreads();
        bf16x8 xf[2][3], df[2][3];              // token rows as operand fragments (token on the lane, e-dimension on k)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            row_frags32(XI, 0, nt, c, g, xf[nt]);
            row_frags32(DZ, 0, nt, c, g, df[nt]);
        }
        float h[2][8], dp[2][8];                // relu(H), dpre = dH o (H > 0): [token tile][hidden 4 g + r | 16 + 4 g + r of the chunk]
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            bf16x8 wf[3], vf[3];
            row_frags32(W1, jc, jt, c, g, wf);
            row_frags32(W2T, jc, jt, c, g, vf);
            const f32x4 bias = *reinterpret_cast<const f32x4*>(C1 + 32 * jc + 16 * jt + 4 * g);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 big = {0.f, 0.f, 0.f, 0.f}, small = big, gb = big, gs = big;
                plane_product(wf, xf[nt], big, small);              // the forward's instructions on the forward's operands: the same bits
                plane_product(vf, df[nt], gb, gs);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre = big[r] + small[r] + bias[r];
                    h[nt][4 * jt + r] = fmaxf(pre, 0.f);
                    dp[nt][4 * jt + r] = pre > 0.f ? gb[r] + gs[r] : 0.f;
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 8; ++e) dc1[e] += dp[nt][e];
        // ---- the chunk's share of dx^T[i][n] = sum_j W1[j][i] dpre[n][j], and dpre as [token][hidden] images in the patch
        Planes8 dpp[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dpp[nt] = split8(dp[nt]);
        {
            bf16x8 dpf[2][3];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) planes_to_frags(dpp[nt], dpf[nt]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x8 w1t[3];                  // W1[hidden of the chunk][i = 16 mt + c] transposed: the e-dimension on m
                tr_frags(W1 + (jc * 2 + mt) * BLK, c, g, w1t);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 big = {0.f, 0.f, 0.f, 0.f}, small = big;
                    plane_product(w1t, dpf[nt], big, small);
                    *reinterpret_cast<f32x4*>(ex + (16 * nt + c) * E + 16 * mt + 4 * g) = big + small;
                }
            }
        }
        // ---- weight gradients: reductions over the group's 32 tokens.  relu(H) first, then dpre, through the wave's patch
        auto to_patch = [&](const Planes8 (&v)[2]) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {    // token 16 nt + c; hidden 4 g .. + 3 -> column block 0, 16 + 4 g .. + 3 -> column block 1
                unsigned char* d = patch + (16 * nt + c) * 32 + 8 * g;
                *reinterpret_cast<u32x2*>(d) = u32x2{v[nt].p0[0], v[nt].p0[1]};
                *reinterpret_cast<u32x2*>(d + PB) = u32x2{v[nt].p1[0], v[nt].p1[1]};
                *reinterpret_cast<u32x2*>(d + 2 * PB) = u32x2{v[nt].p2[0], v[nt].p2[1]};
                *reinterpret_cast<u32x2*>(d + BLK) = u32x2{v[nt].p0[2], v[nt].p0[3]};
                *reinterpret_cast<u32x2*>(d + BLK + PB) = u32x2{v[nt].p1[2], v[nt].p1[3]};
                *reinterpret_cast<u32x2*>(d + BLK + 2 * PB) = u32x2{v[nt].p2[2], v[nt].p2[3]};
            }
        };
        auto wgrad = [&](const unsigned char* other, f32x4 (&accb)[2][2], f32x4 (&accs)[2][2]) {
            // acc[jt][it][r] (+)= sum_tokens patch[token][j = 16 jt + 4 g + r] other[token][i = 16 it + c]
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                bf16x8 af[3];
                tr_frags(patch + jt * BLK, c, g, af);
#pragma unroll
                for (int itl = 0; itl < 2; ++itl) {
                    bf16x8 bfr[3];
                    tr_frags(other + itl * BLK, c, g, bfr);
                    plane_product(af, bfr, accb[jt][itl], accs[jt][itl]);
                }
            }
        };
        {
            Planes8 hp[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) hp[nt] = split8(h[nt]);
            to_patch(hp);
        }
        // (the patch is private to the wave: no workgroup barrier -- the LDS serves a wave's accesses in order; wave_barrier keeps the
        //  compiler from moving the reads above the stores or the next stores above the reads)
        __builtin_amdgcn_wave_barrier();
        wgrad(DZ, w2b, w2s);                    // dW2^T[j][i] += relu(H)[n][j] dz[n][i]
        __builtin_amdgcn_wave_barrier();
        to_patch(dpp);
        __builtin_amdgcn_wave_barrier();
        wgrad(XI, w1b, w1s);                    // dW1[j][i]   += dpre[n][j] x[n][i]
        __syncthreads();                        // the four chunks' shares of dx are in place
        {   // dx[n][i] = dz[n][i] + sum over the chunks, in chunk order: 64 tokens x 8 column quads = one float4 per thread
            const int t = opaque_tid(), n = t >> 3, q = t & 7, gsel = n >> 5;
            const int64_t row = it * 64 + n;
            const float* e0 = Ex + (gsel * 4) * 32 * E + (n & 31) * E + 4 * q;
            const f32x4 s = ((*reinterpret_cast<const f32x4*>(e0) + *reinterpret_cast<const f32x4*>(e0 + 32 * E)) +
                             *reinterpret_cast<const f32x4*>(e0 + 2 * 32 * E)) + *reinterpret_cast<const f32x4*>(e0 + 3 * 32 * E);
            if (row < p.M) {                    // (rows past M were staged as zeros: d is this thread's own dz piece)
                const f32x4 dv = {d.x, d.y, d.z, d.w};
                *reinterpret_cast<f32x4*>(p.dx + row * p.lddx + 4 * q) = s + dv;
                dc2 += dv;
            }
        }
    }
    // ---- publish the partials: slab (workgroup, group) = dW1 chunks | dW2^T chunks | dc1 | dc2 (group 0 carries dc2 of both)
    __syncthreads();
    float* slab = p.part + ((int64_t)blockIdx.x * 2 + grp) * PART;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int itl = 0; itl < 2; ++itl)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 32 * jc + 16 * jt + 4 * g + r, i = 16 * itl + c;
                slab[j * E + i] = w1b[jt][itl][r] + w1s[jt][itl][r];
                slab[HID * E + j * E + i] = w2b[jt][itl][r] + w2s[jt][itl][r];
            }
#pragma unroll
    for (int e = 0; e < 8; ++e) {               // over the 16 lanes that share g (the tokens)
        float v = dc1[e];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
        if (c == 0) slab[2 * HID * E + 32 * jc + 16 * (e >> 2) + 4 * g + (e & 3)] = v;
    }
    // dc2: thread t owned column quad t & 7 in every iteration; sum the 64 threads of each quad through LDS (the shares' area is free)
    float* red = Ex;
    *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = dc2;
    __syncthreads();
    if (threadIdx.x < E) {
        const int q = threadIdx.x >> 2, k = threadIdx.x & 3;
        float s = 0.f;                          // eight sums of eight (a 64-term chain costs a digit on 225 280 tokens)
        for (int a = 0; a < 8; ++a) {
            float t8 = 0.f;
            for (int u = 8 * a; u < 8 * a + 8; ++u) t8 += red[(u * 8 + q) * 4 + k];
            s += t8;
        }
        p.part[((int64_t)blockIdx.x * 2) * PART + 2 * HID * E + HID + threadIdx.x] = s;
        p.part[((int64_t)blockIdx.x * 2 + 1) * PART + 2 * HID * E + HID + threadIdx.x] = 0.f;
    }
}

// out = sum over slabs in a fixed order: dW1 | dW2 (transposed back to [E][HID]) | dc1 | dc2.  A block = 32 consecutive outputs x
// 8 slab groups (every 8th slab from g on: 16 interleaved chains per thread, then a tree), the groups combined through LDS in group
// order.  (One thread per output walking all 512 slabs -- 33 blocks on a 256-CU chip -- took ~130 us: the fixed cost of the backward.)
__global__ __launch_bounds__(256) void ffn_finish_kernel(const float* __restrict__ part, int nslabs, float* __restrict__ dw1,
                                                         float* __restrict__ dw2, float* __restrict__ dc1, float* __restrict__ dc2) {
    __shared__ float red[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + o;
    float acc[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = 0.f;
    if (i < PART) {
        for (int k = grp; k < nslabs; k += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (k + 8 * u < nslabs) acc[u] += part[(int64_t)(k + 8 * u) * PART + i];
        }
    }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    red[grp][o] = acc[0];
    __syncthreads();
    if (grp != 0 || i >= PART) return;
    const float s = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
    if (i < HID * E) dw1[i] = s;
    else if (i < 2 * HID * E) {
        const int r = i - HID * E, j = r / E, col = r % E;
        dw2[col * HID + j] = s;
    } else if (i < 2 * HID * E + HID) dc1[i - 2 * HID * E] = s;
    else dc2[i - 2 * HID * E - HID] = s;
}

int bwd_grid(int64_t M) { return (int)std::min<int64_t>(256, (M + 63) / 64); }

template <typename K>
int launch_lds(K kernel, int grid, size_t lds, hipStream_t st, const FfnArgs& a, const char* who) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        set_error("%s: cannot reserve %zu bytes of LDS", who, lds);
        return MSN_ERR_HIP;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(512), lds, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace msn

using namespace msn;

extern "C" int msn_ffn_supported(int64_t M, int emb, int hidden) { return M > 0 && emb == E && hidden == HID; }

extern "C" int msn_ffn_fwd(const float* x, int64_t ldx, int64_t M, int emb, int hidden, const void* w1_planes, const void* w2t_planes,
                           const float* c1, const float* c2, float* z, int64_t ldz, msn_stream_t stream) {
    MSN_REQUIRE(msn_ffn_supported(M, emb, hidden), "msn_ffn_fwd: emb %d / hidden %d (built for %d / %d)", emb, hidden, E, HID);
    MSN_REQUIRE(x && w1_planes && w2t_planes && c1 && c2 && z, "msn_ffn_fwd: null pointer");
    MSN_REQUIRE(aligned16(x) && aligned16(z) && aligned16(w1_planes) && aligned16(w2t_planes) && ldx % 4 == 0 && ldz % 4 == 0 &&
                    ldx >= E && ldz >= E, "msn_ffn_fwd: rows must be 16-byte aligned");
    FfnArgs a = {};
    a.x = x; a.ldx = ldx; a.w1p = static_cast<const unsigned char*>(w1_planes); a.w2tp = static_cast<const unsigned char*>(w2t_planes);
    a.c1 = c1; a.c2 = c2; a.z = z; a.ldz = ldz; a.M = M;
    const int grid = (int)std::min<int64_t>(512, (M + 255) / 256);      // two workgroups per CU, eight 32-token tiles per pass
    return launch_lds(ffn_fwd_kernel, grid, FWD_LDS, static_cast<hipStream_t>(stream), a, "msn_ffn_fwd");
}

extern "C" size_t msn_ffn_bwd_workspace_bytes(int64_t M, int emb, int hidden) {
    if (!msn_ffn_supported(M, emb, hidden)) return 0;
    return sizeof(float) * (size_t)bwd_grid(M) * 2 * PART;
}

extern "C" int msn_ffn_bwd(const float* x, int64_t ldx, const float* dz, int64_t lddz, int64_t M, int emb, int hidden,
                           const void* w1_planes, const void* w2t_planes, const float* c1, float* dx, int64_t lddx, float* dw1,
                           float* dc1, float* dw2, float* dc2, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(msn_ffn_supported(M, emb, hidden), "msn_ffn_bwd: emb %d / hidden %d (built for %d / %d)", emb, hidden, E, HID);
    MSN_REQUIRE(x && dz && w1_planes && w2t_planes && c1 && dx && dw1 && dc1 && dw2 && dc2, "msn_ffn_bwd: null pointer");
    MSN_REQUIRE(aligned16(x) && aligned16(dz) && aligned16(dx) && aligned16(w1_planes) && aligned16(w2t_planes) && ldx % 4 == 0 &&
                    lddz % 4 == 0 && lddx % 4 == 0 && ldx >= E && lddz >= E && lddx >= E, "msn_ffn_bwd: rows must be 16-byte aligned");
    const size_t need = msn_ffn_bwd_workspace_bytes(M, emb, hidden);
    MSN_REQUIRE(ws && ws_bytes >= need && aligned16(ws), "msn_ffn_bwd: workspace %zu < %zu bytes", ws_bytes, need);
    FfnArgs a = {};
    a.x = x; a.ldx = ldx; a.dz = dz; a.lddz = lddz; a.w1p = static_cast<const unsigned char*>(w1_planes);
    a.w2tp = static_cast<const unsigned char*>(w2t_planes); a.c1 = c1; a.dx = dx; a.lddx = lddx; a.part = static_cast<float*>(ws); a.M = M;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = bwd_grid(M);
    if (int rc = launch_lds(ffn_bwd_kernel, grid, BWD_LDS, st, a, "msn_ffn_bwd")) return rc;
    hipLaunchKernelGGL(ffn_finish_kernel, dim3((unsigned)cdiv(PART, 32)), dim3(256), 0, st, a.part, 2 * grid, dw1, dw2, dc1, dc2);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
