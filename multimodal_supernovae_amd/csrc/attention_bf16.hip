// Self-attention on the bf16 matrix cores for 64-wide heads and up to 256 tokens, no key mask -- the attention of the
// BASELINE cfg5 image tower (build-defined ViT-B/16: T = 197, 12 heads x 64; no reference counterpart -- the reference's
// SelfAttention, src/transformer_utils.py:36-89, is the fp32 path of attention.hip / attention_mfma.hip).
//
// Operands are bf16 in HBM: q | k | v side by side in one (B*T, 3 H 64) matrix as the qkv projection writes it, the
// output and all gradients bf16 in the layouts the neighbouring bf16-resident GEMMs (gemm_bf16res.hip) read.  Scores,
// softmax statistics and every accumulation are fp32 (v_mfma_f32_16x16x32_bf16).
//
// One workgroup (8 waves) per (batch, head); the token matrices the whole workgroup streams over sit in LDS as
// [token][64] bf16 images (128-byte rows, written by LDS-DMA with the XOR swizzle of the GEMM images on the SOURCE
// address), the 16-token tiles a wave owns come straight from global memory into MFMA fragments.  Products that contract
// over the head dimension read row fragments (ds_read_b128); products that contract over TOKENS read the same images
// transposed (ds_read_b64_tr_b16), and their other operand is the probability / score-gradient tile still sitting in the
// accumulator registers of the product that made it (converted to bf16 in place: no LDS round trip) -- its k order is the
// accumulator's (rows 4g..4g+3 of two 16-row tiles), and the transposed reads fetch the matching rows.
//   forward:   S^T = K Q^T  -> online softmax over 32-key blocks ->  O^T += V^T P^T
//   dQ pass:   S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta) ->  dQ^T += K^T dS^T         (waves own query tiles)
//   dK,dV pass: S = Q K^T, dP = dO V^T (lane = key)             ->  dV^T += dO^T P, dK^T += Q^T dS  (waves own key tiles)
#include <algorithm>
#include <math.h>

#include "msn_common.h"

namespace msn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef __attribute__((address_space(1))) const void agptr_t;
typedef __attribute__((address_space(3))) void alptr_t;

constexpr int AHD = 64;                 // head width
constexpr int AMAXT = 256;              // tokens an LDS image holds
constexpr int AIMG = AMAXT * AHD * 2;   // 32 KB
constexpr float ALOG2E = 1.4426950408889634f;
constexpr int AWAVES = 8;               // waves per workgroup: the 16-token tiles of a (batch, head) are dealt round-robin

__device__ __align__(16) const unsigned char g_attn_zero_page[16] = {0};

struct BAttn {
    const u16* qkv; int64_t ld;         // [B*T][ld]: q at column 64 h, k at E + 64 h, v at 2E + 64 h  (E = 64 H)
    u16* out; int64_t ldo;              // fwd: attention output [B*T][ldo], head h at column 64 h.  bwd: the saved output
    const u16* dout; int64_t ldd;       // bwd: gradient of the output
    u16* dqkv;                          // bwd: [B*T][ld], same layout as qkv
    float* lse;                         // [B][H][T] natural-log sum-exp of the scaled scores
    float* delta;                       // bwd scratch [B][H][T]: rowsum(dO o O)
    float* colpart;                     // bwd (nullable): [B][3 E] column sums of dq | dk | dv over the tokens of a sample
    int B, H, T;
    float scale;
};

__device__ __forceinline__ int aswz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ u16 abf(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const u16*>(&b);
}
__device__ __forceinline__ float afl(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ void a_read128(bf16x8& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); }
__device__ __forceinline__ void a_read_tr(bf16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}
// (the same with an immediate byte offset: the tile index of an image read is added by the instruction, not by the vector ALU)
template <int OFF>
__device__ __forceinline__ void a_read128_o(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void a_read_tr_o(bf16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void a_wait_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Stage tokens [0, T) of one 64-wide column block (starting at `src`, row stride ld elements) into a [256][64] LDS image:
// piece = 8 rows x 128 B, lane i of a piece -> row i / 8, chunk slot i % 8 <- source chunk slot ^ swz(row); rows >= T read
// zeros (they enter products as operands: must be finite).  Pieces beyond the padded length are skipped.
__device__ __forceinline__ void stage_image(unsigned char* img, const u16* __restrict__ src, int64_t ld, int T, int wave, int lane) {
    const int rows_needed = (T + 31) / 32 * 32;
#pragma unroll
    for (int q = 0; q < 32 / AWAVES; ++q) {
        const int piece = (32 / AWAVES) * wave + q;       // 32 pieces x 8 rows = 256 rows, dealt to the waves in runs
        if (piece * 8 >= rows_needed) break;
        const int R = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ aswz(R);
        const u16* s = R < T ? src + (int64_t)R * ld + 8 * c : reinterpret_cast<const u16*>(g_attn_zero_page);
        __builtin_amdgcn_global_load_lds((agptr_t*)s, (alptr_t*)(img + piece * 1024), 16, 0, 0);
    }
}

// Row fragment (operand with the head dimension on k) of 16 tokens straight from global memory: lane (token l15, k group
// g) <- 8 consecutive d at 32 kk + 8 g.  Tokens past T are clamped (their results are never stored / are masked).
__device__ __forceinline__ void load_rows(bf16x8 (&f)[2], const u16* __restrict__ base, int64_t ld, int tok0, int T, int l15, int g) {
    const int t = min(tok0 + l15, T - 1);
    const u16* p = base + (int64_t)t * ld + 8 * g;
    f[0] = *reinterpret_cast<const bf16x8*>(p);
    f[1] = *reinterpret_cast<const bf16x8*>(p + 32);
}
// LDS byte offset of the row fragment of token tile `tile` (16 tokens) at k32 half kk inside an image
__device__ __forceinline__ unsigned row_frag_addr(int tile, int kk, int l15, int g) {
    return (unsigned)((16 * tile + l15) * 128 + 16 * ((4 * kk + g) ^ (l15 >> 1)));
}
// LDS byte offset a lane supplies to a transposed read of 4 tokens x 16 d: tokens 16 tile + 4 g + tq, d = 16 dt + 4 tp ..
__device__ __forceinline__ unsigned tr_frag_addr(int tile, int dt, int g, int tq, int tp) {
    const int row = 16 * tile + 4 * g + tq;
    const int chunk = (2 * dt + (tp >> 1)) ^ ((2 * g + (tq >> 1)) & 7);
    return (unsigned)(row * 128 + 16 * chunk + 8 * (tp & 1));
}
__device__ __forceinline__ float group_max4(float v) {      // over the 4 lanes that share l15 (g = 0..3)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ bf16x8 pack8(const float (&a)[4], const float (&b)[4]) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = (__bf16)a[i];
        r[4 + i] = (__bf16)b[i];
    }
    return r;
}

// Column sums (over the tokens of this (batch, head)) of a gradient tile a wave has just stored: v[dt][r] = this lane's
// value for d = 16 dt + 4 g + r of token l15 (0 for tokens past T).  The 16 lanes sharing g are reduced by an xor tree and
// added to the WAVE's own row of an LDS table (single writer: deterministic, no atomics, no registers kept across tiles:
// a second set of accumulators would cost the second workgroup of the CU); the rows are summed in wave order at the end.
__device__ __forceinline__ void add_tile_colsum(const float (&v)[4][4], float* __restrict__ wrow, int l15, int g) {
    // reduce-scatter over the 16 lanes: each halving step trades the half of the values the partner keeps (8 + 4 + 2 + 1
    // shuffles instead of 16 x 4); lane l15 ends up with the 16-lane sum of value index l15 = 4 dt + r
    float w8[8], w4[4], w2[2];
    const bool b3 = l15 & 8, b2 = l15 & 4, b1 = l15 & 2, b0 = l15 & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float lo = v[i >> 2][i & 3], hi = v[2 + (i >> 2)][i & 3];
        w8[i] = (b3 ? hi : lo) + __shfl_xor(b3 ? lo : hi, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = (b2 ? w8[i + 4] : w8[i]) + __shfl_xor(b2 ? w8[i] : w8[i + 4], 4, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = (b1 ? w4[i + 2] : w4[i]) + __shfl_xor(b1 ? w4[i] : w4[i + 2], 2, 64);
    const float t = (b0 ? w2[1] : w2[0]) + __shfl_xor(b0 ? w2[0] : w2[1], 1, 64);
    // no-return LDS add (ds_add_f32): nothing waits for the round trip; one writer per address
    __hip_atomic_fetch_add(&wrow[16 * (l15 >> 2) + 4 * g + (l15 & 3)], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
__device__ __forceinline__ void publish_colsum(const float* __restrict__ table, float* __restrict__ out) {
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < AWAVES; ++w) t += table[w * 64 + threadIdx.x];
        out[threadIdx.x] = t;
    }
}

// ------------------------------------------------------------------------------------------------------- forward
__global__ __launch_bounds__(64 * AWAVES) void battn_fwd_kernel(const BAttn p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * AIMG];      // K image | V image
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, E = p.H * AHD;
    const int l15 = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const u16* base = p.qkv + (int64_t)b * T * p.ld + h * AHD;
    stage_image(lds, base + E, p.ld, T, wave, lane);
    stage_image(lds + AIMG, base + 2 * E, p.ld, T, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned kimg = (unsigned)(uintptr_t)(alptr_t*)lds, vimg = kimg + AIMG;
    const int ntile = (T + 15) / 16, nblk = (T + 31) / 32;
    const float sl = p.scale * ALOG2E;

    // Lane addresses of the image reads, once per kernel: the XOR swizzle depends on the row INSIDE a 16-token tile only, so tile t of an
    // image is +2048 t bytes -- an immediate offset of the read -- and a 32-key block +4096 j: one vector add per address and block.
    // (Recomputed per read they were 24 of the ~165 vector instructions per block against 8 MFMAs; the kernel's vector-instruction
    //  issue time was 3.5 x its matrix time: profiles/r06_battn_counters.txt.)
    const unsigned kb0 = kimg + row_frag_addr(0, 0, l15, g), kb1 = kimg + row_frag_addr(0, 1, l15, g);
    unsigned vb[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vb[dt] = vimg + tr_frag_addr(0, dt, g, tq, tp);
    const bool ragged = (T & 31) != 0;                    // the last 32-key block holds keys past the sequence

    for (int qt = wave; qt < ntile; qt += AWAVES) {
        bf16x8 qf[2];
        load_rows(qf, base, p.ld, 16 * qt, T, l15, g);
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, l = 0.f;                       // running max (log2 domain); this LANE's share of the sum (its keys 4 g .. + 3)
        for (int j = 0; j < nblk; ++j) {
            // scores of 32 keys (tiles 2j, 2j+1): S^T tile = K_tile . Q^T -> lane: query l15, keys 4g..4g+3 of each tile
            const unsigned boff = (unsigned)j * 4096u;
            bf16x8 kf[2][2];
            a_read128_o<0>(kf[0][0], kb0 + boff);
            a_read128_o<0>(kf[0][1], kb1 + boff);
            a_read128_o<2048>(kf[1][0], kb0 + boff);
            a_read128_o<2048>(kf[1][1], kb1 + boff);
            // V^T fragments of the same 32 keys for the four d tiles (transposed reads), requested early
            bf16x4 vf[4][2];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                a_read_tr_o<0>(vf[dt][0], vb[dt] + boff);
                a_read_tr_o<2048>(vf[dt][1], vb[dt] + boff);
            }
            a_wait_lds();
            f32x4 acc[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                acc[t2] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t2][0], qf[0], acc[t2], 0, 0, 0);
                acc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t2][1], qf[1], acc[t2], 0, 0, 0);
            }
            if (ragged && j == nblk - 1) {                  // (uniform) keys past the sequence score -inf: p = 0
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (32 * j + 16 * t2 + 4 * g + r >= T) acc[t2][r] = -INFINITY;
            }
            // block maximum on the raw scores (the scale is positive), scaled once
            float bm = fmaxf(fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3])),
                             fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3])));
            bm = group_max4(bm) * sl;                        // finite: every block holds at least one real key
            if (__any(bm > m)) {                             // (wave-uniform; rare once the running maximum has settled)
                const float mn = fmaxf(m, bm);
                const float f = __builtin_amdgcn_exp2f(m - mn);   // 0 on the first block (m = -inf)
                m = mn;
                l *= f;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] *= f;
            }
            // p = exp2(s sl - m): one multiply-add and one v_exp_f32 per score (exp2f's range handling was five instructions more)
            float s[2][4];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[t2][r] = __builtin_amdgcn_exp2f(fmaf(acc[t2][r], sl, -m));
                    l += s[t2][r];
                }
            const bf16x8 pf = pack8(s[0], s[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vv = __builtin_shufflevector(vf[dt][0], vf[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vv, pf, o[dt], 0, 0, 0);
            }
        }
        l = group_sum4(l);                                   // the four lanes of a query hold its keys 4 g .. + 3 of every tile
        // lane: query 16 qt + l15, d = 16 dt + 4 g .. + 3
        const int q = 16 * qt + l15;
        if (q < T) {
            const float inv = 1.f / l;
            u16* op = p.out + ((int64_t)b * T + q) * p.ldo + h * AHD + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                ushort4 v;
                v.x = abf(o[dt][0] * inv); v.y = abf(o[dt][1] * inv); v.z = abf(o[dt][2] * inv); v.w = abf(o[dt][3] * inv);
                *reinterpret_cast<ushort4*>(op + 16 * dt) = v;
            }
            if (g == 0) p.lse[((int64_t)b * p.H + h) * T + q] = (m + log2f(l)) * (1.f / ALOG2E);
        }
    }
}

// ------------------------------------------------------------------------------------------------------- dQ pass
__global__ __launch_bounds__(64 * AWAVES) void battn_bwd_dq_kernel(const BAttn p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * AIMG];      // K image | V image
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, E = p.H * AHD;
    const int l15 = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const u16* base = p.qkv + (int64_t)b * T * p.ld + h * AHD;
    stage_image(lds, base + E, p.ld, T, wave, lane);
    stage_image(lds + AIMG, base + 2 * E, p.ld, T, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned kimg = (unsigned)(uintptr_t)(alptr_t*)lds, vimg = kimg + AIMG;
    const int ntile = (T + 15) / 16, nblk = (T + 31) / 32;
    const float sl = p.scale * ALOG2E;

    __shared__ float cst[AWAVES * 64];                   // per-wave column sums of dq (bias gradient), see add_tile_colsum
    if (lane < 64) cst[wave * 64 + lane] = 0.f;
    // lane addresses of the image reads, once per kernel; tile t = + 2048 t (immediate), block j = + 4096 j (see the forward)
    const unsigned kb0 = kimg + row_frag_addr(0, 0, l15, g), kb1 = kimg + row_frag_addr(0, 1, l15, g);
    unsigned tb[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) tb[dt] = kimg + tr_frag_addr(0, dt, g, tq, tp);
    const bool ragged = (T & 31) != 0;
    for (int qt = wave; qt < ntile; qt += AWAVES) {
        bf16x8 qf[2], df[2], of[2];
        load_rows(qf, base, p.ld, 16 * qt, T, l15, g);
        load_rows(df, p.dout + (int64_t)b * T * p.ldd + h * AHD, p.ldd, 16 * qt, T, l15, g);
        load_rows(of, p.out + (int64_t)b * T * p.ldo + h * AHD, p.ldo, 16 * qt, T, l15, g);
        const int q = 16 * qt + l15;
        const int64_t stat = ((int64_t)b * p.H + h) * T + min(q, T - 1);
        // delta_q = sum_d dO[q][d] O[q][d]: 16 of the 64 terms in this lane, the rest in the 3 lanes sharing l15
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 8; ++i) dl += (float)df[kk][i] * (float)of[kk][i];
        dl = group_sum4(dl);
        if (g == 0 && q < T) p.delta[stat] = dl;
        const float lse2 = p.lse[stat] * ALOG2E;
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float dls = dl * p.scale;
        for (int j = 0; j < nblk; ++j) {
            const unsigned boff = (unsigned)j * 4096u;
            bf16x8 kf[2][2], vf[2][2];
            a_read128_o<0>(kf[0][0], kb0 + boff);
            a_read128_o<0>(kf[0][1], kb1 + boff);
            a_read128_o<2048>(kf[1][0], kb0 + boff);
            a_read128_o<2048>(kf[1][1], kb1 + boff);
            a_read128_o<AIMG>(vf[0][0], kb0 + boff);            // (the V image follows the K image)
            a_read128_o<AIMG>(vf[0][1], kb1 + boff);
            a_read128_o<AIMG + 2048>(vf[1][0], kb0 + boff);
            a_read128_o<AIMG + 2048>(vf[1][1], kb1 + boff);
            bf16x4 kt[4][2];                                    // K^T fragments (transposed reads) for dQ^T += K^T dS^T
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                a_read_tr_o<0>(kt[dt][0], tb[dt] + boff);
                a_read_tr_o<2048>(kt[dt][1], tb[dt] + boff);
            }
            a_wait_lds();
            float ds[2][4];
            const bool mask_now = ragged && j == nblk - 1;      // (uniform) keys past the sequence: p = 0
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t2][0], qf[0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t2][1], qf[1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t2][0], df[0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t2][1], df[1], dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(fmaf(s[r], sl, -lse2));       // (one multiply-add + v_exp_f32: exp2f carries range handling)
                    if (mask_now && 32 * j + 16 * t2 + 4 * g + r >= T) pr = 0.f;
                    ds[t2][r] = pr * fmaf(dp[r], p.scale, -dls);
                }
            }
            const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kv = __builtin_shufflevector(kt[dt][0], kt[dt][1], 0, 1, 2, 3, 4, 5, 6, 7);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kv, dsf, dq[dt], 0, 0, 0);
            }
        }
        float stored[4][4];
        u16* op = p.dqkv + ((int64_t)b * T + min(q, T - 1)) * p.ld + h * AHD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            ushort4 v;
            v.x = abf(dq[dt][0]); v.y = abf(dq[dt][1]); v.z = abf(dq[dt][2]); v.w = abf(dq[dt][3]);
            if (q < T) *reinterpret_cast<ushort4*>(op + 16 * dt) = v;
            stored[dt][0] = q < T ? afl(v.x) : 0.f; stored[dt][1] = q < T ? afl(v.y) : 0.f;
            stored[dt][2] = q < T ? afl(v.z) : 0.f; stored[dt][3] = q < T ? afl(v.w) : 0.f;
        }
        if (p.colpart) add_tile_colsum(stored, cst + wave * 64, l15, g);   // sums of the values as stored
    }
    if (p.colpart) {   // bias gradient of the packed projection
        __syncthreads();
        publish_colsum(cst, p.colpart + (int64_t)b * 3 * E + h * AHD);
    }
}

// ------------------------------------------------------------------------------------------------------- dK, dV pass
__global__ __launch_bounds__(64 * AWAVES) void battn_bwd_dkv_kernel(const BAttn p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * AIMG];      // Q image | dO image
    __shared__ __attribute__((aligned(16))) float stat_l[AMAXT], stat_d[AMAXT];      // lse * log2e (+inf past the sequence), delta * scale per query
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, E = p.H * AHD;
    const int l15 = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    const u16* base = p.qkv + (int64_t)b * T * p.ld + h * AHD;
    stage_image(lds, base, p.ld, T, wave, lane);
    stage_image(lds + AIMG, p.dout + (int64_t)b * T * p.ldd + h * AHD, p.ldd, T, wave, lane);
    for (int i = threadIdx.x; i < AMAXT; i += 64 * AWAVES) {
        const int64_t st = ((int64_t)b * p.H + h) * T + min(i, T - 1);
        stat_l[i] = i < T ? p.lse[st] * ALOG2E : INFINITY;       // a query past the sequence: p = exp2(-inf) = 0, no mask in the loop
        stat_d[i] = i < T ? p.delta[st] * p.scale : 0.f;         // (delta arrives scaled: ds = p (dp scale - delta scale))
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned qimg = (unsigned)(uintptr_t)(alptr_t*)lds, dimg = qimg + AIMG;
    const int ntile = (T + 15) / 16, nblk = (T + 31) / 32;
    const float sl = p.scale * ALOG2E;

    __shared__ float cst[2 * AWAVES * 64];               // per-wave column sums of dk | dv, see add_tile_colsum
    cst[wave * 64 + lane] = 0.f;
    cst[AWAVES * 64 + wave * 64 + lane] = 0.f;
    // lane addresses of the image reads, once per kernel; tile t = + 2048 t (immediate), block j = + 4096 j, dO image = + AIMG (see the forward)
    const unsigned qb0 = qimg + row_frag_addr(0, 0, l15, g), qb1 = qimg + row_frag_addr(0, 1, l15, g);
    unsigned tb[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) tb[dt] = qimg + tr_frag_addr(0, dt, g, tq, tp);
    for (int kt = wave; kt < ntile; kt += AWAVES) {
        bf16x8 kf[2], vf[2];
        load_rows(kf, base + E, p.ld, 16 * kt, T, l15, g);
        load_rows(vf, base + 2 * E, p.ld, 16 * kt, T, l15, g);
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nblk; ++j) {                        // 32 queries per step
            const unsigned boff = (unsigned)j * 4096u;
            bf16x8 qf[2][2], df[2][2];
            a_read128_o<0>(qf[0][0], qb0 + boff);
            a_read128_o<0>(qf[0][1], qb1 + boff);
            a_read128_o<2048>(qf[1][0], qb0 + boff);
            a_read128_o<2048>(qf[1][1], qb1 + boff);
            a_read128_o<AIMG>(df[0][0], qb0 + boff);
            a_read128_o<AIMG>(df[0][1], qb1 + boff);
            a_read128_o<AIMG + 2048>(df[1][0], qb0 + boff);
            a_read128_o<AIMG + 2048>(df[1][1], qb1 + boff);
            a_wait_lds();
            float pr[2][4], ds[2][4];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                // S tile = Q_tile . K^T -> lane: key l15, queries 16 (2j + t2) + 4 g + r
                f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[t2][0], kf[0], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[t2][1], kf[1], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[t2][0], vf[0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[t2][1], vf[1], dp, 0, 0, 0);
                const f32x4 sl4 = *reinterpret_cast<const f32x4*>(stat_l + 32 * j + 16 * t2 + 4 * g);
                const f32x4 sd4 = *reinterpret_cast<const f32x4*>(stat_d + 32 * j + 16 * t2 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], sl, -sl4[r]));   // (one multiply-add + v_exp_f32; -inf past the sequence)
                    pr[t2][r] = pv;
                    ds[t2][r] = pv * fmaf(dp[r], p.scale, -sd4[r]);
                }
            }
            const bf16x8 pf = pack8(pr[0], pr[1]), dsf = pack8(ds[0], ds[1]);
            // Q^T and dO^T fragments of the same 32 queries (transposed reads), two d tiles at a time: all four at once
            // cost 16 more registers and the second workgroup of the CU (142 VGPRs -> 3 waves per SIMD)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                bf16x4 qT[2][2], dT[2][2];
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    a_read_tr_o<0>(qT[d2][0], tb[2 * half + d2] + boff);
                    a_read_tr_o<2048>(qT[d2][1], tb[2 * half + d2] + boff);
                    a_read_tr_o<AIMG>(dT[d2][0], tb[2 * half + d2] + boff);
                    a_read_tr_o<AIMG + 2048>(dT[d2][1], tb[2 * half + d2] + boff);
                }
                a_wait_lds();
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    const int dt = 2 * half + d2;
                    const bf16x8 dov = __builtin_shufflevector(dT[d2][0], dT[d2][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 qv = __builtin_shufflevector(qT[d2][0], qT[d2][1], 0, 1, 2, 3, 4, 5, 6, 7);
                    dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dov, pf, dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qv, dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        const int key = 16 * kt + l15;
        const bool live = key < T;
        u16* op = p.dqkv + ((int64_t)b * T + min(key, T - 1)) * p.ld + h * AHD + 4 * g;
        float stored[4][4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            ushort4 a;
            a.x = abf(dk[dt][0]); a.y = abf(dk[dt][1]); a.z = abf(dk[dt][2]); a.w = abf(dk[dt][3]);
            if (live) *reinterpret_cast<ushort4*>(op + E + 16 * dt) = a;
            stored[dt][0] = live ? afl(a.x) : 0.f; stored[dt][1] = live ? afl(a.y) : 0.f;
            stored[dt][2] = live ? afl(a.z) : 0.f; stored[dt][3] = live ? afl(a.w) : 0.f;
        }
        if (p.colpart) add_tile_colsum(stored, cst + wave * 64, l15, g);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            ushort4 c;
            c.x = abf(dv[dt][0]); c.y = abf(dv[dt][1]); c.z = abf(dv[dt][2]); c.w = abf(dv[dt][3]);
            if (live) *reinterpret_cast<ushort4*>(op + 2 * E + 16 * dt) = c;
            stored[dt][0] = live ? afl(c.x) : 0.f; stored[dt][1] = live ? afl(c.y) : 0.f;
            stored[dt][2] = live ? afl(c.z) : 0.f; stored[dt][3] = live ? afl(c.w) : 0.f;
        }
        if (p.colpart) add_tile_colsum(stored, cst + AWAVES * 64 + wave * 64, l15, g);
    }
    if (p.colpart) {
        __syncthreads();
        publish_colsum(cst, p.colpart + (int64_t)b * 3 * E + E + h * AHD);
        publish_colsum(cst + AWAVES * 64, p.colpart + (int64_t)b * 3 * E + 2 * E + h * AHD);
    }
}

// out[n] = sum over the batch of part[b][n]: 64 columns x 16 sample groups per workgroup (group rg sums samples rg, rg + 16, ...: eight
// independent loads per wait), the group sums combined through LDS in a fixed order.  (One thread per column walked all 512 samples:
// 64 dependent round trips on nine workgroups, 40 us per launch.)
__global__ __launch_bounds__(1024) void battn_colsum_finish_kernel(const float* __restrict__ part, int nparts, int N, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (n < N) {
        const int mine = (nparts - rg + 15) / 16;
        for (int k0 = 0; k0 < mine; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(int64_t)(rg + 16 * min(k0 + j, mine - 1)) * N + n];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < mine) s += v[j];
        }
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && n < N) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][cl];
        out[n] = t;
    }
}

static int check_battn(const char* who, int B, int H, int T, int64_t ld, int64_t ldo) {
    MSN_REQUIRE(B > 0 && H > 0 && T > 0 && T <= AMAXT, "%s: 1 <= T <= 256 tokens (got %d), B, H > 0", who, T);
    MSN_REQUIRE(ld >= 3 * H * AHD && ld % 8 == 0 && ldo >= H * AHD && ldo % 8 == 0, "%s: bad leading dimensions", who);
    MSN_REQUIRE((int64_t)B * H <= 0x7fffffffLL, "%s: too many (batch, head) pairs", who);
    return MSN_OK;
}

}  // namespace msn

using namespace msn;

extern "C" int msn_attention_bf16_fwd(const void* qkv, int64_t ld, int B, int H, int T, float scale, void* out, int64_t ldo,
                                      float* lse, msn_stream_t stream) {
    if (int rc = check_battn("msn_attention_bf16_fwd", B, H, T, ld, ldo)) return rc;
    MSN_REQUIRE(qkv && out && lse && (reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                "msn_attention_bf16_fwd: null / misaligned pointer");
    BAttn a = {};
    a.qkv = static_cast<const u16*>(qkv); a.ld = ld; a.out = static_cast<u16*>(out); a.ldo = ldo; a.lse = lse;
    a.B = B; a.H = H; a.T = T; a.scale = scale;
    hipLaunchKernelGGL(battn_fwd_kernel, dim3((unsigned)(B * H)), dim3(64 * AWAVES), 0, static_cast<hipStream_t>(stream), a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// bias gradient of the packed projection from inside the passes: colsum_out (3 E floats, nullable) = column sums of dqkv;
// colsum_ws = B x 3 E floats of per-sample partials (summed over the batch in a fixed order)
extern "C" int msn_attention_bf16_bwd(const void* qkv, int64_t ld, const void* out, int64_t ldo, const void* dout, int64_t ldd,
                                      const float* lse, int B, int H, int T, float scale, void* dqkv, float* delta,
                                      float* colsum_out, float* colsum_ws, msn_stream_t stream) {
    if (int rc = check_battn("msn_attention_bf16_bwd", B, H, T, ld, ldo)) return rc;
    MSN_REQUIRE(qkv && out && dout && lse && dqkv && delta && ldd >= H * AHD && ldd % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(qkv) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(dout) & 15) == 0 && (reinterpret_cast<uintptr_t>(dqkv) & 7) == 0,
                "msn_attention_bf16_bwd: null / misaligned pointer");
    BAttn a = {};
    a.qkv = static_cast<const u16*>(qkv); a.ld = ld; a.out = const_cast<u16*>(static_cast<const u16*>(out)); a.ldo = ldo;
    a.dout = static_cast<const u16*>(dout); a.ldd = ldd; a.dqkv = static_cast<u16*>(dqkv); a.lse = const_cast<float*>(lse);
    a.delta = delta; a.B = B; a.H = H; a.T = T; a.scale = scale;
    MSN_REQUIRE(!colsum_out || colsum_ws, "msn_attention_bf16_bwd: column sums need the (B, 3 E) workspace");
    a.colpart = colsum_out ? colsum_ws : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(battn_bwd_dq_kernel, dim3((unsigned)(B * H)), dim3(64 * AWAVES), 0, st, a);      // also writes delta
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(battn_bwd_dkv_kernel, dim3((unsigned)(B * H)), dim3(64 * AWAVES), 0, st, a);
    MSN_LAUNCH_CHECK();
    if (colsum_out) {
        hipLaunchKernelGGL(battn_colsum_finish_kernel, dim3((unsigned)cdiv(3 * H * AHD, 64)), dim3(1024), 0, st, colsum_ws, B,
                           3 * H * AHD, colsum_out);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}
