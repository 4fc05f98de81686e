// Matrix-core attention (head_dim up to 128 in multiples of 16 -- 80..128: eight waves with 256 registers each; the ViT image
// tower; head_dim 4 / 8 / 12: the light-curve transformer, run as 16-wide heads whose missing columns
// are zeros in LDS / registers only).  Exact fp32 on v_mfma_f32_16x16x4_f32.
//
// One workgroup per (batch, head); one wave per 16-row tile of the "fixed" operand, every wave
// sweeping the 16-row tiles of the "streamed" operand, which sits in LDS for the whole workgroup:
//   forward / dQ kernel : fixed = a query tile (fragments in registers), streamed = K and V
//   dK,dV kernel        : fixed = a key tile,                            streamed = Q and dO
// Every product is oriented so that the accumulator of one MFMA chain is directly the A operand of
// the next chain (no transposes, no LDS round trip for P or dS):
//   scores   T[srow][fcol] = sum_d Streamed[srow][d] * Fixed[fcol][d]   (A = LDS rows, B = registers)
//            -> lane (c = lane & 15, g = lane >> 4) holds T[4g + r][c], r = 0..3
//   outputs  O[fcol][d]   += sum_srow T[srow][fcol] * Streamed2[srow][d] (A = T registers: the MFMA k
//            index is the lane group g, which is exactly how T is spread; B = one LDS row per g)
// Softmax statistics of a query are lane-local in the forward / dQ kernels (query = lane column;
// 4 registers per tile + two cross-group shuffles per reduction).  The forward makes two passes over
// the key tiles (row maximum, then exponentials + P.V), recomputing the cheap 16x16 score tiles
// instead of holding T/16 accumulator tiles in registers.
// K-order inside a d-step is free, so lane group g owns d = 16x + 4g .. +3 and fetches them with ONE
// 16-byte LDS read per 4 MFMAs (rows padded by 4 floats).
#include <algorithm>
#include <math.h>

#include "msn_common.h"
#include "attention_args.h"

namespace msn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kFill = -1e7f;  // ref transformer_utils.py:77

// (struct MAttn: attention_args.h)

// (batch, head) of workgroup blockIdx.x.  Workgroups go to the 8 XCDs round-robin; when B % 8 == 0 the ids are re-read so
// that the H heads of a sample are consecutive workgroups of ONE XCD and share the 128-byte lines of its q|k|v rows in
// that XCD's L2 (heads narrower than 32 floats use only part of a line: see attention.hip locate_head).
__device__ __forceinline__ void locate_head(const MAttn& p, int& b, int& hh) {
    b = blockIdx.x / p.H, hh = blockIdx.x % p.H;
    if ((p.B & 7) == 0) {
        const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        b = (int)((j / p.H) * 8 + xcd);
        hh = (int)(j % p.H);
    }
}

// (long kernels: workgroup id -> (sample, head, row block) by locate_block, attention_args.h)

// ---- prologue reads in two halves: REQUEST everything, then COMMIT ----------------------------------------------------
// Written the obvious way (image 1: load, store to LDS; image 2: load, store; mask; fragments, each scaled on arrival) hipcc
// keeps the program order and a workgroup makes 5-7 DEPENDENT memory round trips before its barrier.  So every read of the
// prologue is first REQUESTED into registers -- branch-free, an out-of-range element re-reads element (0, 0) of its matrix --
// and only then COMMITTED (zeroed, scaled, written to LDS): one round trip (two with SPLIT).
// Two images of rows [0, T) (row stride ld, columns col0..col0+hd-1) -> LDS [TP][HD + 4], zeros beyond T / hd.  Thread t owns
// the 16-byte pieces t + u * blockDim.x, u < U, of both images (U = pieces per thread of the kernel's own shapes; what lies
// beyond goes through rest()).  SPLIT (wide heads: 2 U float4 in flight beside the fragments cost a wave of occupancy): the second
// image is requested when the first is committed -- two round trips instead of one.
template <int HD, int U, bool SPLIT>
struct Stager {
    static constexpr int LS = HD + 4, Q4 = HD / 4;
    float4 a[U], b[U];
    const float* s0; const float* s1;
    int64_t ld0, ld1;
    int col0, T, TP, hd;

    __device__ __forceinline__ void load(float4 (&v)[U], const float* __restrict__ src, int64_t ld, int base) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * (int)blockDim.x;
            const int r = idx / Q4, c = 4 * (idx % Q4);
            const bool ok = idx < TP * Q4 && r < T && c < hd;
            v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(ok ? r : 0) * ld + col0 + (ok ? c : 0));
        }
    }
    __device__ __forceinline__ void store(float* dst, const float4 (&v)[U], int base) const {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * (int)blockDim.x;
            const int r = idx / Q4, c = 4 * (idx % Q4);
            if (idx < TP * Q4) {
                float4 w = v[u];                          // (a value, not `ok ? v[u] : zero`: that selects between ADDRESSES
                if (r >= T || c >= hd) w = make_float4(0.f, 0.f, 0.f, 0.f);   //  and pins the pieces in scratch memory)
                *reinterpret_cast<float4*>(dst + r * LS + c) = w;
            }
        }
    }
    __device__ __forceinline__ void request(const float* src0, const float* src1, int64_t l0, int64_t l1, int col, int rows,
                                            int padded, int width) {
        s0 = src0, s1 = src1, ld0 = l0, ld1 = l1, col0 = col, T = rows, TP = padded, hd = width;
        load(a, s0, ld0, threadIdx.x);
        if (!SPLIT) load(b, s1, ld1, threadIdx.x);
    }
    __device__ __forceinline__ void commit_first(float* d0) {
        store(d0, a, threadIdx.x);
        if (SPLIT) load(b, s1, ld1, threadIdx.x);
    }
    __device__ __forceinline__ void commit_second(float* d0, float* d1) {
        store(d1, b, threadIdx.x);
        for (int base = threadIdx.x + U * blockDim.x; base < TP * Q4; base += U * blockDim.x) {   // not the towers' shapes
            load(a, s0, ld0, base);
            load(b, s1, ld1, base);
            store(d0, a, base);
            store(d1, b, base);
        }
    }
};
// fragments of one 16-row tile held by this wave as the B operand: row = lane & 15, d = 16x + 4g + j
template <int HD>
__device__ __forceinline__ void frags_request(float4 (&f)[HD / 16], const float* __restrict__ src, int64_t ld, int col0,
                                              int row, int T, int g, int hd) {
    const float* p = src + (int64_t)(row < T ? row : T - 1) * ld + col0;
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const int d0 = 16 * x + 4 * g;
        f[x] = *reinterpret_cast<const float4*>(p + (d0 < hd ? d0 : 0));
    }
}
template <int HD>
__device__ __forceinline__ void frags_commit(float4 (&f)[HD / 16], int row, int T, int g, float mul, int hd) {
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const int d0 = 16 * x + 4 * g;
        float4 v = f[x];
        if (row >= T || d0 >= hd) v = make_float4(0.f, 0.f, 0.f, 0.f);
        f[x] = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
    }
}
// acc[r] = T[tile row 4g + r][fixed row c] = sum_d tile[4g + r][d] * fixed[c][d]   (A = LDS tile rows, B = fragments)
template <int HD>
__device__ __forceinline__ f32x4 score16(const float* tile, const float4 (&bf)[HD / 16], int c, int g) {
    constexpr int LS = HD + 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const float4 a = *reinterpret_cast<const float4*>(tile + c * LS + 16 * x + 4 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bf[x].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bf[x].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bf[x].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bf[x].w, acc, 0, 0, 0);
    }
    return acc;
}
// out[t][r'] (row 4g + r' of the fixed tile, column 16t + c) += sum over the tile's 16 rows of
// a[row][fixed col] * tile[row][16t + c];  a[r] belongs to tile row 4g + r (the MFMA k index is g)
template <int HD>
__device__ __forceinline__ void accum16(const f32x4& a, const float* tile, f32x4 (&out)[HD / 16], int c, int g) {
    constexpr int LS = HD + 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float* row = tile + (4 * g + r) * LS + c;
#pragma unroll
        for (int t = 0; t < HD / 16; ++t) out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], row[16 * t], out[t], 0, 0, 0);
    }
}
__device__ __forceinline__ float group_max4(float v) {  // over the 4 lane groups (same lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ---- ragged last token (self-attention over T = 16n + 1 tokens: a ViT's class token + a 2^k patch grid) -------------
// Padded to n + 1 tiles, the one extra token costs 2n + 1 of the (n + 1)^2 tile products (9 of 25 at T = 65: the kernels
// took 1.5x the time of T = 64 on the same bytes).  With `tail` set the matrix cores only see the n x n full tiles;
// token z = 16n is done on the vector ALU:
//   * as a streamed row (key z in forward / dQ, query z in dK,dV): every wave adds its rank-1 contribution -- two
//     16-term inner products per lane, shared over the 4 lane groups, and 16 multiply-adds per output product;
//   * as a fixed row (query z / key z): the extra wave n runs lane = streamed row (inner products against the LDS
//     images), then lane = output column (sums over the streamed rows); a few hundred instructions, beside the MFMAs.
// dot of this lane's fragment slice (d = 16x + 4g .. +3) with row `row` of an LDS image, summed over the 4 lane groups
template <int HD>
__device__ __forceinline__ float frag_dot_row(const float4 (&f)[HD / 16], const float* row, int g) {
    float acc = 0.f;
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const float4 r4 = *reinterpret_cast<const float4*>(row + 16 * x + 4 * g);
        acc += f[x].x * r4.x + f[x].y * r4.y + f[x].z * r4.z + f[x].w * r4.w;
    }
    return group_sum4(acc);
}
// out[t][r] += a[row 4g + r of this wave's tile] * row[16t + c]   (a lives in the lane whose column is that row)
template <int HD>
__device__ __forceinline__ void rank1_update(float a, const float* row, f32x4 (&out)[HD / 16], int c, int g) {
    float ar[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ar[r] = __shfl(a, 4 * g + r, 64);
#pragma unroll
    for (int t = 0; t < HD / 16; ++t) {
        const float v = row[16 * t + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) out[t][r] += ar[r] * v;
    }
}
// lane = streamed row j (two passes: j = lane, 64 + lane): inner product of LDS row j of `img` with the broadcast row z
template <int HD>
__device__ __forceinline__ float row_dot(const float* img, int j, const float* z) {
    constexpr int LS = HD + 4;
    float acc = 0.f;
#pragma unroll
    for (int x = 0; x < HD / 4; ++x) {
        const float4 a = *reinterpret_cast<const float4*>(img + j * LS + 4 * x);
        const float4 b = *reinterpret_cast<const float4*>(z + 4 * x);
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    return acc;
}
// lane = output column d: sum_j w[j] * img[j][d] over the first `rows` (a multiple of 4, zero-weighted beyond T) rows
template <int HD>
__device__ __forceinline__ float col_sum(const float* img, const float* w, int rows, int d) {
    constexpr int LS = HD + 4;
    float a0 = 0.f, a1 = 0.f;
    for (int j = 0; j < rows; j += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(w + j);
        a0 += w4.x * img[j * LS + d] + w4.z * img[(j + 2) * LS + d];
        a1 += w4.y * img[(j + 1) * LS + d] + w4.w * img[(j + 3) * LS + d];
    }
    return a0 + a1;
}
// lane = column d: inner product of two LDS rows over the wave (the z-th streamed row against the broadcast row: one
// element, not worth a lane = row pass)
template <int HD>
__device__ __forceinline__ float wave_dot(const float* a, const float* b, int lane) {
    return wave_sum(lane < HD ? a[lane] * b[lane] : 0.f);
}
constexpr int kTailScratch = 384;   // floats of LDS behind the images: 2 broadcast rows (64 each) + 2 weight vectors (128)

__device__ __forceinline__ unsigned short bf16_rne_bits(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const unsigned short*>(&b);
}

// The output tile of one (sample, head) -- Tq rows x hd columns in LDS, row stride HD + 4 -- as planes in the blocked layout of
// msn_plane_split: a thread takes 8 columns of one row (16 bytes of one plane image), consecutive threads consecutive 16-byte
// chunks of ONE image, so a wave's store is a contiguous run of it (the form of the one-pass backward's dqkv output).  Rows of a
// sample start anywhere in a 32-row block: global row = b Tq + row.  Round to nearest even, residuals exact: the bytes
// msn_plane_split writes for the same fp32 values.
template <int HD>
__device__ __forceinline__ void planes_from_tile(const MAttn& p, const float* tile, int b, int col0) {
    constexpr int LS = HD + 4;
    const int NB = p.hd / 16, T = p.Tq, NP = p.o_np;
    unsigned char* cbase = p.oplanes + (int64_t)(col0 / 16) * (NP * 1024);
    const int64_t r0 = (int64_t)b * T;
    for (int idx = threadIdx.x; idx < 2 * T * NB; idx += blockDim.x) {
        const int j = idx / (2 * T), rh = idx - 2 * T * j, row = rh >> 1, half = rh & 1;
        const float* src = tile + row * LS + 16 * j + 8 * half;
        const float4 w0 = *reinterpret_cast<const float4*>(src), w1 = *reinterpret_cast<const float4*>(src + 4);
        float v[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        const int64_t gr = r0 + row;
        unsigned char* dst = cbase + ((gr >> 5) * p.o_cb + j) * (int64_t)(NP * 1024) + (int)(gr & 31) * 32 + 16 * half;
        for (int k = 0; k < NP; ++k) {
            unsigned w[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const unsigned short lo = bf16_rne_bits(v[2 * x]), hi = bf16_rne_bits(v[2 * x + 1]);
                v[2 * x] -= __uint_as_float((unsigned)lo << 16);          // exact
                v[2 * x + 1] -= __uint_as_float((unsigned)hi << 16);
                w[x] = lo | ((unsigned)hi << 16);
            }
            *reinterpret_cast<uint4*>(dst + k * 1024) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// ------------------------------------------------------------------------------------------ forward
template <int HD>
__global__ __launch_bounds__(HD > 64 ? 512 : 1024) void mattn_fwd_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TPk = (p.Tk + 15) / 16 * 16;
    float* Ks = smem;
    // LDS image rows: whole tiles, or -- with the ragged token on the vector ALU -- only the rows that path reads (68 of
    // 80 at T = 65: a fourth workgroup fits on the CU)
    const int rows = p.tail ? (p.Tk + 3) / 4 * 4 : TPk;
    float* Vs = smem + (size_t)rows * LS;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + (size_t)rows * LS);
    float* Zs = reinterpret_cast<float*>(Ms + TPk);      // tail scratch (TPk % 16 == 0: 16-byte aligned)
    int b, hh;
    locate_head(p, b, hh);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int z = p.Tk - 1;                               // the ragged token (tail only)
    const bool tail_wave = p.tail && wave == TPk / 16 - 1;
    // ---- request: K / V rows, key mask, this wave's query fragments (the tail wave: query z), all in flight together
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    Stager<HD, DT, (HD > 32)> sv;
    sv.request(ksrc, vsrc, p.ldk, p.ldv, col0, p.Tk, rows, p.hd);
    const int mj = threadIdx.x;
    uint8_t mk = 1;
    if (p.mask) mk = p.mask[(int64_t)b * p.Tk + (mj < p.Tk ? mj : 0)];
    const int q0 = wave * 16, qrow = q0 + c;
    float4 qf[DT];
    float4 zq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tail_wave) {
        if (lane < HD / 4 && 4 * lane < p.hd)
            zq = *reinterpret_cast<const float4*>(p.q + (int64_t)b * p.q_bs + (int64_t)z * p.ldq + col0 + 4 * lane);
    } else {
        frags_request<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.hd);
    }
    // ---- commit
    sv.commit_first(Ks);
    if (mj < TPk) Ms[mj] = mj < p.Tk ? mk : 1;
    for (int j = mj + blockDim.x; j < TPk; j += blockDim.x) Ms[j] = (j < p.Tk && p.mask) ? p.mask[(int64_t)b * p.Tk + j] : 1;
    if (tail_wave) {
        if (lane < HD / 4)                                // query z, scaled, as a broadcast row
            *reinterpret_cast<float4*>(Zs + 4 * lane) = make_float4(zq.x * p.scale, zq.y * p.scale, zq.z * p.scale, zq.w * p.scale);
    } else {
        frags_commit<HD>(qf, qrow, p.Tq, g, p.scale, p.hd);
    }
    sv.commit_second(Ks, Vs);
    __syncthreads();

    if (tail_wave) {
        float* Ps = Zs + 128;
        float sj[2], m = -INFINITY;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {                  // keys before z, lane = key (a pass without keys is skipped)
            const int j = 64 * h2 + lane;
            sj[h2] = -INFINITY;
            if (j < z) sj[h2] = Ms[j] ? row_dot<HD>(Ks, j, Zs) : kFill;
            m = fmaxf(m, sj[h2]);
        }
        float sz = wave_dot<HD>(Ks + z * LS, Zs, lane);   // key z itself
        sz = Ms[z] ? sz : kFill;
        m = fmaxf(wave_max(m), sz);
        float l = 0.f;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const float e = __expf(sj[h2] - m);           // 0 from z on
            Ps[64 * h2 + lane] = e;
            l += e;
        }
        const float ez = __expf(sz - m);
        if (lane == 0) Ps[z] = ez;
        l = wave_sum(l) + ez;
        float oz = 0.f;
        if (lane < p.hd) {
            oz = col_sum<HD>(Vs, Ps, (p.Tk + 3) / 4 * 4, lane) / l;
            p.out[(int64_t)b * p.o_bs + (int64_t)z * p.ldo + col0 + lane] = oz;
        }
        if (lane == 0) {
            float* st = p.lse + 2 * (((int64_t)b * p.H + hh) * p.Tq + z);
            st[0] = m;
            st[1] = __logf(l);
        }
        if (!p.oplanes) return;
        __syncthreads();                                  // every wave is done with the K / V rows: they become the output tile
        if (lane < HD) Ks[z * LS + lane] = lane < p.hd ? oz : 0.f;
        __syncthreads();
        planes_from_tile<HD>(p, Ks, b, col0);
        return;
    }

    const int nkt = TPk / 16 - (p.tail ? 1 : 0);          // full key tiles on the matrix cores

    float m = -INFINITY;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    // key tiles whose scores fit in registers (4 per tile) next to the output accumulators: sequences up to 256
    // tokens for 16-wide heads (the reference's spectrum tower: 220 tokens), 128 tokens for wider heads
    constexpr int KEEP = HD <= 16 ? 16 : 8;
    if (nkt <= KEEP) {
        // one pass: every score tile is computed once and kept (ViT-S/8 at 64x64: 65 tokens -> 5 tiles)
        f32x4 sc[KEEP];
#pragma unroll
        for (int kt = 0; kt < KEEP; ++kt) {
            if (kt < nkt) {
                sc[kt] = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * g + r;
                    sc[kt][r] = key < p.Tk ? (Ms[key] ? sc[kt][r] : kFill) : -INFINITY;
                    m = fmaxf(m, sc[kt][r]);
                }
            }
        }
        m = group_max4(m);
        float sz = -INFINITY;
        if (p.tail) {                                     // key z: this query column's score, the same in all 4 groups
            sz = frag_dot_row<HD>(qf, Ks + z * LS, g);
            sz = Ms[z] ? sz : kFill;
            m = fmaxf(m, sz);
        }
#pragma unroll
        for (int kt = 0; kt < KEEP; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sc[kt][r] - m);   // exp(-inf) = 0 for the padded keys
                    sc[kt][r] = e;
                    l += e;
                }
                accum16<HD>(sc[kt], Vs + kt * 16 * LS, o, c, g);
            }
        }
        if (p.tail) {
            const float ez = __expf(sz - m);
            if (g == 0) l += ez;                          // once per query (l is summed over the groups below)
            rank1_update<HD>(ez, Vs + z * LS, o, c, g);
        }
    } else {
        // two passes over the key tiles (row maximum, then exponentials + P.V), recomputing the 16x16 score tiles
        for (int kt = 0; kt < nkt; ++kt) {
            const f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                if (key < p.Tk) m = fmaxf(m, Ms[key] ? s[r] : kFill);
            }
        }
        m = group_max4(m);
        for (int kt = 0; kt < nkt; ++kt) {
            f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                const float e = key < p.Tk ? __expf((Ms[key] ? s[r] : kFill) - m) : 0.f;
                s[r] = e;
                l += e;
            }
            accum16<HD>(s, Vs + kt * 16 * LS, o, c, g);
        }
    }
    l = group_sum4(l);
    // o[t][r] belongs to query q0 + 4g + r: fetch that query's normaliser from its column-owner lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lq = __shfl(l, 4 * g + r, 64);
        const int q = q0 + 4 * g + r;
        const float inv = 1.f / lq;
#pragma unroll
        for (int t = 0; t < DT; ++t) o[t][r] *= inv;
        if (q < p.Tq) {
            float* op = p.out + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) op[16 * t] = o[t][r];
        }
    }
    if (g == 0 && qrow < p.Tq) {
        float* st = p.lse + 2 * (((int64_t)b * p.H + hh) * p.Tq + qrow);
        st[0] = m;
        st[1] = __logf(l);
    }
    if (!p.oplanes) return;
    // ---- the same values as a plane matrix: the tile goes through LDS (the K rows are no longer read) and leaves as
    // contiguous runs of the block images (planes_from_tile)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        if (q < p.Tq) {
#pragma unroll
            for (int t = 0; t < DT; ++t) Ks[q * LS + 16 * t + c] = (16 * t + c < p.hd) ? o[t][r] : 0.f;
        }
    }
    __syncthreads();
    planes_from_tile<HD>(p, Ks, b, col0);
}

// ------------------------------------------------------------------------------- backward: dQ, delta
template <int HD>
__global__ __launch_bounds__(HD > 64 ? 512 : 1024) void mattn_bwd_dq_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16;
    constexpr bool LATE = HD > 32;   // fragments requested behind the staged rows (as many registers as before), not beside them
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TPk = (p.Tk + 15) / 16 * 16;
    float* Ks = smem;
    // LDS image rows: whole tiles, or -- with the ragged token on the vector ALU -- only the rows that path reads (68 of
    // 80 at T = 65: a fourth workgroup fits on the CU)
    const int rows = p.tail ? (p.Tk + 3) / 4 * 4 : TPk;
    float* Vs = smem + (size_t)rows * LS;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + (size_t)rows * LS);
    float* Zs = reinterpret_cast<float*>(Ms + TPk);      // tail scratch
    int b, hh;
    locate_head(p, b, hh);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int z = p.Tk - 1;
    const bool tail_wave = p.tail && wave == TPk / 16 - 1;
    // ---- request (see the forward kernel): K / V rows, key mask, this wave's q / dO / O fragments and row statistics
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    Stager<HD, DT, (HD > 32)> sv;
    sv.request(ksrc, vsrc, p.ldk, p.ldv, col0, p.Tk, rows, p.hd);
    const int mj = threadIdx.x;
    uint8_t mk = 1;
    if (p.mask) mk = p.mask[(int64_t)b * p.Tk + (mj < p.Tk ? mj : 0)];
    const int q0 = wave * 16, qrow = q0 + c;
    const bool q_ok = qrow < p.Tq;
    float4 qf[DT], df[DT], of[DT];
    float lm = 0.f, ll = 0.f;
    const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + (q_ok ? qrow : 0);
    float4 zq = make_float4(0.f, 0.f, 0.f, 0.f), zd = zq;
    float dz = 0.f;                                       // tail wave: lane d's term of delta_z = dO_z . O_z
    if (tail_wave) {
        if (lane < HD / 4 && 4 * lane < p.hd) {           // query z and its dO row
            zq = *reinterpret_cast<const float4*>(p.q + (int64_t)b * p.q_bs + (int64_t)z * p.ldq + col0 + 4 * lane);
            zd = *reinterpret_cast<const float4*>(p.dout + (int64_t)b * p.d_bs + (int64_t)z * p.ldd + col0 + 4 * lane);
        }
        if (lane < p.hd)
            dz = p.dout[(int64_t)b * p.d_bs + (int64_t)z * p.ldd + col0 + lane] *
                 p.o[(int64_t)b * p.o_bs + (int64_t)z * p.ldo + col0 + lane];
    } else {
        if (!LATE) {
            frags_request<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.hd);
            frags_request<HD>(df, p.dout + (int64_t)b * p.d_bs, p.ldd, col0, qrow, p.Tq, g, p.hd);
            frags_request<HD>(of, p.o + (int64_t)b * p.o_bs, p.ldo, col0, qrow, p.Tq, g, p.hd);
        }
        lm = p.lse[2 * stat], ll = p.lse[2 * stat + 1];   // row 0's for a padded query: never used (q_ok)
    }
    // ---- commit
    sv.commit_first(Ks);
    if (mj < TPk) Ms[mj] = mj < p.Tk ? mk : 1;
    for (int j = mj + blockDim.x; j < TPk; j += blockDim.x) Ms[j] = (j < p.Tk && p.mask) ? p.mask[(int64_t)b * p.Tk + j] : 1;
    if (tail_wave) {
        if (lane < HD / 4) {                              // query z (scaled) and its dO row as broadcast rows
            *reinterpret_cast<float4*>(Zs + 4 * lane) = make_float4(zq.x * p.scale, zq.y * p.scale, zq.z * p.scale, zq.w * p.scale);
            *reinterpret_cast<float4*>(Zs + 64 + 4 * lane) = zd;
        }
    }
    if (LATE) sv.commit_second(Ks, Vs);
    if (!tail_wave) {
        if (LATE) {                                       // wide heads: the fragments take their own round trip, behind K and V
            frags_request<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.hd);
            frags_request<HD>(df, p.dout + (int64_t)b * p.d_bs, p.ldd, col0, qrow, p.Tq, g, p.hd);
            frags_request<HD>(of, p.o + (int64_t)b * p.o_bs, p.ldo, col0, qrow, p.Tq, g, p.hd);
        }
        frags_commit<HD>(qf, qrow, p.Tq, g, p.scale, p.hd);
        frags_commit<HD>(df, qrow, p.Tq, g, 1.f, p.hd);
        frags_commit<HD>(of, qrow, p.Tq, g, 1.f, p.hd);
    }
    if (!LATE) sv.commit_second(Ks, Vs);
    __syncthreads();

    if (tail_wave) {
        float* Ps = Zs + 128;
        const float delta = wave_sum(dz);
        const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + z;
        const float lm = p.lse[2 * stat], ll = p.lse[2 * stat + 1];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int j = 64 * h2 + lane;
            float ds = 0.f;
            if (j < z && Ms[j]) {
                const float sj = row_dot<HD>(Ks, j, Zs), dp = row_dot<HD>(Vs, j, Zs + 64);
                ds = __expf((sj - lm) - ll) * (dp - delta);
            }
            Ps[j] = ds;
        }
        {
            const float sz = wave_dot<HD>(Ks + z * LS, Zs, lane), dpz = wave_dot<HD>(Vs + z * LS, Zs + 64, lane);
            if (lane == 0) Ps[z] = Ms[z] ? __expf((sz - lm) - ll) * (dpz - delta) : 0.f;
        }
        if (lane < p.hd)
            p.dq[(int64_t)b * p.dq_bs + (int64_t)z * p.lddq + col0 + lane] =
                col_sum<HD>(Ks, Ps, (p.Tk + 3) / 4 * 4, lane) * p.scale;
        if (lane == 0) p.delta[stat] = delta;
        return;
    }

    float delta = 0.f;
#pragma unroll
    for (int x = 0; x < DT; ++x)
        delta += df[x].x * of[x].x + df[x].y * of[x].y + df[x].z * of[x].z + df[x].w * of[x].w;
    delta = group_sum4(delta);
    if (g == 0 && q_ok) p.delta[stat] = delta;

    f32x4 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nkt = TPk / 16 - (p.tail ? 1 : 0);
    for (int kt = 0; kt < nkt; ++kt) {
        const f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
        const f32x4 dp = score16<HD>(Vs + kt * 16 * LS, df, c, g);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            // a padded key contributes nothing; a masked key has a constant score: no gradient to q / k
            const bool live = q_ok && key < p.Tk && Ms[key];
            ds[r] = live ? __expf((s[r] - lm) - ll) * (dp[r] - delta) : 0.f;
        }
        accum16<HD>(ds, Ks + kt * 16 * LS, dq, c, g);
    }
    if (p.tail) {                                         // key z
        const float sz = frag_dot_row<HD>(qf, Ks + z * LS, g), dpz = frag_dot_row<HD>(df, Vs + z * LS, g);
        const float dsz = (q_ok && Ms[z]) ? __expf((sz - lm) - ll) * (dpz - delta) : 0.f;
        rank1_update<HD>(dsz, Ks + z * LS, dq, c, g);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        if (q < p.Tq) {
            float* op = p.dq + (int64_t)b * p.dq_bs + (int64_t)q * p.lddq + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) op[16 * t] = dq[t][r] * p.scale;
        }
    }
}

// ------------------------------------------------------------------------------- backward: dK, dV
template <int HD>
__global__ __launch_bounds__(HD > 64 ? 512 : 1024) void mattn_bwd_dkv_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16;
    constexpr bool LATE = HD > 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TPq = (p.Tq + 15) / 16 * 16;
    float* Qs = smem;
    const int rows = p.tail ? (p.Tq + 3) / 4 * 4 : TPq;   // LDS image rows (see the forward kernel)
    float* Ds = smem + (size_t)rows * LS;
    float* Lm = Ds + (size_t)rows * LS;
    float* Ll = Lm + TPq;
    float* Dl = Ll + TPq;
    float* Zs = Dl + TPq;                                 // tail scratch
    int b, hh;
    locate_head(p, b, hh);
    const int col0 = hh * p.hd;
    const int z = p.Tq - 1;
    const bool tail_wave = p.tail && (int)(threadIdx.x >> 6) == TPq / 16 - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    // ---- request (see the forward kernel): Q / dO rows, the queries' statistics, this wave's key / value fragments + mask
    const float* qsrc = p.q + (int64_t)b * p.q_bs;
    const float* dsrc = p.dout + (int64_t)b * p.d_bs;
    Stager<HD, DT, (HD > 32)> sv;
    sv.request(qsrc, dsrc, p.ldq, p.ldd, col0, p.Tq, rows, p.hd);
    const int tj = threadIdx.x;
    const int64_t sj = ((int64_t)b * p.H + hh) * p.Tq + (tj < p.Tq ? tj : 0);
    const float s_m = p.lse[2 * sj], s_l = p.lse[2 * sj + 1], s_d = p.delta[sj];
    const int k0 = wave * 16, krow = k0 + c;
    float4 kf[DT], vf[DT];
    float4 zk = make_float4(0.f, 0.f, 0.f, 0.f), zv = zk;
    uint8_t mk = 1;
    if (tail_wave) {
        if (lane < HD / 4 && 4 * lane < p.hd) {           // key z and value z
            zk = *reinterpret_cast<const float4*>(p.k + (int64_t)b * p.k_bs + (int64_t)z * p.ldk + col0 + 4 * lane);
            zv = *reinterpret_cast<const float4*>(p.v + (int64_t)b * p.v_bs + (int64_t)z * p.ldv + col0 + 4 * lane);
        }
    } else {
        if (!LATE) {
            frags_request<HD>(kf, p.k + (int64_t)b * p.k_bs, p.ldk, col0, krow, p.Tk, g, p.hd);
            frags_request<HD>(vf, p.v + (int64_t)b * p.v_bs, p.ldv, col0, krow, p.Tk, g, p.hd);
        }
        if (p.mask) mk = p.mask[(int64_t)b * p.Tk + (krow < p.Tk ? krow : 0)];
    }
    // ---- commit
    sv.commit_first(Qs);
    if (tj < TPq) {
        Lm[tj] = tj < p.Tq ? s_m : INFINITY;              // +inf: a padded query row gets p = exp(-inf) = 0
        Ll[tj] = tj < p.Tq ? s_l : 0.f;
        Dl[tj] = tj < p.Tq ? s_d : 0.f;
    }
    for (int t = tj + blockDim.x; t < TPq; t += blockDim.x) {
        const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + (t < p.Tq ? t : 0);
        Lm[t] = t < p.Tq ? p.lse[2 * stat] : INFINITY;
        Ll[t] = t < p.Tq ? p.lse[2 * stat + 1] : 0.f;
        Dl[t] = t < p.Tq ? p.delta[stat] : 0.f;
    }
    bool keep = false;
    if (tail_wave) {
        if (lane < HD / 4) {                              // key z (scaled) and value z as broadcast rows
            *reinterpret_cast<float4*>(Zs + 4 * lane) = make_float4(zk.x * p.scale, zk.y * p.scale, zk.z * p.scale, zk.w * p.scale);
            *reinterpret_cast<float4*>(Zs + 64 + 4 * lane) = zv;
        }
    } else {
        if (LATE) {
            frags_request<HD>(kf, p.k + (int64_t)b * p.k_bs, p.ldk, col0, krow, p.Tk, g, p.hd);
            frags_request<HD>(vf, p.v + (int64_t)b * p.v_bs, p.ldv, col0, krow, p.Tk, g, p.hd);
        }
        frags_commit<HD>(kf, krow, p.Tk, g, p.scale, p.hd);
        frags_commit<HD>(vf, krow, p.Tk, g, 1.f, p.hd);
        keep = krow < p.Tk && mk != 0;
    }
    sv.commit_second(Qs, Ds);
    __syncthreads();

    if (tail_wave) {
        float* Pp = Zs + 128;
        float* Pd = Zs + 256;
        const bool keep_z = p.mask ? p.mask[(int64_t)b * p.Tk + z] != 0 : true;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int i = 64 * h2 + lane;
            float pr = 0.f, ds = 0.f;
            if (i < z) {
                const float si = row_dot<HD>(Qs, i, Zs), dp = row_dot<HD>(Ds, i, Zs + 64);
                pr = __expf(((keep_z ? si : kFill) - Lm[i]) - Ll[i]);
                ds = keep_z ? pr * (dp - Dl[i]) : 0.f;
            }
            Pp[i] = pr;
            Pd[i] = ds;
        }
        {
            const float sz = wave_dot<HD>(Qs + z * LS, Zs, lane), dpz = wave_dot<HD>(Ds + z * LS, Zs + 64, lane);
            const float pr = __expf(((keep_z ? sz : kFill) - Lm[z]) - Ll[z]);
            if (lane == 0) {
                Pp[z] = pr;
                Pd[z] = keep_z ? pr * (dpz - Dl[z]) : 0.f;
            }
        }
        if (lane < p.hd) {
            p.dk[(int64_t)b * p.dk_bs + (int64_t)z * p.lddk + col0 + lane] = col_sum<HD>(Qs, Pd, (p.Tq + 3) / 4 * 4, lane) * p.scale;
            p.dv[(int64_t)b * p.dv_bs + (int64_t)z * p.lddv + col0 + lane] = col_sum<HD>(Ds, Pp, (p.Tq + 3) / 4 * 4, lane);
        }
        return;
    }
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nqt = TPq / 16 - (p.tail ? 1 : 0);
    for (int qt = 0; qt < nqt; ++qt) {
        const f32x4 s = score16<HD>(Qs + qt * 16 * LS, kf, c, g);     // rows = queries, col = this lane's key
        const f32x4 dp = score16<HD>(Ds + qt * 16 * LS, vf, c, g);
        f32x4 pr, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = 16 * qt + 4 * g + r;
            const float e = __expf(((keep ? s[r] : kFill) - Lm[q]) - Ll[q]);
            pr[r] = krow < p.Tk ? e : 0.f;
            ds[r] = keep ? e * (dp[r] - Dl[q]) : 0.f;
        }
        accum16<HD>(pr, Ds + qt * 16 * LS, dv, c, g);
        accum16<HD>(ds, Qs + qt * 16 * LS, dk, c, g);
    }
    if (p.tail) {                                         // query z against this lane's key
        const float sz = frag_dot_row<HD>(kf, Qs + z * LS, g), dpz = frag_dot_row<HD>(vf, Ds + z * LS, g);
        const float e = __expf(((keep ? sz : kFill) - Lm[z]) - Ll[z]);
        rank1_update<HD>(krow < p.Tk ? e : 0.f, Ds + z * LS, dv, c, g);
        rank1_update<HD>(keep ? e * (dpz - Dl[z]) : 0.f, Qs + z * LS, dk, c, g);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + 4 * g + r;
        if (k < p.Tk) {
            float* ok = p.dk + (int64_t)b * p.dk_bs + (int64_t)k * p.lddk + col0 + c;
            float* ov = p.dv + (int64_t)b * p.dv_bs + (int64_t)k * p.lddv + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) {
                    ok[16 * t] = dk[t][r] * p.scale;
                    ov[16 * t] = dv[t][r];
                }
        }
    }
}


#ifdef MSN_ATTN_TIMELINE   // diagnostic build (tools/microbench/attn_timeline.py): shader-clock stamps of wave 0 of every workgroup
constexpr int kDbgRows = 32768;
__device__ unsigned long long g_mattn_dbg[kDbgRows * 8];
extern "C" int msn_mattn_debug_read(unsigned long long* out, int rows) {   // out[rows][8]; clears the buffer
    if (rows > kDbgRows) return 1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mattn_dbg), sizeof(unsigned long long) * 8 * rows) != hipSuccess) return 1;
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_mattn_dbg)) != hipSuccess) return 1;
    return hipMemset(sym, 0, sizeof(unsigned long long) * 8 * kDbgRows) == hipSuccess ? 0 : 1;
}
#define MSN_TL(var) const unsigned long long var = __builtin_readcyclecounter();
#else
#define MSN_TL(var)
#endif
// ---------------------------------------------------------------------------------- backward in ONE pass: dQ, dK, dV
// Self-attention over up to 128 tokens (the ViT towers: 65): the four images a (sample, head) needs -- Q, K, V, dO -- sit in
// LDS TOGETHER (74 KB at T = 65, 64-wide heads: two workgroups per CU), so the backward reads them from HBM once instead of
// twice (the dQ kernel's K | V + fragments, then the dK,dV kernel's Q | dO + fragments + the statistics), delta = dO . O never
// goes through memory, and one launch replaces two.  Same products, same order of accumulation, same ragged-token path as
// mattn_bwd_dq_kernel / mattn_bwd_dkv_kernel above (results equal theirs to the contraction of multiply-adds the compiler
// picks per kernel); the fixed-tile fragments come from the LDS images instead of HBM.  Wave w owns query tile w (dQ phase), then key tile w (dK,dV phase).
// The results leave through LDS (the images' space, once every wave is done with them) in memory order:
//   NP = 0: fp32 rows of dq / dk / dv, 16 bytes per lane;
//   NP = 2 | 3: bf16 PLANES of the packed gradient matrix dqkv (B T x 3 H hd; csrc/pgemm.hip's blocked layout), which is what
//   the two products that consume it read -- the fp32 matrix and the split pass over it (144 us at the headline shape) are
//   gone -- plus the column sums of this sample's rows (the bias gradient's partial sums; finished by colsum_finish).
struct FusedOut {
    unsigned char* planes;    // NP > 0: plane matrix of dqkv
    float* colpart;           // NP > 0, nullable: [B][3 H hd] column sums per sample
    int cb;                   // column blocks of the plane matrix = 2 ceil(3 H hd / 32)
};
template <int HD>
__device__ __forceinline__ void lds_frags(float4 (&f)[HD / 16], const float* img, int row, int g, float mul) {
    constexpr int LS = HD + 4;
#pragma unroll
    for (int x = 0; x < HD / 16; ++x) {
        const float4 v = *reinterpret_cast<const float4*>(img + row * LS + 16 * x + 4 * g);
        f[x] = make_float4(v.x * mul, v.y * mul, v.z * mul, v.w * mul);
    }
}
template <int HD, int NP, bool SHARE>
__global__ __launch_bounds__(512) void mattn_bwd_fused_kernel(const MAttn p, const FusedOut fo) {
    constexpr int LS = HD + 4, DT = HD / 16;
    MSN_TL(tl0)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.Tk;                                   // == p.Tq
    const int TP = (T + 15) / 16 * 16;
    const bool tail = p.tail != 0;
    const int rows = tail ? T : TP;                       // LDS image rows: the full tiles + the ragged token's row
    float* Qs = smem;
    float* Ks = Qs + (size_t)rows * LS;
    float* Vs = Ks + (size_t)rows * LS;
    float* Ds = Vs + (size_t)rows * LS;
    float* Lm = Ds + (size_t)rows * LS;
    float* Ll = Lm + TP;
    float* Dl = Ll + TP;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Dl + TP);
    float* Zp = reinterpret_cast<float*>(Ms + TP);        // ragged token: [3][waves][HD] partial rows of dq_z | dk_z | dv_z
    int b, hh;
    locate_head(p, b, hh);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int nw = blockDim.x >> 6;                       // one wave per FULL 16-row tile
    const int z = T - 1;                                  // the ragged token (tail only)
    const int t0 = wave * 16, trow = t0 + c;              // this wave's tile; this lane's fixed row (query, then key)
    const bool t_ok = trow < T;
    const int dl = lane < HD ? lane : 0;                  // lane = head column in the ragged token's sums
    // ---- request: the four images, the statistics, the key mask, this wave's O fragments (delta), all in flight together
    // (DT + 1 pieces per thread: with the ragged token on the vector ALU the workgroup has one wave per FULL tile only)
    const float* qsrc = p.q + (int64_t)b * p.q_bs;
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    const float* dsrc = p.dout + (int64_t)b * p.d_bs;
    Stager<HD, DT + 1, false> s1, s2;
    s1.request(qsrc, ksrc, p.ldq, p.ldk, col0, T, rows, p.hd);
    s2.request(vsrc, dsrc, p.ldv, p.ldd, col0, T, rows, p.hd);
    const int tj = threadIdx.x;
    const int64_t sj = ((int64_t)b * p.H + hh) * T + (tj < T ? tj : 0);
    const float s_m = p.lse[2 * sj], s_l = p.lse[2 * sj + 1];
    uint8_t mk = 1;
    if (p.mask) mk = p.mask[(int64_t)b * T + (tj < T ? tj : 0)];
    float4 of[DT];
    frags_request<HD>(of, p.o + (int64_t)b * p.o_bs, p.ldo, col0, trow, T, g, p.hd);
    float dz = 0.f;                                       // wave 0: lane d's term of delta_z = dO_z . O_z
    if (tail && wave == 0 && lane < p.hd)
        dz = dsrc[(int64_t)z * p.ldd + col0 + lane] * p.o[(int64_t)b * p.o_bs + (int64_t)z * p.ldo + col0 + lane];
    MSN_TL(tl1)
    // ---- commit
    s1.commit_first(Qs);
    s1.commit_second(Qs, Ks);
    s2.commit_first(Vs);
    s2.commit_second(Vs, Ds);
    for (int t = tj; t < TP; t += blockDim.x) {
        const bool in = t < T;
        float m_ = s_m, l_ = s_l;
        uint8_t k_ = mk;
        if (t != tj) {                                    // (fewer threads than padded rows: not the towers' shapes)
            const int64_t st = ((int64_t)b * p.H + hh) * T + (in ? t : 0);
            m_ = p.lse[2 * st], l_ = p.lse[2 * st + 1];
            k_ = p.mask ? p.mask[(int64_t)b * T + (in ? t : 0)] : 1;
        }
        Lm[t] = in ? m_ : INFINITY;                       // +inf: a padded query row gets p = exp(-inf) = 0
        Ll[t] = in ? l_ : 0.f;
        Ms[t] = in ? k_ : 1;
    }
    frags_commit<HD>(of, trow, T, g, 1.f, p.hd);
    __syncthreads();
    MSN_TL(tl2)

    // ---- delta of this wave's queries (registers for the dQ phase, LDS for every wave's dK,dV phase)
    float4 af[DT], bf[DT];                                // fixed-tile fragments: (q, dO), then (k, v)
    float delta = 0.f;
    lds_frags<HD>(af, Qs, trow, g, p.scale);
    lds_frags<HD>(bf, Ds, trow, g, 1.f);
#pragma unroll
    for (int x = 0; x < DT; ++x)
        delta += bf[x].x * of[x].x + bf[x].y * of[x].y + bf[x].z * of[x].z + bf[x].w * of[x].w;
    delta = group_sum4(delta);
    if (g == 0) Dl[trow] = t_ok ? delta : 0.f;
    if (tail && wave == 0) {                              // rows z .. TP - 1 belong to no wave's tile
        const float dzs = wave_sum(dz);
        if (lane < 16 && z + lane < TP) Dl[z + lane] = lane == 0 ? dzs : 0.f;
    }
    __syncthreads();
    MSN_TL(tl3)
#ifdef MSN_ATTN_TIMELINE
    unsigned long long tl4 = 0;
#endif

    // Ragged token z (T = 16 n + 1): the matrix cores see the n x n full tiles; row z of dq / dk / dv and column z of the
    // scores go over the vector ALU INSIDE the matrix waves.  Every weight that row needs is a by-product of the rank-1 terms
    // the waves form anyway -- ds(i, z), p(i, z) for this wave's queries i in the dQ phase, ds(z, j) for its keys j in the
    // dK,dV phase -- so a wave adds, lane = head column, its 16 rows' share of dk_z, dv_z and dq_z (16 broadcasts + 16
    // multiply-adds each); the shares meet in LDS, wave 0 adds the (z, z) element.  (A fifth wave of vector work per
    // workgroup -- the form of the dQ / dK,dV kernels above -- was the last one to finish: 1 800 cycles of the other four.)
    const int ntile = tail ? TP / 16 - 1 : TP / 16;       // full tiles on the matrix cores
    f32x4 dq[DT], dk[DT], dv[DT];
    f32x4 dqp[SHARE ? 4 : 1][DT];                         // SHARE: this wave's keys' share of dQ, per query tile
    float zdq = 0.f, zdk = 0.f, zdv = 0.f;                // lane d's share of row z
    if constexpr (SHARE) {
        // ---- ONE recomputation of the score / score-gradient tiles serves dQ, dK and dV (up to 4 full tiles: the ViT towers).
        // Wave w owns KEY tile w.  Per query tile: s and dp as in the dK,dV phase below (rows = queries, column = this lane's
        // key), p and ds from them, dv += p^T dO, dk += ds^T Q -- and, new, the tile's ds TRANSPOSED through a 16 x 16-float
        // patch of LDS of this wave (4 stores + one 16-byte read per lane) as the A operand of dq_part[query tile] += ds K(own
        // keys): 80 instead of 112 MFMAs per tile pair.  The four waves' dq parts meet in LDS behind the loop, in wave order.
        float* Tb = Zp + (tail ? 3 * nw * HD : 0) + wave * 256;      // 16 x 16 floats (unpadded: 2-way conflict on 4 stores per tile pair)
        lds_frags<HD>(af, Ks, trow, g, p.scale);          // af = k (scaled), bf = v
        lds_frags<HD>(bf, Vs, trow, g, 1.f);
        const bool keep = t_ok && Ms[trow] != 0;
#pragma unroll
        for (int t = 0; t < DT; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
            for (int t = 0; t < DT; ++t) dqp[qt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qt < ntile) {
                const f32x4 s = score16<HD>(Qs + qt * 16 * LS, af, c, g);
                const f32x4 dp = score16<HD>(Ds + qt * 16 * LS, bf, c, g);
                f32x4 pr, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * g + r;
                    const float e = __expf(((keep ? s[r] : kFill) - Lm[q]) - Ll[q]);
                    pr[r] = t_ok ? e : 0.f;
                    ds[r] = keep ? e * (dp[r] - Dl[q]) : 0.f;
                    Tb[(4 * g + r) * 16 + c] = ds[r];     // Tb[query][key]
                }
                accum16<HD>(pr, Ds + qt * 16 * LS, dv, c, g);
                accum16<HD>(ds, Qs + qt * 16 * LS, dk, c, g);
                const f32x4 dst = *reinterpret_cast<const f32x4*>(Tb + c * 16 + 4 * g);     // ds(query c, keys 4g .. 4g + 3)
                accum16<HD>(dst, Ks + t0 * LS, dqp[qt], c, g);
            }
        }
        if (tail) {
            // query z against this lane's key (as below), then this wave's QUERY tile against key z (the rank-1 term of dq
            // goes into this wave's own part of that tile; its weights are the tile's share of dk_z, dv_z)
            {
                const float sz = frag_dot_row<HD>(af, Qs + z * LS, g), dpz = frag_dot_row<HD>(bf, Ds + z * LS, g);
                const float e = __expf(((keep ? sz : kFill) - Lm[z]) - Ll[z]);
                const float dsj = keep ? e * (dpz - Dl[z]) : 0.f;
                rank1_update<HD>(t_ok ? e : 0.f, Ds + z * LS, dv, c, g);
                rank1_update<HD>(dsj, Qs + z * LS, dk, c, g);
#pragma unroll
                for (int j = 0; j < 16; ++j) zdq += __shfl(dsj, j, 64) * Ks[(t0 + j) * LS + dl];
            }
            lds_frags<HD>(af, Qs, trow, g, p.scale);      // af = q (scaled), bf = dO
            lds_frags<HD>(bf, Ds, trow, g, 1.f);
            const float lm = Lm[trow], ll = Ll[trow];
            const float sz = frag_dot_row<HD>(af, Ks + z * LS, g), dpz = frag_dot_row<HD>(bf, Vs + z * LS, g);
            const bool keep_z = Ms[z] != 0;
            const float prz = t_ok ? __expf(((keep_z ? sz : kFill) - lm) - ll) : 0.f;
            const float dsz = keep_z ? prz * (dpz - delta) : 0.f;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
                if (qt == wave) rank1_update<HD>(dsz, Ks + z * LS, dqp[qt], c, g);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float wds = __shfl(dsz, i, 64), wpr = __shfl(prz, i, 64);
                zdk += wds * Qs[(t0 + i) * LS + dl];
                zdv += wpr * Ds[(t0 + i) * LS + dl];
            }
        }
    } else {
        // ---- dQ of query tile `wave`: af = q (scaled), bf = dO
        const float lm = Lm[trow], ll = Ll[trow];
#pragma unroll
        for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < ntile; ++kt) {
            const f32x4 s = score16<HD>(Ks + kt * 16 * LS, af, c, g);
            const f32x4 dp = score16<HD>(Vs + kt * 16 * LS, bf, c, g);
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                const bool live = t_ok && key < T && Ms[key];
                ds[r] = live ? __expf((s[r] - lm) - ll) * (dp[r] - delta) : 0.f;
            }
            accum16<HD>(ds, Ks + kt * 16 * LS, dq, c, g);
        }
        if (tail) {                                       // key z
            const float sz = frag_dot_row<HD>(af, Ks + z * LS, g), dpz = frag_dot_row<HD>(bf, Vs + z * LS, g);
            const bool keep_z = Ms[z] != 0;
            const float prz = t_ok ? __expf(((keep_z ? sz : kFill) - lm) - ll) : 0.f;       // p(i, z), this lane's query i
            const float dsz = keep_z ? prz * (dpz - delta) : 0.f;                           // ds(i, z)
            rank1_update<HD>(dsz, Ks + z * LS, dq, c, g);
#pragma unroll
            for (int i = 0; i < 16; ++i) {                // this wave's queries' share of dk_z, dv_z (lane i holds query i's)
                const float wds = __shfl(dsz, i, 64), wpr = __shfl(prz, i, 64);
                zdk += wds * Qs[(t0 + i) * LS + dl];
                zdv += wpr * Ds[(t0 + i) * LS + dl];
            }
        }
#ifdef MSN_ATTN_TIMELINE
        tl4 = __builtin_readcyclecounter();
#endif
        // ---- dK, dV of key tile `wave`: af = k (scaled), bf = v
        lds_frags<HD>(af, Ks, trow, g, p.scale);
        lds_frags<HD>(bf, Vs, trow, g, 1.f);
        const bool keep = t_ok && Ms[trow] != 0;
#pragma unroll
        for (int t = 0; t < DT; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int qt = 0; qt < ntile; ++qt) {
            const f32x4 s = score16<HD>(Qs + qt * 16 * LS, af, c, g);     // rows = queries, col = this lane's key
            const f32x4 dp = score16<HD>(Ds + qt * 16 * LS, bf, c, g);
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qt + 4 * g + r;
                const float e = __expf(((keep ? s[r] : kFill) - Lm[q]) - Ll[q]);
                pr[r] = t_ok ? e : 0.f;
                ds[r] = keep ? e * (dp[r] - Dl[q]) : 0.f;
            }
            accum16<HD>(pr, Ds + qt * 16 * LS, dv, c, g);
            accum16<HD>(ds, Qs + qt * 16 * LS, dk, c, g);
        }
        if (tail) {                                       // query z against this lane's key
            const float sz = frag_dot_row<HD>(af, Qs + z * LS, g), dpz = frag_dot_row<HD>(bf, Ds + z * LS, g);
            const float e = __expf(((keep ? sz : kFill) - Lm[z]) - Ll[z]);
            const float dsj = keep ? e * (dpz - Dl[z]) : 0.f;                               // ds(z, j), this lane's key j
            rank1_update<HD>(t_ok ? e : 0.f, Ds + z * LS, dv, c, g);
            rank1_update<HD>(dsj, Qs + z * LS, dk, c, g);
#pragma unroll
            for (int j = 0; j < 16; ++j) zdq += __shfl(dsj, j, 64) * Ks[(t0 + j) * LS + dl];
        }
        }
    if (tail) {
        if (wave == 0) {                                  // the (z, z) element
            const float szz = wave_dot<HD>(Qs + z * LS, Ks + z * LS, lane) * p.scale;
            const float dpzz = wave_dot<HD>(Ds + z * LS, Vs + z * LS, lane);
            const bool keep_z = Ms[z] != 0;
            const float pzz = __expf(((keep_z ? szz : kFill) - Lm[z]) - Ll[z]);
            const float dszz = keep_z ? pzz * (dpzz - Dl[z]) : 0.f;
            zdq += dszz * Ks[z * LS + dl];
            zdk += dszz * Qs[z * LS + dl];
            zdv += pzz * Ds[z * LS + dl];
        }
        if (lane < HD) {
            Zp[(0 * nw + wave) * HD + lane] = zdq;
            Zp[(1 * nw + wave) * HD + lane] = zdk;
            Zp[(2 * nw + wave) * HD + lane] = zdv;
        }
    }
    MSN_TL(tl5)
    __syncthreads();                                      // every wave is done with the images: their space is the stage
    MSN_TL(tl6)

    // ---- stage [3][T][LS]: dq | dk | dv rows of this (sample, head)
    float* St = smem;
    const size_t SS = (size_t)T * LS;
    if constexpr (SHARE) {
        // the waves' dq parts: P[wave][16 ntile rows][LS] over the images' space, summed in wave order by lane = 16 bytes
        // of a row, then held in registers while the stage takes the space
        const int rows_mf = 16 * ntile;
        float* P = smem + (size_t)wave * rows_mf * LS;
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            if (qt < ntile) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < DT; ++t) P[(16 * qt + 4 * g + r) * LS + 16 * t + c] = dqp[qt][t][r];
            }
        }
        __syncthreads();
        constexpr int Q4 = HD / 4, PCS = 4;               // pieces per thread: 64 rows x 16 pieces on 256 threads
        f32x4 sum[PCS];
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            const int idx = threadIdx.x + u * blockDim.x;
            sum[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (idx < rows_mf * Q4) {
                const int row = idx / Q4, cq = 4 * (idx % Q4);
                for (int w = 0; w < nw; ++w) sum[u] += *reinterpret_cast<const f32x4*>(smem + ((size_t)w * rows_mf + row) * LS + cq);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PCS; ++u) {
            const int idx = threadIdx.x + u * blockDim.x;
            if (idx < rows_mf * Q4) {
                const int row = idx / Q4, cq = 4 * (idx % Q4);
                // (rows T .. rows_mf - 1 are padding of the last query tile: in the stage they would be rows 0 .. of the dk block, which
                //  the waves write below without a barrier in between -- a sequence of 16 n + 15 tokens lost dk of its first key now and then)
                if (row < T) *reinterpret_cast<f32x4*>(St + row * LS + cq) = sum[u] * p.scale;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = t0 + 4 * g + r;
        if (row < T) {
#pragma unroll
            for (int t = 0; t < DT; ++t) {
                if constexpr (!SHARE) St[row * LS + 16 * t + c] = dq[t][r] * p.scale;
                St[SS + row * LS + 16 * t + c] = dk[t][r] * p.scale;
                St[2 * SS + row * LS + 16 * t + c] = dv[t][r];
            }
        }
    }
    if (tail) {                                           // row z: the waves' shares in wave order
        for (int u = threadIdx.x; u < 3 * HD; u += blockDim.x) {
            const int sq = u / HD, d = u % HD;
            float a0 = 0.f;
            for (int w = 0; w < nw; ++w) a0 += Zp[(sq * nw + w) * HD + d];
            St[sq * SS + z * LS + d] = sq < 2 ? a0 * p.scale : a0;
        }
    }
    __syncthreads();
    // (compile-time divisors: an integer division by a run-time value costs more here than the 16 bytes it places)
    if constexpr (NP == 0) {
        constexpr int Q4 = HD / 4;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float* base = s == 0 ? p.dq + (int64_t)b * p.dq_bs : s == 1 ? p.dk + (int64_t)b * p.dk_bs : p.dv + (int64_t)b * p.dv_bs;
            const int64_t ld = s == 0 ? p.lddq : s == 1 ? p.lddk : p.lddv;
            for (int idx = threadIdx.x; idx < T * Q4; idx += blockDim.x) {
                const int row = idx / Q4, cq = 4 * (idx % Q4);
                if (cq < p.hd)
                    *reinterpret_cast<float4*>(base + (int64_t)row * ld + col0 + cq) =
                        *reinterpret_cast<const float4*>(St + s * SS + row * LS + cq);
            }
        }
    } else {
        constexpr int NB = HD / 16;                       // the plane form takes whole 16-column blocks: hd == HD
        const int e = p.H * HD;
        const int64_t r0 = (int64_t)b * T;
        // a lane owns 8 columns of one row: consecutive lanes are consecutive 16-byte chunks of ONE plane image (row r, half h at
        // byte 32 r + 16 h), so a wave's store is 1 KB of contiguous bytes per image instead of 64 separate 32-byte segments
        // (measured: the segment form kept the CU's texture path busy for ~6 000 cycles per workgroup)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            unsigned char* cbase = fo.planes + (int64_t)((s * e + col0) / 16) * (NP * 1024);
            for (int idx = threadIdx.x; idx < 2 * T * NB; idx += blockDim.x) {
                int j = 0;
#pragma unroll
                for (int q = 1; q < NB; ++q) j += idx >= 2 * T * q ? 1 : 0;
                const int rh = idx - 2 * T * j, row = rh >> 1, half = rh & 1;
                const float* src = St + s * SS + row * LS + 16 * j + 8 * half;
                const float4 w0 = *reinterpret_cast<const float4*>(src), w1 = *reinterpret_cast<const float4*>(src + 4);
                float v[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                const int64_t gr = r0 + row;
                unsigned char* dst = cbase + ((gr >> 5) * fo.cb + j) * (int64_t)(NP * 1024) + (int)(gr & 31) * 32 + 16 * half;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    unsigned w[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const unsigned short lo = bf16_rne_bits(v[2 * x]), hi = bf16_rne_bits(v[2 * x + 1]);
                        v[2 * x] -= __uint_as_float((unsigned)lo << 16);          // exact
                        v[2 * x + 1] -= __uint_as_float((unsigned)hi << 16);
                        w[x] = lo | ((unsigned)hi << 16);
                    }
                    *reinterpret_cast<uint4*>(dst + k * 1024) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
        if (fo.colpart) {                                 // this sample's column sums, rows in order
            for (int u = threadIdx.x; u < 3 * HD; u += blockDim.x) {
                const int s = u / HD, d = u % HD;
                const float* src = St + s * SS + d;
                float a0 = 0.f, a1 = 0.f;
                int row = 0;
                for (; row + 1 < T; row += 2) a0 += src[row * LS], a1 += src[(row + 1) * LS];
                if (row < T) a0 += src[row * LS];
                fo.colpart[(int64_t)b * 3 * e + s * e + col0 + d] = a0 + a1;
            }
        }
    }
#ifdef MSN_ATTN_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < kDbgRows) {
        unsigned long long* d = g_mattn_dbg + 8 * blockIdx.x;
        d[0] = tl0, d[1] = tl1, d[2] = tl2, d[3] = tl3, d[4] = tl4, d[5] = tl5, d[6] = tl6, d[7] = __builtin_readcyclecounter();
    }
#endif
}

// =====================================================================================================================
// Long sequences (more than 256 tokens: the reference's spectrum transformer on 1024-bin spectra).  Same tile products;
// a workgroup owns a block of up to 128 rows of the fixed operand (8 waves, compiled for two workgroups per CU so that one
// multiplies while the other stages its next chunk) and the streamed operand passes through
// LDS in chunks of CH = 16 * KEEP rows (256 for heads up to 16 wide, 128 beyond), so the score tiles of a chunk still
// fit in registers.  Forward: online softmax ACROSS chunks (running maximum; the output accumulators -- rows are
// queries 4g + r -- take the rescale factor of their query from its column-owner lane), one pass WITHIN a chunk.
// The backward kernels only accumulate over the chunks (probabilities come from the saved row statistics).
// Workgroup -> (sample, head, row block) by locate_block: the workgroups of a sample run on one XCD and share its lines in L2.
template <int HD>
__global__ __launch_bounds__(512, HD > 64 ? 2 : 4) void mattn_fwd_long_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16, KEEP = HD <= 16 ? 16 : 8, CH = 16 * KEEP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (size_t)CH * LS;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + (size_t)CH * LS);
    MSN_TL(tl0)
    const int NB = (p.Tq + 127) / 128;
    int b, hh, blk;
    locate_block(p, NB, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = blk * 128 + wave * 16, qrow = q0 + c;
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    float4 qf[DT];
    Stager<HD, (HD > 32 ? 2 : CH * (HD / 4) / 512), (HD > 32)> sv;   // wide heads: the rest in a second pass (registers)
    uint8_t mk = 1;
    const int tj = threadIdx.x;
    // a chunk's K / V rows and key mask, requested together (see Stager); the first chunk beside the query fragments
    auto request = [&](int k0) {
        const int nt = min(CH, p.Tk - k0), TPc = (nt + 15) / 16 * 16;
        sv.request(ksrc + (int64_t)k0 * p.ldk, vsrc + (int64_t)k0 * p.ldv, p.ldk, p.ldv, col0, nt, TPc, p.hd);
        if (p.mask) mk = p.mask[(int64_t)b * p.Tk + k0 + (tj < nt ? tj : 0)];
    };
    frags_request<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.hd);
    if (HD <= 32) request(0);
    MSN_TL(tla)
    frags_commit<HD>(qf, qrow, p.Tq, g, p.scale, p.hd);
    float m = -INFINITY, l = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.Tk; k0 += CH) {
        const int nt = min(CH, p.Tk - k0), TPc = (nt + 15) / 16 * 16, nkt = TPc / 16;
        __syncthreads();                                  // every wave is done with the previous chunk
        if (k0 > 0 || HD > 32) request(k0);               // (narrow heads: chunk 0 was requested beside the fragments)
        sv.commit_first(Ks);
        sv.commit_second(Ks, Vs);
        MSN_TL(tlc)
        if (tj < TPc) Ms[tj] = tj < nt ? (mk ? 1 : 0) : 2;    // key codes: 1 live, 0 masked out, 2 beyond the sequence
        for (int j = tj + blockDim.x; j < TPc; j += blockDim.x)
            Ms[j] = j < nt ? (p.mask ? (p.mask[(int64_t)b * p.Tk + k0 + j] ? 1 : 0) : 1) : 2;
        __syncthreads();
        MSN_TL(tl1)
        f32x4 sc[KEEP];
        float mc = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KEEP; ++kt) {
            if (kt < nkt) {
                const uint32_t codes = reinterpret_cast<const uint32_t*>(Ms)[4 * kt + g];   // this lane group's 4 keys
                sc[kt] = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t code = (codes >> (8 * r)) & 0xffu;     // branch-free selection
                    sc[kt][r] = code == 1u ? sc[kt][r] : (code == 0u ? kFill : -INFINITY);
                    mc = fmaxf(mc, sc[kt][r]);
                }
            }
        }
        mc = group_max4(mc);
        const float mn = fmaxf(m, mc);                    // finite: a chunk holds at least one key, masked ones score -1e7
        const float alpha = __expf(m - mn);               // exp(-inf) = 0 on the first chunk
        m = mn;
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
            for (int t = 0; t < DT; ++t) o[t][r] *= ar;
        }
#pragma unroll
        for (int kt = 0; kt < KEEP; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(sc[kt][r] - m);
                    sc[kt][r] = e;
                    l += e;
                }
                accum16<HD>(sc[kt], Vs + kt * 16 * LS, o, c, g);
            }
        }
#ifdef MSN_ATTN_TIMELINE
        if (threadIdx.x == 0 && blockIdx.x < kDbgRows && k0 == 0) {
            unsigned long long* d = g_mattn_dbg + 8 * blockIdx.x;
            d[0] = tl0, d[1] = tla, d[2] = tlc, d[3] = tl1, d[4] = __builtin_readcyclecounter();
        }
#endif
    }
    // (the lane's place in the output from an OPAQUE copy of the thread index: kept from the kernel's start, these registers -- and
    //  the width bound __shfl_xor derives from the lane id -- were what the 16-wide instantiation spilled around its chunk loop:
    //  20 bytes of scratch at 128 VGPRs)
    int te = (int)threadIdx.x;
    asm volatile("" : "+v"(te));
    const int ce = te & 15, ge = (te >> 4) & 3, q0e = blk * 128 + (te >> 6) * 16;
    l += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((te ^ 16) & 63) << 2, __builtin_bit_cast(int, l)));
    l += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((te ^ 32) & 63) << 2, __builtin_bit_cast(int, l)));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lq = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((te & 48) + 4 * ge + r) << 2, __builtin_bit_cast(int, l)));
        const int q = q0e + 4 * ge + r;
        if (q < p.Tq) {
            float* op = p.out + (int64_t)b * p.o_bs + (int64_t)q * p.ldo + col0 + ce;
            const float inv = 1.f / lq;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + ce < p.hd) op[16 * t] = o[t][r] * inv;
        }
    }
    if (ge == 0 && q0e + ce < p.Tq) {
        float* st = p.lse + 2 * (((int64_t)b * p.H + hh) * p.Tq + q0e + ce);
        st[0] = m;
        st[1] = __logf(l);
    }
#ifdef MSN_ATTN_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && blockIdx.x < kDbgRows) g_mattn_dbg[8 * blockIdx.x + 5] = __builtin_readcyclecounter();
#endif
}

template <int HD>
__global__ __launch_bounds__(512, HD > 64 ? 2 : 4) void mattn_bwd_dq_long_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16, KEEP = HD <= 16 ? 16 : 8, CH = 16 * KEEP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + (size_t)CH * LS;
    uint8_t* Ms = reinterpret_cast<uint8_t*>(Vs + (size_t)CH * LS);
    const int NB = (p.Tq + 127) / 128;
    int b, hh, blk;
    locate_block(p, NB, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int q0 = blk * 128 + wave * 16, qrow = q0 + c;
    const bool q_ok = qrow < p.Tq;
    const float* ksrc = p.k + (int64_t)b * p.k_bs;
    const float* vsrc = p.v + (int64_t)b * p.v_bs;
    float4 qf[DT], df[DT], of[DT];
    Stager<HD, (HD > 32 ? 2 : CH * (HD / 4) / 512), (HD > 32)> sv;   // wide heads: the rest in a second pass (registers)
    uint8_t mk = 1;
    const int tj = threadIdx.x;
    auto request = [&](int k0) {                          // see the forward kernel
        const int nt = min(CH, p.Tk - k0), TPc = (nt + 15) / 16 * 16;
        sv.request(ksrc + (int64_t)k0 * p.ldk, vsrc + (int64_t)k0 * p.ldv, p.ldk, p.ldv, col0, nt, TPc, p.hd);
        if (p.mask) mk = p.mask[(int64_t)b * p.Tk + k0 + (tj < nt ? tj : 0)];
    };
    frags_request<HD>(qf, p.q + (int64_t)b * p.q_bs, p.ldq, col0, qrow, p.Tq, g, p.hd);
    frags_request<HD>(df, p.dout + (int64_t)b * p.d_bs, p.ldd, col0, qrow, p.Tq, g, p.hd);
    frags_request<HD>(of, p.o + (int64_t)b * p.o_bs, p.ldo, col0, qrow, p.Tq, g, p.hd);
    const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + (q_ok ? qrow : 0);
    const float lm = p.lse[2 * stat], ll = p.lse[2 * stat + 1];   // row 0's for a padded query: never used (q_ok)
    if (HD <= 32) request(0);
    frags_commit<HD>(qf, qrow, p.Tq, g, p.scale, p.hd);
    frags_commit<HD>(df, qrow, p.Tq, g, 1.f, p.hd);
    frags_commit<HD>(of, qrow, p.Tq, g, 1.f, p.hd);
    float delta = 0.f;
#pragma unroll
    for (int x = 0; x < DT; ++x)
        delta += df[x].x * of[x].x + df[x].y * of[x].y + df[x].z * of[x].z + df[x].w * of[x].w;
    delta = group_sum4(delta);
    if (g == 0 && q_ok) p.delta[stat] = delta;
    f32x4 dq[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.Tk; k0 += CH) {
        const int nt = min(CH, p.Tk - k0), TPc = (nt + 15) / 16 * 16, nkt = TPc / 16;
        __syncthreads();
        if (k0 > 0 || HD > 32) request(k0);               // (narrow heads: chunk 0 was requested beside the fragments)
        sv.commit_first(Ks);
        sv.commit_second(Ks, Vs);
        if (tj < TPc) Ms[tj] = tj < nt ? (mk ? 1 : 0) : 2;    // key codes: 1 live, 0 masked out, 2 beyond the sequence
        for (int j = tj + blockDim.x; j < TPc; j += blockDim.x)
            Ms[j] = j < nt ? (p.mask ? (p.mask[(int64_t)b * p.Tk + k0 + j] ? 1 : 0) : 1) : 2;
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const uint32_t codes = reinterpret_cast<const uint32_t*>(Ms)[4 * kt + g];
            const f32x4 s = score16<HD>(Ks + kt * 16 * LS, qf, c, g);
            const f32x4 dp = score16<HD>(Vs + kt * 16 * LS, df, c, g);
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = q_ok && ((codes >> (8 * r)) & 0xffu) == 1u;
                ds[r] = live ? __expf((s[r] - lm) - ll) * (dp[r] - delta) : 0.f;
            }
            accum16<HD>(ds, Ks + kt * 16 * LS, dq, c, g);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + 4 * g + r;
        if (q < p.Tq) {
            float* op = p.dq + (int64_t)b * p.dq_bs + (int64_t)q * p.lddq + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) op[16 * t] = dq[t][r] * p.scale;
        }
    }
}

template <int HD>
__global__ __launch_bounds__(512, HD > 64 ? 2 : 4) void mattn_bwd_dkv_long_kernel(const MAttn p) {
    constexpr int LS = HD + 4, DT = HD / 16, KEEP = HD <= 16 ? 16 : 8, CH = 16 * KEEP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;
    float* Ds = smem + (size_t)CH * LS;
    float* Lm = Ds + (size_t)CH * LS;
    float* Ll = Lm + CH;
    float* Dl = Ll + CH;
    const int NB = (p.Tk + 127) / 128;
    int b, hh, blk;
    locate_block(p, NB, b, hh, blk);
    const int col0 = hh * p.hd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int k0 = blk * 128 + wave * 16, krow = k0 + c;
    const float* qsrc = p.q + (int64_t)b * p.q_bs;
    const float* dsrc = p.dout + (int64_t)b * p.d_bs;
    float4 kf[DT], vf[DT];
    Stager<HD, (HD > 32 ? 2 : CH * (HD / 4) / 512), (HD > 32)> sv;   // wide heads: the rest in a second pass (registers)
    float s_m = 0.f, s_l = 0.f, s_d = 0.f;
    const int tj = threadIdx.x;
    auto request = [&](int i0) {                          // a chunk's Q / dO rows and row statistics, requested together
        const int nt = min(CH, p.Tq - i0), TPc = (nt + 15) / 16 * 16;
        sv.request(qsrc + (int64_t)i0 * p.ldq, dsrc + (int64_t)i0 * p.ldd, p.ldq, p.ldd, col0, nt, TPc, p.hd);
        const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + i0 + (tj < nt ? tj : 0);
        s_m = p.lse[2 * stat], s_l = p.lse[2 * stat + 1], s_d = p.delta[stat];
    };
    frags_request<HD>(kf, p.k + (int64_t)b * p.k_bs, p.ldk, col0, krow, p.Tk, g, p.hd);
    frags_request<HD>(vf, p.v + (int64_t)b * p.v_bs, p.ldv, col0, krow, p.Tk, g, p.hd);
    const bool in_seq = krow < p.Tk;
    uint8_t mk = 1;
    if (p.mask) mk = p.mask[(int64_t)b * p.Tk + (in_seq ? krow : 0)];
    if (HD <= 32) request(0);
    frags_commit<HD>(kf, krow, p.Tk, g, p.scale, p.hd);
    frags_commit<HD>(vf, krow, p.Tk, g, 1.f, p.hd);
    const bool keep = in_seq && mk != 0;
    f32x4 dk[DT], dv[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) dk[t] = dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < p.Tq; i0 += CH) {
        const int nt = min(CH, p.Tq - i0), TPc = (nt + 15) / 16 * 16, nqt = TPc / 16;
        __syncthreads();
        if (i0 > 0 || HD > 32) request(i0);               // (narrow heads: chunk 0 was requested beside the fragments)
        sv.commit_first(Qs);
        sv.commit_second(Qs, Ds);
        if (tj < TPc) {
            Lm[tj] = tj < nt ? s_m : INFINITY;            // +inf: a padded query row gets p = exp(-inf) = 0
            Ll[tj] = tj < nt ? s_l : 0.f;
            Dl[tj] = tj < nt ? s_d : 0.f;
        }
        for (int t = tj + blockDim.x; t < TPc; t += blockDim.x) {
            const int64_t stat = ((int64_t)b * p.H + hh) * p.Tq + i0 + (t < nt ? t : 0);
            Lm[t] = t < nt ? p.lse[2 * stat] : INFINITY;     // +inf: a padded query row gets p = exp(-inf) = 0
            Ll[t] = t < nt ? p.lse[2 * stat + 1] : 0.f;
            Dl[t] = t < nt ? p.delta[stat] : 0.f;
        }
        __syncthreads();
        for (int qt = 0; qt < nqt; ++qt) {
            const f32x4 s = score16<HD>(Qs + qt * 16 * LS, kf, c, g);     // rows = queries, col = this lane's key
            const f32x4 dp = score16<HD>(Ds + qt * 16 * LS, vf, c, g);
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * qt + 4 * g + r;
                const float e = __expf(((keep ? s[r] : kFill) - Lm[q]) - Ll[q]);
                pr[r] = in_seq ? e : 0.f;
                ds[r] = keep ? e * (dp[r] - Dl[q]) : 0.f;
            }
            accum16<HD>(pr, Ds + qt * 16 * LS, dv, c, g);
            accum16<HD>(ds, Qs + qt * 16 * LS, dk, c, g);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + 4 * g + r;
        if (k < p.Tk) {
            float* ok = p.dk + (int64_t)b * p.dk_bs + (int64_t)k * p.lddk + col0 + c;
            float* ov = p.dv + (int64_t)b * p.dv_bs + (int64_t)k * p.lddv + col0 + c;
#pragma unroll
            for (int t = 0; t < DT; ++t)
                if (16 * t + c < p.hd) {
                    ok[16 * t] = dk[t][r] * p.scale;
                    ov[16 * t] = dv[t][r];
                }
        }
    }
}

template <typename K>
static int launch_big_lds(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, const MAttn& a) {
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            set_error("attention: cannot reserve %zu bytes of LDS", lds);
            return MSN_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, st, a);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// Is the matrix-core path applicable?  (16-B aligned operands, head width a multiple of 4 up to 128: a width that is not a
// multiple of 16 runs as the next one, its missing columns zeros in LDS / registers only.)
static int padded_hd(int hd) { return (hd + 15) / 16 * 16; }
bool mattn_applicable(const MAttn& a, bool shared_q) {
    if (a.hd % 4 != 0 || a.hd > 128 || a.hd < 4 || a.Tq > 65535 || a.Tk > 65535 || (a.q_bs == 0 && !shared_q)) return false;
    if ((int64_t)a.B * a.H * ((std::max(a.Tq, a.Tk) + 127) / 128) > 0x7fffffffLL) return false;
    const int64_t lds[] = {a.ldq, a.ldk, a.ldv, a.q_bs, a.k_bs, a.v_bs};
    for (int64_t v : lds)
        if (v % 4 != 0) return false;
    const void* ptrs[] = {a.q, a.k, a.v};
    for (const void* ptr : ptrs)
        if (reinterpret_cast<uintptr_t>(ptr) & 15) return false;
    return true;
}

#define MSN_MATTN_DISPATCH(KERNEL, ...)                                                    \
    switch (padded_hd(a.hd)) {                                                                      \
        case 16: rc = launch_big_lds(KERNEL<16>, __VA_ARGS__); break;                      \
        case 32: rc = launch_big_lds(KERNEL<32>, __VA_ARGS__); break;                      \
        case 48: rc = launch_big_lds(KERNEL<48>, __VA_ARGS__); break;                      \
        case 64: rc = launch_big_lds(KERNEL<64>, __VA_ARGS__); break;                      \
        case 80: rc = launch_big_lds(KERNEL<80>, __VA_ARGS__); break;                      \
        case 96: rc = launch_big_lds(KERNEL<96>, __VA_ARGS__); break;                      \
        case 112: rc = launch_big_lds(KERNEL<112>, __VA_ARGS__); break;                    \
        default: rc = launch_big_lds(KERNEL<128>, __VA_ARGS__); break;                     \
    }

// ragged-token path: self-attention over 16n + 1 tokens, one-pass forward (n + 1 <= 8 tiles), both weight passes of
// the tail wave inside 128 rows
static bool use_tail(const MAttn& a) { return a.Tq == a.Tk && a.Tk % 16 == 1 && a.Tk > 16 && a.Tk <= 113 && a.hd <= 64; }

static bool is_long(const MAttn& a) { return a.Tq > 128 || a.Tk > 128; }
static int chunk_rows(int hd) { return padded_hd(hd) <= 16 ? 256 : 128; }
static unsigned long_block(int T) { return 64u * (unsigned)std::min(8, (T + 15) / 16); }

bool pattn_forward_aligned(const MAttn& a);
bool pattn_backward_aligned(const MAttn& a);

int mattn_forward(const MAttn& a0, hipStream_t st) {
    MAttn a = a0;
    a.tail = use_tail(a) ? 1 : 0;
    // long sequences of narrow heads: fp32-grade products on the bf16 matrix cores (attention_planes.hip)
    if (is_long(a) && pattn_applicable(a) && pattn_forward_aligned(a)) return pattn_forward(a, st);
    if (is_long(a)) {
        const int CH = chunk_rows(a.hd), NB = (a.Tq + 127) / 128;
        const size_t lds = sizeof(float) * 2 * (size_t)CH * (padded_hd(a.hd) + 4) + (size_t)CH;
        int rc;
        MSN_MATTN_DISPATCH(mattn_fwd_long_kernel, dim3(a.B * a.H * NB), dim3(long_block(a.Tq)), lds, st, a)
        return rc;
    }
    const int TPk = (a.Tk + 15) / 16 * 16, nq = (a.Tq + 15) / 16;
    const int rows = a.tail ? (a.Tk + 3) / 4 * 4 : TPk;
    const size_t lds = sizeof(float) * (2 * (size_t)rows * (padded_hd(a.hd) + 4) + kTailScratch) + (size_t)TPk;
    int rc;
    MSN_MATTN_DISPATCH(mattn_fwd_kernel, dim3(a.B * a.H), dim3(64 * nq), lds, st, a)
    return rc;
}

// One-pass backward (mattn_bwd_fused_kernel): self-attention, up to 128 tokens, heads up to 64 wide, everything 16-byte aligned.
static int g_attn_fused = 1;
static int g_attn_share = 1;
void mattn_set_fused(int on) { g_attn_fused = on & 1, g_attn_share = (on & 2) ? 0 : 1; }      // bit 1: the 7-product form of the one-pass kernel
static size_t fused_lds(const MAttn& a, bool tail) {
    const int TP = (a.Tk + 15) / 16 * 16;
    const int rows = tail ? a.Tk : TP;
    const int waves = tail ? a.Tk / 16 : TP / 16;
    const size_t ragged = tail ? 3 * (size_t)waves * padded_hd(a.hd) : 0;          // the waves' shares of row z
    const size_t patches = (size_t)waves * 256;                                     // a 16 x 16 transpose patch per wave
    return sizeof(float) * (4 * (size_t)rows * (padded_hd(a.hd) + 4) + 3 * (size_t)TP + ragged + patches) + (size_t)TP;
}
bool mattn_fused_applicable(const MAttn& a) {
    if (!mattn_applicable(a) || a.Tq != a.Tk || a.Tk > 128 || a.hd > 64 || a.hd % 4 != 0) return false;
    const int64_t al[] = {a.ldd, a.d_bs, a.ldo, a.o_bs};
    bool ok = ((reinterpret_cast<uintptr_t>(a.dout) | reinterpret_cast<uintptr_t>(a.o)) & 15) == 0;
    for (int64_t v : al) ok = ok && (v % 4 == 0);
    return ok && fused_lds(a, use_tail(a)) <= 160 * 1024;
}
template <int NP>
static int launch_fused(const MAttn& a, const FusedOut& fo, hipStream_t st) {
    const size_t lds = fused_lds(a, a.tail != 0);
    const dim3 grid(a.B * a.H), block(64 * (a.tail ? a.Tk / 16 : (a.Tk + 15) / 16));      // one wave per full 16-row tile
    // one recomputation for dQ and dK,dV where the query tiles' dQ parts fit registers and the images' space: up to 4 full tiles
    const int full_tiles = a.tail ? a.Tk / 16 : (a.Tk + 15) / 16;
    const bool share = g_attn_share && full_tiles <= 4;
#define MSN_FUSED_CASE(HDV)                                                                                               \
    {                                                                                                                     \
        auto kern = share ? mattn_bwd_fused_kernel<HDV, NP, true> : mattn_bwd_fused_kernel<HDV, NP, false>;               \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
            set_error("attention: cannot reserve %zu bytes of LDS", lds);                                                 \
            return MSN_ERR_HIP;                                                                                           \
        }                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, block, lds, st, a, fo);                                                            \
    }
    switch (padded_hd(a.hd)) {
        case 16: MSN_FUSED_CASE(16) break;
        case 32: MSN_FUSED_CASE(32) break;
        case 48: MSN_FUSED_CASE(48) break;
        default: MSN_FUSED_CASE(64) break;
    }
#undef MSN_FUSED_CASE
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
// dqkv as planes (+ per-sample column sums): q, k, v are the three column groups of ONE packed matrix
int mattn_backward_planes(const MAttn& a0, int planes, unsigned char* out, int cb, float* colpart, hipStream_t st) {
    MAttn a = a0;
    a.tail = use_tail(a) ? 1 : 0;
    FusedOut fo = {out, colpart, cb};
    return planes == 3 ? launch_fused<3>(a, fo, st) : launch_fused<2>(a, fo, st);
}

int mattn_backward(const MAttn& a0, hipStream_t st) {
    MAttn a = a0;
    a.tail = use_tail(a) ? 1 : 0;
    const int TPk = (a.Tk + 15) / 16 * 16, TPq = (a.Tq + 15) / 16 * 16;
    int rc;
    const int64_t al[] = {a.ldd, a.d_bs, a.ldo, a.o_bs};
    bool ok = ((reinterpret_cast<uintptr_t>(a.dout) | reinterpret_cast<uintptr_t>(a.o)) & 15) == 0;
    for (int64_t v : al) ok = ok && (v % 4 == 0);
    if (!ok) {
        set_error("attention backward: out / dout must be 16-byte aligned with strides %% 4 == 0");
        return MSN_ERR_SHAPE;
    }
    if (g_attn_fused && mattn_fused_applicable(a)) return launch_fused<0>(a, FusedOut{nullptr, nullptr, 0}, st);
    if (is_long(a) && pattn_applicable(a) && pattn_backward_aligned(a)) return pattn_backward(a, st);
    if (is_long(a)) {
        const int CH = chunk_rows(a.hd);
        {
            const size_t lds = sizeof(float) * 2 * (size_t)CH * (padded_hd(a.hd) + 4) + (size_t)CH;
            MSN_MATTN_DISPATCH(mattn_bwd_dq_long_kernel, dim3(a.B * a.H * ((a.Tq + 127) / 128)), dim3(long_block(a.Tq)), lds, st, a)
            if (rc != MSN_OK) return rc;
        }
        const size_t lds = sizeof(float) * (2 * (size_t)CH * (padded_hd(a.hd) + 4) + 3 * (size_t)CH);
        MSN_MATTN_DISPATCH(mattn_bwd_dkv_long_kernel, dim3(a.B * a.H * ((a.Tk + 127) / 128)), dim3(long_block(a.Tk)), lds, st, a)
        return rc;
    }
    {
        const int rows = a.tail ? (a.Tk + 3) / 4 * 4 : TPk;
        const size_t lds = sizeof(float) * (2 * (size_t)rows * (padded_hd(a.hd) + 4) + kTailScratch) + (size_t)TPk;
        MSN_MATTN_DISPATCH(mattn_bwd_dq_kernel, dim3(a.B * a.H), dim3(64 * (TPq / 16)), lds, st, a)
        if (rc != MSN_OK) return rc;
    }
    {
        const int rows = a.tail ? (a.Tq + 3) / 4 * 4 : TPq;
        const size_t lds = sizeof(float) * (2 * (size_t)rows * (padded_hd(a.hd) + 4) + 3 * (size_t)TPq + kTailScratch);
        MSN_MATTN_DISPATCH(mattn_bwd_dkv_kernel, dim3(a.B * a.H), dim3(64 * (TPk / 16)), lds, st, a)
    }
    return rc;
}

}  // namespace msn

using namespace msn;

extern "C" int msn_set_attention_fused(int on) {
    MSN_REQUIRE(on >= 0 && on <= 3, "msn_set_attention_fused: 0, 1 or 3");
    mattn_set_fused(on);
    return MSN_OK;
}

extern "C" size_t msn_attention_bwd_planes_workspace_bytes(int B, int H, int head_dim) {
    return sizeof(float) * ((size_t)B + COLSUM_SLICES) * 3 * (size_t)H * (size_t)head_dim;
}

// Self-attention forward over a packed qkv matrix (rows = (sample, token), columns q | k | v) whose output leaves BOTH as the fp32
// matrix the backward reads and as the plane matrix the output projection multiplies (ref src/transformer_utils.py:36-89: the
// `unifyheads` Linear follows the attention directly) -- the split pass between the two launches is gone.
extern "C" int msn_attention_fwd_planes(const float* qkv, int64_t ldqkv, const uint8_t* key_mask, int B, int H, int T, int head_dim,
                                        float scale, float* out, int64_t ldo, float* lse, int planes, void* out_planes,
                                        msn_stream_t stream) {
    MSN_REQUIRE(qkv && out && lse && out_planes, "msn_attention_fwd_planes: null pointer");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_attention_fwd_planes: planes must be 2 or 3 (got %d)", planes);
    MSN_REQUIRE(B > 0 && H > 0 && T > 0 && head_dim > 0, "msn_attention_fwd_planes: bad shape");
    MSN_REQUIRE(head_dim % 16 == 0 && head_dim <= 64, "msn_attention_fwd_planes: head width %d (16, 32, 48 or 64)", head_dim);
    const int e = H * head_dim;
    MSN_REQUIRE(ldqkv >= 3 * (int64_t)e && ldo >= e, "msn_attention_fwd_planes: row strides shorter than the rows");
    MSN_REQUIRE((reinterpret_cast<uintptr_t>(out_planes) & 15) == 0, "msn_attention_fwd_planes: the plane matrix must be 16-byte aligned");
    MAttn m = {};
    m.q = qkv; m.k = qkv + e; m.v = qkv + 2 * e; m.out = out;
    m.mask = key_mask; m.lse = lse;
    m.ldq = m.ldk = m.ldv = ldqkv; m.ldo = ldo;
    m.q_bs = m.k_bs = m.v_bs = (int64_t)T * ldqkv; m.o_bs = (int64_t)T * ldo;
    m.B = B; m.H = H; m.Tq = m.Tk = T; m.hd = head_dim; m.scale = scale;
    MSN_REQUIRE(mattn_applicable(m) && T <= 128,
                "msn_attention_fwd_planes: up to 128 tokens, 16-byte aligned operands, row strides %% 4 == 0 (T = %d, head %d)", T, head_dim);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t M = (int64_t)B * T;
    const int cb = 2 * (int)cdiv(e, 32);
    unsigned char* dst = static_cast<unsigned char*>(out_planes);
    if (M % 32 != 0) {            // rows behind the matrix in its last row block are zeros
        const size_t blk = (size_t)cb * planes * 1024;
        if (hipMemsetAsync(dst + (size_t)(M / 32) * blk, 0, blk, st) != hipSuccess) {
            set_error("msn_attention_fwd_planes: memset failed");
            return MSN_ERR_HIP;
        }
    }
    if (e % 32 != 0) {            // e % 32 == 16: the last column block of every row block is padding; the kernel stores real blocks only
        const size_t blk = (size_t)planes * 1024;
        if (hipMemset2DAsync(dst + (size_t)(cb - 1) * blk, (size_t)cb * blk, 0, blk, (size_t)cdiv(M, 32), st) != hipSuccess) {
            set_error("msn_attention_fwd_planes: memset of the padding column block failed");
            return MSN_ERR_HIP;
        }
    }
    m.oplanes = dst; m.o_np = planes; m.o_cb = cb;
    return mattn_forward(m, st);
}

extern "C" int msn_attention_bwd_planes(const float* qkv, int64_t ldqkv, const uint8_t* key_mask, int B, int H, int T,
                                        int head_dim, float scale, const float* out, int64_t ldo, const float* lse,
                                        const float* dout, int64_t ldd, int planes, void* dqkv_planes, float* colsum_out,
                                        void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(qkv && out && lse && dout && dqkv_planes, "msn_attention_bwd_planes: null pointer");
    MSN_REQUIRE(planes == 2 || planes == 3, "msn_attention_bwd_planes: planes must be 2 or 3 (got %d)", planes);
    MSN_REQUIRE(B > 0 && H > 0 && T > 0 && head_dim > 0, "msn_attention_bwd_planes: bad shape");
    MSN_REQUIRE(head_dim % 16 == 0 && head_dim <= 64, "msn_attention_bwd_planes: head width %d (16, 32, 48 or 64)", head_dim);
    const int e = H * head_dim;
    MSN_REQUIRE(ldqkv >= 3 * (int64_t)e && ldo >= e && ldd >= e, "msn_attention_bwd_planes: row strides shorter than the rows");
    MSN_REQUIRE((reinterpret_cast<uintptr_t>(dqkv_planes) & 15) == 0, "msn_attention_bwd_planes: the plane matrix must be 16-byte aligned");
    MAttn m = {};
    m.q = qkv; m.k = qkv + e; m.v = qkv + 2 * e; m.o = out; m.dout = dout;
    m.mask = key_mask; m.lse = const_cast<float*>(lse);
    m.ldq = m.ldk = m.ldv = ldqkv; m.ldo = ldo; m.ldd = ldd;
    m.q_bs = m.k_bs = m.v_bs = (int64_t)T * ldqkv; m.o_bs = (int64_t)T * ldo; m.d_bs = (int64_t)T * ldd;
    m.B = B; m.H = H; m.Tq = m.Tk = T; m.hd = head_dim; m.scale = scale;
    MSN_REQUIRE(mattn_fused_applicable(m),
                "msn_attention_bwd_planes: up to 128 tokens, 16-byte aligned operands, row strides %% 4 == 0 (T = %d, head %d)", T, head_dim);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = nullptr;
    if (colsum_out) {
        MSN_REQUIRE(ws && ws_bytes >= msn_attention_bwd_planes_workspace_bytes(B, H, head_dim),
                    "msn_attention_bwd_planes: workspace too small for the column sums");
        part = static_cast<float*>(ws);
    }
    const int64_t M = (int64_t)B * T;
    const int cb = 2 * (int)cdiv(3 * e, 32);
    unsigned char* dst = static_cast<unsigned char*>(dqkv_planes);
    if (M % 32 != 0) {            // rows behind the matrix in its last row block are zeros
        const size_t blk = (size_t)cb * planes * 1024;
        if (hipMemsetAsync(dst + (size_t)(M / 32) * blk, 0, blk, st) != hipSuccess) {
            set_error("msn_attention_bwd_planes: memset failed");
            return MSN_ERR_HIP;
        }
    }
    if ((3 * e) % 32 != 0) {      // 3e % 32 == 16: the last column block of every row block is padding; the kernel stores real blocks only
        const size_t blk = (size_t)planes * 1024;
        if (hipMemset2DAsync(dst + (size_t)(cb - 1) * blk, (size_t)cb * blk, 0, blk, (size_t)cdiv(M, 32), st) != hipSuccess) {
            set_error("msn_attention_bwd_planes: memset of the padding column block failed");
            return MSN_ERR_HIP;
        }
    }
    if (int rc = mattn_backward_planes(m, planes, dst, cb, part, st)) return rc;
    if (colsum_out) return colsum_finish(part, B, 3 * e, colsum_out, st);
    return MSN_OK;
}
