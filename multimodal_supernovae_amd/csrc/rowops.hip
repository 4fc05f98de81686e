// Row-wise HBM-bound kernels: LayerNorm fwd/bwd, L2-normalise fwd/bwd, time/band embedding
// fwd/bwd, masked pooling fwd/bwd.  All are streaming kernels: 16-byte loads, a row is owned by a
// group of LPR lanes of one wave (LPR = 4 / 16 / 64 by row length) so the row statistics are
// wave-shuffle reductions; column reductions (d gamma, d beta, d embedding weights) are
// block partials in caller scratch + a fixed-order final pass (deterministic, no atomics).
#include <algorithm>
#include <cstdlib>
#include <math.h>

#include "msn_common.h"

namespace msn {

constexpr int NCH = 4;  // 16-byte chunks per lane -> rows of up to LPR * 16 floats

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct RowGeom {
    int64_t rows;
    int cols;      // multiple of 4
    int64_t ld;    // multiple of 4
};

template <int LPR>
__device__ __forceinline__ void load_row(float4 (&v)[NCH], const float* __restrict__ p, int cols, int lr) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (lr + k * LPR);
        v[k] = c < cols ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int LPR>
__device__ __forceinline__ void store_row(const float4 (&v)[NCH], float* __restrict__ p, int cols, int lr) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (lr + k * LPR);
        if (c < cols) *reinterpret_cast<float4*>(p + c) = v[k];
    }
}
// the same row as bf16 (round to nearest even), 8 bytes per chunk -- operands of the bf16-resident GEMMs (gemm_bf16res.hip)
__device__ __forceinline__ unsigned short f2bf_bits(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const unsigned short*>(&b);
}
template <int LPR>
__device__ __forceinline__ void load_row_bf16(float4 (&v)[NCH], const unsigned short* __restrict__ p, int cols, int lr) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (lr + k * LPR);
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cols) {
            const uint2 w = *reinterpret_cast<const uint2*>(p + c);
            v[k] = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                               __uint_as_float(w.y & 0xffff0000u));
        }
    }
}
template <int LPR>
__device__ __forceinline__ void store_row_bf16(const float4 (&v)[NCH], unsigned short* __restrict__ p, int cols, int lr) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (lr + k * LPR);
        if (c < cols) {
            ushort4 o;
            o.x = f2bf_bits(v[k].x); o.y = f2bf_bits(v[k].y); o.z = f2bf_bits(v[k].z); o.w = f2bf_bits(v[k].w);
            *reinterpret_cast<ushort4*>(p + c) = o;
        }
    }
}
// the same row as bf16 PLANES in the blocked layout of msn_plane_split (include/msn_hip.h; csrc/pgemm.hip): the operand of
// the fp32-grade plane GEMMs, written by the kernel that produces the row instead of a separate split pass.  A lane's chunk
// of 4 columns is 8 bytes of one block row per plane; the four rows a workgroup normalises together complete a 128-byte line.
struct PlaneOut {
    unsigned char* base;   // nullptr: no plane output
    int np, cb;            // planes (2 | 3), column blocks = 2 ceil(cols / 32)
    int cols;              // columns of the matrix: the padding behind them is written as +0
};
template <int NP>
__device__ __forceinline__ void split4(const float4& v, uint2 (&o)[NP]) {
    float r[4] = {v.x, v.y, v.z, v.w};
    unsigned short b[4];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            b[e] = f2bf_bits(r[e]);
            r[e] -= __uint_as_float((unsigned)b[e] << 16);       // exact
        }
        o[k].x = b[0] | ((unsigned)b[1] << 16);
        o[k].y = b[2] | ((unsigned)b[3] << 16);
    }
}
template <int LPR, int NP>
__device__ __forceinline__ void store_row_planes(const float4 (&v)[NCH], const PlaneOut& po, int64_t r, int lr) {
    unsigned char* rowp = po.base + (r >> 5) * (int64_t)po.cb * (NP * 1024) + (int)(r & 31) * 32;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = 4 * (lr + k * LPR);
        if (c < po.cb * 16) {                                     // columns past `cols` hold zeros (load_row): the padding
            uint2 o[NP];
            split4<NP>(c < po.cols ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f), o);
            unsigned char* dst = rowp + (c >> 4) * (NP * 1024) + (c & 15) * 2;
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dst + q * 1024) = o[q];
        }
    }
}
template <int LPR>
__device__ __forceinline__ void store_row_planes_any(const float4 (&v)[NCH], const PlaneOut& po, int64_t r, int lr) {
    if (po.np == 3) store_row_planes<LPR, 3>(v, po, r, lr);
    else store_row_planes<LPR, 2>(v, po, r, lr);
}
__device__ __forceinline__ float sum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ------------------------------------------------------------------------------------- LayerNorm
// one row in place: v -> (v - mean) rstd gamma + beta; the row's statistics in mu / rs (every lane of the row's group)
template <int LPR>
__device__ __forceinline__ void ln_fwd_row(float4 (&v)[NCH], const float4 (&gm)[NCH], const float4 (&bt)[NCH], int cols, int lr,
                                           float inv_n, float eps, float& mu, float& rs) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) s += sum4(v[k]);
    mu = group_sum<LPR>(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (4 * (lr + k * LPR) < cols) {
            const float a = v[k].x - mu, b = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
            q += a * a + b * b + c * c + d * d;
        }
    }
    rs = rsqrtf(group_sum<LPR>(q) * inv_n + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        v[k].x = (v[k].x - mu) * rs * gm[k].x + bt[k].x;
        v[k].y = (v[k].y - mu) * rs * gm[k].y + bt[k].y;
        v[k].z = (v[k].z - mu) * rs * gm[k].z + bt[k].z;
        v[k].w = (v[k].w - mu) * rs * gm[k].w + bt[k].w;
    }
}
template <int LPR>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t ldx, RowGeom g,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, float* __restrict__ y, int64_t ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     unsigned short* __restrict__ yb, PlaneOut po) {
    constexpr int RG = 256 / LPR;
    const int lr = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    float4 gm[NCH], bt[NCH];
    load_row<LPR>(gm, gamma, g.cols, lr);
    load_row<LPR>(bt, beta, g.cols, lr);
    const float inv_n = 1.f / (float)g.cols;
    for (int64_t r = (int64_t)blockIdx.x * RG + rg; r < g.rows; r += (int64_t)gridDim.x * RG) {
        float4 v[NCH];
        load_row<LPR>(v, x + r * ldx, g.cols, lr);
        float mu, rs;
        ln_fwd_row<LPR>(v, gm, bt, g.cols, lr, inv_n, eps, mu, rs);
        if (y) store_row<LPR>(v, y + r * ldy, g.cols, lr);
        if (yb) store_row_bf16<LPR>(v, yb + r * ldy, g.cols, lr);
        if (po.base) store_row_planes_any<LPR>(v, po, r, lr);
        if (lr == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

// LayerNorm forward whose ONLY output is the plane matrix (the operand of the next product).  The row-at-a-time kernel above
// writes a lane's 4 columns as 8 bytes per plane -- 32-byte segments, 3.4 TB/s.  Here a workgroup normalises one whole ROW
// BLOCK (32 rows: 8 waves x 4 rows), builds the block's plane images in LDS in their final layout, and then copies the block
// -- cb x NP x 1 KB of CONTIGUOUS memory -- out in 16-byte lanes.  Same arithmetic as ln_fwd_kernel<64> (ln_fwd_row).
template <int NP>
__global__ __launch_bounds__(512) void ln_fwd_planes_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                            unsigned char* __restrict__ out, int cb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ln_img[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gm[NCH], bt[NCH];
    load_row<64>(gm, gamma, cols, lane);
    load_row<64>(bt, beta, cols, lane);
    const float inv_n = 1.f / (float)cols;
    const int64_t r0 = (int64_t)blockIdx.x * 32 + wave * 4;
    float4 v[4][NCH];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                             // the wave's four rows in flight together
        const int64_t r = r0 + i;
        load_row<64>(v[i], x + (r < rows ? r : 0) * ldx, cols, lane);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + i;
        float mu, rs;
        ln_fwd_row<64>(v[i], gm, bt, cols, lane, inv_n, eps, mu, rs);
        const bool live = r < rows;
        // In LDS the rows of column block j are rotated by j & 7 positions: the 16 lanes of one ds_write_b64 group hold 4
        // neighbouring column blocks, whose images are 3 KB = 0 banks apart -- unrotated, a 4-way conflict
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = 4 * (lane + k * 64);
            if (c < cb * 16) {                                // columns past `cols` and rows past `rows`: the padding, +0
                uint2 o[NP];
                split4<NP>((live && c < cols) ? v[i][k] : make_float4(0.f, 0.f, 0.f, 0.f), o);
                const int j = c >> 4;
                unsigned char* dst = ln_img + j * (NP * 1024) + (((wave * 4 + i + (j & 7)) & 31) * 32) + (c & 15) * 2;
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dst + q * 1024) = o[q];
            }
        }
        if (lane == 0 && live) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
    __syncthreads();
    const int total = cb * NP * 1024;
    unsigned char* dst = out + (int64_t)blockIdx.x * total;
    for (int off = threadIdx.x * 16; off < total; off += 512 * 16) {
        const int image = off >> 10, j = image / NP;
        const int src = (image << 10) + (((off & 1023) + 32 * (j & 7)) & 1023);
        *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(ln_img + src);
    }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  block partials of dgamma / dbeta
template <int LPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                     const float* __restrict__ x, int64_t ldx, RowGeom g,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, float* __restrict__ dx,
                                                     int64_t lddx, float* __restrict__ part,
                                                     const float* __restrict__ add, int64_t ldadd,
                                                     unsigned short* __restrict__ dxb, int want_dxsum, PlaneOut po) {
    constexpr int RG = 256 / LPR;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [RG][2 or 3][cols]
    const int lr = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const bool dy_bf16 = (want_dxsum & 2) != 0;   // dy holds bf16 (lddy in bf16 elements): the input-gradient product wrote it so (cfg5)
    want_dxsum &= 1;
    const int nacc = want_dxsum ? 3 : 2;     // third accumulator: column sums of dx (the bias gradient of the Linear
                                             // that produced this LayerNorm's input through a residual add)
    float4 gm[NCH], dg[NCH], db[NCH], sx[NCH];
    load_row<LPR>(gm, gamma, g.cols, lr);
#pragma unroll
    for (int k = 0; k < NCH; ++k) dg[k] = db[k] = sx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_n = 1.f / (float)g.cols;
    for (int64_t r = (int64_t)blockIdx.x * RG + rg; r < g.rows; r += (int64_t)gridDim.x * RG) {
        float4 v[NCH], d[NCH];
        load_row<LPR>(v, x + r * ldx, g.cols, lr);
        if (dy_bf16) load_row_bf16<LPR>(d, reinterpret_cast<const unsigned short*>(dy) + r * lddy, g.cols, lr);
        else load_row<LPR>(d, dy + r * lddy, g.cols, lr);
        const float mu = mean[r], rs = rstd[r];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const bool on = 4 * (lr + k * LPR) < g.cols;
            v[k].x = on ? (v[k].x - mu) * rs : 0.f;  // xhat
            v[k].y = on ? (v[k].y - mu) * rs : 0.f;
            v[k].z = on ? (v[k].z - mu) * rs : 0.f;
            v[k].w = on ? (v[k].w - mu) * rs : 0.f;
            dg[k].x += d[k].x * v[k].x; dg[k].y += d[k].y * v[k].y; dg[k].z += d[k].z * v[k].z; dg[k].w += d[k].w * v[k].w;
            db[k].x += d[k].x; db[k].y += d[k].y; db[k].z += d[k].z; db[k].w += d[k].w;
            d[k].x *= gm[k].x; d[k].y *= gm[k].y; d[k].z *= gm[k].z; d[k].w *= gm[k].w;  // g * dy
            s1 += sum4(d[k]);
            s2 += dot4(d[k], v[k]);
        }
        const float m1 = group_sum<LPR>(s1) * inv_n, m2 = group_sum<LPR>(s2) * inv_n;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            d[k].x = rs * (d[k].x - m1 - v[k].x * m2);
            d[k].y = rs * (d[k].y - m1 - v[k].y * m2);
            d[k].z = rs * (d[k].z - m1 - v[k].z * m2);
            d[k].w = rs * (d[k].w - m1 - v[k].w * m2);
        }
        if (add) {  // gradient arriving through the residual branch around this LayerNorm
            float4 a[NCH];
            load_row<LPR>(a, add + r * ldadd, g.cols, lr);
#pragma unroll
            for (int k = 0; k < NCH; ++k) { d[k].x += a[k].x; d[k].y += a[k].y; d[k].z += a[k].z; d[k].w += a[k].w; }
        }
        store_row<LPR>(d, dx + r * lddx, g.cols, lr);
        if (dxb) store_row_bf16<LPR>(d, dxb + r * lddx, g.cols, lr);
        if (po.base) store_row_planes_any<LPR>(d, po, r, lr);
        if (want_dxsum) {
#pragma unroll
            for (int k = 0; k < NCH; ++k) { sx[k].x += d[k].x; sx[k].y += d[k].y; sx[k].z += d[k].z; sx[k].w += d[k].w; }
        }
    }
    // reduce the RG row-groups of this block, then publish [nacc][cols] for the final pass
    float* mine = lds + (int64_t)rg * nacc * g.cols;
    store_row<LPR>(dg, mine, g.cols, lr);
    store_row<LPR>(db, mine + g.cols, g.cols, lr);
    if (want_dxsum) store_row<LPR>(sx, mine + 2 * g.cols, g.cols, lr);
    __syncthreads();
    for (int i = threadIdx.x; i < nacc * g.cols; i += 256) {
        float s = 0.f;
        for (int k = 0; k < RG; ++k) s += lds[(int64_t)k * nacc * g.cols + i];
        part[(int64_t)blockIdx.x * nacc * g.cols + i] = s;
    }
}

// The backward whose dx ALSO leaves as a plane matrix, by row blocks (the counterpart of ln_fwd_planes_kernel): ln_bwd_kernel
// writes a lane's 4 columns of a plane as 8 bytes -- 32-byte segments of the blocked layout, 72 of them per 384-wide row.  Here a
// workgroup takes whole 32-row blocks (8 waves x 4 rows, one row in flight per wave: 16 waves per CU keep ~70 KB of loads in
// the air), stores dx row by row as before, builds the block's plane images in LDS and copies them out as contiguous memory.
// Same arithmetic per row as ln_bwd_kernel<64>; the column partials are kept per lane over the workgroup's blocks and published
// in the same [workgroup][nacc][cols] layout for sum_slabs_kernel.  NC = 16-byte chunks per lane (cols <= 256 NC).
template <int NP, int NC>
__global__ __launch_bounds__(512) void ln_bwd_planes_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                            int64_t ldx, int64_t rows, int cols, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, int64_t lddx, float* __restrict__ part,
                                                            const float* __restrict__ add, int64_t ldadd, int want_dxsum,
                                                            unsigned char* __restrict__ out, int cb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ln_img[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nacc = want_dxsum ? 3 : 2;
    float4 gm[NC], dg[NC], db[NC], sx[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = 4 * (lane + k * 64);
        gm[k] = c < cols ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[k] = db[k] = sx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_n = 1.f / (float)cols;
    const int64_t nrb = (rows + 31) >> 5;
    const int total = cb * NP * 1024;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        for (int i = 0; i < 4; ++i) {                         // (two rows in flight per wave, `#pragma unroll 2`: 127 registers, same time)
            const int64_t r = rb * 32 + wave * 4 + i;
            const bool live = r < rows;                       // wave-uniform
            float4 d[NC];
            if (live) {
                float4 v[NC];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int c = 4 * (lane + k * 64);
                    const bool on = c < cols;
                    v[k] = on ? *reinterpret_cast<const float4*>(x + r * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    d[k] = on ? *reinterpret_cast<const float4*>(dy + r * lddy + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                const float mu = mean[r], rs = rstd[r];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const bool on = 4 * (lane + k * 64) < cols;
                    v[k].x = on ? (v[k].x - mu) * rs : 0.f;  // xhat
                    v[k].y = on ? (v[k].y - mu) * rs : 0.f;
                    v[k].z = on ? (v[k].z - mu) * rs : 0.f;
                    v[k].w = on ? (v[k].w - mu) * rs : 0.f;
                    dg[k].x += d[k].x * v[k].x; dg[k].y += d[k].y * v[k].y; dg[k].z += d[k].z * v[k].z; dg[k].w += d[k].w * v[k].w;
                    db[k].x += d[k].x; db[k].y += d[k].y; db[k].z += d[k].z; db[k].w += d[k].w;
                    d[k].x *= gm[k].x; d[k].y *= gm[k].y; d[k].z *= gm[k].z; d[k].w *= gm[k].w;  // g * dy
                    s1 += sum4(d[k]);
                    s2 += dot4(d[k], v[k]);
                }
                const float m1 = group_sum<64>(s1) * inv_n, m2 = group_sum<64>(s2) * inv_n;
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    d[k].x = rs * (d[k].x - m1 - v[k].x * m2);
                    d[k].y = rs * (d[k].y - m1 - v[k].y * m2);
                    d[k].z = rs * (d[k].z - m1 - v[k].z * m2);
                    d[k].w = rs * (d[k].w - m1 - v[k].w * m2);
                }
                if (add) {  // gradient arriving through the residual branch around this LayerNorm
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        const int c = 4 * (lane + k * 64);
                        if (c < cols) {
                            const float4 a = *reinterpret_cast<const float4*>(add + r * ldadd + c);
                            d[k].x += a.x; d[k].y += a.y; d[k].z += a.z; d[k].w += a.w;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    const int c = 4 * (lane + k * 64);
                    if (c < cols) *reinterpret_cast<float4*>(dx + r * lddx + c) = d[k];
                }
                if (want_dxsum) {
#pragma unroll
                    for (int k = 0; k < NC; ++k) { sx[k].x += d[k].x; sx[k].y += d[k].y; sx[k].z += d[k].z; sx[k].w += d[k].w; }
                }
            }
            // the row's plane images (rows of column block j rotated by j & 7 positions: see ln_fwd_planes_kernel)
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const int c = 4 * (lane + k * 64);
                if (c < cb * 16) {                            // columns past `cols` and rows past `rows`: the padding, +0
                    uint2 o[NP];
                    split4<NP>((live && c < cols) ? d[k] : make_float4(0.f, 0.f, 0.f, 0.f), o);
                    const int j = c >> 4;
                    unsigned char* dst = ln_img + j * (NP * 1024) + (((wave * 4 + i + (j & 7)) & 31) * 32) + (c & 15) * 2;
#pragma unroll
                    for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dst + q * 1024) = o[q];
                }
            }
        }
        __syncthreads();
        unsigned char* dst = out + rb * total;
        for (int off = threadIdx.x * 16; off < total; off += 512 * 16) {
            const int image = off >> 10, j = image / NP;
            const int src = (image << 10) + (((off & 1023) + 32 * (j & 7)) & 1023);
            *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(ln_img + src);
        }
        __syncthreads();
    }
    // the eight waves' column partials -> one [nacc][cols] slab of this workgroup (the image area is free now: 96 cols bytes)
    float* red = reinterpret_cast<float*>(ln_img) + (int64_t)wave * nacc * cols;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        const int c = 4 * (lane + k * 64);
        if (c < cols) {
            *reinterpret_cast<float4*>(red + c) = dg[k];
            *reinterpret_cast<float4*>(red + cols + c) = db[k];
            if (want_dxsum) *reinterpret_cast<float4*>(red + 2 * cols + c) = sx[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nacc * cols; i += 512) {
        float s = 0.f;
        for (int w = 0; w < 8; ++w) s += reinterpret_cast<const float*>(ln_img)[(int64_t)w * nacc * cols + i];
        part[(int64_t)blockIdx.x * nacc * cols + i] = s;
    }
}

// out[i] = sum_b part[b][i], i < n : 16 slab groups x 64 columns per block (each thread keeps 4 independent loads
// in flight), LDS-combined in a fixed order
__global__ __launch_bounds__(1024) void sum_slabs_kernel(const float* __restrict__ part, int nslabs, int n,
                                                         float* __restrict__ out0, float* __restrict__ out1, int split,
                                                         float* __restrict__ out2 = nullptr) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x % 64, g = threadIdx.x / 64;
    const int i = blockIdx.x * 64 + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int b = g;
        for (; b + 48 < nslabs; b += 64) {
            s0 += part[(int64_t)b * n + i];
            s1 += part[(int64_t)(b + 16) * n + i];
            s2 += part[(int64_t)(b + 32) * n + i];
            s3 += part[(int64_t)(b + 48) * n + i];
        }
        for (; b < nslabs; b += 16) s0 += part[(int64_t)b * n + i];
    }
    red[g][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        if (i < split) out0[i] = t;
        else if (i < 2 * split || !out2) out1[i - split] = t;
        else out2[i - 2 * split] = t;
    }
}

// ------------------------------------------------------------------------------------ L2 normalise
template <int LPR>
__global__ __launch_bounds__(256) void l2n_fwd_kernel(const float* __restrict__ x, int64_t ldx, RowGeom g,
                                                      float* __restrict__ y, int64_t ldy, float* __restrict__ inv) {
    constexpr int RG = 256 / LPR;
    const int lr = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    for (int64_t r = (int64_t)blockIdx.x * RG + rg; r < g.rows; r += (int64_t)gridDim.x * RG) {
        float4 v[NCH];
        load_row<LPR>(v, x + r * ldx, g.cols, lr);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) q += dot4(v[k], v[k]);
        const float nrm = sqrtf(group_sum<LPR>(q));
        const float iv = 1.f / nrm;  // no epsilon: ref models_multimodal.py:279
#pragma unroll
        for (int k = 0; k < NCH; ++k) { v[k].x /= nrm; v[k].y /= nrm; v[k].z /= nrm; v[k].w /= nrm; }
        store_row<LPR>(v, y + r * ldy, g.cols, lr);
        if (lr == 0) inv[r] = iv;
    }
}
// dx = inv * (dy - y * <y, dy>)
template <int LPR>
__global__ __launch_bounds__(256) void l2n_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                      const float* __restrict__ y, int64_t ldy, RowGeom g,
                                                      const float* __restrict__ inv, float* __restrict__ dx,
                                                      int64_t lddx) {
    constexpr int RG = 256 / LPR;
    const int lr = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    for (int64_t r = (int64_t)blockIdx.x * RG + rg; r < g.rows; r += (int64_t)gridDim.x * RG) {
        float4 v[NCH], d[NCH];
        load_row<LPR>(v, y + r * ldy, g.cols, lr);
        load_row<LPR>(d, dy + r * lddy, g.cols, lr);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) s += dot4(v[k], d[k]);
        s = group_sum<LPR>(s);
        const float iv = inv[r];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            d[k].x = iv * (d[k].x - v[k].x * s);
            d[k].y = iv * (d[k].y - v[k].y * s);
            d[k].z = iv * (d[k].z - v[k].z * s);
            d[k].w = iv * (d[k].w - v[k].w * s);
        }
        store_row<LPR>(d, dx + r * lddx, g.cols, lr);
    }
}

// -------------------------------------------------------------------------- time / band embedding
// out[b,t,c] = x[b,t] * w[c] + bw[c] + (c even ? sin : cos)(t[b,t] * omega[c/2]) + band[band(t)][c]
// ref transformer_utils.py:166-176 (interleaved sin/cos) and :214-231.
__global__ void time_embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t rows, int T,
                                      int e, const float* __restrict__ w, const float* __restrict__ bw,
                                      const float* __restrict__ omega, const float* __restrict__ band, int nband,
                                      float* __restrict__ out) {
    const int64_t total = rows * e;
    const int per_band = nband > 1 ? T / nband : T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / e;
        const int c = (int)(i % e);
        const float ang = t[r] * omega[c >> 1];
        float v = x[r] * w[c] + bw[c];
        v += (c & 1) ? cosf(ang) : sinf(ang);
        if (nband > 1) v += band[(int64_t)(((int)(r % T)) / per_band) * e + c];
        out[i] = v;
    }
}

// One block per sample b: seg[b][k][c] = sum_{t in band k} dy[b,t,c];  sdx[b][c] = sum_t dy[b,t,c] * x[b,t]
__global__ __launch_bounds__(256) void time_embed_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             int T, int e, int nband, float* __restrict__ seg,
                                                             float* __restrict__ sdx) {
    extern __shared__ float lds[];  // [groups][e]
    const int b = blockIdx.x;
    const int cpt = e < 256 ? e : 256;           // threads along columns
    const int groups = 256 / cpt;                // row groups
    const int c0 = threadIdx.x % cpt, gidx = threadIdx.x / cpt;
    const int per_band = T / nband;
    const float* dyb = dy + (int64_t)b * T * e;
    const float* xb = x + (int64_t)b * T;
    for (int cb = 0; cb < e; cb += cpt) {
        const int c = cb + c0;
        float sx = 0.f;
        for (int k = 0; k < nband; ++k) {
            float s = 0.f;
            if (c < e && gidx < groups)
                for (int tt = k * per_band + gidx; tt < (k + 1) * per_band; tt += groups) {
                    const float d = dyb[(int64_t)tt * e + c];
                    s += d;
                    sx = fmaf(d, xb[tt], sx);
                }
            __syncthreads();
            if (gidx < groups) lds[gidx * cpt + c0] = s;
            __syncthreads();
            if (gidx == 0 && c < e) {
                float tot = 0.f;
                for (int q = 0; q < groups; ++q) tot += lds[q * cpt + c0];
                seg[((int64_t)b * nband + k) * e + c] = tot;
            }
        }
        __syncthreads();
        if (gidx < groups) lds[gidx * cpt + c0] = sx;
        __syncthreads();
        if (gidx == 0 && c < e) {
            float tot = 0.f;
            for (int q = 0; q < groups; ++q) tot += lds[q * cpt + c0];
            sdx[(int64_t)b * e + c] = tot;
        }
    }
}
// dw[c] = sum_b sdx[b][c]; dband[k][c] = sum_b seg[b][k][c]; dbw[c] = sum_k dband[k][c].
// One block per 64 columns; its 16 thread groups each sum a sixteenth of the samples (fixed order, 4 independent
// partial sums in flight per thread), LDS-combined.
__global__ __launch_bounds__(1024) void time_embed_bwd_finish_kernel(const float* __restrict__ seg,
                                                                     const float* __restrict__ sdx, int B, int e,
                                                                     int nband, float* __restrict__ dw,
                                                                     float* __restrict__ dbw, float* __restrict__ dband) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    auto column_sum = [&](const float* __restrict__ src, int64_t stride) {   // sum over b of src[b * stride + c]
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (c < e) {
            int b = g;
            for (; b + 48 < B; b += 64) {
                s0 += src[(int64_t)b * stride + c];
                s1 += src[(int64_t)(b + 16) * stride + c];
                s2 += src[(int64_t)(b + 32) * stride + c];
                s3 += src[(int64_t)(b + 48) * stride + c];
            }
            for (; b < B; b += 16) s0 += src[(int64_t)b * stride + c];
        }
        __syncthreads();
        red[g][cl] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        return t;
    };
    const float sw = column_sum(sdx, e);
    if (g == 0 && c < e) dw[c] = sw;
    float sb = 0.f;
    for (int k = 0; k < nband; ++k) {
        const float t = column_sum(seg + (int64_t)k * e, (int64_t)nband * e);
        if (g == 0 && c < e && dband) dband[(int64_t)k * e + c] = t;
        sb += t;
    }
    if (g == 0 && c < e) dbw[c] = sb;
}

// ---------------------------------------------------------------------------------- masked pooling
// ref transformer_utils.py:234-239: x = x * mask; mean: sum_t / sum_t mask (0/0 -> NaN for an empty
// row, as the reference); max: max over ALL t of the zeroed tokens (padded zeros take part).
constexpr int POOL_MEAN = 0, POOL_MAX = 1;
// One workgroup per sample: 4 thread groups x 64 channels, group q owns the q-th quarter of the sequence (contiguous,
// so the first maximum wins ties exactly as in a sequential scan) and keeps 4 loads in flight; the groups are combined
// through LDS in order.  (One thread per channel walking all T tokens ran at 0.8 TB/s.)
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, int T,
                                                       int e, int mode, float* __restrict__ out, int* __restrict__ arg,
                                                       float* __restrict__ count) {
    __shared__ float red_v[4][64];
    __shared__ float red_c[4][64];
    __shared__ int red_i[4][64];
    const int b = blockIdx.x;
    const float* xb = x + (int64_t)b * T * e;
    const uint8_t* mb = mask + (int64_t)b * T;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int per = (T + 3) / 4, t0 = q * per, t1 = min(T, t0 + per);
    for (int c0 = 0; c0 < e; c0 += 64) {
        const int c = c0 + cl;
        float s = 0.f, cnt = 0.f, best = -INFINITY;
        int bi = t0;
        if (c < e) {
            if (mode == POOL_MEAN) {
                float s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int t = t0;
                for (; t + 3 < t1; t += 4) {
                    const float m0 = mb[t] ? 1.f : 0.f, m1 = mb[t + 1] ? 1.f : 0.f, m2 = mb[t + 2] ? 1.f : 0.f,
                                m3 = mb[t + 3] ? 1.f : 0.f;
                    s += xb[(int64_t)t * e + c] * m0;
                    s1 += xb[(int64_t)(t + 1) * e + c] * m1;
                    s2 += xb[(int64_t)(t + 2) * e + c] * m2;
                    s3 += xb[(int64_t)(t + 3) * e + c] * m3;
                    cnt += (m0 + m1) + (m2 + m3);
                }
                for (; t < t1; ++t) {
                    const float m = mb[t] ? 1.f : 0.f;
                    s += xb[(int64_t)t * e + c] * m;
                    cnt += m;
                }
                s = (s + s1) + (s2 + s3);
            } else {
                for (int t = t0; t < t1; ++t) {
                    const float v = xb[(int64_t)t * e + c] * (mb[t] ? 1.f : 0.f);
                    if (v > best) { best = v; bi = t; }
                }
            }
        }
        __syncthreads();
        red_v[q][cl] = mode == POOL_MEAN ? s : best;
        red_c[q][cl] = cnt;
        red_i[q][cl] = bi;
        __syncthreads();
        if (q == 0 && c < e) {
            if (mode == POOL_MEAN) {
                const float tot = (red_v[0][cl] + red_v[1][cl]) + (red_v[2][cl] + red_v[3][cl]);
                const float n = (red_c[0][cl] + red_c[1][cl]) + (red_c[2][cl] + red_c[3][cl]);
                out[(int64_t)b * e + c] = tot / n;
                if (c == 0) count[b] = n;
            } else {
                float bb = red_v[0][cl];
                int ii = red_i[0][cl];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (red_v[k][cl] > bb) { bb = red_v[k][cl]; ii = red_i[k][cl]; }
                out[(int64_t)b * e + c] = bb;
                arg[(int64_t)b * e + c] = ii;
            }
        }
    }
}
__global__ void pool_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ mask, int64_t B, int T,
                                int e, int mode, const int* __restrict__ arg, const float* __restrict__ count,
                                float* __restrict__ dx) {
    const int64_t total = B * T * e;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % e);
        const int64_t bt = i / e;
        const int t = (int)(bt % T);
        const int64_t b = bt / T;
        const float m = mask[bt] ? 1.f : 0.f;
        float v;
        if (mode == POOL_MEAN) {
            v = dout[b * e + c] * m / count[b];
        } else {
            v = (arg[b * e + c] == t) ? dout[b * e + c] * m : 0.f;
        }
        dx[i] = v;
    }
}
// y[b,t,:] = x[b,t,:] * mask[b,t]   (agg = "pretraining" tokens, and the zeroing before attn pooling)
__global__ void mask_tokens_kernel(const float* __restrict__ x, const uint8_t* __restrict__ mask, int64_t rows, int e,
                                   float* __restrict__ y) {
    const int64_t total = rows * e;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = mask[i / e] ? x[i] : 0.f * x[i];
}

// ---------------------------------------------------------------- ViT token assembly (build-defined)
// tok[b][0] = cls + pos[0];  tok[b][1+i] = patch[b][i] + pos[1+i]      (T = 1 + n_patches)
__global__ void vit_tokens_fwd_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                      const float* __restrict__ pos, int64_t B, int T, int e, float* __restrict__ tok) {
    const int64_t total = B * T * e;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % e);
        const int t = (int)((i / e) % T);
        const int64_t b = i / ((int64_t)e * T);
        const float base = t == 0 ? cls[c] : patch[(b * (T - 1) + (t - 1)) * e + c];
        tok[i] = base + pos[(int64_t)t * e + c];
    }
}
// dpatch[b][i] = dtok[b][1+i]  (compaction for the patch-embedding wgrad GEMM)
__global__ void vit_tokens_bwd_kernel(const float* __restrict__ dtok, int64_t B, int T, int e, float* __restrict__ dpatch) {
    const int64_t total = B * (T - 1) * e;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % e);
        const int t = (int)((i / e) % (T - 1));
        const int64_t b = i / ((int64_t)e * (T - 1));
        dpatch[i] = dtok[(b * T + t + 1) * e + c];
    }
}

// feat[r] = (x*m, t*inv_norm*m, m, 0): the 4 input channels of the build-defined 1-D CNN encoder
__global__ void series_features_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                       const uint8_t* __restrict__ mask, int64_t rows, float inv_norm,
                                       float4* __restrict__ feat) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[r] ? 1.f : 0.f;
        feat[r] = make_float4(x[r] * m, t[r] * inv_norm * m, m, 0.f);
    }
}

// ------------------------------------------------------------- masked MSE (masked-light-curve pretraining)
// loss = mean over {i : sel[i]} of (pred[i] - target[i])^2   (ref src/models_pretraining.py:201-231:
// nn.MSELoss()(x[mask_pred], x_pred[mask_pred])); stats[0] = loss, stats[1] = number of selected elements.
__global__ __launch_bounds__(1024) void masked_mse_fwd_kernel(const float* __restrict__ pred,
                                                              const float* __restrict__ target,
                                                              const uint8_t* __restrict__ sel, int64_t n,
                                                              float* __restrict__ stats) {
    __shared__ float red[16];
    float s = 0.f, c = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
        if (sel[i]) {
            const float d = pred[i] - target[i];
            s = fmaf(d, d, s);
            c += 1.f;
        }
    const float ts = block_sum(s, red);
    const float tc = block_sum(c, red);
    if (threadIdx.x == 0) {
        stats[0] = ts / tc;
        stats[1] = tc;
    }
}
__global__ void masked_mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                      const uint8_t* __restrict__ sel, int64_t n, const float* __restrict__ stats,
                                      const float* __restrict__ grad_out, float* __restrict__ dpred) {
    const float f = 2.f * (*grad_out) / stats[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dpred[i] = sel[i] ? f * (pred[i] - target[i]) : 0.f;
}

// ----------------------------------------------------------- on-device augmentation (NoisyDataLoader.__iter__)
// ref src/dataloader.py:88-287: images  <- rot90^k( img + (2u - 1) * level * std(batch) ), k per sample;
// series <- x + g * err * level.  The random fields u ~ U[0,1), g ~ N(0,1) and k are INPUTS (any RNG).
// stats[0] = sum, stats[1] = sum of squared deviations from the mean (two passes -> unbiased std like torch.std)
__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ part) {
    __shared__ double red[16];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        part[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void sqdev_kernel(const float* __restrict__ x, int64_t n, const double* __restrict__ part,
                                                     int nparts, double* __restrict__ part2) {
    __shared__ double red[16];
    double tot = 0.0;
    for (int k = 0; k < nparts; ++k) tot += part[k];
    const double mean = tot / (double)n;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)x[i] - mean;
        s += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        part2[blockIdx.x] = t;
    }
}
__global__ void augment_images_kernel(const float* __restrict__ img, const float* __restrict__ u,
                                      const int* __restrict__ rot, int64_t B, int C, int S, float level,
                                      const double* __restrict__ part2, int nparts, int64_t n, float* __restrict__ out) {
    double ss = 0.0;
    for (int k = 0; k < nparts; ++k) ss += part2[k];
    const float range = level * (float)sqrt(ss / (double)(n - 1));
    const int64_t total = B * C * S * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % S), y = (int)((i / S) % S);
        const int64_t bc = i / ((int64_t)S * S);
        const int k = rot[bc / C] & 3;
        // torch.rot90 (counter-clockwise, k times): k=1: out[y][x] = in[x][S-1-y]
        int sy = y, sx = x;
        if (k == 1) { sy = x; sx = S - 1 - y; }
        else if (k == 2) { sy = S - 1 - y; sx = S - 1 - x; }
        else if (k == 3) { sy = S - 1 - x; sx = y; }
        const int64_t src = (bc * S + sy) * S + sx;
        out[i] = img[src] + (2.f * u[src] - 1.f) * range;
    }
}
__global__ void augment_series_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                      const float* __restrict__ err, int64_t n, float level, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] + g[i] * err[i] * level;
}

// -------------------------------------------------------------------------------------------- dropout
// Counter-based: element i is kept iff hash(seed, i) >= p, so the SAME call on the gradient reproduces the mask
// (nothing is stored).  y = keep ? x / (1 - p) : 0  (+ res).  nn.Dropout semantics in train mode; the stream of
// random numbers differs from torch's Philox sequence (only the distribution can match).
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t i, float p, float inv_keep) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    const float u = (float)(x >> 40) * (1.f / 16777216.f);   // 24 random bits -> [0, 1)
    return u >= p ? inv_keep : 0.f;
}
// seed_base != NULL: the seed is seed_base[0] + seed (device-resident base: a launch recorded in a HIP graph draws a new
// mask at every replay, msn_seed_advance moves the base once per step)
__global__ void dropout_kernel(const float* __restrict__ x, int64_t n, float p, uint64_t seed,
                               const uint64_t* __restrict__ seed_base, const float* __restrict__ res,
                               float* __restrict__ y) {
    if (seed_base) seed += seed_base[0];
    const float inv_keep = 1.f / (1.f - p);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i] * keep_scale(seed, i, p, inv_keep);
        if (res) v += res[i];
        y[i] = v;
    }
}

static int pick_lpr(int cols) {
    const int chunks = cols / 4;
    if (chunks <= 4 * NCH) return 4;
    if (chunks <= 16 * NCH) return 16;
    return 64;
}
static int check_rows(const char* who, int64_t rows, int cols, std::initializer_list<int64_t> lds,
                      std::initializer_list<const void*> ptrs) {
    MSN_REQUIRE(rows > 0 && cols > 0, "%s: empty input", who);
    MSN_REQUIRE(cols % 4 == 0 && cols <= 64 * NCH * 4, "%s: row length %d must be a multiple of 4 and <= %d", who, cols,
                64 * NCH * 4);
    for (int64_t ld : lds) MSN_REQUIRE(ld >= cols && ld % 4 == 0, "%s: leading dimension %lld", who, (long long)ld);
    for (const void* p : ptrs)
        MSN_REQUIRE(p && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "%s: null or unaligned pointer", who);
    return MSN_OK;
}

#define MSN_LPR_DISPATCH(KERNEL, lpr, grid, lds, st, ...)                                         \
    if (lpr == 4) hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), lds, st, __VA_ARGS__);         \
    else if (lpr == 16) hipLaunchKernelGGL((KERNEL<16>), grid, dim3(256), lds, st, __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), lds, st, __VA_ARGS__);

static int ln_grid(int64_t rows, int lpr) {
    return (int)std::min<int64_t>(cdiv(rows, 256 / lpr), 2048);
}
static int ln_bwd_grid(int64_t rows, int lpr) {
    // 2 workgroups per CU (8 waves): measured optimum with the 16-group slab sum behind it (tools/bench_rowops.py)
    return (int)std::min<int64_t>(cdiv(rows, 4 * (256 / lpr)), 512);
}

}  // namespace msn

using namespace msn;

extern "C" int msn_layernorm_fwd(const float* x, int64_t ldx, int64_t rows, int cols, const float* gamma,
                                 const float* beta, float eps, float* y, int64_t ldy, float* mean, float* rstd,
                                 msn_stream_t stream) {
    if (int rc = check_rows("msn_layernorm_fwd", rows, cols, {ldx, ldy}, {x, y, gamma, beta})) return rc;
    MSN_REQUIRE(mean && rstd, "msn_layernorm_fwd: null statistics pointer");
    const int lpr = pick_lpr(cols);
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MSN_LPR_DISPATCH(ln_fwd_kernel, lpr, dim3(ln_grid(rows, lpr)), 0, st, x, ldx, g, gamma, beta, eps, y, ldy, mean, rstd,
                     (unsigned short*)nullptr, PlaneOut{nullptr, 0, 0, 0})
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_layernorm_fwd_bf16(const float* x, int64_t ldx, int64_t rows, int cols, const float* gamma,
                                      const float* beta, float eps, void* y_bf16, int64_t ldy, float* mean, float* rstd,
                                      msn_stream_t stream) {
    if (int rc = check_rows("msn_layernorm_fwd_bf16", rows, cols, {ldx, ldy}, {x, gamma, beta})) return rc;
    MSN_REQUIRE(mean && rstd && y_bf16 && (reinterpret_cast<uintptr_t>(y_bf16) & 7) == 0, "msn_layernorm_fwd_bf16: bad pointer");
    const int lpr = pick_lpr(cols);
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MSN_LPR_DISPATCH(ln_fwd_kernel, lpr, dim3(ln_grid(rows, lpr)), 0, st, x, ldx, g, gamma, beta, eps, (float*)nullptr, ldy,
                     mean, rstd, static_cast<unsigned short*>(y_bf16), PlaneOut{nullptr, 0, 0, 0})
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" size_t msn_layernorm_bwd_workspace_bytes(int64_t rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return sizeof(float) * 2 * (size_t)cols * (size_t)ln_bwd_grid(rows, pick_lpr(cols));
}

extern "C" int msn_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int cols,
                                 const float* mean, const float* rstd, const float* gamma, const float* add,
                                 int64_t ldadd, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* ws,
                                 size_t ws_bytes, msn_stream_t stream) {
    if (int rc = check_rows("msn_layernorm_bwd", rows, cols, {lddy, ldx, lddx}, {dy, x, dx, gamma})) return rc;
    MSN_REQUIRE(mean && rstd && dgamma && dbeta, "msn_layernorm_bwd: null pointer");
    MSN_REQUIRE(!add || (ldadd >= cols && ldadd % 4 == 0 && (reinterpret_cast<uintptr_t>(add) & 15) == 0),
                "msn_layernorm_bwd: bad residual-gradient operand");
    const int lpr = pick_lpr(cols);
    const int grid = ln_bwd_grid(rows, lpr);
    MSN_REQUIRE(ws && ws_bytes >= sizeof(float) * 2 * (size_t)cols * grid, "msn_layernorm_bwd: workspace too small");
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const size_t lds = sizeof(float) * 2 * (size_t)cols * (256 / lpr);
    MSN_LPR_DISPATCH(ln_bwd_kernel, lpr, dim3(grid), lds, st, dy, lddy, x, ldx, g, mean, rstd, gamma, dx, lddx, part, add, ldadd,
                     (unsigned short*)nullptr, 0, PlaneOut{nullptr, 0, 0, 0})
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv(2 * cols, 64)), dim3(1024), 0, st, part, grid, 2 * cols,
                       dgamma, dbeta, cols);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// the same backward that ALSO writes a bf16 copy of dx (row stride lddx): the gradient is consumed twice, as the fp32
// residual-stream gradient and as the bf16 operand of the next weight / input gradient product
extern "C" int msn_layernorm_bwd_bf16(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int cols,
                                      const float* mean, const float* rstd, const float* gamma, const float* add,
                                      int64_t ldadd, float* dx, int64_t lddx, void* dx_bf16, float* dgamma, float* dbeta,
                                      float* dx_colsum, int dy_is_bf16, void* ws, size_t ws_bytes, msn_stream_t stream) {
    if (int rc = check_rows("msn_layernorm_bwd_bf16", rows, cols, {lddy, ldx, lddx}, {dy, x, dx, gamma})) return rc;
    MSN_REQUIRE(mean && rstd && dgamma && dbeta && dx_bf16 && (reinterpret_cast<uintptr_t>(dx_bf16) & 7) == 0,
                "msn_layernorm_bwd_bf16: null pointer");
    MSN_REQUIRE(!add || (ldadd >= cols && ldadd % 4 == 0 && (reinterpret_cast<uintptr_t>(add) & 15) == 0),
                "msn_layernorm_bwd_bf16: bad residual-gradient operand");
    const int lpr = pick_lpr(cols);
    const int grid = ln_bwd_grid(rows, lpr);
    const int nacc = dx_colsum ? 3 : 2;
    MSN_REQUIRE(ws && ws_bytes >= sizeof(float) * nacc * (size_t)cols * grid, "msn_layernorm_bwd_bf16: workspace too small");
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    const size_t lds = sizeof(float) * nacc * (size_t)cols * (256 / lpr);
    MSN_LPR_DISPATCH(ln_bwd_kernel, lpr, dim3(grid), lds, st, dy, lddy, x, ldx, g, mean, rstd, gamma, dx, lddx, part, add, ldadd,
                     static_cast<unsigned short*>(dx_bf16), (dx_colsum ? 1 : 0) | (dy_is_bf16 ? 2 : 0), PlaneOut{nullptr, 0, 0, 0})
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv(nacc * cols, 64)), dim3(1024), 0, st, part, grid, nacc * cols,
                       dgamma, dbeta, cols, dx_colsum);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// ---- LayerNorm writing bf16 PLANES (blocked layout of msn_plane_split): the operand of the plane GEMMs (pgemm.hip)
static int plane_tail_zero(void* planes_out, int64_t rows, int cols, int np, hipStream_t st) {
    // rows past `rows` in the last 32-row block are part of the plane matrix and must be zero
    if (rows % 32 == 0) return MSN_OK;
    const int64_t cb = 2 * cdiv(cols, 32);
    const size_t block_row = (size_t)cb * np * 1024;
    if (hipMemsetAsync(static_cast<unsigned char*>(planes_out) + (rows / 32) * block_row, 0, block_row, st) != hipSuccess) {
        set_error("plane output: hipMemsetAsync failed");
        return MSN_ERR_HIP;
    }
    return MSN_OK;
}

// Row-block kernels (ln_fwd_planes_kernel / ln_bwd_planes_kernel) from 32 768 rows on; below, the coarse grid of whole row blocks
// loses to the row-at-a-time kernels (25.7 vs 25.9 us at 8 320 rows: r05 log, item 14).  Both write the same bytes.
static constexpr int64_t kLnBlockRows = 32 * 1024;
extern "C" int msn_layernorm_fwd_planes(const float* x, int64_t ldx, int64_t rows, int cols, const float* gamma,
                                        const float* beta, float eps, int planes, void* y_planes, float* y, int64_t ldy,
                                        float* mean, float* rstd, msn_stream_t stream) {
    if (int rc = check_rows("msn_layernorm_fwd_planes", rows, cols, {ldx}, {x, gamma, beta, y_planes})) return rc;
    MSN_REQUIRE(mean && rstd && (planes == 2 || planes == 3), "msn_layernorm_fwd_planes: bad argument");
    MSN_REQUIRE(!y || (ldy >= cols && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0), "msn_layernorm_fwd_planes: bad fp32 output");
    const int lpr = pick_lpr(cols);
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cbn = 2 * (int)cdiv(cols, 32);
    const size_t img = (size_t)cbn * planes * 1024;
    // whole row blocks through LDS (two workgroups per CU); from 1024 row blocks on (4 per CU) -- below that the grid is too
    // coarse: 520 blocks (16 640 rows) 18.4 us against the row kernel's 15.1, 2080 blocks 45.5 against 55.5
    if (!y && lpr == 64 && img <= 78 * 1024 && rows >= kLnBlockRows) {
        unsigned char* o = static_cast<unsigned char*>(y_planes);
        const dim3 grid((unsigned)cdiv(rows, 32)), block(512);
        if (planes == 3) {
            if (img > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(ln_fwd_planes_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img);
            hipLaunchKernelGGL(ln_fwd_planes_kernel<3>, grid, block, img, st, x, ldx, rows, cols, gamma, beta, eps, mean, rstd, o, cbn);
        } else {
            if (img > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(ln_fwd_planes_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img);
            hipLaunchKernelGGL(ln_fwd_planes_kernel<2>, grid, block, img, st, x, ldx, rows, cols, gamma, beta, eps, mean, rstd, o, cbn);
        }
        MSN_LAUNCH_CHECK();
        return MSN_OK;
    }
    if (int rc = plane_tail_zero(y_planes, rows, cols, planes, st)) return rc;
    const PlaneOut po{static_cast<unsigned char*>(y_planes), planes, 2 * (int)cdiv(cols, 32), cols};
    MSN_LPR_DISPATCH(ln_fwd_kernel, lpr, dim3(ln_grid(rows, lpr)), 0, st, x, ldx, g, gamma, beta, eps, y, ldy, mean, rstd,
                     (unsigned short*)nullptr, po)
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// backward that ALSO writes the planes of dx: the gradient is consumed as the fp32 residual-stream gradient and as the plane
// operand of the next weight / input gradient products; dx_colsum (nullable): column sums of dx (a bias gradient)
extern "C" int msn_layernorm_bwd_planes(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t rows, int cols,
                                        const float* mean, const float* rstd, const float* gamma, const float* add,
                                        int64_t ldadd, float* dx, int64_t lddx, int planes, void* dx_planes, float* dgamma,
                                        float* dbeta, float* dx_colsum, void* ws, size_t ws_bytes, msn_stream_t stream) {
    if (int rc = check_rows("msn_layernorm_bwd_planes", rows, cols, {lddy, ldx, lddx}, {dy, x, dx, gamma, dx_planes})) return rc;
    MSN_REQUIRE(mean && rstd && dgamma && dbeta && (planes == 2 || planes == 3), "msn_layernorm_bwd_planes: bad argument");
    MSN_REQUIRE(!add || (ldadd >= cols && ldadd % 4 == 0 && (reinterpret_cast<uintptr_t>(add) & 15) == 0),
                "msn_layernorm_bwd_planes: bad residual-gradient operand");
    const int lpr = pick_lpr(cols);
    const int grid = ln_bwd_grid(rows, lpr);
    const int nacc = dx_colsum ? 3 : 2;
    MSN_REQUIRE(ws && ws_bytes >= sizeof(float) * nacc * (size_t)cols * grid, "msn_layernorm_bwd_planes: workspace too small");
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    {   // whole row blocks through LDS (ln_bwd_planes_kernel): the same rule as the forward's
        const int cbn = 2 * (int)cdiv(cols, 32);
        const size_t img = (size_t)cbn * planes * 1024;
        if (lpr == 64 && img <= 78 * 1024 && rows >= kLnBlockRows) {
            const int bgrid = (int)std::min<int64_t>(cdiv(rows, 32), std::min(grid, 512));       // <= grid: the workspace holds it
            const int nc = (int)cdiv(cols, 256);
            unsigned char* o = static_cast<unsigned char*>(dx_planes);
#define MSN_LN_BWD_BLOCK(NP_, NC_)                                                                                               \
    {                                                                                                                            \
        if (img > 64 * 1024)                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_planes_kernel<NP_, NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)img); \
        hipLaunchKernelGGL((ln_bwd_planes_kernel<NP_, NC_>), dim3(bgrid), dim3(512), img, st, dy, lddy, x, ldx, rows, cols, mean, rstd, \
                           gamma, dx, lddx, part, add, ldadd, dx_colsum ? 1 : 0, o, cbn);                                        \
    }
            if (planes == 3) {
                if (nc == 1) MSN_LN_BWD_BLOCK(3, 1) else if (nc == 2) MSN_LN_BWD_BLOCK(3, 2) else if (nc == 3) MSN_LN_BWD_BLOCK(3, 3) else MSN_LN_BWD_BLOCK(3, 4)
            } else {
                if (nc == 1) MSN_LN_BWD_BLOCK(2, 1) else if (nc == 2) MSN_LN_BWD_BLOCK(2, 2) else if (nc == 3) MSN_LN_BWD_BLOCK(2, 3) else MSN_LN_BWD_BLOCK(2, 4)
            }
#undef MSN_LN_BWD_BLOCK
            MSN_LAUNCH_CHECK();
            hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv(nacc * cols, 64)), dim3(1024), 0, st, part, bgrid, nacc * cols,
                               dgamma, dbeta, cols, dx_colsum);
            MSN_LAUNCH_CHECK();
            return MSN_OK;
        }
    }
    if (int rc = plane_tail_zero(dx_planes, rows, cols, planes, st)) return rc;
    const PlaneOut po{static_cast<unsigned char*>(dx_planes), planes, 2 * (int)cdiv(cols, 32), cols};
    const size_t lds = sizeof(float) * nacc * (size_t)cols * (256 / lpr);
    MSN_LPR_DISPATCH(ln_bwd_kernel, lpr, dim3(grid), lds, st, dy, lddy, x, ldx, g, mean, rstd, gamma, dx, lddx, part, add, ldadd,
                     (unsigned short*)nullptr, dx_colsum ? 1 : 0, po)
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv(nacc * cols, 64)), dim3(1024), 0, st, part, grid, nacc * cols,
                       dgamma, dbeta, cols, dx_colsum);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_l2norm_fwd(const float* x, int64_t ldx, int64_t rows, int cols, float* y, int64_t ldy,
                              float* inv_norm, msn_stream_t stream) {
    if (int rc = check_rows("msn_l2norm_fwd", rows, cols, {ldx, ldy}, {x, y})) return rc;
    MSN_REQUIRE(inv_norm, "msn_l2norm_fwd: null inv_norm");
    const int lpr = pick_lpr(cols);
    const RowGeom g{rows, cols, ldx};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MSN_LPR_DISPATCH(l2n_fwd_kernel, lpr, dim3(ln_grid(rows, lpr)), 0, st, x, ldx, g, y, ldy, inv_norm)
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, int64_t rows, int cols,
                              const float* inv_norm, float* dx, int64_t lddx, msn_stream_t stream) {
    if (int rc = check_rows("msn_l2norm_bwd", rows, cols, {lddy, ldy, lddx}, {dy, y, dx})) return rc;
    MSN_REQUIRE(inv_norm, "msn_l2norm_bwd: null inv_norm");
    const int lpr = pick_lpr(cols);
    const RowGeom g{rows, cols, ldy};
    hipStream_t st = static_cast<hipStream_t>(stream);
    MSN_LPR_DISPATCH(l2n_bwd_kernel, lpr, dim3(ln_grid(rows, lpr)), 0, st, dy, lddy, y, ldy, g, inv_norm, dx, lddx)
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_time_embed_fwd(const float* x, const float* t, int64_t B, int T, int e, const float* w,
                                  const float* bw, const float* omega, const float* band, int nband, float* out,
                                  msn_stream_t stream) {
    MSN_REQUIRE(B > 0 && T > 0 && e > 0 && e % 2 == 0, "msn_time_embed_fwd: bad sizes B=%lld T=%d e=%d", (long long)B, T, e);
    MSN_REQUIRE(x && t && w && bw && omega && out, "msn_time_embed_fwd: null pointer");
    MSN_REQUIRE(nband >= 1 && (nband == 1 || (band && T % nband == 0)),
                "msn_time_embed_fwd: T=%d must be divisible by nband=%d", T, nband);
    const int64_t total = B * T * e;
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 4096);
    hipLaunchKernelGGL(time_embed_fwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, t, B * T, T,
                       e, w, bw, omega, band, nband, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" size_t msn_time_embed_bwd_workspace_bytes(int64_t B, int e, int nband) {
    if (B <= 0 || e <= 0 || nband <= 0) return 0;
    return sizeof(float) * (size_t)B * (size_t)e * (size_t)(nband + 1);
}

extern "C" int msn_time_embed_bwd(const float* dy, const float* x, int64_t B, int T, int e, int nband, float* dw,
                                  float* dbw, float* dband, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(B > 0 && T > 0 && e > 0 && nband >= 1 && T % nband == 0, "msn_time_embed_bwd: bad sizes");
    MSN_REQUIRE(dy && x && dw && dbw && (nband == 1 || dband), "msn_time_embed_bwd: null pointer");
    MSN_REQUIRE(ws && ws_bytes >= msn_time_embed_bwd_workspace_bytes(B, e, nband), "msn_time_embed_bwd: workspace too small");
    MSN_REQUIRE(B < (1ll << 31), "msn_time_embed_bwd: batch too large");
    float* seg = static_cast<float*>(ws);
    float* sdx = seg + (size_t)B * nband * e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int cpt = std::min(e, 256);
    const size_t lds = sizeof(float) * (256 / cpt) * cpt;
    hipLaunchKernelGGL(time_embed_bwd_kernel, dim3((unsigned)B), dim3(256), lds, st, dy, x, T, e, nband, seg, sdx);
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(time_embed_bwd_finish_kernel, dim3((unsigned)cdiv(e, 64)), dim3(1024), 0, st, seg, sdx, (int)B, e,
                       nband, dw, dbw, nband > 1 ? dband : nullptr);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_masked_pool_fwd(const float* x, const uint8_t* mask, int64_t B, int T, int e, int mode, float* out,
                                   int* argmax, float* count, msn_stream_t stream) {
    MSN_REQUIRE(B > 0 && T > 0 && e > 0 && B < (1ll << 31), "msn_masked_pool_fwd: bad sizes");
    MSN_REQUIRE(x && mask && out && ((mode == POOL_MEAN && count) || (mode == POOL_MAX && argmax)),
                "msn_masked_pool_fwd: bad arguments");
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)B), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, mask, T, e, mode, out, argmax, count);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_masked_pool_bwd(const float* dout, const uint8_t* mask, int64_t B, int T, int e, int mode,
                                   const int* argmax, const float* count, float* dx, msn_stream_t stream) {
    MSN_REQUIRE(B > 0 && T > 0 && e > 0, "msn_masked_pool_bwd: bad sizes");
    MSN_REQUIRE(dout && mask && dx && ((mode == POOL_MEAN && count) || (mode == POOL_MAX && argmax)),
                "msn_masked_pool_bwd: bad arguments");
    const int64_t total = B * T * e;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dout, mask, B, T, e, mode, argmax, count, dx);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_mask_tokens(const float* x, const uint8_t* mask, int64_t rows, int e, float* y, msn_stream_t stream) {
    MSN_REQUIRE(rows > 0 && e > 0 && x && mask && y, "msn_mask_tokens: bad arguments");
    const int64_t total = rows * e;
    hipLaunchKernelGGL(mask_tokens_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, mask, rows, e, y);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// dst[r][c] += src[r][c] over a (rows x cols) window of two row-strided matrices (cols % 4 == 0, 16-byte aligned rows).
__global__ void add_rows_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src, int64_t lds,
                                int64_t rows, int cols4) {
    const int64_t total = rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols4;
        const int c = (int)(i % cols4) * 4;
        float4 a = *reinterpret_cast<const float4*>(dst + r * ldd + c);
        const float4 b = *reinterpret_cast<const float4*>(src + r * lds + c);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        *reinterpret_cast<float4*>(dst + r * ldd + c) = a;
    }
}

extern "C" int msn_add_rows(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t rows, int cols,
                            msn_stream_t stream) {
    MSN_REQUIRE(dst && src && rows > 0 && cols > 0 && cols % 4 == 0 && ldd % 4 == 0 && lds % 4 == 0 && ldd >= cols &&
                    lds >= cols && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0,
                "msn_add_rows: bad arguments (cols and both leading dimensions must be multiples of 4, 16-byte aligned)");
    const int64_t total = rows * (cols / 4);
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dst, ldd, src, lds, rows, cols / 4);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_vit_tokens_fwd(const float* patch, const float* cls, const float* pos, int64_t B, int T, int e,
                                  float* tok, msn_stream_t stream) {
    MSN_REQUIRE(patch && cls && pos && tok && B > 0 && T > 1 && e > 0, "msn_vit_tokens_fwd: bad arguments");
    hipLaunchKernelGGL(vit_tokens_fwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv(B * T * e, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), patch, cls, pos, B, T, e, tok);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_vit_tokens_bwd(const float* dtok, int64_t B, int T, int e, float* dpatch, msn_stream_t stream) {
    MSN_REQUIRE(dtok && dpatch && B > 0 && T > 1 && e > 0, "msn_vit_tokens_bwd: bad arguments");
    hipLaunchKernelGGL(vit_tokens_bwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv(B * (T - 1) * e, 256), 4096)),
                       dim3(256), 0, static_cast<hipStream_t>(stream), dtok, B, T, e, dpatch);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_series_features(const float* x, const float* t, const uint8_t* mask, int64_t rows, float inv_norm,
                                   float* feat, msn_stream_t stream) {
    MSN_REQUIRE(x && t && mask && feat && rows > 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0,
                "msn_series_features: bad arguments");
    hipLaunchKernelGGL(series_features_kernel, dim3((unsigned)std::min<int64_t>(cdiv(rows, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, t, mask, rows, inv_norm, reinterpret_cast<float4*>(feat));
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_masked_mse_fwd(const float* pred, const float* target, const uint8_t* select, int64_t n, float* stats,
                                  msn_stream_t stream) {
    MSN_REQUIRE(pred && target && select && stats && n > 0, "msn_masked_mse_fwd: bad arguments");
    hipLaunchKernelGGL(masked_mse_fwd_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), pred, target,
                       select, n, stats);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
extern "C" int msn_masked_mse_bwd(const float* pred, const float* target, const uint8_t* select, int64_t n,
                                  const float* stats, const float* grad_out, float* dpred, msn_stream_t stream) {
    MSN_REQUIRE(pred && target && select && stats && grad_out && dpred && n > 0, "msn_masked_mse_bwd: bad arguments");
    hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 2048)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pred, target, select, n, stats, grad_out, dpred);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

constexpr int AUG_PARTS = 256;
extern "C" size_t msn_augment_workspace_bytes(void) { return sizeof(double) * 2 * AUG_PARTS; }

// out = rot90^{rot[b]}( img + (2 u - 1) * noise_level * std(img) ), square (B, C, S, S) images, std over the whole
// batch (unbiased, torch.std);  u: uniform [0,1) field of the same shape;  rot: B ints (quarter turns, CCW).
extern "C" int msn_augment_images(const float* img, const float* u, const int* rot, int64_t B, int C, int S,
                                  float noise_level, float* out, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(img && u && rot && out && B > 0 && C > 0 && S > 0, "msn_augment_images: bad arguments");
    MSN_REQUIRE(ws && ws_bytes >= msn_augment_workspace_bytes(), "msn_augment_images: workspace too small");
    const int64_t n = B * C * S * S;
    MSN_REQUIRE(n > 1, "msn_augment_images: need more than one pixel for a standard deviation");
    double* part = static_cast<double*>(ws);
    double* part2 = part + AUG_PARTS;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(sum_kernel, dim3(AUG_PARTS), dim3(1024), 0, st, img, n, part);
    hipLaunchKernelGGL(sqdev_kernel, dim3(AUG_PARTS), dim3(1024), 0, st, img, n, part, AUG_PARTS, part2);
    hipLaunchKernelGGL(augment_images_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0, st, img, u,
                       rot, B, C, S, noise_level, part2, AUG_PARTS, n, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
// out = x + g * err * noise_level   (magnitudes / spectra; g: standard-normal field)
extern "C" int msn_augment_series(const float* x, const float* g, const float* err, int64_t n, float noise_level,
                                  float* out, msn_stream_t stream) {
    MSN_REQUIRE(x && g && err && out && n > 0, "msn_augment_series: bad arguments");
    hipLaunchKernelGGL(augment_series_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, g, err, n, noise_level, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// y = dropout(x; p, seed) (+ residual).  In place (y == x) allowed.  0 <= p < 1.
extern "C" int msn_dropout(const float* x, int64_t n, float p, uint64_t seed, const float* residual, float* y,
                           msn_stream_t stream) {
    MSN_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "msn_dropout: bad arguments (p = %f)", (double)p);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n, p, seed, static_cast<const uint64_t*>(nullptr), residual, y);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

// The same with the seed = seed_base[0] (device) + seed_offset: for a training step recorded in a HIP graph.
extern "C" int msn_dropout_dev(const float* x, int64_t n, float p, const uint64_t* seed_base, uint64_t seed_offset,
                               const float* residual, float* y, msn_stream_t stream) {
    MSN_REQUIRE(x && y && seed_base && n > 0 && p >= 0.f && p < 1.f, "msn_dropout_dev: bad arguments (p = %f)", (double)p);
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n, p, seed_offset, seed_base, residual, y);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
__global__ void seed_advance_kernel(uint64_t* seed_base) { seed_base[0] = seed_base[0] * 6364136223846793005ull + 1442695040888963407ull; }
// seed_base[0] <- next value of a 64-bit LCG: one launch per (recorded) training step
extern "C" int msn_seed_advance(uint64_t* seed_base, msn_stream_t stream) {
    MSN_REQUIRE(seed_base, "msn_seed_advance: null pointer");
    hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), seed_base);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
