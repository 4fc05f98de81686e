// Pieces shared by the fp32 (gemm.hip) and bf16 (gemm_bf16.hip) MFMA GEMM kernels.
#pragma once
#include <type_traits>

#include "msn_common.h"

namespace msn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int KPAD = 4;  // floats; keeps 16-B alignment and makes b128 fragment reads conflict-free

// Implicit-GEMM convolution: one operand of the product is the column matrix of a channels-last image that is never
// written -- the LDS-DMA lanes compute their own source address per K-step and taps that fall outside the image read a
// page of zeros.  The matrix index that walks pixels (the row m of the A operand for forward / dgrad, the k index of
// the B operand for wgrad) enumerates (b, ry, rx) over RH x RW; tap (ky, kx) of that pixel reads the source pixel
// (ry * sy + y0 + dir * ky, rx * sx + x0 + dir * kx) of src[b][SH][SW][SC]; the other index walks (tap, channel)
// with the channel fastest.  forward: rows = output pixels, y0 = -pad, dir = +1; dgrad (stride 1): rows = input
// pixels, src = dY, y0 = +pad, dir = -1; wgrad: k = output pixels, columns = (tap, input channel).
struct ConvGather {
    const float* src;
    const float* zero;             // at least 16 bytes of zeros
    int SH, SW, SC, RH, RW, sy, sx, y0, x0, dir, kw;
    unsigned long long rw_magic, rh_magic;   // floor(2^36 / d) + 1: n / d == (n * magic) >> 36 for n < 2^27, d <= 512
};

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    float* aux;
    int64_t M, N, K;
    int64_t lda, ldb, ldc, ldaux;
    int epilogue;
    int tiles_m, tiles_n;
    int k_per_split;  // multiple of BK
    int splits;
    float* partial;   // [splits][M][N] when splits > 1
    // The last `tail_tiles` tiles (those that would run as a partly filled final round of workgroups) are cut
    // into `tail_splits` K-slabs of `tail_kps`; each slab's accumulators go to tail_partial in register order and
    // tail_finish_kernel sums them in slab order and applies the epilogue.  0 tiles = no tail.
    int tail_tiles, tail_splits, tail_kps;
    float* tail_partial;   // [tail_tiles][tail_splits][BM * BN]
    // In-kernel finish: one arrival counter per tail tile (zero between launches: the finishing workgroup resets it).
    // The workgroup that stores a tile's LAST slab sums the slabs and applies the epilogue itself.  Null = the
    // slabs are summed by a tail_finish_kernel launch instead.
    unsigned* tail_counter;
    ConvGather cg;         // kernels instantiated with CONV != 0 only
    float* colsum;         // wgrad + bias grad: [splits][M] sums over k of A (opA = T) or null
#ifdef MSN_TIMELINE
    unsigned long long* dbg;   // per workgroup: 4 timestamps + HW_ID + XCC_ID (diagnostic builds only)
#endif
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective for any nwg: XCD x (= bid % 8) owns a contiguous chunk of logical ids
    const int q = nwg / 8, r = nwg % 8, x = bid % 8;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + bid / 8;
}


// Which tile and K range this workgroup owns.  Regular workgroups come first in one flat XCD-mapped run over
// (split, tile): XCD x owns a contiguous piece of it, so the tiles of one K-slab -- which re-read the same operand
// rows -- share that XCD's L2.  Tail workgroups follow (dispatched last, dealt round-robin over the XCDs).
struct TileCoord {
    int logical, split, tail_slab;   // tail_slab < 0: regular workgroup
    int64_t k_begin, k_end;
};
// `w` = position in the launch-order list of workgroups (blockIdx.x for one workgroup per item; a persistent
// workgroup walks w = blockIdx.x, + gridDim.x, ... -- gridDim.x is a multiple of 8, so it stays on its XCD's run).
__device__ __forceinline__ TileCoord locate_tile(const GemmArgs& p, int w) {
    const int nreg = p.tiles_m * p.tiles_n - p.tail_tiles;
    const int regular = nreg * p.splits;
    TileCoord c;
    if (w < regular) {
        const int flat = xcd_remap(w, regular);
        c.logical = flat % nreg;
        c.split = flat / nreg;
        c.tail_slab = -1;
        c.k_begin = (int64_t)c.split * p.k_per_split;
        c.k_end = min(p.K, c.k_begin + (int64_t)p.k_per_split);
    } else {
        const int t = w - regular;
        c.logical = nreg + t / p.tail_splits;
        c.split = t % p.tail_splits;
        c.tail_slab = t;
        c.k_begin = (int64_t)c.split * p.tail_kps;
        c.k_end = min(p.K, c.k_begin + (int64_t)p.tail_kps);
    }
    return c;
}
__device__ __forceinline__ TileCoord locate_tile(const GemmArgs& p) { return locate_tile(p, (int)blockIdx.x); }
__device__ __forceinline__ int work_items(const GemmArgs& p) {
    return (p.tiles_m * p.tiles_n - p.tail_tiles) * p.splits + p.tail_tiles * p.tail_splits;
}

__device__ __forceinline__ int l32_of(int lane) { return lane & 31; }
// Accumulators of one tail K-slab, in register order (every store instruction writes 256 contiguous bytes).
template <int TM, int TN>
__device__ __forceinline__ void dump_tail(const f32x16 (&acc)[TM][TN], const GemmArgs& p, int slab, int wave,
                                          int nwaves, int lane) {
    float* base = p.tail_partial + ((int64_t)slab * nwaves + wave) * (TM * TN * 16 * 64) + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) base[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
}

// Epilogue of a (TM x TN) grid of 32x32 accumulator tiles (C/D layout of every 32x32 MFMA on gfx950,
// fp32 and bf16 alike: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 h).
// The epilogue kind and "this wave's tile lies wholly inside C" are template parameters, selected once per wave by
// a uniform switch: the interior case is straight-line code (no per-element compare / exec mask), and each kind
// carries only its own arithmetic.  The aux operand of a tile is read as one batch of 16 loads before any
// arithmetic, so the loads overlap instead of paying one memory round trip per element.
template <int TM, int TN, int EPI, bool FULL>
__device__ __forceinline__ void epilogue_body(const f32x16 (&acc)[TM][TN], const GemmArgs& p, float* __restrict__ out,
                                              int64_t ldo, int64_t m0, int64_t n0, int wm0, int wn0, int l32, int h) {
    constexpr bool READS_AUX = EPI == MSN_EPI_RELU_BWD || EPI == MSN_EPI_GELU_BWD || EPI == MSN_EPI_ADD;
    const bool saves_aux = EPI == MSN_EPI_GELU && p.aux != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t col = n0 + wn0 + 32 * j + l32;
            const bool col_ok = FULL || col < p.N;
            const float bv = (p.bias && col_ok) ? p.bias[col] : 0.f;
            const int64_t row0 = m0 + wm0 + 32 * i + 4 * h;
            float* __restrict__ o = out + row0 * ldo + col;
            float* __restrict__ x = p.aux + row0 * p.ldaux + col;   // only dereferenced by the kinds that use aux
            float av[16];
            if (READS_AUX) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    av[r] = (FULL || (col_ok && row0 + dr < p.M)) ? x[dr * p.ldaux] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                const bool ok = FULL || (col_ok && row0 + dr < p.M);
                float v = acc[i][j][r] + bv;
                if (EPI == MSN_EPI_RELU) v = fmaxf(v, 0.f);
                else if (EPI == MSN_EPI_GELU) {
                    // erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7: measured 4.6e-7 / 2.7e-7 on gelu / gelu' for
                    // |x| <= 12 against fp64), sharing exp(-v^2 / 2) with the pdf term: 1 rcp + 1 exp + ~12 FMAs per
                    // element instead of the ~40 instructions of erff + a second exp -- 8 % of the ff1 product's time
                    const float z = v * 0.70710678118654752f, az = fabsf(z);
                    const float t = __frcp_rn(fmaf(0.3275911f, az, 1.f));
                    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f),
                                                        -0.284496736f), 0.254829592f);
                    const float e = __expf(-z * z);
                    const float cdf = 0.5f + 0.5f * copysignf(1.f - poly * e, z);
                    if (saves_aux && ok) x[dr * p.ldaux] = cdf + v * 0.39894228040143268f * e;
                    v *= cdf;
                } else if (EPI == MSN_EPI_RELU_BWD) v = av[r] > 0.f ? v : 0.f;
                else if (EPI == MSN_EPI_GELU_BWD) v *= av[r];
                else if (EPI == MSN_EPI_ADD) v += av[r];
                if (ok) o[dr * ldo] = v;
            }
        }
}

template <int TM, int TN, int EPI>
__device__ __forceinline__ void epilogue_kind(const f32x16 (&acc)[TM][TN], const GemmArgs& p, float* out, int64_t ldo,
                                              int64_t m0, int64_t n0, int wm0, int wn0, int l32, int h) {
    const bool full = m0 + wm0 + 32 * TM <= p.M && n0 + wn0 + 32 * TN <= p.N;   // wave-uniform
    if (full) epilogue_body<TM, TN, EPI, true>(acc, p, out, ldo, m0, n0, wm0, wn0, l32, h);
    else epilogue_body<TM, TN, EPI, false>(acc, p, out, ldo, m0, n0, wm0, wn0, l32, h);
}

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[TM][TN], const GemmArgs& p, int64_t m0, int64_t n0,
                                              int wm0, int wn0, int l32, int h, int split) {
    if (p.splits > 1) {   // split-K slab: plain store into this split's [M][N] partial (bias is null by contract)
        epilogue_kind<TM, TN, MSN_EPI_NONE>(acc, p, p.partial + (int64_t)split * p.M * p.N, p.N, m0, n0, wm0, wn0, l32, h);
        return;
    }
    switch (p.epilogue) {
        case MSN_EPI_RELU: epilogue_kind<TM, TN, MSN_EPI_RELU>(acc, p, p.C, p.ldc, m0, n0, wm0, wn0, l32, h); break;
        case MSN_EPI_GELU: epilogue_kind<TM, TN, MSN_EPI_GELU>(acc, p, p.C, p.ldc, m0, n0, wm0, wn0, l32, h); break;
        case MSN_EPI_RELU_BWD: epilogue_kind<TM, TN, MSN_EPI_RELU_BWD>(acc, p, p.C, p.ldc, m0, n0, wm0, wn0, l32, h); break;
        case MSN_EPI_GELU_BWD: epilogue_kind<TM, TN, MSN_EPI_GELU_BWD>(acc, p, p.C, p.ldc, m0, n0, wm0, wn0, l32, h); break;
        case MSN_EPI_ADD: epilogue_kind<TM, TN, MSN_EPI_ADD>(acc, p, p.C, p.ldc, m0, n0, wm0, wn0, l32, h); break;
        default: epilogue_kind<TM, TN, MSN_EPI_NONE>(acc, p, p.C, p.ldc, m0, n0, wm0, wn0, l32, h); break;
    }
}

// One K-slab of a tail tile is done: store it; with arrival counters, the workgroup that stores the tile's last slab
// sums all of them IN SLAB ORDER (its own included, re-read: the result does not depend on which workgroup finishes)
// and applies the epilogue -- the finishing pass without a launch of its own.  `flag` = one LDS word no wave still
// reads (the operand buffers are dead after the K loop; the barrier below orders that).
// Visibility across the 8 XCDs (one L2 each) WITHOUT a device-scope fence -- a release fence writes back the whole
// L2 of the XCD (measured: +2 us on every launch with a tail, 85.0 -> 87.8 ms per step): the slab words themselves are
// stored and re-read as relaxed device-scope atomics (write-through / coherent reads, sc1), every thread waits for
// its own stores (vmcnt(0)) before the workgroup barrier, and only then is the arrival counted.
template <int TM, int TN>
__device__ __forceinline__ void finish_tail(f32x16 (&acc)[TM][TN], const GemmArgs& p, const TileCoord& tc, int wave,
                                            int nwaves, int lane, int64_t m0, int64_t n0, int wm0, int wn0,
                                            unsigned* flag) {
    if (p.tail_counter == nullptr) {
        dump_tail<TM, TN>(acc, p, tc.tail_slab, wave, nwaves, lane);
        return;
    }
    {
        float* base = p.tail_partial + ((int64_t)tc.tail_slab * nwaves + wave) * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(base + ((i * TN + j) * 16 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int tile = tc.tail_slab / p.tail_splits;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's slab words have reached the coherence point
    __syncthreads();                                   // ... and so have those of every thread of the workgroup
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(p.tail_counter + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = prev + 1u == (unsigned)p.tail_splits ? 1u : 0u;
        if (last) __hip_atomic_store(p.tail_counter + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < p.tail_splits; ++s) {
        const float* base = p.tail_partial + (((int64_t)tile * p.tail_splits + s) * nwaves + wave) * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += __hip_atomic_load(base + ((i * TN + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gemm_epilogue<TM, TN>(acc, p, m0, n0, wm0, wn0, l32_of(lane), lane >> 5, 0);
}

// ---- LDS-DMA operand tiles (gemm.hip, gemm_pw.hip) ------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int ROWS, bool KMAJOR, int DBK>
struct DmaTile {
    static constexpr int kFloats = ROWS * DBK;            // unpadded image
    static constexpr int kPieces = ROWS * DBK / 256;      // 1-KB LDS-DMA pieces per K-step
    static constexpr int CPR = DBK / 4;                   // 16-byte chunks per row of a K-contiguous image
    static constexpr int RPB = DBK >= 64 ? 1 : 64 / DBK;  // rows of a K-contiguous image per 256-B bank row
    // XOR swizzle of a row's chunk positions.  ds_read_b128 serves 16 lanes (16 different rows, same logical chunk)
    // per LDS cycle; they must land on 16 different 16-B slots of the bank row.  Rows r and r' share a slot range
    // when r % RPB == r' % RPB, so the swizzle key must differ between them: (r / RPB) % CPR does for every row set
    // {0-3, 12-15, 20-27} + 4j + 32h the instruction groups (MI355X_MICROARCH.md, LDS).
    __device__ static __forceinline__ int swz(int r) { return (r / RPB) % CPR; }
    // source address of this lane for piece q of the K-step starting at k0
    __device__ static __forceinline__ const float* src(const float* __restrict__ g, int64_t ld, int64_t row0,
                                                       int64_t nrows, int64_t k0, int q, int lane) {
        if (KMAJOR) {   // image [k][ROWS]: piece = 256 / ROWS consecutive k-rows
            constexpr int LPR = ROWS / 4;                                  // lanes per k-row
            const int kk = q * (64 / LPR) + lane / LPR;
            int64_t c = row0 + 4 * (lane % LPR);
            c = c + 3 < nrows ? c : nrows - 4;
            return g + (k0 + kk) * ld + c;
        } else {        // image [ROWS][DBK]: piece = 64 / CPR rows; chunk c of row r sits at position c ^ swz(r)
            const int r = q * (64 / CPR) + lane / CPR, pos = lane % CPR;
            int64_t rr = row0 + r;
            rr = rr < nrows ? rr : nrows - 1;
            return g + rr * ld + k0 + 4 * (pos ^ swz(r));
        }
    }
    // LDS reads are issued from inline asm: hipcc's waitcnt pass cannot prove that a ds_read does not alias an
    // LDS-DMA write still in flight and would put s_waitcnt vmcnt(0) in front of every k-step's first read
    // (draining the whole ring); an asm read is invisible to that pass, so its completion is counted by hand
    // (lgkmcnt) in the kernel.  `tile_addr` = LDS byte address of the image.
    // A fragment = this lane's 4 consecutive k values of one row.  The asm outputs bind straight to the registers
    // the MFMAs read (no copy may sit between the asynchronous read and the hand-placed s_waitcnt).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    struct FragKM { f32x2 lo, hi; };
    struct FragKC { float4 v; };
    using Frag = std::conditional_t<KMAJOR, FragKM, FragKC>;
    template <int C>
    __device__ static __forceinline__ float get(const Frag& f) {
        if constexpr (KMAJOR) return C == 0 ? f.lo.x : C == 1 ? f.lo.y : C == 2 ? f.hi.x : f.hi.y;
        else return C == 0 ? f.v.x : C == 1 ? f.v.y : C == 2 ? f.v.z : f.v.w;
    }
    static constexpr int kReads = KMAJOR ? 2 : 1;   // LDS instructions per fragment
    __device__ static __forceinline__ void frag_issue(Frag& f, unsigned tile_addr, int row, int ko, int h) {
        if constexpr (KMAJOR) {   // k, k+1 | k+2, k+3 of this lane half: two ds_read2_b32 (dword offsets 0, ROWS)
            static_assert(ROWS <= 255, "ds_read2_b32 offset1 is 8 bits (dwords)");
            const unsigned a = tile_addr + 4u * ((8 * ko + 4 * h) * ROWS + row);
            asm volatile("ds_read2_b32 %0, %1 offset1:%2" : "=v"(f.lo) : "v"(a), "n"(ROWS));
            asm volatile("ds_read2_b32 %0, %1 offset1:%2" : "=v"(f.hi) : "v"(a + 8u * ROWS), "n"(ROWS));
        } else {
            const unsigned a = tile_addr + 4u * (row * DBK + 4 * ((2 * ko + h) ^ swz(row)));
            asm volatile("ds_read_b128 %0, %1" : "=v"(f.v) : "v"(a));
        }
    }
};


// bf16 matrix-core variant (gemm_bf16.hip): planes = 1 -> operands rounded to bf16, planes = 2 -> each
// operand split hi + lo and three products accumulated (fp32-grade accuracy).  Needs the vector-load
// conditions of the VEC fp32 kernels; the caller falls back to the fp32 kernels otherwise.
int launch_bgemm(const GemmArgs& a, int opA, int opB, int planes, int bm, int bn, hipStream_t st);

// gemm_list.hip: the work-list (stream-K) launch
bool gemm_list_takes(int n, const msn_gemm_desc* d);
size_t gemm_list_ws_bytes();
int gemm_list_launch(int n, const msn_gemm_desc* d, void* ws, size_t ws_bytes, hipStream_t st);
// gemm.hip: one zeroed slice of 512 arrival counters per (device, stream); null when all slices are taken
unsigned* gemm_counter_slice(hipStream_t st);

inline unsigned gemm_grid(const GemmArgs& a) {
    return (unsigned)((a.tiles_m * a.tiles_n - a.tail_tiles) * a.splits + a.tail_tiles * a.tail_splits);
}

}  // namespace msn
