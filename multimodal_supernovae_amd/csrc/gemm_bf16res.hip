// GEMMs with bf16-RESIDENT operands on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate) -- the
// BASELINE cfg5 image tower (build-defined ViT-B/16; no reference counterpart: the reference has no mixed precision).
//
// The older msn::bgemm_kernel keeps every tensor fp32 in HBM and rounds to bf16 between the global load and the LDS
// write: with 128 x 128 tiles that moves 32 KB L2 -> LDS per MFLOP and is L2-bound at 13 % of the matrix peak.  Here
// the operands ARE bf16 in HBM (LayerNorm / GELU / attention epilogues write them; weights are rounded once per step),
// so a K-tile travels global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) at half
// the bytes, and the tile is 256 x 256 (16 KB L2 -> LDS per MFLOP).
//
//   msn_bgemm_nt :  C[M][N] = epi( A[M][K] . B[N][K]^T + bias )      both operands K-contiguous
//                   (forward: activations x weight; dgrad: dY x W^T with the pre-transposed bf16 weight copy)
//   msn_bgemm_tn :  C[N][K] = sum_m A[m][N]^T . B[m][K]               both operands reduction-MAJOR
//                   (wgrad: dY^T . X; fragments come out of LDS transposed by ds_read_b64_tr_b16; split over m,
//                    fp32 partial slabs summed in a fixed order)
//
// Workgroup = 8 waves (NT: 4 x 2, TN: 2 x 4), tile 256 x 256, K-step 64, two LDS K-tile buffers of 64 KB.  A wave owns 64 x 128
// (TN: 128 x 64) of the tile = 4 x 8 MFMA tiles of 16 x 16 (128 accumulator registers) and walks a K-tile in four phases of 16
// MFMAs (k32 half x 64-column half); the fragments of the next phase are requested before the MFMAs of the current one
// (ds_read from inline asm with hand-counted lgkmcnt: hipcc would otherwise drain the LDS-DMA queue with vmcnt(0)
// before every LDS read), the LDS-DMA of K-tile t + 2 is issued in the middle of K-tile t (one full K-tile of lead),
// ONE raw s_barrier per K-tile.  NT: the weight rows of a tile sit PERMUTED in LDS so that a lane ends up with eight (fp32
// output: two runs of four) CONSECUTIVE output columns of a row and sixteen lanes side by side on it: a store instruction is
// 4 rows x 256 contiguous bytes (see `offB`).  LDS images are written linearly by the DMA and XOR-swizzled on the
// SOURCE address, with the same XOR on the fragment reads: conflict-free for both read shapes (derivation at `swz`).
#include <algorithm>
#include <type_traits>

#include "msn_common.h"

namespace msn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int BT = 256;            // tile rows / columns
constexpr int BKS = 64;            // K-step (elements)
constexpr int OPER_BYTES = BT * BKS * 2;      // one operand of one K-tile: 32 KB
constexpr int STAGE_BYTES = 2 * OPER_BYTES;   // 64 KB
constexpr int EPI_B_NONE = 0, EPI_B_GELU = 1, EPI_B_GELU_BWD = 2, EPI_B_ADD = 3;

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ u16 f2bf(float f) {          // round to nearest even; NaN stays NaN (plain cast)
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const u16*>(&b);
}

// GELU (erf form) and its derivative for bf16 outputs, without transcendentals: Phi(x) - 1/2 and gelu'(x) - 1/2 are odd
// functions; on |x| <= 4 (clamped: Phi(4) = 1 - 3e-5) they are evaluated as odd polynomials in t = x / 4 of degree 15 / 17
// (least-squares minimax fit, fp32 Horner: |error| <= 2.4e-5 / 7.7e-5 against the erf form -- bf16 keeps 4e-3 relative).
// The epilogue of a 256 x 256 tile evaluates 128 of these per thread with nothing to overlap them (one 128-KB workgroup
// per CU): libm's erff costs a third of the tile's MFMA time, Abramowitz-Stegun 7.1.26 (1 rcp + 1 exp + 13 FMAs) a
// quarter, the polynomial (8 - 9 FMAs + 4) a sixth.
// (Round 6: with the stores out of the way the epilogue's arithmetic shows -- 55 us of a 595-us GELU launch.  The polynomials are evaluated
//  in x itself -- the coefficients below are the fitted ones times 4^-(2k+1) -- and the clamp is ONE v_med3_f32: 12 instead of 14 vector
//  instructions per value.)
__device__ __forceinline__ float gelu_fast(float x) {
    constexpr float c[8] = {3.988475204e-01f, -6.617539376e-02f, 9.664885700e-03f, -1.048208098e-03f, 8.066803275e-05f, -4.100923888e-06f, 1.217137253e-07f, -1.580833464e-09f};
    const float xc = __builtin_amdgcn_fmed3f(x, -4.f, 4.f);
    const float u = xc * xc;
    float a = c[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) a = fmaf(a, u, c[k]);
    return x * fmaf(a, xc, 0.5f);                                      // x * Phi(x)
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    constexpr float c[9] = {7.976095676e-01f, -2.648265958e-01f, 5.845616758e-02f, -8.716349490e-03f, 9.073331021e-04f, -6.495839625e-05f, 3.028406582e-06f, -8.219024039e-08f, 9.796399247e-10f};
    const float xc = __builtin_amdgcn_fmed3f(x, -4.f, 4.f);
    const float u = xc * xc;
    float a = c[8];
#pragma unroll
    for (int k = 7; k >= 0; --k) a = fmaf(a, u, c[k]);
    return fmaf(a, xc, 0.5f);                                          // Phi(x) + x phi(x)
}

struct BgemmArgs {
    const u16* A; const u16* B;
    void* C; void* aux; const float* bias;
    int64_t lda, ldb, ldc, ldaux;
    int64_t M;                 // NT: rows of A / C.  TN: reduction length
    int N, K;                  // NT: C is M x N, reduction K.  TN: C is N x K
    int tiles_m, tiles_n;      // NT: tile grid.  TN: tiles over N and K
    int super_rows;            // NT: tile-rows walked together (L2 blocking)
    float* colpart;            // NT (nullable): [4 * tiles_m][N] column sums of the values written to C, per 64-row slab
    int splits; int64_t rows_per_split;   // TN: reduction split
    float* slabs;              // TN: [splits][N][K] partials (splits > 1)
};

// ---------------------------------------------------------------------------------------------------------------
// NT: LDS image of one operand K-tile = [256 rows][128 B] (row = 64 bf16 of k), 16-byte chunk c of row r stored at
// chunk position c ^ swz(r), swz(r) = (r >> 1) & 7.  A ds_read_b128 of the 16x16x32 operand takes, per 16-lane issue
// group, rows {0-3, 12-15} at chunk c and rows {4-11} at chunk c ^ 1 (lane l: row l & 15, chunk 4 kk + (l >> 4)):
// 16-byte slot in the 256-byte bank row = 8 (r & 1) + (c ^ swz(r)); even rows land on 8 distinct slots of the lower
// half, odd rows on 8 distinct slots of the upper half -> conflict-free.
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

__device__ __forceinline__ void ds_read128(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}

#ifndef MSN_BGEMM_STAGGER
#define MSN_BGEMM_STAGGER 0
#endif
constexpr bool g_stagger = MSN_BGEMM_STAGGER != 0;

template <int EPI, bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void bgemm_nt_kernel(const BgemmArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // static priority for the second-dispatched half of the workgroup: waves 4-7 lose the per-SIMD issue arbitration (priority, then
    // age) on every phase otherwise (MI355X_MICROARCH.md, two waves per SIMD); cfg5 step -0.4 % in three interleaved pairs.  (The plane
    // GEMMs of pgemm_kernels.h: +0.3 % with the same line -- not there.)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    // Persistent: workgroup b multiplies the tiles t = b, b + gridDim.x, ... (one workgroup per CU; gridDim.x is a multiple
    // of 8 or the number of tiles).  The ring of K-tiles runs on ACROSS tile boundaries: the first two K-tiles of the next
    // tile are requested during the last two K-tiles of this one, so they land under the epilogue instead of in front of
    // the next tile's first MFMA -- with one workgroup per CU nothing else would cover that prologue.
    // XCD-aware order: tile ids are dealt round-robin to the 8 XCDs (t % 8 == blockIdx.x % 8); give each XCD a contiguous
    // run of tiles, and walk the tiles N-fastest so that the workgroups of one XCD share A row panels (and the whole
    // weight) in its L2.
    // Inside an XCD's run: super-rows of 8 tile-rows, row-fastest within a tile column -- the 32 workgroups an XCD has in
    // flight then cover 8 A row panels (3 MB at K = 768) x 4 weight column panels, so a weight panel is fetched once per
    // super-row instead of once per tile-row (rocprofv3 FETCH_SIZE of the N = 3072, K = 768 products: 2.1 GB per launch
    // against 0.78 GB of operands with the N-fastest order; the weight alone is 4.7 MB against 4 MB of L2)
    // (only while the super-row's A panels fit an L2 next to the weight stream: SR = 4096 / K tile-rows, 1 for K >= 2304)
    const int total = p.tiles_m * p.tiles_n;
    auto locate = [&](int t, int64_t& m0_, int& n0_, int& tm_) {
        const int q = total / 8, r = total % 8, x = t % 8, i = t / 8;
        const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
        const int SR = p.super_rows;
        const int sr = bid / (SR * p.tiles_n), j = bid % (SR * p.tiles_n);
        const int rows_sr = min(SR, p.tiles_m - sr * SR);
        tm_ = sr * SR + j % rows_sr;
        m0_ = (int64_t)tm_ * BT;
        n0_ = (j / rows_sr) * BT;
    };
    // ---- LDS-DMA sources: wave w moves pieces 4w .. 4w+3 (8 rows x 128 B each) of both operands, by BUFFER loads
    // (buffer_load_dwordx4 ... lds): one resource descriptor per operand and requested tile (base = the tile's first row,
    // num_records = the bytes from there to the end of the matrix, in SGPRs) + a 32-bit per-lane byte offset.  Rows past the
    // edge of the matrix are out of range for the descriptor and read as zeros -- no clamping -- and the eight 64-bit per-lane
    // pointers of rounds 2-3 (16 VGPRs: the kernel stood at 255 with 12 bytes of scratch per lane) are gone.
    // (The bounds check covers the per-lane offset, not the scalar one: the K-tile advance goes into the scalar offset, the
    // row of a piece into the per-lane one.)
    const int lrow = lane >> 3;
    __amdgpu_buffer_rsrc_t rsrcA, rsrcB;      // of the tile whose K-tiles are being REQUESTED: the next tile's from its last-but-one K-tile on
    auto make_src = [&](int64_t m0_, int n0_) {
#ifdef MSN_ABL_BF_SAMEPANEL               // diagnostic build: every tile reads the panels of tile (0, 0) -- L2 hits only
        m0_ = 0, n0_ = 0;
#endif
        const int64_t bytesA = std::max<int64_t>(p.M - m0_, 0) * p.lda * 2, bytesB = std::max<int64_t>((int64_t)p.N - n0_, 0) * p.ldb * 2;
        rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.A + m0_ * p.lda), (short)0,
                                                  (int)std::min<int64_t>(bytesA, 0x7fffffff), 0x00020000);
        rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(p.B + (int64_t)n0_ * p.ldb), (short)0,
                                                  (int)std::min<int64_t>(bytesB, 0x7fffffff), 0x00020000);
    };
    // per-lane byte offsets inside a piece: LDS row 8 q' + lrow (q' = 4 wave + q), 16-byte chunk (lane & 7) ^ swz(LDS row);
    // swz(row) = (row >> 1) & 7 = (4 q' + (lrow >> 1)) & 7, and 4 q' & 7 is 0 or 4 by the parity of q: two offsets per operand.
    // A: LDS row = tile row.  B: the LDS image holds the tile's 256 weight rows (= output columns) PERMUTED -- LDS row 128 h + 16 j + l
    // (h: the wave column, j: n-tile, l: the MFMA column index) is output column
    //     bf16 output:  128 h + 8 l + j               (lane l of n-tile j = 0 .. 7: EIGHT consecutive columns per lane, 16 bytes)
    //     fp32 output:  128 h + 64 (j >> 2) + 4 l + (j & 3)   (two runs of FOUR consecutive columns per lane, 16 bytes each)
    // so that, with the accumulator layout of the un-swapped product (lane = column index l, registers = rows 4 g + r), ONE store
    // instruction writes 4 rows x 256 contiguous bytes: sixteen lanes side by side on a row.  That shape is what the memory pipe of
    // a CU takes fastest, by far (tools/microbench/store_patterns.hip, 128 KB per tile and CU between 20-us idle phases: 1.9 us;
    // 16 rows x 64 B 4.3; 8 rows x 128 B and 16 rows x 32 B 8 - 9), and behind a 1.9-us K-tile the epilogue's stores are what a
    // persistent workgroup waits for (profiles/r06_bgemm_epilogue.txt).  The permutation costs nothing: an LDS-DMA piece is 8 rows of
    // 128 B either way; its rows are STR = 8 (4) weight rows apart instead of adjacent.
    constexpr int STR = OUT_BF16 ? 8 : 4;
    unsigned offA[2], offB[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int sw = ((4 * par) + (lrow >> 1)) & 7;
        offA[par] = (unsigned)(lrow * p.lda * 2 + 16 * ((lane & 7) ^ sw));
        offB[par] = (unsigned)(STR * lrow * p.ldb * 2 + 16 * ((lane & 7) ^ sw));
    }
    const unsigned pieceA = (unsigned)(8 * p.lda * 2);    // bytes from one piece of A to the next
    // first weight row of piece q' = 4 wave + q (LDS rows 8 q' ..: h = q' >> 4, j = (q' >> 1) & 7, l = 8 (q' & 1) + lrow), in bytes
    unsigned baseB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int qq = 4 * wave + q, h = qq >> 4, j = (qq >> 1) & 7, half = qq & 1;
        const int row = OUT_BF16 ? 128 * h + 64 * half + j : 128 * h + 64 * (j >> 2) + 32 * half + (j & 3);
        baseB[q] = (unsigned)(row * p.ldb * 2);
    }
    int t = blockIdx.x, tm, n0, tm_next = 0, n0_next = 0;
    int64_t m0, m0_next = 0;
    locate(t, m0, n0, tm);
    make_src(m0, n0);
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)lds;   // LDS byte address of the array
    auto issue = [&](int buf, int kt) {
#ifdef MSN_ABL_BF_NODMA
        return;                            // diagnostic build (tools/microbench/build_ablate.sh): no operand traffic at all
#endif
        unsigned char* dst = lds + buf * STAGE_BYTES + wave * 4096;
        const int koff = kt * (BKS * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t*)(dst + q * 1024), 16, offA[q & 1] + (unsigned)(4 * wave + q) * pieceA, koff, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lptr_t*)(dst + OPER_BYTES + q * 1024), 16, offB[q & 1] + baseB[q], koff, 0, 0);
    };

    // ---- fragment addresses: wave (wm, wn) owns rows 64 wm .. + 63 of A (4 m-tiles), LDS rows 128 wn .. + 127 of B (8 n-tiles)
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;
    // LDS row = base + 16 i + l15: swz(row) = l15 >> 1 for every tile;  chunk(kk) = (4 kk + g) ^ swz
    const unsigned fragA = (unsigned)((wm * 64 + l15) * 128 + 16 * (g ^ (l15 >> 1)));
    const unsigned fragB = (unsigned)(OPER_BYTES + (wn * 128 + l15) * 128 + 16 * (g ^ (l15 >> 1)));

    f32x4 acc[4][8];
    bf16x8 fa[2][4], fb[2][4];

    // set S of B fragments <- n-tiles 4 nh .. 4 nh + 3 at k32 half kk;  set S of A fragments <- all 4 m-tiles at kk
#ifdef MSN_ABL_BF_NOFRAG                   // diagnostic build: no fragment reads (the MFMAs multiply whatever the registers hold)
    for (int s_ = 0; s_ < 2; ++s_)
        for (int i = 0; i < 4; ++i) { asm volatile("" : "=v"(fa[s_][i])); asm volatile("" : "=v"(fb[s_][i])); }
#define REQ_B(S, buf, kk, nh)
#define REQ_A(S, buf, kk)
#else
#define REQ_B(S, buf, kk, nh)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                  \
        ds_read128(fb[S][j], lds0 + (buf) * STAGE_BYTES + (fragB ^ ((kk) << 6)) + (4 * (nh) + j) * 2048);
#define REQ_A(S, buf, kk)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
        ds_read128(fa[S][i], lds0 + (buf) * STAGE_BYTES + (fragA ^ ((kk) << 6)) + i * 2048);
#endif
    // D tile = A_frag . B_frag^T: lane (l15, g) gets column index l15 of n-tile j and rows 4 g .. 4 g + 3 of m-tile i
#define MULT(SA, SB, nh)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)                    \
        acc[i][4 * (nh) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[SA][i], fb[SB][j], acc[i][4 * (nh) + j], 0, 0, 0);
#define WAIT_LGKM(n)                                           \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0);

    // store instructions of a FULL tile's epilogue per wave (what the first K-tile behind it may leave in flight)
    constexpr int kEpiStores = (EPI == EPI_B_GELU || !OUT_BF16) ? 32 : 16;

    auto store_tile = [&](int64_t m0, int n0, int tm) {
        // ---- epilogue: lane (l15, g) holds rows m = 16 i + 4 g + r (register r of m-tile i) and, of n-tile j, one column (above):
        // eight consecutive columns cb .. cb + 7 (bf16 output; acc[i][j][r] = column cb + j) or two runs of four, cf(hh) .. + 3
        // (fp32 output; acc[i][4 hh + jj][r] = column cf(hh) + jj).  One m-tile at a time.  Whatever the epilogue READS (saved
        // pre-activation, residual) is requested BEFORE the m-tile's stores are issued -- the next m-tile's pieces between this
        // one's arithmetic and its stores, into the same registers: vector-memory operations complete in order (one counter for
        // loads and stores), and a load issued behind a store waited for it (GELU' launch 738 -> 678 us with the loads first).
        // Every access goes through a buffer descriptor of the TILE's rows (scalar registers; rows past the matrix are out of its
        // range: stores dropped, loads zero) + one 32-bit lane offset + a scalar row offset -- sixteen 64-bit row pointers per lane
        // were 32 registers the K loop's 128 accumulators do not leave.
        // (lane constants and row strides of the epilogue from OPAQUE copies of the thread index / the strides: computed at kernel start
        //  they stay alive across the K loop -- three vector registers and 32 scalar row offsets the allocator then spills)
        int tid_e = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int l15 = tid_e & 15, g = (tid_e >> 4) & 3, wave_e = __builtin_amdgcn_readfirstlane(tid_e >> 6), wm = wave_e >> 1, wn = wave_e & 1;
        constexpr int ES = OUT_BF16 ? 2 : 4;                       // bytes per element of C
        constexpr int XS = EPI == EPI_B_ADD ? 4 : 2;               // ... of aux
        constexpr unsigned OOB = 0x40000000u;                      // beyond any tile (256 rows x ld < 2^20 elements: host)
        const int cb = n0 + wn * 128 + 8 * l15;
        auto cf = [&](int hh) { return n0 + wn * 128 + 64 * hh + 4 * l15; };
        // column of acc[.][j] and whether its group of four columns lies inside the matrix (N % 4 == 0: host)
        auto ncol4 = [&](int q4) { return OUT_BF16 ? cb + 4 * q4 : cf(q4); };      // group q4 = tiles 4 q4 .. 4 q4 + 3
        const bool ok4[2] = {ncol4(0) + 3 < p.N, ncol4(1) + 3 < p.N};
        [[maybe_unused]] const bool rows16_c = p.ldc % 8 == 0, rows16_x = p.ldaux % 8 == 0;   // bf16 rows start on 16 bytes
        const int rows_here = (int)std::min<int64_t>(std::max<int64_t>(p.M - m0, 0), BT);
        auto tile_rsrc = [&](const void* base, int64_t ld, int es) {
            const uint64_t a0 = reinterpret_cast<uint64_t>(base) + (uint64_t)((m0 * ld + n0) * es);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a0), hi = __builtin_amdgcn_readfirstlane((unsigned)(a0 >> 32));
            const int nb = __builtin_amdgcn_readfirstlane((int)(((int64_t)rows_here * ld - n0) * es));
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, nb, 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rsC = tile_rsrc(p.C, p.ldc, ES);
        [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsX = tile_rsrc(p.aux ? p.aux : p.C, p.aux ? p.ldaux : p.ldc, XS);
        // lane offsets of the two column groups in row 64 wm + 4 g of the tile (bytes); the row of (i, r) goes into the scalar offset
        const int lrow0 = wm * 64 + 4 * g;
        unsigned vc[2], vx[2];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
            vc[q4] = ok4[q4] ? (unsigned)((lrow0 * (int)p.ldc + ncol4(q4) - n0) * ES) : OOB;
            vx[q4] = ok4[q4] ? (unsigned)((lrow0 * (int)p.ldaux + ncol4(q4) - n0) * XS) : OOB;
        }
        int srowC = (int)p.ldc * ES, srowX = (int)p.ldaux * XS;
        asm volatile("" : "+s"(srowC), "+s"(srowX));
        const int rows_left = rows_here - lrow0;                   // rows 16 i + r < rows_left of this lane are inside the matrix
        float4 bias4[2];
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4)
            bias4[q4] = (p.bias && ok4[q4]) ? *reinterpret_cast<const float4*>(p.bias + ncol4(q4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x4 cs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // column sums over this lane's rows (bias gradient)
        constexpr bool READS = EPI == EPI_B_GELU_BWD || EPI == EPI_B_ADD;
        using AuxT = std::conditional_t<EPI == EPI_B_ADD, u32x4, u32x2>;
        AuxT ax[READS ? 4 : 1][2];                                  // [r][group of four columns]
        auto request = [&](int i) {
            if constexpr (READS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int so = (16 * i + r) * srowX;
                    if constexpr (EPI == EPI_B_ADD) {
#pragma unroll
                        for (int q4 = 0; q4 < 2; ++q4) ax[r][q4] = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)vx[q4], so, 0);
                    } else {
#ifndef MSN_ABL_BF_NOAUX
                        if (OUT_BF16 && rows16_x && ok4[1]) {
                            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)vx[0], so, 0);
                            ax[r][0] = u32x2{v[0], v[1]}, ax[r][1] = u32x2{v[2], v[3]};
                        } else {
#pragma unroll
                            for (int q4 = 0; q4 < 2; ++q4) ax[r][q4] = __builtin_amdgcn_raw_buffer_load_b64(rsX, (int)vx[q4], so, 0);
                        }
#else
#pragma unroll
                        for (int q4 = 0; q4 < 2; ++q4) ax[r][q4] = u32x2{0x3f803f80u + (unsigned)r, 0x3f803f80u};
#endif
                    }
                }
            }
        };
        // (one v_cvt_pk_bf16_f32 per pair: the vector conversion; two scalar casts + a merge were three instructions)
        auto pack2 = [](float a, float b) {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t ab = {a, b};
            return __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bf16x2_t));
        };
        // eight bf16 of one row: one 16-byte store, or 8 bytes per in-range group of four
        auto put_bf16 = [&](__amdgpu_buffer_rsrc_t rs, const unsigned (&v)[2], int so, const unsigned (&o)[4], bool rows16) {
            if (rows16 && ok4[1]) {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{o[0], o[1], o[2], o[3]}, rs, (int)v[0], so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{o[0], o[1]}, rs, (int)v[0], so, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{o[2], o[3]}, rs, (int)v[1], so, 0);
            }
        };
        request(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // arithmetic, in place (acc[i][4 q4 + jj][r]: row r, column ncol4(q4) + jj)
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
                const float bj[4] = {bias4[q4].x, bias4[q4].y, bias4[q4].z, bias4[q4].w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    f32x4 v = acc[i][4 * q4 + jj] + bj[jj];
                    if constexpr (EPI == EPI_B_GELU_BWD) {         // x gelu'(saved bf16 pre-activation)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned w = ax[r][q4][jj >> 1];
                            const float pre = __uint_as_float((jj & 1) ? (w & 0xffff0000u) : (w << 16));
#ifndef MSN_ABL_BF_NOGELU
                            v[r] *= gelu_grad_fast(pre);
#else
                            v[r] *= pre;
#endif
                        }
                    } else if constexpr (EPI == EPI_B_ADD) {       // + fp32 residual
                        v += f32x4{__uint_as_float(ax[0][q4][jj]), __uint_as_float(ax[1][q4][jj]), __uint_as_float(ax[2][q4][jj]),
                                   __uint_as_float(ax[3][q4][jj])};
                    }
                    acc[i][4 * q4 + jj] = v;
                }
            }
            if (i + 1 < 4) request(i + 1);
            // stores: row r of the m-tile per instruction -- 4 rows (g) x 256 contiguous bytes
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int soC = (16 * i + r) * srowC;
                if constexpr (EPI == EPI_B_GELU) {                 // aux <- pre-activation (bf16), C <- gelu(pre)
                    unsigned o[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) o[x] = pack2(acc[i][2 * x][r], acc[i][2 * x + 1][r]);
#ifndef MSN_ABL_BF_NOAUX               // diagnostic builds (tools/microbench/build_ablate.sh BF_NOAUX / BF_NOGELU / BF_NOCST): timing only
                    put_bf16(rsX, vx, (16 * i + r) * srowX, o, rows16_x);
#endif
#ifndef MSN_ABL_BF_NOGELU
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j][r] = gelu_fast(acc[i][j][r]);
#endif
                }
                if constexpr (OUT_BF16) {
                    unsigned o[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) o[x] = pack2(acc[i][2 * x][r], acc[i][2 * x + 1][r]);
#ifdef MSN_ABL_BF_NOCST
                    if (o[0] == 0x7fc17fc1u)
#endif
                    put_bf16(rsC, vc, soC, o, rows16_c);
                    if (p.colpart && 16 * i + r < rows_left) {     // sums of the values AS STORED (what the next products read)
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            cs[x >> 1][2 * (x & 1)] += __uint_as_float(o[x] << 16);
                            cs[x >> 1][2 * (x & 1) + 1] += __uint_as_float(o[x] & 0xffff0000u);
                        }
                    }
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        const f32x4 v = {acc[i][4 * q4][r], acc[i][4 * q4 + 1][r], acc[i][4 * q4 + 2][r], acc[i][4 * q4 + 3][r]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)vc[q4], soC, 0);
                        if (p.colpart && 16 * i + r < rows_left) cs[q4] += v;
                    }
                }
            }
        }
        if (p.colpart) {   // the four lanes sharing l15 hold the wave's 64 rows: two xor steps, then the lanes of g == 0 write their columns
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    float t = cs[q4][x];
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    cs[q4][x] = t;
                }
            if (g == 0) {
                float* row = p.colpart + (int64_t)(4 * tm + wm) * p.N;
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4)
                    if (ok4[q4]) *reinterpret_cast<f32x4*>(row + ncol4(q4)) = cs[q4];
            }
        }
    };

    const int nkt = p.K / BKS;             // >= 2 whenever gridDim.x < total (host)
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nkt > 1) issue(1, 1);
    unsigned gs = 0;                       // K-tiles this workgroup has started, over all its tiles: K-tile gs sits in buffer gs & 1
    bool behind_full_tile = false;         // the previous tile's epilogue issued exactly kEpiStores store instructions per wave
    for (;;) {
        const int t_next = t + (int)gridDim.x;
        const bool more = t_next < total;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        REQ_B(0, gs & 1, 0, 0)
        REQ_A(0, gs & 1, 0)
        for (int kt = 0; kt < nkt; ++kt, ++gs) {
            const int cur = gs & 1;
            // phase 0: k32 half 0, n-tiles 0-3
            REQ_B(1, cur, 0, 1)
            WAIT_LGKM(4)
            MULT(0, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            // phase 1: k32 half 0, n-tiles 4-7
            REQ_B(0, cur, 1, 0)
            REQ_A(1, cur, 1)
            WAIT_LGKM(8)
            MULT(0, 1, 1)
            __builtin_amdgcn_sched_barrier(0);
            // phase 2: k32 half 1, n-tiles 0-3
            REQ_B(1, cur, 1, 1)
            WAIT_LGKM(4)
            MULT(1, 0, 0)
            __builtin_amdgcn_sched_barrier(0);
            // phase 3: every fragment of this K-tile is in registers -> the buffer may be refilled; the next K-tile of the
            // stream (issued one K-tile ago; the next tile's first one at the end of a tile) must have landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // (vector-memory operations complete IN ORDER, loads and stores on one counter: K-tile 1 of a tile was requested BEFORE
            //  the previous tile's epilogue issued its stores, so behind a FULL tile's epilogue -- a fixed number of store instructions
            //  per wave -- the first K-tile waits for everything but those stores instead of draining them; they have until K-tile 2,
            //  requested behind them, is waited for)
            if (kt == 0 && behind_full_tile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEpiStores) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // (The 8 LDS-DMA instructions of a wave are what the loop pays most for -- profiles/r03_bgemm_kloop_ablation.txt: without
            // them it runs 40 % faster, with every panel an L2 hit 15 %.  Letting the two waves of a SIMD take turns -- one issues
            // here, the other behind its MULT -- gains 5 % on long K in isolation and nothing in the training step; issuing later
            // still loses: the pieces land too late for the next barrier.  Build with -DMSN_BGEMM_STAGGER=1 to try.)
            auto refill = [&]() {
                if (kt + 2 < nkt) issue(cur, kt + 2);
                else if (more) {                               // the ring runs on into the next tile
                    if (kt + 2 == nkt) {                       // (this tile's last K-tile was requested one K-tile ago)
                        locate(t_next, m0_next, n0_next, tm_next);
                        make_src(m0_next, n0_next);
                    }
                    issue(cur, kt + 2 - nkt);
                }
            };
            if (wave < 4 || !g_stagger) refill();
            if (kt + 1 < nkt) {
                REQ_B(0, cur ^ 1, 0, 0)
                REQ_A(0, cur ^ 1, 0)
            }
            __builtin_amdgcn_sched_barrier(0);
            MULT(1, 1, 1)
            __builtin_amdgcn_sched_barrier(0);
            if (wave >= 4 && g_stagger) refill();
            __builtin_amdgcn_sched_barrier(0);
        }
        store_tile(m0, n0, tm);
        behind_full_tile = m0 + BT <= p.M && n0 + BT <= p.N;
        if (!more) break;
        t = t_next, m0 = m0_next, n0 = n0_next, tm = tm_next;
    }
#undef REQ_A
#undef REQ_B
#undef MULT
}

// ---------------------------------------------------------------------------------------------------------------
// TN: both operands are [m][cols] with the reduction index m on the ROWS.  LDS image of one operand K-step =
// [64 rows of m][512 B] (256 bf16 of n or k); the MFMA wants, per lane, 8 consecutive m of one column: two
// ds_read_b64_tr_b16 (each delivers a 4-row x 16-column block column-major to a 16-lane group; lane 4q + p of the group
// supplies the address of row q, columns 4p .. 4p+3; lane i receives column i).  Banking is per 32-lane half = two
// groups = rows {8 g2 + 4 h + q : g2 in 0..1, q in 0..3} x 32 bytes: the 32-byte chunk index (low three bits) of row r
// is XORed with tsw(r) = (r & 3) | ((r >> 3) & 1) << 2, which is distinct over those eight rows -> conflict-free.
__device__ __forceinline__ int tsw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ds_read_tr(bf16x4& dst, unsigned addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}

__global__ __launch_bounds__(512, 2) void bgemm_tn_kernel(const BgemmArgs p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // static priority for the second-dispatched half of the workgroup: waves 4-7 lose the per-SIMD issue arbitration (priority, then
    // age) on every phase otherwise (MI355X_MICROARCH.md, two waves per SIMD); cfg5 step -0.4 % in three interleaved pairs.  (The plane
    // GEMMs of pgemm_kernels.h: +0.3 % with the same line -- not there.)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    const int tiles = p.tiles_m * p.tiles_n;               // tiles over (N, K)
    // workgroup ids go round-robin to the 8 XCDs: give each XCD a contiguous run of (split, tile) pairs, i.e. the tiles of
    // one reduction range, so that its dY / X panels are shared in ONE L2 instead of being fetched by all eight
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, x = bid % 8, i = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int split = bid / tiles, tile = bid % tiles;
    const int tn = tile / p.tiles_n, tk = tile % p.tiles_n;
    const int n0 = tn * BT, k0 = tk * BT;
    const int64_t m_begin = (int64_t)split * p.rows_per_split;
    const int64_t m_end = std::min<int64_t>(p.M, m_begin + p.rows_per_split);
    const int nkt = (int)((m_end - m_begin + BKS - 1) / BKS);

    // ---- LDS-DMA: a piece = 2 rows x 512 B; wave w moves pieces 4w .. 4w+3 (rows 8w .. 8w+7) of both operands.
    // lane i of a piece: row i / 32, 16-byte slot s = i % 32 <- source chunk c = (s >> 1) with its low 3 bits ^ tsw(row)
    // Sources through BUFFER descriptors (round 5): one per operand in scalar registers, based at this split's first row and
    // covering exactly its rows -- a row of the K-step past the reduction range lies beyond the descriptor and reads zeros (the
    // older form chose between the row and a zero page per piece: a compare, an exec mask and a 64-bit address per instruction,
    // 24 vector instructions per K-step which the matrix pipe does not overlap) -- the lane's byte offset inside a K-step in one
    // register per piece, the K-step as the scalar offset.
    const int rows_here = (int)(m_end - m_begin);
    auto uniform_rsrc = [&](const u16* base, int64_t bytes) -> __amdgpu_buffer_rsrc_t {
        const uint64_t a = reinterpret_cast<uint64_t>(base);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const int nb = __builtin_amdgcn_readfirstlane((int)std::min<int64_t>(bytes, 0x7fffffff));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, nb, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc(p.A + m_begin * p.lda, 2 * ((int64_t)(rows_here - 1) * p.lda + p.N));
    const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc(p.B + m_begin * p.ldb, 2 * ((int64_t)(rows_here - 1) * p.ldb + p.K));
    unsigned offA[4], offB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int R = 2 * (4 * wave + q) + (lane >> 5);
        const int s = lane & 31;
        const int c = (((s >> 1) & 7) ^ tsw(R)) | ((s >> 1) & 8);
        const int col = 16 * c + 8 * (s & 1);                     // element column inside the 256-wide tile
        // columns past N / K: clamped (their products are never stored)
        offA[q] = (unsigned)(2 * ((int64_t)R * p.lda + std::min(n0 + col, p.N - 8)));
        offB[q] = (unsigned)(2 * ((int64_t)R * p.ldb + std::min(k0 + col, p.K - 8)));
    }
    const int kstepA = (int)(2 * BKS * p.lda), kstepB = (int)(2 * BKS * p.ldb);      // bytes from one K-step to the next
    auto issue = [&](int buf, int kt) {
        unsigned char* dst = lds + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t*)(dst + q * 1024), 16, (int)offA[q], kt * kstepA, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t*)(dst + OPER_BYTES + q * 1024), 16, (int)offB[q], kt * kstepB, 0, 0);
    };

    // ---- fragments.  Output tile rows = n (operand A = dY), columns = k (operand B = X).  Wave (wr, wc): n in
    // [128 wr, +128) = 8 tiles, k in [64 wc, +64) = 4 tiles.  Swapped MFMA operands again: D[k][n] -> lane holds
    // n = l15 and k = 4 g .. 4 g + 3 ... see the epilogue.
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
    // a tr read: lane supplies row (32 kk + 8 g + 4 h + tq), byte (2 * col0 + 8 tp) of that row, chunk-swizzled.
    // tsw(row) depends on (tq, g & 1) only: (tq & 3) | ((8 g + 4 h + tq) >> 3 & 1) << 2 = tq | (g & 1) << 2
    const int sw = tq | ((g & 1) << 2);
    auto tr_addr = [&](int oper, int col0, int kk, int h) -> unsigned {
        const int row = 32 * kk + 8 * g + 4 * h + tq;
        const int byte = 2 * col0 + 8 * tp;                        // col0 % 16 == 0 -> 32-byte chunk index = byte >> 5
        const int chunk = byte >> 5;
        const int sbyte = (((chunk & 7) ^ sw) | (chunk & 8)) * 32 + (byte & 31);
        return (unsigned)(oper * OPER_BYTES + row * 512 + sbyte);
    };
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)lds;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x4 fa[2][4][2], fb[2][4][2];     // [set][tile][h]: elements m = 8 g + 4 h .. + 3 of the k32 block

#define TREQ_A(S, buf, kk, nh)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int h = 0; h < 2; ++h)                    \
        ds_read_tr(fa[S][i][h], lds0 + (buf) * STAGE_BYTES + tr_addr(0, wr * 128 + 16 * (4 * (nh) + i), kk, h));
#define TREQ_B(S, buf, kk)                                                                                         \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int h = 0; h < 2; ++h)                    \
        ds_read_tr(fb[S][j][h], lds0 + (buf) * STAGE_BYTES + tr_addr(1, wc * 64 + 16 * j, kk, h));
    // D = X_frag^T-as-A . dY_frag-as-B: rows of D = k, columns = n: lane holds n = l15, k = 4 g .. 4 g + 3
#define TMULT(SA, SB, nh)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                  \
        const bf16x8 av = __builtin_shufflevector(fa[SA][i][0], fa[SA][i][1], 0, 1, 2, 3, 4, 5, 6, 7);             \
        const bf16x8 bv = __builtin_shufflevector(fb[SB][j][0], fb[SB][j][1], 0, 1, 2, 3, 4, 5, 6, 7);             \
        acc[4 * (nh) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[4 * (nh) + i][j], 0, 0, 0);     \
    }

    if (nkt > 0) {
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (nkt > 1) issue(1, 1);
        TREQ_A(0, 0, 0, 0)
        TREQ_B(0, 0, 0)
    }
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        // (lgkmcnt counts at most 15 outstanding: never more than 8 younger reads behind the ones waited for)
        TREQ_A(1, cur, 0, 1)
        WAIT_LGKM(8)
        TMULT(0, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        TREQ_A(0, cur, 1, 0)
        WAIT_LGKM(8)
        TMULT(1, 0, 1)
        __builtin_amdgcn_sched_barrier(0);
        TREQ_B(1, cur, 1)
        TREQ_A(1, cur, 1, 1)
        WAIT_LGKM(8)
        TMULT(0, 1, 0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nkt) issue(cur, kt + 2);
        if (kt + 1 < nkt) {
            TREQ_A(0, cur ^ 1, 0, 0)
            TREQ_B(0, cur ^ 1, 0)
        }
        __builtin_amdgcn_sched_barrier(0);
        TMULT(1, 1, 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef TREQ_A
#undef TREQ_B
#undef TMULT
#undef WAIT_LGKM

    // ---- epilogue: C[n][k] (or this split's slab): lane holds n = n0 + wr 128 + 16 i + l15, k = k0 + wc 64 + 16 j + 4 g ..
    float* out = p.splits > 1 ? p.slabs + (int64_t)split * p.N * p.K : static_cast<float*>(p.C);
    const int64_t ldo = p.splits > 1 ? p.K : p.ldc;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int n = n0 + wr * 128 + 16 * i + l15;
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wc * 64 + 16 * j + 4 * g;
            if (k + 3 >= p.K) continue;
            *reinterpret_cast<float4*>(out + (int64_t)n * ldo + k) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// C[i] = sum_s slab[s][i] in split order (float4 per thread, eight independent loads per wait)
__global__ void bgemm_slab_sum_kernel(const float* __restrict__ slabs, int splits, int64_t n4, int K4, int64_t ldc4,
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

// ------------------------------------------------------------------------------------------------- casts, sums
// y = bf16(x), 8 elements per thread
__global__ void cast_bf16_kernel(const float* __restrict__ x, int64_t n8, u16* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        uint4 o;
        o.x = f2bf(a.x) | ((unsigned)f2bf(a.y) << 16); o.y = f2bf(a.z) | ((unsigned)f2bf(a.w) << 16);
        o.z = f2bf(b.x) | ((unsigned)f2bf(b.y) << 16); o.w = f2bf(b.z) | ((unsigned)f2bf(b.w) << 16);
        reinterpret_cast<uint4*>(y)[i] = o;
    }
}

// y[c][r] = bf16(x[r][c]) through a 32 x 33 LDS tile (weights: a few MB per step)
__global__ __launch_bounds__(256) void cast_bf16_t_kernel(const float* __restrict__ x, int R, int C, u16* __restrict__ y) {
    __shared__ float t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        t[i][tx] = (r0 + i < R && c0 + tx < C) ? x[(int64_t)(r0 + i) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) y[(int64_t)(c0 + i) * R + r0 + tx] = f2bf(t[tx][i]);
}

// A LIST of weight matrices rounded to bf16 in ONE launch (msn_cast_bf16_list): the bf16-resident trunk rounds four weights per block in
// its forward and makes four transposed copies in its backward -- 94 launches of 6 - 10 us per step one by one.  Workgroup -> (entry,
// 32 x 32 tile) through the entries' first-workgroup table (kernel argument); plain entries
// copy row-wise through the same tile shape, transposed ones go through the 32 x 33 LDS tile of cast_bf16_t_kernel.
constexpr int CAST_LIST_MAX = 64;
struct CastEntry {
    const float* x; u16* y;
    int R, C, transposed, first, tiles_x;
};
struct CastTable {
    int n;
    CastEntry e[CAST_LIST_MAX];
};
__global__ __launch_bounds__(256) void cast_bf16_list_kernel(const CastTable tb) {
    __shared__ float t[32][33];
    int k = 0;
    for (int j = 1; j < tb.n; ++j) k = (int)blockIdx.x >= tb.e[j].first ? j : k;          // (uniform; at most 64 entries)
    const CastEntry e = tb.e[k];
    const int tile = (int)blockIdx.x - e.first;
    const int r0 = (tile / e.tiles_x) * 32, c0 = (tile % e.tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (e.transposed) {
        for (int i = ty; i < 32; i += 8)
            t[i][tx] = (r0 + i < e.R && c0 + tx < e.C) ? e.x[(int64_t)(r0 + i) * e.C + c0 + tx] : 0.f;
        __syncthreads();
        for (int i = ty; i < 32; i += 8)
            if (c0 + i < e.C && r0 + tx < e.R) e.y[(int64_t)(c0 + i) * e.R + r0 + tx] = f2bf(t[tx][i]);
    } else {
        for (int i = ty; i < 32; i += 8)
            if (r0 + i < e.R && c0 + tx < e.C) e.y[(int64_t)(r0 + i) * e.C + c0 + tx] = f2bf(e.x[(int64_t)(r0 + i) * e.C + c0 + tx]);
    }
}

// out[n] = sum_m X[m][n] for bf16 X: block partials of 8 columns per thread, then one fixed-order pass
__global__ __launch_bounds__(256) void bcolsum_part_kernel(const u16* __restrict__ x, int64_t ld, int64_t M, int N8,
                                                           int rows_per_block, float* __restrict__ part) {
    const int c8 = blockIdx.x * 32 + (threadIdx.x & 31);          // group of 8 columns
    const int ry = threadIdx.x >> 5;                               // 8 row lanes
    __shared__ float red[8][32][8];
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c8 < N8) {
        const int64_t r_begin = (int64_t)blockIdx.y * rows_per_block;
        const int64_t r_end = std::min<int64_t>(M, r_begin + rows_per_block);
        for (int64_t r = r_begin + ry; r < r_end; r += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + r * ld + 8 * c8);
            s[0] += bf2f((u16)(v.x & 0xffff)); s[1] += bf2f((u16)(v.x >> 16));
            s[2] += bf2f((u16)(v.y & 0xffff)); s[3] += bf2f((u16)(v.y >> 16));
            s[4] += bf2f((u16)(v.z & 0xffff)); s[5] += bf2f((u16)(v.z >> 16));
            s[6] += bf2f((u16)(v.w & 0xffff)); s[7] += bf2f((u16)(v.w >> 16));
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[ry][threadIdx.x & 31][k] = s[k];
    __syncthreads();
    if (ry == 0 && c8 < N8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = 0.f;
#pragma unroll
            for (int y = 0; y < 8; ++y) t += red[y][threadIdx.x & 31][k];
            part[(int64_t)blockIdx.y * (8 * N8) + 8 * c8 + k] = t;
        }
    }
}
// out[n] = sum_k part[k][n]: 64 columns x 16 row groups per workgroup (group rg sums the parts k = rg, rg + 16, ... eight independent
// loads per wait), the sixteen group sums combined through LDS in a fixed order.  (Four groups of 256 threads took 30 us for the
// 1 576 x 3 072 partials of a GELU' launch -- 50 dependent round trips per thread on 48 workgroups.)
__global__ __launch_bounds__(1024) void bcolsum_finish_kernel(const float* __restrict__ part, int nparts, int N,
                                                              float* __restrict__ out) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (n < N) {
        const int mine = (nparts - rg + 15) / 16;               // parts this group owns
        for (int k0 = 0; k0 < mine; k0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(int64_t)(rg + 16 * std::min(k0 + j, mine - 1)) * N + n];
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

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int tn_splits(int tiles, int64_t M) {
    // enough workgroups for the 256 CUs (one 128-KB workgroup per CU), each split a whole number of K-steps
    int s = std::max(1, 256 / tiles);
    const int64_t steps = cdiv(M, BKS);
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, steps));
}

}  // namespace msn

using namespace msn;

extern "C" size_t msn_bgemm_nt_colsum_workspace_bytes(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return sizeof(float) * 4 * (size_t)cdiv(M, BT) * (size_t)N;
}

extern "C" int msn_bgemm_nt(int64_t M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                            int64_t ldc, int c_bf16, const float* bias, int epilogue, void* aux, int64_t ldaux,
                            float* colsum_out, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "msn_bgemm_nt: empty operand");
    MSN_REQUIRE(K % BKS == 0 && N % 4 == 0, "msn_bgemm_nt: K = %d must be a multiple of 64 and N = %d of 4", K, N);
    MSN_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B),
                "msn_bgemm_nt: operand rows must be 16-byte aligned (lda %lld, ldb %lld)", (long long)lda, (long long)ldb);
    MSN_REQUIRE(ldc >= N && ldc % 4 == 0 && aligned16(C) && (!bias || aligned16(bias)), "msn_bgemm_nt: bad output / bias");
    // (the epilogue addresses a tile's 256 rows through one buffer descriptor and 32-bit offsets below 2^30)
    MSN_REQUIRE(ldc < (1 << 20) && ldaux < (1 << 20), "msn_bgemm_nt: ldc = %lld / ldaux = %lld must be below 2^20 elements", (long long)ldc,
                (long long)ldaux);
    MSN_REQUIRE(epilogue >= EPI_B_NONE && epilogue <= EPI_B_ADD, "msn_bgemm_nt: unknown epilogue %d", epilogue);
    MSN_REQUIRE(epilogue == EPI_B_NONE || (aux && ldaux >= N && ldaux % 4 == 0 && aligned16(aux)),
                "msn_bgemm_nt: epilogue %d needs an aux matrix", epilogue);
    MSN_REQUIRE(!(epilogue == EPI_B_GELU && !c_bf16) && !(epilogue == EPI_B_ADD && c_bf16),
                "msn_bgemm_nt: GELU writes bf16 (activation + saved pre-activation), ADD writes the fp32 residual stream");
    BgemmArgs a = {};
    a.A = static_cast<const u16*>(A); a.B = static_cast<const u16*>(B); a.C = C; a.aux = aux; a.bias = bias;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldaux = ldaux; a.M = M; a.N = N; a.K = K;
    a.tiles_m = (int)cdiv(M, BT); a.tiles_n = (int)cdiv(N, BT);
    a.super_rows = std::max(1, std::min(8, 4096 / K));
    a.colpart = nullptr;
    if (colsum_out) {
        const size_t need = msn_bgemm_nt_colsum_workspace_bytes(M, N);
        MSN_REQUIRE(ws && ws_bytes >= need && aligned16(ws) && N % 4 == 0, "msn_bgemm_nt: column-sum workspace %zu < %zu bytes", ws_bytes, need);
        a.colpart = static_cast<float*>(ws);
    }
    // one persistent workgroup per CU walking the tiles (the ring of K-tiles runs on across tile boundaries) once there is
    // more than one tile per CU and a tile has at least two K-tiles; otherwise one workgroup per tile
    const int total_tiles = a.tiles_m * a.tiles_n;
    const bool persistent = total_tiles > 256 && K / BKS >= 2;       // (one workgroup per tile at every size: bit-identical, 3 - 6 % slower on K = 768; r02)
    const dim3 grid((unsigned)(persistent ? 256 : total_tiles)), block(512);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (epilogue == EPI_B_GELU) hipLaunchKernelGGL((bgemm_nt_kernel<EPI_B_GELU, true>), grid, block, 0, st, a);
    else if (epilogue == EPI_B_ADD) hipLaunchKernelGGL((bgemm_nt_kernel<EPI_B_ADD, false>), grid, block, 0, st, a);
    else if (epilogue == EPI_B_GELU_BWD) {
        if (c_bf16) hipLaunchKernelGGL((bgemm_nt_kernel<EPI_B_GELU_BWD, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bgemm_nt_kernel<EPI_B_GELU_BWD, false>), grid, block, 0, st, a);
    } else {
        if (c_bf16) hipLaunchKernelGGL((bgemm_nt_kernel<EPI_B_NONE, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((bgemm_nt_kernel<EPI_B_NONE, false>), grid, block, 0, st, a);
    }
    MSN_LAUNCH_CHECK();
    if (colsum_out) {   // every (row slab, column) partial was written by exactly one wave: fixed-order sum over the slabs
        hipLaunchKernelGGL(bcolsum_finish_kernel, dim3((unsigned)cdiv(N, 64)), dim3(1024), 0, st, a.colpart, 4 * a.tiles_m, N,
                           colsum_out);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}

extern "C" size_t msn_bgemm_tn_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int tiles = (int)(cdiv(N, BT) * cdiv(K, BT));
    const int s = tn_splits(tiles, M);
    return s > 1 ? sizeof(float) * (size_t)s * N * K : 0;
}

extern "C" int msn_bgemm_tn(int64_t M, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, float* C,
                            int64_t ldc, void* ws, size_t ws_bytes, msn_stream_t stream) {
    MSN_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, "msn_bgemm_tn: empty operand");
    MSN_REQUIRE(N % 8 == 0 && K % 8 == 0 && lda >= N && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && aligned16(A) && aligned16(B),
                "msn_bgemm_tn: N, K and both leading dimensions must be multiples of 8, operands 16-byte aligned");
    MSN_REQUIRE(ldc >= K && ldc % 4 == 0 && aligned16(C), "msn_bgemm_tn: bad output");
    BgemmArgs a = {};
    a.A = static_cast<const u16*>(A); a.B = static_cast<const u16*>(B); a.C = C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.tiles_m = (int)cdiv(N, BT); a.tiles_n = (int)cdiv(K, BT);
    const int tiles = a.tiles_m * a.tiles_n;
    a.splits = tn_splits(tiles, M);
    a.rows_per_split = cdiv(cdiv(M, a.splits), BKS) * BKS;
    a.splits = (int)cdiv(M, a.rows_per_split);
    // (the kernel addresses a split's rows through buffer descriptors with 32-bit byte offsets)
    MSN_REQUIRE((a.rows_per_split + BKS) * std::max(lda, ldb) * 2 < (1ll << 31),
                "msn_bgemm_tn: a reduction split spans 2 GB of an operand (%lld rows x %lld elements)", (long long)a.rows_per_split,
                (long long)std::max(lda, ldb));
    const size_t need = a.splits > 1 ? sizeof(float) * (size_t)a.splits * N * K : 0;
    MSN_REQUIRE(need == 0 || (ws && ws_bytes >= need && aligned16(ws)), "msn_bgemm_tn: workspace %zu < %zu bytes", ws_bytes, need);
    a.slabs = static_cast<float*>(ws);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bgemm_tn_kernel, dim3((unsigned)(tiles * a.splits)), dim3(512), 0, st, a);
    MSN_LAUNCH_CHECK();
    if (a.splits > 1) {
        const int64_t n4 = (int64_t)N * K / 4;
        hipLaunchKernelGGL(bgemm_slab_sum_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4, 256), 2048)), dim3(256), 0, st,
                           a.slabs, a.splits, n4, K / 4, ldc / 4, C);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}

extern "C" int msn_cast_bf16(const float* x, int64_t n, void* y, msn_stream_t stream) {
    MSN_REQUIRE(x && y && n > 0 && n % 8 == 0 && aligned16(x) && aligned16(y), "msn_cast_bf16: n must be a multiple of 8, 16-byte aligned");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n / 8, 256), 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n / 8, static_cast<u16*>(y));
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_cast_bf16_transposed(const float* x, int R, int C, void* y, msn_stream_t stream) {
    MSN_REQUIRE(x && y && R > 0 && C > 0, "msn_cast_bf16_transposed: bad arguments");
    hipLaunchKernelGGL(cast_bf16_t_kernel, dim3((unsigned)cdiv(C, 32), (unsigned)cdiv(R, 32)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, R, C, static_cast<u16*>(y));
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}

extern "C" int msn_cast_bf16_list(int n, const msn_cast_item* items, msn_stream_t stream) {
    MSN_REQUIRE(items && n > 0, "msn_cast_bf16_list: empty list");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int base = 0; base < n; base += CAST_LIST_MAX) {
        CastTable tb = {};
        tb.n = std::min(n - base, CAST_LIST_MAX);
        int64_t wgs = 0;
        for (int k = 0; k < tb.n; ++k) {
            const msn_cast_item& it = items[base + k];
            MSN_REQUIRE(it.x && it.y && it.R > 0 && it.C > 0 && it.R < (1ll << 31) && it.C < (1ll << 31), "msn_cast_bf16_list: bad item %d", base + k);
            CastEntry& e = tb.e[k];
            e.x = it.x, e.y = static_cast<u16*>(it.y), e.R = (int)it.R, e.C = (int)it.C, e.transposed = it.transposed ? 1 : 0;
            e.first = (int)wgs;
            e.tiles_x = (int)cdiv(it.C, 32);
            wgs += cdiv(it.R, 32) * cdiv(it.C, 32);
            MSN_REQUIRE(wgs < (1ll << 31), "msn_cast_bf16_list: list too large");
        }
        hipLaunchKernelGGL(cast_bf16_list_kernel, dim3((unsigned)wgs), dim3(256), 0, st, tb);
        MSN_LAUNCH_CHECK();
    }
    return MSN_OK;
}

extern "C" size_t msn_bcolsum_workspace_bytes(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int parts = (int)std::min<int64_t>(cdiv(M, 64), 256);
    return sizeof(float) * (size_t)parts * N;
}

extern "C" int msn_bcolsum(const void* X, int64_t ldx, int64_t M, int N, float* out, void* ws, size_t ws_bytes,
                           msn_stream_t stream) {
    MSN_REQUIRE(X && out && M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0 && ldx >= N && aligned16(X),
                "msn_bcolsum: N and ldx must be multiples of 8, rows 16-byte aligned");
    const int parts = (int)std::min<int64_t>(cdiv(M, 64), 256);
    const int rows_per_block = (int)cdiv(M, parts);
    MSN_REQUIRE(ws && ws_bytes >= sizeof(float) * (size_t)parts * N, "msn_bcolsum: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(bcolsum_part_kernel, dim3((unsigned)cdiv(N / 8, 32), (unsigned)parts), dim3(256), 0, st,
                       static_cast<const u16*>(X), ldx, M, N / 8, rows_per_block, part);
    MSN_LAUNCH_CHECK();
    hipLaunchKernelGGL(bcolsum_finish_kernel, dim3((unsigned)cdiv(N, 64)), dim3(1024), 0, st, part, parts, N, out);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
