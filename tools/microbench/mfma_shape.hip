// Which bf16 MFMA shape holds the higher clock under load?  Register-only loops, random operands, 128 accumulator registers
// per wave, 1 or 2 waves per SIMD: v_mfma_f32_32x32x16_bf16 (8 accumulators of 16) against v_mfma_f32_16x16x32_bf16 (32 of 4).
// Same flops per instruction-cycle; the question is the clock the chip holds (MI355X_MICROARCH.md, DVFS give-back (7)).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool SMALL>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ rnd, float* out, int iters, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        const uint4 x = rnd[(threadIdx.x * 8 + i) & 4095], y = rnd[(threadIdx.x * 8 + 4 + i) & 4095];
        a[i] = __builtin_bit_cast(bf16x8, x);
        b[i] = __builtin_bit_cast(bf16x8, y);
    }
    float s = 0.f;
    if constexpr (SMALL) {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)      // the same flops per iteration: 16 x 32768 = 32 x 16384
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
template <bool SMALL>
void run(const uint4* rnd, int threads, const char* what) {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
    unsigned long long* clk; hipMalloc(&clk, 16);
    const int iters = 60000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {                        // the last repetition is the warm one
        hipEventRecord(e0);
        k<SMALL><<<256, threads>>>(rnd, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * (threads / 64) * iters * 32.0 * 16384.0;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%s, %d waves per SIMD: %.0f TFLOP/s (%.1f ms), in-kernel clock %.3f GHz, %.1f cycles per 16384-flop MFMA-equivalent per wave\n", what, threads / 256,
           flops / ms / 1e9, ms, (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / (iters * 32.0));
    hipFree(clk); hipFree(out);
}
int main(int argc, char** argv) {
    const bool zeros = argc > 1;
    uint4* h = (uint4*)malloc(4096 * 16);
    srand(1);
    for (int i = 0; i < 4096 * 4; ++i) {
        // random bf16 pairs of moderate magnitude: sign, exponent 120..134, random mantissa
        unsigned lo = ((rand() & 1) << 15) | ((120 + rand() % 15) << 7) | (rand() & 127);
        unsigned hi = ((rand() & 1) << 15) | ((120 + rand() % 15) << 7) | (rand() & 127);
        ((unsigned*)h)[i] = zeros ? 0u : (lo | (hi << 16));
    }
    uint4* d; hipMalloc(&d, 4096 * 16); hipMemcpy(d, h, 4096 * 16, hipMemcpyHostToDevice);
    printf("%s operands\n", zeros ? "zero" : "random");
    run<false>(d, 256, "v_mfma_f32_32x32x16_bf16");
    run<true>(d, 256, "v_mfma_f32_16x16x32_bf16");
    run<false>(d, 512, "v_mfma_f32_32x32x16_bf16");
    run<true>(d, 512, "v_mfma_f32_16x16x32_bf16");
    return 0;
}
