// Calibration: sustained rate of v_mfma_f32_32x32x2_f32 with NACC independent accumulators per wave and
// `waves` waves per SIMD, operands in registers (no memory).  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// clk[0..1]: shader-clock cycles (s_memtime) and 100-MHz ticks (s_memrealtime) spent by workgroup 0 -> core clock
template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
template <int NACC>
void run(int threads, int blocks_per_cu) {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    const int iters = 2000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    k<NACC><<<blocks, threads>>>(out, 10, 1.f, 2.f, nullptr);
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, 1.f, 2.f, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * (threads / 64) * iters * 16.0 * NACC * 4096.0;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("NACC=%d threads/block=%d blocks/CU=%d  -> %.1f TFLOP/s (%.3f ms)  core clock %.3f GHz\n", NACC, threads, blocks_per_cu,
           flops / ms / 1e9, ms, (double)h[0] / ((double)h[1] * 10.0));
    hipFree(clk);
    hipFree(out);
}
int main() {
    run<1>(256, 1); run<2>(256, 1); run<4>(256, 1); run<4>(256, 2); run<4>(512, 1); run<2>(512, 1); run<4>(256, 4);
    return 0;
}
