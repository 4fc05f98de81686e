// Calibration: what a workgroup barrier per K-step costs the fp32 matrix pipe.  4 waves per workgroup (one per SIMD), WPC
// workgroups per CU, every wave issues NM v_mfma_f32_32x32x2_f32 (4 independent accumulators) per "K-step", then
//   MODE 0: nothing            MODE 1: s_barrier            MODE 2: s_barrier + 16 ds_read_b128 of a private LDS region
//   MODE 3: as the GEMM: the barrier sits before the last quarter of the step's MFMAs
// Build: hipcc --offload-arch=gfx950 -O3 mfma_barrier.hip -o mfma_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NM>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0, float b0, int lds_pad) {
    extern __shared__ float dyn[];
    __shared__ float4 buf[1024];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    if (MODE == 2) {
        for (int i = threadIdx.x; i < 1024; i += 256) buf[i] = make_float4(a, b, a, b);
        __syncthreads();
    }
    float4 f[4];
    for (int i = 0; i < 4; ++i) f[i] = make_float4(a, b, b, a);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < NM * 3 / 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int u = 0; u < NM / 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
        } else {
            if (MODE == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = buf[(threadIdx.x + 64 * q + it) & 1023];
            }
#pragma unroll
            for (int u = 0; u < NM; ++u)
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(MODE == 2 ? f[u & 3].x : a, MODE == 2 ? f[u & 3].y : b, acc[u & 3], 0, 0, 0);
            if (MODE >= 1) __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (lds_pad < 0 ? dyn[0] : 0.f);
}

template <int MODE, int NM>
void run(int wpc) {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    const int iters = 4000 * 64 / NM, blocks = 256 * wpc;
    // one workgroup per CU: pad the dynamic LDS so that a second one does not fit
    const size_t dyn = wpc == 1 ? 100 * 1024 : 0;
    hipFuncSetAttribute((const void*)k<MODE, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, NM><<<blocks, 256, dyn>>>(out, 10, 1.f, 2.f, 0);
    hipEventRecord(e0);
    k<MODE, NM><<<blocks, 256, dyn>>>(out, iters, 1.f, 2.f, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * (double)NM * 4096.0;
    printf("mode %d  MFMAs per step %3d  workgroups/CU %d -> %.1f TFLOP/s (%.3f ms)\n", MODE, NM, wpc, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    for (int wpc = 1; wpc <= 2; ++wpc) {
        run<0, 64>(wpc);
        run<1, 64>(wpc);
        run<1, 128>(wpc);
        run<1, 32>(wpc);
        run<1, 16>(wpc);
        run<2, 64>(wpc);
        run<3, 64>(wpc);
    }
    return 0;
}
