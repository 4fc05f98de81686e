// Issue rate of fp32 vector instructions on gfx950, per SIMD: dependent-free chains of v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32
// and the same interleaved with v_mfma_f32_16x16x32_bf16 from the SAME wave and from a SECOND wave of the SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/valu_rate.hip -o tools/microbench/valu_rate && tools/microbench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int ITER = 4096, CH = 8;
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int waves_mfma) {
    const int wave = threadIdx.x >> 6;
    float a[CH], b[CH];
    f32x2 p[CH];
    f32x4 acc[4] = {};
    bf16x8 fa, fb;
    for (int i = 0; i < 8; ++i) fa[i] = (__bf16)(float)(threadIdx.x + i), fb[i] = (__bf16)1.f;
    for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 1e-3f + i, b[i] = 1.0001f, p[i] = f32x2{a[i], a[i] + 1.f};
    const float c0 = out[0], c1 = out[1];
    const f32x2 c2 = {c0, c0}, c3 = {c1, c1};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool do_mfma = (MODE >= 4) && (MODE == 4 || wave >= (int)(blockDim.x >> 6) - waves_mfma);
    const bool do_valu = MODE < 5 || !do_mfma;
    for (int it = 0; it < ITER; ++it) {
        if (do_valu) {
            if (MODE == 0 || MODE >= 4) {
#pragma unroll
                for (int i = 0; i < CH; ++i) a[i] = __builtin_fmaf(a[i], c0, c1);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < CH; ++i) p[i] = __builtin_elementwise_fma(p[i], c2, c3);
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < CH; ++i) p[i] = p[i] + c3;
            } else if (MODE == 3) {
#pragma unroll
                for (int i = 0; i < CH; ++i) p[i] = p[i] * c2;
            }
        }
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < CH; ++i) s += a[i] + p[i][0] + p[i][1];
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int waves_mfma, float* out, unsigned long long* cyc) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, waves_mfma);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, waves_mfma);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int wps = threads / 256;   // waves per SIMD
    printf("%-58s %d waves/SIMD: %7.1f us, wave 0: %8llu clocks (100 MHz ticks x ?) ; per iteration per SIMD %.2f ns\n", name, wps, ms * 1e3, h[0], ms * 1e6 / ITER);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * (2 + 256 * 512));
    hipMalloc(&cyc, 8 * 256 * 8);
    const float init[2] = {1.0000001f, 1e-9f};
    hipMemcpy(out, init, sizeof(init), hipMemcpyHostToDevice);
    printf("per iteration: 8 vector instructions (and/or 4 MFMA 16x16x32 bf16 = 4 x 16 cycles nominal)\n");
    for (int threads : {256, 512}) {
        run<0>("8 x v_fma_f32", threads, 0, out, cyc);
        run<1>("8 x v_pk_fma_f32", threads, 0, out, cyc);
        run<2>("8 x v_pk_add_f32", threads, 0, out, cyc);
        run<3>("8 x v_pk_mul_f32", threads, 0, out, cyc);
        run<4>("8 x v_fma_f32 + 4 MFMA, same wave", threads, 0, out, cyc);
    }
    run<5>("4 MFMA in wave 4-7, 8 x v_fma_f32 in waves 0-3 (one each per SIMD)", 512, 4, out, cyc);
    run<5>("4 MFMA only (waves 0-3 idle: 256 threads all MFMA)", 256, 4, out, cyc);
    return 0;
}
