// The plane-output epilogue of pgemm_nt_kernel as a store pattern: a wave writes, per 32 x 32 tile, SIX 1-KB block images that lie
// side by side in HBM (6 KB), four tiles per wave and 256 x 128 workgroup tile, between idle phases (the K loop).
//   A: one contiguous 1-KB image per instruction (as shipped)      B: four 256-byte quarters of four DIFFERENT images per instruction
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/store_planes.hip -o tools/microbench/store_planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int RB = 66560 / 32, CB = 96;        // row blocks, column blocks (N = 1536) of a three-plane matrix: [RB][CB][3][1 KB]
template <int P>
__global__ __launch_bounds__(512) void k(unsigned char* C, int total, unsigned seed, int spin_ticks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int tm = t / (CB / 8), tn = t % (CB / 8);      // tile = 8 row blocks x 8 column blocks
        if (spin_ticks) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(4);
            __syncthreads();
        }
        const uint4 v = make_uint4(seed + t, lane, wave, 7u);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            unsigned char* grp = C + ((int64_t)(tm * 8 + wm * 2 + i) * CB + tn * 8 + wn * 4 + 2 * j) * 3072;    // 6 KB: two column blocks x 3 planes
            if (P == 0) {
                for (int s = 0; s < 6; ++s) *reinterpret_cast<uint4*>(grp + s * 1024 + lane * 16) = v;
            } else if (P == 1) {
                const int gq = lane >> 4;
                for (int s = 0; s < 6; ++s) {
                    int im = s + gq; im = im >= 6 ? im - 6 : im;
                    *reinterpret_cast<uint4*>(grp + im * 1024 + gq * 256 + (lane & 15) * 16) = v;
                }
            } else if (P == 2) {     // eight 128-byte pieces of eight different ... (6 images: pieces of 128 B, lane >> 3 picks the image offset)
                const int g8 = lane >> 3;
                for (int s = 0; s < 6; ++s) {
                    // piece index within the 6-KB group: 48 pieces of 128 B; instruction s takes pieces {g8 * 6 + s}
                    *reinterpret_cast<uint4*>(grp + (g8 * 6 + s) * 128 + (lane & 7) * 16) = v;
                }
            }
        }
    }
}
template <int P> void run(const char* name, unsigned char* C, int grid, int spin) {
    const int total = (RB / 8) * (CB / 8);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<P>, dim3(grid), dim3(512), 0, 0, C, total, 1u, spin);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<P>, dim3(grid), dim3(512), 0, 0, C, total, 2u + r, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double tiles_cu = (total + grid - 1) / grid;
    printf("%-64s grid %3d spin %4.1f us: %7.1f us  (%.2f us per 196-KB tile and CU beyond the spin)\n", name, grid, spin * 0.01, ms * 1e3, ms * 1e3 / tiles_cu - spin * 0.01);
}
int main() {
    unsigned char* C; hipMalloc(&C, (size_t)RB * CB * 3072);
    for (int spin : {0, 3000}) for (int grid : {256, 32}) {
        run<0>("one 1-KB image per instruction (as shipped)", C, grid, spin);
        run<1>("four 256-B quarters of four images per instruction", C, grid, spin);
        run<2>("eight 128-B pieces, 768 B apart, per instruction", C, grid, spin);
    }
    return 0;
}
