// Store throughput of a persistent 512-thread workgroup per CU writing 256 x 256 bf16 tiles of a [100864][3072] matrix -- the epilogue
// of bgemm_nt_kernel without the product -- in different lane -> address patterns.  One wave = rows 128 wm .. + 127, columns 64 wn .. + 63.
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/store_patterns.hip -o tools/microbench/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short u16;
constexpr int64_t M = 100864; constexpr int N = 3072;
template <int P>
__global__ __launch_bounds__(512) void k(u16* C, int tiles_n, int total, unsigned seed, int spin_ticks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 2, wn = wave & 3, l15 = lane & 15, g = lane >> 4;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int64_t m0 = (int64_t)(t / tiles_n) * 256; const int n0 = (t % tiles_n) * 256;
        const unsigned v = seed + t;
        if (spin_ticks) {      // "the K loop": every wave idles for spin_ticks x 10 ns, then the workgroup meets (as the K loop's barriers make it)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(4);
            __syncthreads();
        }
        if (P == 0) {          // 8 bytes per lane: rows 16 i + l15, columns 16 j + 4 g
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j)
                *reinterpret_cast<uint2*>(C + (m0 + wm * 128 + 16 * i + l15) * N + n0 + wn * 64 + 16 * j + 4 * g) = make_uint2(v, v + i);
        } else if (P == 1) {   // 16 bytes per lane: rows 16 i + l15, columns 32 pr + 8 g
            for (int i = 0; i < 8; ++i) for (int pr = 0; pr < 2; ++pr)
                *reinterpret_cast<uint4*>(C + (m0 + wm * 128 + 16 * i + l15) * N + n0 + wn * 64 + 32 * pr + 8 * g) = make_uint4(v, v + i, v, v);
        } else if (P == 2) {   // 16 bytes per lane, a wave instruction = 8 rows x 128 contiguous bytes (lane = row l >> 3, piece l & 7)
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<uint4*>(C + (m0 + wm * 128 + 8 * i + (lane >> 3)) * N + n0 + wn * 64 + 8 * (lane & 7)) = make_uint4(v, v + i, v, v);
        } else if (P == 3) {   // whole 512-byte tile rows: a wave instruction = 2 rows x 512 bytes (wave w: rows 32 w .. + 31)
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<uint4*>(C + (m0 + wave * 32 + 2 * i + (lane >> 5)) * N + n0 + 8 * (lane & 31)) = make_uint4(v, v + i, v, v);
        } else if (P == 4) {   // 64 rows x 16 bytes per instruction (lane = row): wave = rows 32 w' .., 16 column chunks
            for (int i = 0; i < 16; ++i) {
                const int slot = wave * 16 + i;            // 128 slots = 4 row groups of 64 x 32 chunks of 8 columns
                *reinterpret_cast<uint4*>(C + (m0 + (slot >> 5) * 64 + lane) * N + n0 + 8 * (slot & 31)) = make_uint4(v, v + i, v, v);
            }
        } else if (P == 5) {   // 32 rows x 32 bytes per instruction
            for (int i = 0; i < 16; ++i) {
                const int slot = wave * 16 + i;            // 128 slots = 8 row groups of 32 x 16 chunks of 16 columns
                *reinterpret_cast<uint4*>(C + (m0 + (slot >> 4) * 32 + (lane >> 1)) * N + n0 + 16 * (slot & 15) + 8 * (lane & 1)) = make_uint4(v, v + i, v, v);
            }
        } else if (P == 6) {   // 4 rows x 256 bytes per instruction
            for (int i = 0; i < 16; ++i) {
                const int slot = wave * 16 + i;            // 128 slots = 64 row groups of 4 x 2 halves of 128 columns
                *reinterpret_cast<uint4*>(C + (m0 + (slot >> 1) * 4 + (lane >> 4)) * N + n0 + 128 * (slot & 1) + 8 * (lane & 15)) = make_uint4(v, v + i, v, v);
            }
        } else if (P == 7) {   // fp32 tile rows (the fp32-output epilogue): 16 rows x 64 B, 32 instructions
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j)
                *reinterpret_cast<uint4*>(reinterpret_cast<float*>(C) + (m0 + wm * 128 + 16 * i + l15) * (N / 2) + (n0 + wn * 64 + 16 * j + 4 * g) % (N / 2)) = make_uint4(v, v + i, v, v);
        }
    }
}
template <int P> void run(const char* name, u16* C, int grid, int spin = 0) {
    const int tiles_n = N / 256, total = (int)(M / 256) * tiles_n;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<P>, dim3(grid), dim3(512), 0, 0, C, tiles_n, total, 1u, spin);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<P>, dim3(grid), dim3(512), 0, 0, C, tiles_n, total, 2u + r, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double tiles_cu = (total + grid - 1) / grid;
    printf("%-72s grid %3d spin %4.1f us: %7.1f us  %6.2f TB/s  (%.2f us per tile and CU beyond the spin)\n", name, grid, spin * 0.01, ms * 1e3, M * N * 2.0 / ms * 1e-9,
           ms * 1e3 / tiles_cu - spin * 0.01);
}
int main() {
    u16* C; hipMalloc(&C, M * N * 2);
    for (int grid : {256, 32}) {
        run<0>("8 B / lane, 16 rows x 32 B per instruction (old epilogue)", C, grid);
        run<1>("16 B / lane, 16 rows x 64 B per instruction (new epilogue)", C, grid);
        run<2>("16 B / lane, 8 rows x 128 B per instruction", C, grid);
        run<3>("16 B / lane, 2 rows x 512 B per instruction", C, grid);
        run<4>("16 B / lane, 64 rows x 16 B per instruction", C, grid);
        run<5>("16 B / lane, 32 rows x 32 B per instruction", C, grid);
        run<6>("16 B / lane, 4 rows x 256 B per instruction", C, grid);
    }
    printf("-- with a 20-us idle phase per tile (the product), every CU\n");
    run<0>("8 B / lane, 16 rows x 32 B per instruction (old epilogue)", C, 256, 2000);
    run<1>("16 B / lane, 16 rows x 64 B per instruction (new epilogue)", C, 256, 2000);
    run<2>("16 B / lane, 8 rows x 128 B per instruction", C, 256, 2000);
    run<6>("16 B / lane, 4 rows x 256 B per instruction", C, 256, 2000);
    return 0;
}
