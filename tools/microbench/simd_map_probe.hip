// Which SIMD does wave w of a 512-thread workgroup run on, and which TG_ID slots do the workgroups that share a CU get?
// (gfx950; the question behind rotating the key tiles of the one-pass plane attention backward by workgroup so that the
// SIMD that hosts two live waves in a ragged last key block differs between the two workgroups of a CU.)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/simd_map_probe.hip -o tools/microbench/simd_map_probe && tools/microbench/simd_map_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) void probe(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned acc = 0;
    for (int i = 0; i < spin; ++i) {                     // keep the workgroup resident for a while
        acc += smem[(threadIdx.x * 4 + i) & 1023];
        __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + ((size_t)blockIdx.x * 8 + wave) * 6;
        o[0] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_ID
        o[1] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // XCC_ID
        o[2] = (unsigned)t0;
        o[3] = (unsigned)t1;
        o[4] = acc;
        o[5] = (unsigned)(t0 >> 32);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 2048, spin = argc > 2 ? atoi(argv[2]) : 400;
    const size_t lds = 71168;                               // the fused backward's 69.5 KB: two workgroups per CU
    unsigned* d;
    hipMalloc(&d, (size_t)blocks * 8 * 6 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), lds, 0, d, spin);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(512), lds, 0, d, spin);
    hipDeviceSynchronize();
    std::vector<unsigned> h((size_t)blocks * 8 * 6);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // 1. wave index -> SIMD_ID
    int hist[8][4] = {};
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) hist[w][(h[((size_t)b * 8 + w) * 6] >> 4) & 3]++;
    printf("wave -> SIMD_ID histogram (rows: wave of the workgroup, columns: SIMD 0..3)\n");
    for (int w = 0; w < 8; ++w) printf("  wave %d: %6d %6d %6d %6d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    // 2. workgroups that overlap in time on one CU: their TG_IDs and block indices
    struct WG { int blk, tg; unsigned t0, t1; int s0; };
    std::map<unsigned, std::vector<WG>> cu;
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[(size_t)b * 8 * 6], xcc = h[(size_t)b * 8 * 6 + 1] & 15;
        const unsigned key = (xcc << 16) | (hw & 0xff00);          // XCC, SE, SH, CU
        cu[key].push_back({b, (int)((hw >> 16) & 15), h[(size_t)b * 8 * 6 + 2], h[(size_t)b * 8 * 6 + 3], (int)((hw >> 4) & 3)});
    }
    printf("CUs seen: %zu\n", cu.size());
    long ds0[4] = {};
    long same_parity = 0, diff_parity = 0, same_tg = 0, pairs = 0;
    std::map<int, long> dblk;
    int tg_hist[16] = {};
    for (auto& kv : cu) {
        auto& v = kv.second;
        for (size_t i = 0; i < v.size(); ++i) {
            tg_hist[v[i].tg]++;
            for (size_t j = i + 1; j < v.size(); ++j) {
                const bool overlap = (int)(v[i].t1 - v[j].t0) > 0 && (int)(v[j].t1 - v[i].t0) > 0;
                if (!overlap) continue;
                ++pairs;
                ds0[(v[i].s0 - v[j].s0) & 3]++;
                if (v[i].tg == v[j].tg) ++same_tg;
                if ((v[i].tg & 1) == (v[j].tg & 1)) ++same_parity; else ++diff_parity;
                int d = abs(v[i].blk - v[j].blk);
                dblk[d < 64 ? d : (d < 256 ? 64 : (d == 256 ? 256 : (d < 512 ? 257 : (d == 512 ? 512 : 513))))]++;
            }
        }
    }
    printf("TG_ID histogram:");
    for (int i = 0; i < 16; ++i) printf(" %d", tg_hist[i]);
    printf("\nco-resident workgroup pairs: %ld; same TG_ID %ld; same TG_ID parity %ld, different parity %ld\n", pairs, same_tg, same_parity, diff_parity);
    printf("SIMD of wave 0, difference between co-resident workgroups (0..3): %ld %ld %ld %ld\n", ds0[0], ds0[1], ds0[2], ds0[3]);
    printf("block index distance of co-resident pairs (bucket: count; 64 = 64..255, 257 = 257..511, 513 = above 512):");
    for (auto& kv : dblk) printf(" %d:%ld", kv.first, kv.second);
    printf("\n");
    auto& first = cu.begin()->second;
    printf("one CU's workgroups (block, TG_ID, start, end in 10-ns ticks):");
    for (size_t i = 0; i < first.size() && i < 12; ++i) printf(" (%d, %d, %u, %u)", first[i].blk, first[i].tg, first[i].t0 - first[0].t0, first[i].t1 - first[0].t0);
    printf("\n");
    return 0;
}
