// What v_permlane32_swap / v_permlane16_swap return (gfx950): lane-id probe.  hipcc --offload-arch=gfx950 -O3 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    const u32x2 s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + threadIdx.x] = s[0];
    out[192 + threadIdx.x] = s[1];
    const u32x2 t = __builtin_amdgcn_permlane16_swap(a, a, false, false);      // both operands the same value
    out[256 + threadIdx.x] = t[0];
    out[320 + threadIdx.x] = t[1];
    const u32x2 w = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    out[384 + threadIdx.x] = w[0];
    out[448 + threadIdx.x] = w[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 2048);
    k<<<1, 64>>>(d);
    unsigned h[512]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
    const char* names[8] = {"swap32(a,b)[0]", "swap32(a,b)[1]", "swap16(a,b)[0]", "swap16(a,b)[1]", "swap16(a,a)[0]", "swap16(a,a)[1]", "swap32(a,a)[0]", "swap32(a,a)[1]"};
    printf("in : a = lane, b = 100 + lane; shown at lanes 0, 16, 32, 48\n");
    for (int v = 0; v < 8; ++v) { printf("%s:", names[v]); for (int i = 0; i < 64; i += 16) printf(" [%d]=%u", i, h[64 * v + i]); printf("\n"); }
    return 0;
}
