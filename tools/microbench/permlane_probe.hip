// What v_permlane32_swap returns (gfx950): lane-id probe.  hipcc --offload-arch=gfx950 -O3 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512);
    k<<<1, 64>>>(d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("in : a = lane, b = 100 + lane\nr[0]:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[i]);
    printf("\nr[1]:"); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[64 + i]);
    printf("\n");
    return 0;
}
