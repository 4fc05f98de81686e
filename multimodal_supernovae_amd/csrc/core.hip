// Library-level entry points: version, error slot, device probe, shader-clock probe.
#include <stdarg.h>

#include "msn_common.h"

namespace msn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace msn

extern "C" int msn_version(void) { return 100; }  // 0.1.0
extern "C" const char* msn_last_error(void) { return msn::g_err; }
extern "C" int msn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// The shader clock the chip holds WHILE other kernels run (measurement, bench.py): one wave stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz), sleeps until `microseconds` of real time have passed, and stamps again: clock = d cycles / d ticks x
// 100 MHz (MI355X_MICROARCH.md, "in-kernel clock").  Launched on a side stream beside a training step it reads the clock under
// that step's load -- the matrix-core-dense plane kernels pull it from 2.4 GHz to ~2.0, and boxes differ.
namespace msn {
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0;
    while (t1 - t0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
        t1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0;
    out[1] = t1 - t0;
}
}  // namespace msn
extern "C" int msn_clock_probe(unsigned long long* out2, int microseconds, msn_stream_t stream) {
    MSN_REQUIRE(out2 && microseconds > 0 && microseconds <= 2000000, "msn_clock_probe: an output pair and 1 .. 2 000 000 us");
    hipLaunchKernelGGL(msn::clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out2,
                       (unsigned long long)microseconds * 100ull);
    MSN_LAUNCH_CHECK();
    return MSN_OK;
}
