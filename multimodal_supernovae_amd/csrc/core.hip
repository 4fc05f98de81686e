// Library-level entry points: version, error slot, device probe.
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
