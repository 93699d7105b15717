// Error reporting, version and device check for libgansynth_hip.so.
#include <stdarg.h>
#include "gs_common.h"
#include "gs_prof.h"

namespace gs {
thread_local char g_err[512] = "";
ProfState g_prof;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace gs

extern "C" const char* gs_last_error(void) { return gs::g_err; }
extern "C" int gs_version(void) { return 100; }

extern "C" int gs_init(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gs::fail(GS_ERR_HIP, "gs_init: no HIP device");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return gs::fail(GS_ERR_HIP, "gs_init: hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return gs::fail(GS_ERR_UNSUPPORTED, "gs_init: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    return 0;
}
