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

extern "C" int gs_streams_create(int n, void** streams) {
    GS_CHECK_ARG(n >= 0 && (streams || n == 0), "gs_streams_create: bad arguments");
    for (int i = 0; i < n; ++i) streams[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        hipStream_t s = nullptr;
        const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (e != hipSuccess) return gs::fail(GS_ERR_HIP, "gs_streams_create: stream %d of %d: %s", i, n, hipGetErrorString(e));
        streams[i] = (void*)s;
    }
    return 0;
}
extern "C" int gs_streams_destroy(int n, void** streams) {
    int bad = 0;
    for (int i = 0; streams && i < n; ++i)
        if (streams[i]) { bad += hipStreamDestroy((hipStream_t)streams[i]) != hipSuccess; streams[i] = nullptr; }
    return bad ? gs::fail(GS_ERR_HIP, "gs_streams_destroy: %d streams could not be destroyed", bad) : 0;
}
