// C-ABI plumbing: version, thread-local error string, device query.
#include <stdarg.h>

#include "common.h"

namespace sculpt {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int num_cus() {
    static int cached = 0;
    if (cached) return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cached;
}

}  // namespace sculpt

extern "C" {

int sculpt_version(void) { return SCULPT_ABI_VERSION; }

const char *sculpt_last_error(void) { return sculpt::g_err.c_str(); }

int sculpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
