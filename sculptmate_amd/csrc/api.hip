// C-ABI plumbing: version, thread-local error string, device query.
#include <stdarg.h>
#include <string.h>

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

#ifndef SCULPT_SOURCE_DIGEST
#define SCULPT_SOURCE_DIGEST "unknown"
#endif
const char *sculpt_source_digest(void) { return SCULPT_SOURCE_DIGEST; }

const char *sculpt_last_error(void) { return sculpt::g_err.c_str(); }

int sculpt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int sculpt_stream_create_cu_mask(int first_cu, int n_cus, sculpt_stream_t *stream_out) {
    SC_REQUIRE(stream_out && first_cu >= 0 && n_cus >= 1, "stream_create_cu_mask: bad arguments");
    const int total = sculpt::num_cus();
    SC_REQUIRE(first_cu + n_cus <= total, "stream_create_cu_mask: CUs [%d, %d) of %d", first_cu, first_cu + n_cus, total);
    uint32_t mask[16] = {0};  // up to 512 CUs; bit i = CU i in the driver's enumeration (striped over the XCDs)
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    SC_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)((total + 31) / 32), mask));
    *stream_out = reinterpret_cast<sculpt_stream_t>(st);
    return 0;
}

int sculpt_ply_face_records(const int64_t *faces_host, size_t n, uint8_t *records_host) {
    SC_REQUIRE(n == 0 || (faces_host && records_host), "ply_face_records: null argument");
    for (size_t i = 0; i < n; ++i) {
        uint8_t *r = records_host + 13 * i;
        const int32_t v[3] = {(int32_t)faces_host[3 * i], (int32_t)faces_host[3 * i + 1], (int32_t)faces_host[3 * i + 2]};
        r[0] = 3;
        memcpy(r + 1, v, 12);  // little-endian host (x86-64), unaligned destination
    }
    return 0;
}

int sculpt_stream_destroy(sculpt_stream_t stream) {
    SC_HIP(hipStreamDestroy(sculpt::as_stream(stream)));
    return 0;
}

}  // extern "C"
