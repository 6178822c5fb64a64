// Shared host-side helpers of libislam_hip.so: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/islam_hip.h"

namespace islam {

char* err_buf();   // thread-local, 512 bytes (defined in abi.hip)

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define ISLAM_HIP_CHECK(expr)                                                                       \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::islam::fail(ISLAM_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
    } while (0)

#define ISLAM_LAUNCH_CHECK() ISLAM_HIP_CHECK(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

}  // namespace islam
