// Shared host-side helpers of libptocr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include "../../include/ptocr_hip.h"

namespace ptocr {

extern thread_local char g_err[512];

inline int fail(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
    return 1;
}

#define PT_HIP(call)                                                                              \
    do { hipError_t e_ = (call);                                                                  \
         if (e_ != hipSuccess) return ::ptocr::fail("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define PT_CHECK(cond, ...) do { if (!(cond)) return ::ptocr::fail(__VA_ARGS__); } while (0)

inline int launch_ok(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace ptocr
