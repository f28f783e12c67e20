// Shared host-side helpers of libptocr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <mutex>
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

// Per-device launch state (round 6).  A kernel that needs more dynamic LDS than the default has the limit raised once per (kernel,
// DEVICE) -- hipFuncSetAttribute is a per-device setting -- and a persistent kernel sizes its grid by the CU count of the CURRENT device.
// Both used to be function statics set by whichever thread and device came first: a second device in the process launched without the
// attribute, a concurrent first call raced.  Now: looked up per device under a mutex (a launch costs one hipGetDevice more).
constexpr int PT_MAX_DEVICES = 64;
struct DynLds { std::mutex mu; bool done[PT_MAX_DEVICES] = {}; };
inline int raise_dyn_lds(DynLds &st, const void *fn, int bytes) {
    int dev = 0;
    PT_HIP(hipGetDevice(&dev));
    PT_CHECK(dev >= 0 && dev < PT_MAX_DEVICES, "device ordinal %d beyond the %d this library keeps launch state for", dev, PT_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(st.mu);
    if (!st.done[dev]) {
        PT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        st.done[dev] = true;
    }
    return 0;
}
inline int current_device_cus(int *n_cu) {
    static std::mutex mu;
    static int cus[PT_MAX_DEVICES] = {};
    int dev = 0;
    PT_HIP(hipGetDevice(&dev));
    PT_CHECK(dev >= 0 && dev < PT_MAX_DEVICES, "device ordinal %d beyond the %d this library keeps launch state for", dev, PT_MAX_DEVICES);
    std::lock_guard<std::mutex> lk(mu);
    if (!cus[dev]) PT_HIP(hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev));
    *n_cu = cus[dev];
    return 0;
}

// Device memory the LIBRARY owns (post-process workspaces, the LSTM exchange buffers) comes from the allocator installed with
// ptocr_set_allocator (include/ptocr_hip.h), hipMalloc / hipFree by default.
extern ptocr_alloc_fn g_alloc;
extern ptocr_free_fn g_free;
extern long g_live_allocs;
template <typename T>
inline hipError_t dev_malloc(T **p, size_t bytes) {
    hipError_t e;
    if (g_alloc) e = g_alloc(reinterpret_cast<void **>(p), bytes) == 0 ? hipSuccess : hipErrorOutOfMemory;
    else e = hipMalloc(reinterpret_cast<void **>(p), bytes);
    if (e == hipSuccess) __atomic_add_fetch(&g_live_allocs, 1, __ATOMIC_RELAXED);
    return e;
}
inline hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    __atomic_sub_fetch(&g_live_allocs, 1, __ATOMIC_RELAXED);
    if (g_free) return g_free(p) == 0 ? hipSuccess : hipErrorInvalidValue;
    return hipFree(p);
}

}  // namespace ptocr
