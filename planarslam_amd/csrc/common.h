// planarslam_amd/csrc/common.h — shared host-side plumbing for libplanar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/planar_abi.h"

namespace planar {

void set_error(const char* fmt, ...);

#define PLANAR_HIP_CHECK(expr)                                                                        \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            ::planar::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (_e == hipErrorOutOfMemory) ? PLANAR_ENOMEM : PLANAR_EDEVICE;                      \
        }                                                                                             \
    } while (0)

#define PLANAR_REQUIRE(cond, code, msg)                  \
    do {                                                 \
        if (!(cond)) {                                   \
            ::planar::set_error("%s: %s", __func__, msg); \
            return (code);                               \
        }                                                \
    } while (0)

template <typename T>
static inline T align_up(T v, T a) { return (v + a - 1) / a * a; }

// Device memory owner (no exceptions).
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e)); return PLANAR_ENOMEM; }
        bytes = n;
        return PLANAR_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    ~DevBuf() { release(); }
    template <typename T> T* as() const { return (T*)p; }
};

}  // namespace planar

struct planar_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // the stream work is enqueued on (own_stream unless overridden)
};
