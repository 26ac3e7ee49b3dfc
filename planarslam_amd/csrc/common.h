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

// Host<->device staging for the synchronous host-pointer entry points: one device block, inputs
// copied in on the stream, outputs copied back after the launch.
struct Stager {
    struct Item { const void* h_in; void* h_out; size_t bytes; size_t off; };
    std::vector<Item> items;
    size_t total = 0;
    DevBuf buf;
    // returns the index of the item; call dev<T>(idx) after upload()
    int in(const void* h, size_t bytes) { return add(h, nullptr, bytes); }
    int out(void* h, size_t bytes) { return add(nullptr, h, bytes); }
    int inout(void* h, size_t bytes) { return add(h, h, bytes); }
    int add(const void* hin, void* hout, size_t bytes) {
        items.push_back({hin, hout, bytes, total});
        total += align_up(bytes ? bytes : (size_t)1, (size_t)256);
        return (int)items.size() - 1;
    }
    int upload(hipStream_t st) {
        int rc = buf.alloc(total);
        if (rc) return rc;
        for (const Item& it : items)
            if (it.h_in && it.bytes) {
                hipError_t e = hipMemcpyAsync(buf.as<uint8_t>() + it.off, it.h_in, it.bytes, hipMemcpyHostToDevice, st);
                if (e != hipSuccess) { set_error("staging upload failed: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
            }
        return PLANAR_OK;
    }
    template <typename T> T* dev(int idx) const { return (T*)(buf.as<uint8_t>() + items[idx].off); }
    int download(hipStream_t st) {
        for (const Item& it : items)
            if (it.h_out && it.bytes) {
                hipError_t e = hipMemcpyAsync(it.h_out, buf.as<uint8_t>() + it.off, it.bytes, hipMemcpyDeviceToHost, st);
                if (e != hipSuccess) { set_error("staging download failed: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
            }
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("stream sync failed: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
        return PLANAR_OK;
    }
};

// XCD-aware (frame, block) from a ONE-dimensional grid of per_frame * B workgroups.  MI355X has 8 XCDs, each with its own 4 MB L2, and the dispatcher deals consecutive
// workgroup ids round-robin to them (block b -> XCD b % 8: observed, MI355X_MICROARCH.md; speed only, nothing depends on it for correctness).  A kernel whose
// workgroups of one frame re-read the same bytes (overlapping patches / support regions of one image) wants a frame's workgroups on ONE XCD, so that the frame's
// image is fetched into one L2 once instead of into all eight: ids cycle through groups of 8 frames (id % 8 picks the frame of the group = the XCD), a frame's
// workgroups follow each other on their XCD.  The last B % 8 frames keep the plain order.  Bijective over [0, per_frame * B).
#ifdef __HIPCC__
__device__ __forceinline__ void xcd_frame_block(int per_frame, int B, int& frame, int& blk) {
    const int n = (int)blockIdx.x, group = 8 * per_frame, full = (B >> 3) * group;
    if (n < full) { const int g = n / group, r = n - g * group; frame = g * 8 + (r & 7); blk = r >> 3; }
    else { const int r = n - full; frame = (B & ~7) + r / per_frame; blk = r - (r / per_frame) * per_frame; }
}
#endif

}  // namespace planar

struct planar_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // the stream work is enqueued on (own_stream unless overridden)
    planar::DevBuf scratch;         // grow-only device scratch for entry points that need temporaries
    // Optional side stream for the one-wavefront-per-frame kernels of the extractors (PEAC clustering, LSD region growing): planar_ctx_set_seq_stream.  Such a kernel
    // is latency-bound and occupies a CU for tens of milliseconds with four wavefronts; on a CU-masked stream (planar_cu_stream_create) it stays on a subset of the
    // CUs while the wide kernels of `stream` keep the rest.  seq_begin() / seq_end() bracket the launch: fork by event from `stream`, join back into it.
    hipStream_t seq_stream = nullptr;
    hipEvent_t seq_fork = nullptr, seq_join = nullptr;
    hipStream_t seq_begin() {
        if (!seq_stream) return stream;
        (void)hipEventRecord(seq_fork, stream);
        (void)hipStreamWaitEvent(seq_stream, seq_fork, 0);
        return seq_stream;
    }
    void seq_end() {
        if (!seq_stream) return;
        (void)hipEventRecord(seq_join, seq_stream);
        (void)hipStreamWaitEvent(stream, seq_join, 0);
    }
    int ensure_scratch(size_t bytes) {
        if (scratch.bytes >= bytes) return PLANAR_OK;
        // a previous kernel may still be reading the old block
        if (scratch.p) (void)hipStreamSynchronize(stream);
        return scratch.alloc(bytes);
    }
    // grow-only PINNED host block (planar_local_ba stages its whole problem through it: one copy each way instead of ~25 from pageable arrays)
    void* host_scratch = nullptr;
    size_t host_scratch_bytes = 0;
    int ensure_host_scratch(size_t bytes) {
        if (host_scratch_bytes >= bytes) return PLANAR_OK;
        if (host_scratch) { (void)hipStreamSynchronize(stream); (void)hipHostFree(host_scratch); host_scratch = nullptr; host_scratch_bytes = 0; }
        const size_t want = planar::align_up(bytes + bytes / 4, (size_t)4096);
        hipError_t e = hipHostMalloc(&host_scratch, want, hipHostMallocDefault);
        if (e != hipSuccess) { (void)hipGetLastError(); host_scratch = nullptr; planar::set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return PLANAR_EDEVICE; }
        host_scratch_bytes = want;
        return PLANAR_OK;
    }
};
