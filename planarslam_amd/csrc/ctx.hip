// planarslam_amd/csrc/ctx.hip — context / error plumbing of libplanar_hip.so.
#include "common.h"

namespace planar {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace planar

extern "C" {

const char* planar_last_error(void) { return planar::g_err; }
int planar_abi_version(void) { return 100; }

int planar_ctx_create(planar_ctx** out, int device) {
    PLANAR_REQUIRE(out != nullptr, PLANAR_EINVAL, "out is null");
    *out = nullptr;
    int n = 0;
    PLANAR_HIP_CHECK(hipGetDeviceCount(&n));
    PLANAR_REQUIRE(device >= 0 && device < n, PLANAR_EDEVICE, "no such HIP device");
    PLANAR_HIP_CHECK(hipSetDevice(device));
    planar_ctx* c = new (std::nothrow) planar_ctx();
    PLANAR_REQUIRE(c != nullptr, PLANAR_ENOMEM, "host allocation failed");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; planar::set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return PLANAR_EDEVICE; }
    c->stream = c->own_stream;
    *out = c;
    return PLANAR_OK;
}

void planar_ctx_destroy(planar_ctx* ctx) {
    if (!ctx) return;
    if (ctx->seq_fork) (void)hipEventDestroy(ctx->seq_fork);
    if (ctx->seq_join) (void)hipEventDestroy(ctx->seq_join);
    if (ctx->host_scratch) (void)hipHostFree(ctx->host_scratch);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int planar_ctx_set_stream(planar_ctx* ctx, void* s) {
    PLANAR_REQUIRE(ctx != nullptr, PLANAR_EINVAL, "ctx is null");
    ctx->stream = s ? (hipStream_t)s : ctx->own_stream;
    return PLANAR_OK;
}

int planar_cu_stream_create(int device, const uint32_t* cu_mask, int n_words, void** out_stream) {
    PLANAR_REQUIRE(out_stream && cu_mask && n_words >= 1 && n_words <= 32, PLANAR_EINVAL, "bad argument");
    *out_stream = nullptr;
    bool any = false;
    for (int i = 0; i < n_words; i++) any = any || cu_mask[i] != 0u;
    PLANAR_REQUIRE(any, PLANAR_EINVAL, "empty CU mask");
    PLANAR_HIP_CHECK(hipSetDevice(device));
    hipStream_t s = nullptr;
    PLANAR_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask));
    *out_stream = (void*)s;
    return PLANAR_OK;
}

void planar_cu_stream_destroy(void* stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }

int planar_ctx_set_seq_stream(planar_ctx* ctx, void* s) {
    PLANAR_REQUIRE(ctx != nullptr, PLANAR_EINVAL, "ctx is null");
    PLANAR_HIP_CHECK(hipSetDevice(ctx->device));
    if (s && !ctx->seq_fork) {
        PLANAR_HIP_CHECK(hipEventCreateWithFlags(&ctx->seq_fork, hipEventDisableTiming));
        PLANAR_HIP_CHECK(hipEventCreateWithFlags(&ctx->seq_join, hipEventDisableTiming));
    }
    ctx->seq_stream = (hipStream_t)s;
    return PLANAR_OK;
}

void* planar_ctx_get_stream(planar_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int planar_ctx_sync(planar_ctx* ctx) {
    PLANAR_REQUIRE(ctx != nullptr, PLANAR_EINVAL, "ctx is null");
    PLANAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PLANAR_OK;
}

}  // extern "C"
