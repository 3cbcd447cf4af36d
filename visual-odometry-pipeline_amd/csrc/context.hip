// Context life cycle, error reporting and device-buffer management of libvohip.so.
#include "vohip_internal.h"

#include <new>

static thread_local char g_create_error[256] = {0};

int vohip_fail(vohip_ctx *ctx, int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    if (ctx)
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    else
        vsnprintf(g_create_error, sizeof(g_create_error), fmt, ap);
    va_end(ap);
    return status;
}

int vohip_reserve(vohip_ctx *ctx, vohip_buf &b, size_t bytes)
{
    if (bytes <= b.cap && b.ptr) return VOHIP_OK;
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
    if (b.ptr) {
        VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        VOHIP_HIP(ctx, hipFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
    }
    VOHIP_HIP(ctx, hipMalloc(&b.ptr, want));
    b.cap = want;
    b.pinned_host = false;
    return VOHIP_OK;
}

int vohip_reserve_pinned(vohip_ctx *ctx, vohip_buf &b, size_t bytes)
{
    if (bytes <= b.cap && b.ptr) return VOHIP_OK;
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
    if (b.ptr) {
        VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        VOHIP_HIP(ctx, hipHostFree(b.ptr));
        b.ptr = nullptr;
        b.cap = 0;
    }
    VOHIP_HIP(ctx, hipHostMalloc(&b.ptr, want, hipHostMallocDefault));
    b.cap = want;
    b.pinned_host = true;
    return VOHIP_OK;
}

void vohip_release(vohip_buf &b)
{
    if (!b.ptr) return;
    if (b.pinned_host)
        (void)hipHostFree(b.ptr);
    else
        (void)hipFree(b.ptr);
    b.ptr = nullptr;
    b.cap = 0;
}

extern "C" {

const char *vohip_version(void) { return "0.1"; }

int vohip_create(vohip_ctx **out, int device_id, void *hip_stream)
{
    if (!out) return vohip_fail(nullptr, VOHIP_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return vohip_fail(nullptr, VOHIP_ERR_NO_DEVICE, "no HIP device visible (%s)",
                          e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return vohip_fail(nullptr, VOHIP_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)",
                          device_id, ndev);
    vohip_ctx *ctx = new (std::nothrow) vohip_ctx();
    if (!ctx) return vohip_fail(nullptr, VOHIP_ERR_INVALID_ARGUMENT, "out of host memory");
    ctx->device = device_id;
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        delete ctx;
        return vohip_fail(nullptr, VOHIP_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    }
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->owns_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return vohip_fail(nullptr, VOHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->owns_stream = true;
    }
    // the static scratch of the allocation-free entry point: its address on this device
    if (const int st = vohip_ransac_scratch_init(ctx)) {
        vohip_fail(nullptr, st, "%s", ctx->err);
        if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return st;
    }
    *out = ctx;
    return VOHIP_OK;
}

int vohip_destroy(vohip_ctx *ctx)
{
    if (!ctx) return VOHIP_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < VOHIP_NBUF; ++i) vohip_release(ctx->bufs[i]);
    if (ctx->klt_ev) {
        for (int i = 0; i < 2 * vohip_ctx::KLT_TIME_PAIRS; ++i)
            if (ctx->klt_ev[i]) (void)hipEventDestroy(ctx->klt_ev[i]);
        delete[] ctx->klt_ev;
    }
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VOHIP_OK;
}

const char *vohip_last_error(const vohip_ctx *ctx) { return ctx ? ctx->err : g_create_error; }

void *vohip_stream(vohip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int vohip_set_option(vohip_ctx *ctx, const char *name, int value)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, name, "NULL option name");
    if (std::strcmp(name, "klt_stats") == 0) {
        ctx->opt_klt_stats = value ? 1 : 0;
        return VOHIP_OK;
    }
    if (std::strcmp(name, "klt_time") == 0) {
        if (value && !ctx->klt_ev) {
            VOHIP_HIP(ctx, hipSetDevice(ctx->device));
            ctx->klt_ev = new (std::nothrow) hipEvent_t[2 * vohip_ctx::KLT_TIME_PAIRS]();
            if (!ctx->klt_ev) return vohip_fail(ctx, VOHIP_ERR_HIP, "out of host memory");
            for (int i = 0; i < 2 * vohip_ctx::KLT_TIME_PAIRS; ++i) VOHIP_HIP(ctx, hipEventCreate(&ctx->klt_ev[i]));
        }
        ctx->opt_klt_time = value ? 1 : 0;
        ctx->klt_ev_used = 0;
        return VOHIP_OK;
    }
    if (std::strcmp(name, "ransac_score_variant") == 0) {
        VOHIP_REQUIRE(ctx, value >= -1 && value <= 5, "ransac_score_variant must be -1 (auto) or 0..5");
        ctx->opt_ransac_variant = value;
        return VOHIP_OK;
    }
    return vohip_fail(ctx, VOHIP_ERR_INVALID_ARGUMENT, "unknown option '%s'", name);
}

int vohip_synchronize(vohip_ctx *ctx)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

}  // extern "C"
