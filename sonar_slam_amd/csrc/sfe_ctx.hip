// Context, device memory, stream timing: the plumbing of libsonarfe (no torch, no CPU fallback).
#include "sfe_internal.h"

#include <cstring>

static std::string g_create_err;

int sfe_set_err(sfe_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_err = buf;
    return code;
}

void *sfe_scratch(sfe_ctx *ctx, int slot, size_t bytes)
{
    auto &b = ctx->scratch[slot];
    if (bytes <= b.cap && b.p)
        return b.p;
    if (b.p) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->stream2)
            (void)hipStreamSynchronize(ctx->stream2); // the ICP target preparation may run there
        if (ctx->stream_copy)
            (void)hipStreamSynchronize(ctx->stream_copy); // an upload into the old block may still be in flight
        // a "known to be zero" note about this buffer dies with it: hipMalloc commonly hands the same address out
        // again, with whatever the allocation holds (extract_dev trusted the pointer alone)
        if (ctx->bm_clean_ptr == b.p) {
            ctx->bm_clean_ptr = nullptr;
            ctx->bm_clean_bytes = 0;
        }
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 4 + 4096;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        sfe_set_err(ctx, SFE_ERR_HIP, "hipMalloc(%zu) for scratch slot %d failed", want, slot);
        return nullptr;
    }
    b.cap = want;
    return b.p;
}

void *sfe_pinned_begin(sfe_ctx *ctx, size_t bytes)
{
    auto &b = ctx->pin[ctx->pin_next];
    if (b.pending) {
        if (hipEventSynchronize(b.ev) != hipSuccess) {
            sfe_set_err(ctx, SFE_ERR_HIP, "hipEventSynchronize on a pinned staging block failed");
            return nullptr;
        }
        b.pending = false;
    }
    if (bytes > b.cap || !b.p) {
        if (b.p)
            (void)hipHostFree(b.p);
        b.p = nullptr;
        b.cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&b.p, want, hipHostMallocDefault) != hipSuccess) {
            b.p = nullptr;
            sfe_set_err(ctx, SFE_ERR_HIP, "hipHostMalloc(%zu) for the pinned staging block failed", want);
            return nullptr;
        }
        b.cap = want;
    }
    return b.p;
}

int sfe_pinned_end(sfe_ctx *ctx, hipStream_t s)
{
    auto &b = ctx->pin[ctx->pin_next];
    SFE_HIP(ctx, hipEventRecord(b.ev, s));
    b.pending = true;
    ctx->pin_next ^= 1;
    return 0;
}

extern "C" {

const char *sfe_version(void) { return "sonarfe 0.1 (gfx950)"; }

int sfe_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        if (count)
            *count = 0;
        return sfe_set_err(nullptr, SFE_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    if (count)
        *count = n;
    return 0;
}

int sfe_ctx_create(int device, sfe_ctx **out)
{
    if (!out)
        return sfe_set_err(nullptr, SFE_ERR_ARG, "sfe_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return sfe_set_err(nullptr, SFE_ERR_NODEV,
                           "no HIP device available (%s); libsonarfe has no CPU fallback",
                           e != hipSuccess ? hipGetErrorString(e) : "count = 0");
    if (device < 0 || device >= n)
        return sfe_set_err(nullptr, SFE_ERR_ARG, "device %d out of range [0,%d)", device, n);
    e = hipSetDevice(device);
    if (e != hipSuccess)
        return sfe_set_err(nullptr, SFE_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess)
        return sfe_set_err(nullptr, SFE_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return sfe_set_err(nullptr, SFE_ERR_NODEV, "device %d is %s; libsonarfe is built for gfx950 only",
                           device, prop.gcnArchName);
    sfe_ctx *c = new sfe_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_loop, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pin[0].ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->pin[1].ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return sfe_set_err(nullptr, SFE_ERR_HIP, "stream/event creation failed on device %d", device);
    }
    *out = c;
    return 0;
}

void sfe_ctx_destroy(sfe_ctx *ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamSynchronize(ctx->stream_copy);
    for (auto &b : ctx->scratch)
        if (b.p)
            (void)hipFree(b.p);
    for (auto &b : ctx->pin) {
        if (b.p)
            (void)hipHostFree(b.p);
        if (b.ev)
            (void)hipEventDestroy(b.ev);
    }
    for (auto &b : ctx->pin_io)
        if (b.p)
            (void)hipHostFree(b.p);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipEventDestroy(ctx->ev_prep);
    (void)hipEventDestroy(ctx->ev_loop);
    (void)hipEventDestroy(ctx->ev_copy);
    (void)hipEventDestroy(ctx->ev_compute);
    (void)hipStreamDestroy(ctx->stream_copy);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *sfe_last_error(sfe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int sfe_sync(sfe_ctx *ctx)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_device_name(sfe_ctx *ctx, char *buf, int cap)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, buf && cap > 0);
    hipDeviceProp_t prop;
    SFE_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, (size_t)cap, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int sfe_malloc(sfe_ctx *ctx, size_t bytes, void **dptr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, dptr);
    *dptr = nullptr;
    SFE_HIP(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return 0;
}

int sfe_free(sfe_ctx *ctx, void *dptr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    // nothing this context has enqueued may still touch the block: kernels, the ICP preparation on the side stream,
    // uploads on the copy stream (sfe_memcpy_h2d_async)
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream_copy));
    if (dptr)
        SFE_HIP(ctx, hipFree(dptr));
    return 0;
}

int sfe_memcpy_h2d(sfe_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int sfe_memcpy_d2h(sfe_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- streamed inputs: pinned host memory, uploads on a copy stream next to the kernels ----
int sfe_host_alloc(sfe_ctx *ctx, size_t bytes, void **hptr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, hptr);
    *hptr = nullptr;
    SFE_HIP(ctx, hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return 0;
}

int sfe_host_free(sfe_ctx *ctx, void *hptr)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream_copy));
    if (hptr)
        SFE_HIP(ctx, hipHostFree(hptr));
    return 0;
}

int sfe_memcpy_h2d_async(sfe_ctx *ctx, void *dst_dev, const void *src_pinned, size_t bytes)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, dst_dev && src_pinned);
    SFE_HIP(ctx, hipMemcpyAsync(dst_dev, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream_copy));
    SFE_HIP(ctx, hipEventRecord(ctx->ev_copy, ctx->stream_copy));
    return 0;
}

int sfe_stream_fence(sfe_ctx *ctx, int what)
{
    if (int rc = sfe_use(ctx))
        return rc;
    if (what == 0) { // kernels enqueued from now on run behind every upload enqueued so far
        SFE_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_copy, 0));
    } else if (what == 1) { // uploads enqueued from now on run behind every kernel enqueued so far
        SFE_HIP(ctx, hipEventRecord(ctx->ev_compute, ctx->stream));
        SFE_HIP(ctx, hipStreamWaitEvent(ctx->stream_copy, ctx->ev_compute, 0));
    } else if (what == 2) { // host waits for the uploads
        SFE_HIP(ctx, hipStreamSynchronize(ctx->stream_copy));
    } else {
        return sfe_set_err(ctx, SFE_ERR_ARG, "sfe_stream_fence: what = %d", what);
    }
    return 0;
}

int sfe_memset(sfe_ctx *ctx, void *dst_dev, int value, size_t bytes)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return 0;
}

// debug: copy the first `bytes` of scratch slot `slot` to the host (the kernels' intermediate results: tools/dbg_*.py)
int sfe_debug_read_scratch(sfe_ctx *ctx, int slot, void *dst_host, size_t bytes)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_ARG(ctx, slot >= 0 && slot < SFE_NSCRATCH && dst_host && ctx->scratch[slot].p && ctx->scratch[slot].cap >= bytes);
    SFE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SFE_HIP(ctx, hipMemcpy(dst_host, ctx->scratch[slot].p, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int sfe_timer_start(sfe_ctx *ctx)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

int sfe_timer_stop(sfe_ctx *ctx, float *elapsed_ms)
{
    if (int rc = sfe_use(ctx))
        return rc;
    SFE_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    SFE_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    SFE_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (elapsed_ms)
        *elapsed_ms = ms;
    return 0;
}

} // extern "C"
