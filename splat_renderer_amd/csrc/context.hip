// context.hip — ctx, buffers, timing.  Replaces the reference's GPUDevice/GPUBuffer plumbing
// (/root/reference/src/main.ts:16-36; createBuffer/writeBuffer/mapAsync call sites throughout).
#include "common.h"

#include <cstdio>
#include <cstring>

static thread_local std::string g_tls_err = "";

int ctx_fail(splat_ctx *ctx, int code, const char *what, hipError_t e) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    else
        snprintf(buf, sizeof buf, "%s", what);
    if (ctx) ctx->err = buf;
    g_tls_err = buf;
    return code;
}

int ctx_ensure_scan_ws(splat_ctx *ctx, size_t bytes) {
    if (ctx->scan_ws_bytes >= bytes) return SPLAT_OK;
    if (ctx->scan_ws) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ctx->scan_ws));
        ctx->scan_ws = nullptr;
        ctx->scan_ws_bytes = 0;
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (hipMalloc(&ctx->scan_ws, want) != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "scan workspace hipMalloc");
    ctx->scan_ws_bytes = want;
    return SPLAT_OK;
}

int ctx_ensure_pinned(splat_ctx *ctx, size_t bytes) {
    if (ctx->pinned_bytes >= bytes) return SPLAT_OK;
    if (ctx->pinned) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    size_t want = bytes < (1u << 16) ? (1u << 16) : bytes;
    if (hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault) != hipSuccess)
        return ctx_fail(ctx, SPLAT_ERR_OOM, "pinned staging hipHostMalloc");
    ctx->pinned_bytes = want;
    return SPLAT_OK;
}

void stage_begin(splat_ctx *ctx, int stage) {
    if (!ctx->timing || !((ctx->timing_mask >> stage) & 1u)) return;
    StageTimer &t = ctx->timers[stage];
    if (t.used == t.beg.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        t.beg.push_back(a);
        t.end.push_back(b);
    }
    (void)hipEventRecord(t.beg[t.used], ctx->stream);
}

// For a stage that is ONE kernel: the event pair to attach to the launch itself (hipExtLaunchKernelGGL
// takes the kernel's own start/stop timestamps, with no marker packets around it in the queue — a
// hipEventRecord pair costs ~6 us of idle GPU per launch).  False when the stage is not being timed.
bool stage_event_pair(splat_ctx *ctx, int stage, hipEvent_t *start, hipEvent_t *stop) {
    if (!ctx->timing || !((ctx->timing_mask >> stage) & 1u)) return false;
    StageTimer &t = ctx->timers[stage];
    if (ctx->timing_every > 1 && (t.tick++ % ctx->timing_every) != 0) return false; // (splat_set_timing_sampling; a count per stage)
    if (t.used == t.beg.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
        t.beg.push_back(a);
        t.end.push_back(b);
    }
    *start = t.beg[t.used];
    *stop = t.end[t.used];
    ++t.used;
    return true;
}

void stage_end(splat_ctx *ctx, int stage) {
    if (!ctx->timing || !((ctx->timing_mask >> stage) & 1u)) return;
    StageTimer &t = ctx->timers[stage];
    if (t.used >= t.beg.size()) return;
    (void)hipEventRecord(t.end[t.used], ctx->stream);
    ++t.used;
}

static int ctx_create_impl(int device, void *stream, bool have_stream, splat_ctx **out) {
    if (!out) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "splat_ctx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return ctx_fail(nullptr, SPLAT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)", e);
    if (device < 0 || device >= count) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "device ordinal out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return ctx_fail(nullptr, SPLAT_ERR_HIP, "hipSetDevice", e);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return ctx_fail(nullptr, SPLAT_ERR_HIP, "hipGetDeviceProperties", e);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        std::string m = std::string("device is ") + prop.gcnArchName + "; libsplat_hip is built for gfx950 (MI355X) only";
        return ctx_fail(nullptr, SPLAT_ERR_NO_DEVICE, m.c_str());
    }
    splat_ctx *ctx = new splat_ctx();
    ctx->device = device;
    if (have_stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return ctx_fail(nullptr, SPLAT_ERR_HIP, "hipStreamCreate", e);
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return SPLAT_OK;
}

extern "C" {

int splat_abi_version(void) { return SPLAT_ABI_VERSION; }

int splat_ctx_create(int device_ordinal, splat_ctx **out) { return ctx_create_impl(device_ordinal, nullptr, false, out); }

int splat_ctx_create_on_stream(int device_ordinal, void *hip_stream, splat_ctx **out) {
    return ctx_create_impl(device_ordinal, hip_stream, true, out);
}

void splat_ctx_destroy(splat_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->timers) {
        for (auto e : t.beg) (void)hipEventDestroy(e);
        for (auto e : t.end) (void)hipEventDestroy(e);
    }
    if (ctx->d_consumed) (void)hipFree(ctx->d_consumed);
    for (auto &h : ctx->px_hist)
        if (h.mem) (void)hipFree(h.mem);
    if (ctx->scan_ws) (void)hipFree(ctx->scan_ws);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *splat_last_error(splat_ctx *ctx) { return ctx ? ctx->err.c_str() : g_tls_err.c_str(); }

int splat_sync(splat_ctx *ctx) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SPLAT_OK;
}

int splat_set_timing(splat_ctx *ctx, int enabled) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ctx->timing = enabled != 0;
    if (ctx->timing) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto &t : ctx->timers) t.used = 0; // start a new sample set
        if (ctx->d_consumed) HIP_TRY(ctx, hipMemsetAsync(ctx->d_consumed, 0, (size_t)ctx->consumed_tiles * 16, ctx->stream));
    }
    return SPLAT_OK;
}

int splat_set_timing_sampling(splat_ctx *ctx, uint32_t every) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, every >= 1);
    ctx->timing_every = every;
    for (auto &t : ctx->timers) t.tick = 0;
    return SPLAT_OK;
}

int splat_set_timing_stages(splat_ctx *ctx, uint32_t stage_mask) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ctx->timing_mask = stage_mask;
    return SPLAT_OK;
}

} // extern "C"

int ctx_ensure_consumed(splat_ctx *ctx, uint32_t tiles) {
    if (tiles <= ctx->consumed_tiles) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // (grown mid-run: counts gathered so far at the old size are kept)
    unsigned long long *bigger = nullptr;
    // (two counters per tile: entries staged, entries consumed)
    if (hipMalloc((void **)&bigger, (size_t)tiles * 16) != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "consumed counters hipMalloc");
    // (on the context's own stream: it is non-blocking, so null-stream work is not ordered before what is launched next)
    HIP_TRY(ctx, hipMemsetAsync(bigger, 0, (size_t)tiles * 16, ctx->stream));
    if (ctx->d_consumed) {
        HIP_TRY(ctx, hipMemcpyAsync(bigger, ctx->d_consumed, (size_t)ctx->consumed_tiles * 16, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->d_consumed);
    }
    ctx->d_consumed = bigger;
    ctx->consumed_tiles = tiles;
    return SPLAT_OK;
}

extern "C" {

int splat_timing_consumed(splat_ctx *ctx, uint64_t *staged, uint64_t *consumed) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, staged != nullptr && consumed != nullptr);
    *staged = *consumed = 0;
    if (!ctx->d_consumed) return SPLAT_OK; // no timed frame has run
    std::vector<unsigned long long> host((size_t)ctx->consumed_tiles * 2);
    HIP_TRY(ctx, hipMemcpyAsync(host.data(), ctx->d_consumed, host.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long s = 0, c = 0;
    for (size_t t = 0; t < host.size(); t += 2) {
        s += host[t];
        c += host[t + 1];
    }
    *staged = s;
    *consumed = c;
    return SPLAT_OK;
}

int splat_stage_time_stats(splat_ctx *ctx, int stage, uint32_t *samples, double *total_ms) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, stage >= 0 && stage < SPLAT_STAGE_COUNT && samples && total_ms);
    StageTimer &t = ctx->timers[stage];
    *samples = (uint32_t)t.used;
    *total_ms = 0.0;
    for (size_t i = 0; i < t.used; ++i) {
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventSynchronize(t.end[i]));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, t.beg[i], t.end[i]));
        *total_ms += ms;
    }
    return SPLAT_OK;
}

int splat_stage_time_ms(splat_ctx *ctx, int stage, float *ms) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, stage >= 0 && stage < SPLAT_STAGE_COUNT && ms);
    StageTimer &t = ctx->timers[stage];
    if (t.used == 0) return ctx_fail(ctx, SPLAT_ERR_STATE, "stage has not been timed (call splat_set_timing first)");
    HIP_TRY(ctx, hipEventSynchronize(t.end[t.used - 1]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, t.beg[t.used - 1], t.end[t.used - 1]));
    return SPLAT_OK;
}

int splat_buf_alloc(splat_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, dptr != nullptr);
    *dptr = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) return ctx_fail(ctx, SPLAT_ERR_OOM, "hipMalloc", e);
    return SPLAT_OK;
}

int splat_buf_free(splat_ctx *ctx, void *dptr) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    if (!dptr) return SPLAT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(dptr));
    return SPLAT_OK;
}

int splat_buf_upload(splat_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, dst && (src || bytes == 0));
    if (bytes == 0) return SPLAT_OK;
    // pageable source: hipMemcpyAsync stages it before returning, so src may be reused at once
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SPLAT_OK;
}

int splat_buf_download(splat_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, src && (dst || bytes == 0));
    if (bytes == 0) return SPLAT_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SPLAT_OK;
}

int splat_buf_zero(splat_ctx *ctx, void *dptr, size_t bytes) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, dptr || bytes == 0);
    if (bytes == 0) return SPLAT_OK;
    HIP_TRY(ctx, hipMemsetAsync(dptr, 0, bytes, ctx->stream));
    return SPLAT_OK;
}

int splat_buf_copy(splat_ctx *ctx, void *dst_dptr, const void *src_dptr, size_t bytes) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, bytes == 0 || (dst_dptr && src_dptr));
    if (bytes == 0) return SPLAT_OK;
    HIP_TRY(ctx, hipMemcpyAsync(dst_dptr, src_dptr, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return SPLAT_OK;
}

} // extern "C"
