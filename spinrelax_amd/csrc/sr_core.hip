// sr_core.hip -- context, device memory, stream and HIP-event timing of libspinrelax_hip.so
#include "sr_internal.h"
#include <cstring>

static thread_local char g_err[1024] = "";

void sr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void *sr_workspace(sr_ctx *ctx, int slot, size_t bytes)
{
    if (slot < 0 || slot >= SR_NSLOTS) { sr_set_error("bad workspace slot %d", slot); return nullptr; }
    if (bytes == 0) bytes = 16;
    if (ctx->slot_bytes[slot] >= bytes) return ctx->slot[slot];
    if (ctx->slot[slot]) {
        // One context is driven from several streams (spinrelax_amd/pipeline.py: main, auxiliary, one per batch in
        // flight), so work that still reads the old buffer may sit on ANY of them: wait for the whole device, not for
        // the stream that happens to be current.  Growth is rare (sizes repeat from batch to batch).
        (void)hipDeviceSynchronize();
        (void)hipFree(ctx->slot[slot]);
        ctx->slot[slot] = nullptr;
        ctx->slot_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 8;            // 12.5 % head-room so near-equal sizes do not realloc
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        sr_set_error("hipMalloc(%zu bytes) for workspace slot %d failed: %s", want, slot, hipGetErrorString(e));
        return nullptr;
    }
    ctx->slot[slot] = p;
    ctx->slot_bytes[slot] = want;
    return p;
}

int sr_grant_lds(sr_ctx *ctx, int kid, const void *func, size_t bytes)
{
    static int mu = 0;                         // spin lock (the table is touched a handful of times per process)
    static size_t granted[64][8];              // [device][kernel family], largest size granted in this process
    if (ctx->device < 0 || ctx->device >= 64 || kid < 0 || kid >= 8) {
        hipError_t e_ = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e_ != hipSuccess) { sr_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) -> %s", bytes, hipGetErrorString(e_)); return -100 - (int)e_; }
        return 0;
    }
    while (__atomic_exchange_n(&mu, 1, __ATOMIC_ACQUIRE)) {}
    int rc = 0;
    if (bytes > granted[ctx->device][kid]) {
        hipError_t e_ = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e_ != hipSuccess) {
            sr_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) -> %s", bytes, hipGetErrorString(e_));
            rc = -100 - (int)e_;
        } else {
            granted[ctx->device][kid] = bytes;
        }
    }
    __atomic_store_n(&mu, 0, __ATOMIC_RELEASE);
    return rc;
}

extern "C" {

int sr_abi_version(void) { return SR_ABI_VERSION; }

#ifndef SR_BUILD_ID
#define SR_BUILD_ID "unknown"
#endif
const char *sr_build_id(void) { return SR_BUILD_ID; }

const char *sr_last_error(void) { return g_err; }

sr_ctx *sr_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        sr_set_error("no HIP device available (%s); libspinrelax_hip has no CPU fallback",
                     e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) { sr_set_error("device %d out of range (0..%d)", device, n - 1); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { sr_set_error("hipSetDevice(%d) failed", device); return nullptr; }
    sr_ctx *ctx = new sr_ctx();
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    ctx->stream = nullptr;
    ctx->fit_waves = 2;
    ctx->fit_lds = 1;
    ctx->fit_geo = 1;
    ctx->ct_fft = 3;
    ctx->ct_traceless = 0;
    ctx->fft_table_ready = 0;
    ctx->fft32_table_ready = 0;
    if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) {
        sr_set_error("hipGetDeviceProperties failed");
        delete ctx;
        return nullptr;
    }
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        sr_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, ctx->prop.gcnArchName);
        delete ctx;
        return nullptr;
    }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        sr_set_error("hipEventCreate failed");
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void sr_destroy(sr_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();      // the work areas may be in use on any stream the caller has driven the context from
    for (int i = 0; i < SR_NSLOTS; ++i)
        if (ctx->slot[i]) (void)hipFree(ctx->slot[i]);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    for (int i = 0; i < 2; ++i)
        if (ctx->stage[i]) {
            (void)hipHostFree(ctx->stage[i]);
            (void)hipEventDestroy(ctx->stage_ev[i]);
        }
    delete ctx;
}

int sr_set_option(sr_ctx *ctx, const char *name, int value)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(name != nullptr, -2, "sr_set_option: name is NULL");
    if (!strcmp(name, "fit_waves")) {
        SR_REQUIRE(value == 1 || value == 2 || value == 4, -3, "sr_set_option: fit_waves must be 1, 2 or 4");
        ctx->fit_waves = value;
        return 0;
    }
    if (!strcmp(name, "ct_fft")) {
        SR_REQUIRE(value >= 0 && value <= 4, -3, "sr_set_option: ct_fft must be 0 .. 4");
        ctx->ct_fft = value;
        return 0;
    }
    if (!strcmp(name, "ct_wg_per_cu")) {
        SR_REQUIRE(value >= 0 && value <= 8, -3, "sr_set_option: ct_wg_per_cu must be 0 .. 8");
        ctx->ct_wg_per_cu = value;
        return 0;
    }
    if (!strcmp(name, "ct_traceless")) {
        SR_REQUIRE(value == 0 || value == 1, -3, "sr_set_option: ct_traceless must be 0 or 1");
        ctx->ct_traceless = value;
        return 0;
    }
    if (!strcmp(name, "fit_geo")) {
        SR_REQUIRE(value == 0 || value == 1, -3, "sr_set_option: fit_geo must be 0 or 1");
        ctx->fit_geo = value;
        return 0;
    }
    if (!strcmp(name, "fit_lds")) {
        SR_REQUIRE(value == 0 || value == 1, -3, "sr_set_option: fit_lds must be 0 or 1");
        ctx->fit_lds = value;
        return 0;
    }
    sr_set_error("sr_set_option: unknown option '%s'", name);
    return -3;
}

int sr_set_stream(sr_ctx *ctx, void *hip_stream)
{
    SR_CHECK_CTX(ctx);
    ctx->stream = (hipStream_t)hip_stream;
    return 0;
}

int sr_stream_create(sr_ctx *ctx, const uint32_t *cu_mask, int n_words, int priority, void **stream_out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(stream_out != nullptr, -2, "sr_stream_create: stream_out is NULL");
    SR_REQUIRE(n_words >= 0 && n_words <= 32, -2, "sr_stream_create: n_words out of range");
    hipStream_t st = nullptr;
    if (cu_mask && n_words > 0) {
        bool any = false;
        for (int i = 0; i < n_words; ++i) any = any || cu_mask[i] != 0u;
        SR_REQUIRE(any, -2, "sr_stream_create: empty CU mask");
        SR_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask));
    } else {
        SR_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority));
    }
    *stream_out = (void *)st;
    return 0;
}

int sr_stream_destroy(sr_ctx *ctx, void *hip_stream)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(hip_stream != nullptr, -2, "sr_stream_destroy: NULL stream");
    if (ctx->stream == (hipStream_t)hip_stream) ctx->stream = nullptr;
    SR_HIP(hipStreamDestroy((hipStream_t)hip_stream));
    return 0;
}

// ---- signals: a stream waits for a value that a KERNEL writes (hipStreamWaitValue32 on signal memory) ----
uint32_t *sr_signal_alloc(sr_ctx *ctx)
{
    if (!ctx) { sr_set_error("sr_signal_alloc: NULL context"); return nullptr; }
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) != hipSuccess || !can) {
        sr_set_error("sr_signal_alloc: the device cannot wait on memory values from a stream");
        return nullptr;
    }
    uint32_t *p = nullptr;
    if (hipExtMallocWithFlags((void **)&p, 8, hipMallocSignalMemory) != hipSuccess || !p) {
        sr_set_error("sr_signal_alloc: hipExtMallocWithFlags(hipMallocSignalMemory) failed");
        return nullptr;
    }
    p[0] = 0u;
    p[1] = 0u;
    return p;
}

// The ordering rule of signals, enforced: the RELEASE of a value must have been SUBMITTED (sr_stream_write_signal; a kernel-side
// release is always paired with one behind its launch) before a wait for that value is queued.  Streams of one priority are
// multiplexed onto a few hardware queues; a wait queued ahead of its own release on a shared queue sits in front of it and never
// ends (seen in round 4: a test queued the wait first and hung the box until the time limit).  The library keeps, per signal,
// the largest value whose release has been submitted and refuses a wait for more.
static int g_sig_mu = 0;
static uint32_t *g_sig_ptr[64];
static uint32_t g_sig_submitted[64];
static int g_sig_n = 0;
static void sig_lock() { while (__atomic_exchange_n(&g_sig_mu, 1, __ATOMIC_ACQUIRE)) { } }
static void sig_unlock() { __atomic_store_n(&g_sig_mu, 0, __ATOMIC_RELEASE); }
static int sig_slot(uint32_t *sig, bool create)          // under the lock
{
    for (int i = 0; i < g_sig_n; ++i)
        if (g_sig_ptr[i] == sig) return i;
    if (!create || g_sig_n >= 64) return -1;
    g_sig_ptr[g_sig_n] = sig;
    g_sig_submitted[g_sig_n] = 0u;
    return g_sig_n++;
}

int sr_signal_free(sr_ctx *ctx, uint32_t *sig)
{
    SR_CHECK_CTX(ctx);
    if (sig) {
        sig_lock();
        const int i = sig_slot(sig, false);
        if (i >= 0) {
            g_sig_ptr[i] = g_sig_ptr[g_sig_n - 1];
            g_sig_submitted[i] = g_sig_submitted[g_sig_n - 1];
            --g_sig_n;
        }
        sig_unlock();
        SR_HIP(hipFree(sig));
    }
    return 0;
}

int sr_stream_wait_signal(sr_ctx *ctx, uint32_t *sig, uint32_t value)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(sig != nullptr, -2, "sr_stream_wait_signal: NULL signal");
    sig_lock();
    const int i = sig_slot(sig, false);
    const uint32_t have = i >= 0 ? g_sig_submitted[i] : 0u;
    sig_unlock();
    SR_REQUIRE(value == 0u || have >= value, -6,
               "sr_stream_wait_signal: no release of value %u has been submitted for this signal (largest submitted: %u). Submit the "
               "releasing launch and its sr_stream_write_signal BEFORE the wait: a wait queued first can share a hardware queue with "
               "its own release and never end", value, have);
    SR_HIP(hipStreamWaitValue32(ctx->stream, sig, value, hipStreamWaitValueGte, 0xFFFFFFFFu));
    return 0;
}

int sr_stream_write_signal(sr_ctx *ctx, uint32_t *sig, uint32_t value)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(sig != nullptr, -2, "sr_stream_write_signal: NULL signal");
    SR_HIP(hipStreamWriteValue32(ctx->stream, sig, value, 0));
    sig_lock();
    const int i = sig_slot(sig, true);
    if (i >= 0 && value > g_sig_submitted[i]) g_sig_submitted[i] = value;
    sig_unlock();
    SR_REQUIRE(i >= 0, -4, "sr_stream_write_signal: more than 64 live signals");
    return 0;
}

int sr_sync(sr_ctx *ctx)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_device_info(sr_ctx *ctx, int *n_cu, int64_t *hbm_bytes, int *lds_per_cu, char *name, int name_len)
{
    SR_CHECK_CTX(ctx);
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    if (lds_per_cu) *lds_per_cu = (int)ctx->prop.maxSharedMemoryPerMultiProcessor;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    return 0;
}

void *sr_malloc(sr_ctx *ctx, size_t bytes)
{
    if (!ctx) { sr_set_error("null sr_ctx"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sr_set_error("hipSetDevice failed"); return nullptr; }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) { sr_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

int sr_free(sr_ctx *ctx, void *p)
{
    SR_CHECK_CTX(ctx);
    if (p) SR_HIP(hipFree(p));
    return 0;
}

int sr_memcpy_h2d(sr_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_memcpy_d2h(sr_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_memcpy_d2h_async(sr_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(dst && src, -2, "sr_memcpy_d2h_async: null pointer");
    SR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

int sr_memcpy_h2d_async(sr_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(dst && src, -2, "sr_memcpy_h2d_async: null pointer");
    SR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

void *sr_host_alloc(sr_ctx *ctx, size_t bytes)
{
    if (!ctx) { sr_set_error("null sr_ctx"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sr_set_error("hipSetDevice failed"); return nullptr; }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) { sr_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

int sr_host_free(sr_ctx *ctx, void *p)
{
    SR_CHECK_CTX(ctx);
    if (p) SR_HIP(hipHostFree(p));
    return 0;
}

int sr_device_sync(sr_ctx *ctx)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipDeviceSynchronize());
    return 0;
}

int sr_memset(sr_ctx *ctx, void *dst, int value, size_t bytes)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return 0;
}

int sr_timer_start(sr_ctx *ctx)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return 0;
}

int sr_timer_stop_ms(sr_ctx *ctx, float *ms)
{
    SR_CHECK_CTX(ctx);
    SR_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    SR_HIP(hipEventSynchronize(ctx->ev1));
    float t = 0.f;
    SR_HIP(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    if (ms) *ms = t;
    return 0;
}

}  // extern "C"
