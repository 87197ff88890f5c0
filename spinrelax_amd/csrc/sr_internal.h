// sr_internal.h -- shared declarations of libspinrelax_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/spinrelax_hip.h"

#define SR_NSLOTS 12

struct sr_ctx {
    int device;
    hipStream_t stream;
    hipDeviceProp_t prop;
    hipEvent_t ev0, ev1;
    // growable device workspaces, one per purpose, reused across calls (no hipMalloc in steady state)
    void *slot[SR_NSLOTS];
    size_t slot_bytes[SR_NSLOTS];
};

enum {
    SR_WS_VECS = 0,     // staged host vectors (frame-major)
    SR_WS_SOA,          // packed planes
    SR_WS_PSUM,         // C(t) raw sums
    SR_WS_OUT0, SR_WS_OUT1, SR_WS_OUT2, SR_WS_OUT3,
    SR_WS_IN0, SR_WS_IN1, SR_WS_IN2, SR_WS_IN3,
    SR_WS_MISC
};

void sr_set_error(const char *fmt, ...);
void *sr_workspace(sr_ctx *ctx, int slot, size_t bytes);   // returns NULL (error set) on failure

#define SR_HIP(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            sr_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return -100 - (int)e_;                                                          \
        }                                                                                   \
    } while (0)

#define SR_CHECK_CTX(ctx)                                                                   \
    do {                                                                                    \
        if (!(ctx)) { sr_set_error("null sr_ctx"); return -1; }                            \
        SR_HIP(hipSetDevice((ctx)->device));                                                \
    } while (0)

#define SR_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                    \
        if (!(cond)) { sr_set_error(__VA_ARGS__); return (code); }                          \
    } while (0)

// largest LDS allocation one workgroup may ask for (160 KiB on gfx950)
static inline size_t sr_lds_limit(const sr_ctx *ctx)
{
    size_t a = ctx->prop.maxSharedMemoryPerMultiProcessor, b = ctx->prop.sharedMemPerBlock;
    return a > b ? a : b;
}

static inline int64_t sr_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
