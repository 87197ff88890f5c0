// sr_internal.h -- shared declarations of libspinrelax_hip.so (not part of the public ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "../../include/spinrelax_hip.h"

#define SR_NSLOTS 15

struct sr_ctx {
    int device;
    hipStream_t stream;
    hipDeviceProp_t prop;
    hipEvent_t ev0, ev1;
    // growable device workspaces, one per purpose, reused across calls (no hipMalloc in steady state)
    void *slot[SR_NSLOTS];
    size_t slot_bytes[SR_NSLOTS];
    // tuning (sr_set_option)
    int fit_waves;      // waves per residue in the model-order search: 1, 2 or 4
    int fit_geo;        // 1 (default): on a uniform time grid the fit kernels form exp(-t/tau) of a thread's points by multiplication
                        // (sr_fit.hip, Residue::stage); 0: exp() per point whatever the grid
    int fit_lds;        // 1: stage t, y, 1/sigma of a residue in LDS when it fits; 0: read them from global memory
    int ct_fft;         // kernel 1 when the chunk length allows: 3 (default) = float32 real-input FFT (k_ct_rfft32) for 4096 < F + L <=
                        // 8192, the float64 complex FFT below; 4 = float32 transforms for every 1024 < F + L <= 8192;
                        // 2 = the float64 real-input FFT (k_ct_rfft) for 4096 < F + L <= 8192;
                        // 1 = complex float64 FFT formulation (k_ct_fft) everywhere, 0 = always the direct kernel
    int ct_wg_per_cu;   // k_ct_rfft32: at most this many workgroups per CU (0 = as many as fit: 4 for M = 6144); see sr_ct32.hip
    int ct_traceless;   // 1: k_ct_rfft<12> in its traceless five-signal form (faster alone, slower inside the pipeline: default 0)
    int fft_table_ready;
    int fft32_table_ready;
    // strided host -> device copies of bond vectors (sr_vectors.hip): two pinned staging buffers, and what went through them
    void *stage[2];
    hipEvent_t stage_ev[2];
    int stage_busy[2];
    unsigned long long h2d_bytes, h2d_calls;
};

enum { SR_K_CT1 = 0, SR_K_CT4, SR_K_VECHIST, SR_K_DQ, SR_K_MISC };

enum {
    SR_WS_VECS = 0,     // staged host vectors (frame-major)
    SR_WS_SOA,          // packed planes
    SR_WS_PSUM,         // C(t) raw sums
    SR_WS_OUT0, SR_WS_OUT1, SR_WS_OUT2, SR_WS_OUT3,
    SR_WS_IN0, SR_WS_IN1, SR_WS_IN2, SR_WS_IN3,
    SR_WS_MISC,
    SR_WS_FIT,          // residual work space of the fit kernel (when the caller passes none)
    SR_WS_FFT,          // twiddle table of the FFT formulation of kernel 1
    SR_WS_FFT32         // tables of its float32 form (sr_ct32.hip)
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

// allow `func` to be launched with `bytes` of dynamic LDS on this context's device.  hipFuncSetAttribute is a per-DEVICE
// function attribute: the largest size granted so far is remembered per (device, kernel family) for the whole PROCESS
// (monotonic, under a mutex; sr_core.hip), so two contexts on one device can never lower each other's grant.
int sr_grant_lds(sr_ctx *ctx, int kid, const void *func, size_t bytes);

static inline int64_t sr_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// sr_ct32.hip: the float32 real-input FFT form of kernel 1 (raw lag sums per chunk, like k_ct_rfft); the chunk starts may be null
int sr_launch_ct_rfft32(sr_ctx *ctx, const float *soa, int64_t Npad, const int64_t *cs_host, const int64_t *cs_dev, double *psum,
                        int R, int F, int L, int Lp, int64_t series);

#ifdef __HIPCC__
// ---- wave-level float64 sum on the VALU only (DPP + readlane), no LDS round trips -----------------
// __shfl_xor on a double compiles to two ds_bpermute_b32 per step (LDS crossbar, ~100 cycles of
// dependent latency each); the fit kernel reduces 54 values per Jacobian and was parked on those waits
// half of its life (rocprofv3: SQ_WAIT_ANY 51 %).  DPP moves are ordinary VALU instructions.
__device__ __forceinline__ double sr_dpp_f64(double v, const int ctrl_sel)
{
    union { double d; int i[2]; } a, b;
    a.d = v;
    switch (ctrl_sel) {
        case 0:  // quad_perm [1,0,3,2]
            b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0xB1, 0xF, 0xF, true);
            b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0xB1, 0xF, 0xF, true);
            break;
        case 1:  // quad_perm [2,3,0,1]
            b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x4E, 0xF, 0xF, true);
            b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x4E, 0xF, 0xF, true);
            break;
        case 2:  // row_half_mirror
            b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x141, 0xF, 0xF, true);
            b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x141, 0xF, 0xF, true);
            break;
        default:  // row_mirror
            b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x140, 0xF, 0xF, true);
            b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x140, 0xF, 0xF, true);
            break;
    }
    return b.d;
}

__device__ __forceinline__ double sr_readlane_f64(double v, int lane)
{
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    b.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return b.d;
}

// sum over the 64 lanes of a wave, result identical in every lane; fixed association order
__device__ __forceinline__ double sr_wave_sum_f64(double v)
{
    v += sr_dpp_f64(v, 0);
    v += sr_dpp_f64(v, 1);
    v += sr_dpp_f64(v, 2);
    v += sr_dpp_f64(v, 3);            // every lane of a 16-lane row holds the row sum
    const double r0 = sr_readlane_f64(v, 0), r1 = sr_readlane_f64(v, 16);
    const double r2 = sr_readlane_f64(v, 32), r3 = sr_readlane_f64(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// ---- many sums at once: NV values per lane -> lane L ends up with the 64-lane total of value bitrev6(L) ------------
// Reducing NV values one by one costs NV x (4 DPP steps + 8 readlanes + 3 adds) ~ 25 instructions each; the fit kernel's
// Jacobian reduces 54 (J^T J and J^T f at n = 9): 1 350 instructions per wave, a third of the Jacobian pass itself.
// Here every level HALVES the number of live registers instead: for a pair of values (a, b), the lanes whose level bit is
// 0 keep a and receive the partner lane's a, the others keep b and receive the partner's b -- one exchange and one add
// turn two registers into one.  54 -> 27 -> 14 -> 7 -> 4 -> 2 -> 1 registers: ~220 instructions.
//   level 1, 2: lane bits 5 and 4 with v_permlane32_swap / v_permlane16_swap (gfx950: swap the upper half / the odd
//               rows of one register with the lower half / the even rows of the other: 2 instructions per double)
//   level 3, 4: bits 3 and 2 inside a 16-lane row: row_mirror / row_half_mirror DPP moves under bank masks
//   level 5, 6: bits 1 and 0 inside a quad: quad_perm DPP moves + per-lane selects
// Association order (fixed, the same for every value): lanes are paired l <-> l+32, then rows r <-> r^1, then i <-> 15-i
// inside a row, i <-> 7-i inside its halves, i <-> 3-i inside its quads, finally neighbours.  A missing partner value
// (odd count) is 0, which adds exactly.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ int sr_dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK_MASK, false);
}

template <int LEVEL>
__device__ __forceinline__ double sr_reduce_pair(double a, double b, int lane)
{
    union U { double d; int i[2]; unsigned u[2]; };
    U ua, ub, k, t;
    ua.d = a; ub.d = b;
    if (LEVEL == 1) {            // after the swap: ua = [a.lo32, b.lo32], ub = [a.hi32, b.hi32]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            auto r = __builtin_amdgcn_permlane32_swap(ua.u[h], ub.u[h], false, false);
            ua.u[h] = r[0]; ub.u[h] = r[1];
        }
        return ua.d + ub.d;
    } else if (LEVEL == 2) {     // rows: ua = [a0, b0, a2, b2], ub = [a1, b1, a3, b3]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            auto r = __builtin_amdgcn_permlane16_swap(ua.u[h], ub.u[h], false, false);
            ua.u[h] = r[0]; ub.u[h] = r[1];
        }
        return ua.d + ub.d;
    } else if (LEVEL == 3 || LEVEL == 4) {
        constexpr int ctrl = LEVEL == 3 ? 0x140 : 0x141;            // row_mirror / row_half_mirror
        constexpr int m0 = LEVEL == 3 ? 0x3 : 0x5, m1 = LEVEL == 3 ? 0xC : 0xA;   // banks whose level bit is 0 / 1
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            t.i[h] = sr_dpp_i32<ctrl, m0>(0, ua.i[h]);               // bit 0 lanes: the partner's a
            t.i[h] = sr_dpp_i32<ctrl, m1>(t.i[h], ub.i[h]);          // bit 1 lanes: the partner's b
            k.i[h] = sr_dpp_i32<0xE4, m1>(ua.i[h], ub.i[h]);         // keep a (bit 0 lanes) or b (bit 1 lanes)
        }
        return k.d + t.d;
    } else {
        constexpr int ctrl = LEVEL == 5 ? 0x1B : 0xB1;               // quad_perm [3,2,1,0] / [1,0,3,2]
        const bool bit = (lane & (LEVEL == 5 ? 2 : 1)) != 0;
        U s;
        k.d = bit ? b : a;
        s.d = bit ? a : b;
#pragma unroll
        for (int h = 0; h < 2; ++h) t.i[h] = sr_dpp_i32<ctrl, 0xF>(0, s.i[h]);
        return k.d + t.d;
    }
}

template <int LEVEL, int N>
__device__ __forceinline__ void sr_reduce_level(double *v, int lane)
{
#pragma unroll
    for (int m = 0; m < (N + 1) / 2; ++m) v[m] = sr_reduce_pair<LEVEL>(v[2 * m], 2 * m + 1 < N ? v[2 * m + 1] : 0.0, lane);
}

// v[0 .. NV) per lane, NV <= 64 (destroyed); returns the total of value sr_reduced_index(lane) (garbage when that index
// is >= NV)
template <int NV>
__device__ __forceinline__ double sr_wave_sum_many_f64(double *v, int lane)
{
    constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2;
    sr_reduce_level<1, NV>(v, lane);
    sr_reduce_level<2, N1>(v, lane);
    sr_reduce_level<3, N2>(v, lane);
    sr_reduce_level<4, N3>(v, lane);
    sr_reduce_level<5, N4>(v, lane);
    sr_reduce_level<6, N5>(v, lane);
    return v[0];
}
__device__ __forceinline__ int sr_reduced_index(int lane) { return (int)(__builtin_bitreverse32((unsigned)lane) >> 26); }
#endif
