// sr_ct.hip -- kernel 0 (frame-major -> per-vector planes) and kernel 1 (Palmer-chunked P2
// autocorrelation) for gfx950.
//
// Reference semantics: calculate_Ct_Palmer, calculate-Ct-from-traj.py:200-238 (see include/spinrelax_hip.h).
//
// Kernel 1 exists in two formulations that produce the same raw sums S[lag] = sum_j (u(j).u(j+lag))^2 per chunk:
//   k_ct_fft     float64 Wiener-Khinchin (six autocorrelations by FFT, the whole transform in LDS): production path for
//                1024 < F + L <= 8192, ~4 % of the direct flop count, accurate to 1e-15 (see further down);
//   k_ct_palmer  direct shifted products in float32 (or float64 in validation mode): every other chunk length.
//
// Direct kernel design:
//   * one workgroup stages ONE (chunk r, vector v) time series of F frames into LDS as three float
//     planes (x, y, z) -- 48 KB at F = 4096, so three workgroups share a CU's 160 KB;
//   * the (j, lag) plane is cut into lag blocks of 128 lags; a wave owns a lag block, its 64 lanes
//     are 4 j-strips x 16 lag-lanes, each lag-lane owns 8 consecutive lags.  Per step a lane needs
//     8 a-values and a 16-frame b window per component and issues 8x8x4 = 256 FMAs (3 for u(j).u(j+lag), 1 for
//     the square-accumulate).  Consecutive steps' b windows overlap by 8 frames: the window is kept as two
//     8-frame halves that swap roles, so a step reads 6 + 6 = 12 ds_read_b128 (18 without the rotation);
//   * LDS layout is "chunk-parity split, xyz-interleaved": 16-byte chunk c (4 frames of one component) lives
//     in half (c & 1) at slot (c >> 1); a slot is 48 bytes = [x-chunk | y-chunk | z-chunk].  Lag-lanes whose
//     windows start 8 floats (2 chunks) apart therefore read slots 48 bytes apart -- 3*l mod 16 is a
//     permutation, so the 16 lanes of a ds_read_b128 group hit 16 different 16-byte bank groups -- and all
//     18 reads of a step use immediate offsets from four base registers;
//   * the 64 lanes are mapped to (strip, lag-lane) along the hardware's ds_read_b128 lane groups
//     {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,44-47,52-59} {36-43,48-51,60-63} (MI355X_MICROARCH.md,
//     LDS): every group belongs to ONE strip, so its a-window read is a broadcast and its b-window reads are
//     conflict-free for any strip length;
//   * partial sums: float32, 4 independent accumulators per lag, at most 16 terms each, folded into
//     float64 every 8 steps; strips are combined with two float64 wave shuffles; no atomics;
//   * lags that do not fill a 128-lag block (for F = 4096 only lag 2048) and the validation mode run
//     through a simple float64 path in the same launch.
#include "sr_internal.h"

namespace {

constexpr int kLagBlock = 128;     // lags per wave pass
constexpr int kLagsPerLane = 8;
constexpr int kJT = 8;             // j values per lane step
constexpr int kFlush = 8;          // lane steps between float32 -> float64 folds
constexpr float kCenter = 8.0f;    // accumulators start at -kCenter so the <=16 terms (each in [0,1]) keep
                                   // the running float32 sum near zero: halves the accumulation rounding
constexpr int kPad = 192;          // zero padding behind the series (max overshoot of a window: 190)

__host__ __device__ inline int64_t ct_Fp(int64_t F)
{
    // smallest Fp >= F + kPad with Fp % 64 == 32 (so the two parity halves are 16 banks apart)
    int64_t x = F + kPad;
    int64_t base = (x / 64) * 64 + 32;
    if (base < x) base += 64;
    return base;
}

// float index of frame e, component comp in the interleaved parity-split layout; Hf = floats per half
__device__ __forceinline__ int lds_pos(int e, int comp, int Hf)
{
    const int c = e >> 2;
    return (c & 1) * Hf + (c >> 1) * 12 + comp * 4 + (e & 3);
}

// lane -> (strip g, lag-lane l16) following the ds_read_b128 lane groups, and back
__device__ __forceinline__ void lane_to_strip(int lane, int &g, int &l16)
{
    const int h = lane >> 5, m = lane & 31;
    const bool inA = (m < 4) || (m >= 12 && m < 16) || (m >= 20 && m < 28);
    g = 2 * h + (inA ? 0 : 1);
    if (inA) l16 = m < 4 ? m : (m < 16 ? m - 8 : m - 12);
    else l16 = m < 12 ? m - 4 : (m < 20 ? m - 8 : m - 16);
}
__device__ __forceinline__ int strip_to_lane(int g, int l16)
{
    const int h = g >> 1;
    int m;
    if ((g & 1) == 0) m = l16 < 4 ? l16 : (l16 < 8 ? l16 + 8 : l16 + 12);
    else m = l16 < 8 ? l16 + 4 : (l16 < 12 ? l16 + 8 : l16 + 16);
    return 32 * h + m;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// kernel 0: (N, Vtot, 3) float32 -> planes soa[(v*3+c)*Npad + n], zero for n in [N, Npad)
// ------------------------------------------------------------------------------------------
constexpr int kPackFrames = 64;
constexpr int kPackVecs = 32;

// The general form (any number of vectors, any alignment): a 64-frame x 32-vector tile transposed through LDS.
// 16-byte accesses on both sides when it can: a frame's 32 vectors are 96 consecutive floats (24 float4 when the row start is
// 16-byte aligned, i.e. Vtot*3 and (v0+vb)*3 multiples of 4), a plane row of 64 frames is 16 float4.
__global__ __launch_bounds__(256) void k_pack_soa_ragged(const float *__restrict__ vecs, int64_t N, int64_t Vtot,
                                                         int64_t v0, int64_t nV, float *__restrict__ soa, int64_t Npad)
{
    __shared__ float tile[kPackVecs * 3][kPackFrames + 1];
    const int64_t n0 = (int64_t)blockIdx.x * kPackFrames;
    const int64_t vb = (int64_t)blockIdx.y * kPackVecs;
    const int nvec = (int)min((int64_t)kPackVecs, nV - vb);
    const int row = nvec * 3;
    const int tid = threadIdx.x;
    const bool vec4 = ((Vtot * 3) & 3) == 0 && (((v0 + vb) * 3) & 3) == 0 && (row & 3) == 0;
    if (vec4) {
        const int q4 = row >> 2;                              // float4 per frame
        for (int idx = tid; idx < kPackFrames * q4; idx += 256) {
            const int n = idx / q4, q = idx - n * q4;
            const int64_t fr = n0 + n;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fr < N) val = *reinterpret_cast<const float4 *>(vecs + (fr * Vtot + v0 + vb) * 3 + 4 * q);
            tile[4 * q + 0][n] = val.x;
            tile[4 * q + 1][n] = val.y;
            tile[4 * q + 2][n] = val.z;
            tile[4 * q + 3][n] = val.w;
        }
    } else {
        for (int idx = tid; idx < kPackFrames * row; idx += 256) {
            const int n = idx / row, k = idx - n * row;
            const int64_t fr = n0 + n;
            float val = 0.f;
            if (fr < N) val = vecs[(fr * Vtot + v0 + vb) * 3 + k];
            tile[k][n] = val;
        }
    }
    __syncthreads();
    // Npad % 4 == 0 and n0 % 64 == 0: every group of four frames is either fully inside the planes or fully outside
    for (int idx = tid; idx < (kPackFrames / 4) * row; idx += 256) {
        const int k = idx / (kPackFrames / 4), n = (idx - k * (kPackFrames / 4)) * 4;
        const int64_t fr = n0 + n;
        if (fr < Npad) {
            const float4 o = make_float4(tile[k][n], tile[k][n + 1], tile[k][n + 2], tile[k][n + 3]);
            *reinterpret_cast<float4 *>(soa + (vb * 3 + k) * Npad + fr) = o;
        }
    }
}

// The production form (whole 32-vector tiles, 16-byte aligned rows: cfg3 / cfg4 and every shard of them): the transposition in
// REGISTERS, no LDS.  A thread owns three 4 x 4 blocks -- four consecutive frames x one 16-byte column of a frame's row (four
// consecutive components) -- twelve 16-byte loads all in flight, then twelve 16-byte stores; the lanes of a wave are 8 columns x 8
// frame groups, so a wave-load reads 8 full 128-byte lines (8 frames) and a wave-store writes 8 full lines (128 B of each of 8
// planes).  Both sides non-temporal: the vectors are read once, the planes are next read by another kernel.  Measured against the
// LDS tile above (round 4, scripts/dev/interference.py, same box each time): alone 0.208-0.223 against 0.229-0.239 ms (5.5-5.9
// TB/s), and what one pack costs 20 back-to-back C(t) launches it runs beside 0.165-0.185 against 0.197-0.207 ms; 20-step
// benchmark 2.27-2.32 against 2.35-2.38 ms per step.  The LDS tile's scattered 4-byte LDS writes and its 54 instructions per 16
// bytes were what the C(t) waves on the same CU paid for.  (An LDS-DMA fill of the same tile: no gain; 16 or 4 columns per wave,
// temporal accesses, four loads in flight instead of twelve: worse or equal.)
constexpr int kPackRegFrames = 128;
__global__ __launch_bounds__(256) void k_pack_soa(const float *__restrict__ vecs, int64_t N, int64_t Vtot,
                                                  int64_t v0, int64_t nV, float *__restrict__ soa, int64_t Npad)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int64_t n0 = (int64_t)blockIdx.x * kPackRegFrames;
    const int64_t vb = (int64_t)blockIdx.y * kPackVecs;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 7, gl = lane >> 3;
    v4f r[3][4];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int blk = it * 4 + wave;                        // 12 wave-blocks: 3 column groups x 4 groups of 8 frame groups
        const int q = (blk % 3) * 8 + ql;                     // 16-byte column of the 96-float row
        const int64_t fr = n0 + 4 * ((blk / 3) * 8 + gl);
#pragma unroll
        for (int j = 0; j < 4; ++j) {                         // no branch around a load: past the end, frame N - 1 again
            const int64_t f = min(fr + j, N - 1);
            r[it][j] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(vecs + (f * Vtot + v0 + vb) * 3 + 4 * q));
        }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int blk = it * 4 + wave;
        const int q = (blk % 3) * 8 + ql;
        const int64_t fr = n0 + 4 * ((blk / 3) * 8 + gl);
        if (fr >= Npad) continue;                             // Npad % 4 == 0: a frame group is inside the planes or outside
        if (fr + 3 >= N) {                                    // frames in [N, Npad): zeros
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (fr + j >= N) r[it][j] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        float *o = soa + (vb * 3 + 4 * q) * Npad + fr;
        __builtin_nontemporal_store(v4f{r[it][0].x, r[it][1].x, r[it][2].x, r[it][3].x}, reinterpret_cast<v4f *>(o));
        __builtin_nontemporal_store(v4f{r[it][0].y, r[it][1].y, r[it][2].y, r[it][3].y}, reinterpret_cast<v4f *>(o + Npad));
        __builtin_nontemporal_store(v4f{r[it][0].z, r[it][1].z, r[it][2].z, r[it][3].z}, reinterpret_cast<v4f *>(o + 2 * Npad));
        __builtin_nontemporal_store(v4f{r[it][0].w, r[it][1].w, r[it][2].w, r[it][3].w}, reinterpret_cast<v4f *>(o + 3 * Npad));
    }
}

// The same transposition with the de-tumbling folded in: every frame's vectors are rotated by that frame's unit
// quaternion (float64, rotate_vector_simd's operation order, transforms3d_supplement.py:270-296) and rounded to the
// float32 the planes hold.  SURVEY.md section 8(f)-1: lab-frame vectors + colvar-qorient in, body-frame C(t) out.
__global__ __launch_bounds__(256) void k_pack_soa_rot(const float *__restrict__ vecs, int64_t N, int64_t Vtot, int64_t v0,
                                                      int64_t nV, const double *__restrict__ quat,
                                                      float *__restrict__ soa, int64_t Npad)
{
#pragma clang fp contract(off)
    __shared__ float tile[kPackVecs * 3][kPackFrames + 1];
    const int64_t n0 = (int64_t)blockIdx.x * kPackFrames;
    const int64_t vb = (int64_t)blockIdx.y * kPackVecs;
    const int nvec = (int)min((int64_t)kPackVecs, nV - vb);
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kPackFrames * nvec; idx += 256) {
        const int n = idx / nvec, k = idx - n * nvec;
        const int64_t fr = n0 + n;
        float ox = 0.f, oy = 0.f, oz = 0.f;
        if (fr < N) {
            const float *p = vecs + (fr * Vtot + v0 + vb + k) * 3;
            const double vx = (double)p[0], vy = (double)p[1], vz = (double)p[2];
            const double qw = quat[fr * 4 + 0], qx = quat[fr * 4 + 1], qy = quat[fr * 4 + 2], qz = quat[fr * 4 + 3];
            const double ax = (qy * vz - qz * vy) + qw * vx;
            const double ay = (qz * vx - qx * vz) + qw * vy;
            const double az = (qx * vy - qy * vx) + qw * vz;
            const double bx = qy * az - qz * ay;
            const double by = qz * ax - qx * az;
            const double bz = qx * ay - qy * ax;
            ox = (float)((bx + bx) + vx);
            oy = (float)((by + by) + vy);
            oz = (float)((bz + bz) + vz);
        }
        tile[k * 3 + 0][n] = ox;
        tile[k * 3 + 1][n] = oy;
        tile[k * 3 + 2][n] = oz;
    }
    __syncthreads();
    const int row = nvec * 3;
    for (int idx = tid; idx < kPackFrames * row; idx += 256) {
        const int k = idx / kPackFrames, n = idx - k * kPackFrames;
        const int64_t fr = n0 + n;
        if (fr < Npad) soa[(vb * 3 + k) * Npad + fr] = tile[k][n];
    }
}

// ------------------------------------------------------------------------------------------
// kernel 1
// ------------------------------------------------------------------------------------------
struct CtArgs {
    const float *soa;
    int64_t Npad;
    const int64_t *chunk_start;   // device, may be null
    double *psum;                 // (nV, R, Lp)
    int R, F, Fp, L, Lp, nslab, mode;
};

#ifndef SR_CT_WAVES_EU
#define SR_CT_WAVES_EU 3
#endif
template <int W>
__global__ __launch_bounds__(W * 64, SR_CT_WAVES_EU) void k_ct_palmer(CtArgs a)
{
    extern __shared__ __align__(16) float lds[];
    // This is the throughput kernel of the pipeline; the fit wavefronts of the previous batch share its SIMDs.  Raised
    // issue priority makes them fill the slots this kernel leaves idle instead of taking turns with it.
    __builtin_amdgcn_s_setprio(3);
    const int Fp = a.Fp, Hf = (Fp >> 3) * 12, F = a.F;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int series = blockIdx.x / a.nslab;
    const int slab = blockIdx.x - series * a.nslab;
    const int v = series / a.R;
    const int r = series - v * a.R;

    // ---- stage the series (coalesced dword loads; zero padding behind frame F) ----
    {
        const int64_t start = a.chunk_start ? a.chunk_start[r] : (int64_t)r * F;
        const float *px = a.soa + ((int64_t)v * 3 + 0) * a.Npad + start;
        const float *py = px + a.Npad;
        const float *pz = py + a.Npad;
        for (int e = tid; e < Fp; e += W * 64) {
            const int p = lds_pos(e, 0, Hf);
            const bool in = e < F;
            lds[p] = in ? px[e] : 0.f;
            lds[p + 4] = in ? py[e] : 0.f;
            lds[p + 8] = in ? pz[e] : 0.f;
        }
    }
    __syncthreads();

    const int NW = a.nslab * W;            // workers (waves) per series
    const int wid = slab * W + wave;
    double *out = a.psum + ((int64_t)v * a.R + r) * a.Lp;
    const int nb = (a.mode == 0) ? (a.L + 1) / kLagBlock : 0;

    // ---- fast path: full lag blocks, serpentine assignment balances the (F - lag) work ----
    int g, l16;
    lane_to_strip(lane, g, l16);
    for (int i = 0; i * NW < nb; ++i) {
        const int k = (i & 1) ? i * NW + (NW - 1 - wid) : i * NW + wid;
        if (k >= nb) continue;
        const int dw = k * kLagBlock;
        const int nj = F - dw;
        const int S = (((nj + 3) >> 2) + 15) & ~15;        // strip length, multiple of 16: an even number of steps
        const int iters = S >> 3;
        const float *pa0 = lds + ((g * S) >> 3) * 12;                              // even chunks of the a window
        const float *pa1 = pa0 + Hf;                                               // odd chunks
        const float *pb0 = lds + ((g * S + dw + kLagsPerLane * l16) >> 3) * 12;    // even chunks of the b window
        const float *pb1 = pb0 + Hf;
        double acc64[kLagsPerLane];
#pragma unroll
        for (int d = 0; d < kLagsPerLane; ++d) acc64[d] = 0.0;

        // The 16-frame b window of a step is [P | Q]: P = its first 8 frames, Q = the next 8.  The following step's
        // window starts 8 frames later, i.e. with this step's Q -- so only ONE new half is read per step and the two
        // halves swap roles (12 instead of 18 ds_read_b128 per 256 FMAs).
        float Px[8], Py[8], Pz[8], Qx[8], Qy[8], Qz[8];
#define SR_CT_LOAD_HALF(HX, HY, HZ, OFF)                                                         \
        {                                                                                        \
            const float4 t0 = *reinterpret_cast<const float4 *>(pb0 + (OFF));                   \
            const float4 t1 = *reinterpret_cast<const float4 *>(pb0 + (OFF) + 4);               \
            const float4 t2 = *reinterpret_cast<const float4 *>(pb0 + (OFF) + 8);               \
            const float4 u0 = *reinterpret_cast<const float4 *>(pb1 + (OFF));                   \
            const float4 u1 = *reinterpret_cast<const float4 *>(pb1 + (OFF) + 4);               \
            const float4 u2 = *reinterpret_cast<const float4 *>(pb1 + (OFF) + 8);               \
            HX[0] = t0.x; HX[1] = t0.y; HX[2] = t0.z; HX[3] = t0.w; HX[4] = u0.x; HX[5] = u0.y; HX[6] = u0.z; HX[7] = u0.w; \
            HY[0] = t1.x; HY[1] = t1.y; HY[2] = t1.z; HY[3] = t1.w; HY[4] = u1.x; HY[5] = u1.y; HY[6] = u1.z; HY[7] = u1.w; \
            HZ[0] = t2.x; HZ[1] = t2.y; HZ[2] = t2.z; HZ[3] = t2.w; HZ[4] = u2.x; HZ[5] = u2.y; HZ[6] = u2.z; HZ[7] = u2.w; \
        }
#define SR_CT_STEP(LX, LY, LZ, HX, HY, HZ)                                                       \
        {                                                                                        \
            float ax[kJT], ay[kJT], az[kJT], bx[16], by[16], bz[16];                             \
            {                                                                                    \
                const float4 tx = *reinterpret_cast<const float4 *>(pa0);                       \
                const float4 ty = *reinterpret_cast<const float4 *>(pa0 + 4);                   \
                const float4 tz = *reinterpret_cast<const float4 *>(pa0 + 8);                   \
                const float4 ux = *reinterpret_cast<const float4 *>(pa1);                       \
                const float4 uy = *reinterpret_cast<const float4 *>(pa1 + 4);                   \
                const float4 uz = *reinterpret_cast<const float4 *>(pa1 + 8);                   \
                ax[0] = tx.x; ax[1] = tx.y; ax[2] = tx.z; ax[3] = tx.w; ax[4] = ux.x; ax[5] = ux.y; ax[6] = ux.z; ax[7] = ux.w; \
                ay[0] = ty.x; ay[1] = ty.y; ay[2] = ty.z; ay[3] = ty.w; ay[4] = uy.x; ay[5] = uy.y; ay[6] = uy.z; ay[7] = uy.w; \
                az[0] = tz.x; az[1] = tz.y; az[2] = tz.z; az[3] = tz.w; az[4] = uz.x; az[5] = uz.y; az[6] = uz.z; az[7] = uz.w; \
            }                                                                                    \
            SR_CT_LOAD_HALF(HX, HY, HZ, 12)                                                      \
            _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                      \
                bx[t] = LX[t]; by[t] = LY[t]; bz[t] = LZ[t];                                     \
                bx[8 + t] = HX[t]; by[8 + t] = HY[t]; bz[8 + t] = HZ[t];                         \
            }                                                                                    \
            _Pragma("unroll") for (int jj = 0; jj < kJT; ++jj) {                                 \
                _Pragma("unroll") for (int d = 0; d < kLagsPerLane; ++d) {                       \
                    float dot = ax[jj] * bx[jj + d];                                             \
                    dot = fmaf(ay[jj], by[jj + d], dot);                                         \
                    dot = fmaf(az[jj], bz[jj + d], dot);                                         \
                    acc[d][jj & 3] = fmaf(dot, dot, acc[d][jj & 3]);                             \
                }                                                                                \
            }                                                                                    \
            pa0 += 12; pa1 += 12; pb0 += 12; pb1 += 12;                                          \
        }
        SR_CT_LOAD_HALF(Px, Py, Pz, 0)
        for (int it0 = 0; it0 < iters; it0 += kFlush) {
            float acc[kLagsPerLane][4];
#pragma unroll
            for (int d = 0; d < kLagsPerLane; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[d][q] = -kCenter;
            const int n = min(kFlush, iters - it0);            // even
            for (int ii = 0; ii < n; ii += 2) {
                SR_CT_STEP(Px, Py, Pz, Qx, Qy, Qz)
                SR_CT_STEP(Qx, Qy, Qz, Px, Py, Pz)
            }
#pragma unroll
            for (int d = 0; d < kLagsPerLane; ++d) {
                const float s = (acc[d][0] + acc[d][1]) + (acc[d][2] + acc[d][3]);
                acc64[d] += (double)s + 4.0 * (double)kCenter;
            }
        }
#undef SR_CT_STEP
#undef SR_CT_LOAD_HALF
        // combine the 4 j strips: the lanes of strip 0 collect the partial sums of strips 1..3
        {
            const int s1 = strip_to_lane(1, l16), s2 = strip_to_lane(2, l16), s3 = strip_to_lane(3, l16);
#pragma unroll
            for (int d = 0; d < kLagsPerLane; ++d) {
                const double v0 = acc64[d];
                const double v1 = __shfl(v0, s1, 64), v2 = __shfl(v0, s2, 64), v3 = __shfl(v0, s3, 64);
                acc64[d] = (v0 + v1) + (v2 + v3);
            }
        }
        if (g == 0) {
            double *o = out + dw + kLagsPerLane * l16;
#pragma unroll
            for (int d = 0; d < kLagsPerLane; ++d) o[d] = acc64[d];
        }
    }

    // ---- float64 path: remaining lags (and every lag in validation mode) ----
    {
        int lo = nb * kLagBlock;
        if (lo < 1) lo = 1;
        for (int d = lo + wid; d <= a.L; d += NW) {
            double s = 0.0;
            for (int j = lane; j + d < F; j += 64) {
                const int pa = lds_pos(j, 0, Hf), pb = lds_pos(j + d, 0, Hf);
                const double x = (double)lds[pa] * (double)lds[pb] + (double)lds[pa + 4] * (double)lds[pb + 4] +
                                 (double)lds[pa + 8] * (double)lds[pb + 8];
                s += x * x;
            }
            s = wave_sum_f64(s);
            if (lane == 0) out[d] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// kernel 1, FFT formulation
// ------------------------------------------------------------------------------------------
// S[lag] = sum_j (u(j).u(j+lag))^2 is the sum of six ordinary autocorrelations: with
//   a = (x^2, y^2, z^2, xy, xz, yz),  (u.u')^2 = a1 a1' + a2 a2' + a3 a3' + 2 (a4 a4' + a5 a5' + a6 a6'),
// so S = IFFT( sum_c w_c |FFT(a_c)|^2 ) on the chunk zero-padded to M >= F + L points (Wiener-Khinchin).  In float64
// this is ~13x fewer operations than the 4 F L / 2 FMAs of the direct kernel at F = 4096, and more accurate (1e-14
// instead of the float32 dot products' 1e-8).  What makes it a one-workgroup-per-series kernel is the 160 KB of LDS:
// a complete 8192-point complex float64 transform (128 KB + padding) stays on the CU.
//
// One workgroup of 256 threads owns one (chunk, vector) series.  M = N1 * 256, N1 = 8, 16 or 32; four-step
// decomposition N1 x 32 x 8 with every small transform in registers:
//   1. thread n2 holds the N1 samples n = n2 + 256 n1, transforms them (radix-2 DIF, constant twiddles), applies
//      the twiddle w_M^(n2 k1);
//   2. exchange through LDS; thread (k1, n2 mod 8) transforms 32 samples n2 = lo + 8 h, twiddle w_256^(lo k2a);
//   3. exchange; thread q transforms the 8 samples of group g = k1 + N1 k2a: X[g + 32 N1 k2b].
// Real signals are transformed in pairs (p + i q); the power spectra come out of Z(k) and conj Z(M-k), exchanged
// through LDS once more.  The weighted power spectrum (real, even) then runs through the same transform; its real
// part / M is S[lag], written where the direct kernel writes (raw sums per chunk; k_ct_finalize is shared).
// LDS addresses are padded (one slot per 8, eight per 256) so that all three access patterns are conflict-free.
struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
    return {fma(a.re, b.re, -(a.im * b.im)), fma(a.re, b.im, a.im * b.re)};
}

// d * exp(-2 pi i e / 32), e a compile-time constant after unrolling
template <int E>
__device__ __forceinline__ cplx mul_w32(cplx d)
{
    if (E == 0) return d;
    if (E == 8) return {d.im, -d.re};
    constexpr double c[16] = {1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476,
                              0.5555702330196023, 0.38268343236508984, 0.19509032201612833, 0.0,
                              -0.1950903220161282, -0.3826834323650897, -0.555570233019602, -0.7071067811865475,
                              -0.8314696123025453, -0.9238795325112867, -0.9807852804032304};
    constexpr double s[16] = {0.0, 0.19509032201612825, 0.3826834323650898, 0.5555702330196022, 0.7071067811865475,
                              0.8314696123025452, 0.9238795325112867, 0.9807852804032304, 1.0, 0.9807852804032304,
                              0.9238795325112867, 0.8314696123025455, 0.7071067811865476, 0.5555702330196022,
                              0.3826834323650899, 0.1950903220161286};
    return {fma(d.re, c[E], d.im * s[E]), fma(d.im, c[E], -(d.re * s[E]))};
}

// d * exp(-2 pi i e / 32) for e = 0..15 known after unrolling (a switch the optimiser folds)
__device__ __forceinline__ cplx mul_w32_rt(cplx d, int e)
{
    switch (e) {
        case 0: return mul_w32<0>(d);
        case 1: return mul_w32<1>(d);
        case 2: return mul_w32<2>(d);
        case 3: return mul_w32<3>(d);
        case 4: return mul_w32<4>(d);
        case 5: return mul_w32<5>(d);
        case 6: return mul_w32<6>(d);
        case 7: return mul_w32<7>(d);
        case 8: return mul_w32<8>(d);
        case 9: return mul_w32<9>(d);
        case 10: return mul_w32<10>(d);
        case 11: return mul_w32<11>(d);
        case 12: return mul_w32<12>(d);
        case 13: return mul_w32<13>(d);
        case 14: return mul_w32<14>(d);
        default: return mul_w32<15>(d);
    }
}

template <int LOGN, int S, int BLK, int J>
struct FftStage {
    __device__ static __forceinline__ void run(cplx *v)
    {
        constexpr int N = 1 << LOGN;
        constexpr int half = N >> (S + 1);
        constexpr int i = BLK * 2 * half + J;
        const cplx a = v[i], b = v[i + half];
        v[i] = {a.re + b.re, a.im + b.im};
        const cplx d = {a.re - b.re, a.im - b.im};
        v[i + half] = mul_w32<((J << S) * (32 / N)) & 15>(d);
        if constexpr (J + 1 < half) FftStage<LOGN, S, BLK, J + 1>::run(v);
        else if constexpr (BLK + 1 < (1 << S)) FftStage<LOGN, S, BLK + 1, 0>::run(v);
        else if constexpr (S + 1 < LOGN) FftStage<LOGN, S + 1, 0, 0>::run(v);
    }
};
// in-register radix-2 decimation-in-frequency transform of N = 2^LOGN <= 32 points; v[p] ends up holding X[rev(p)]
template <int LOGN>
__device__ __forceinline__ void fft_reg(cplx *v)
{
    FftStage<LOGN, 0, 0, 0>::run(v);
}
template <int LOGN>
__host__ __device__ constexpr int bitrev(int p)
{
    int r = 0;
    for (int b = 0; b < LOGN; ++b) r |= ((p >> b) & 1) << (LOGN - 1 - b);
    return r;
}
// d * exp(-2 pi i e / 24), e a compile-time constant
template <int E>
__device__ __forceinline__ cplx mul_w24(cplx d)
{
    if (E == 0) return d;
    if (E == 6) return {d.im, -d.re};
    if (E == 12) return {-d.re, -d.im};
    constexpr double c[15] = {1.0, 0.9659258262890683, 0.8660254037844387, 0.7071067811865476, 0.5000000000000001,
                              0.25881904510252074, 0.0, -0.25881904510252063, -0.4999999999999998, -0.7071067811865475,
                              -0.8660254037844387, -0.9659258262890682, -1.0, -0.9659258262890683, -0.8660254037844388};
    constexpr double s[15] = {0.0, 0.25881904510252074, 0.49999999999999994, 0.7071067811865475, 0.8660254037844386,
                              0.9659258262890683, 1.0, 0.9659258262890683, 0.8660254037844387, 0.7071067811865476,
                              0.49999999999999994, 0.258819045102521, 0.0, -0.2588190451025208, -0.4999999999999997};
    return {fma(d.re, c[E], d.im * s[E]), fma(d.im, c[E], -(d.re * s[E]))};
}

// First stage of the four-step transform: N1 samples per thread -> N1 frequencies k1, in place; v[p] holds X[k1(p)].
template <int N1>
struct Stage1 {
    static constexpr int LOG = N1 == 8 ? 3 : (N1 == 16 ? 4 : 5);
    __host__ __device__ static constexpr int k1(int p) { return bitrev<LOG>(p); }
    __device__ static __forceinline__ void run(cplx *v) { fft_reg<LOG>(v); }
};
// 24 = 3 x 8 (transform length 6144 = F + L for the F = 4096 chunks: a quarter less work than 8192): n1 = 8 a + b,
// k1 = ka + 3 kb; 3-point transforms over a, twiddle w_24^(b ka), 8-point transforms over b
template <int B>
__device__ __forceinline__ void dft3_col(cplx *v, cplx (*y)[8])
{
    constexpr double h = 0.8660254037844386;            // sqrt(3)/2
    const cplx x0 = v[B], x1 = v[8 + B], x2 = v[16 + B];
    const cplx t = {x1.re + x2.re, x1.im + x2.im}, d = {x1.re - x2.re, x1.im - x2.im};
    const cplx m = {fma(-0.5, t.re, x0.re), fma(-0.5, t.im, x0.im)};
    const cplx r = {h * d.im, -h * d.re};                // -i sqrt(3)/2 (x1 - x2)
    y[0][B] = {x0.re + t.re, x0.im + t.im};
    y[1][B] = mul_w24<B>(cplx{m.re + r.re, m.im + r.im});
    y[2][B] = mul_w24<2 * B>(cplx{m.re - r.re, m.im - r.im});
    if constexpr (B + 1 < 8) dft3_col<B + 1>(v, y);
}
template <>
struct Stage1<24> {
    __host__ __device__ static constexpr int k1(int p) { return (p >> 3) + 3 * bitrev<3>(p & 7); }
    __device__ static __forceinline__ void run(cplx *v)
    {
        cplx y[3][8];
        dft3_col<0>(v, y);
#pragma unroll
        for (int ka = 0; ka < 3; ++ka) {
            fft_reg<3>(y[ka]);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[8 * ka + q] = y[ka][q];
        }
    }
};

// v[p] *= base^k(p), k(p) < N: base^k = A[k & 7] * B[k >> 3] with 8 + N/8 powers held in registers (a full table of N
// powers would cost 4 N VGPRs next to the 4 N of the data)
template <int N, class KOF>
__device__ __forceinline__ void apply_twiddles(cplx *v, cplx base)
{
    constexpr int NA = N < 8 ? N : 8, NB = N / 8 > 0 ? N / 8 : 1;
    cplx A[NA], B[NB];
    A[0] = {1.0, 0.0};
#pragma unroll
    for (int k = 1; k < NA; ++k) A[k] = k == 1 ? base : cmul(A[k >> 1], A[k - (k >> 1)]);
    B[0] = {1.0, 0.0};
    if (NB > 1) {
        B[1] = cmul(A[4], A[4]);
#pragma unroll
        for (int k = 2; k < NB; ++k) B[k] = cmul(B[k >> 1], B[k - (k >> 1)]);
    }
#pragma unroll
    for (int p = 0; p < N; ++p) {
        const int k = KOF::k1(p);
        if (k == 0) continue;
        const cplx t = (k >> 3) == 0 ? A[k & 7] : ((k & 7) == 0 ? B[k >> 3] : cmul(A[k & 7], B[k >> 3]));
        v[p] = cmul(v[p], t);
    }
}

// LDS slot of logical element a: one pad slot per 8 elements and eight per 256.  Every access pattern below splits into
// a per-thread part and a compile-time part without carries between them, so each access is `base + immediate`.
__host__ __device__ constexpr int fft_pad(int a) { return a + (a >> 3) + 8 * (a >> 8); }
__host__ __device__ constexpr int fft_lds_slots(int M) { return M + (M >> 3) + 8 * (M >> 8); }

// tab[2t], tab[2t+1] = cos, -sin of 2 pi t / 8192 for t < 1024, followed by the same for 2 pi t / 6144, t < 256 (k_ct_fft);
// then, from complex index 1280, the three 256-entry tables of k_ct_rfft for N1 = 12 and for N1 = 16:
//   w_H^t (H = 256 N1), w_256^t, w_M^t (M = 512 N1)
constexpr int kFftTabDoubles = 2 * (1280 + 2 * 768);
__global__ void k_fft_init_table(double *tab)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double sn, cs;
    if (t < 1280) {
        if (t < 1024) sincospi((double)t / 4096.0, &sn, &cs);
        else sincospi((double)(t - 1024) / 3072.0, &sn, &cs);
    } else if (t < 1280 + 2 * 768) {
        const int u = t - 1280, set = u / 768, j = u - set * 768, which = j >> 8, i = j & 255;
        const double H = set == 0 ? 3072.0 : 4096.0;
        const double len = which == 0 ? H : (which == 1 ? 256.0 : 2.0 * H);
        sincospi(2.0 * (double)i / len, &sn, &cs);
    } else {
        return;
    }
    tab[2 * t] = cs;
    tab[2 * t + 1] = -sn;
}

struct CtFftArgs {
    const float *soa;
    int64_t Npad;
    const int64_t *chunk_start;   // device, may be null
    const double *tab;            // w_8192^t, t < 1024, then w_6144^t, t < 256
    double *psum;                 // (nV, R, Lp)
    int R, F, L, Lp;
};

// one full transform of the thread's N1 samples v[] (natural order, sample n = tid + 256 n1) -> the thread's
// G = N1/8 groups of 8 spectrum values w[j][p] = X[g + 32 N1 rev3(p)], g = tid + 256 j.  Ends with a barrier.
template <int N1>
__device__ __forceinline__ void fft_workgroup(cplx *v, cplx (*w)[8], cplx *lds, const double *__restrict__ tab, int tid)
{
    constexpr int G = N1 / 8;
    constexpr bool kPow2 = (N1 & (N1 - 1)) == 0;
    // step 1: N1-point transforms over n1, twiddle w_M^(n2 k1), to LDS as element k1*256 + n2
    Stage1<N1>::run(v);
    {
        const int ti = kPow2 ? (8192 / (N1 * 256)) * tid : 1024 + tid;      // w_M^tid
        apply_twiddles<N1, Stage1<N1>>(v, cplx{tab[2 * ti], tab[2 * ti + 1]});
        cplx *b = lds + tid + (tid >> 3);
#pragma unroll
        for (int p = 0; p < N1; ++p) b[fft_pad(Stage1<N1>::k1(p) * 256)] = v[p];
    }
    __syncthreads();
    // step 2: thread (k1, lo), active while k1 < N1: 32-point transforms over h (n2 = lo + 8 h), twiddle w_256^(lo k2a)
    cplx u[32];
    const int k1 = tid >> 3, lo = tid & 7;
    const bool act = k1 < N1;
    if (act) {
        const cplx *b = lds + fft_pad(256) * k1 + lo;
#pragma unroll
        for (int h = 0; h < 32; ++h) u[h] = b[9 * h];
        fft_reg<5>(u);
        apply_twiddles<32, Stage1<32>>(u, cplx{tab[2 * (32 * lo)], tab[2 * (32 * lo) + 1]});
    }
    __syncthreads();
    if (act) {
        // element (k1 + N1 k2a)*8 + lo
        if constexpr (kPow2) {
            cplx *b = lds + 9 * k1 + lo;
#pragma unroll
            for (int p = 0; p < 32; ++p) b[fft_pad(8 * N1 * bitrev<5>(p))] = u[p];
        } else {
            // 8 N1 is not a power of two: the per-thread and the constant part of the slot can carry into each other
            const int t = 8 * k1 + lo;
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const int c = 8 * N1 * bitrev<5>(p);
                lds[t + c + ((t + c) >> 3) + 8 * ((t + c) >> 8)] = u[p];
            }
        }
    }
    __syncthreads();
    // step 3: thread q, groups g = q + 256 j: 8-point transforms over lo
    {
        const cplx *b = lds + 9 * tid + 8 * (tid >> 5);
#pragma unroll
        for (int j = 0; j < G; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[j][e] = b[fft_pad(2048 * j) + e];
            fft_reg<3>(w[j]);
        }
    }
    __syncthreads();
}

// HALF: the chunk fills at most 256 NZ samples (NZ = N1/2, or 16 of 24: the F = 4096 case): the thread's samples
// beyond NZ are known to be zero and are not loaded
template <int N1, bool HALF>
__global__ __launch_bounds__(256) void k_ct_fft(CtFftArgs a)
{
    extern __shared__ __align__(16) unsigned char fft_smem[];
    cplx *lds = reinterpret_cast<cplx *>(fft_smem);
    constexpr int M = N1 * 256;
    constexpr int G = N1 / 8;
    constexpr int NZ = HALF ? (N1 == 24 ? 16 : N1 / 2) : N1;
    const int tid = threadIdx.x;
    const int v = blockIdx.x / a.R, r = blockIdx.x - v * a.R;
    const int F = a.F;
    const int64_t start = a.chunk_start ? a.chunk_start[r] : (int64_t)r * F;
    const float *px = a.soa + ((int64_t)v * 3 + 0) * a.Npad + start;
    const float *py = px + a.Npad;
    const float *pz = py + a.Npad;
    cplx *fb = lds + tid + (tid >> 3);                 // frequency / natural order: element tid + 256 j + 32 N1 k'

    double W[G][8];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) W[j][e] = 0.0;

    // three packed pairs: (x^2, y^2) weights 1,1; (z^2, xy) weights 1,2; (xz, yz) weights 2,2.  The samples are
    // re-read from the planes for every pair (L2 hits) rather than kept in 96 registers across the transforms.
#pragma unroll 1
    for (int pair = 0; pair < 3; ++pair) {
        // the samples are re-read for every pair (L2 hits; keeping them in registers across the loop makes the
        // compiler hoist all six products, 384 VGPRs).  All 3 N1 loads are issued before the first use -- with one
        // wave per SIMD a load-use-load-use sequence pays the memory latency N1 times (1.5 ms of 3.6 ms).
        asm volatile("" ::: "memory");
        float xr[N1], yr[N1], zr[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            if (n1 >= NZ) {
                xr[n1] = yr[n1] = zr[n1] = 0.f;
                continue;
            }
            const int n = tid + 256 * n1;
            const bool in = n < F;
            // unconditional loads from a clamped index + select: a conditional load becomes a branch, and a branch per
            // sample serialises the memory latency (that alone was 1.5 ms of 3.6 ms)
            const int nc = in ? n : 0;
            const float xv = px[nc], yv = py[nc], zv = pz[nc];
            xr[n1] = in ? xv : 0.f;
            yr[n1] = in ? yv : 0.f;
            zr[n1] = in ? zv : 0.f;
        }
        cplx sig[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const double x = (double)xr[n1], y = (double)yr[n1], z = (double)zr[n1];
            if (pair == 0) sig[n1] = {x * x, y * y};
            else if (pair == 1) sig[n1] = {z * z, x * y};
            else sig[n1] = {x * z, y * z};
        }
        cplx w[G][8];
        fft_workgroup<N1>(sig, w, lds, a.tab, tid);
        // spectrum to LDS in frequency order, then every thread reads the mirror frequency of its own ones
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int p = 0; p < 8; ++p) fb[fft_pad(256 * j + 32 * N1 * bitrev<3>(p))] = w[j][p];
        __syncthreads();
        const double wp = pair == 2 ? 2.0 : 1.0, wq = pair == 0 ? 1.0 : 2.0;
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int k = tid + 256 * j + 32 * N1 * bitrev<3>(p);
                const int km = k == 0 ? 0 : M - k;
                const cplx zm = lds[km + (km >> 3) + 8 * (km >> 8)];
                const cplx zk = w[j][p];
                // P = (Z(k) + conj Z(M-k)) / 2, Q = (Z(k) - conj Z(M-k)) / (2i)
                const double sr = zk.re + zm.re, si = zk.im - zm.im;
                const double dr = zk.re - zm.re, di = zk.im + zm.im;
                W[j][p] += 0.25 * (wp * (sr * sr + si * si) + wq * (dr * dr + di * di));
            }
        __syncthreads();
    }
    // the weighted power spectrum (real, even) back through the same transform
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int p = 0; p < 8; ++p) fb[fft_pad(256 * j + 32 * N1 * bitrev<3>(p))] = {W[j][p], 0.0};
    __syncthreads();
    {
        cplx sig[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) sig[n1] = fb[fft_pad(256 * n1)];
        __syncthreads();
        cplx w[G][8];
        fft_workgroup<N1>(sig, w, lds, a.tab, tid);
        double *out = a.psum + ((int64_t)v * a.R + r) * a.Lp;
        const double inv = 1.0 / (double)M;
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int lag = tid + 256 * j + 32 * N1 * bitrev<3>(p);
                if (lag >= 1 && lag <= a.L) out[lag] = w[j][p].re * inv;
            }
    }
}

template <int N1, bool HALF>
int launch_ct_fft_h(sr_ctx *ctx, const CtFftArgs &a, int64_t series)
{
    const size_t lds = (size_t)fft_lds_slots(256 * N1) * sizeof(cplx);
    if (lds > 64 * 1024)
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ct_fft<N1, HALF>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_ct_fft<N1, HALF>), dim3((unsigned)series), dim3(256), lds, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}
template <int N1>
int launch_ct_fft(sr_ctx *ctx, const CtFftArgs &a, int64_t series)
{
    constexpr int NZ = N1 == 24 ? 16 : N1 / 2;
    return a.F <= 256 * NZ ? launch_ct_fft_h<N1, true>(ctx, a, series) : launch_ct_fft_h<N1, false>(ctx, a, series);
}

// ------------------------------------------------------------------------------------------
// kernel 1, REAL-input FFT formulation (production for 4096 < F + L <= 8192)
// ------------------------------------------------------------------------------------------
// Same mathematics as k_ct_fft (six autocorrelations by Wiener-Khinchin, float64), restructured around occupancy: the six
// signals are real, so each goes through a complex transform of HALF the padded length (z[m] = a[2m] + i a[2m+1],
// H = M/2 points) and the real even power spectrum comes back through ONE half-length transform.  The LDS image of a
// transform shrinks from 96 KB to 52 KB (H = 3072): THREE workgroups share a CU's 160 KB instead of one, at <= 168
// registers per lane -- k_ct_fft runs at one wave per SIMD (256 VGPR + 196 AGPR) and is latency-bound.  Seven half-length
// transforms per series replace four full-length ones (20 % fewer flop).
//
// H = N1 * 256 (N1 = 12: M = 6144, the F = 4096 chunks; N1 = 16: M = 8192), 256 threads, three steps N1 x 16 x 16 with
// every small transform in registers:
//   1. thread n2 holds z[n2 + 256 n1], n1 < N1: N1-point transform (12 = 3 x 4), twiddle w_H^(n2 k1), to LDS as
//      element k1*256 + n2 (one pad slot per 16);
//   2. thread (k1, lo), k1 < N1 (16 N1 of the 256 threads): 16-point transform over h (n2 = lo + 16 h), twiddle
//      w_256^(lo k2a), to LDS row (k1*16 + k2a), column lo (rows of 17 slots);
//   3. thread (k1, k2a): reads its own row, 16-point transform over lo: X[k1 + N1 (k2a + 16 k2b)], k2b < 16.
// Real-signal spectrum from Z = FFT_H(z):  A[k] = (Z[k] + conj Z[H-k])/2 - (i/2) w_M^k (Z[k] - conj Z[H-k]); the partner
// frequency H - k lives in thread (N1-k1, 15-k2a) at 15-k2b (k1 = 0 apart), fetched through the row layout.
// Back: with P[k] the weighted power spectrum (P[M-k] = P[k]),  Y[k] = (P[k] + P[H-k]) + i (P[k] - P[H-k]) conj(w_M^k);
// FFT_H(Y)[m] = M (S[2m] + i S[2m-1]): the even lags in the real part, the odd ones in the imaginary part.
// All LDS accesses are 16-byte (one complex) and conflict-free for the lane groups of ds_read_b128 / ds_write_b128
// (MI355X_MICROARCH.md, LDS) except a 2-way case in the natural-order read of Y.
template <int N1>
struct RStage1 {                                           // N1 = 16
    __host__ __device__ static constexpr int k1(int p) { return bitrev<4>(p); }
    __device__ static __forceinline__ void run(cplx *v) { fft_reg<4>(v); }
};
template <int B>
__device__ __forceinline__ void dft3_col12(cplx *v, cplx (*y)[4])
{
    constexpr double h = 0.8660254037844386;            // sqrt(3)/2
    const cplx x0 = v[B], x1 = v[4 + B], x2 = v[8 + B];
    const cplx t = {x1.re + x2.re, x1.im + x2.im}, d = {x1.re - x2.re, x1.im - x2.im};
    const cplx m = {fma(-0.5, t.re, x0.re), fma(-0.5, t.im, x0.im)};
    const cplx r = {h * d.im, -h * d.re};                // -i sqrt(3)/2 (x1 - x2)
    y[0][B] = {x0.re + t.re, x0.im + t.im};
    y[1][B] = mul_w24<2 * B>(cplx{m.re + r.re, m.im + r.im});          // w_12^B
    y[2][B] = mul_w24<4 * B>(cplx{m.re - r.re, m.im - r.im});          // w_12^(2B)
    if constexpr (B + 1 < 4) dft3_col12<B + 1>(v, y);
}
template <>
struct RStage1<12> {                                       // n1 = 4 a + b, k1 = ka + 3 kb
    __host__ __device__ static constexpr int k1(int p) { return (p >> 2) + 3 * bitrev<2>(p & 3); }
    __device__ static __forceinline__ void run(cplx *v)
    {
        cplx y[3][4];
        dft3_col12<0>(v, y);
#pragma unroll
        for (int ka = 0; ka < 3; ++ka) {
            fft_reg<2>(y[ka]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * ka + q] = y[ka][q];
        }
    }
};

// Hide a value's provenance from the optimiser.  Twiddle bases depend only on the thread, so everything derived from them
// (11 + 15 + 16 complex powers) is invariant across the six transforms of a series: left alone, the compiler computes them
// once, parks 100+ registers and spills them (83 scratch stores in the prologue, ~150 reloads per transform; the kernel
// then waits on scratch 85 % of the time).  Recomputing them per transform is a few dozen multiplies.
__device__ __forceinline__ cplx opaque(cplx z)
{
    asm volatile("" : "+v"(z.re), "+v"(z.im));
    return z;
}
// The same for the thread index: every LDS / global address of the kernel is a function of it, and the ~80 addresses of one
// transform would otherwise be computed once and spilled.
__device__ __forceinline__ int opaque(int t)
{
    asm volatile("" : "+v"(t));
    return t;
}

// v[p] *= base^k(p): the powers are built one after the other (k = 1 .. N-1), no table of N powers in registers
template <int N, class KOF>
__device__ __forceinline__ void apply_twiddles_seq(cplx *v, cplx base)
{
    cplx cur = base;
#pragma unroll
    for (int k = 1; k < N; ++k) {
#pragma unroll
        for (int p = 0; p < N; ++p)
            if (KOF::k1(p) == k) v[p] = cmul(v[p], cur);
        if (k + 1 < N) cur = cmul(cur, base);
    }
}

// v[p] *= base^k(p) with the powers built as a tree (base^k = base^(k/2) * base^(k - k/2)): dependency depth log2 N
// instead of N (the sequential chain cost 12 % of the kernel: every wave waits on it at one or two waves per SIMD)
template <int N, class KOF>
__device__ __forceinline__ void apply_twiddles_tree(cplx *v, cplx base)
{
    cplx pw[N];
    pw[1] = base;
#pragma unroll
    for (int k = 2; k < N; ++k) pw[k] = cmul(pw[k >> 1], pw[k - (k >> 1)]);
#pragma unroll
    for (int p = 0; p < N; ++p) {
        const int k = KOF::k1(p);
        if (k != 0) v[p] = cmul(v[p], pw[k]);
    }
}

struct CtRfftArgs {
    const float *soa;
    int64_t Npad;
    const int64_t *chunk_start;   // device, may be null
    const double *tab;            // 3 x 256 complex: w_H^t, w_256^t, w_M^t
    double *psum;                 // (nV, R, Lp)
    int R, F, L, Lp;
};

__host__ __device__ constexpr int rfft_lds_slots(int N1) { return 256 * N1 + 256 * N1 / N1 + 16; }   // natural order, one pad per N1

// One half-length transform: the thread's N1 inputs v[] (natural order, element tid + 256 n1) -> for the 16 N1 threads
// (k1, k2a) = (tid >> 4, tid & 15), k1 < N1: w[p] = X[k1 + N1 (k2a + 16 rev4(p))].  The caller has made sure nobody still
// reads the LDS image; on return every thread has read what it needs from it (row tid is the thread's own).
template <int N1>
__device__ __forceinline__ void rfft_workgroup(cplx *v, cplx *w, cplx *lds, cplx base1, int tid)
{
    RStage1<N1>::run(v);
    apply_twiddles_tree<N1, RStage1<N1>>(v, base1);
    {
        cplx *b = lds + tid + (tid >> 4);                         // element k1*256 + tid, one pad slot per 16
#pragma unroll
        for (int p = 0; p < N1; ++p) b[272 * RStage1<N1>::k1(p)] = v[p];
    }
    __syncthreads();
    const int k1 = tid >> 4, lo = tid & 15;
    const bool act = k1 < N1;
    cplx u[16];
    if (act) {
        const cplx *b = lds + 272 * k1 + lo;                      // element k1*256 + lo + 16 h -> + 17 h
#pragma unroll
        for (int h = 0; h < 16; ++h) u[h] = b[17 * h];
        fft_reg<4>(u);
        {
            const cplx *tw = lds + rfft_lds_slots(N1) + lo;       // w_256^(lo k2a) at [k2a*16 + lo], filled at kernel start
#pragma unroll
            for (int p = 1; p < 16; ++p) u[p] = cmul(u[p], tw[16 * bitrev<4>(p)]);
        }
        // in place: element k1*256 + lo + 16 h sits in row (k1*16 + h), column lo -- the very cells this thread has just
        // read are the ones it writes as row (k1*16 + k2a), column lo: no barrier between its reads and its writes
        cplx *bw = lds + 272 * k1 + lo;
#pragma unroll
        for (int p = 0; p < 16; ++p) bw[17 * bitrev<4>(p)] = u[p];
    }
    // The row thread tid reads next (17 tid .. 17 tid + 15) was written by the 16 threads (k1, lo = 0..15) = 16 k1 .. 16 k1 + 15:
    // its own 16-lane group.  LDS operations of one wave complete in order, so no workgroup barrier is needed here -- only
    // the compiler must not move the reads above the writes.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (act) {
        const cplx *b = lds + 17 * tid;
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e] = b[e];
        fft_reg<4>(w);
    }
}

// HALF: the chunk fills at most 2/3 (N1 = 12) or 1/2 (N1 = 16) of the padded length: the thread's inputs beyond NZ are
// known to be zero and are neither loaded nor multiplied.
//
// TR ("traceless"): FIVE forward transforms instead of six.  With T = u (x) u and s = |u|^2,
//     (u.u')^2 = sum_ij T_ij T'_ij = sum_ij Q_ij Q'_ij + s s' / 3,        Q = T - (s/3) 1   (traceless, 5 components),
//     sum_ij Q_ij Q'_ij = d1 d1'/2 + d2 d2'/6 + 2 (xy x'y' + xz x'z' + yz y'z'),  d1 = x^2 - y^2,  d2 = 2 z^2 - x^2 - y^2
// (an orthonormal change of basis on the diagonal (x^2, y^2, z^2); exact for ANY vectors).  The bond vectors are unit
// vectors rounded to float32: s = 1 + e with |e| < 3e-7, so the trace term needs no transform,
//     sum_{j < F-d} s_j s_{j+d} = (F - d) + P[F-d] + (P[F] - P[d]) + O(F e^2),      P[k] = sum_{j<k} e_j  (prefix sums),
// and the neglected O(e^2) part is < 1e-13 of C(t).  P[F-d] + (P[F] - P[d]) = G[0] + G[d] with G[d] = sum_{j=d}^{F-d-1} e_j, the
// sum over a window that shrinks from both ends: a suffix scan over HALF the series.  The prologue (which holds x, y, z for the
// first signal anyway) forms e, the workgroup scans it once in float32 with DPP adds, and the finished term
// ((F - d) + G[0] + G[d]) / 3 stays in LDS as float64 (16 KB) until the lags are written: one look-up and one fma per lag.
// Measured (rocprofv3 PMC, cfg3): 7.1 % fewer VALU instructions per launch than the six-signal kernel (a seventh of the
// transforms minus this bookkeeping), 3.4 % fewer wave cycles, 0.95 -> 0.915 ms: the kernel's waves spend 36 % of their life
// at barriers / waitcnt and 22 % in issue stalls, which a shorter instruction stream does not shorten.  (The first version --
// float64 prefix sums over the whole series through ds_bpermute shuffles, three look-ups per lag -- cost as much as it saved.)
// A series with any |e| >= kUnitTol (not a unit vector: zero vectors from the 0/0 guard of vecnorm_NDarray, callers with
// unnormalised input) runs the sixth transform on s instead -- decided per workgroup, same kernel.
constexpr double kUnitTol = 5e-7;

template <bool TR> __device__ __forceinline__ int rfft_plane_a(int c) { return TR ? (c == 4 ? 1 : 0) : (c < 3 ? c : (c == 5 ? 1 : 0)); }
template <bool TR> __device__ __forceinline__ int rfft_plane_b(int c) { return TR ? (c == 3 || c == 4 ? 2 : 1) : (c < 3 ? c : (c == 3 ? 1 : 2)); }
// weight / 4 of signal c in the power spectrum
template <bool TR> __device__ __forceinline__ double rfft_weight4(int c)
{
    if (!TR) return c < 3 ? 0.25 : 0.5;
    return c == 0 ? 1.0 / 24.0 : (c == 1 ? 0.125 : (c == 5 ? 1.0 / 12.0 : 0.5));
}

// (A register budget below the 256 that two waves per SIMD allow -- amdgpu_num_vgpr, which counts in units of TWO registers on
// gfx90a and later -- was tried to leave the bandwidth kernels room beside a C(t) + fit pair of waves: 240 / 232 / 224 VGPRs
// cost 36-52 B of scratch in the transform loop, 0.93 -> 1.01 / 1.01 / 1.10 ms alone, no hiding gained)
template <int N1, bool HALF, bool TR>
__global__ __launch_bounds__(256, 2) void k_ct_rfft(CtRfftArgs a)
{
    extern __shared__ __align__(16) unsigned char fft_smem[];
    cplx *lds = reinterpret_cast<cplx *>(fft_smem);
    constexpr int H = N1 * 256, M = 2 * H;
    constexpr int NZ = HALF ? (N1 == 12 ? 8 : N1 / 2) : N1;
    // PF: the samples of signal c + 1 are loaded one transform ahead (2 NZ float2 registers held across the transform).  With
    // all 16 input blocks in use that is 64 VGPRs the 256-register budget does not have (188 B of scratch, reloaded inside the
    // transform): the M = 8192 kernel loads them right before it forms the signal and leaves the latency to the other
    // workgroup of the CU.
    constexpr bool PF = NZ <= 8;
    float *Pl = reinterpret_cast<float *>(lds + rfft_lds_slots(N1) + 512);     // TR: the trace term's table (2049 doubles), then scan scratch
    const int tid0 = threadIdx.x;
    const int v = blockIdx.x / a.R, r = blockIdx.x - v * a.R;
    const int F = a.F;
    const int64_t start = a.chunk_start ? a.chunk_start[r] : (int64_t)r * F;
    const float *px = a.soa + ((int64_t)v * 3 + 0) * a.Npad + start;
    const bool even = ((start | a.Npad | (int64_t)F) & 1) == 0;   // frames 2m, 2m + 1 of every plane share an aligned 8 bytes,
                                                                  // and no pair straddles the end of the chunk
    // partner thread holding the frequencies H - k (see the header comment): pt; thread 0 pairs k2b with (16 - k2b) & 15,
    // everybody else with 15 - k2b: column (15 - k2b + off0) & 15, which only wraps for thread 0 at k2b = 0
    const cplx wbase = {a.tab[2 * (512 + ((tid0 >> 4) < N1 ? (tid0 >> 4) + N1 * (tid0 & 15) : 0))],
                        a.tab[2 * (512 + ((tid0 >> 4) < N1 ? (tid0 >> 4) + N1 * (tid0 & 15) : 0)) + 1]};   // w_M^(k1 + N1 k2a)

    // Power spectrum, by PAIRS of frequencies (k, H - k): thread (k1, k2a) owns the pairs whose k has k2b < 8; it keeps
    // Wk[q] = P[k] (k2b = q) and Wm[q] = P[H - k] (the partner thread's frequency 15 - q).  With S = Z[k] + conj Z[H-k],
    // D = Z[k] - conj Z[H-k], T = w_M^k D:   4 |A[k]|^2 = |S - i T|^2   and   4 |A[H-k]|^2 = |S + i T|^2  -- one complex
    // multiply serves both.  Thread 0 is its own partner with k2b <-> 16 - k2b: its slot q = 0 holds k = 0 (Wk) and k = H
    // (Wm), and the self-paired frequency k = H/2 (k2b = 8) gets the scalar Wmid.
    double Wk[8], Wm[8], Wmid = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) Wk[q] = Wm[q] = 0.0;

    // signal c is a product of two of the three planes (TR: c = 1 is x^2 - y^2; c = 0 and 5 need all three, see below).
    // Thread t holds the pairs of frames (2m, 2m + 1), m = t + 256 n1; unconditional range-checked loads (no branch per
    // sample), one 8-byte load per plane when the pair is aligned.  The loads of signal c + 2 are issued right after the
    // samples of signal c + 1 have been turned into its input, i.e. a whole transform before they are needed (15 % of the
    // kernel was spent waiting for them at the top of every transform).
    // Loads go through buffer resources that cover exactly the chunk's F frames of a plane: a frame past the chunk reads
    // as 0 by the hardware range check -- no clamp, no select, and the address is one 32-bit byte offset per load instead
    // of a 64-bit add (13 % of the transform loop's instructions were address arithmetic and masks).
    float2 ar[NZ], br[NZ];
#define SR_RFFT_LOAD1(DST, PLANE, T)                                                             \
    {                                                                                            \
        /* the plane index is workgroup-uniform: say so, or the descriptor is built in VGPRs and every load becomes a */ \
        /* waterfall loop (readfirstlane + compare + masked load), 13 instructions and a serialisation each          */ \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                    \
            const_cast<float *>(px + (int64_t)__builtin_amdgcn_readfirstlane(PLANE) * a.Npad), (short)0, F * 4, 0x00020000); \
        if (even) {                                                                              \
            _Pragma("unroll") for (int n1 = 0; n1 < NZ; ++n1)                                    \
                DST[n1] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_, 8 * ((T) + 256 * n1), 0, 0)); \
        } else {                                                                                 \
            _Pragma("unroll") for (int n1 = 0; n1 < NZ; ++n1) {                                  \
                const int ob_ = 8 * ((T) + 256 * n1);                                            \
                DST[n1] = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, ob_, 0, 0)),      \
                                      __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, ob_ + 4, 0, 0))); \
            }                                                                                    \
        }                                                                                        \
    }
#define SR_RFFT_LOAD(C, T)                                                                       \
    {                                                                                            \
        const int cc_ = (C);                                                                     \
        SR_RFFT_LOAD1(ar, rfft_plane_a<TR>(cc_), T)                                              \
        SR_RFFT_LOAD1(br, rfft_plane_b<TR>(cc_), T)                                              \
    }
    {
        // step-2 twiddles w_256^(lo k2a), transposed so that the 16 lanes of a ds_read_b128 group (consecutive lo) hit 16
        // consecutive slots; ordered before their first use by the first barrier of the first transform
        const int j = ((tid0 & 15) * (tid0 >> 4)) & 255;
        lds[rfft_lds_slots(N1) + tid0] = cplx{a.tab[2 * (256 + j)], a.tab[2 * (256 + j) + 1]};
    }

    // step-1 twiddle base w_H^tid: the same for every transform of the series.  From a table in LDS, not from global
    // memory: vmcnt counts in order, so waiting for a global load issued behind the sample prefetch drains the prefetch too
    // (measured: 0.94 -> 1.00 ms although a seventh of the transforms was gone), and carrying it in registers across the
    // transforms costs four of the VGPRs the loop does not have.  Ordered before its first read by the first barrier below.
    lds[rfft_lds_slots(N1) + 256 + tid0] = cplx{a.tab[2 * tid0], a.tab[2 * tid0 + 1]};
    cplx sig[N1];                 // input of the next transform (entries >= NZ stay zero)
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) sig[n1] = cplx{0.0, 0.0};
    int nsig = 6;
    if (TR) {
        // ---- prologue of the traceless form: signal 0 = 2 z^2 - x^2 - y^2, and the trace term's table ----
        // e_j = |u_j|^2 - 1 (float64, then rounded to float32: |e| < 3e-7, so 1e-14 absolute).  What the lags need is
        //     P[F-d] + (P[F] - P[d]) = G[0] + G[d],      G[d] = sum_{j = d}^{F-d-1} e_j  (the window that shrinks from both ends),
        // and G is a suffix sum of h_i = e_i + e_{F-1-i} (i < F-1-i; the centre frame once): a scan over HALF the series,
        // in float32 (sums of < 4096 terms of 1e-7: rounding 1e-12 absolute against F - d > 2000), with DPP adds.
        float emax = 0.f;
        float *E = reinterpret_cast<float *>(lds);          // scratch in the still unused transform image
        float *aux = Pl + 2 * 2056;                         // behind the table: [0 .. 4) wave totals, [4 .. 8) wave maxima of |e|
        {
            float2 zr[NZ];
            SR_RFFT_LOAD(0, tid0)                      // x, y
            SR_RFFT_LOAD1(zr, 2, tid0)
            const bool full = F == 512 * NZ;           // no frame of the loaded blocks lies behind the chunk
#pragma unroll
            for (int n1 = 0; n1 < NZ; ++n1) {
                const double x0 = (double)ar[n1].x, x1 = (double)ar[n1].y, y0 = (double)br[n1].x, y1 = (double)br[n1].y;
                const double z0 = (double)zr[n1].x, z1 = (double)zr[n1].y;
                const double q0 = fma(x0, x0, y0 * y0), q1 = fma(x1, x1, y1 * y1), zz0 = z0 * z0, zz1 = z1 * z1;
                sig[n1] = cplx{(zz0 + zz0) - q0, (zz1 + zz1) - q1};
                const int f0 = 2 * (tid0 + 256 * n1);
                float ea = (float)((q0 + zz0) - 1.0), eb = (float)((q1 + zz1) - 1.0);
                if (!full) {
                    ea = f0 < F ? ea : 0.f;
                    eb = f0 + 1 < F ? eb : 0.f;
                }
                *reinterpret_cast<float2 *>(E + f0) = make_float2(ea, eb);
                emax = fmaxf(emax, fmaxf(fabsf(ea), fabsf(eb)));
            }
        }
        SR_RFFT_LOAD(1, tid0)
        const int lane = tid0 & 63, wave = tid0 >> 6;
        {
            // wave maximum with DPP moves (an inclusive max-scan: lane 63 ends up with the maximum; |e| >= 0, so 0 is neutral)
#define SR_DPP_MAX(CTRL, RM) emax = fmaxf(emax, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, emax), CTRL, RM, 0xF, false)));
            SR_DPP_MAX(0x111, 0xF) SR_DPP_MAX(0x112, 0xF) SR_DPP_MAX(0x114, 0xF) SR_DPP_MAX(0x118, 0xF) SR_DPP_MAX(0x142, 0xA) SR_DPP_MAX(0x143, 0xC)
#undef SR_DPP_MAX
            if (lane == 63) aux[4 + wave] = emax;
        }
        __syncthreads();
        // thread t owns i = 8 b .. 8 b + 7 with b = 255 - t: an inclusive PREFIX scan over t is the suffix sum over i
        const int i0 = 8 * (255 - tid0);
        float sfx[8], incl;
        {
            const float4 ea = *reinterpret_cast<const float4 *>(E + i0), eb = *reinterpret_cast<const float4 *>(E + i0 + 4);
            const float ei[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
            float h[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k, j = F - 1 - i;               // j > 0: F > 2730 for this transform length
                const float ej = E[j];
                h[k] = i < j ? ei[k] + ej : (i == j ? ei[k] : 0.f);
            }
            sfx[7] = h[7];
#pragma unroll
            for (int k = 6; k >= 0; --k) sfx[k] = h[k] + sfx[k + 1];
            // wave-wide inclusive scan of the thread totals: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then row_bcast15 /
            // row_bcast31 (the AMDGPU atomic optimiser's sequence); lanes without a source add 0
            float vsc = sfx[0];
            vsc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vsc), 0x111, 0xF, 0xF, false));
            vsc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vsc), 0x112, 0xF, 0xF, false));
            vsc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vsc), 0x114, 0xF, 0xF, false));
            vsc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vsc), 0x118, 0xF, 0xF, false));
            vsc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vsc), 0x142, 0xA, 0xF, false));
            vsc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, vsc), 0x143, 0xC, 0xF, false));
            incl = vsc;
            if (lane == 63) aux[wave] = incl;
        }
        __syncthreads();                                     // every read of E is done: the first transform may use the image
        {
            float off = incl - sfx[0];
#pragma unroll
            for (int w2 = 0; w2 < 3; ++w2) off += w2 < wave ? aux[w2] : 0.f;
            const float G0 = (aux[0] + aux[1]) + (aux[2] + aux[3]);          // sum of every e of the series
            // Tt[d] = ((F - d) + G[0] + G[d]) / 3: the finished trace term of lag d, float64 -- one fma per lag at the end
            double *Tt = reinterpret_cast<double *>(Pl);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                Tt[i0 + k] = ((double)(F - (i0 + k)) + (double)(G0 + (sfx[k] + off))) * (1.0 / 3.0);
            if (tid0 == 0) Tt[2048] = ((double)(F - 2048) + (double)G0) * (1.0 / 3.0);
            const float mx = fmaxf(fmaxf(aux[4], aux[5]), fmaxf(aux[6], aux[7]));
            nsig = __builtin_amdgcn_readfirstlane(mx < (float)kUnitTol ? 5 : 6);
        }
    } else {
        SR_RFFT_LOAD(0, tid0)
#pragma unroll
        for (int n1 = 0; n1 < NZ; ++n1)
            sig[n1] = cplx{(double)ar[n1].x * (double)br[n1].x, (double)ar[n1].y * (double)br[n1].y};
        if (PF) SR_RFFT_LOAD(1, tid0)
    }
#pragma unroll 1
    for (int c = 0; c < nsig; ++c) {
        asm volatile("" ::: "memory");
        const int tid = opaque(tid0);
        const int k1 = tid >> 4, k2a = tid & 15;
        const bool act = k1 < N1;
        const int pt = k1 != 0 ? (N1 - k1) * 16 + (15 - k2a) : (k2a != 0 ? 16 - k2a : 0);
        const int off0 = tid == 0 ? 1 : 0;
        const cplx base1 = opaque(lds[rfft_lds_slots(N1) + 256 + tid]);
        cplx w[16];
        rfft_workgroup<N1>(sig, w, lds, base1, tid);
        // own row again, now in frequency order k2b; then every thread reads the partner frequencies of its 8 pairs
        if (act) {
            cplx *b = lds + 17 * tid;
#pragma unroll
            for (int p = 0; p < 16; ++p) b[bitrev<4>(p)] = w[p];
        }
        __syncthreads();
        if (act) {
            const double wgt = rfft_weight4<TR>(c);               // weight / 4
            const cplx *b = lds + 17 * pt + off0;
            const cplx wb = opaque(wbase);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const cplx zk = w[bitrev<4>(q)];
                if (q == 0 && off0) {                              // thread 0: k = 0 and k = H from Z[0] alone
                    const double e0 = zk.re + zk.im, eh = zk.re - zk.im;
                    Wk[0] = fma(4.0 * wgt, e0 * e0, Wk[0]);
                    Wm[0] = fma(4.0 * wgt, eh * eh, Wm[0]);
                    continue;
                }
                const cplx zm = b[15 - q];
                const cplx S = {zk.re + zm.re, zk.im - zm.im}, D = {zk.re - zm.re, zk.im + zm.im};
                const cplx T = cmul(mul_w32_rt(wb, q), D);
                const double pr = S.re + T.im, pi = S.im - T.re;      // S - i T
                const double mr = S.re - T.im, mi = S.im + T.re;      // S + i T
                Wk[q] = fma(wgt, fma(pr, pr, pi * pi), Wk[q]);
                Wm[q] = fma(wgt, fma(mr, mr, mi * mi), Wm[q]);
            }
            if (off0) {                                            // k = H/2 (k2b = 8) mirrors onto itself
                const cplx zk = w[bitrev<4>(8)];
                const cplx S = {2.0 * zk.re, 0.0}, D = {0.0, 2.0 * zk.im};
                const cplx T = cmul(mul_w32_rt(wb, 8), D);
                const double pr = S.re + T.im, pi = S.im - T.re;
                Wmid = fma(wgt, fma(pr, pr, pi * pi), Wmid);
            }
        }
        // the next signal's input from the samples loaded one transform ago (w is dead here: few live registers), and the loads
        // of the one after it.  Unconditional (behind the last signal the values are simply not used): a conditional
        // assignment would keep the transform's in-place leftovers in `sig` alive through the spectrum step.
        {
            const int cn = c + 1;
            if (!PF && cn < nsig) SR_RFFT_LOAD(cn, tid)
            // keep these products HERE: nothing ties them to this point but their inputs, and scheduled above the spectrum
            // step (where w[16] is live) they push the accumulators into scratch
#pragma unroll
            for (int n1 = 0; n1 < NZ; ++n1)
                asm volatile("" : "+v"(ar[n1].x), "+v"(ar[n1].y), "+v"(br[n1].x), "+v"(br[n1].y));
#pragma unroll
            for (int n1 = NZ; n1 < N1; ++n1) sig[n1] = cplx{0.0, 0.0};
            if (TR && cn == 1) {                                   // x^2 - y^2
#pragma unroll
                for (int n1 = 0; n1 < NZ; ++n1) {
                    const double a0 = (double)ar[n1].x, a1 = (double)ar[n1].y, b0 = (double)br[n1].x, b1 = (double)br[n1].y;
                    sig[n1] = cplx{fma(a0, a0, -(b0 * b0)), fma(a1, a1, -(b1 * b1))};
                }
            } else {
#pragma unroll
                for (int n1 = 0; n1 < NZ; ++n1)
                    sig[n1] = cplx{(double)ar[n1].x * (double)br[n1].x, (double)ar[n1].y * (double)br[n1].y};
            }
            if (TR && cn == 5 && nsig == 6) {          // not a unit vector: s = x^2 + y^2 + z^2 itself (rare; the z load is exposed)
#pragma unroll
                for (int n1 = 0; n1 < NZ; ++n1) {
                    const double a0 = (double)ar[n1].x, a1 = (double)ar[n1].y, b0 = (double)br[n1].x, b1 = (double)br[n1].y;
                    sig[n1] = cplx{fma(a0, a0, b0 * b0), fma(a1, a1, b1 * b1)};
                }
                SR_RFFT_LOAD1(ar, 2, tid)
#pragma unroll
                for (int n1 = 0; n1 < NZ; ++n1) {
                    const double z0 = (double)ar[n1].x, z1 = (double)ar[n1].y;
                    sig[n1] = cplx{fma(z0, z0, sig[n1].re), fma(z1, z1, sig[n1].im)};
                }
            } else if (PF && c + 2 < nsig) {
                SR_RFFT_LOAD(c + 2, tid)
            }
        }
        __syncthreads();
    }

    // ---- back: Y[k] = (P[k] + P[H-k]) + i (P[k] - P[H-k]) conj(w_M^k), through the same transform.  The pair owner has
    // both P[k] and P[H-k]:  Y[k] = (E - d sin, d cos),  Y[H-k] = (E + d sin, d cos)  with E = P[k] + P[H-k],
    // d = P[k] - P[H-k], w_M^k = (cos, -sin).  Natural order with one pad slot per N1 elements: k + k2a + 16 k2b. ----
    const int tid = opaque(tid0);
    const int k1 = tid >> 4, k2a = tid & 15;
    const bool act = k1 < N1;
    const int pt = k1 != 0 ? (N1 - k1) * 16 + (15 - k2a) : (k2a != 0 ? 16 - k2a : 0);
    const int off0 = tid == 0 ? 1 : 0;
    if (act) {
        cplx *bk = lds + k1 + (N1 + 1) * k2a;                                    // own frequencies, column k2b = q
        cplx *bm = lds + (pt >> 4) + (N1 + 1) * (pt & 15) + 16 * (N1 + 1) * off0;  // the partner's, column 15 - q (+ 1 for thread 0)
        const cplx wb = opaque(wbase);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double E = Wk[q] + Wm[q], d = Wk[q] - Wm[q];
            const cplx wk = mul_w32_rt(wb, q);                                  // (cos, -sin)
            bk[16 * (N1 + 1) * q] = {fma(d, wk.im, E), d * wk.re};
            if (!(q == 0 && off0)) bm[16 * (N1 + 1) * (15 - q)] = {fma(-d, wk.im, E), d * wk.re};
        }
        if (off0) bk[16 * (N1 + 1) * 8] = {2.0 * Wmid, 0.0};                     // k = H/2: E = 2 P, d = 0
    }
    __syncthreads();
    {
        cplx yin[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const int k = tid + 256 * n1;
            yin[n1] = lds[k + k / N1];
        }
        __syncthreads();
        cplx w[16];
        rfft_workgroup<N1>(yin, w, lds, opaque(lds[rfft_lds_slots(N1) + 256 + tid]), tid);
        if (act) {
            double *out = a.psum + ((int64_t)v * a.R + r) * a.Lp;
            const double inv = 1.0 / (double)M;
            const bool unit = TR && nsig == 5;
            const double *Tt = reinterpret_cast<const double *>(Pl);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int m = k1 + N1 * (k2a + 16 * bitrev<4>(p));
                const int le = 2 * m, lod = 2 * m - 1;
                if (le >= 1 && le <= a.L) out[le] = unit ? fma(w[p].re, inv, Tt[le]) : w[p].re * inv;
                if (lod >= 1 && lod <= a.L) out[lod] = unit ? fma(w[p].im, inv, Tt[lod]) : w[p].im * inv;
            }
        }
    }
#undef SR_RFFT_LOAD
#undef SR_RFFT_LOAD1
}

template <int N1, bool HALF, bool TR>
constexpr size_t rfft_lds_bytes()
{
    // transform image + the 16 x 16 step-2 twiddles + the 256 step-1 twiddle bases (+ TR: the trace term's table Tt[0 .. 2048]
    // as float64, then 4 wave totals and 4 wave maxima)
    return (size_t)(rfft_lds_slots(N1) + 512) * sizeof(cplx) + (TR ? (size_t)(2 * 2056 + 16) * sizeof(float) : 0);
}

template <int N1, bool HALF, bool TR>
int launch_ct_rfft_h(sr_ctx *ctx, const CtRfftArgs &a, int64_t series)
{
    const size_t lds = rfft_lds_bytes<N1, HALF, TR>();
    if (lds > 64 * 1024)
        SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ct_rfft<N1, HALF, TR>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_ct_rfft<N1, HALF, TR>), dim3((unsigned)series), dim3(256), lds, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}
// With L = F/2 the two transform lengths are tied to the chunk length: M = 6144 serves 4096 < 1.5 F <= 6144, i.e. F <= 4096
// (at most 8 of the 12 input blocks are non-zero: HALF), M = 8192 serves 4096 < F <= 5461 (more than half: not HALF).
// Only those instantiations exist.  The traceless form of the M = 6144 kernel is an OPTION (sr_set_option "ct_traceless"):
// alone it is 4 % faster (0.95 -> 0.915 ms for cfg3), inside the pipeline -- where a C(t) workgroup shares its CU with a
// fit workgroup -- 3 % slower per step (same-box A/B), so the six-signal kernel stays the default.
int launch_ct_rfft(sr_ctx *ctx, const CtRfftArgs &a, int64_t series)
{
    if (a.F + a.L <= 6144) {
        SR_REQUIRE(a.F <= 4096, -3, "k_ct_rfft<12>: F=%d does not fit 8 input blocks", a.F);
        return ctx->ct_traceless ? launch_ct_rfft_h<12, true, true>(ctx, a, series) : launch_ct_rfft_h<12, true, false>(ctx, a, series);
    }
    SR_REQUIRE(a.F + a.L <= 8192, -3, "k_ct_rfft<16>: F=%d too long", a.F);
    return launch_ct_rfft_h<16, false, false>(ctx, a, series);
}

// mean / std over the R replicate chunks, calculate-Ct-from-traj.py:226-228.  One workgroup owns a tile of kFinV vectors x
// kFinD lags: the raw sums are read along the lags (a wave = 64 consecutive lags of one vector), the results leave in BOTH
// orientations -- (lags, vectors) as the reference holds them, through an LDS tile so that 16 consecutive vectors of a lag
// go out together, and (vectors, lags) for the fit, straight from the registers.  (The thread-per-element version wrote
// the (lags, vectors) arrays with a stride of nV doubles: 445 MB of HBM traffic for 218 MB of data, and two transposition
// launches behind it.)
constexpr int kFinV = 16, kFinD = 64;
constexpr int kFinR = 32;          // replicate chunks a thread keeps in registers (more: two passes over memory)
__global__ __launch_bounds__(256) void k_ct_finalize(const double *__restrict__ psum, int R, int F, int L, int Lp,
                                                     int64_t nV, double *__restrict__ Ct, double *__restrict__ dCt,
                                                     double *__restrict__ CtT, double *__restrict__ dCtT)
{
    __shared__ double tm[kFinV][kFinD + 1], ts[kFinV][kFinD + 1];
    const int tid = threadIdx.x;
    const int d0 = blockIdx.x * kFinD;                  // lag index - 1 of the tile's first column
    const int64_t v0 = (int64_t)blockIdx.y * kFinV;
    const double rootR = sqrt((double)R) - 1.0;
    {
        const int dl = tid & 63;
        const int d = d0 + dl + 1;
#pragma unroll
        for (int i = 0; i < kFinV / 4; ++i) {
            const int vl = (tid >> 6) + 4 * i;
            const int64_t v = v0 + vl;
            if (d > L || v >= nV) continue;
            const double *p = psum + v * R * Lp + d;
            const double n = (double)(F - d);
            double m = 0.0, s = 0.0;
            if (R <= kFinR) {
                // the replicate values stay in registers between the two passes of numpy.std: the raw sums (201 MB for cfg3)
                // are read once, all loads in flight together
                double pr[kFinR];
#pragma unroll
                for (int r = 0; r < kFinR; ++r) pr[r] = r < R ? p[(int64_t)r * Lp] : 0.0;
#pragma unroll
                for (int r = 0; r < kFinR; ++r) {
                    pr[r] = 1.5 * (pr[r] / n) - 0.5;
                    if (r < R) m += pr[r];
                }
                m /= (double)R;
#pragma unroll
                for (int r = 0; r < kFinR; ++r) {
                    const double e = pr[r] - m;
                    if (r < R) s += e * e;
                }
            } else {
                for (int r = 0; r < R; ++r) m += 1.5 * (p[(int64_t)r * Lp] / n) - 0.5;
                m /= (double)R;
                for (int r = 0; r < R; ++r) {
                    const double e = (1.5 * (p[(int64_t)r * Lp] / n) - 0.5) - m;
                    s += e * e;
                }
            }
            const double sd = sqrt(s / (double)R) / rootR;
            tm[vl][dl] = m;
            ts[vl][dl] = sd;
            if (CtT) {
                CtT[v * L + (d - 1)] = m;
                dCtT[v * L + (d - 1)] = sd;
            }
        }
    }
    __syncthreads();
    {
        const int vl = tid & 15;
        const int64_t v = v0 + vl;
#pragma unroll
        for (int i = 0; i < kFinD / 16; ++i) {
            const int dl = (tid >> 4) + 16 * i;
            const int d = d0 + dl + 1;
            if (d > L || v >= nV) continue;
            const int64_t o = (int64_t)(d - 1) * nV + v;
            Ct[o] = tm[vl][dl];
            dCt[o] = ts[vl][dl];
        }
    }
}

__global__ __launch_bounds__(256) void k_transpose_f64(const double *__restrict__ in, int64_t rows, int64_t cols,
                                                       double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = in[(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) out[(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

template <int W>
int launch_ct(sr_ctx *ctx, const CtArgs &a, int64_t nblocks, size_t lds_bytes)
{
    if (int rc = sr_grant_lds(ctx, W == 1 ? SR_K_CT1 : SR_K_CT4, reinterpret_cast<const void *>(&k_ct_palmer<W>), lds_bytes))
        return rc;
    hipLaunchKernelGGL(k_ct_palmer<W>, dim3((unsigned)nblocks), dim3(W * 64), lds_bytes, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int sr_transpose_f64_dev(sr_ctx *ctx, const double *in, int64_t rows, int64_t cols, double *out)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(in && out && rows > 0 && cols > 0, -2, "sr_transpose_f64_dev: bad arguments");
    const int64_t gx = (cols + 31) / 32, gy = (rows + 31) / 32;
    SR_REQUIRE(gy <= 65535, -3, "sr_transpose_f64_dev: too many rows");
    hipLaunchKernelGGL(k_transpose_f64, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, ctx->stream, in, rows, cols, out);
    SR_HIP(hipGetLastError());
    return 0;
}

int64_t sr_ct_psum_stride(int64_t F) { return sr_round_up(F / 2 + 1 + kLagBlock, 8); }

int64_t sr_ct_max_frames_per_chunk(sr_ctx *ctx)
{
    if (!ctx) return -1;
    const int64_t lds = (int64_t)sr_lds_limit(ctx);
    int64_t F = lds / 12 - kPad - 64;
    return F > 0 ? F : 0;
}

int sr_pack_soa_f32_dev(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                        float *soa, int64_t Npad)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && soa, -2, "sr_pack_soa_f32_dev: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3,
               "sr_pack_soa_f32_dev: bad shape N=%lld Vtot=%lld v0=%lld nV=%lld", (long long)N, (long long)Vtot,
               (long long)v0, (long long)nV);
    SR_REQUIRE(Npad >= N && Npad % 4 == 0, -3, "sr_pack_soa_f32_dev: Npad=%lld must be >= N and a multiple of 4",
               (long long)Npad);
    const int64_t gx = (Npad + kPackFrames - 1) / kPackFrames;
    const int64_t gy = (nV + kPackVecs - 1) / kPackVecs;
    SR_REQUIRE(gy <= 65535, -3, "sr_pack_soa_f32_dev: too many vectors in one call (%lld)", (long long)nV);
    if (nV % kPackVecs == 0 && ((Vtot * 3) & 3) == 0 && ((v0 * 3) & 3) == 0 && (((uintptr_t)vecs | (uintptr_t)soa) & 15) == 0)
        hipLaunchKernelGGL(k_pack_soa, dim3((unsigned)((Npad + kPackRegFrames - 1) / kPackRegFrames), (unsigned)gy), dim3(256), 0,
                           ctx->stream, vecs, N, Vtot, v0, nV, soa, Npad);
    else
        hipLaunchKernelGGL(k_pack_soa_ragged, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, ctx->stream, vecs, N, Vtot, v0, nV,
                           soa, Npad);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_pack_soa_rot_f32_dev(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV,
                            const double *quat, float *soa, int64_t Npad)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && soa && quat, -2, "sr_pack_soa_rot_f32_dev: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3,
               "sr_pack_soa_rot_f32_dev: bad shape N=%lld Vtot=%lld v0=%lld nV=%lld", (long long)N, (long long)Vtot,
               (long long)v0, (long long)nV);
    SR_REQUIRE(Npad >= N && Npad % 4 == 0, -3, "sr_pack_soa_rot_f32_dev: Npad=%lld must be >= N and a multiple of 4",
               (long long)Npad);
    const int64_t gx = (Npad + kPackFrames - 1) / kPackFrames;
    const int64_t gy = (nV + kPackVecs - 1) / kPackVecs;
    SR_REQUIRE(gy <= 65535, -3, "sr_pack_soa_rot_f32_dev: too many vectors in one call (%lld)", (long long)nV);
    hipLaunchKernelGGL(k_pack_soa_rot, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, ctx->stream, vecs, N, Vtot, v0, nV,
                       quat, soa, Npad);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_ct_palmer_sums_f32_dev(sr_ctx *ctx, const float *soa, int64_t Npad, int64_t R, int64_t F, int64_t nV,
                              const int64_t *chunk_start_host, int mode, double *psum)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(soa && psum, -2, "sr_ct_palmer_sums_f32_dev: null pointer");
    SR_REQUIRE(R >= 1 && F >= 2 && nV >= 1, -3, "sr_ct_palmer_sums_f32_dev: bad shape R=%lld F=%lld nV=%lld", (long long)R,
               (long long)F, (long long)nV);
    SR_REQUIRE(mode == 0 || mode == 1, -3, "sr_ct_palmer_sums_f32_dev: mode must be 0 or 1");
    const int64_t Fp = ct_Fp(F);
    const size_t lds_bytes = (size_t)Fp * 3 * sizeof(float);
    SR_REQUIRE(lds_bytes <= sr_lds_limit(ctx), -4,
               "sr_ct_palmer_sums_f32_dev: F=%lld frames per chunk need %zu B of LDS (> %zu); max F is %lld",
               (long long)F, lds_bytes, sr_lds_limit(ctx),
               (long long)sr_ct_max_frames_per_chunk(ctx));
    SR_REQUIRE(R * nV < (int64_t)1 << 30, -3, "sr_ct_palmer_sums_f32_dev: too many series");
    const int64_t L = F / 2;
    const int64_t Lp = sr_ct_psum_stride(F);
    if (chunk_start_host) {
        for (int64_t r = 0; r < R; ++r)
            SR_REQUIRE(chunk_start_host[r] >= 0 && chunk_start_host[r] + F <= Npad, -3,
                       "sr_ct_palmer_sums_f32_dev: chunk %lld start %lld out of range", (long long)r,
                       (long long)chunk_start_host[r]);
    } else {
        SR_REQUIRE(R * F <= Npad, -3, "sr_ct_palmer_sums_f32_dev: R*F=%lld exceeds Npad=%lld", (long long)(R * F),
                   (long long)Npad);
    }
    int64_t *cs_dev = nullptr;
    if (chunk_start_host) {
        cs_dev = (int64_t *)sr_workspace(ctx, SR_WS_MISC, (size_t)R * sizeof(int64_t));
        if (!cs_dev) return -5;
        SR_HIP(hipMemcpyAsync(cs_dev, chunk_start_host, (size_t)R * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        SR_HIP(hipStreamSynchronize(ctx->stream));      // tiny table: the caller's array is free again when this returns
    }
    CtArgs a;
    a.soa = soa; a.Npad = Npad; a.chunk_start = cs_dev; a.psum = psum;
    a.R = (int)R; a.F = (int)F; a.Fp = (int)Fp; a.L = (int)L; a.Lp = (int)Lp; a.mode = mode;
    const int nb = mode == 0 ? (int)((L + 1) / kLagBlock) : 0;
    const int64_t series = R * nV;
    int rc;
    // FFT formulation: chunk + lags must fit a 2048 / 4096 / 8192-point transform (shorter chunks are cheap anyway)
    const int64_t need = F + L;
    if (mode == 0 && ctx->ct_fft && need > 1024 && need <= 8192) {
        double *tab = (double *)sr_workspace(ctx, SR_WS_FFT, kFftTabDoubles * sizeof(double));
        if (!tab) return -5;
        if (!ctx->fft_table_ready) {
            hipLaunchKernelGGL(k_fft_init_table, dim3((kFftTabDoubles / 2 + 255) / 256), dim3(256), 0, ctx->stream, tab);
            SR_HIP(hipGetLastError());
            SR_HIP(hipStreamSynchronize(ctx->stream));      // once per context: later launches may come on other streams
            ctx->fft_table_ready = 1;
        }
        if (ctx->ct_fft == 4 || (ctx->ct_fft == 3 && need > 4096))       // float32 transforms (sr_ct32.hip)
            return sr_launch_ct_rfft32(ctx, soa, Npad, chunk_start_host, cs_dev, psum, (int)R, (int)F, (int)L, (int)Lp, series);
        if (ctx->ct_fft >= 2 && need > 4096) {
            // real-input formulation: half-length transforms, two workgroups per CU
            CtRfftArgs ra;
            ra.soa = soa; ra.Npad = Npad; ra.chunk_start = cs_dev; ra.psum = psum;
            ra.R = (int)R; ra.F = (int)F; ra.L = (int)L; ra.Lp = (int)Lp;
            ra.tab = need <= 6144 ? tab + 2 * 1280 : tab + 2 * (1280 + 768);
            return launch_ct_rfft(ctx, ra, series);
        }
        CtFftArgs fa;
        fa.soa = soa; fa.Npad = Npad; fa.chunk_start = cs_dev; fa.tab = tab; fa.psum = psum;
        fa.R = (int)R; fa.F = (int)F; fa.L = (int)L; fa.Lp = (int)Lp;
        rc = need <= 2048 ? launch_ct_fft<8>(ctx, fa, series)
             : need <= 4096 ? launch_ct_fft<16>(ctx, fa, series)
             : need <= 6144 ? launch_ct_fft<24>(ctx, fa, series) : launch_ct_fft<32>(ctx, fa, series);
    } else if (mode == 1) {
        a.nslab = 1;
        rc = launch_ct<4>(ctx, a, series, lds_bytes);
    } else if (nb >= 16) {
        a.nslab = nb >= 64 ? nb / 32 : 1;            // about 4-8 lag blocks per wave
        rc = launch_ct<4>(ctx, a, series * a.nslab, lds_bytes);
    } else {
        a.nslab = nb > 0 ? nb : 1;                   // one wave per workgroup, one lag block per wave
        rc = launch_ct<1>(ctx, a, series * a.nslab, lds_bytes);
    }
    return rc;
}

int sr_ct_finalize_t_f64_dev(sr_ctx *ctx, const double *psum, int64_t R, int64_t F, int64_t nV, double *Ct, double *dCt,
                             double *CtT, double *dCtT)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(psum && Ct && dCt, -2, "sr_ct_finalize_f64_dev: null pointer");
    SR_REQUIRE((CtT == nullptr) == (dCtT == nullptr), -2, "sr_ct_finalize_t_f64_dev: CtT and dCtT go together");
    SR_REQUIRE(R >= 1 && F >= 2 && nV >= 1, -3, "sr_ct_finalize_f64_dev: bad shape");
    const int64_t L = F / 2;
    const int64_t Lp = sr_ct_psum_stride(F);
    const int64_t gx = (L + kFinD - 1) / kFinD, gy = (nV + kFinV - 1) / kFinV;
    SR_REQUIRE(gy <= 65535, -3, "sr_ct_finalize_f64_dev: too many vectors in one call (%lld)", (long long)nV);
    hipLaunchKernelGGL(k_ct_finalize, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, ctx->stream, psum, (int)R, (int)F, (int)L,
                       (int)Lp, nV, Ct, dCt, CtT, dCtT);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_ct_finalize_f64_dev(sr_ctx *ctx, const double *psum, int64_t R, int64_t F, int64_t nV, double *Ct, double *dCt)
{
    return sr_ct_finalize_t_f64_dev(ctx, psum, R, F, nV, Ct, dCt, nullptr, nullptr);
}

int sr_ct_palmer_f32_dev(sr_ctx *ctx, const float *soa, int64_t Npad, int64_t R, int64_t F, int64_t nV,
                         const int64_t *chunk_start_host, int mode, double *psum_ws, double *Ct, double *dCt)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(soa && Ct && dCt, -2, "sr_ct_palmer_f32_dev: null pointer");
    SR_REQUIRE(R >= 1 && F >= 2 && nV >= 1, -3, "sr_ct_palmer_f32_dev: bad shape R=%lld F=%lld nV=%lld", (long long)R,
               (long long)F, (long long)nV);
    double *psum = psum_ws;
    if (!psum) {
        psum = (double *)sr_workspace(ctx, SR_WS_PSUM, (size_t)(nV * R * sr_ct_psum_stride(F)) * sizeof(double));
        if (!psum) return -5;
    }
    int rc = sr_ct_palmer_sums_f32_dev(ctx, soa, Npad, R, F, nV, chunk_start_host, mode, psum);
    if (rc) return rc;
    return sr_ct_finalize_f64_dev(ctx, psum, R, F, nV, Ct, dCt);
}

int sr_ct_palmer_f32(sr_ctx *ctx, const float *vecs, int64_t N, int64_t Vtot, int64_t v0, int64_t nV, int64_t R,
                     int64_t F, const int64_t *chunk_start_host, int mode, double *Ct, double *dCt)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(vecs && Ct && dCt, -2, "sr_ct_palmer_f32: null pointer");
    SR_REQUIRE(N > 0 && Vtot > 0 && nV > 0 && v0 >= 0 && v0 + nV <= Vtot, -3, "sr_ct_palmer_f32: bad shape");
    // the rank's columns [v0, v0 + nV) only: 12 N nV bytes over PCIe (sr_vectors.hip), then kernel 0 and kernel 1
    sr_vectors *h = sr_vectors_create(ctx, nV, N);
    if (!h) return -5;
    int rc = sr_vectors_append_f32(ctx, h, vecs, N, Vtot, v0);
    if (!rc) rc = sr_vectors_ct_f32(ctx, h, R, F, chunk_start_host, mode, Ct, dCt);
    sr_vectors_destroy(ctx, h);
    return rc;
}

}  // extern "C"
