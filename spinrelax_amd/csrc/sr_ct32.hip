// sr_ct32.hip -- kernel 1 in the reference's own arithmetic type: the Wiener-Khinchin form of the Palmer-chunked P2
// autocorrelation with FLOAT32 transforms (k_ct_rfft32): production for 4096 < F + L <= 8192, option for 1024 < F + L <= 4096.
//
// Reference semantics: calculate_Ct_Palmer, calculate-Ct-from-traj.py:200-238 -- which computes in float32
// (`Ct = np.zeros(..., dtype=vecs.dtype)`, :219; the shifted products :222-228).  The float64 kernels of sr_ct.hip
// (k_ct_rfft, k_ct_fft) are wider than the reference; this one matches its type and is what the pipeline runs.
//
// Why a float32 transform needs care, and what is done about it.  S[d] = sum_j (u_j.u_{j+d})^2 is a sum of ordinary
// autocorrelations of signals a_c (products of two components).  A float32 transform carries a rounding error proportional
// to the RMS of what it transforms, and the a_c have a large constant part (x^2 averages 1/3; an ordered bond vector has
// <xy> != 0): transformed as they are, C(t) comes out at 2e-7 and dC(t) misses its bar.  So, per (chunk, signal):
//   * a constant m_c (the chunk mean of a_c, computed in float32 -- ANY constant gives an exact identity) is subtracted
//     while the signal is formed, d_c = fma(a, b, -m_c), one rounding;
//   * the d_c go through the float32 transforms, power spectra summed with the weights w_c, one inverse transform;
//   * what the subtraction removed is restored in float64, exactly:  with a_c = d_c + m_c on the chunk's F frames,
//         sum_c w_c sum_j a_c[j] a_c[j+d] = sum_c w_c sum_j d_c[j] d_c[j+d] + (PE[F-d] + PE[F] - PE[d]) + (F - d) K,
//         e[j] = sum_c w_c m_c d_c[j],   PE = prefix sums of e,   K = sum_c w_c m_c^2:
//     ONE scalar signal e (accumulated in registers while the d_c are formed) and ONE float64 scan of it per series.
//     PE[F-d] + PE[F] - PE[d] = G[0] + G[d] with G[d] = sum_{j=d}^{F-1-d} e[j], the window that shrinks from both ends: a
//     suffix scan over HALF the series.
// Signals: the five components of the traceless tensor Q = u (x) u - |u|^2/3  (2z^2-x^2-y^2, x^2-y^2, xy, xz, yz with
// weights 1/6, 1/2, 2, 2, 2; exact for any vectors) plus the trace term |u|^2 |u'|^2 / 3.  For unit vectors (|u|^2 = 1 + eps,
// |eps| < 5e-7: float32-rounded unit vectors) the trace term is (F - d)/3 + window sums of eps/3 + O(F eps^2): eps/3 simply
// joins e[j] and 1/3 joins K -- five forward transforms instead of six.  A series with any |eps| >= 5e-7 (zero vectors of
// the 0/0 guard of vecnorm_NDarray, unnormalised input) transforms |u|^2 (weight 1/3) as a sixth signal, same launch.
// Accuracy on the cfg3 trajectory: C(t) 2-3e-8 relative, dC(t) 2-3e-9 absolute against the float64 oracle -- the class of the
// direct float32 kernel k_ct_palmer; the reference chain (fit orders, R1/R2/NOE) stays within 2e-7 (scripts/dev/
// f32fft_feasibility.py, tests/test_gpu_chain.py).
//
// Structure: one 256-thread workgroup per (chunk, vector) series, real-input transforms of half length H = N1 * 256
// (N1 = 12: M = 6144, F <= 4096;  N1 = 16: M = 8192, F <= 5461), three steps N1 x 16 x 16 in registers with two LDS
// exchanges, spectrum by frequency pairs -- the layout of k_ct_rfft (sr_ct.hip), every index mapping the same.  A complex
// float is 8 bytes: the transform image is 26 KB (N1 = 12) and a thread's 12 + 16 points fit 128 VGPRs -- FOUR waves per SIMD
// and four workgroups per CU where the float64 kernel has two and two.
#include "sr_internal.h"

namespace {

// ---- packed complex float32 arithmetic ------------------------------------------------------------------------------
// A complex number is one 64-bit register pair (re, im) and every operation below is ONE or TWO v_pk_*_f32 instructions.
// Why it matters (scripts/dev/probe/pk_rate.hip, profiles/r05_pk_issue_rate.txt): a gfx950 wave issues a plain float32
// VALU instruction every ~6 cycles whatever its neighbours do, a packed one every ~7 -- twice the arithmetic per issue; the SIMD
// only saturates on plain instructions with three waves issuing at once, and this kernel, with its barriers and LDS round
// trips, has about one.  (Left to the SLP vectoriser the packing costs a v_mov per operand pair and 270 B of scratch; here
// the swaps and sign flips of complex arithmetic ride on the op_sel / neg modifiers, spelled out in inline assembly where
// the compiler does not fold them itself.)
typedef float c32 __attribute__((ext_vector_type(2)));     // .x = re, .y = im
#define SR_PK __device__ __forceinline__
SR_PK c32 pk_fma(c32 a, c32 b, c32 c) { return __builtin_elementwise_fma(a, b, c); }
SR_PK c32 splat(float c) { return c32{c, c}; }
// a + (-i) b = (a.re + b.im, a.im - b.re)
SR_PK c32 add_mi(c32 a, c32 b)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + i b = (a.re - b.im, a.im + b.re)
SR_PK c32 add_pi(c32 a, c32 b)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (-i) (a - b) = (a.im - b.im, b.re - a.re)
SR_PK c32 mi_sub(c32 a, c32 b)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b), a - conj(b)
SR_PK c32 add_conj(c32 a, c32 b)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
SR_PK c32 sub_conj(c32 a, c32 b)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (s.re + t.im, s.re - t.im) and (s.im - t.re, s.im + t.re): real and imaginary parts of the pair (s - i t, s + i t)
SR_PK c32 pair_re(c32 s, c32 t)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(s), "v"(t));
    return r;
}
SR_PK c32 pair_im(c32 s, c32 t)
{
    c32 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(s), "v"(t));
    return r;
}
// a * w, both variable
SR_PK c32 cmulf(c32 a, c32 w)
{
    c32 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                       // (a.im w.im, a.im w.re)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a1 *= w1, a2 *= w2 as ONE block: a packed instruction that consumes the result of the packed instruction right before it costs
// a wait state (the compiler puts an s_nop between the two halves of cmulf); two products interleaved need none.
SR_PK void cmulf2(c32 &a1, c32 w1, c32 &a2, c32 w2)
{
    c32 t1, t2;
    asm("v_pk_mul_f32 %2, %0, %4 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %1, %5 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %0, %4, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]\n\t"
        "v_pk_fma_f32 %1, %1, %5, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
        : "+v"(a1), "+v"(a2), "=&v"(t1), "=&v"(t2) : "v"(w1), "v"(w2));
}
// d * (C - i S), C and S compile-time constants: (d.re C + d.im S, d.im C - d.re S)
template <int CBITS, int SBITS>
SR_PK c32 mul_const(c32 d)
{
    const float C = __builtin_bit_cast(float, CBITS), S = __builtin_bit_cast(float, SBITS);
    return pk_fma(d.yx, c32{S, -S}, d * splat(C));
}
constexpr int fbits(float f) { return __builtin_bit_cast(int, f); }

// d * exp(-2 pi i E / 32), E a compile-time constant
template <int E>
SR_PK c32 mulf_w32(c32 d)
{
    constexpr float c[16] = {1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f, 0.7071067811865476f,
                             0.5555702330196023f, 0.38268343236508984f, 0.19509032201612833f, 0.0f,
                             -0.1950903220161282f, -0.3826834323650897f, -0.555570233019602f, -0.7071067811865475f,
                             -0.8314696123025453f, -0.9238795325112867f, -0.9807852804032304f};
    constexpr float s[16] = {0.0f, 0.19509032201612825f, 0.3826834323650898f, 0.5555702330196022f, 0.7071067811865475f,
                             0.8314696123025452f, 0.9238795325112867f, 0.9807852804032304f, 1.0f, 0.9807852804032304f,
                             0.9238795325112867f, 0.8314696123025455f, 0.7071067811865476f, 0.5555702330196022f,
                             0.3826834323650899f, 0.1950903220161286f};
    if constexpr (E == 0) return d;
    else if constexpr (E == 8) return c32{d.y, -d.x};
    else if constexpr (E == 4) return add_mi(d, d) * splat(c[4]);              // sqrt(1/2) (d.re + d.im, d.im - d.re)
    else if constexpr (E == 12) return add_pi(d, d) * splat(-c[4]);            // -sqrt(1/2) (d.re - d.im, d.im + d.re)
    else return mul_const<fbits(c[E]), fbits(s[E])>(d);
}
SR_PK c32 mulf_w32_rt(c32 d, int e)      // e = 0..15 known after unrolling
{
    switch (e) {
        case 0: return mulf_w32<0>(d);
        case 1: return mulf_w32<1>(d);
        case 2: return mulf_w32<2>(d);
        case 3: return mulf_w32<3>(d);
        case 4: return mulf_w32<4>(d);
        case 5: return mulf_w32<5>(d);
        case 6: return mulf_w32<6>(d);
        case 7: return mulf_w32<7>(d);
        case 8: return mulf_w32<8>(d);
        case 9: return mulf_w32<9>(d);
        case 10: return mulf_w32<10>(d);
        case 11: return mulf_w32<11>(d);
        case 12: return mulf_w32<12>(d);
        case 13: return mulf_w32<13>(d);
        case 14: return mulf_w32<14>(d);
        default: return mulf_w32<15>(d);
    }
}
// d * exp(-2 pi i E / 24)
template <int E>
SR_PK c32 mulf_w24(c32 d)
{
    constexpr float c[15] = {1.0f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f, 0.5000000000000001f,
                             0.25881904510252074f, 0.0f, -0.25881904510252063f, -0.4999999999999998f, -0.7071067811865475f,
                             -0.8660254037844387f, -0.9659258262890682f, -1.0f, -0.9659258262890683f, -0.8660254037844388f};
    constexpr float s[15] = {0.0f, 0.25881904510252074f, 0.49999999999999994f, 0.7071067811865475f, 0.8660254037844386f,
                             0.9659258262890683f, 1.0f, 0.9659258262890683f, 0.8660254037844387f, 0.7071067811865476f,
                             0.49999999999999994f, 0.258819045102521f, 0.0f, -0.2588190451025208f, -0.4999999999999997f};
    if constexpr (E == 0) return d;
    else if constexpr (E == 6) return c32{d.y, -d.x};
    else if constexpr (E == 12) return -d;
    else return mul_const<fbits(c[E]), fbits(s[E])>(d);
}

template <int LOGN>
__host__ __device__ constexpr int bitrevf(int p)
{
    int r = 0;
    for (int b = 0; b < LOGN; ++b) r |= ((p >> b) & 1) << (LOGN - 1 - b);
    return r;
}

template <int LOGN, int S, int BLK, int J>
struct FftStageF {
    __device__ static __forceinline__ void run(c32 *v)
    {
        constexpr int N = 1 << LOGN;
        constexpr int half = N >> (S + 1);
        constexpr int i = BLK * 2 * half + J;
        constexpr int E = ((J << S) * (32 / N)) & 15;
        const c32 a = v[i], b = v[i + half];
        v[i] = a + b;
        if constexpr (E == 8) v[i + half] = mi_sub(a, b);              // the -i of the twiddle rides on the subtraction
        else v[i + half] = mulf_w32<E>(a - b);
        if constexpr (J + 1 < half) FftStageF<LOGN, S, BLK, J + 1>::run(v);
        else if constexpr (BLK + 1 < (1 << S)) FftStageF<LOGN, S, BLK + 1, 0>::run(v);
        else if constexpr (S + 1 < LOGN) FftStageF<LOGN, S + 1, 0, 0>::run(v);
    }
};
// in-register radix-2 decimation-in-frequency transform of N = 2^LOGN <= 16 points; v[p] ends up holding X[rev(p)]
template <int LOGN>
__device__ __forceinline__ void fftf_reg(c32 *v)
{
    FftStageF<LOGN, 0, 0, 0>::run(v);
}

template <int N1>
struct FStage1 {                                           // N1 = 4, 8, 16
    static constexpr int LOG = N1 == 4 ? 2 : (N1 == 8 ? 3 : 4);
    __host__ __device__ static constexpr int k1(int p) { return bitrevf<LOG>(p); }
    __device__ static __forceinline__ void run(c32 *v) { fftf_reg<LOG>(v); }
};
template <int B>
__device__ __forceinline__ void dft3f_col12(c32 *v, c32 (*y)[4])
{
    constexpr float h = 0.8660254037844386f;             // sqrt(3)/2
    const c32 x0 = v[B], x1 = v[4 + B], x2 = v[8 + B];
    const c32 t = x1 + x2, d = x1 - x2;
    const c32 m = pk_fma(splat(-0.5f), t, x0);
    const c32 hd = d * splat(h);
    y[0][B] = x0 + t;
    y[1][B] = mulf_w24<2 * B>(add_mi(m, hd));              // w_12^B (m - i h d)
    y[2][B] = mulf_w24<4 * B>(add_pi(m, hd));              // w_12^(2B) (m + i h d)
    if constexpr (B + 1 < 4) dft3f_col12<B + 1>(v, y);
}
template <>
struct FStage1<12> {                                       // n1 = 4 a + b, k1 = ka + 3 kb
    __host__ __device__ static constexpr int k1(int p) { return (p >> 2) + 3 * bitrevf<2>(p & 3); }
    __device__ static __forceinline__ void run(c32 *v)
    {
        c32 y[3][4];
        dft3f_col12<0>(v, y);
#pragma unroll
        for (int ka = 0; ka < 3; ++ka) {
            fftf_reg<2>(y[ka]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * ka + q] = y[ka][q];
        }
    }
};

// Hide a value's provenance from the optimiser (see sr_ct.hip: thread-invariant twiddles and addresses would otherwise be
// computed once per kernel, parked in registers the loop does not have, and spilled).
__device__ __forceinline__ c32 opaquef(c32 z)
{
    asm volatile("" : "+v"(z));
    return z;
}
__device__ __forceinline__ int opaquei(int t)
{
    asm volatile("" : "+v"(t));
    return t;
}

// v[p] *= base^k(p), base = w_H^tid.  The thread reads base^1, base^2, base^4, base^8 from four float32 tables (each entry
// rounded once from float64) and multiplies them up: k = 3, 5, 6, 9, 10, 12 cost one float32 complex multiply (one more
// rounding), 7, 11, 13, 14 two, 15 three -- against four levels of a multiply tree started from base alone, and against 43
// float64 instructions + 22 conversions per transform for exactly rounded powers.
template <int N, class KOF>
__device__ __forceinline__ void applyf_twiddles(c32 *v, const c32 *tw, int tid)
{
    c32 pw[16];
    pw[1] = opaquef(tw[tid]);
    pw[2] = opaquef(tw[256 + tid]);
    pw[4] = opaquef(tw[512 + tid]);
    pw[8] = opaquef(tw[768 + tid]);
    static_assert(N == 4 || N == 8 || N == 12 || N == 16, "the step-1 sizes");
    static_assert(KOF::k1(0) == 0, "entry 0 carries no twiddle");
    if constexpr (N == 4) {                                // k1(p) = 0, 2, 1, 3
        pw[3] = cmulf(pw[1], pw[2]);
        cmulf2(v[1], pw[2], v[2], pw[1]);
        v[3] = cmulf(v[3], pw[3]);
        return;
    } else if constexpr (N == 8) {                         // k1(p) = 0, 4, 2, 6, 1, 5, 3, 7
        pw[3] = pw[1]; pw[5] = pw[1];
        cmulf2(pw[3], pw[2], pw[5], pw[4]);
        pw[6] = pw[2]; pw[7] = pw[3];
        cmulf2(pw[6], pw[4], pw[7], pw[4]);
        v[1] = cmulf(v[1], pw[4]);
#pragma unroll
        for (int p = 2; p + 1 < N; p += 2) cmulf2(v[p], pw[KOF::k1(p)], v[p + 1], pw[KOF::k1(p + 1)]);
        return;
    }
    pw[3] = pw[1]; pw[5] = pw[1]; pw[6] = pw[2]; pw[9] = pw[1]; pw[10] = pw[2];
    cmulf2(pw[3], pw[2], pw[5], pw[4]);
    cmulf2(pw[6], pw[4], pw[9], pw[8]);
    pw[7] = pw[3]; pw[11] = pw[3];
    if constexpr (N == 12) {
        cmulf2(pw[10], pw[8], pw[7], pw[4]);
        cmulf2(pw[11], pw[8], v[1], pw[KOF::k1(1)]);       // k1(1) = 6
    } else {
        pw[12] = pw[4];
        cmulf2(pw[10], pw[8], pw[12], pw[8]);
        pw[13] = pw[5]; pw[14] = pw[6];
        cmulf2(pw[7], pw[4], pw[11], pw[8]);
        cmulf2(pw[13], pw[8], pw[14], pw[8]);
        pw[15] = pw[7];
        cmulf2(pw[15], pw[8], v[1], pw[KOF::k1(1)]);       // k1(1) = 8
    }
    // v[p] *= pw[k1(p)], two at a time (p = 0 carries no twiddle, p = 1 went with the last power)
#pragma unroll
    for (int p = 2; p + 1 < N; p += 2) cmulf2(v[p], pw[KOF::k1(p)], v[p + 1], pw[KOF::k1(p + 1)]);
}

// Tables (computed in float64, rounded once; per N1): w_H^(j t) for j = 1, 2, 4, 8 (step-1 twiddle bases), w_256^t, w_M^t, t < 256
struct Ct32Tab {
    float w1[4][2 * 256];
    float w2[2 * 256];
    float w3[2 * 256];
};
__host__ __device__ constexpr int f32_tab_set(int N1) { return N1 / 4 - 1; }       // N1 = 4, 8, 12, 16 -> 0 .. 3
__global__ void k_ct32_init_table(Ct32Tab *tab)          // tab[N1 / 4 - 1], H = 256 N1
{
    const int t = threadIdx.x, set = blockIdx.x;
    const double H = 1024.0 * (double)(set + 1);
    double sn, cs;
    for (int j = 0; j < 4; ++j) {
        const int e = (t << j) % (int)H;                   // exact argument reduction
        sincospi(2.0 * (double)e / H, &sn, &cs);
        tab[set].w1[j][2 * t] = (float)cs;
        tab[set].w1[j][2 * t + 1] = (float)-sn;
    }
    sincospi(2.0 * (double)t / 256.0, &sn, &cs);
    tab[set].w2[2 * t] = (float)cs;
    tab[set].w2[2 * t + 1] = (float)-sn;
    sincospi((double)t / H, &sn, &cs);
    tab[set].w3[2 * t] = (float)cs;
    tab[set].w3[2 * t + 1] = (float)-sn;
}

struct Ct32Args {
    const float *soa;
    int64_t Npad;
    const int64_t *chunk_start;   // device, may be null
    const Ct32Tab *tab;
    double *psum;                 // (nV, R, Lp)
    int R, F, L, Lp;
};

__host__ __device__ constexpr int f32_img_slots(int N1) { return 256 * N1 + 256 + 16; }     // as rfft_lds_slots (sr_ct.hip)

#ifdef SR_CT32_STAMPS           // development: s_memtime stamps around the phases of a pass, summed per wave, left behind lag L of the
#define SR_STAMP(I) { const long long t_ = __builtin_amdgcn_s_memtime(); stamp_acc[I] += t_ - stamp_t; stamp_t = t_; }   // series' sums
#else
#define SR_STAMP(I)
#endif
#ifdef SR_CT32_EXP_NOBAR        // timing experiment: no workgroup barriers (wrong results)
#define SR_CT32_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define SR_CT32_SYNC() __syncthreads()
#endif

// One half-length transform: the thread's N1 inputs v[] (natural order, element tid + 256 n1) -> for the 16 N1 threads
// (k1, k2a) = (tid >> 4, tid & 15), k1 < N1: w[p] = X[k1 + N1 (k2a + 16 rev4(p))].  The caller has made sure nobody still
// reads the LDS image; on return every thread has read what it needs from it (row tid is the thread's own).
// SPECTRUM: the forward transforms of the loop -- the half of the row a partner thread reads (frequencies k2b >= 8: the partner
// of (k, k2b' < 8) sits at 15 - k2b'; thread 0 pairs k2b' with 16 - k2b': 9 .. 15 and the pad slot, where it leaves Z[0]) goes
// back to the thread's own row in frequency order, INSIDE the block that computed it: behind the block all 16 values would be live
// at once across a branch merge (10 registers over the budget of four waves per SIMD, spilled and reloaded every pass).
template <int N1, bool SPECTRUM>
__device__ __forceinline__ void rfft32_workgroup(c32 *v, c32 *w, c32 *lds, const c32 *tw1, int tid
#ifdef SR_CT32_STAMPS
                                                 , long long *stamp_acc, long long &stamp_t
#endif
)
{
    FStage1<N1>::run(v);
    applyf_twiddles<N1, FStage1<N1>>(v, tw1, tid);
    {
        c32 *b = lds + tid + (tid >> 4);                          // element k1*256 + tid, one pad slot per 16
#pragma unroll
        for (int p = 0; p < N1; ++p) b[272 * FStage1<N1>::k1(p)] = v[p];
    }
    SR_STAMP(0)
    SR_CT32_SYNC();
    SR_STAMP(1)
    const int k1 = tid >> 4, lo = tid & 15;
    const bool act = k1 < N1;
    if (act) {
        c32 u[16];
        const c32 *b = lds + 272 * k1 + lo;                       // element k1*256 + lo + 16 h -> + 17 h
#pragma unroll
        for (int h = 0; h < 16; ++h) u[h] = b[17 * h];
#ifdef SR_CT32_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SR_STAMP(2)
#endif
        fftf_reg<4>(u);
        {
            const c32 *tw = lds + f32_img_slots(N1) + lo;         // w_256^(lo k2a) at [k2a*16 + lo], filled at kernel start
#pragma unroll
            for (int p = 1; p < 15; p += 2) cmulf2(u[p], tw[16 * bitrevf<4>(p)], u[p + 1], tw[16 * bitrevf<4>(p + 1)]);
            u[15] = cmulf(u[15], tw[16 * bitrevf<4>(15)]);
        }
        // in place (see k_ct_rfft): the cells this thread has read are the ones it writes
        c32 *bw = lds + 272 * k1 + lo;
#pragma unroll
        for (int p = 0; p < 16; ++p) bw[17 * bitrevf<4>(p)] = u[p];
        SR_STAMP(3)
    }
    // row tid was written by the thread's own 16-lane group, and the LDS operations of one wave complete in order: only the
    // compiler must not move the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (act) {
        const c32 *b = lds + 17 * tid;
#pragma unroll
        for (int e = 0; e < 16; ++e) w[e] = b[e];
#ifdef SR_CT32_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SR_STAMP(4)
#endif
        fftf_reg<4>(w);
        SR_STAMP(5)
        if (SPECTRUM) {
            c32 *bo = lds + 17 * tid;
#pragma unroll
            for (int p = 0; p < 16; ++p)
                if (bitrevf<4>(p) >= 8) bo[bitrevf<4>(p)] = w[p];
            if (tid == 0) bo[16] = w[0];
        }
    }
}

constexpr float kUnitTolF = 5e-7f;

// inclusive float64 prefix scan over the 64 lanes of a wave: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then
// row_bcast15 / row_bcast31; lanes without a source add 0
__device__ __forceinline__ double wave_scan_f64(double v)
{
    union U { double d; int i[2]; };
#define SR_SCAN_STEP(CTRL, RM)                                                                   \
    {                                                                                            \
        U a_, b_;                                                                                \
        a_.d = v;                                                                                \
        b_.i[0] = __builtin_amdgcn_update_dpp(0, a_.i[0], CTRL, RM, 0xF, false);                 \
        b_.i[1] = __builtin_amdgcn_update_dpp(0, a_.i[1], CTRL, RM, 0xF, false);                 \
        v += b_.d;                                                                               \
    }
    SR_SCAN_STEP(0x111, 0xF) SR_SCAN_STEP(0x112, 0xF) SR_SCAN_STEP(0x114, 0xF) SR_SCAN_STEP(0x118, 0xF)
    SR_SCAN_STEP(0x142, 0xA) SR_SCAN_STEP(0x143, 0xC)
#undef SR_SCAN_STEP
    return v;
}
// wave totals of float values with the same sequence (lane 63 ends up with the sum)
__device__ __forceinline__ float wave_total_f32(float v)
{
#define SR_TOT_STEP(CTRL, RM) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xF, false));
    SR_TOT_STEP(0x111, 0xF) SR_TOT_STEP(0x112, 0xF) SR_TOT_STEP(0x114, 0xF) SR_TOT_STEP(0x118, 0xF) SR_TOT_STEP(0x142, 0xA) SR_TOT_STEP(0x143, 0xC)
#undef SR_TOT_STEP
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v)       // v >= 0: 0 is neutral
{
#define SR_MAX_STEP(CTRL, RM) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xF, false)));
    SR_MAX_STEP(0x111, 0xF) SR_MAX_STEP(0x112, 0xF) SR_MAX_STEP(0x114, 0xF) SR_MAX_STEP(0x118, 0xF) SR_MAX_STEP(0x142, 0xA) SR_MAX_STEP(0x143, 0xC)
#undef SR_MAX_STEP
    return v;
}

// planes of signal c (0 = x, 1 = y, 2 = z): c = 1, 2: x y;  3: x z;  4: y z;  5 (|u|^2): x y, then z
__device__ __forceinline__ int f32_plane_a(int c) { return c == 4 ? 1 : 0; }
__device__ __forceinline__ int f32_plane_b(int c) { return c == 3 || c == 4 ? 2 : 1; }

// The transforms' inputs.  The epilogue forms them a second time for e[j] and relies on getting the SAME bits: one definition.
SR_PK c32 f32_sig0(c32 x, c32 y, c32 z, float m) { return pk_fma(z + z, z, -pk_fma(x, x, pk_fma(y, y, splat(m)))); }   // 2 z^2 - x^2 - y^2 - m
SR_PK c32 f32_sig1(c32 x, c32 y, float m) { return pk_fma(x, x, -pk_fma(y, y, splat(m))); }                            // x^2 - y^2 - m
SR_PK c32 f32_sigp(c32 a, c32 b, float m) { return pk_fma(a, b, splat(-m)); }                                          // a b - m
SR_PK c32 f32_sig5(c32 x, c32 y, c32 z, float m) { return pk_fma(z, z, pk_fma(x, x, pk_fma(y, y, splat(-m)))); }       // |u|^2 - m

#ifndef SR_CT32_PF
#define SR_CT32_PF 0
#endif
#ifndef SR_CT32_NT
#define SR_CT32_NT 2           // cache policy (aux bits of the buffer loads: 2 = nt) of the plane loads whose next use is two or more
                              // passes away -- z in the prologue, y of signal 2, x of signal 3, everything in the epilogue.  The planes a
                              // pass re-reads do not fit the L2 of an XCD at four workgroups per CU (128 x 48 KB); with the far reuses
                              // out of the way the near ones hit: 2.30 -> 1.53 GB fetched per launch (PMC, 0 = every load temporal),
                              // same duration alone and in the pipeline
#endif
#ifndef SR_CT32_WAVES16
#define SR_CT32_WAVES16 3        // the same for the N1 = 16 kernel (M = 8192: 16 input blocks): 146 VGPRs, no scratch, three workgroups per CU
#endif
#ifndef SR_CT32_WAVES
#define SR_CT32_WAVES 4          // waves per SIMD the N1 = 12 kernel is compiled for: 124 VGPRs, no scratch, four workgroups per CU
#endif
// FULL: the chunk fills the loaded input blocks exactly (F = 512 NZ) and frames 2m, 2m + 1 share an aligned 8 bytes: no
// frame masks, 8-byte loads only (cfg3 / cfg4: F = 4096 with N1 = 12).
template <int N1, bool FULL>
__global__ __launch_bounds__(256, (N1 == 16 ? SR_CT32_WAVES16 : SR_CT32_WAVES)) void k_ct_rfft32(Ct32Args a)
{
    extern __shared__ __align__(16) unsigned char f32_smem[];
    c32 *lds = reinterpret_cast<c32 *>(f32_smem);
    constexpr int H = N1 * 256, M = 2 * H;
    // input blocks (of 512 frames) that can hold frames: F + L <= M with L = F/2, i.e. F <= 1365 / 2730 / 4096 / 5461 for N1 = 4 / 8 / 12 / 16
    constexpr int NZ = N1 == 4 ? 3 : (N1 == 8 ? 6 : (N1 == 12 ? 8 : 16));
    constexpr int IPT = N1 == 4 ? 3 : (N1 == 8 ? 6 : (N1 == 12 ? 8 : 11));     // scan: half-series elements per thread (256 IPT >= ceil(F/2))
    c32 *tw1 = lds + f32_img_slots(N1) + 256;                      // 4 x 256 step-1 twiddle bases: w_H^(j tid), j = 1, 2, 4, 8
    c32 *tw3 = tw1 + 1024;                                         // 256 x w_M^t: the spectrum step's twiddle of thread (k1, k2a) at k1 + N1 k2a
    float *aux = reinterpret_cast<float *>(tw3 + 256);             // [0 .. 32): wave partial sums; [32 .. 49): m_c, w_c m_c, weight of eps;
                                                                   // [50]: P[H/2] (thread 0's self-paired frequency); [52 .. 54): K (double)
    const int tid0 = threadIdx.x;
    const int v = blockIdx.x / a.R, r = blockIdx.x - v * a.R;
    const int F = FULL ? 512 * NZ : a.F, L = FULL ? 256 * NZ : a.L;      // FULL: compile-time (most of the back transform's
                                                                         // outputs are lags beyond L, and fold away)
    const int64_t start = a.chunk_start ? a.chunk_start[r] : (int64_t)r * F;
#ifdef SR_CT32_EXP_HOTLOADS     // timing experiment (wrong results): every series reads the first series' samples -- all loads hit the caches
    const float *px = a.soa + (start & 1);
#else
    const float *px = a.soa + ((int64_t)v * 3 + 0) * a.Npad + start;
#endif
    const bool even = FULL || ((start | a.Npad | (int64_t)F) & 1) == 0;   // frames 2m, 2m + 1 of every plane share an aligned 8 bytes

    // buffer resources that cover exactly the chunk's F frames of a plane: a frame past the chunk reads as 0
#define SR_F32_LOAD1(DST, PLANE, T) SR_F32_LOADP(DST, PLANE, T, 0, NZ, 0)
#define SR_F32_LOADR(DST, PLANE, T, LO, HI) SR_F32_LOADP(DST, PLANE, T, LO, HI, 0)
#define SR_F32_LOAD1NT(DST, PLANE, T) SR_F32_LOADP(DST, PLANE, T, 0, NZ, SR_CT32_NT)
#define SR_F32_LOADP(DST, PLANE, T, LO, HI, POL)                                                 \
    {                                                                                            \
        const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(                    \
            const_cast<float *>(px + (int64_t)__builtin_amdgcn_readfirstlane(PLANE) * a.Npad), (short)0, F * 4, 0x00020000); \
        if (even) {                                                                              \
            _Pragma("unroll") for (int n1 = (LO); n1 < (HI); ++n1)                              \
                DST[n1] = __builtin_bit_cast(c32, __builtin_amdgcn_raw_buffer_load_b64(rs_, 8 * ((T) + 256 * n1), 0, (POL))); \
        } else {                                                                                 \
            _Pragma("unroll") for (int n1 = (LO); n1 < (HI); ++n1) {                            \
                const int ob_ = 8 * ((T) + 256 * n1);                                            \
                DST[n1] = c32{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, ob_, 0, (POL))),           \
                              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_, ob_ + 4, 0, (POL)))};      \
            }                                                                                    \
        }                                                                                        \
    }
    // frames behind the chunk hold 0 in every signal (the zero padding of the transform) and in e
#define SR_F32_MASK(D0, D1, T, N1_)                                                              \
    if (!FULL) {                                                                                 \
        D0 = 2 * ((T) + 256 * (N1_)) < F ? D0 : 0.f;                                             \
        D1 = 2 * ((T) + 256 * (N1_)) + 1 < F ? D1 : 0.f;                                         \
    }
    {
        const int j = ((tid0 & 15) * (tid0 >> 4)) & 255;
        lds[f32_img_slots(N1) + tid0] = c32{a.tab->w2[2 * j], a.tab->w2[2 * j + 1]};
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) tw1[256 * jj + tid0] = c32{a.tab->w1[jj][2 * tid0], a.tab->w1[jj][2 * tid0 + 1]};
        tw3[tid0] = c32{a.tab->w3[2 * tid0], a.tab->w3[2 * tid0 + 1]};
        if (tid0 == 0) aux[50] = 0.f;
    }

    c32 sig[N1];                  // input of the next transform (entries >= NZ stay zero)
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) sig[n1] = c32{0.f, 0.f};
    int nsig;
    {
        // ---- prologue: chunk means of the signals, eps = |u|^2 - 1, signal 0 ----
        c32 xr[NZ], yr[NZ], zr[NZ];
        SR_F32_LOAD1(xr, 0, tid0)
        SR_F32_LOAD1(yr, 1, tid0)
        SR_F32_LOAD1NT(zr, 2, tid0)
        c32 s0 = {0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0, s4 = s0, s5 = s0;
        float emax = 0.f;
#pragma unroll
        for (int n1 = 0; n1 < NZ; ++n1) {
            const c32 x = xr[n1], y = yr[n1], z = zr[n1];
            const c32 xx = x * x, yy = y * y, zz = z * z, q = xx + yy;
            s0 += (zz + zz) - q;
            s1 += xx - yy;
            s2 = pk_fma(x, y, s2);
            s3 = pk_fma(x, z, s3);
            s4 = pk_fma(y, z, s4);
            s5 += q + zz;
            // |u|^2 - 1 to 1e-7: only the unit-vector test (5e-7 +- 1e-7 either way is a valid choice of algorithm)
            c32 ev = (q + zz) - splat(1.0f);
            SR_F32_MASK(ev.x, ev.y, tid0, n1)
            emax = fmaxf(emax, fmaxf(fabsf(ev.x), fabsf(ev.y)));
        }
        const float t0 = wave_total_f32(s0.x + s0.y), t1 = wave_total_f32(s1.x + s1.y), t2 = wave_total_f32(s2.x + s2.y);
        const float t3 = wave_total_f32(s3.x + s3.y), t4 = wave_total_f32(s4.x + s4.y), t5 = wave_total_f32(s5.x + s5.y);
        emax = wave_max_f32(emax);
        if ((tid0 & 63) == 63) {
            float *o = aux + 8 * (tid0 >> 6);
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = t5; o[6] = emax;
        }
        __syncthreads();
        // m_c: the chunk mean with its low four mantissa bits cleared -- any constant serves, and this one makes 3 m_c and
        // 12 m_c exact in float32, so that the SAME w_c m_c enters e[j] (float32) and K (float64)
        float m[6];
        const float invF = 1.0f / (float)F;
#pragma unroll
        for (int c = 0; c < 6; ++c)
            m[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
                       __builtin_bit_cast(int, ((aux[c] + aux[8 + c]) + (aux[16 + c] + aux[24 + c])) * invF) & (int)0xFFFFFFF0));
        const float mx = fmaxf(fmaxf(aux[6], aux[14]), fmaxf(aux[22], aux[30]));
        const bool unit = __builtin_amdgcn_readfirstlane(mx < kUnitTolF ? 1 : 0) != 0;
        nsig = unit ? 5 : 6;
        // weights x 6 (1/6, 1/2, 2, 2, 2 and 1/3 for the trace): exact in float32; the lag sums are divided by 6 at the end
        const float wgt[6] = {1.0f, 3.0f, 12.0f, 12.0f, 12.0f, 2.0f};
        double Kc = unit ? 2.0 : 0.0;          // 6 x (sum_c w_c m_c^2 (+ 1/3 for unit vectors)); kept in LDS until the epilogue
#pragma unroll
        for (int c = 0; c < 6; ++c)
            if (c < 5 || !unit) Kc = fma((double)wgt[c] * (double)m[c], (double)m[c], Kc);
        if (tid0 == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                aux[32 + c] = m[c];
                aux[40 + c] = wgt[c] * m[c];
            }
            aux[48] = unit ? 2.0f : 0.f;                       // 6 / 3: the weight of eps in e[j]
            *reinterpret_cast<double *>(aux + 52) = Kc;
        }
#pragma unroll
        for (int n1 = 0; n1 < NZ; ++n1) {
            c32 d = f32_sig0(xr[n1], yr[n1], zr[n1], m[0]);
            SR_F32_MASK(d.x, d.y, tid0, n1)
            sig[n1] = d;
        }
    }
    // power spectrum by pairs of frequencies (k, H - k), see k_ct_rfft: W2[q] = (P[k], P[H - k]) for the thread's k with k2b = q
    c32 W2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) W2[q] = c32{0.f, 0.f};
#ifdef SR_CT32_STAMPS
    long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = __builtin_amdgcn_s_memtime();
    const long long stamp_begin = stamp_t;
#define SR_STAMP_ARGS , stamp_acc, stamp_t
#else
#define SR_STAMP_ARGS
#endif
#pragma unroll 1
    for (int c = 0; c < nsig; ++c) {
        asm volatile("" ::: "memory");
        const int tid = opaquei(tid0);
        const int k1 = (tid >> 4) & 15, k2a = tid & 15;
        const bool act = k1 < N1;
        const int pt = k1 != 0 ? (N1 - k1) * 16 + (15 - k2a) : (k2a != 0 ? 16 - k2a : 0);
        const int off0 = tid == 0 ? 1 : 0;
        // the samples of the next signal: issued at the top of the pass (SR_CT32_PF = 1: a whole transform ahead), behind the
        // transform (2), or where they are consumed (0)
        const int cn = c + 1;
        c32 ar[NZ], br[NZ];
        if (SR_CT32_PF == 1 && cn < nsig) {
            SR_F32_LOAD1(ar, f32_plane_a(cn), tid)
            SR_F32_LOAD1(br, f32_plane_b(cn), tid)
        }
        c32 w[16];
        rfft32_workgroup<N1, true>(sig, w, lds, tw1, tid SR_STAMP_ARGS);
        // (the transform has left the partner-visible half of the thread's row in frequency order; thread 0 pairs k = 0 with "k = H",
        // which is Z[0] again, read from the pad slot behind its row: the general formula then gives P[0] and P[H], w_M^0 = 1)
        SR_STAMP(6)
        if (SR_CT32_PF == 2 && cn < nsig) {
            SR_F32_LOAD1(ar, f32_plane_a(cn), tid)
            SR_F32_LOAD1(br, f32_plane_b(cn), tid)
        }
        SR_CT32_SYNC();
        SR_STAMP(7)
        if (act) {
            const float wgt = c == 0 ? 0.25f : (c == 1 ? 0.75f : (c == 5 ? 0.5f : 3.0f));    // 6 x weight / 4
            const c32 *b = lds + 17 * pt + off0;
            const c32 wb = tw3[k1 + N1 * k2a];
            // S = Z[k] + conj Z[H-k], D = Z[k] - conj Z[H-k], T = w_M^k D:  4 |A[k]|^2 = |S - i T|^2, 4 |A[H-k]|^2 = |S + i T|^2
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                const c32 zk0 = w[bitrevf<4>(q)], zm0 = b[15 - q], zk1 = w[bitrevf<4>(q + 1)], zm1 = b[14 - q];
                const c32 S0 = add_conj(zk0, zm0), S1 = add_conj(zk1, zm1);
                c32 T0 = sub_conj(zk0, zm0), T1 = sub_conj(zk1, zm1);
                cmulf2(T0, mulf_w32_rt(wb, q), T1, mulf_w32_rt(wb, q + 1));
                const c32 A0 = pair_re(S0, T0), B0 = pair_im(S0, T0), A1 = pair_re(S1, T1), B1 = pair_im(S1, T1);
                W2[q] = pk_fma(splat(wgt), pk_fma(A0, A0, B0 * B0), W2[q]);
                W2[q + 1] = pk_fma(splat(wgt), pk_fma(A1, A1, B1 * B1), W2[q + 1]);
            }
            if (off0) {                                            // k = H/2 (k2b = 8) mirrors onto itself
                const c32 zk = w[bitrevf<4>(8)];
                const c32 D = {0.0f, 2.0f * zk.y};
                const c32 T = cmulf(D, mulf_w32_rt(wb, 8));
                const float pr = 2.0f * zk.x + T.y, pi = -T.x;
                aux[50] = fmaf(wgt, fmaf(pr, pr, pi * pi), aux[50]);
            }
        }
        SR_STAMP(8)
        // the next signal (w is dead here): d = a b - m_c
        if (cn < nsig) {
            asm volatile("" ::: "memory");
            const float mc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aux[32 + cn])));
            // the samples in groups of eight input blocks (NZ = 16: two groups, the second loaded behind the first one's signal --
            // all sixteen at once are 64 registers on top of the signal's 32)
#pragma unroll
            for (int g0 = 0; g0 < NZ; g0 += 8) {
                constexpr int GE = 8 < NZ ? 8 : NZ;         // blocks per group
                if (g0 > 0) asm volatile("" ::: "memory");
                if (SR_CT32_PF == 0 || g0 > 0) {
                    // cache policy by the distance to the plane's next use (SR_CT32_NT): y of signal 2 (next: signal 4) and x of
                    // signal 3 (next: the epilogue) are not wanted again for two passes
                    if (SR_CT32_NT != 0 && cn == 3) {
                        SR_F32_LOADP(ar, 0, tid, g0, g0 + GE, SR_CT32_NT)
                    } else {
                        SR_F32_LOADR(ar, f32_plane_a(cn), tid, g0, g0 + GE)
                    }
                    if (SR_CT32_NT != 0 && cn == 2) {
                        SR_F32_LOADP(br, 1, tid, g0, g0 + GE, SR_CT32_NT)
                    } else {
                        SR_F32_LOADR(br, f32_plane_b(cn), tid, g0, g0 + GE)
                    }
                }
                if (cn == 1) {                                     // x^2 - y^2
#pragma unroll
                    for (int n1 = g0; n1 < g0 + GE; ++n1) ar[n1] = f32_sig1(ar[n1], br[n1], mc);
                } else if (cn == 5) {                              // not unit vectors: |u|^2 (rare; the z load is exposed)
                    c32 zr[NZ];
                    SR_F32_LOADR(zr, 2, tid, g0, g0 + GE)
#pragma unroll
                    for (int n1 = g0; n1 < g0 + GE; ++n1) ar[n1] = f32_sig5(ar[n1], br[n1], zr[n1], mc);
                } else {
#pragma unroll
                    for (int n1 = g0; n1 < g0 + GE; ++n1) ar[n1] = f32_sigp(ar[n1], br[n1], mc);
                }
#pragma unroll
                for (int n1 = g0; n1 < g0 + GE; ++n1) {
                    c32 d = ar[n1];
                    SR_F32_MASK(d.x, d.y, tid, n1)
                    sig[n1] = d;
                }
            }
#pragma unroll
            for (int n1 = NZ; n1 < N1; ++n1) sig[n1] = c32{0.f, 0.f};
        }
        SR_STAMP(9)
        SR_CT32_SYNC();
        SR_STAMP(10)
    }
#ifdef SR_CT32_STAMPS
    const long long stamp_loop_end = stamp_t;
#endif

    // ---- back: Y[k] = (P[k] + P[H-k]) + i (P[k] - P[H-k]) conj(w_M^k), through the same transform (see k_ct_rfft) ----
    const int tid = opaquei(tid0);
    const int k1 = (tid >> 4) & 15, k2a = tid & 15;
    const bool act = k1 < N1;
    const int pt = k1 != 0 ? (N1 - k1) * 16 + (15 - k2a) : (k2a != 0 ? 16 - k2a : 0);
    const int off0 = tid == 0 ? 1 : 0;
    if (act) {
        c32 *bk = lds + k1 + (N1 + 1) * k2a;                                    // own frequencies, column k2b = q
        c32 *bm = lds + (pt >> 4) + (N1 + 1) * (pt & 15) + 16 * (N1 + 1) * off0;  // the partner's, column 15 - q (+ 1 for thread 0)
        const c32 wb = tw3[k1 + N1 * k2a];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float E = W2[q].x + W2[q].y, d = W2[q].x - W2[q].y;
            const c32 wk = mulf_w32_rt(wb, q);                                  // (cos, -sin)
            bk[16 * (N1 + 1) * q] = c32{fmaf(d, wk.y, E), d * wk.x};
            if (!(q == 0 && off0)) bm[16 * (N1 + 1) * (15 - q)] = c32{fmaf(-d, wk.y, E), d * wk.x};
        }
        if (off0) bk[16 * (N1 + 1) * 8] = c32{2.0f * aux[50], 0.0f};             // k = H/2: E = 2 P, d = 0
    }
    __syncthreads();
    c32 w[16];
    {
        c32 yin[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const int k = tid + 256 * n1;
            yin[n1] = lds[k + k / N1];
        }
        __syncthreads();
        rfft32_workgroup<N1, false>(yin, w, lds, tw1, tid SR_STAMP_ARGS);
    }
    __syncthreads();                                     // every read of the image is done: it now serves the scan

    // ---- the mean terms (x 6), float64: Tt[d] = (F - d) K + G[0] + G[d],  G[d] = sum_{j=d}^{F-1-d} e[j] ----
    // e[j] = sum_c w_c m_c d_c[j] (+ eps_j / 3 for unit vectors), here, from the samples once more: the d_c are formed by the
    // SAME float32 operations as the transforms' inputs (bit-identical), eps in float64.  (Carried through the transform loop in
    // registers instead, e costs 2 NZ VGPRs the loop does not have at four waves per SIMD: 150-290 B of scratch.)
    float *E = reinterpret_cast<float *>(lds);
    double *Tt = reinterpret_cast<double *>(lds);
    double *tot = reinterpret_cast<double *>(aux);       // wave totals (the float sums of the prologue are dead)
    {
        c32 xr[NZ], yr[NZ], zr[NZ];
        SR_F32_LOAD1NT(xr, 0, tid)
        SR_F32_LOAD1NT(yr, 1, tid)
        SR_F32_LOAD1NT(zr, 2, tid)
        float mm[6], wm[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            mm[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aux[32 + c])));
            wm[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aux[40 + c])));
        }
        const float weps = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, aux[48])));
        const bool unit = nsig == 5;
#pragma unroll
        for (int n1 = 0; n1 < NZ; ++n1) {
            const c32 x = xr[n1], y = yr[n1], z = zr[n1];
            c32 acc;
            if (unit) {
                acc.x = weps * (float)fma((double)x.x, (double)x.x, fma((double)y.x, (double)y.x, fma((double)z.x, (double)z.x, -1.0)));
                acc.y = weps * (float)fma((double)x.y, (double)x.y, fma((double)y.y, (double)y.y, fma((double)z.y, (double)z.y, -1.0)));
            } else {
                acc = splat(wm[5]) * f32_sig5(x, y, z, mm[5]);
            }
            acc = pk_fma(splat(wm[0]), f32_sig0(x, y, z, mm[0]), acc);
            acc = pk_fma(splat(wm[1]), f32_sig1(x, y, mm[1]), acc);
            acc = pk_fma(splat(wm[2]), f32_sigp(x, y, mm[2]), acc);
            acc = pk_fma(splat(wm[3]), f32_sigp(x, z, mm[3]), acc);
            acc = pk_fma(splat(wm[4]), f32_sigp(y, z, mm[4]), acc);
            SR_F32_MASK(acc.x, acc.y, tid, n1)
            *reinterpret_cast<c32 *>(E + 2 * (tid + 256 * n1)) = acc;
        }
    }
    __syncthreads();
    // thread t owns i = IPT b .. IPT b + IPT - 1 with b = 255 - t: an inclusive PREFIX scan over t is the suffix sum over i.
    // h_i = e_i + e_{F-1-i} (i < F-1-i), e_i (the centre frame of an odd F), 0 beyond
    const int i0 = IPT * (255 - tid);
    double sfx[IPT], incl;
    {
        double hsum[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int i = i0 + k, j = F - 1 - i;
            const float ei = E[min(i, F - 1)], ej = E[max(j, 0)];
            hsum[k] = (i <= j ? (double)ei : 0.0) + (i < j ? (double)ej : 0.0);
        }
        sfx[IPT - 1] = hsum[IPT - 1];
#pragma unroll
        for (int k = IPT - 2; k >= 0; --k) sfx[k] = hsum[k] + sfx[k + 1];
        incl = wave_scan_f64(sfx[0]);
        if ((tid & 63) == 63) tot[tid >> 6] = incl;
    }
    __syncthreads();                                     // every read of E is done, the wave totals are there
    {
        const int wave = tid >> 6;
        double off = incl - sfx[0];
#pragma unroll
        for (int w2 = 0; w2 < 3; ++w2) off += w2 < wave ? tot[w2] : 0.0;
        const double G0 = (tot[0] + tot[1]) + (tot[2] + tot[3]);
        const double Kc = *reinterpret_cast<const double *>(aux + 52);
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int d = i0 + k;
            if (d <= L) Tt[d] = fma((double)(F - d), Kc, G0 + (sfx[k] + off));
        }
        if (tid == 0 && L >= 256 * IPT) Tt[L] = fma((double)(F - L), Kc, G0);
    }
    __syncthreads();
    if (act) {
        double *out = a.psum + ((int64_t)v * a.R + r) * a.Lp;
        const double inv = 1.0 / (double)M, sixth = 1.0 / 6.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int m = k1 + N1 * (k2a + 16 * bitrevf<4>(p));
            const int le = 2 * m, lod = 2 * m - 1;
            if (le >= 1 && le <= L) out[le] = fma((double)w[p].x, inv, Tt[le]) * sixth;
            if (lod >= 1 && lod <= L) out[lod] = fma((double)w[p].y, inv, Tt[lod]) * sixth;
        }
    }
#ifdef SR_CT32_STAMPS
    if ((tid & 63) == 0) {             // behind lag L of this series' sums (the stride leaves 128 unused slots): 16 values per wave
        double *dbg = a.psum + ((int64_t)v * a.R + r) * a.Lp + L + 2 + 16 * (tid >> 6);
        for (int i = 0; i < 11; ++i) dbg[i] = (double)stamp_acc[i];
        dbg[11] = (double)(stamp_loop_end - stamp_begin);
        dbg[12] = (double)(__builtin_amdgcn_s_memtime() - stamp_begin);
    }
#endif
#undef SR_F32_LOAD1
#undef SR_F32_LOAD1NT
#undef SR_F32_LOADR
#undef SR_F32_LOADP
#undef SR_F32_MASK
}

template <int N1>
constexpr size_t f32_lds_bytes()
{
    // transform image + the 16 x 16 step-2 twiddles + the 4 x 256 step-1 bases + the 256 spectrum twiddles + reduction scratch
    return (size_t)(f32_img_slots(N1) + 256 + 1024 + 256) * sizeof(c32) + 64 * sizeof(float);
}

template <int N1, bool FULL>
int launch_ct_rfft32_h(sr_ctx *ctx, const Ct32Args &a, int64_t series)
{
    size_t lds = f32_lds_bytes<N1>();
    static_assert(f32_lds_bytes<N1>() <= 64 * 1024, "k_ct_rfft32: the image is meant to fit the default LDS grant");
    // option ct_wg_per_cu (0 = as many as fit): a larger LDS request caps the workgroups a CU holds, which leaves registers free on
    // every SIMD for the waves of the bandwidth kernels that run beside this one in the pipeline (pack, histogram)
    if (ctx->ct_wg_per_cu > 0) {
        const size_t cap = ((size_t)160 * 1024 / (size_t)ctx->ct_wg_per_cu) & ~(size_t)1023;
        if (cap > lds && cap <= 64 * 1024) lds = cap;
    }
    // the scan's tables share the image: F floats, then L + 1 doubles
    constexpr size_t Fmax = N1 == 4 ? 1365 : (N1 == 8 ? 2730 : (N1 == 12 ? 4096 : 5461));
    static_assert((size_t)f32_img_slots(N1) * sizeof(c32) >= (Fmax + 3) * 4, "k_ct_rfft32: E does not fit the image");
    static_assert((size_t)f32_img_slots(N1) * sizeof(c32) >= (Fmax / 2 + 2) * 8, "k_ct_rfft32: Tt does not fit the image");
    hipLaunchKernelGGL((k_ct_rfft32<N1, FULL>), dim3((unsigned)series), dim3(256), lds, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}

}  // namespace

// Called by sr_ct_palmer_sums_f32_dev (sr_ct.hip): ct_fft = 3 (default) for 4096 < F + L <= 8192, ct_fft = 4 for every 1024 < F + L <= 8192.
// (Why the shorter chunks keep float64 transforms by default: at cfg2's size the launch is 80 us either way, and the drop-in chain THROUGH
// ITS TEXT FILES -- _Ctint.dat keeps 8 digits -- reproduces the reference's printed digits only with float64 C(t); with float32 C(t) one
// of 16 cfg2 residues lands 1.1e-6 from the reference's through-files table, tests/test_gpu_cli.py.)  chunk starts: cs_host (may be
// null: chunk r starts at r F) is what decides the aligned fast path, cs_dev is what the kernel reads.
int sr_launch_ct_rfft32(sr_ctx *ctx, const float *soa, int64_t Npad, const int64_t *cs_host, const int64_t *cs_dev, double *psum,
                        int R, int F, int L, int Lp, int64_t series)
{
    Ct32Tab *tab = (Ct32Tab *)sr_workspace(ctx, SR_WS_FFT32, 4 * sizeof(Ct32Tab));
    if (!tab) return -5;
    if (!ctx->fft32_table_ready) {
        hipLaunchKernelGGL(k_ct32_init_table, dim3(4), dim3(256), 0, ctx->stream, tab);
        SR_HIP(hipGetLastError());
        SR_HIP(hipStreamSynchronize(ctx->stream));      // once per context: later launches may come on other streams
        ctx->fft32_table_ready = 1;
    }
    Ct32Args a;
    a.soa = soa; a.Npad = Npad; a.chunk_start = cs_dev; a.psum = psum;
    a.R = R; a.F = F; a.L = L; a.Lp = Lp;
    bool aligned = (Npad & 1) == 0 && (F & 1) == 0 && (((uintptr_t)soa) & 7) == 0;
    if (cs_host)
        for (int r = 0; r < R; ++r) aligned = aligned && (cs_host[r] & 1) == 0;
    const int need = F + L;
    SR_REQUIRE(need > 1024 && need <= 8192, -3, "k_ct_rfft32: F=%d outside the transform lengths", F);
    if (need <= 2048) {
        a.tab = tab + f32_tab_set(4);
        return launch_ct_rfft32_h<4, false>(ctx, a, series);
    }
    if (need <= 4096) {
        a.tab = tab + f32_tab_set(8);
        return launch_ct_rfft32_h<8, false>(ctx, a, series);
    }
    if (need <= 6144) {
        a.tab = tab + f32_tab_set(12);
        return aligned && F == 4096 ? launch_ct_rfft32_h<12, true>(ctx, a, series) : launch_ct_rfft32_h<12, false>(ctx, a, series);
    }
    a.tab = tab + f32_tab_set(16);
    return launch_ct_rfft32_h<16, false>(ctx, a, series);
}
