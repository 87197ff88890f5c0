// sr_fit.hip -- kernel 3b: multi-exponential C(t) model, residual/Jacobian evaluation and a
// batched bounded non-linear least-squares solver that runs entirely on the device.
//
// Reference call site: conduct_curve_fitting, fitting_Ct_functions.py:322-324
//     curve_fit(curvefit_exponential, DeltaT, Decay, sigma=dDecay, p0=..., bounds=(0, [1..,tauMax..,1]))
// The arithmetic of that call lives in a third-party dependency that is not part of the reference
// tree: scipy (requirements.txt:2 "scipy>=0.17.1", unpinned; the build container has 1.15.3).
// With bounds, curve_fit -> least_squares(method='trf', jac='2-point', x_scale=1, ftol=xtol=gtol=1e-8,
// max_nfev=100*n, tr_solver='exact').  k_trf restates that published algorithm (Branch, Coleman & Li
// 1999 trust-region-reflective with More's 1977 trust-region sub-problem iteration, as implemented in
// scipy/optimize/_lsq/trf.py::trf_bounds and _lsq/common.py, and the forward-difference Jacobian of
// scipy/optimize/_numdiff.py::approx_derivative) so that the GPU follows the same iteration path:
// same step sizes h = sqrt(eps)*sign(x)*max(1,|x|) adjusted to the bounds, same Coleman-Li scaling,
// same reflective / Cauchy step selection, same radius update and termination tests.  The only
// structural difference: scipy factorises the augmented Jacobian with an SVD, here the n x n normal
// matrix B = (J d)^T (J d) + diag_h is Cholesky-factorised for every Levenberg parameter alpha that
// More's iteration visits -- phi(alpha) and phi'(alpha) are the same functions, evaluated differently.
//
// Mapping: one workgroup of W waves per residue (option fit_waves; default 2 since round 5, four workgroups per CU).  The L
// data points are spread over the W*64 threads for model evaluation and the J^T J / J^T f reductions (wave shuffles, then a
// fixed-order combine through LDS); the n <= 11 dimensional algebra is workgroup-uniform, run by the leader wave (the kernel
// is templated on n so every small array is statically indexed).  The critical path of a fit is serial (up to 100 n
// dependent iterations): more waves per residue shorten an evaluation a little (18.8 us at W = 4, 20.4 at 2) and idle
// during the leader's algebra; with the chip full W = 2 costs 0.40 ms per 512-residue batch, W = 4 0.60.
// Floating-point contraction is OFF for this file (spinrelax_amd/build.py): every fused multiply-add is written as
// fma(), so k_trf<n> and the search_order<n> instances of k_order_search -- separately compiled copies of the same
// solver -- round identically whatever the optimiser does around them (with -ffp-contract=fast the two differed in
// the last bit of a few sums, enough to send an ill-conditioned nine-parameter fit down another path).
#include "sr_internal.h"
#include <type_traits>

namespace {

constexpr double kEPS = 2.220446049250313e-16;
constexpr int kNmax = 11;

struct FitArgs {
    const double *t, *y, *sigma;      // (nRes, L); sigma may be null
    const double *p0;                 // (nRes, N)
    const unsigned char *skip;        // (nRes) or null: 1 = leave this residue alone
    double tau_max;
    int nRes, L, max_nfev, jac_mode;  // jac_mode 0: 2-point finite differences (scipy default), 1: analytic
    double ftol, xtol, gtol;
    double *popt, *pcov, *chisq;      // (nRes,N), (nRes,N,N), (nRes)
    int *status, *nfev;               // (nRes)
    double *fws;                      // (nRes, L) weights, only used when a residue does not fit into LDS
    int geo;                          // option fit_geo: 1 = exponentials of a uniform time grid by multiplication (Residue::stage)
};

__device__ __forceinline__ double wsum(double v) { return sr_wave_sum_f64(v); }

// Instruction-count reductions (bit-identical results unless noted; cfg3 benchmark batch, model-order search alone /
// saturated per batch / whole pipeline step at 20 steps):
//   exp(-t/tau) as one fused sequence (the device library's exp() without the selects a non-positive argument cannot
//       need) on top of shared-divisor (Markstein) divisions for t/tau and for the forward-difference quotient, divisors
//       and reciprocals held in SGPRs (uni()), no branch:                               8.95 / 0.836 / 3.40 ms (20 steps)
//       (steps on the way, each a build switch at the time: Markstein for both quotients 9.25 / 0.868 / 3.45, for t/tau only
//       9.39 / 0.905, IEEE divisions 9.87 / 0.937 / 3.55)
//   SR_FIT_LEADER=1 (default) only the leader wave runs the n x n algebra and broadcasts the trial point through LDS: 2 % in
//                            every measure once the reductions below were out of the way (6.69 / 0.728 against 6.84 / 0.746).
//   SR_FIT_REDUCE_MANY=1 (default) the 54 lane sums of a Jacobian by the register-halving reduction of sr_internal.h
//                            (different association: the fits move in their last bits): 8.95 -> 6.85 / 0.832 -> 0.745
//   Round 5, NOT bit-identical (within 1e-15 per exponential; every parity test re-run, per-trial tallies refreshed):
//   exp(-t/tau) on a uniform time grid by multiplication along a thread's points (Residue::stage, eval_f, eval_jac; option
//       fit_geo): one exp() per exponential and thread instead of one per point -- 2 instructions per exponential and point
//       instead of 23: 0.709 -> 0.607 ms saturated, an evaluation of the slowest residue 22.6 -> 18.7 us;
//   W = 2 and 37 KB of LDS per residue (C(t), weights, one time per thread; the times themselves are only read by the
//       exp()-per-point paths, from global memory) -> four workgroups per CU: 0.607 -> 0.397 ms saturated.
// Tried and dropped (bit-identical): evaluating trial points with a fused model + Jacobian pass (an accepted step then needs
// no second pass: the exponentials of the model once per evaluation instead of twice): 0.716 -> 0.692 ms saturated but
// 6.58 -> 6.79 ms alone (the slowest residue rejects many trial steps, each now paying for a Jacobian; scratch 872 -> 1 192 B).
// Round 1 had tried Markstein with the reciprocals in VGPRs and a branch for huge quotients: the kernel lives at the
// 256-register limit of two waves per SIMD, the extra live values spilled inside the Jacobian loop and the branch kept
// the scheduler from interleaving the independent exponentials of a point (12.7 -> 16.8 ms).  Uniform values in SGPRs
// cost no vector registers; the Jacobian loop went from 535 to 437 instructions per point at n = 9.
#ifndef SR_FIT_LEADER
#define SR_FIT_LEADER 1
#endif
#ifndef SR_FIT_REDUCE_MANY
#define SR_FIT_REDUCE_MANY 1   // lane sums of J^T J and J^T f by the register-halving reduction (sr_internal.h)
#endif
#ifndef SR_FIT_LF_LDS
#define SR_FIT_LF_LDS 1      // Cholesky factor in a per-wave LDS area instead of 90 VGPRs
#endif
// Tried and dropped (round 3; the code is in the history, commit 1dd6a1d): the trust-region sub-problem LANE-PARALLEL on the
// leader wave (one matrix row per lane, readlane / DPP row reductions) instead of sequential on workgroup-uniform values:
// saturated 0.719 -> 0.704 ms per batch, scratch 780 -> 424 B per lane, every parity test passed -- but another summation
// order in the 9-term dot products sends the ONE nine-parameter fit of the benchmark batch that never converges to the
// evaluation limit (900 evaluations, 18 ms, instead of ~390 / 8 ms), which a 20-step run pays in full as pipeline drain.
// Tried and dropped (round 3): parking the solver's workgroup-uniform n-vectors (x, g, d, g_h, the trial point) in a per-wave LDS
// area across the model / Jacobian passes -- explicit stores before, volatile loads behind -- so that the register allocator
// has nothing of them to spill: it spilled MORE (1 256 B per lane instead of 776, 216 scratch loads per solver iteration instead
// of 75): the copies and the volatile loads lengthen other live ranges in the n x n algebra.  The frame that remains is the
// allocator's choice of victims in ~7 000 instructions of straight-line algebra, not a set of values one can name.
#ifndef SR_FIT_SETPRIO
#define SR_FIT_SETPRIO 3     // wave priority of the model-order search (0 = default priority)
#endif

// a / b for a divisor b shared by many numerators, with r = 1.0 / b computed once (correctly rounded).  q0 = a*r is
// within 2 ulp; one residual correction makes it faithful, the second gives the correctly rounded quotient
// (Markstein 1990; the same tail v_div_fmas_f64 executes) -- 5 instructions instead of the 11 of a full IEEE
// division with its scaling.  Exception the theorem leaves open: divisors whose significand is all ones.
// Divisor and reciprocal are workgroup-uniform: uni() moves them into SGPRs (v_fma_f64 takes one scalar operand), so
// the hoisted reciprocals cost no VGPRs in loops that already sit at the register limit.
__device__ __forceinline__ double uni(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double div_shared(double a, double b, double r)
{
    double q = a * r;
    double e = fma(-b, q, a);
    q = fma(e, r, q);
    e = fma(-b, q, a);
    return fma(e, r, q);
}
// exp(a / b) for a <= 0 < b with the quotient formed as above: the instruction sequence of the device library's
// exp() (ROCm 7.2 ocml, read off its ISA: argument reduction by ln2 hi/lo, degree-11 polynomial in Horner form, ldexp)
// minus what a non-positive argument cannot need -- the overflow select -- and with one v_max_f64 on the quotient in
// place of the underflow select (which doubles as the guard for huge / infinite / NaN quotients).  Same bits as
// exp(a / b) (scripts/dev/fit_dump.py: every fit of the fixtures ends on identical parameters); 23 instructions instead
// of 11 + 23.
extern __shared__ __align__(16) double fit_smem[];
// Measured and dropped (round 4): a table-driven exp -- 2^(j/64) from a 64-entry LDS table, degree-4 polynomial on |t| <= ln2/128,
// 5 float64 instructions fewer per exponential, below one ulp like this sequence but not the same bits: 0.743 -> 0.718 ms per
// batch with the chip full (-3.4 %: the exponentials are half of the point loops' float64 work, the point loops a quarter of an
// evaluation's time on the leader wave), every over-parameterised fit takes another path (order-9 evaluations 5 192 -> 5 205);
// not worth re-measuring every parity tally for 1 % of a step.
__device__ __forceinline__ double exp_neg_quotient(double a, double b, double r)
{
    // quotients below -1100 (and the NaN a zero tau makes of the corrected quotient: maxNum returns the other operand)
    // all end in ldexp(p, <= -1587) = 0, the value exp() has there
    const double q = fmax(div_shared(a, b, r), -1100.0);
    const double dn = rint(q * 0x1.71547652b82fep+0);
    double t = fma(-0x1.62e42fefa39efp-1, dn, q);
    t = fma(-0x1.abc9e3b39803fp-56, dn, t);
    double p = fma(0x1.ade156a5dcb37p-26, t, 0x1.28af3fca7ab0cp-22);
    p = fma(t, p, 0x1.71dee623fde64p-19);
    p = fma(t, p, 0x1.a01997c89e6b0p-16);
    p = fma(t, p, 0x1.a01a014761f6ep-13);
    p = fma(t, p, 0x1.6c16c1852b7b0p-10);
    p = fma(t, p, 0x1.1111111122322p-7);
    p = fma(t, p, 0x1.55555555502a1p-5);
    p = fma(t, p, 0x1.5555555555511p-3);
    p = fma(t, p, 0x1.000000000000bp-1);
    p = fma(t, p, 1.0);
    p = fma(t, p, 1.0);
    return ldexp(p, (int)dn);
}

// ---- model: curvefit_exponential, fitting_Ct_functions.py:419-427 -----------------------------
template <int N>
struct Model {
    static constexpr int K = N / 2;
    static constexpr bool kFreeS2 = (N % 2) == 1;

    __device__ static __forceinline__ double S2_of(const double *x)
    {
#pragma clang fp contract(off)
        if (kFreeS2) return x[N - 1];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += x[k];
        return 1.0 - s;
    }
    // e[k] = exp(-1.0*t/tau_k)
    __device__ static __forceinline__ void exps(const double *x, double t, double *e)
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < K; ++k) e[k] = exp((-1.0 * t) / x[K + k]);
    }
    // the same with the divisors tau_k and rtau[k] = 1.0 / tau_k hoisted out of the loop over the data points (SGPRs)
    __device__ static __forceinline__ void recips(const double *x, double *tau_u, double *rtau)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            tau_u[k] = uni(x[K + k]);
            rtau[k] = uni(1.0 / x[K + k]);
        }
    }
    __device__ static __forceinline__ void exps(const double *tau_u, const double *rtau, double t, double *e)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) e[k] = exp_neg_quotient(-1.0 * t, tau_u[k], rtau[k]);
    }
    __device__ static __forceinline__ double value(const double *x, const double *e)
    {
#pragma clang fp contract(off)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) s += x[k] * e[k];
        return S2_of(x) + s;
    }
};

// ---- small dense helpers on packed symmetric matrices (i >= j stored at i*(i+1)/2 + j) ----------
__host__ __device__ constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

template <int N>
__device__ __forceinline__ double dotN(const double *a, const double *b)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) s = fma(a[i], b[i], s);
    return s;
}
template <int N>
__device__ __forceinline__ double normN(const double *a) { return sqrt(dotN<N>(a, a)); }

// s^T B s
template <int N>
__device__ __forceinline__ double quadN(const double *B, const double *s)
{
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) r = fma(B[tri(i, j)], s[j], r);
        q = fma(s[i], r, q);
    }
    return q;
}
// u^T B s
template <int N>
__device__ __forceinline__ double bilinN(const double *B, const double *u, const double *s)
{
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double r = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) r = fma(B[tri(i, j)], s[j], r);
        q = fma(u[i], r, q);
    }
    return q;
}

// Cholesky of (B + alpha I): Lf holds the strict lower triangle of L, inv[i] = 1/L_ii (the factor is only
// ever used through multiplications by these reciprocals: one rsqrt per column instead of a sqrt and
// n divisions -- float64 sqrt/div are ~15-instruction dependent chains and dominated the serial part).
// Returns false when a pivot is not positive; lmin2 = smallest squared diagonal of L.
template <int N>
__device__ __forceinline__ bool cholN(const double *B, double alpha, double *Lf, double *inv, double &lmin2)
{
    bool ok = true;
    lmin2 = 1e300;
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = B[tri(i, j)] + (i == j ? alpha : 0.0);
#pragma unroll
            for (int k = 0; k < j; ++k) s = fma(-Lf[tri(i, k)], Lf[tri(j, k)], s);
            if (i == j) {
                if (!(s > 0.0)) { ok = false; s = 1.0; }
                lmin2 = fmin(lmin2, s);
                inv[i] = rsqrt(s);
            } else {
                Lf[tri(i, j)] = s * inv[j];
            }
        }
    }
    return ok;
}
// solve L z = b (forward)
template <int N>
__device__ __forceinline__ void fwdN(const double *Lf, const double *inv, const double *b, double *z)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s = fma(-Lf[tri(i, k)], z[k], s);
        z[i] = s * inv[i];
    }
}
// solve L^T p = z (backward)
template <int N>
__device__ __forceinline__ void bwdN(const double *Lf, const double *inv, const double *z, double *p)
{
#pragma unroll
    for (int i = N - 1; i >= 0; --i) {
        double s = z[i];
#pragma unroll
        for (int k = i + 1; k < N; ++k) s = fma(-Lf[tri(k, i)], p[k], s);
        p[i] = s * inv[i];
    }
}

// phi(alpha) = ||p(alpha)|| - Delta, phi' = -(p^T (B+alpha I)^-1 p)/||p||, p = -(B+alpha I)^-1 g.
// (common.py:phi_and_derivative expressed through the Cholesky factor: with L L^T = B + alpha I,
//  z = L^-1 g, p = -L^-T z, q = L^-1 p:  p^T (B+alpha I)^-1 p = ||q||^2.)
template <int N>
__device__ __forceinline__ void phi_from_factor(const double *Lf, const double *inv, const double *g, double Delta,
                                                double *p, double &phi, double &phi_prime)
{
    double z[N], q[N];
    fwdN<N>(Lf, inv, g, z);
    bwdN<N>(Lf, inv, z, p);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = -p[i];
    fwdN<N>(Lf, inv, p, q);
    const double pn = normN<N>(p);
    phi = pn - Delta;
    phi_prime = -dotN<N>(q, q) / pn;
}

// common.py:solve_lsq_trust_region
template <int N>
__device__ void solve_tr(const double *B, const double *g, int m, double Delta, double &alpha, double *p, double *lfbuf)
{
    // rank test of scipy: s_min > EPS*m*s_max on the singular values of the augmented Jacobian.  Here:
    // B is accepted as full rank when its Cholesky factorisation succeeds with pivots above the
    // equivalent threshold (EPS*m)^2 * max diag.
    double Lf_regs[SR_FIT_LF_LDS ? 1 : N * (N + 1) / 2], inv[N], lmin2;
    double *Lf = SR_FIT_LF_LDS ? lfbuf : Lf_regs;
    bool full_rank = cholN<N>(B, 0.0, Lf, inv, lmin2);
    if (full_rank) {
        double dmax = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) dmax = fmax(dmax, B[tri(i, i)]);
        const double thr = kEPS * (double)m;
        if (!(lmin2 > thr * thr * dmax)) full_rank = false;
    }
    double alpha_upper = normN<N>(g) / Delta;
    double alpha_lower = 0.0;
    if (full_rank) {
        double phi, phip;
        phi_from_factor<N>(Lf, inv, g, Delta, p, phi, phip);     // p = Gauss-Newton step
        if (phi <= 0.0) { alpha = 0.0; return; }                 // norm(p) <= Delta
        alpha_lower = -phi / phip;
    }
    if (!full_rank && alpha == 0.0) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (alpha < alpha_lower || alpha > alpha_upper) alpha = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double phi, phip;
        cholN<N>(B, alpha, Lf, inv, lmin2);
        phi_from_factor<N>(Lf, inv, g, Delta, p, phi, phip);
        if (phi < 0) alpha_upper = alpha;
        const double ratio = phi / phip;
        alpha_lower = fmax(alpha_lower, alpha - ratio);
        alpha -= (phi + Delta) * ratio / Delta;
        if (fabs(phi) < 0.01 * Delta) break;
    }
    {
        double z[N];
        cholN<N>(B, alpha, Lf, inv, lmin2);
        fwdN<N>(Lf, inv, g, z);
        bwdN<N>(Lf, inv, z, p);
    }
    const double sc = -Delta / normN<N>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] *= sc;
}

// common.py:step_size_to_bound (returns min step; hits[i] = sign(s_i) where the minimum is attained)
template <int N>
__device__ __forceinline__ double step_to_bound(const double *x, const double *s, const double *lb, const double *ub,
                                                int *hits)
{
    double steps[N], mn = INFINITY;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        steps[i] = INFINITY;
        if (s[i] != 0.0) steps[i] = fmax((lb[i] - x[i]) / s[i], (ub[i] - x[i]) / s[i]);
        mn = fmin(mn, steps[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) hits[i] = (steps[i] == mn) ? (s[i] > 0 ? 1 : (s[i] < 0 ? -1 : 0)) : 0;
    return mn;
}

// common.py:minimize_quadratic_1d
__device__ __forceinline__ void min_quad_1d(double a, double b, double lo, double hi, double c, double &t, double &y)
{
    t = lo;
    y = lo * (a * lo + b) + c;
    const double yh = hi * (a * hi + b) + c;
    if (yh < y) { y = yh; t = hi; }
    if (a != 0) {
        const double ex = -0.5 * b / a;
        if (lo < ex && ex < hi) {
            const double ye = ex * (a * ex + b) + c;
            if (ye < y) { y = ye; t = ex; }
        }
    }
}

// trf.py:select_step.  B = J_h^T J_h + diag_h, so every quadratic form of the original is a form in B.
template <int N>
__device__ void select_step(const double *x, const double *B, const double *g_h, double *p, double *p_h,
                            const double *d, double Delta, const double *lb, const double *ub, double theta,
                            double *step, double *step_h, double &predicted)
{
    bool inb = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double xn = x[i] + p[i];
        inb = inb && (xn >= lb[i]) && (xn <= ub[i]);
    }
    if (inb) {
        const double pv = 0.5 * quadN<N>(B, p_h) + dotN<N>(g_h, p_h);
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = p[i]; step_h[i] = p_h[i]; }
        predicted = -pv;
        return;
    }
    int hits[N];
    const double p_stride = step_to_bound<N>(x, p, lb, ub, hits);
    double r_h[N], r[N], xb[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        r_h[i] = hits[i] != 0 ? -p_h[i] : p_h[i];
        r[i] = d[i] * r_h[i];
        p[i] *= p_stride;
        p_h[i] *= p_stride;
        xb[i] = x[i] + p[i];
    }
    // intersect_trust_region(p_h, r_h, Delta): positive root
    double to_tr;
    {
        const double a = dotN<N>(r_h, r_h), b = dotN<N>(p_h, r_h), c = dotN<N>(p_h, p_h) - Delta * Delta;
        const double dd = sqrt(b * b - a * c);
        const double q = -(b + copysign(dd, b));
        const double t1 = q / a, t2 = c / q;
        to_tr = t1 < t2 ? t2 : t1;
    }
    int hits2[N];
    const double to_bound = step_to_bound<N>(xb, r, lb, ub, hits2);
    double r_stride = fmin(to_bound, to_tr), r_stride_l, r_stride_u;
    if (r_stride > 0) {
        r_stride_l = (1 - theta) * p_stride / r_stride;
        r_stride_u = (r_stride == to_bound) ? theta * to_bound : to_tr;
    } else {
        r_stride_l = 0;
        r_stride_u = -1;
    }
    double r_value;
    if (r_stride_l <= r_stride_u) {
        // build_quadratic_1d(J_h, g_h, r_h, s0=p_h, diag=diag_h)
        const double a = 0.5 * quadN<N>(B, r_h);
        const double b = dotN<N>(g_h, r_h) + bilinN<N>(B, p_h, r_h);
        const double c = 0.5 * quadN<N>(B, p_h) + dotN<N>(g_h, p_h);
        min_quad_1d(a, b, r_stride_l, r_stride_u, c, r_stride, r_value);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            r_h[i] = r_h[i] * r_stride + p_h[i];
            r[i] = r_h[i] * d[i];
        }
    } else {
        r_value = INFINITY;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { p[i] *= theta; p_h[i] *= theta; }
    const double p_value = 0.5 * quadN<N>(B, p_h) + dotN<N>(g_h, p_h);

    double ag_h[N], ag[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { ag_h[i] = -g_h[i]; ag[i] = d[i] * ag_h[i]; }
    const double to_tr2 = Delta / normN<N>(ag_h);
    int hits3[N];
    const double to_bound2 = step_to_bound<N>(x, ag, lb, ub, hits3);
    double ag_stride = (to_bound2 < to_tr2) ? theta * to_bound2 : to_tr2;
    double ag_value;
    {
        const double a = 0.5 * quadN<N>(B, ag_h);
        const double b = dotN<N>(g_h, ag_h);
        min_quad_1d(a, b, 0.0, ag_stride, 0.0, ag_stride, ag_value);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) { ag_h[i] *= ag_stride; ag[i] *= ag_stride; }

    if (p_value < r_value && p_value < ag_value) {
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = p[i]; step_h[i] = p_h[i]; }
        predicted = -p_value;
    } else if (r_value < p_value && r_value < ag_value) {
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = r[i]; step_h[i] = r_h[i]; }
        predicted = -r_value;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) { step[i] = ag[i]; step_h[i] = ag_h[i]; }
        predicted = -ag_value;
    }
}

// common.py:make_strictly_feasible
template <int N>
__device__ __forceinline__ void strictly_feasible(double *x, const double *lb, const double *ub, double rstep)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double lower_dist = x[i] - lb[i], upper_dist = ub[i] - x[i];
        bool lo_act, up_act;
        if (rstep == 0) {
            lo_act = x[i] <= lb[i];
            up_act = x[i] >= ub[i];
        } else {
            const double lt = rstep * fmax(1.0, fabs(lb[i])), ut = rstep * fmax(1.0, fabs(ub[i]));
            lo_act = lower_dist <= fmin(upper_dist, lt);
            up_act = upper_dist <= fmin(lower_dist, ut);
        }
        double xn = x[i];
        if (rstep == 0) {
            if (lo_act) xn = nextafter(lb[i], ub[i]);
            if (up_act) xn = nextafter(ub[i], lb[i]);
        } else {
            if (lo_act) xn = lb[i] + rstep * fmax(1.0, fabs(lb[i]));
            if (up_act) xn = ub[i] - rstep * fmax(1.0, fabs(ub[i]));
        }
        if (xn < lb[i] || xn > ub[i]) xn = 0.5 * (lb[i] + ub[i]);
        x[i] = xn;
    }
}

// Per-residue data access.  LDS = true: t, y and the weights 1/sigma of the residue live in LDS for the whole
// solve (3*L doubles after the reduction scratch) -- every model evaluation used to re-read them from L2/HBM
// with one wave per SIMD and nothing to hide the latency behind, which was most of the ~12 us an iteration cost
// even at n = 2.  LDS = false (L too long for LDS): global memory, weights precomputed into the work array.
constexpr int kRedStride = kNmax * (kNmax + 1) / 2 + kNmax + 2;   // doubles per wave in the reduction scratch
constexpr int kBcast = 16;
constexpr int kMat = kNmax * (kNmax + 1) / 2;                        // one packed symmetric n x n matrix
constexpr int kStateDoubles = 16;                                    // SearchState of k_order_search (workgroup-uniform)
// LDS of a workgroup: reduction scratch, broadcast area, matrices, search state, and -- when the residue is staged -- C(t) and the
// weights of its L points plus the times of the first W * 64 (one per thread: all the uniform-grid form needs; the paths that need
// every time read them from global memory).  L = 2048: 37.4 KB at W = 2 (four workgroups per CU), 39.7 KB at W = 4.
__host__ __device__ constexpr size_t fit_lds_doubles(int W, int64_t L_staged)
{
    return (size_t)W * kRedStride + kBcast + (2 + W) * kMat + kStateDoubles + (L_staged ? (size_t)W * 64 + 2 * (size_t)L_staged : 0);
}

template <int W, bool LDS>
struct Residue {
    static constexpr int NTH = W * 64;
    static constexpr int NW = W;
    static constexpr int BC = W * kRedStride;          // kBcast doubles: leader wave -> workgroup broadcast
    static constexpr int MA = BC + kBcast;             // J^T J of the current point (packed), shared by all threads
    static constexpr int MB = MA + kMat;               // the scaled trust-region matrix B
    static constexpr int LF = MB + kMat;               // one Cholesky factor per wave (every lane writes the same values)
    static constexpr int ST = LF + W * kMat;           // the model-order search's state (uniform; every thread writes the same values)
    static constexpr int RED = ST + kStateDoubles;     // start of the staged residue

    const double *tg, *yg, *wg;   // global (LDS == false)
    const double *sg;             // sigma of this residue (global) or null
    int L, tid;
    int geo;                      // the time axis is a uniform grid and a thread owns more than one point of it: see stage()
    double tstep;                 // t[NTH] - t[0], the distance between consecutive points of one thread

    __device__ __forceinline__ double *red() const { return fit_smem; }
    __device__ __forceinline__ double *bcast() const { return fit_smem + BC; }
    // The two n x n matrices every thread would otherwise hold in registers (90 VGPRs each at n = 9: the kernel sat
    // at the 256-register limit and spilled ~1 GB per launch to scratch).  They are workgroup-uniform, read with
    // broadcast ds_read_b64 whose addresses are known up front.
    __device__ __forceinline__ double *matA() const { return fit_smem + MA; }
    __device__ __forceinline__ double *matB() const { return fit_smem + MB; }
    __device__ __forceinline__ double *matLf() const { return fit_smem + LF + (tid >> 6) * kMat; }
    __device__ __forceinline__ double ld_t(int l) const { return tg[l]; }        // any time: global (exp() per point, chi^2)
    __device__ __forceinline__ double ld_t0() const { return LDS ? fit_smem[RED + tid] : tg[tid < L ? tid : 0]; }   // the thread's first
    __device__ __forceinline__ double ld_y(int l) const { return LDS ? fit_smem[RED + NTH + l] : yg[l]; }
    __device__ __forceinline__ double ld_w(int l) const { return LDS ? fit_smem[RED + NTH + L + l] : wg[l]; }

    // stage the residue: weights 1/sigma (curve_fit: transform = 1/sigma, _minpack_py.py:985), t, y
    __device__ void stage(const double *t, const double *y, const double *sg_, double *wbuf, int geo_allowed)
    {
        sg = sg_;
        if (LDS) fit_smem[RED + tid] = t[tid < L ? tid : 0];
        for (int l = tid; l < L; l += NTH) {
            const double w = sg ? 1.0 / sg[l] : 1.0;
            if (LDS) {
                fit_smem[RED + NTH + l] = y[l];
                fit_smem[RED + NTH + L + l] = w;
            } else {
                wbuf[l] = w;
            }
        }
        tg = t; yg = y; wg = wbuf;
        // Uniform grid?  A thread owns the points tid, tid + NTH, tid + 2 NTH, ...; when t[tid + j NTH] = t[tid] + j (t[NTH] - t[0]) to
        // rounding (8 ulp; dt * arange(L) is within 1.5), the exponentials of its points form a geometric sequence and the point
        // loops multiply instead of evaluating exp() per point (eval_f / eval_jac).  Any other time axis -- the interface takes
        // arbitrary t -- keeps the exp() per point, and so does L <= NTH (one point per thread: nothing to multiply).
        int odd = 0;
        tstep = 0.0;
        if (L > NTH) {
            tstep = t[NTH] - t[0];
            const double t0 = t[tid];
            int j = 0;
            for (int l = tid; l < L; l += NTH, ++j) {
                const double tl = t[l];
                if (!(fabs(tl - fma((double)j, tstep, t0)) <= 1.8e-15 * fabs(tl))) odd = 1;
            }
            if (!(tstep > 0.0)) odd = 1;
        }
        // workgroup OR through the reduction scratch (free here); the first barrier also publishes the staged residue
        const bool wave_odd = __builtin_amdgcn_ballot_w64(odd != 0) != 0;
        if ((tid & 63) == 0) red()[(tid >> 6) * kRedStride] = wave_odd ? 1.0 : 0.0;
        __syncthreads();
        double any = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) any += red()[w * kRedStride];
        geo = (any == 0.0 && L > NTH && geo_allowed) ? 1 : 0;
        __syncthreads();
    }

    // sum of v over the workgroup, identical in every thread (fixed combination order)
    template <int CNT>
    __device__ __forceinline__ void block_sums(double *vals) const
    {
        const int lane = tid & 63, wave = tid >> 6;
        double *r = red();
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            const double w = wsum(vals[k]);
            if (lane == 0) r[wave * kRedStride + k] = w;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CNT; ++k) {
            double acc = r[k];
#pragma unroll
            for (int w = 1; w < W; ++w) acc += r[w * kRedStride + k];
            vals[k] = acc;
        }
        __syncthreads();
    }
};

// cost = 0.5 f.f of the residuals f = w*(model - y); finite=false when any f is not finite
// Point cache (one-exponential models, N <= 3): the exponentials and the residual of a thread's first kCacheCap points as
// eval_f computed them.  A Jacobian is only ever evaluated at the point the last eval_f call was made at (the initial
// point, or the trial point that has just been accepted), so eval_jac takes them from here instead of recomputing them --
// the same values by construction.  2 x 8 doubles per thread; at N <= 3 the solver has the registers.
constexpr int kCacheCap = 8;
template <int N> constexpr bool fit_point_cache() { return N <= 3; }     // (N <= 5 measured: 48 more live registers, no gain)
template <int N>
struct PointCache {
    static constexpr int K = N / 2;
    double e[fit_point_cache<N>() ? kCacheCap * (K > 0 ? K : 1) : 1];
    double f[fit_point_cache<N>() ? kCacheCap : 1];
};

template <int N, class R>
__device__ __forceinline__ double eval_f(const R &T, const double *x, bool &finite, PointCache<N> &pc)
{
    constexpr int K = N / 2;
    constexpr bool CACHE = fit_point_cache<N>();
    using M = Model<N>;
    const int tid = T.tid, L = T.L;
    constexpr int NTH = R::NTH;
    double acc = 0.0, bad = 0.0;
    double tau_u[K > 0 ? K : 1], rtau[K > 0 ? K : 1];
    M::recips(x, tau_u, rtau);
    // GEO (uniform time grid, Residue::stage): exp(-t/tau) at the thread's first point, then one multiplication by
    // exp(-(t[NTH] - t[0])/tau) per further point -- the points of a thread are NTH grid steps apart.  Within 1e-15 of the
    // exp() per point (<= 7 products), 2 instructions per exponential and point instead of 23.
    auto run = [&](auto GEO) {
        double er[K > 0 ? K : 1], Rk[K > 0 ? K : 1];
        if (GEO) {
            M::exps(tau_u, rtau, T.ld_t0(), er);
            M::exps(tau_u, rtau, T.tstep, Rk);
#pragma unroll
            for (int k = 0; k < K; ++k) Rk[k] = uni(Rk[k]);
        }
        auto point = [&](int l, double *e) {
            if (GEO) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    e[k] = er[k];
                    er[k] *= Rk[k];
                }
            } else {
                M::exps(tau_u, rtau, T.ld_t(l), e);
            }
            double f;
            {
#pragma clang fp contract(off)
                f = T.ld_w(l) * (M::value(x, e) - T.ld_y(l));
            }
            if (!isfinite(f)) bad = 1.0; else acc = fma(f, f, acc);
            return f;
        };
        int l0 = tid;
        if (CACHE) {
#pragma unroll
            for (int j = 0; j < kCacheCap; ++j) {
                const int l = tid + j * NTH;
                if (l < L) {
                    double e[K > 0 ? K : 1];
                    pc.f[j] = point(l, e);
#pragma unroll
                    for (int k = 0; k < K; ++k) pc.e[j * K + k] = e[k];
                }
            }
            l0 = tid + kCacheCap * NTH;
        }
        for (int l = l0; l < L; l += NTH) {
            double e[K > 0 ? K : 1];
            point(l, e);
        }
    };
    if (T.geo) run(std::true_type{});
    else run(std::false_type{});
    double v[2] = {acc, bad};
    T.template block_sums<2>(v);
    finite = v[1] == 0.0;
    return 0.5 * v[0];
}

// J^T J (packed) and J^T f at x (f is recomputed from the same expression eval_f uses: identical bits)
template <int N, class R>
__device__ __forceinline__ void eval_jac(const R &T, const double *x, const double *lb, const double *ub, int mode, double *A,
                                         double *g, const PointCache<N> &pc)
{
    constexpr bool CACHE = fit_point_cache<N>();
    constexpr int K = N / 2;
    constexpr int NT = N * (N + 1) / 2;
    using M = Model<N>;
    const int tid = T.tid, L = T.L;
    constexpr int NTH = R::NTH;
    double h[N], dx[N], rdx[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        // _numdiff.py:_compute_absolute_step + _adjust_scheme_to_bounds('1-sided', num_steps=1)
        const double sgn = x[i] >= 0 ? 1.0 : -1.0;
        double hi = 1.4901161193847656e-08 * sgn * fmax(1.0, fabs(x[i]));
        const double ld = x[i] - lb[i], ud = ub[i] - x[i];
        const double xp = x[i] + hi;
        const bool violated = (xp < lb[i]) || (xp > ub[i]);
        const bool fitting = fabs(hi) <= fmax(ld, ud);
        if (violated && fitting) hi = -hi;
        if (!fitting) hi = (ud >= ld) ? ud : -ld;
        h[i] = hi;
        dx[i] = (x[i] + hi) - x[i];
        dx[i] = uni(dx[i]);
        rdx[i] = uni(1.0 / dx[i]);
    }
    double tau_u[K > 0 ? K : 1], rtau[K > 0 ? K : 1], tauh_u[K > 0 ? K : 1], rtau_h[K > 0 ? K : 1];
    M::recips(x, tau_u, rtau);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        tauh_u[k] = uni(x[K + k] + h[K + k]);
        rtau_h[k] = uni(1.0 / (x[K + k] + h[K + k]));
    }
    double acc[NT + N];                       // J^T J (packed) followed by J^T f: reduced together below
    double *Aacc = acc, *gacc = acc + NT;
#pragma unroll
    for (int i = 0; i < NT + N; ++i) acc[i] = 0.0;
    // GEO (uniform time grid): the exponentials of a thread's points by multiplication, as in eval_f -- at x (the same products:
    // identical bits to eval_f's) and at the forward-difference points tau_k + h_k
    auto run = [&](auto GEO) {
    double er[K > 0 ? K : 1], Rk[K > 0 ? K : 1], erh[K > 0 ? K : 1], Rh[K > 0 ? K : 1];
    if (GEO) {
        const double t0 = T.ld_t0();
        M::exps(tau_u, rtau, t0, er);
        M::exps(tau_u, rtau, T.tstep, Rk);
        if (mode == 0) {
            M::exps(tauh_u, rtau_h, t0, erh);
            M::exps(tauh_u, rtau_h, T.tstep, Rh);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            Rk[k] = uni(Rk[k]);
            if (mode == 0) Rh[k] = uni(Rh[k]);
        }
    }
    // one data point: `cached` >= 0 takes the exponentials and the residual at x from the point cache (slot `cached`)
    auto point = [&](int l, int cached) {
        const double w = T.ld_w(l), yl = T.ld_y(l);
        double tl = 0.0;
        if (!GEO || mode != 0) tl = T.ld_t(l);          // (the uniform-grid form with finite differences never needs the time itself)
        double e[K > 0 ? K : 1], Jr[N], f0;
        if (CACHE && cached >= 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) e[k] = pc.e[cached * K + k];
            f0 = pc.f[cached];
        } else {
            if (GEO) {
#pragma unroll
                for (int k = 0; k < K; ++k) e[k] = er[k];
            } else {
                M::exps(tau_u, rtau, tl, e);
            }
            {
#pragma clang fp contract(off)
                f0 = w * (M::value(x, e) - yl);
            }
        }
        if (GEO) {
#pragma unroll
            for (int k = 0; k < K; ++k) er[k] *= Rk[k];
        }
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                // model at x + h_i e_i, recomputed the way fun(x + h) evaluates it
                double xi[N], ei[K > 0 ? K : 1];
#pragma unroll
                for (int j = 0; j < N; ++j) xi[j] = x[j];
                xi[i] = x[i] + h[i];
#pragma unroll
                for (int k = 0; k < K; ++k) ei[k] = e[k];
                if (i >= K && i < 2 * K) {
                    if (GEO) {
                        ei[i - K] = erh[i - K];
                        erh[i - K] *= Rh[i - K];
                    } else {
                        ei[i - K] = exp_neg_quotient(-1.0 * tl, tauh_u[i - K], rtau_h[i - K]);
                    }
                }
                double fi;
                {
#pragma clang fp contract(off)
                    fi = w * (M::value(xi, ei) - yl);
                }
                Jr[i] = div_shared(fi - f0, dx[i], rdx[i]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Jr[k] = w * (e[k] - (M::kFreeS2 ? 0.0 : 1.0));
                Jr[K + k] = w * (x[k] * e[k] * tl / (x[K + k] * x[K + k]));
            }
            if (M::kFreeS2) Jr[N - 1] = w;
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            gacc[i] = fma(Jr[i], f0, gacc[i]);
#pragma unroll
            for (int j = 0; j <= i; ++j) Aacc[tri(i, j)] = fma(Jr[i], Jr[j], Aacc[tri(i, j)]);
        }
    };
    int l0 = tid;
    if (CACHE) {
#pragma unroll
        for (int j = 0; j < kCacheCap; ++j) {          // static slot indices: the cache stays in registers
            const int l = tid + j * NTH;
            if (l < L) point(l, j);
        }
        l0 = tid + kCacheCap * NTH;
    }
    for (int l = l0; l < L; l += NTH) point(l, -1);
    };
    if (T.geo) run(std::true_type{});
    else run(std::false_type{});
    // workgroup sums in the fixed order of block_sums (lanes by DPP butterfly, then waves 0..W-1): J^T J goes to the
    // shared matrix in LDS (element k by thread k), J^T f to every thread's registers
    {
        const int lane = tid & 63, wave = tid >> 6;
        double *r = T.red();
        if (SR_FIT_REDUCE_MANY && NT + N <= 64) {
            // all NT + N lane sums in ~220 instructions (sr_wave_sum_many_f64) instead of 25 each; the lane that ends up
            // with value k's total stores it
            const double tot = sr_wave_sum_many_f64<(NT + N <= 64 ? NT + N : 1)>(acc, lane);
            const int k = sr_reduced_index(lane);
            if (k < NT + N) r[wave * kRedStride + k] = tot;
        } else {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const double w = wsum(Aacc[k]);
                if (lane == 0) r[wave * kRedStride + k] = w;
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const double w = wsum(gacc[k]);
                if (lane == 0) r[wave * kRedStride + NT + k] = w;
            }
        }
        __syncthreads();
        for (int k = tid; k < NT; k += NTH) {
            double acc = r[k];
#pragma unroll
            for (int w = 1; w < R::NW; ++w) acc += r[w * kRedStride + k];
            A[k] = acc;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
            double acc = r[NT + k];
#pragma unroll
            for (int w = 1; w < R::NW; ++w) acc += r[w * kRedStride + NT + k];
            g[k] = acc;
        }
        __syncthreads();
    }
}

struct SolveParams {
    double tau_max, ftol, xtol, gtol;
    int max_nfev, jac_mode;
};

// Workgroup-uniform n-vectors of the solver in SGPRs (round 4).  x, the gradient g, the Coleman-Li scaling v / dv, d = sqrt(v) and
// g_h = d g are the same in every lane; the compiler cannot know that and keeps them in VGPRs -- 2 per double, ~110 registers at
// n = 9 that are live across the whole trust-region step and were its spill victims (42 scratch reloads per trial step, each a
// round trip in the leader wave's dependent chain).  uni() (v_readfirstlane) hands them over as scalar registers, which a float64
// VALU instruction takes as an operand: same values, same operations, bit-identical fits.  Measured on the cfg3 batch: scratch
// 808 -> 596 B per lane, 42 -> 19 scratch reloads per trial step, a lone launch 6.56 -> 6.27 ms (20.4 instead of 21.4 us per
// evaluation of the slowest residue), 0.710 -> 0.700 ms per batch with the chip full.  Going further (the scalars cost, Delta,
// alpha, theta and x at the initial point as well) runs out of SGPRs: 102 in use, the overflow goes to VGPR lanes and the frame
// grows again (644 B, 25 reloads).
// curve_fit(...) of conduct_curve_fitting for the staged residue T, starting from p0 (registers).
// Outputs: x (optimum), pc (packed covariance, valid when cov_ok), chi (calc_chiSq), status, nfev.
// DIAG: pc receives only the N diagonal elements of the covariance (what the model-order search needs for its dP > P
// test) -- computed by the same column solves as the full matrix, so the values are bit-identical; it just does not keep
// 36 doubles per lane that nobody reads (they were a third of search_order<9>'s scratch frame).
template <int N, bool DIAG, class R>
__device__ __forceinline__ void trf_solve(const R &T, const double *p0, const SolveParams &P, double *x, double *pc,
                                          bool &cov_ok, double &chi, int &status, int &nfev)
{
    constexpr int K = N / 2;
    constexpr int NT = N * (N + 1) / 2;
    const int m = T.L;
    const bool leader = !SR_FIT_LEADER || (R::NTH == 64) || (T.tid < 64);
    double lb[N], ub[N];
    bool inb = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        x[i] = p0[i];
        lb[i] = 0.0;
        ub[i] = (i >= K && i < 2 * K) ? P.tau_max : 1.0;
        inb = inb && (x[i] >= lb[i]) && (x[i] <= ub[i]);
    }
    status = -99;
    nfev = 0;
#ifdef SR_FIT_DEV_NJEV          // development: Jacobians evaluated (accepted steps + 1), returned in the high half of nfev
    int njev = 0;
#define SR_NJEV_INC() (++njev)
#else
#define SR_NJEV_INC()
#endif
    double cost = INFINITY;
    double *A = T.matA(), *B = T.matB();     // LDS, workgroup-uniform
    double g[N];
    bool have_fit = false;

    if (!inb) {
        status = -2;              // least_squares: "Initial guess is outside of provided bounds" (ValueError)
    } else {
        strictly_feasible<N>(x, lb, ub, 1e-10);
        bool finite;
        PointCache<N> pcache;
        cost = eval_f<N>(T, x, finite, pcache);
        nfev = 1;
        if (!finite) {
            status = -3;          // "Residuals are not finite in the initial point"
        } else {
            have_fit = true;
            eval_jac<N>(T, x, lb, ub, P.jac_mode, A, g, pcache);
            SR_NJEV_INC();
            const int max_nfev = P.max_nfev > 0 ? P.max_nfev : 100 * N;
            double v[N], dv[N];
            // CL_scaling_vector
#pragma unroll
            for (int i = 0; i < N; ++i) {
                v[i] = 1.0; dv[i] = 0.0;
                if (g[i] < 0) { v[i] = ub[i] - x[i]; dv[i] = -1.0; }
                if (g[i] > 0) { v[i] = x[i] - lb[i]; dv[i] = 1.0; }
            }
            double Delta;
            {
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) { const double q = x[i] / sqrt(v[i]); s = fma(q, q, s); }
                Delta = sqrt(s);
                if (Delta == 0) Delta = 1.0;
            }
            double alpha = 0.0;
            int term = 0;      // 0 = none
            bool term_set = false;
            while (true) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    g[i] = uni(g[i]);
                    v[i] = 1.0; dv[i] = 0.0;
                    if (g[i] < 0) { v[i] = ub[i] - x[i]; dv[i] = -1.0; }
                    if (g[i] > 0) { v[i] = x[i] - lb[i]; dv[i] = 1.0; }
                    v[i] = uni(v[i]);
                    dv[i] = uni(dv[i]);
                }
                double g_norm = 0.0;
#pragma unroll
                for (int i = 0; i < N; ++i) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
                if (g_norm < P.gtol) { term = 1; term_set = true; }
                if (term_set || nfev == max_nfev) break;

                // The trust-region sub-problem (Cholesky per Levenberg parameter, reflective / Cauchy step selection) is
                // a few thousand dependent float64 operations on workgroup-uniform data.  Everything long-lived in it
                // (J^T J, B, the Cholesky factor) lives in LDS, not in VGPRs; moving d and g_h there as well did not pay.
                // (SR_FIT_LEADER=1: only the leader wave solves it and broadcasts the trial point.)
                const double theta = fmax(0.995, 1 - g_norm);
                double d[N], g_h[N];
#pragma unroll
                for (int i = 0; i < N; ++i) { d[i] = uni(sqrt(v[i])); g_h[i] = uni(d[i] * g[i]); }
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        const double b = (A[tri(i, j)] * d[i]) * d[j] + (i == j ? g[i] * dv[i] : 0.0);
                        if (T.tid == 0) B[tri(i, j)] = b;
                    }
                __syncthreads();

                double actual_reduction = -1.0, cost_new = cost;
                double xn[N];
                while (actual_reduction <= 0 && nfev < max_nfev) {
                    double predicted, step_h_norm, step_norm;
                    if (leader) {
                        double p_h[N], p[N], step[N], step_h[N];
                        solve_tr<N>(B, g_h, m, Delta, alpha, p_h, T.matLf());
#pragma unroll
                        for (int i = 0; i < N; ++i) p[i] = d[i] * p_h[i];
                        select_step<N>(x, B, g_h, p, p_h, d, Delta, lb, ub, theta, step, step_h, predicted);
#pragma unroll
                        for (int i = 0; i < N; ++i) xn[i] = x[i] + step[i];
                        strictly_feasible<N>(xn, lb, ub, 0.0);
                        step_h_norm = normN<N>(step_h);
                        step_norm = normN<N>(step);
                        if (SR_FIT_LEADER && R::NTH > 64 && T.tid == 0) {
                            double *bc = T.bcast();
#pragma unroll
                            for (int i = 0; i < N; ++i) bc[i] = xn[i];
                            bc[N] = predicted; bc[N + 1] = step_h_norm; bc[N + 2] = step_norm;
                        }
                    }
                    if (SR_FIT_LEADER && R::NTH > 64) {
                        __syncthreads();
                        const double *bc = T.bcast();
#pragma unroll
                        for (int i = 0; i < N; ++i) xn[i] = bc[i];
                        predicted = bc[N]; step_h_norm = bc[N + 1]; step_norm = bc[N + 2];
                    }
                    bool finite2;
                    cost_new = eval_f<N>(T, xn, finite2, pcache);
                    nfev += 1;
                    if (!finite2) {
                        Delta = 0.25 * step_h_norm;
                        continue;
                    }
                    actual_reduction = cost - cost_new;
                    // update_tr_radius
                    double ratio;
                    if (predicted > 0) ratio = actual_reduction / predicted;
                    else if (predicted == 0 && actual_reduction == 0) ratio = 1;
                    else ratio = 0;
                    double Delta_new = Delta;
                    if (ratio < 0.25) Delta_new = 0.25 * step_h_norm;
                    else if (ratio > 0.75 && step_h_norm > 0.95 * Delta) Delta_new = Delta * 2.0;
                    // check_termination
                    const bool ft = (actual_reduction < P.ftol * cost) && (ratio > 0.25);
                    const bool xt = step_norm < P.xtol * (P.xtol + normN<N>(x));
                    if (ft && xt) { term = 4; term_set = true; }
                    else if (ft) { term = 2; term_set = true; }
                    else if (xt) { term = 3; term_set = true; }
                    if (term_set) break;
                    alpha *= Delta / Delta_new;
                    Delta = Delta_new;
                }
                if (actual_reduction > 0) {
#pragma unroll
                    for (int i = 0; i < N; ++i) x[i] = uni(xn[i]);
                    cost = cost_new;
                    eval_jac<N>(T, x, lb, ub, P.jac_mode, A, g, pcache);
                    SR_NJEV_INC();
                }
            }
            status = term_set ? term : 0;
        }
    }
#ifdef SR_FIT_DEV_NJEV
    nfev |= njev << 16;
#endif
#undef SR_NJEV_INC

    // ---- outputs: popt, pcov = (J^T J)^-1 * 2 cost / (m - n)  (curve_fit, _minpack_py.py:1040-1055), chi ----
    cov_ok = false;
    if (have_fit && m > N) {
        double Lf_regs[SR_FIT_LF_LDS ? 1 : NT], inv[N], lmin2;
        double *Lf = SR_FIT_LF_LDS ? T.matLf() : Lf_regs;
        cov_ok = cholN<N>(A, 0.0, Lf, inv, lmin2);
        if (cov_ok) {
            // conditioning guard equivalent to scipy's singular-value cut eps*max(m,n)*s_max
            double dmax = 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) dmax = fmax(dmax, A[tri(i, i)]);
            const double thr = kEPS * (double)m;
            if (!(lmin2 > thr * thr * dmax)) cov_ok = false;
        }
        if (cov_ok) {
            const double s_sq = 2.0 * cost / (double)(m - N);
#pragma unroll
            for (int c = 0; c < N; ++c) {
                double e[N], z[N], col[N];
#pragma unroll
                for (int i = 0; i < N; ++i) e[i] = (i == c) ? 1.0 : 0.0;
                fwdN<N>(Lf, inv, e, z);
                bwdN<N>(Lf, inv, z, col);
                if (DIAG) {
                    pc[c] = col[c] * s_sq;
                } else {
#pragma unroll
                    for (int i = c; i < N; ++i) pc[tri(i, c)] = col[i] * s_sq;
                }
            }
        }
    }
    chi = INFINITY;
    if (have_fit) {
        // calc_chiSq, fitting_Ct_functions.py:272-276: mean((model - y)^2 / sigma)
        double acc[1] = {0.0};
        for (int l = T.tid; l < T.L; l += R::NTH) {
            double e[K > 0 ? K : 1];
            Model<N>::exps(x, T.ld_t(l), e);
            const double r = Model<N>::value(x, e) - T.ld_y(l);
            acc[0] += T.sg ? (r * r) / T.sg[l] : r * r;
        }
        T.template block_sums<1>(acc);
        chi = acc[0] / (double)T.L;
    }
}

template <int N, int W, bool LDS>
__global__ __launch_bounds__(W * 64) void k_trf(FitArgs a)
{
    constexpr int NT = N * (N + 1) / 2;
    const int res = blockIdx.x;
    if (a.skip && a.skip[res]) return;
    const int tid = threadIdx.x;
    Residue<W, LDS> T;
    T.L = a.L;
    T.tid = tid;
    const double *sg_res = a.sigma ? a.sigma + (int64_t)res * a.L : nullptr;
    T.stage(a.t + (int64_t)res * a.L, a.y + (int64_t)res * a.L, sg_res, LDS ? nullptr : a.fws + (int64_t)res * a.L, a.geo);
    double p0[N], x[N], pc[NT], chi;
    bool cov_ok;
    int status, nfev;
#pragma unroll
    for (int i = 0; i < N; ++i) p0[i] = a.p0[(int64_t)res * N + i];
    SolveParams P;
    P.tau_max = a.tau_max; P.ftol = a.ftol; P.xtol = a.xtol; P.gtol = a.gtol; P.max_nfev = a.max_nfev; P.jac_mode = a.jac_mode;
    trf_solve<N, false>(T, p0, P, x, pc, cov_ok, chi, status, nfev);
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) a.popt[(int64_t)res * N + i] = x[i];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                a.pcov[((int64_t)res * N + i) * N + j] = cov_ok ? pc[tri(i, j)] : INFINITY;
        a.chisq[res] = chi;
        a.status[res] = status;
        a.nfev[res] = nfev;
    }
}

// ---- model-order search on the device -----------------------------------------------------------------
// optimised_curve_fitting (fitting_Ct_functions.py:278-304) with conduct_curve_fitting(bReInitialise=True)
// (:306-345) and initialise_for_fit_advanced (:359-374) for ONE residue per workgroup, every model order in one
// launch: initial guess -> bounded fit -> quality flags -> accept / reject -> next order.  The host used to sit in
// this loop (one launch, six copies and ~0.6 ms of numpy per order); the only thing it still supplies is the
// log-spaced tau guesses (numpy.logspace involves pow(), whose last bit the device cannot promise to reproduce).
constexpr int kMaxOrders = 8;

struct SearchArgs {
    const double *t, *y, *sigma;      // (nRes, L); sigma may be null
    int nRes, L, nOrders;
    int orders[kMaxOrders];           // numbers of parameters, in the order they are tried
    int tau_off[kMaxOrders];          // offset of each order's tau guesses inside a tau_guess row
    const double *tau_guess;          // (1 or nRes, sum of orders/2)
    int64_t tau_stride;               // 0: one row shared by every residue
    double tau_max, chi_thr, ftol, xtol, gtol;
    int Pmax, Kmax;
    double *popt, *dP, *chisq;        // (nOrders, nRes, Pmax), same, (nOrders, nRes)
    int *status, *nfev;               // (nOrders, nRes); status -100 = order not attempted
    int *best;                        // (nRes) index into orders of the selected model, -1 = none satisfactory
    double *sel_S2, *sel_C, *sel_tau, *sel_chi;   // (nRes), (nRes,Kmax), (nRes,Kmax), (nRes): components sorted by tau
    int *sel_K;                       // (nRes) number of components of the selected model (0 = none)
    double *fws;                      // (nRes, L) weights when a residue does not fit into LDS
    int64_t t_stride;                 // L, or 0: one time axis shared by every residue
    const int *order;                 // null, or nRes residue indices: workgroup b solves residue order[b] (results stay at the
                                      // residue's own index; only WHEN a residue starts changes)
    unsigned int *tail_signal;        // null, or signal memory: the LAST workgroup of the grid stores tail_value there when it
    unsigned int tail_value;          // starts -- workgroups are dispatched in index order, so from then on the launch only drains
    int geo;                          // option fit_geo, as in FitArgs
};

// numpy.mean of n <= 128 contiguous float64 values (pairwise summation of numpy/_core/src/umath/loops_utils.h.src)
template <class R>
__device__ __forceinline__ double np_mean_y(const R &T, int start, int n)
{
#pragma clang fp contract(off)
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) res += T.ld_y(start + i);
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = T.ld_y(start + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += T.ld_y(start + i + j);
        }
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += T.ld_y(start + i);
    }
    return res / (double)n;
}

struct SearchState {
    bool first, done;
    int best;
    double best_chi;
    double xbest[kNmax];
};

// Not inlined on purpose: one register allocation per model order (inlining the eight solvers into the kernel body
// made the allocator spill 1.1 KB per lane); the residue descriptor travels by value in registers, the small
// search state lives on the stack and is only touched before and after a solve.
template <int N, class R>
__device__ __noinline__ void search_order(const R T, const SearchArgs &a, int j, int res, double avgBeg, double avgEnd,
                                          SearchState &st)
{
    constexpr int K = N / 2;
    constexpr bool kFree = (N % 2) == 1;
    // the state as the previous order left it, read BEFORE the solve: st lives in LDS and every wave updates it at the end of
    // an order, so a wave that reaches the accept / reject step late must not see what an earlier wave has already written
    // for THIS order (the solve's barriers keep the waves less than one order apart)
    const bool st_first = st.first;
    const double st_best_chi = st.best_chi;
    double p0[N], c0, sumC, S2_0;
    {
#pragma clang fp contract(off)
        // initialise_for_fit_advanced, fitting_Ct_functions.py:359-374
        c0 = fabs(avgBeg - avgEnd) / (double)K;
        sumC = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) sumC += c0;
        S2_0 = kFree ? avgEnd : 1.0 - sumC / (double)K;
    }
    const double *tg = a.tau_guess + a.tau_stride * res + a.tau_off[j];
#pragma unroll
    for (int k = 0; k < K; ++k) { p0[k] = c0; p0[K + k] = tg[k]; }
    if (kFree) p0[N - 1] = S2_0;

    SolveParams P;
    P.tau_max = a.tau_max; P.ftol = a.ftol; P.xtol = a.xtol; P.gtol = a.gtol; P.max_nfev = 100 * N; P.jac_mode = 0;
    double x[N], pc[N], chi;
    bool cov_ok;
    int status, nfev;
    trf_solve<N, true>(T, p0, P, x, pc, cov_ok, chi, status, nfev);

    // quality flags, fitting_Ct_functions.py:320-338 (the sum > 1 test runs on the INITIAL guess: reference quirk)
    const bool ok = status > 0;
    bool q1 = true;
    double dP[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        dP[i] = cov_ok ? sqrt(pc[i]) : INFINITY;
        if (dP[i] > x[i]) q1 = false;
    }
    bool q2;
    {
#pragma clang fp contract(off)
        const double S2chk = kFree ? S2_0 : 1.0 - sumC;
        q2 = !(S2chk + sumC > 1.0);
    }
    const bool allq = ok && q1 && q2;
    const double chiSq = ok ? chi : INFINITY;
    if (T.tid == 0) {
        const int64_t o = (int64_t)j * a.nRes + res;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            a.popt[o * a.Pmax + i] = x[i];
            a.dP[o * a.Pmax + i] = dP[i];
        }
        a.chisq[o] = chiSq;
        a.status[o] = status;
        a.nfev[o] = nfev;
    }
    // accept / reject, optimised_curve_fitting :278-304
    bool take = false;
    if (st_first) {
        if (allq) { take = true; st.first = false; }
    } else {
        if (!allq || chiSq >= st_best_chi * a.chi_thr) st.done = true;
        else take = true;
    }
    if (take) {
        st.best = j;
        st.best_chi = chiSq;
#pragma unroll
        for (int i = 0; i < N; ++i) st.xbest[i] = x[i];
    }
}

#ifndef SR_FIT_WAVES_EU
#define SR_FIT_WAVES_EU 2     // wavefronts per SIMD the register budget allows: 2 -> 256 VGPRs, 3 -> 168
#endif
// SR_FIT_WAVES_EU_LOW: register budget of the variant that only knows orders <= 5.  Orders 2, 3 and 5 are a third of the search's
// cost with the chip full (0.128 + 0.105 + 0.043 of 0.72 ms per batch, scripts/dev/fit_by_order.py) although their fits are
// tiny, so more resident workgroups were tried for them: 3 / 4 waves per SIMD (168 / 128 VGPRs: 592 / 840 B of scratch) with the
// residue in LDS or in global memory -- 0.289 / 0.337 ms (LDS), 0.316 / 0.360 (global) against 0.276 at two waves: they are
// issue-bound like the high orders, not waiting for latency.
#ifndef SR_FIT_WAVES_EU_LOW
#define SR_FIT_WAVES_EU_LOW SR_FIT_WAVES_EU
#endif
template <int NMAX, int W, bool LDS>
__global__ __launch_bounds__(W * 64, NMAX <= 5 ? SR_FIT_WAVES_EU_LOW : SR_FIT_WAVES_EU) void k_order_search(SearchArgs a)
{
    const int res = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x;
    if (a.tail_signal && blockIdx.x == gridDim.x - 1 && tid == 0)
        __hip_atomic_store(a.tail_signal, a.tail_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // The fits are the latency chain of a group: their waves issue before the waves of the bandwidth kernels that fill the
    // launch's tail (the group's histograms: 24.0 -> 23.1 ms for the merged launch of 20 batches with them beside it).
    __builtin_amdgcn_s_setprio(SR_FIT_SETPRIO);
    Residue<W, LDS> T;
    T.L = a.L;
    T.tid = tid;
    const double *sg_res = a.sigma ? a.sigma + (int64_t)res * a.L : nullptr;
    T.stage(a.t + (int64_t)res * a.t_stride, a.y + (int64_t)res * a.L, sg_res, LDS ? nullptr : a.fws + (int64_t)res * a.L, a.geo);

    const int ns = a.L < 10 ? a.L : 10;                 // nSample = 10, fitting_Ct_functions.py:359
    const double avgBeg = np_mean_y(T, 0, ns);
    const double avgEnd = np_mean_y(T, a.L - ns, ns);

    if (tid == 0) {
        for (int j = 0; j < a.nOrders; ++j) {
            const int64_t o = (int64_t)j * a.nRes + res;
            for (int i = 0; i < a.Pmax; ++i) { a.popt[o * a.Pmax + i] = NAN; a.dP[o * a.Pmax + i] = NAN; }
            a.chisq[o] = INFINITY;
            a.status[o] = -100;
            a.nfev[o] = 0;
        }
    }
    // The search state lives in LDS, not on the stack: it crosses the (non-inlined) per-order solver calls by reference, and a
    // stack object costs every lane a scratch frame.  It is workgroup-uniform: every thread stores the same values and reads
    // them back behind its own stores (LDS operations of a wave complete in order), so no barrier is needed.
    static_assert(sizeof(SearchState) <= kStateDoubles * sizeof(double), "SearchState does not fit its LDS slot");
    SearchState &st = *reinterpret_cast<SearchState *>(fit_smem + Residue<W, LDS>::ST);
    st.first = true; st.done = false; st.best = -1; st.best_chi = INFINITY;
#pragma unroll
    for (int i = 0; i < kNmax; ++i) st.xbest[i] = 0.0;

    for (int j = 0; j < a.nOrders && !st.done; ++j) {
        switch (a.orders[j]) {
            case 2: search_order<2>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 3: search_order<3>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 4: search_order<4>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 5: search_order<5>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 6: if (NMAX >= 6) search_order<(NMAX >= 6 ? 6 : 2)>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 7: if (NMAX >= 7) search_order<(NMAX >= 7 ? 7 : 2)>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 8: if (NMAX >= 8) search_order<(NMAX >= 8 ? 8 : 2)>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 9: if (NMAX >= 9) search_order<(NMAX >= 9 ? 9 : 2)>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 10: if (NMAX >= 10) search_order<(NMAX >= 10 ? 10 : 2)>(T, a, j, res, avgBeg, avgEnd, st); break;
            case 11: if (NMAX >= 11) search_order<(NMAX >= 11 ? 11 : 2)>(T, a, j, res, avgBeg, avgEnd, st); break;
            default: break;
        }
    }

    // selected model with its components sorted by tau (sort_components, fitting_Ct_functions.py:203-209)
    if (tid == 0) {
        a.best[res] = st.best;
        double Cs[kNmax / 2], ts[kNmax / 2];
        int K = 0;
        double S2 = 0.0, chi = NAN;
        if (st.best >= 0) {
#pragma clang fp contract(off)
            const int nP = a.orders[st.best];
            K = nP / 2;
            double sum = 0.0;
            for (int k = 0; k < kNmax / 2; ++k) {
                // st.xbest is indexed with compile-time constants only (registers): unrolled selects
                double c = 0.0, tt = 0.0;
#pragma unroll
                for (int i = 0; i < kNmax; ++i) {
                    if (i == k) c = st.xbest[i];
                    if (i == K + k) tt = st.xbest[i];
                }
                if (k < K) { Cs[k] = c; ts[k] = tt; sum += c; }
            }
            double last = 0.0;
#pragma unroll
            for (int i = 0; i < kNmax; ++i)
                if (i == nP - 1) last = st.xbest[i];
            S2 = (nP & 1) ? last : 1.0 - sum;
            chi = st.best_chi;
            for (int i = 1; i < K; ++i) {               // stable insertion sort by tau
                const double tc = ts[i], cc = Cs[i];
                int q = i - 1;
                while (q >= 0 && ts[q] > tc) { ts[q + 1] = ts[q]; Cs[q + 1] = Cs[q]; --q; }
                ts[q + 1] = tc; Cs[q + 1] = cc;
            }
        }
        a.sel_S2[res] = S2;
        a.sel_chi[res] = chi;
        a.sel_K[res] = K;
        for (int k = 0; k < a.Kmax; ++k) {
            a.sel_C[(int64_t)res * a.Kmax + k] = k < K ? Cs[k] : 0.0;
            a.sel_tau[(int64_t)res * a.Kmax + k] = k < K ? ts[k] : 1.0;
        }
    }
}

template <int NMAX, int W>
int launch_search_w(sr_ctx *ctx, const SearchArgs &a)
{
    const size_t lds_small = fit_lds_doubles(W, 0) * sizeof(double);
    const size_t lds_full = fit_lds_doubles(W, a.L) * sizeof(double);
    if (ctx->fit_lds && lds_full <= sr_lds_limit(ctx)) {
        if (lds_full > 64 * 1024)
            SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_order_search<NMAX, W, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full));
        hipLaunchKernelGGL((k_order_search<NMAX, W, true>), dim3((unsigned)a.nRes), dim3(W * 64), lds_full, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((k_order_search<NMAX, W, false>), dim3((unsigned)a.nRes), dim3(W * 64), lds_small, ctx->stream, a);
    }
    SR_HIP(hipGetLastError());
    return 0;
}

template <int NMAX>
int launch_search(sr_ctx *ctx, const SearchArgs &a)
{
    switch (ctx->fit_waves) {
        case 1: return launch_search_w<NMAX, 1>(ctx, a);
        case 2: return launch_search_w<NMAX, 2>(ctx, a);
        default: return launch_search_w<NMAX, 4>(ctx, a);
    }
}

// residual + analytic Jacobian for arbitrary parameter sets
__global__ __launch_bounds__(256) void k_resjac(const double *__restrict__ t, const double *__restrict__ y,
                                                const double *__restrict__ sigma, const double *__restrict__ params,
                                                int L, int P, double *__restrict__ resid, double *__restrict__ jac)
{
#pragma clang fp contract(off)
    const int res = blockIdx.x;
    const int K = P / 2;
    const bool freeS2 = (P % 2) == 1;
    const double *x = params + (int64_t)res * P;
    double S2;
    if (freeS2) S2 = x[P - 1];
    else {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += x[k];
        S2 = 1.0 - s;
    }
    for (int l = threadIdx.x; l < L; l += 256) {
        const int64_t o = (int64_t)res * L + l;
        const double tl = t[o], w = sigma ? 1.0 / sigma[o] : 1.0;
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double e = exp((-1.0 * tl) / x[K + k]);
            s += x[k] * e;
            if (jac) {
                jac[o * P + k] = w * (e - (freeS2 ? 0.0 : 1.0));
                jac[o * P + K + k] = w * (x[k] * e * tl / (x[K + k] * x[K + k]));
            }
        }
        if (jac && freeS2) jac[o * P + P - 1] = w;
        resid[o] = w * ((S2 + s) - y[o]);
    }
}

template <int N, int W>
int launch_trf_w(sr_ctx *ctx, const FitArgs &a)
{
    const size_t lds_small = fit_lds_doubles(W, 0) * sizeof(double);
    const size_t lds_full = fit_lds_doubles(W, a.L) * sizeof(double);
    if (lds_full <= sr_lds_limit(ctx)) {
        if (lds_full > 64 * 1024)
            SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trf<N, W, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_full));
        hipLaunchKernelGGL((k_trf<N, W, true>), dim3((unsigned)a.nRes), dim3(W * 64), lds_full, ctx->stream, a);
    } else {
        hipLaunchKernelGGL((k_trf<N, W, false>), dim3((unsigned)a.nRes), dim3(W * 64), lds_small, ctx->stream, a);
    }
    SR_HIP(hipGetLastError());
    return 0;
}

// The single-order solver uses the same number of waves per residue as the model-order search (sr_set_option
// "fit_waves"): the workgroup sums are combined wave by wave, so only equal wave counts give bit-identical fits --
// which is what lets the host-driven search (one sr_expfit_lm_f64 call per order) reproduce the one-launch search exactly.
template <int N>
int launch_trf(sr_ctx *ctx, const FitArgs &a)
{
    switch (ctx->fit_waves) {
        case 1: return launch_trf_w<N, 1>(ctx, a);
        case 2: return launch_trf_w<N, 2>(ctx, a);
        default: return launch_trf_w<N, 4>(ctx, a);
    }
}

int dispatch_trf(sr_ctx *ctx, int P, const FitArgs &a)
{
    switch (P) {
        case 2: return launch_trf<2>(ctx, a);
        case 3: return launch_trf<3>(ctx, a);
        case 4: return launch_trf<4>(ctx, a);
        case 5: return launch_trf<5>(ctx, a);
        case 6: return launch_trf<6>(ctx, a);
        case 7: return launch_trf<7>(ctx, a);
        case 8: return launch_trf<8>(ctx, a);
        case 9: return launch_trf<9>(ctx, a);
        case 10: return launch_trf<10>(ctx, a);
        case 11: return launch_trf<11>(ctx, a);
        default:
            sr_set_error("sr_expfit_lm_f64: P=%d parameters not supported (2..%d)", P, kNmax);
            return -3;
    }
}

}  // namespace

extern "C" {

int sr_expfit_lm_f64_dev(sr_ctx *ctx, const double *t, const double *C, const double *sigma, int nRes, int L, int P,
                         const double *p0, double tau_max, int max_iter, const unsigned char *skip, double *work,
                         double *popt, double *pcov, double *chisq, int *status, int *n_iter)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && p0 && popt && pcov && chisq && status && n_iter, -2, "sr_expfit_lm_f64_dev: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && P >= 2 && P <= kNmax, -3, "sr_expfit_lm_f64_dev: bad sizes nRes=%d L=%d P=%d", nRes, L, P);
    double *fws = work ? work : (double *)sr_workspace(ctx, SR_WS_FIT, (size_t)nRes * 2 * L * sizeof(double));
    if (!fws) return -5;
    FitArgs a;
    a.t = t; a.y = C; a.sigma = sigma; a.p0 = p0; a.skip = skip; a.tau_max = tau_max;
    a.nRes = nRes; a.L = L; a.max_nfev = max_iter; a.jac_mode = 0;
    a.ftol = a.xtol = a.gtol = 1e-8;
    a.popt = popt; a.pcov = pcov; a.chisq = chisq; a.status = status; a.nfev = n_iter; a.fws = fws; a.geo = ctx->fit_geo;
    if (max_iter < 0) { a.max_nfev = -max_iter; a.jac_mode = 1; }     // negative: analytic Jacobian variant
    return dispatch_trf(ctx, P, a);
}

int sr_expfit_order_search_f64_dev(sr_ctx *ctx, const double *t, const double *C, const double *sigma, int nRes, int L,
                                   const int *orders, int nOrders, const double *tau_guess, int tau_guess_rows,
                                   double tau_max, double chi_threshold, double *work, double *popt, double *dP,
                                   double *chisq, int *status, int *nfev, int *best, double *sel_S2, double *sel_C,
                                   double *sel_tau, double *sel_chi, int *sel_K)
{
    return sr_expfit_order_search_batched_f64_dev(ctx, t, nRes, C, sigma, nRes, L, orders, nOrders, tau_guess, tau_guess_rows, tau_max,
                                                  chi_threshold, nullptr, nullptr, 0u, work, popt, dP, chisq, status, nfev, best,
                                                  sel_S2, sel_C, sel_tau, sel_chi, sel_K);
}

int sr_expfit_order_search_batched_f64_dev(sr_ctx *ctx, const double *t, int t_rows, const double *C, const double *sigma, int nRes,
                                           int L, const int *orders, int nOrders, const double *tau_guess, int tau_guess_rows,
                                           double tau_max, double chi_threshold, const int *dispatch_order,
                                           uint32_t *tail_signal, uint32_t tail_value, double *work,
                                           double *popt, double *dP, double *chisq, int *status, int *nfev, int *best,
                                           double *sel_S2, double *sel_C, double *sel_tau, double *sel_chi, int *sel_K)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t_rows == 1 || t_rows == nRes, -3, "sr_expfit_order_search_batched_f64_dev: t_rows must be 1 or nRes");
    SR_REQUIRE(t && C && orders && tau_guess && popt && dP && chisq && status && nfev && best && sel_S2 && sel_C && sel_tau &&
                   sel_chi && sel_K, -2, "sr_expfit_order_search_f64_dev: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && nOrders >= 1 && nOrders <= kMaxOrders, -3,
               "sr_expfit_order_search_f64_dev: bad sizes nRes=%d L=%d nOrders=%d (at most %d orders)", nRes, L, nOrders, kMaxOrders);
    SR_REQUIRE(tau_guess_rows == 1 || tau_guess_rows == nRes, -3, "sr_expfit_order_search_f64_dev: tau_guess_rows must be 1 or nRes");
    SearchArgs a;
    int pmax = 0, off = 0;
    for (int j = 0; j < nOrders; ++j) {
        SR_REQUIRE(orders[j] >= 2 && orders[j] <= kNmax, -3, "sr_expfit_order_search_f64_dev: order %d not supported (2..%d)",
                   orders[j], kNmax);
        a.orders[j] = orders[j];
        a.tau_off[j] = off;
        off += orders[j] / 2;
        pmax = orders[j] > pmax ? orders[j] : pmax;
    }
    for (int j = nOrders; j < kMaxOrders; ++j) { a.orders[j] = 0; a.tau_off[j] = 0; }
    const size_t lds_full = fit_lds_doubles(ctx->fit_waves, L) * sizeof(double);
    double *fws = work;
    if (!fws && (!ctx->fit_lds || lds_full > sr_lds_limit(ctx))) {
        fws = (double *)sr_workspace(ctx, SR_WS_FIT, (size_t)nRes * L * sizeof(double));
        if (!fws) return -5;
    }
    a.t = t; a.y = C; a.sigma = sigma; a.nRes = nRes; a.L = L; a.nOrders = nOrders;
    a.t_stride = t_rows == 1 ? 0 : L; a.order = dispatch_order;
    a.tail_signal = tail_signal; a.tail_value = tail_value; a.geo = ctx->fit_geo;
    a.tau_guess = tau_guess; a.tau_stride = tau_guess_rows == 1 ? 0 : off;
    a.tau_max = tau_max; a.chi_thr = chi_threshold; a.ftol = a.xtol = a.gtol = 1e-8;
    a.Pmax = pmax; a.Kmax = pmax / 2;
    a.popt = popt; a.dP = dP; a.chisq = chisq; a.status = status; a.nfev = nfev; a.best = best;
    a.sel_S2 = sel_S2; a.sel_C = sel_C; a.sel_tau = sel_tau; a.sel_chi = sel_chi; a.sel_K = sel_K; a.fws = fws;
#ifdef SR_FIT_DEV_FAST
    SR_REQUIRE(pmax <= 9, -3, "development build: orders up to 9 only");
    return launch_search<9>(ctx, a);
#else
    if (pmax <= 5) return launch_search<5>(ctx, a);
    if (pmax <= 9) return launch_search<9>(ctx, a);
    return launch_search<11>(ctx, a);
#endif
}

int sr_expfit_order_search_f64(sr_ctx *ctx, const double *t, const double *C, const double *sigma, int nRes, int L,
                               const int *orders, int nOrders, const double *tau_guess, int tau_guess_rows, double tau_max,
                               double chi_threshold, double *popt, double *dP, double *chisq, int *status, int *nfev,
                               int *best, double *sel_S2, double *sel_C, double *sel_tau, double *sel_chi, int *sel_K)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && orders && tau_guess && popt && dP && chisq && status && nfev && best && sel_S2 && sel_C && sel_tau &&
                   sel_chi && sel_K, -2, "sr_expfit_order_search_f64: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && nOrders >= 1 && nOrders <= kMaxOrders, -3, "sr_expfit_order_search_f64: bad sizes");
    SR_REQUIRE(tau_guess_rows == 1 || tau_guess_rows == nRes, -3, "sr_expfit_order_search_f64: tau_guess_rows must be 1 or nRes");
    int pmax = 0, sumK = 0;
    for (int j = 0; j < nOrders; ++j) {
        SR_REQUIRE(orders[j] >= 2 && orders[j] <= kNmax, -3, "sr_expfit_order_search_f64: order %d not supported", orders[j]);
        pmax = orders[j] > pmax ? orders[j] : pmax;
        sumK += orders[j] / 2;
    }
    const size_t nL = (size_t)nRes * L, nR = (size_t)nRes, nO = (size_t)nOrders, kmax = (size_t)(pmax / 2);
    const size_t ntau = (size_t)tau_guess_rows * sumK;
    double *in = (double *)sr_workspace(ctx, SR_WS_IN0, (3 * nL + ntau) * sizeof(double));
    const size_t nd = 2 * nO * nR * pmax + nO * nR + nR * (2 + 2 * kmax);
    double *out = (double *)sr_workspace(ctx, SR_WS_OUT0, nd * sizeof(double));
    int *iout = (int *)sr_workspace(ctx, SR_WS_OUT1, (2 * nO * nR + 2 * nR) * sizeof(int));
    if (!in || !out || !iout) return -5;
    double *t_d = in, *y_d = in + nL, *s_d = in + 2 * nL, *tg_d = in + 3 * nL;
    SR_HIP(hipMemcpyAsync(t_d, t, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(y_d, C, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (sigma) SR_HIP(hipMemcpyAsync(s_d, sigma, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(tg_d, tau_guess, ntau * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    double *popt_d = out, *dP_d = popt_d + nO * nR * pmax, *chi_d = dP_d + nO * nR * pmax, *sS2_d = chi_d + nO * nR;
    double *sC_d = sS2_d + nR, *st_d = sC_d + nR * kmax, *schi_d = st_d + nR * kmax;
    int *status_d = iout, *nfev_d = iout + nO * nR, *best_d = nfev_d + nO * nR, *sK_d = best_d + nR;
    int rc = sr_expfit_order_search_f64_dev(ctx, t_d, y_d, sigma ? s_d : nullptr, nRes, L, orders, nOrders, tg_d, tau_guess_rows,
                                            tau_max, chi_threshold, nullptr, popt_d, dP_d, chi_d, status_d, nfev_d, best_d,
                                            sS2_d, sC_d, st_d, schi_d, sK_d);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(popt, popt_d, nO * nR * pmax * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(dP, dP_d, nO * nR * pmax * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(chisq, chi_d, nO * nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(sel_S2, sS2_d, nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(sel_C, sC_d, nR * kmax * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(sel_tau, st_d, nR * kmax * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(sel_chi, schi_d, nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(status, status_d, nO * nR * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(nfev, nfev_d, nO * nR * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(best, best_d, nR * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(sel_K, sK_d, nR * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_expfit_lm_f64(sr_ctx *ctx, const double *t, const double *C, const double *sigma, int nRes, int L, int P,
                     const double *p0, double tau_max, int max_iter, double *popt, double *pcov, double *chisq,
                     int *status, int *n_iter)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && p0 && popt && pcov && chisq && status && n_iter, -2, "sr_expfit_lm_f64: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && P >= 2 && P <= kNmax, -3, "sr_expfit_lm_f64: bad sizes nRes=%d L=%d P=%d", nRes, L, P);
    const size_t nL = (size_t)nRes * L, nP = (size_t)nRes * P;
    double *in = (double *)sr_workspace(ctx, SR_WS_IN0, (3 * nL + nP) * sizeof(double));
    double *out = (double *)sr_workspace(ctx, SR_WS_OUT0, (nP + nP * P + nRes) * sizeof(double));
    int *iout = (int *)sr_workspace(ctx, SR_WS_OUT1, (size_t)nRes * 2 * sizeof(int));
    if (!in || !out || !iout) return -5;
    double *t_d = in, *y_d = in + nL, *s_d = in + 2 * nL, *p0_d = in + 3 * nL;
    SR_HIP(hipMemcpyAsync(t_d, t, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(y_d, C, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (sigma) SR_HIP(hipMemcpyAsync(s_d, sigma, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(p0_d, p0, nP * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    double *popt_d = out, *pcov_d = out + nP, *chi_d = pcov_d + nP * P;
    int rc = sr_expfit_lm_f64_dev(ctx, t_d, y_d, sigma ? s_d : nullptr, nRes, L, P, p0_d, tau_max, max_iter, nullptr, nullptr, popt_d, pcov_d,
                                  chi_d, iout, iout + nRes);
    if (rc) return rc;
    SR_HIP(hipMemcpyAsync(popt, popt_d, nP * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(pcov, pcov_d, nP * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(chisq, chi_d, (size_t)nRes * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(status, iout, (size_t)nRes * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(n_iter, iout + nRes, (size_t)nRes * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_expfit_resjac_f64(sr_ctx *ctx, const double *t, const double *C, const double *sigma, const double *params,
                         int nRes, int L, int P, double *resid, double *jac)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(t && C && params && resid, -2, "sr_expfit_resjac_f64: null pointer");
    SR_REQUIRE(nRes >= 1 && L >= 1 && P >= 1 && P <= 64, -3, "sr_expfit_resjac_f64: bad sizes");
    const size_t nL = (size_t)nRes * L, nP = (size_t)nRes * P;
    double *in = (double *)sr_workspace(ctx, SR_WS_IN0, (3 * nL + nP) * sizeof(double));
    double *r_d = (double *)sr_workspace(ctx, SR_WS_OUT0, nL * sizeof(double));
    double *j_d = jac ? (double *)sr_workspace(ctx, SR_WS_OUT1, nL * P * sizeof(double)) : nullptr;
    if (!in || !r_d || (jac && !j_d)) return -5;
    double *t_d = in, *y_d = in + nL, *s_d = in + 2 * nL, *p_d = in + 3 * nL;
    SR_HIP(hipMemcpyAsync(t_d, t, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(y_d, C, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (sigma) SR_HIP(hipMemcpyAsync(s_d, sigma, nL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SR_HIP(hipMemcpyAsync(p_d, params, nP * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_resjac, dim3((unsigned)nRes), dim3(256), 0, ctx->stream, t_d, y_d, sigma ? s_d : nullptr, p_d, L, P,
                       r_d, j_d);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(resid, r_d, nL * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (jac) SR_HIP(hipMemcpyAsync(jac, j_d, nL * P * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
