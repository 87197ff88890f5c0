// sr_relax.hip -- kernel 3a: batched spectral densities J(omega) and R1 / R2 / NOE / rho.
//
// Reference semantics:
//   Jomega ufunc  x/(x*x+y*y)                      Jomega/Jomega.c:49-66
//   _do_Jsum                                        spectral_densities.py:1961-1972
//   D_/A_coefficients_symmtop                       spectral_densities.py:1874-1906
//   J_direct_transform / J_combine_isotropic_exp_decayN / J_combine_symmtop_exp_decayN
//                                                   spectral_densities.py:2024-2077
//   get_relax_from_J(_simd), get_rho_from_J(_simd)  spectral_densities.py:1680-1786   (noe_mode 0)
//   spinRelaxationR1/R2/NOE.eval                    spectral_densities.py:820-907     (noe_mode 1)
//   weighted_average_stdev                          general_maths.py:100-110
//   _obtain_R1R2NOErho, _obtain_Jomega              calculate-relaxations-from-Ct.py:82-191
//
// One workgroup per (residue, experiment).  For the symmetric top the bin-centre vectors are the same
// for every residue, and J(bin, w) = sum_j A_j(bin) * G[j][w] with the 3x5 table
// G[j][w] = S2*Jomega(D_j, w) + sum_k C_k*Jomega(D_j + 1/tau_k, w) independent of the bin, so a bin
// costs 15 FMAs.  Weighted mean and weighted sigma over the bins use two passes (mean first), like
// numpy.average followed by average((x-mean)^2).
#include "sr_internal.h"

namespace {

constexpr int kMaxK = 8;
constexpr int kNQ = 14;      // R1, R2, NOE(old), rho, Nterm(6J4-J2), J0..J4, a1, b1, a2, b2 (R = a + f_CSA*b)
constexpr int kNC = 2;       // covariances: (a1,b1), (a2,b2)

struct RelaxArgs {
    int model, E, nRes, Kmax, B, noe_mode;
    double D0, D1;
    double zeta;          // S2 and C are multiplied by this on load (1.0: already scaled by the caller)
    const double *omega, *f_DD, *f_CSA, *time_fact, *gamma_ratio;
    const double *S2, *C, *tau;
    const int *nComps;
    const double *binvecs, *weights;
    double *out, *Jout, *stats;
};

__device__ __forceinline__ double jomega(double x, double y) { return x / (x * x + y * y); }

__device__ __forceinline__ void quantities(const double *J, double fDD, double fCSA, double tf, double gr, double *q)
{
    const double J0 = J[0], J1 = J[1], J2 = J[2], J3 = J[3], J4 = J[4];
    const double R1 = tf * (fDD * (J2 + 3 * J1 + 6 * J4) + fCSA * J1);
    const double R2 = tf * (0.5 * fDD * (4 * J0 + J2 + 3 * J1 + 6 * J4 + 6 * J3) + 1.0 / 6.0 * fCSA * (4 * J0 + 3 * J1));
    const double Nt = 6 * J4 - J2;
    q[0] = R1;
    q[1] = R2;
    q[2] = 1.0 + tf * gr / R1 * fDD * Nt;
    q[3] = J1 / J0;
    q[4] = Nt;
    q[5] = J0; q[6] = J1; q[7] = J2; q[8] = J3; q[9] = J4;
    // CSA enters only through f_CSA (proportional to csa^2): R1 = a1 + f_CSA*b1, R2 = a2 + f_CSA*b2
    q[10] = tf * (fDD * (J2 + 3 * J1 + 6 * J4));
    q[11] = tf * J1;
    q[12] = tf * (0.5 * fDD * (4 * J0 + J2 + 3 * J1 + 6 * J4 + 6 * J3));
    q[13] = tf * (1.0 / 6.0 * (4 * J0 + 3 * J1));
}

template <int CNT>
__device__ __forceinline__ void block_sum(double *vals, double *red, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < CNT; ++k) {
        const double t = sr_wave_sum_f64(vals[k]);
        if (lane == 0) red[wave * 20 + k] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CNT; ++k) vals[k] = ((red[k] + red[20 + k]) + red[40 + k]) + red[60 + k];
    __syncthreads();
}

// G[j][w] = S2 g(D_j, w) + sum_k C_k g(D_j + 1/tau_k, w): the bin-independent part of J (threads 0..14 fill it)
__device__ __forceinline__ void fill_G(int model, double D0, double D1, double S2, double zeta, const double *C, const double *tau,
                                       int K, const double *om, double (*G)[5], int tid)
{
    if (tid < 15) {
        const int j = tid / 5, w = tid - j * 5;
        double g = 0.0;
        if (model == 2) {
            const double Dpar = D0, Dperp = D1;
            const double DJ = j == 0 ? 5 * Dperp + Dpar : (j == 1 ? 2 * Dperp + 4 * Dpar : 6 * Dperp);
            g = S2 * jomega(DJ, om[w]);
            for (int k = 0; k < K; ++k) { const double ck = zeta * C[k]; g += ck * jomega(DJ + 1. / tau[k], om[w]); }
        } else if (j == 0) {
            if (model == 1) {
                const double tg = 1.0 / (6.0 * D0);
                g = S2 * tg / (1. + (om[w] * tg) * (om[w] * tg));
                for (int k = 0; k < K; ++k) {
                    const double kk = (1.0 / tg) + (1.0 / tau[k]);
                    const double ck = zeta * C[k];
                    g += ck * kk / (kk * kk + om[w] * om[w]);
                }
            } else {
                for (int k = 0; k < K; ++k) { const double ck = zeta * C[k]; g += ck * tau[k] / (1 + (tau[k] * om[w]) * (tau[k] * om[w])); }
            }
        }
        G[j][w] = g;
    }
}

// Weighted mean and variance over the B histogram bins (numpy.average / weighted_average_stdev: two passes) of the kNQ quantities
// of `quantities()` at one (residue, experiment): J(bin, w) = sum_j A_j(bin) G[j][w].  Shared, inlined, by k_relax and by the
// per-residue CSA search of the legacy `--opt new` mode, whose objective must reproduce k_relax's numbers exactly.
__device__ __forceinline__ void relax_bins(const double (*G)[5], double fDD, double fCSA, double tf, double gr, bool prolate, int B,
                                           const double *binvecs, const double *wgt, double *red, int tid, double *mean, double *var)
{
    double wsum;
    {   // pass 1: weighted means
        double s[kNQ + 1];
        for (int k = 0; k <= kNQ; ++k) s[k] = 0.0;
        for (int b = tid; b < B; b += 256) {
            const double w_ = wgt ? wgt[b] : 1.0;
            const double *v = binvecs + (int64_t)b * 3;
            const double z = prolate ? v[2] : v[0];
            const double z2 = z * z, w1 = 1 - z2;
            const double A0 = 3.0 * (z2 * w1), A1 = 0.75 * (w1 * w1), A2 = 0.25 * ((3 * z2 - 1) * (3 * z2 - 1));
            double J[5], q[kNQ];
            for (int w = 0; w < 5; ++w) J[w] = A0 * G[0][w] + A1 * G[1][w] + A2 * G[2][w];
            quantities(J, fDD, fCSA, tf, gr, q);
            for (int k = 0; k < kNQ; ++k) s[k] += w_ * q[k];
            s[kNQ] += w_;
        }
        block_sum<kNQ + 1>(s, red, tid);
        wsum = s[kNQ];
        for (int k = 0; k < kNQ; ++k) mean[k] = s[k] / wsum;
    }
    {   // pass 2: weighted variance about the mean (+ the two covariances the closed-form CSA objective needs)
        double s[kNQ + kNC];
        for (int k = 0; k < kNQ + kNC; ++k) s[k] = 0.0;
        for (int b = tid; b < B; b += 256) {
            const double w_ = wgt ? wgt[b] : 1.0;
            const double *v = binvecs + (int64_t)b * 3;
            const double z = prolate ? v[2] : v[0];
            const double z2 = z * z, w1 = 1 - z2;
            const double A0 = 3.0 * (z2 * w1), A1 = 0.75 * (w1 * w1), A2 = 0.25 * ((3 * z2 - 1) * (3 * z2 - 1));
            double J[5], q[kNQ];
            for (int w = 0; w < 5; ++w) J[w] = A0 * G[0][w] + A1 * G[1][w] + A2 * G[2][w];
            quantities(J, fDD, fCSA, tf, gr, q);
            for (int k = 0; k < kNQ; ++k) { const double d = q[k] - mean[k]; s[k] += w_ * (d * d); }
            s[kNQ] += w_ * ((q[10] - mean[10]) * (q[11] - mean[11]));
            s[kNQ + 1] += w_ * ((q[12] - mean[12]) * (q[13] - mean[13]));
        }
        block_sum<kNQ + kNC>(s, red, tid);
        for (int k = 0; k < kNQ + kNC; ++k) var[k] = s[k] / wsum;
    }
}

__global__ __launch_bounds__(256) void k_relax(RelaxArgs a)
{
    __shared__ double G[3][5];
    __shared__ double red[80];
    const int i = blockIdx.x, e = blockIdx.y, tid = threadIdx.x;
    const double *om = a.omega + e * 5;
    const double fDD = a.f_DD[e], fCSA = a.f_CSA[(int64_t)e * a.nRes + i], tf = a.time_fact[e], gr = a.gamma_ratio[e];
    const double S2 = a.zeta * a.S2[i];
    const double *C = a.C + (int64_t)i * a.Kmax, *tau = a.tau + (int64_t)i * a.Kmax;
    const int K = a.nComps[i];
    const double zeta = a.zeta;
    const bool prolate = a.D0 > a.D1;
    double *out = a.out + ((int64_t)e * a.nRes + i) * 8;
    double *Jout = a.Jout ? a.Jout + ((int64_t)e * a.nRes + i) * 10 : nullptr;
    double *stats = a.stats ? a.stats + ((int64_t)e * a.nRes + i) * 12 : nullptr;

    fill_G(a.model, a.D0, a.D1, S2, zeta, C, tau, K, om, G, tid);
    __syncthreads();

    if (a.model != 2 || a.B == 0) {
        // one J per residue: no distribution, sigma = 0
        if (tid == 0) {
            double J[5], q[kNQ];
            if (a.model == 2) {
                const double *v = a.binvecs + (int64_t)i * 3;
                const double z = prolate ? v[2] : v[0];
                const double z2 = z * z, w1 = 1 - z2;
                const double A0 = 3.0 * (z2 * w1), A1 = 0.75 * (w1 * w1), A2 = 0.25 * ((3 * z2 - 1) * (3 * z2 - 1));
                for (int w = 0; w < 5; ++w) J[w] = A0 * G[0][w] + A1 * G[1][w] + A2 * G[2][w];
            } else {
                for (int w = 0; w < 5; ++w) J[w] = G[0][w];
            }
            quantities(J, fDD, fCSA, tf, gr, q);
            for (int k = 0; k < 4; ++k) { out[2 * k] = q[k]; out[2 * k + 1] = 0.0; }
            if (Jout) for (int w = 0; w < 5; ++w) { Jout[2 * w] = J[w]; Jout[2 * w + 1] = 0.0; }
            if (stats) {
                for (int k = 0; k < 12; ++k) stats[k] = 0.0;
                stats[0] = q[10]; stats[1] = q[11]; stats[5] = q[12]; stats[6] = q[13]; stats[10] = q[4];
            }
        }
        return;
    }

    const double *wgt = a.weights ? a.weights + (int64_t)i * a.B : nullptr;
    double mean[kNQ], var[kNQ + kNC];
    relax_bins(G, fDD, fCSA, tf, gr, prolate, a.B, a.binvecs, wgt, red, tid, mean, var);
    if (tid == 0) {
        out[0] = mean[0]; out[1] = sqrt(var[0]);
        out[2] = mean[1]; out[3] = sqrt(var[1]);
        if (a.noe_mode == 0) {
            out[4] = mean[2]; out[5] = sqrt(var[2]);
        } else {
            // NOE_b = 1 + c*N_b with c = tf*gr*fDD/<R1>: mean and sigma of an affine map of N
            const double c = tf * gr / mean[0] * fDD;
            out[4] = 1.0 + c * mean[4];
            out[5] = fabs(c) * sqrt(var[4]);
        }
        out[6] = mean[3]; out[7] = sqrt(var[3]);
        if (Jout) for (int w = 0; w < 5; ++w) { Jout[2 * w] = mean[5 + w]; Jout[2 * w + 1] = sqrt(var[5 + w]); }
        if (stats) {
            // [a1, b1, Var a1, Cov(a1,b1), Var b1, a2, b2, Var a2, Cov(a2,b2), Var b2, N, Var N]
            stats[0] = mean[10]; stats[1] = mean[11]; stats[2] = var[10]; stats[3] = var[kNQ]; stats[4] = var[11];
            stats[5] = mean[12]; stats[6] = mean[13]; stats[7] = var[12]; stats[8] = var[kNQ + 1]; stats[9] = var[13];
            stats[10] = mean[4]; stats[11] = var[4];
        }
    }
}

__global__ __launch_bounds__(256) void k_jomega(const double *__restrict__ x, const double *__restrict__ y,
                                                double *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = jomega(x[i], y[i]);
}

template <typename T>
T *upload(sr_ctx *ctx, int slot, const T *host, size_t count, int *rc)
{
    T *d = (T *)sr_workspace(ctx, slot, count * sizeof(T));
    if (!d) { *rc = -5; return nullptr; }
    hipError_t e = hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { sr_set_error("upload: %s", hipGetErrorString(e)); *rc = -100 - (int)e; return nullptr; }
    return d;
}


// ---- residue-specific CSA search (new class API) -------------------------------------------------------------------
// spectral_densities.py:1371-1382 + 1430-1447: for every residue, fmin_powell over ONE variable (its CSA) of the
// mean of (value - target)^2 / (sigma_value^2 + sigma_target^2) over the experiments that cover the residue.  With
// the 12 sufficient statistics k_relax already produces, value and sigma are closed forms of csa^2, so a whole search
// is a few dozen evaluations of ~20 float64 operations per experiment: one thread per residue runs it start to end.
// The optimiser is scipy's (unpinned in requirements.txt; 1.15.3 in the build container), restated for N = 1:
//   _minimize_powell        scipy/optimize/_optimize.py:3375-3585   (xtol = ftol = 1e-4, maxiter = maxfev = 1000)
//   _linesearch_powell      :3176-3222  -> Brent without bounds, tol = 100 xtol
//   bracket                 :2916-3073  (xa = 0, xb = 1, grow_limit = 110), _recover_from_bracket_error :3079-3110
//   Brent.optimize          :2463-2572
// Every expression keeps Python's operation order and is compiled without contraction, so the search walks the
// same path as the host loop it replaces and ends on the same last-evaluated CSA (the value the reference keeps:
// it ignores fmin_powell's return value and relies on the objective's side effect).
struct RscsaArgs {
    int E, n, has_err, maxfun, maxiter;
    double step, xtol, ftol;
    const double *stats;          // (E, n, 12)
    const int *col;               // (E) 0 = R1, 1 = R2, 2 = NOE
    const double *pref;           // (E) f_CSA = csa^2 * pref
    const double *cnoe;           // (E) time_fact * gamma_B / gamma_A
    const double *fdd;            // (E)
    const double *y, *dy;         // (E, n) targets and their uncertainties (0 where there is none)
    const unsigned char *cover;   // (E, n)
    const double *csa0;           // (n)
    double *csa, *val, *err, *fopt;   // (n), (E, n), (E, n), (n)
    int *nfev;                    // (n)
};

struct RscsaObjective {
    const RscsaArgs &a;
    int i, ncover, calls;
    bool stop;                    // _MaxFuncCallError raised: unwind to _minimize_powell's loop
    double last;

    __device__ void closed_form(int e, double csa, double &v, double &dv) const
    {
#pragma clang fp contract(off)
        const double *st = a.stats + ((size_t)e * a.n + i) * 12;
        const double f = csa * csa * a.pref[e];
        double var;
        if (a.col[e] == 0) {
            v = st[0] + f * st[1];
            var = st[2] + 2.0 * f * st[3] + f * f * st[4];
        } else if (a.col[e] == 1) {
            v = st[5] + f * st[6];
            var = st[7] + 2.0 * f * st[8] + f * f * st[9];
        } else {
            const double R1 = st[0] + f * st[1];
            const double c = a.cnoe[e] / R1 * a.fdd[e];
            v = 1.0 + c * st[10];
            var = c * c * st[11];
        }
        dv = a.has_err ? sqrt(0.0 > var ? 0.0 : var) : 0.0;
    }

    __device__ double operator()(double csa)
    {
#pragma clang fp contract(off)
        if (calls >= a.maxfun) { stop = true; return NAN; }
        calls += 1;
        last = csa;
        double chisq = 0.0;
        for (int e = 0; e < a.E; ++e) {
            if (!a.cover[(size_t)e * a.n + i]) continue;
            double v, dv;
            closed_form(e, csa, v, dv);
            const double dt = a.dy[(size_t)e * a.n + i];
            double w = dv * dv + dt * dt;
            if (w == 0) w = 1.0;
            const double r = v - a.y[(size_t)e * a.n + i];
            chisq += r * r / w;
        }
        return chisq / (double)ncover;
    }
};

// min over alpha of f(p + alpha*xi); returns alpha_min and the value there (Brent on bracket(0, 1)).
template <class F>
__device__ void powell_line_min(F &f, double p, double xi, double tol, double &alpha_min, double &fret)
{
#pragma clang fp contract(off)
    auto g = [&](double alpha) { return f(p + alpha * xi); };
    // ---- bracket ----
    const double gold = 1.618034, verysmall = 1e-21, grow_limit = 110.0;
    double xa = 0.0, xb = 1.0;
    double fa = g(xa); if (f.stop) return;
    double fb = g(xb); if (f.stop) return;
    if (fa < fb) { double t = xa; xa = xb; xb = t; t = fa; fa = fb; fb = t; }
    double xc = xb + gold * (xb - xa);
    double fc = g(xc); if (f.stop) return;
    int it = 0;
    while (fc < fb) {
        const double tmp1 = (xb - xa) * (fb - fc);
        const double tmp2 = (xb - xc) * (fb - fa);
        const double val = tmp2 - tmp1;
        const double denom = fabs(val) < verysmall ? 2.0 * verysmall : 2.0 * val;
        double w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom;
        const double wlim = xb + grow_limit * (xc - xb);
        if (it > 1000) { f.stop = true; return; }       // scipy raises RuntimeError here
        it += 1;
        double fw;
        if ((w - xc) * (xb - w) > 0.0) {
            fw = g(w); if (f.stop) return;
            if (fw < fc) { xa = xb; xb = w; fa = fb; fb = fw; break; }
            else if (fw > fb) { xc = w; fc = fw; break; }
            w = xc + gold * (xc - xb);
            fw = g(w); if (f.stop) return;
        } else if ((w - wlim) * (wlim - xc) >= 0.0) {
            w = wlim;
            fw = g(w); if (f.stop) return;
        } else if ((w - wlim) * (xc - w) > 0.0) {
            fw = g(w); if (f.stop) return;
            if (fw < fc) {
                xb = xc; xc = w; w = xc + gold * (xc - xb);
                fb = fc; fc = fw;
                fw = g(w); if (f.stop) return;
            }
        } else {
            w = xc + gold * (xc - xb);
            fw = g(w); if (f.stop) return;
        }
        xa = xb; xb = xc; xc = w;
        fa = fb; fb = fc; fc = fw;
    }
    const bool cond1 = (fb < fc && fb <= fa) || (fb < fa && fb <= fc);
    const bool cond2 = (xa < xb && xb < xc) || (xc < xb && xb < xa);
    const bool cond3 = isfinite(xa) && isfinite(xb) && isfinite(xc);
    if (!(cond1 && cond2 && cond3)) {
        // _recover_from_bracket_error: the best of the three points (numpy.argmin: first minimum, NaN wins)
        if (isnan(xa) || isnan(xb) || isnan(xc) || isnan(fa) || isnan(fb) || isnan(fc)) { alpha_min = NAN; fret = NAN; return; }
        alpha_min = xa; fret = fa;
        if (fb < fret) { alpha_min = xb; fret = fb; }
        if (fc < fret) { alpha_min = xc; fret = fc; }
        return;
    }
    // ---- Brent ----
    const double mintol = 1.0e-11, cg = 0.3819660;
    double x = xb, w = xb, v = xb, fx = fb, fw = fb, fv = fb;
    double lo = xa < xc ? xa : xc, hi = xa < xc ? xc : xa;
    double deltax = 0.0, rat = 0.0;
    for (int iter = 0; iter < 500; ++iter) {
        const double tol1 = tol * fabs(x) + mintol;
        const double tol2 = 2.0 * tol1;
        const double xmid = 0.5 * (lo + hi);
        if (fabs(x - xmid) < (tol2 - 0.5 * (hi - lo))) break;
        if (fabs(deltax) <= tol1) {
            deltax = (x >= xmid) ? lo - x : hi - x;
            rat = cg * deltax;
        } else {
            double tmp1 = (x - w) * (fx - fv);
            double tmp2 = (x - v) * (fx - fw);
            double pp = (x - v) * tmp2 - (x - w) * tmp1;
            tmp2 = 2.0 * (tmp2 - tmp1);
            if (tmp2 > 0.0) pp = -pp;
            tmp2 = fabs(tmp2);
            const double dx_temp = deltax;
            deltax = rat;
            if ((pp > tmp2 * (lo - x)) && (pp < tmp2 * (hi - x)) && (fabs(pp) < fabs(0.5 * tmp2 * dx_temp))) {
                rat = pp * 1.0 / tmp2;
                const double u = x + rat;
                if ((u - lo) < tol2 || (hi - u) < tol2) rat = (xmid - x >= 0) ? tol1 : -tol1;
            } else {
                deltax = (x >= xmid) ? lo - x : hi - x;
                rat = cg * deltax;
            }
        }
        double u;
        if (fabs(rat) < tol1) u = (rat >= 0) ? x + tol1 : x - tol1;
        else u = x + rat;
        const double fu = g(u); if (f.stop) return;
        if (fu > fx) {
            if (u < x) lo = u; else hi = u;
            if ((fu <= fw) || (w == x)) { v = w; w = u; fv = fw; fw = fu; }
            else if ((fu <= fv) || (v == x) || (v == w)) { v = u; fv = fu; }
        } else {
            if (u >= x) lo = x; else hi = x;
            v = w; w = x; x = u;
            fv = fw; fw = fx; fx = fu;
        }
    }
    alpha_min = x;
    fret = fx;
}

// scipy.optimize._optimize._minimize_powell for ONE variable (N = 1, no bounds), operation for operation; f.stop unwinds like
// _MaxFuncCallError.  x: Powell's current point, fval: the value there.
template <class F>
__device__ void powell_min_1d(F &f, double x0, double step, double xtol, double ftol, int maxiter, int maxfun, double &x, double &fval)
{
#pragma clang fp contract(off)
    x = x0;
    double direc = step;
    fval = f(x);
    double x1 = x;
    int iter = 0;
    while (!f.stop) {
        const double fx = fval;
        double delta = 0.0;
        {
            const double fx2 = fval;
            if (direc != 0.0) {
                double amin = 0.0, fret = fval;
                powell_line_min(f, x, direc, xtol * 100, amin, fret);
                if (f.stop) break;
                x = x + amin * direc;          // the scaled direction is a local of the loop: direc[0] itself is kept
                fval = fret;
            }
            if ((fx2 - fval) > delta) delta = fx2 - fval;
        }
        iter += 1;
        const double bnd = ftol * (fabs(fx) + fabs(fval)) + 1e-20;
        if (2.0 * (fx - fval) <= bnd) break;
        if (f.calls >= maxfun) break;
        if (iter >= maxiter) break;
        if (isnan(fx) && isnan(fval)) break;
        double direc1 = x - x1;
        x1 = x;
        const double x2 = x + direc1;          // min(lmax, 1) * direc1 with lmax = 1
        const double fx2 = f(x2);
        if (f.stop) break;
        if (fx > fx2) {
            double t = 2.0 * (fx + fx2 - 2.0 * fval);
            double temp = fx - fval - delta;
            t *= temp * temp;
            temp = fx - fx2;
            t -= delta * temp * temp;
            if (t < 0.0) {
                if (direc1 != 0.0) {
                    double amin = 0.0, fret = fval;
                    powell_line_min(f, x, direc1, xtol * 100, amin, fret);
                    if (f.stop) break;
                    direc1 = amin * direc1;
                    x = x + direc1;
                    fval = fret;
                }
                if (direc1 != 0.0) direc = direc1;
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_rscsa_search(RscsaArgs a)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= a.n) return;
    RscsaObjective f{a, i, 0, 0, false, a.csa0[i]};
    for (int e = 0; e < a.E; ++e) f.ncover += a.cover[(size_t)e * a.n + i] ? 1 : 0;
    a.nfev[i] = 0;
    a.csa[i] = a.csa0[i];
    a.fopt[i] = NAN;
    if (f.ncover == 0) return;          // the reference skips residues no experiment covers
    double x, fval;
    powell_min_1d(f, a.csa0[i], a.step, a.xtol, a.ftol, a.maxiter, a.maxfun, x, fval);
    a.nfev[i] = f.calls;
    a.csa[i] = f.last;
    a.fopt[i] = fval;
    for (int e = 0; e < a.E; ++e) {
        if (!a.cover[(size_t)e * a.n + i]) continue;
        double v, dv;
        f.closed_form(e, f.last, v, dv);
        a.val[(size_t)e * a.n + i] = v;
        a.err[(size_t)e * a.n + i] = dv;
    }
}

// ---- legacy `--opt new`: the per-residue CSA refinement (calculate-relaxations-from-Ct.py:935-1000) ------------------------
// For every residue: fmin_powell over ONE variable, the CSA, of optfunc_R1R2NOE_new (:210-258) -- R1, R2, NOE (old API: per-vector
// R1 in the NOE) as weighted mean and sigma over the histogram bins, cast to float32 like the reference's datablock, against the
// measured triple: mean_k (model_k - exp_k)^2 / (sigma_exp_k^2 + sigma_model_k^2).  The NOE of the old API is not affine in
// f_CSA, so every objective call walks the 2 592 bins: one workgroup per residue runs scipy's Powell search workgroup-uniformly
// (every thread the same control flow on the same values) and the objective is a workgroup reduction -- relax_bins, the very
// code k_relax runs for the host-driven search, so both searches see the same numbers and take the same path.
struct LegacyArgs {
    int nRes, Kmax, B, maxiter, maxfun;
    double D0, D1, fDD, g2, tf, gr, step, xtol, ftol;
    const double *omega;          // (5)
    const double *S2, *C, *tau;   // (nRes), (nRes, Kmax) x 2: already scaled by zeta
    const int *nComps;            // (nRes)
    const double *binvecs;        // (B, 3)
    const double *weights;        // (nRes, B)
    const double *expt;           // (nRes, 3, 2): R1, R2, NOE x (value, sigma)
    const double *csa0;           // (nRes)
    double *csa, *fopt;           // (nRes)
    int *nfev;                    // (nRes)
};

struct LegacyObjective {
    const LegacyArgs &a;
    const double (*G)[5];
    double *red;
    const double *wgt;
    const double *ex;             // this residue's (3, 2)
    bool prolate;
    int tid, calls;
    bool stop;

    __device__ double operator()(double csa)
    {
        if (calls >= a.maxfun) { stop = true; return NAN; }
        calls += 1;
        double fcsa;
        {
#pragma clang fp contract(off)
            fcsa = ((2.0 / 15.0) * (csa * csa)) * a.g2;          // get_f_CSA: 2.0/15.0 * csa**2.0 * (gamma B0)**2
        }
        double mean[kNQ], var[kNQ + kNC];
        relax_bins(G, a.fDD, fcsa, a.tf, a.gr, prolate, a.B, a.binvecs, wgt, red, tid, mean, var);
        double acc = 0.0;
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d0 = (float)mean[k], d1 = (float)sqrt(var[k]);      // the reference's float32 datablock
                const double r = (double)d0 - ex[2 * k];
                const float sq = d1 * d1;                                       // numpy squares the float32 column in float32
                const double sig = ex[2 * k + 1] * ex[2 * k + 1] + (double)sq;
                acc += (r * r) / sig;
            }
            acc = acc / 3.0;
        }
        return acc;
    }
};

__global__ __launch_bounds__(256) void k_legacy_csa_search(LegacyArgs a)
{
    __shared__ double G[3][5];
    __shared__ double red[80];
    const int i = blockIdx.x, tid = threadIdx.x;
    fill_G(2, a.D0, a.D1, a.S2[i], 1.0, a.C + (int64_t)i * a.Kmax, a.tau + (int64_t)i * a.Kmax, a.nComps[i], a.omega, G, tid);
    __syncthreads();
    LegacyObjective f{a, G, red, a.weights + (int64_t)i * a.B, a.expt + (int64_t)i * 6, a.D0 > a.D1, tid, 0, false};
    double x, fval;
    powell_min_1d(f, a.csa0[i], a.step, a.xtol, a.ftol, a.maxiter, a.maxfun, x, fval);
    if (tid == 0) {
        a.csa[i] = x;
        a.fopt[i] = fval;
        a.nfev[i] = f.calls;
    }
}

}  // namespace

extern "C" {

int sr_jomega_f64(sr_ctx *ctx, const double *x, const double *y, double *out, int64_t n)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(x && y && out && n >= 0, -2, "sr_jomega_f64: bad arguments");
    if (n == 0) return 0;
    int rc = 0;
    double *dx = upload(ctx, SR_WS_IN0, x, (size_t)n, &rc);
    double *dy = upload(ctx, SR_WS_IN1, y, (size_t)n, &rc);
    double *dout = (double *)sr_workspace(ctx, SR_WS_OUT0, (size_t)n * sizeof(double));
    if (rc || !dx || !dy || !dout) return rc ? rc : -5;
    hipLaunchKernelGGL(k_jomega, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dx, dy, dout, n);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_jomega_relax_f64(sr_ctx *ctx, int model, const double *D, int E, const double *omega, const double *f_DD,
                        const double *f_CSA, const double *time_fact, const double *gamma_ratio, int nRes, int Kmax,
                        const double *S2, const double *C, const double *tau, const int *nComps, int B,
                        const double *binvecs, const double *weights, int weights_on_device, int noe_mode, double *out,
                        double *Jout, double *stats)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(model >= 0 && model <= 2, -3, "sr_jomega_relax_f64: model must be 0, 1 or 2");
    SR_REQUIRE(E >= 1 && nRes >= 1 && Kmax >= 1 && Kmax <= kMaxK && B >= 0, -3, "sr_jomega_relax_f64: bad sizes");
    SR_REQUIRE(E <= 65535, -3, "sr_jomega_relax_f64: too many experiments");
    SR_REQUIRE(omega && f_DD && f_CSA && time_fact && gamma_ratio && S2 && C && tau && nComps && out, -2,
               "sr_jomega_relax_f64: null pointer");
    SR_REQUIRE(model == 0 || D, -2, "sr_jomega_relax_f64: D required");
    SR_REQUIRE(model != 2 || binvecs, -2, "sr_jomega_relax_f64: symmetric top needs vectors");
    SR_REQUIRE(noe_mode == 0 || noe_mode == 1, -3, "sr_jomega_relax_f64: noe_mode must be 0 or 1");
    for (int i = 0; i < nRes; ++i)
        SR_REQUIRE(nComps[i] >= 0 && nComps[i] <= Kmax, -3, "sr_jomega_relax_f64: nComps[%d]=%d out of range", i, nComps[i]);
    // pack every input into one staging buffer
    const size_t nE = (size_t)E, nR = (size_t)nRes;
    const size_t cnt = nE * 5 + nE + nE * nR + nE + nE + nR + 2 * nR * Kmax +
                       (model == 2 ? (B > 0 ? (size_t)B * 3 : nR * 3) : 0) +
                       ((B > 0 && weights && !weights_on_device) ? nR * B : 0);
    double *stage = (double *)sr_workspace(ctx, SR_WS_IN0, cnt * sizeof(double));
    int *ncomp_d = (int *)sr_workspace(ctx, SR_WS_IN1, nR * sizeof(int));
    double *out_d = (double *)sr_workspace(ctx, SR_WS_OUT0, nE * nR * 8 * sizeof(double));
    double *J_d = Jout ? (double *)sr_workspace(ctx, SR_WS_OUT1, nE * nR * 10 * sizeof(double)) : nullptr;
    double *st_d = stats ? (double *)sr_workspace(ctx, SR_WS_OUT2, nE * nR * 12 * sizeof(double)) : nullptr;
    if (!stage || !ncomp_d || !out_d || (Jout && !J_d) || (stats && !st_d)) return -5;
    RelaxArgs a;
    a.model = model; a.E = E; a.nRes = nRes; a.Kmax = Kmax; a.B = (model == 2) ? B : 0; a.noe_mode = noe_mode;
    a.D0 = D ? D[0] : 0.0;
    a.D1 = (D && model == 2) ? D[1] : 0.0;
    a.zeta = 1.0;
    double *p = stage;
    auto put = [&](const double *src, size_t n) -> const double * {
        hipError_t e = hipMemcpyAsync(p, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return nullptr;
        const double *r = p;
        p += n;
        return r;
    };
    a.omega = put(omega, nE * 5);
    a.f_DD = put(f_DD, nE);
    a.f_CSA = put(f_CSA, nE * nR);
    a.time_fact = put(time_fact, nE);
    a.gamma_ratio = put(gamma_ratio, nE);
    a.S2 = put(S2, nR);
    a.C = put(C, nR * Kmax);
    a.tau = put(tau, nR * Kmax);
    a.binvecs = nullptr;
    a.weights = nullptr;
    if (model == 2) a.binvecs = put(binvecs, B > 0 ? (size_t)B * 3 : nR * 3);
    if (model == 2 && B > 0 && weights) a.weights = weights_on_device ? weights : put(weights, nR * B);
    SR_REQUIRE(a.omega && a.f_DD && a.f_CSA && a.time_fact && a.gamma_ratio && a.S2 && a.C && a.tau, -6,
               "sr_jomega_relax_f64: host to device copy failed");
    SR_HIP(hipMemcpyAsync(ncomp_d, nComps, nR * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    a.nComps = ncomp_d;
    a.out = out_d;
    a.Jout = J_d;
    a.stats = st_d;
    hipLaunchKernelGGL(k_relax, dim3((unsigned)nRes, (unsigned)E), dim3(256), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(out, out_d, nE * nR * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (Jout) SR_HIP(hipMemcpyAsync(Jout, J_d, nE * nR * 10 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (stats) SR_HIP(hipMemcpyAsync(stats, st_d, nE * nR * 12 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_jomega_relax_f64_dev(sr_ctx *ctx, int model, const double *D, int E, const double *omega, const double *f_DD,
                            const double *f_CSA, const double *time_fact, const double *gamma_ratio, int nRes, int Kmax,
                            double zeta, const double *S2, const double *C, const double *tau, const int *nComps, int B,
                            const double *binvecs, const double *weights, int noe_mode, double *out, double *Jout,
                            double *stats)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(model >= 0 && model <= 2, -3, "sr_jomega_relax_f64_dev: model must be 0, 1 or 2");
    SR_REQUIRE(E >= 1 && E <= 65535 && nRes >= 1 && Kmax >= 1 && Kmax <= kMaxK && B >= 0, -3, "sr_jomega_relax_f64_dev: bad sizes");
    SR_REQUIRE(omega && f_DD && f_CSA && time_fact && gamma_ratio && S2 && C && tau && nComps && out, -2,
               "sr_jomega_relax_f64_dev: null pointer");
    SR_REQUIRE(model == 0 || D, -2, "sr_jomega_relax_f64_dev: D required");
    SR_REQUIRE(model != 2 || binvecs, -2, "sr_jomega_relax_f64_dev: symmetric top needs vectors");
    SR_REQUIRE(noe_mode == 0 || noe_mode == 1, -3, "sr_jomega_relax_f64_dev: noe_mode must be 0 or 1");
    RelaxArgs a;
    a.model = model; a.E = E; a.nRes = nRes; a.Kmax = Kmax; a.B = (model == 2) ? B : 0; a.noe_mode = noe_mode;
    a.D0 = D ? D[0] : 0.0;
    a.D1 = (D && model == 2) ? D[1] : 0.0;
    a.zeta = zeta;
    a.omega = omega; a.f_DD = f_DD; a.f_CSA = f_CSA; a.time_fact = time_fact; a.gamma_ratio = gamma_ratio;
    a.S2 = S2; a.C = C; a.tau = tau; a.nComps = nComps;
    a.binvecs = model == 2 ? binvecs : nullptr;
    a.weights = (model == 2 && B > 0) ? weights : nullptr;
    a.out = out; a.Jout = Jout; a.stats = stats;
    hipLaunchKernelGGL(k_relax, dim3((unsigned)nRes, (unsigned)E), dim3(256), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    return 0;
}

int sr_rscsa_search_f64(sr_ctx *ctx, int E, int nRes, const double *stats, const int *column, const double *csa_prefactor,
                        const double *noe_factor, const double *f_DD, const double *target, const double *dtarget,
                        const unsigned char *cover, int has_err, const double *csa0, double step, double xtol, double ftol,
                        double *csa, double *values, double *errors, double *fopt, int *nfev)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(E >= 1 && nRes >= 1, -3, "sr_rscsa_search_f64: bad sizes");
    SR_REQUIRE(stats && column && csa_prefactor && noe_factor && f_DD && target && dtarget && cover && csa0 && csa && values &&
               errors && fopt && nfev, -2, "sr_rscsa_search_f64: null pointer");
    for (int e = 0; e < E; ++e)
        SR_REQUIRE(column[e] >= 0 && column[e] <= 2, -3, "sr_rscsa_search_f64: column[%d]=%d is not 0 (R1), 1 (R2) or 2 (NOE)", e, column[e]);
    const size_t nE = (size_t)E, nR = (size_t)nRes, EN = nE * nR;
    const size_t nd_in = EN * 12 + 3 * nE + 2 * EN + nR;
    double *din = (double *)sr_workspace(ctx, SR_WS_IN0, nd_in * sizeof(double));
    int *iin = (int *)sr_workspace(ctx, SR_WS_IN1, nE * sizeof(int) + EN);
    double *dout = (double *)sr_workspace(ctx, SR_WS_OUT0, (2 * EN + 2 * nR) * sizeof(double) + nR * sizeof(int));
    if (!din || !iin || !dout) return -5;
    double *p = din;
    auto put = [&](const double *src, size_t n) -> const double * {
        hipError_t e = hipMemcpyAsync(p, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return nullptr;
        const double *r = p;
        p += n;
        return r;
    };
    RscsaArgs a;
    a.E = E; a.n = nRes; a.has_err = has_err ? 1 : 0; a.maxfun = 1000; a.maxiter = 1000;
    a.step = step; a.xtol = xtol; a.ftol = ftol;
    a.stats = put(stats, EN * 12);
    a.pref = put(csa_prefactor, nE);
    a.cnoe = put(noe_factor, nE);
    a.fdd = put(f_DD, nE);
    a.y = put(target, EN);
    a.dy = put(dtarget, EN);
    a.csa0 = put(csa0, nR);
    SR_REQUIRE(a.stats && a.pref && a.cnoe && a.fdd && a.y && a.dy && a.csa0, -6, "sr_rscsa_search_f64: host to device copy failed");
    SR_HIP(hipMemcpyAsync(iin, column, nE * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    unsigned char *cov_d = (unsigned char *)(iin + nE);
    SR_HIP(hipMemcpyAsync(cov_d, cover, EN, hipMemcpyHostToDevice, ctx->stream));
    a.col = iin;
    a.cover = cov_d;
    a.val = dout; a.err = dout + EN; a.csa = dout + 2 * EN; a.fopt = a.csa + nR;
    a.nfev = (int *)(a.fopt + nR);
    SR_HIP(hipMemsetAsync(dout, 0, 2 * EN * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(k_rscsa_search, dim3((unsigned)((nRes + 63) / 64)), dim3(64), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(values, a.val, EN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(errors, a.err, EN * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(csa, a.csa, nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(fopt, a.fopt, nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(nfev, a.nfev, nR * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int sr_legacy_csa_search_f64(sr_ctx *ctx, const double *D, const double *omega, double f_DD, double gammaB0_sq, double time_fact,
                             double gamma_ratio, int nRes, int Kmax, const double *S2, const double *C, const double *tau,
                             const int *nComps, int B, const double *binvecs, const double *weights, const double *expt,
                             const double *csa0, double step, double xtol, double ftol, int maxiter, int maxfun, double *csa,
                             double *fopt, int *nfev)
{
    SR_CHECK_CTX(ctx);
    SR_REQUIRE(nRes >= 1 && Kmax >= 1 && Kmax <= kMaxK && B >= 1, -3, "sr_legacy_csa_search_f64: bad sizes");
    SR_REQUIRE(D && omega && S2 && C && tau && nComps && binvecs && weights && expt && csa0 && csa && fopt && nfev, -2,
               "sr_legacy_csa_search_f64: null pointer");
    SR_REQUIRE(maxiter >= 1 && maxfun >= 1, -3, "sr_legacy_csa_search_f64: maxiter / maxfun must be positive");
    for (int i = 0; i < nRes; ++i)
        SR_REQUIRE(nComps[i] >= 0 && nComps[i] <= Kmax, -3, "sr_legacy_csa_search_f64: nComps[%d]=%d out of range", i, nComps[i]);
    const size_t nR = (size_t)nRes;
    const size_t cnt = 5 + nR + 2 * nR * Kmax + (size_t)B * 3 + nR * B + nR * 6 + nR;
    double *stage = (double *)sr_workspace(ctx, SR_WS_IN0, cnt * sizeof(double));
    int *ncomp_d = (int *)sr_workspace(ctx, SR_WS_IN1, nR * sizeof(int));
    double *dout = (double *)sr_workspace(ctx, SR_WS_OUT0, 2 * nR * sizeof(double) + nR * sizeof(int));
    if (!stage || !ncomp_d || !dout) return -5;
    double *p = stage;
    auto put = [&](const double *src, size_t n) -> const double * {
        hipError_t e = hipMemcpyAsync(p, src, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return nullptr;
        const double *r = p;
        p += n;
        return r;
    };
    LegacyArgs a;
    a.nRes = nRes; a.Kmax = Kmax; a.B = B; a.maxiter = maxiter; a.maxfun = maxfun;
    a.D0 = D[0]; a.D1 = D[1]; a.fDD = f_DD; a.g2 = gammaB0_sq; a.tf = time_fact; a.gr = gamma_ratio;
    a.step = step; a.xtol = xtol; a.ftol = ftol;
    a.omega = put(omega, 5);
    a.S2 = put(S2, nR);
    a.C = put(C, nR * Kmax);
    a.tau = put(tau, nR * Kmax);
    a.binvecs = put(binvecs, (size_t)B * 3);
    a.weights = put(weights, nR * B);
    a.expt = put(expt, nR * 6);
    a.csa0 = put(csa0, nR);
    SR_REQUIRE(a.omega && a.S2 && a.C && a.tau && a.binvecs && a.weights && a.expt && a.csa0, -6,
               "sr_legacy_csa_search_f64: host to device copy failed");
    SR_HIP(hipMemcpyAsync(ncomp_d, nComps, nR * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    a.nComps = ncomp_d;
    a.csa = dout; a.fopt = dout + nR; a.nfev = (int *)(dout + 2 * nR);
    hipLaunchKernelGGL(k_legacy_csa_search, dim3((unsigned)nRes), dim3(256), 0, ctx->stream, a);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(csa, a.csa, nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(fopt, a.fopt, nR * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipMemcpyAsync(nfev, a.nfev, nR * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SR_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // extern "C"
