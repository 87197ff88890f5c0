"""
sr_oracle.py -- CPU restatement (numpy, float64) of the SpinRelax hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under spinrelax_amd/ imports this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as the checker.

Every function restates the algorithm of the cited reference lines (paths relative to
/root/reference).  Parity of this restatement against the real reference is pinned by
oracle/gen_golden.py, which imports the reference in the build container, runs both on the same
seeded inputs and stores the reference's outputs under tests/golden/ (see tests/test_oracle_golden.py).

Third-party arithmetic the reference itself delegates to (scipy.optimize.curve_fit -> least_squares
TRF, numpy.histogramdd, numpy.average) is called here the same way the reference calls it.
"""
import numpy as np

# ----------------------------------------------------------------------------------------------
# C(t): calculate-Ct-from-traj.py
# ----------------------------------------------------------------------------------------------

def reformat_vecs_by_tau(vec_list, dt, tau):
    """calculate-Ct-from-traj.py:245-275.  list of (nFr_i, V, 3) -> (R, F, V, 3); the tail of
    each file that does not fill a block of F=int(tau/dt) frames is dropped per file."""
    F = int(tau / dt)
    kept = [np.asarray(v)[: (v.shape[0] // F) * F] for v in vec_list]
    cat = np.concatenate(kept, axis=0)
    return cat.reshape(cat.shape[0] // F, F, cat.shape[1], cat.shape[2])


def calculate_dt(dt, tau):
    """calculate-Ct-from-traj.py:240-243."""
    return (np.arange(int(0.5 * tau / dt)) + 1.0) * dt


def calculate_Ct_Palmer(vecs, dtype=np.float64):
    """calculate-Ct-from-traj.py:200-238 evaluated in `dtype` (float64 = parity oracle; float32 =
    what the reference produces from MDTraj float32 coordinates).

    vecs (R, F, V, 3).  For delta = 1..F//2:
        p[r, v]   = (1/(F-delta)) * sum_j ( 1.5 (u[r,j,v] . u[r,j+delta,v])^2 - 0.5 )
        Ct[d-1,v] = mean_r p ;  dCt[d-1,v] = std_r(p, ddof=0) / (sqrt(R) - 1)
    """
    u = np.asarray(vecs, dtype=dtype)
    R, F, V, _ = u.shape
    L = F // 2
    Ct = np.zeros((L, V), dtype=dtype)
    dCt = np.zeros((L, V), dtype=dtype)
    for d in range(1, L + 1):
        dots = np.einsum('rjvc,rjvc->rjv', u[:, :-d], u[:, d:])
        p2 = -0.5 + 1.5 * np.square(dots)
        p = np.einsum('rjv->rv', p2) / (F - d)
        Ct[d - 1] = np.mean(p, axis=0)
        with np.errstate(divide='ignore', invalid='ignore'):
            dCt[d - 1] = np.std(p, axis=0) / (np.sqrt(R) - 1.0)
    return Ct, dCt


def calculate_Ct_Palmer_perchunk(vecs):
    """Per-replicate means p[r, d-1, v] in float64 (the quantity the GPU kernel produces before the
    replicate statistics); explicit-loop restatement used for cross-checks."""
    u = np.asarray(vecs, dtype=np.float64)
    R, F, V, _ = u.shape
    L = F // 2
    p = np.empty((R, L, V))
    for d in range(1, L + 1):
        x = np.sum(u[:, :-d] * u[:, d:], axis=-1)
        p[:, d - 1] = 1.5 * np.mean(x * x, axis=1) - 0.5
    return p


def calculate_Ct_fft(vecs):
    """Independent cross-check (NOT the reference algorithm): sum_t (u(t).u(t+d))^2 equals the sum
    of the ordinary autocorrelations of the 9 products u_a u_b."""
    u = np.asarray(vecs, dtype=np.float64)
    R, F, V, _ = u.shape
    L = F // 2
    T = np.einsum('rjva,rjvb->rjvab', u, u).reshape(R, F, V, 9)
    n = 1
    while n < 2 * F:
        n *= 2
    ft = np.fft.rfft(T, n=n, axis=1)
    ac = np.fft.irfft(ft * np.conj(ft), n=n, axis=1)[:, 1:L + 1].sum(axis=-1)   # (R, L, V)
    cnt = (F - np.arange(1, L + 1))[None, :, None]
    p = 1.5 * ac / cnt - 0.5
    Ct = p.mean(axis=0)
    with np.errstate(divide='ignore', invalid='ignore'):
        dCt = p.std(axis=0) / (np.sqrt(R) - 1.0)
    return Ct, dCt


def exact_triples(R, F, V):
    """Number of P2(u.u') evaluations one C(t) call performs."""
    L = F // 2
    return R * V * (L * F - L * (L + 1) // 2)

# ----------------------------------------------------------------------------------------------
# rotation, spherical coordinates, Lambert histogram, mean vector, S2
# ----------------------------------------------------------------------------------------------

def vecnorm_NDarray(v, axis=-1):
    """transforms3d_supplement.py:40-52 (0/0 -> 0 through nan_to_num)."""
    v = np.asarray(v)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.nan_to_num(v / np.linalg.norm(v, axis=axis, keepdims=True))


def rotate_vector_simd(v, q):
    """transforms3d_supplement.py:270-296 with axis=-1: q normalised, a = qv x v + qw v,
    b = qv x a, v' = v + 2 b.  q is (4,) or broadcastable (..., 4); result is float64 when q is."""
    q = vecnorm_NDarray(np.asarray(q, dtype=np.float64))
    v = np.asarray(v)
    qw = q[..., 0:1]
    qv = q[..., 1:4]
    a = np.cross(qv, v) + qw * v
    b = np.cross(qv, a)
    return b + b + v


def xyz_to_rtp(uv):
    """general_maths.py:118-158, non-unit branch, vaxis=-1."""
    uv = np.asarray(uv)
    out = np.zeros_like(uv)
    out[..., 0] = np.linalg.norm(uv, axis=-1)
    out[..., 1] = np.arctan2(uv[..., 1], uv[..., 0])
    with np.errstate(divide='ignore', invalid='ignore'):
        out[..., 2] = np.arccos(uv[..., 2] / out[..., 0])
    return out


def lambert_edges(nphi=72):
    """The edges numpy.histogramdd builds for bins=(nphi, nphi//2),
    range=((-pi,pi),(-1,1)) (calculate-Ct-from-traj.py:618)."""
    ncos = int(nphi / 2)
    return [np.linspace(-np.pi, np.pi, nphi + 1), np.linspace(-1.0, 1.0, ncos + 1)]


def lambert_histogram(vecs, nphi=72):
    """calculate-Ct-from-traj.py:585-626 (the `normed=False` kwarg, removed from numpy 2, dropped).
    vecs (N, V, 3) in the frame to be histogrammed.  Returns hist (V, nphi, nphi//2), edges."""
    ncos = int(nphi / 2)
    rtp = np.transpose(xyz_to_rtp(vecs), axes=(1, 0, 2))
    rtp = np.delete(rtp, 0, axis=2)
    rtp[..., 1] = np.cos(rtp[..., 1])
    V = rtp.shape[0]
    hist = np.zeros((V, nphi, ncos), dtype=rtp.dtype)
    edges = None
    for i in range(V):
        h, e = np.histogramdd(rtp[i], bins=(nphi, ncos), range=((-np.pi, np.pi), (-1, 1)))
        if edges is None:
            edges = e
        hist[i] = h
    return hist, edges


def mean_vector(vecs):
    """calculate-Ct-from-traj.py:579-583 + general_scripts.py:11-16."""
    m = np.mean(vecs, axis=0)
    return m / np.sqrt((m ** 2).sum(-1))[..., np.newaxis]


def calculate_S2_by_outerProduct(vecs, delta_t=-1, tau_memory=-1):
    """calculate-Ct-from-traj.py:96-145, 3-D branch (time, residue, 3)."""
    vecs = np.asarray(vecs)
    N, V, _ = vecs.shape
    if delta_t < 0 or tau_memory < 0:
        m = np.einsum('ijk,ijl->jkl', vecs, vecs) / N
        return 1.5 * np.einsum('...ij,...ij->...', m, m) - 0.5
    F = int(tau_memory / delta_t)
    nB = int(N / F)
    blk = vecs[: nB * F].reshape(nB, F, V, 3)
    m = np.einsum('ijkl,ijkm->iklm', blk, blk) / F
    s = 1.5 * np.einsum('...ij,...ij->...', m, m) - 0.5
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.stack((np.mean(s, axis=0), np.std(s, axis=0) / (np.sqrt(nB) - 1.0)), axis=-1)

# ----------------------------------------------------------------------------------------------
# multi-exponential C(t) fit: fitting_Ct_functions.py
# ----------------------------------------------------------------------------------------------

def curvefit_exponential(t, *params):
    """fitting_Ct_functions.py:419-427.  params = [C_1..C_K, tau_1..tau_K (, S2)]."""
    n = len(params)
    K = n // 2
    C = np.array(params[:K], dtype=float)
    tau = np.array(params[K:2 * K], dtype=float)
    S2 = params[-1] if n % 2 == 1 else 1.0 - np.sum(C)
    return S2 + np.sum(C[:, None] * np.exp(-1.0 * t[None, :] / tau[:, None]), axis=0)


def calc_chiSq(t, y, dy, S2, C, tau, zeta=1.0):
    """fitting_Ct_functions.py:266-276: mean(resid^2 / sigma) -- sigma, not sigma^2."""
    model = zeta * (S2 + np.sum(np.asarray(C)[:, None] * np.exp(-1.0 * t[None, :] / np.asarray(tau)[:, None]), axis=0))
    if dy is None:
        return np.mean(np.square(model - y))
    return np.mean(np.square(model - y) / dy)


def initial_guess(t, y, nParams, nSample=10):
    """fitting_Ct_functions.py:359-382: tau log-spaced strictly between <dt> and 2 t_max;
    C_k = |mean(first 10) - mean(last 10)| / K; S2 = mean(last 10) when free, else 1 - mean(C)."""
    K = nParams // 2
    free_S2 = (nParams % 2 == 1)
    tau = np.logspace(np.log10(np.mean(t[1:] - t[:-1])), np.log10(t[-1] * 2.0), K + 2)[1:-1]
    beg = np.mean(y[:nSample])
    end = np.mean(y[-nSample:])
    C = np.array([np.fabs(beg - end) / K] * K)
    S2 = end if free_S2 else 1.0 - np.mean(C)
    return C, tau, S2, free_S2


def conduct_curve_fitting(t, y, dy, nParams):
    """fitting_Ct_functions.py:306-345 with bReInitialise=True, including the reference's quirk
    that the quality checks run on the *initial guess* (C, S2) before the optimum is stored."""
    from scipy.optimize import curve_fit
    C0, tau0, S20, free_S2 = initial_guess(t, y, nParams)
    K = nParams // 2
    p0 = list(C0) + list(tau0) + ([S20] if free_S2 else [])
    ub = [1.0] * K + [t[-1] * 10] * K + ([1.0] if free_S2 else [])
    quality = [True, True, True]
    out = dict(nParams=nParams, p0=np.array(p0), free_S2=free_S2)
    try:
        popt, pcov = curve_fit(curvefit_exponential, t, y, sigma=dy, p0=p0, bounds=(0.0, ub))
    except Exception:
        quality[0] = False
        out.update(chiSq=np.inf, quality=quality, ok=False)
        return out
    with np.errstate(invalid='ignore'):
        dP = np.sqrt(np.diag(pcov))
    S2_chk = S20 if free_S2 else 1.0 - np.sum(C0)
    if np.any(dP > popt):
        quality[1] = False
    if S2_chk + np.sum(C0) > 1.0:
        quality[2] = False
    C = popt[:K]
    tau = popt[K:2 * K]
    S2 = popt[-1] if free_S2 else 1.0 - np.sum(C)
    dC = np.array(dP[:K], dtype=float)
    dtau = np.array(dP[K:2 * K], dtype=float)
    dS2 = dP[-1] if free_S2 else 0.0
    chi = calc_chiSq(t, y, dy, S2, C, tau)
    order = np.argsort(tau)
    out.update(ok=True, popt=popt, dP=dP, chiSq=chi, quality=quality,
               C=C[order], tau=tau[order], dC=dC[order], dtau=dtau[order], S2=S2, dS2=dS2)
    return out


def optimised_curve_fitting(t, y, dy, listDoG=(2, 3, 5, 7, 9), chiSqThreshold=0.5):
    """fitting_Ct_functions.py:278-304.  Returns (selected fit dict or None, list of all trials)."""
    trials = []
    best = None
    first = True
    for nP in listDoG:
        fit = conduct_curve_fitting(t, y, dy, nP)
        trials.append(fit)
        if first:
            if np.all(fit['quality']):
                best = fit
                first = False
            continue
        if not np.all(fit['quality']):
            break
        if fit['chiSq'] >= best['chiSq'] * chiSqThreshold:
            break
        best = fit
    return best, trials

# ----------------------------------------------------------------------------------------------
# spectral densities and relaxation: Jomega/Jomega.c, spectral_densities.py
# ----------------------------------------------------------------------------------------------

GAMMA = {'1H': 267.513e6, '13C': 67.262e6, '15N': -27.116e6}     # spectral_densities.py:50-67
CSA_DEFAULT = {'15N': -170e-6, '13C': -130e-6}                     # spectral_densities.py:39-48
TIME_FACT = {'ps': 1.0e-12, 'ns': 1.0e-9, 'us': 1.0e-6, 'ms': 1.0e-3, 's': 1.0}


def Jomega(x, y):
    """Jomega/Jomega.c:49-66: x / (x*x + y*y), elementwise with numpy broadcasting."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    return x / (x * x + y * y)


def do_Jsum(om, A_J, D_J):
    """spectral_densities.py:1961-1972: einsum('...j,jk', A_J, Jomega.outer(D_J, om))."""
    Dmat = Jomega(np.asarray(D_J)[:, None], np.asarray(om)[None, :])
    return np.einsum('...j,jk', A_J, Dmat)


def D_coefficients_symmtop(Dpar, Dperp):
    """spectral_densities.py:1874-1884."""
    return np.array([5 * Dperp + Dpar, 2 * Dperp + 4 * Dpar, 6 * Dperp])


def A_coefficients_symmtop(v, bProlate=True):
    """spectral_densities.py:1886-1906."""
    v = np.asarray(v)
    z2 = np.square(v[..., 2] if bProlate else v[..., 0])
    w = 1 - z2
    return np.stack((3.0 * (z2 * w), 0.75 * np.square(w), 0.25 * np.square(3 * z2 - 1)), axis=-1)


def symmtop_from_iso(Diso, aniso):
    """calculate-relaxations-from-Ct.py:621-622: returns (Dpar, Dperp)."""
    Dperp = 3. * Diso / (2 + aniso)
    return aniso * Dperp, Dperp


def J_direct_transform(om, consts, taus):
    """spectral_densities.py:2024-2033."""
    om = np.asarray(om)
    J = np.zeros(len(om))
    for c, t in zip(consts, taus):
        J = J + c * t / (1 + (t * om) ** 2.)
    return J


def J_combine_isotropic_exp_decayN(om, tau_iso, S2, consts, taus):
    """spectral_densities.py:2038-2050."""
    om = np.asarray(om)
    k = (1.0 / tau_iso) + (1.0 / np.array(taus, dtype=float))
    J = S2 * tau_iso / (1. + (om * tau_iso) ** 2.)
    for i in range(len(consts)):
        J = J + consts[i] * k[i] / (k[i] ** 2. + om ** 2.)
    return J


def J_combine_symmtop_exp_decayN(om, v, Dpar, Dperp, S2, consts, taus):
    """spectral_densities.py:2057-2077."""
    D_J = D_coefficients_symmtop(Dpar, Dperp)
    A_J = A_coefficients_symmtop(v, bProlate=(Dpar > Dperp))
    J = do_Jsum(om, S2 * A_J, D_J)
    for i in range(len(consts)):
        J = J + do_Jsum(om, consts[i] * A_J, D_J + 1. / taus[i])
    return J


def B0_from_Hz(Hz):
    """calculate-relaxations-from-Ct.py:567."""
    return 2.0 * np.pi * Hz / 267.513e6


def omega_set(B0, timeUnit='ps', nucX='15N'):
    """spectral_densities.py:1630-1645 then set_time_unit 1619-1625.  The object is built in 'ns'
    and then rescaled by time_fact(tu)/time_fact('ns'); that operation order is kept."""
    om = np.zeros(5)
    tf_ns = TIME_FACT['ns']
    om[3] = -1.0 * GAMMA['1H'] * B0 * tf_ns
    om[1] = -1.0 * GAMMA[nucX] * B0 * tf_ns
    om[2] = om[3] - om[1]
    om[4] = om[3] + om[1]
    om *= TIME_FACT[timeUnit] / tf_ns
    return om


def omega_set_new(fieldMHz, timeUnit='ps', nucA='15N', nucB='1H'):
    """spectral_densities.py:153-175 (angularFrequencies), field given as 1H frequency in MHz."""
    B0 = 2.0 * np.pi * fieldMHz / 267.513
    tf = TIME_FACT[timeUnit]
    om = np.zeros(5)
    om[1] = -1.0 * GAMMA[nucA] * B0 * tf
    om[3] = -1.0 * GAMMA[nucB] * B0 * tf
    om[2] = om[3] - om[1]
    om[4] = om[3] + om[1]
    return om, B0


def factor_DD(nucX='15N', rXH_nm=1.02e-1):
    """spectral_densities.py:1696 / :239."""
    return 0.10 * 1.1121216813552401e-82 * GAMMA['1H'] ** 2.0 * GAMMA[nucX] ** 2.0 * (rXH_nm * 1e-9) ** -6.0


def factor_DD_new(nucA='15N', nucB='1H', rAB_nm=1.02e-1):
    """spectral_densities.py:239 (new API): same formula with gamma_A^2 * gamma_B^2 in that order,
    which rounds 1 ulp differently from the old API's gamma_H^2 * gamma_X^2."""
    return 0.10 * 1.1121216813552401e-82 * GAMMA[nucA] ** 2.0 * GAMMA[nucB] ** 2.0 * (rAB_nm * 1e-9) ** -6.0


def factor_CSA(csa, B0, nucX='15N'):
    """spectral_densities.py:1699-1701 / :243."""
    return 2.0 / 15.0 * csa ** 2.0 * (GAMMA[nucX] * B0) ** 2


def relax_from_J(J, B0, csa, time_fact, nucX='15N'):
    """Old API, spectral_densities.py:1680-1737: NOE uses the per-vector R1."""
    J = np.asarray(J)
    fDD = factor_DD(nucX)
    fCSA = factor_CSA(csa, B0, nucX)
    J0, J1, J2, J3, J4 = (J[..., i] for i in range(5))
    R1 = time_fact * (fDD * (J2 + 3 * J1 + 6 * J4) + fCSA * J1)
    R2 = time_fact * (0.5 * fDD * (4 * J0 + J2 + 3 * J1 + 6 * J4 + 6 * J3) + 1.0 / 6.0 * fCSA * (4 * J0 + 3 * J1))
    NOE = 1.0 + time_fact * GAMMA['1H'] / (GAMMA[nucX] * R1) * fDD * (6 * J4 - J2)
    return R1, R2, NOE


def rho_from_J(J):
    """spectral_densities.py:1775-1786."""
    J = np.asarray(J)
    return J[..., 1] / J[..., 0]


def weighted_average_stdev(values, weights):
    """general_maths.py:100-110."""
    avg = np.average(values, axis=-1, weights=weights)
    return avg, np.sqrt(np.average((values - avg) ** 2.0, axis=-1, weights=weights))


def convert_LambertCylindricalHist_to_vecs(hist, edges):
    """spectral_densities.py:2334-2350: bin-centre unit vectors (phi-major flattening) + weights."""
    phis = 0.5 * (edges[0][:-1] + edges[0][1:])
    thetas = np.arccos(0.5 * (edges[1][:-1] + edges[1][1:]))
    P, T = np.meshgrid(phis, thetas, indexing='ij')
    bv = np.stack((np.cos(P) * np.sin(T), np.sin(P) * np.sin(T), np.cos(T)), axis=-1)
    nP = hist.shape[1] * hist.shape[2]
    return np.repeat(bv.reshape(nP, 3)[None], hist.shape[0], axis=0), np.reshape(hist, (hist.shape[0], nP))


def obtain_R1R2NOErho(model, D, B0, S2, consts, taus, vecXH=None, weights=None, csa=None,
                      timeUnit='ps', nucX='15N', cast32=True):
    """calculate-relaxations-from-Ct.py:125-191 (old API).
    model 'direct_transform' | 'rigid_sphere' (D = Diso) | 'rigid_symmtop' (D = (Dpar, Dperp)).
    S2/consts/taus are per-residue lists (already zeta-scaled, :747-750).  Returns the datablock
    (4, n) or (4, n, 2); cast to float32 like the reference unless cast32=False."""
    n = len(S2)
    tf = TIME_FACT[timeUnit]
    om = omega_set(B0, timeUnit, nucX)
    if csa is None:
        csa = np.repeat(CSA_DEFAULT[nucX], n)
    csa = np.broadcast_to(np.asarray(csa, dtype=float), (n,))
    dt = np.float32 if cast32 else np.float64
    if model in ('direct_transform', 'rigid_sphere') or (vecXH is not None and np.ndim(vecXH) == 2):
        blk = np.zeros((4, n), dtype=dt)
        for i in range(n):
            if model == 'direct_transform':
                J = J_direct_transform(om, consts[i], taus[i])
            elif model == 'rigid_sphere':
                J = J_combine_isotropic_exp_decayN(om, 1.0 / (6.0 * D), S2[i], consts[i], taus[i])
            else:
                J = J_combine_symmtop_exp_decayN(om, vecXH[i], D[0], D[1], S2[i], consts[i], taus[i])
            R1, R2, NOE = relax_from_J(J, B0, csa[i], tf, nucX)
            blk[:, i] = [R1, R2, NOE, rho_from_J(J)]
        return blk
    blk = np.zeros((4, n, 2), dtype=dt)
    for i in range(n):
        J = J_combine_symmtop_exp_decayN(om, vecXH[i], D[0], D[1], S2[i], consts[i], taus[i])
        R1, R2, NOE = relax_from_J(J, B0, csa[i], tf, nucX)
        rho = rho_from_J(J)
        for k, arr in enumerate((R1, R2, NOE, rho)):
            if weights is None:
                blk[k, i] = [np.mean(arr), np.std(arr)]
            else:
                blk[k, i] = weighted_average_stdev(arr, weights[i])
    return blk


def new_api_eval(kind, fieldMHz, Diso, aniso, S2, consts, taus, zeta, binvecs, weights, csa,
                 timeUnit='ps', nucA='15N', nucB='1H'):
    """New class API, spectral_densities.py:820-907 with an axisymmetric model holding a vector
    distribution (463-603): R1/R2 as the old formulas; NOE uses R1 first *averaged over vectors*
    (881-892).  binvecs (B,3), weights (n,B), csa scalar or (n,).  zeta multiplies S2 and C inside
    calc_Jomega_one (552-557).  Returns values (n,), errors (n,)."""
    n = len(S2)
    om, B0 = omega_set_new(fieldMHz, timeUnit, nucA, nucB)
    tf = TIME_FACT[timeUnit]
    Dpar, Dperp = symmtop_from_iso(Diso, aniso)
    D_J = D_coefficients_symmtop(Dpar, Dperp)
    A_J = A_coefficients_symmtop(binvecs, bProlate=(aniso > 1))
    gA = GAMMA[nucA]
    gB = GAMMA[nucB]
    fDD = factor_DD_new(nucA, nucB)
    csa = np.broadcast_to(np.asarray(csa, dtype=float), (n,))
    vals = np.zeros(n)
    errs = np.zeros(n)
    for i in range(n):
        J = do_Jsum(om, zeta * S2[i] * A_J, D_J)
        for j in range(len(consts[i])):
            J = J + do_Jsum(om, zeta * consts[i][j] * A_J, D_J + 1. / taus[i][j])
        fCSA = 2.0 / 15.0 * csa[i] ** 2.0 * (gA * B0) ** 2
        J0, J1, J2, J3, J4 = (J[..., k] for k in range(5))
        R1 = tf * (fDD * (J2 + 3 * J1 + 6 * J4) + fCSA * J1)
        if kind == 'R1':
            x = R1
        elif kind == 'R2':
            x = tf * (0.5 * fDD * (4 * J0 + J2 + 3 * J1 + 6 * J4 + 6 * J3) + 1.0 / 6.0 * fCSA * (4 * J0 + 3 * J1))
        elif kind == 'NOE':
            R1m = np.average(R1, weights=weights[i])
            x = 1.0 + tf * gB / (gA * R1m) * fDD * (6 * J4 - J2)
        else:
            raise ValueError(kind)
        v = np.average(x, weights=weights[i])
        vals[i] = v
        errs[i] = np.sqrt(np.average((x - v) ** 2.0, weights=weights[i]))
    return vals, errs


# ----------------------------------------------------------------------------------------------
# global rotational diffusion: calculate-dq-distribution.py
# ----------------------------------------------------------------------------------------------

def quat_mult_simd(q1, q2):
    """transforms3d_supplement.py:163-183."""
    q1 = np.asarray(q1, dtype=np.float64)
    q2 = np.asarray(q2, dtype=np.float64)
    out = np.zeros_like(q1)
    out[..., 0] = q1[..., 0] * q2[..., 0] - np.einsum('...i,...i', q1[..., 1:4], q2[..., 1:4])
    out[..., 1:4] = q1[..., 0, None] * q2[..., 1:4] + q2[..., 0, None] * q1[..., 1:4] + np.cross(q1[..., 1:4], q2[..., 1:4])
    return out


def obtain_self_dq(q, delta):
    """calculate-dq-distribution.py:102-109: quat_reduce_simd(quat_mult_simd(quat_invert(q[:-delta]), q[delta:]))."""
    q = np.asarray(q)
    d = quat_mult_simd(q[:-delta] * [1.0, -1.0, -1.0, -1.0], q[delta:])          # quat_invert :185-186
    sgn = np.sign(d[..., 0])                                                     # quat_reduce_simd :219-227, qref = (1,0,0,0)
    sgn[sgn == 0] = 1.0
    return d * sgn[:, None]


def average_LegendreP1quat(vq):
    """calculate-dq-distribution.py:111-112 AS WRITTEN: apply_along_axis(..., axis=0) hands LegendreP1_quat one COLUMN
    (all samples of one component), so the value is mean over the three components of 1 - 2 sum_i v_ic^2, i.e.
    1 - (2/3) sum_i |v_i|^2 -- not the sample mean of 1 - 2|v|^2.  Restated as the reference computes it."""
    return np.mean([1.0 - 2.0 * np.sum(np.square(vq[:, c])) for c in range(3)])


def average_anisotropic_tensor(vq):
    """calculate-dq-distribution.py:118-126 without a frame rotation: mean of the outer products."""
    return np.mean(np.einsum('ij,ik->ijk', vq, vq), axis=0)


def dq_chunk_ranges(ndat, nchunk):
    """calculate-dq-distribution.py:128-144: nblock = ceil(ndat/nchunk); chunk i = [nblock i, min(ndat, nblock (i+1)))."""
    nblock = int(np.ceil(1.0 * ndat / nchunk))
    return [(nblock * i, min(ndat, nblock * (i + 1))) for i in range(nchunk)]


def dq_moments(q, lags, nchunk=1):
    """(nlags, nchunk, 7): sums of xx yy zz xy xz yz of the vector part of dq over each chunk, and the sample count --
    the quantity libspinrelax_hip's sr_dq_moments_f32 produces."""
    q = np.asarray(q)
    out = np.zeros((len(lags), nchunk, 7))
    for k, d in enumerate(lags):
        v = obtain_self_dq(q, int(d))[:, 1:4]
        for c, (a, b) in enumerate(dq_chunk_ranges(v.shape[0], nchunk)):
            w = v[a:b]
            if w.shape[0] == 0:
                continue
            out[k, c, :6] = [np.sum(w[:, 0] ** 2), np.sum(w[:, 1] ** 2), np.sum(w[:, 2] ** 2), np.sum(w[:, 0] * w[:, 1]),
                             np.sum(w[:, 0] * w[:, 2]), np.sum(w[:, 1] * w[:, 2])]
            out[k, c, 6] = w.shape[0]
    return out


# ----------------------------------------------------------------------------------------------
# trajectory front end: calculate-Ct-from-traj.py:64-86, 466-467
# ----------------------------------------------------------------------------------------------

def obtain_XHvecs(xyz, indexX, indexH):
    """calculate-Ct-from-traj.py:82-84 on a coordinate array: take(H) - take(X), vecnorm_NDarray(axis=2); the dtype of
    xyz (float32 for MDTraj) is kept, exactly as numpy does for the reference."""
    v = np.take(xyz, indexH, axis=1) - np.take(xyz, indexX, axis=1)
    with np.errstate(divide='ignore', invalid='ignore'):
        return vecnorm_NDarray(v, axis=2)


def kabsch_rotations(xyz, ref_xyz, fit_indices):
    """Per-frame optimal proper rotation (3 x 3, float64) that superposes the fit atoms of every frame onto the reference
    about their centroids: SVD form of Kabsch's algorithm -- deliberately a different algorithm from the device's
    closed-form quaternion (Horn).  This is what MDTraj's superpose(ref, frame=0, atom_indices=...) applies
    (calculate-Ct-from-traj.py:467; MDTraj is absent from the image, so this step is pinned by the mathematics only)."""
    P = np.asarray(xyz, dtype=np.float64)[:, fit_indices]
    Q = np.asarray(ref_xyz, dtype=np.float64)[fit_indices]
    P = P - P.mean(axis=1, keepdims=True)
    Q = Q - Q.mean(axis=0)
    H = np.einsum('nia,ib->nab', P, Q)                       # sum_i p_i q_i^T
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(np.einsum('nab,nbc->nac', U, Vt)))
    D = np.zeros_like(H)
    D[:, 0, 0] = 1.0
    D[:, 1, 1] = 1.0
    D[:, 2, 2] = d
    # R = V D U^T maps p onto q
    return np.einsum('nba,nbc,ndc->nad', Vt, D, U)


def superposed_XHvecs(xyz, ref_xyz, fit_indices, indexX, indexH):
    """Unit bond vectors after the superposition, float64: R_n (x_H - x_X) normalised."""
    R = kabsch_rotations(xyz, ref_xyz, fit_indices)
    d = (np.take(xyz, indexH, axis=1) - np.take(xyz, indexX, axis=1)).astype(np.float64)
    r = np.einsum('nab,nvb->nva', R, d)
    return r / np.linalg.norm(r, axis=2)[..., None], R
