"""
Global rotational diffusion from an orientation trajectory -- host side of SURVEY.md section 8(f)-2, mirroring the
functions of the reference's calculate-dq-distribution.py (same names, argument meaning, messages and file formats).

The per-lag reductions (difference quaternions dq_i = q_i^-1 q_{i+delta}, their isotropic decay and second-moment tensor,
for the whole trajectory and for every sub-chunk; reference :102-144 inside the loop :554-609) run on the GPU in ONE
launch for all lags: `sr_dq_moments_f32` (csrc/sr_dq.hip) returns the six second moments of the vector part of dq per
(lag, chunk).  Everything the reference derives from the samples is a function of those moments:

    <v (x) v>                      = M / n                                          (average_anisotropic_tensor :118-126)
    in the frame q_frame           = R M R^T / n   (the reference rotates every sample, then averages: same tensor)
    average_LegendreP1quat AS WRITTEN (:111-112; np.apply_along_axis over axis 0 hands the function a whole column)
                                   = mean_c(1 - 2 sum_i v_ic^2) = 1 - (2/3) tr M   -- NOT <1 - 2|v|^2>; reproduced.

The 3 x 3 eigen-decompositions, the frame quaternion and the one-parameter Powell fits (scipy.optimize.fmin_powell, as in
the reference) stay on the host.  There is no CPU fallback for the reductions.
"""
import math

import numpy as np
from scipy.optimize import fmin_powell

from . import general_scripts as gs
from . import quaternions as qops


def _ctx(ctx):
    from . import hip
    return ctx if ctx is not None else hip.default_context()


# ---- anisotropy measures of a diagonalised diffusion tensor (reference :31-96) ----------------------------------------
def aniso(D):
    return 2 * D[2] / (D[1] + D[0])


def rhomb(D):
    return 3 * (D[1] - D[0]) / (2 * D[2] - D[1] - D[0])


def calculate_aniso_nosort(D):
    """(Dx <= Dy <= Dz) -> (iso, ani and rhombicity taking z as unique axis, the same taking x)."""
    return (np.mean(D), aniso(D), rhomb(D), aniso(D[::-1]), rhomb(D[::-1]))


def calculate_anisotropies(D, chunkD=[]):
    """Without chunkD: the five measures of sorted D.  With it: [(value, std over the chunks), ...] where the chunks
    are put in D's rank order (reference :72-96)."""
    if len(chunkD) == 0:
        return calculate_aniso_nosort(np.sort(D))
    order = np.argsort(D)
    val = calculate_aniso_nosort(D[order])
    samples = np.array([calculate_aniso_nosort(x[order]) for x in chunkD])
    errors = np.std(samples, axis=0)
    return [(val[i], errors[i]) for i in range(len(val))]


# ---- device reductions -------------------------------------------------------------------------------------------
def dq_moments(q, lags, nchunk=1, ctx=None):
    """(nlags, nchunk, 7) float64 from the GPU: sums of xx yy zz xy xz yz of vec(q_i^-1 q_{i+lag}) and the count."""
    q = np.asarray(q)
    return _ctx(ctx).dq_moments(np.ascontiguousarray(q, dtype=np.float64 if q.dtype == np.float64 else np.float32), lags, nchunk)


def moments_to_tensor(m):
    """(..., 7) moments -> (..., 3, 3) mean outer product."""
    m = np.asarray(m)
    t = np.empty(m.shape[:-1] + (3, 3))
    n = m[..., 6]
    for (i, j), k in (((0, 0), 0), ((1, 1), 1), ((2, 2), 2), ((0, 1), 3), ((0, 2), 4), ((1, 2), 5)):
        t[..., i, j] = m[..., k] / n
        t[..., j, i] = t[..., i, j]
    return t


def average_LegendreP1quat(m):
    """The reference's average_LegendreP1quat(ndat, vq) expressed through the moments of vq (see module docstring for
    what that function really averages)."""
    m = np.asarray(m)
    return np.mean(np.stack([1 - 2 * m[..., 0], 1 - 2 * m[..., 1], 1 - 2 * m[..., 2]], axis=0), axis=0)


def average_anisotropic_tensor(m, qframe=(1, 0, 0, 0)):
    """average_anisotropic_tensor(ndat, vq, qframe) from the moments of vq."""
    t = moments_to_tensor(m)
    if not qops.nearly_equivalent(qframe, (1, 0, 0, 0)):
        R = qops.rotation_matrix(qframe)
        t = R @ t @ R.T
    return t


# ---- decay models and fits (reference :146-208) --------------------------------------------------------------------
def isotropic_decay(x, a):
    return 1.5 * np.exp(-x / a) - 0.5


def anisotropic_decay_noc(x, a):
    return 0.5 * np.exp(-x / a) + 0.5


def powell_expdecay(pos, *args):
    """mean squared deviation of y = C0 exp(-x/A) + C1; accumulated point by point like the reference's loop so that
    fmin_powell sees bit-identical objective values and takes the same path."""
    x, y, C0, C1 = args
    A = pos
    chi2 = 0.0
    nval = len(x)
    for i in range(nval):
        ymodel = C0 * math.exp(-x[i] / A) + C1
        chi2 += (ymodel - y[i]) ** 2
    return chi2 / nval


def obtain_exponential_guess(x, y, C1):
    return (x[0] - x[1]) / math.log((y[1] - C1) / (y[0] - C1))


def conduct_exponential_fit(xlist, ylist, C0, C1):
    print('= = Begin exponential fit.')
    guess = obtain_exponential_guess([xlist[0], xlist[1]], [ylist[0], ylist[1]], C1)
    print('= = = guessed initial tau: ', guess)
    fitOut = fmin_powell(powell_expdecay, guess, args=(xlist, ylist, C0, C1), full_output=True)
    print('= = = = Tau obtained: ', fitOut[0][0])
    return fitOut[0][0]


def get_flex_bounds(x, samples, nsig=1):
    """x with the asymmetric bounds that express the spread of the sub-chunk values around it."""
    mean = np.mean(samples)
    sig = np.std(samples)
    return [x, nsig * sig + x - mean, nsig * sig + mean - x]


# ---- output (reference :222-339, 393-403) --------------------------------------------------------------------------
def format_header(style_str, tau, taus=[]):
    l = []
    if style_str == 'iso':
        l.append('# model fit, tau = %e [ps]' % (tau))
        l.append("# Converted D_iso = %e [s^-1]" % (0.5e12 / tau))
        l.append("# t cos(th) P2[cos(th)] cos(th/2) th")
    elif style_str == 'iso_err':
        b = get_flex_bounds(tau, taus)
        l.append('# model fit, tau = %e +- %e %e [ps]' % (b[0], b[1], b[2]))
        Dval = 0.5e12 / tau
        Dvals = [0.5e12 / taus[i] for i in range(len(taus))]
        b = get_flex_bounds(Dval, Dvals)
        l.append('# Converted D_iso = %e +- %e %e [s^-1]' % (b[0], b[1], b[2]))
        for i in range(len(taus)):
            l.append('# Chunk_%d D_iso = %e [s^-1]' % (i, Dvals[i]))
        l.append("# t cos(th) P2[cos(th)] cos(th/2) th")
    elif style_str == 'aniso':
        Dval = 0.5e12 / tau
        for i in range(3):
            l.append("# model fit, e_%i tau = %e [ps]" % (i, tau[i]))
            l.append("# Converted D_%i = %e [s^-1]" % (i, Dval[i]))
        anis = calculate_anisotropies(Dval)
        l.append("# Converted Diso = %e [s^-1]" % (anis[0]))
        for name, v in zip(('Dani_L', 'Drho_L', 'Dani_S', 'Drho_S'), anis[1:]):
            l.append("# Converted %s = %f" % (name, v))
        l.append("# t <1-2x^2> <1-2y^2> <1-2z^2>")
    elif style_str == 'aniso_err':
        Dval = 0.5e12 / tau
        Dvals = 0.5e12 / taus
        for i in range(3):
            b = get_flex_bounds(tau[i], taus[:, i])
            l.append('# model fit, e_%i tau = %e +- %e %e [ps]' % (i, b[0], b[1], b[2]))
            b = get_flex_bounds(Dval[i], Dvals[:, i])
            l.append('# Converted D_%i = %e +- %e %e [s^-1]' % (i, b[0], b[1], b[2]))
        anis = calculate_anisotropies(Dval, Dvals)
        l.append("# Converted Diso = %e +- %e [s^-1]" % anis[0])
        for name, v in zip(('Dani_L', 'Drho_L', 'Dani_S', 'Drho_S'), anis[1:]):
            l.append("# Converted %s = %f +- %f" % ((name,) + tuple(v)))
        for j in range(len(taus)):
            for i in range(3):
                l.append('# Chunk_%d D_%d = %e [s^-1]' % (j, i, Dvals[j, i]))
        l.append("# t <1-2x^2> <1-2y^2> <1-2z^2>")
    return l


def format_header_quat(q):
    return '# Quaternion orientation frame: %f %f %f %f' % (q[0], q[1], q[2], q[3])


def print_model_fits_gen(fname, ydims, str_header, xlist, ylist):
    """xmgrace text: ydims 1 = one curve, 2 = several curves on one graph, 3 = several graphs of several curves."""
    ndat = len(xlist)
    with open(fname, 'w') as fp:
        for line in str_header:
            print("%s" % line, file=fp)
        if ydims == 1:
            for i in range(ndat):
                print("%g %g" % (xlist[i], ylist[i]), file=fp)
        elif ydims == 2:
            for s, curve in enumerate(ylist):
                print("@target g%d.s%d" % (0, s), file=fp)
                for i in range(ndat):
                    print("%g %g" % (xlist[i], curve[i]), file=fp)
                print("&", file=fp)
        elif ydims == 3:
            dim1 = len(ylist)
            print("dim1: ", dim1)
            for g in range(dim1):
                print("@g%d on" % g, file=fp)
            for g in range(dim1):
                print("dim2: ", len(ylist[g]))
                for s, curve in enumerate(ylist[g]):
                    print("@target g%d.s%d" % (g, s), file=fp)
                    for i in range(ndat):
                        print("%g %g" % (xlist[i], curve[i]), file=fp)
                    print("&", file=fp)
            print("@arrange(%i, %i, 0.1, 0.1, 0.1)" % (2, int(0.5 * dim1 + 0.5)), file=fp)
            for i in range(dim1):
                print("@with g%i" % i, file=fp)
                if i == 0:
                    print("@subtitle \"Aggregate Data\"", file=fp)
                print("@autoscale", file=fp)
        else:
            print("= = = Critical ERROR: invalid dimension specifier in print_model_fits_gen!")
            raise SystemExit(1)


def print_axes_as_xyz(fname, mats):
    with open(fname, 'w') as fp:
        for m in mats:
            print("3", file=fp)
            print("AXES", file=fp)
            for name, row in zip('XYZ', m):
                print("%s %g %g %g" % (name, row[0], row[1], row[2]), file=fp)


def rotmatrix_to_quaternion(time, matrix, bInvert=False):
    """gmx rotmat rows (9 numbers per frame) -> (5, nPts) block [time, w, x, y, z] (reference :406-423)."""
    nPts = len(time)
    if nPts != len(matrix):
        print("= = = ERROR in rotmatrix_to_quaternion: lengths are not the same!")
        return
    out = np.zeros((5, nPts))
    for i in range(nPts):
        out[0, i] = time[i]
        q = qops.mat2quat(matrix[i])
        out[1:5, i] = qops.qinverse(q) if bInvert else q
    return out


# ---- the analysis of calculate-dq-distribution.py's main (:539-735) -------------------------------------------------
def frame_intervals(data_delta_t, min_dt, max_dt, skip_dt):
    """(min_int, max_int, skip_int) in frames, with the reference's integer conversions (:541-549)."""
    skip_int = max(1, int(skip_dt / data_delta_t))
    min_int = max(skip_int, int(min_dt / data_delta_t))
    max_int = int(max_dt / data_delta_t)
    return min_int, max_int, skip_int


def analyse(data, min_dt=0.0, max_dt=1000.0, skip_dt=0.0, num_chunk=0, bDoIso=True, bDoAniso=True, bDoFullTensor=False,
            ctx=None):
    """data: (>= 5, ndat) block [time, q.w, q.x, q.y, q.z, ...] as the PLUMED reader returns it (float32).  Returns a dict
    with the per-lag lists the reference's main loop fills (out_dtlist, out_isolist, out_aniso1list, out_aniso2list,
    out_qlist, out_moilist, q_frame, chunk_isolist, chunk_aniso2list, out_RT)."""
    ndat = data.shape[1]
    data_delta_t = data[0, 1] - data[0, 0]
    min_int, max_int, skip_int = frame_intervals(data_delta_t, min_dt, max_dt, skip_dt)
    num_int = int(np.floor((max_int - min_int) / skip_int) + 1)
    min_delta_t = min_int * data_delta_t
    max_delta_t = max_int * data_delta_t
    print("= = Will calculate statistics for %i intervals between %g - %g ps, every %g ps) = =" % (num_int, min_delta_t, max_delta_t, skip_dt))
    print("= = ...corresponding to %i - %i frames, every %i frames. = =" % (min_int, max_int, skip_int))
    if max_delta_t > (data[0, -1] - data[0, 0]) / 2.0:
        print("= = = ERROR: max_dt requested (%g ps) is greater than half of the entire trajectory (%g ps)!" % (max_delta_t, (data[0, -1] - data[0, 0]) / 2.0))
        print("             ...will refuse to calculate correlation.")
        raise SystemExit(1)
    bDoSubchunk = num_chunk > 1
    lags = list(range(min_int, max_int + 1, skip_int))
    tot_int = len(lags)
    # float32 from the PLUMED reader (plumedcolvario.py:14-15), float64 from the gmx-rotmat route: each keeps its precision
    q32 = np.ascontiguousarray(data[1:5].T, dtype=np.float64 if data.dtype == np.float64 else np.float32)
    nch = num_chunk if bDoSubchunk else 1
    mom = dq_moments(q32, lags, nch, ctx=ctx)                  # ONE launch: (tot_int, nch, 7)
    total = mom.sum(axis=1)

    res = dict(out_dtlist=np.zeros(tot_int), out_isolist=np.zeros(tot_int), out_aniso1list=np.zeros((3, tot_int)),
               out_aniso2list=np.zeros((3, tot_int)), out_qlist=np.zeros((4, tot_int)), out_moilist=np.zeros((tot_int, 3, 3)),
               q_frame=(1, 0, 0, 0), lags=np.array(lags), moments=mom)
    if bDoFullTensor:
        res['out_RT'] = np.zeros((tot_int, 3, 3))
    if bDoSubchunk:
        res['chunk_isolist'] = np.zeros((num_chunk, tot_int))
        res['chunk_aniso2list'] = np.zeros((num_chunk, 3, tot_int))
    q_frame = (1, 0, 0, 0)
    bFirst = True
    for index, delta in enumerate(lags):
        res['out_dtlist'][index] = delta * data_delta_t
        moi = average_anisotropic_tensor(total[index])
        moiR1 = average_anisotropic_tensor(total[index], q_frame)
        print(" = = %i of %i intervals summed." % ((delta - min_int) / skip_int + 1, tot_int))
        if bDoIso:
            res['out_isolist'][index] = average_LegendreP1quat(total[index])
        if bDoAniso:
            eigval, eigvec = np.linalg.eigh(moi)
            moi_axes = eigvec.T
            q_rot = qops.quat_frame_transform_min(moi_axes)
            if bFirst:
                bFirst = False
                q_frame = q_rot
                moiR1 = average_anisotropic_tensor(total[index], q_frame)
                print("= = = PAF Axes in REF frame:")
                print(moi_axes[0], moi_axes[1], moi_axes[2])
                print("= = = Eigenvalues in REF frame and PAF frame:")
                print(eigval)
                print("= = = FRAME rotation from REF frame to PAF frame.")
                print(q_rot, moi_axes[0], moi_axes[1], moi_axes[2])
            res['out_aniso1list'][:, index] = 1 - 2 * eigval
            res['out_aniso2list'][:, index] = [1 - 2 * moiR1[0, 0], 1 - 2 * moiR1[1, 1], 1 - 2 * moiR1[2, 2]]
            res['out_qlist'][:, index] = q_rot
            res['out_moilist'][index] = moi_axes
        if bDoFullTensor:
            res['out_RT'][index] = moiR1
        if bDoSubchunk:
            res['chunk_isolist'][:, index] = average_LegendreP1quat(mom[index])
            t = average_anisotropic_tensor(mom[index], q_frame)
            res['chunk_aniso2list'][:, :, index] = np.stack([1 - 2 * t[:, 0, 0], 1 - 2 * t[:, 1, 1], 1 - 2 * t[:, 2, 2]], axis=1)
    res['q_frame'] = q_frame
    return res


def fit_and_write(res, out_pref, num_chunk=0, bDoIso=True, bDoAniso=True, bDoFullTensor=False):
    """The fitting + output half of the reference's main (:661-725): <pref>-iso.dat, -aniso2.dat, -aniso_q.dat, -moi.xyz,
    -tensor.dat.  Returns the fitted decay times."""
    bDoSubchunk = num_chunk > 1
    dt = res['out_dtlist']
    fitted = {}
    if bDoIso:
        tau = conduct_exponential_fit(dt, res['out_isolist'], 1.5, -0.5)
        model = isotropic_decay(dt, tau)
        fitted['iso_tau'] = tau
        if bDoSubchunk:
            chtaus = [conduct_exponential_fit(dt, res['chunk_isolist'][i], 1.5, -0.5) for i in range(num_chunk)]
            printlist = [[res['out_isolist'], model]]
            for i in range(num_chunk):
                printlist.append([res['chunk_isolist'][i], isotropic_decay(dt, chtaus[i])])
            print_model_fits_gen(out_pref + "-iso.dat", 3, format_header('iso_err', tau, chtaus), dt, printlist)
            fitted['iso_chunk_taus'] = np.array(chtaus)
        else:
            print_model_fits_gen(out_pref + "-iso.dat", 2, format_header('iso', tau), dt, [res['out_isolist'], model])
    if bDoAniso:
        print("= = = Running exponential fitting of fully anisotropic D...")
        a2 = res['out_aniso2list']
        taus = np.array([conduct_exponential_fit(dt, a2[i], 0.5, 0.5) for i in range(3)])
        models = anisotropic_decay_noc(dt, taus.reshape((3, 1)))
        fitted['aniso_taus'] = taus
        if bDoSubchunk:
            print("= = = Running exponential fitting over sub-chunks as well for uncertainty analysis...")
            chmodels = np.zeros((num_chunk, 3, len(dt)))
            chtaus = np.zeros((num_chunk, 3))
            for i in range(num_chunk):
                for j in range(3):
                    chtaus[i, j] = conduct_exponential_fit(dt, res['chunk_aniso2list'][i][j], 0.5, 0.5)
                chmodels[i] = anisotropic_decay_noc(dt, chtaus[i].reshape((3, 1)))
            header = format_header('aniso_err', taus, chtaus)
            header.append(format_header_quat(res['q_frame']))
            printlist = [np.concatenate((a2, models))]
            for i in range(num_chunk):
                printlist.append(np.concatenate((res['chunk_aniso2list'][i], chmodels[i])))
            print_model_fits_gen(out_pref + "-aniso2.dat", 3, header, dt, printlist)
            fitted['aniso_chunk_taus'] = chtaus
        else:
            header = format_header('aniso', taus)
            header.append(format_header_quat(res['q_frame']))
            print_model_fits_gen(out_pref + "-aniso2.dat", 2, header, dt, np.concatenate((a2, models)))
        gs.print_xylist(out_pref + "-aniso_q.dat", dt, res['out_qlist'], bCols=True)
        print_axes_as_xyz(out_pref + "-moi.xyz", res['out_moilist'])
    if bDoFullTensor:
        gs.print_xylist(out_pref + "-tensor.dat", dt, res['out_RT'].reshape(len(dt), 9).T)
    return fitted
