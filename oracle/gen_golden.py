#!/usr/bin/env python3
"""
gen_golden.py -- generate tests/golden/ fixtures by running the REAL reference (imported from
/root/reference by oracle/ref_loader.py) on seeded synthetic inputs.

Run in the build container only:   make -C oracle ref && python oracle/gen_golden.py

Fixtures are data (inputs / expected outputs of the reference's own functions and the files its own
writers produce); no reference source text is stored.  Inputs come from spinrelax_amd.synth (pure
integer-hash generator); their SHA-256 is stored so tests can prove they regenerated the same bytes.
Third-party versions in use when the fixtures were made are recorded in tests/golden/MANIFEST.json.
"""
import contextlib
import hashlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader                                   # noqa: E402
from spinrelax_amd import synth                     # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(GOLD, exist_ok=True)
ref = ref_loader.load()
manifest = {}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **arrs)
    manifest[name] = {k: list(np.shape(v)) for k, v in arrs.items()}
    print('wrote', name, os.path.getsize(path) // 1024, 'KiB')


# ------------------------------------------------------------------------------------------
# (i) C(t): calculate_Ct_Palmer on cfg1, cfg2 and a 16-vector slice of cfg3
# ------------------------------------------------------------------------------------------
def gen_ct(tag, cfg, nvec=None):
    s = synth.config_shapes(cfg)
    vecs = synth.synth_config(cfg, nvec=nvec)
    v4 = quiet(ref.calcCt.reformat_vecs_by_tau, [vecs], s['dt'], s['tau_memory'])
    assert v4.shape[:2] == (s['R'], s['F'])
    t = ref.calcCt.calculate_dt(s['dt'], s['tau_memory'])
    Ct32, dCt32 = quiet(ref.calcCt.calculate_Ct_Palmer, v4)
    Ct64, dCt64 = quiet(ref.calcCt.calculate_Ct_Palmer, v4.astype(np.float64))
    assert Ct32.dtype == np.float32 and Ct64.dtype == np.float64
    save('%s_ct.npz' % tag, input_sha=sha(vecs), cfg=cfg, nvec=vecs.shape[1], t=t,
         Ct64=Ct64, dCt64=dCt64, Ct32=Ct32, dCt32=dCt32)
    return s, vecs, v4, t, (Ct32, dCt32), (Ct64, dCt64)


# ------------------------------------------------------------------------------------------
# (ii) rotation + spherical histogram + mean vector + S2  (calculate-Ct-from-traj.py:535-646)
# ------------------------------------------------------------------------------------------
def ref_vec_stage(s, v4, q, nphi=72):
    sh = v4.shape
    v3 = v4.reshape((sh[0] * sh[1], sh[-2], sh[-1]))
    rot = ref.qs.rotate_vector_simd(v3, np.array(q))                         # :567
    avg = ref.gs.normalise_vector_array(np.mean(rot, axis=0))                # :581
    rtp = ref.gm.xyz_to_rtp(rot)                                             # :588
    rtp = np.transpose(rtp, axes=(1, 0, 2))                                  # :600
    rtp = np.delete(rtp, 0, axis=2)                                          # :611
    rtp[..., 1] = np.cos(rtp[..., 1])                                        # :613
    nB = rtp.shape[0]
    hist = np.zeros((nB, nphi, int(nphi / 2)), dtype=rtp.dtype)              # :615
    edges = None
    for i in range(nB):                                                      # :617-626 minus `normed`
        h, e = np.histogramdd(rtp[i], bins=(nphi, int(nphi / 2)), range=((-np.pi, np.pi), (-1, 1)))
        if edges is None:
            edges = e
        hist[i] = h
    S2 = quiet(ref.calcCt.calculate_S2_by_outerProduct, rot, s['dt'], s['tau_memory'])   # :641
    S2n = quiet(ref.calcCt.calculate_S2_by_outerProduct, rot)                              # :644
    return v3, rot, avg, hist, edges, S2, S2n


def gen_vec(tag, s, v4, q):
    v3, rot, avg, hist, edges, S2, S2n = ref_vec_stage(s, v4, q)
    assert rot.dtype == np.float64 and hist.dtype == np.float64
    rs = np.random.RandomState(7)
    idx_n = rs.randint(0, rot.shape[0], 256)
    idx_v = rs.randint(0, rot.shape[1], 256)
    save('%s_vec.npz' % tag, q=np.array(q), hist=hist.astype(np.uint32), hist_sum=hist.sum(),
         edges_phi=edges[0], edges_cos=edges[1], avgvec=avg, S2_tau=S2, S2_all=S2n,
         rot_idx_n=idx_n, rot_idx_v=idx_v, rot_sample=rot[idx_n, idx_v])
    return rot, avg, hist, edges, S2


# ------------------------------------------------------------------------------------------
# (iii) multi-exponential fits (calculate-fitted-Ct.py:149-180, fitting_Ct_functions.py:278-345)
# ------------------------------------------------------------------------------------------
PMAX = 9


def ref_fit_all(names, tlist, ylist, dylist, listDoG=(2, 3, 5, 7, 9)):
    nres = len(names)
    nord = len(listDoG)
    trial_chi = np.full((nres, nord), np.nan)
    trial_popt = np.full((nres, nord, PMAX), np.nan)
    trial_dP = np.full((nres, nord, PMAX), np.nan)
    trial_p0 = np.full((nres, nord, PMAX), np.nan)
    trial_q = np.zeros((nres, nord, 3), dtype=bool)
    sel_nP = np.zeros(nres, dtype=int)
    sel_chi = np.full(nres, np.nan)
    sel_S2 = np.full(nres, np.nan)
    sel_dS2 = np.full(nres, np.nan)
    sel_C = np.full((nres, 4), np.nan)
    sel_tau = np.full((nres, 4), np.nan)
    sel_dC = np.full((nres, 4), np.nan)
    sel_dtau = np.full((nres, 4), np.nan)
    ac = ref.fitCt.autoCorrelations()
    ac.import_target_array(keys=names, DeltaT=tlist, Decay=ylist, dDecay=dylist)
    for i, k in enumerate(ac.DeltaT.keys()):
        # per-order trials at full precision (same calls optimised_curve_fitting makes)
        for j, nP in enumerate(listDoG):
            m = ref.fitCt.autoCorrelationModel(name=k)
            m.set_nParams(nP)
            m.initialise_for_fit_advanced(ac.DeltaT[k], ac.Decay[k])
            trial_p0[i, j, :nP] = m.get_params_as_list()
            chi, qual = quiet(m.conduct_curve_fitting, ac.DeltaT[k], ac.Decay[k], ac.dDecay[k], bReInitialise=True)
            trial_chi[i, j] = chi
            trial_q[i, j] = qual
            if qual[0]:
                # conduct_curve_fitting sorts components afterwards; keep the sorted view
                trial_popt[i, j, :nP] = m.get_params_as_list()
                trial_dP[i, j, :nP] = m.get_uncertainties_as_list()
        obj = ac.add_model(k)
        quiet(obj.optimised_curve_fitting, ac.DeltaT[k], ac.Decay[k], ac.dDecay[k],
              listDoG=list(listDoG), chiSqThreshold=0.5)
        sel_nP[i] = obj.nParams
        sel_chi[i] = obj.chiSq
        sel_S2[i] = obj.S2
        sel_dS2[i] = obj.dS2
        K = obj.nComps
        sel_C[i, :K] = obj.C
        sel_tau[i, :K] = obj.tau
        sel_dC[i, :K] = obj.dC
        sel_dtau[i, :K] = obj.dtau
    res = dict(listDoG=np.array(listDoG), trial_chi=trial_chi, trial_popt=trial_popt, trial_dP=trial_dP,
               trial_p0=trial_p0, trial_quality=trial_q, sel_nParams=sel_nP, sel_chi=sel_chi, sel_S2=sel_S2,
               sel_dS2=sel_dS2, sel_C=sel_C, sel_tau=sel_tau, sel_dC=sel_dC, sel_dtau=sel_dtau)
    return ac, res


def gen_fit(tag, names, t, Ct, dCt, keep_text=False):
    """Goes through the reference's own text round trip (_Ctint.dat) like run-all.bash does."""
    fn = os.path.join(GOLD, '%s_Ctint.dat' % tag)
    ref.gs.print_sxylist(fn, names, t, np.stack((Ct.T, dCt.T), axis=-1))      # calculate-Ct-from-traj.py:531
    legs, tl, yl, dyl = ref.gs.load_sxydylist(fn, 'legend')                      # calculate-fitted-Ct.py:107
    legs = [int(x) for x in legs]
    ac, res = ref_fit_all(legs, tl, yl, dyl)
    fit_fn = os.path.join(GOLD, '%s_fittedCt.dat' % tag)
    ac.export(fileName=fit_fn, style='xmgrace')                                  # calculate-fitted-Ct.py:180
    save('%s_fit.npz' % tag, names=np.array(legs), t=tl, y=yl, dy=dyl, **res)
    if not keep_text:
        os.remove(fn)
        os.remove(fit_fn)
        return ac, res, None
    return ac, res, fit_fn


# ------------------------------------------------------------------------------------------
# (iv) J(omega), R1/R2/NOE/rho old API + new API (calculate-relaxations-from-Ct.py:541-1036)
# ------------------------------------------------------------------------------------------
FIELDS = (500.133, 600.133, 800.133)


def pad_params(S2, C, tau):
    n = len(S2)
    K = np.array([len(c) for c in C])
    Cp = np.zeros((n, 4))
    Tp = np.ones((n, 4))
    for i in range(n):
        Cp[i, :K[i]] = C[i]
        Tp[i, :K[i]] = tau[i]
    return np.array(S2, dtype=float), Cp, Tp, K


def gen_relax(tag, fit_fn, hist, edges, names):
    zeta = synth.ZETA
    npz_fn = os.path.join(GOLD, '%s_vecHistogram.npz' % tag)
    # calculate-Ct-from-traj.py:629 with the ragged `edges` wrapped as an object array (numpy>=1.24)
    np.savez_compressed(npz_fn, names=names, dataType='LambertCylindrical', bHistogram=True,
                        edges=np.array(edges, dtype=object), axisLabels=['phi', 'cos(theta)'], data=hist)
    autoCorrs = ref.fitCt.read_fittedCt_parameters(fit_fn)                     # :692
    S2_list, consts_list, taus_list, _ = autoCorrs.get_params_as_list()         # :747
    S2_raw, C_raw, T_raw, K = pad_params(S2_list, consts_list, taus_list)
    for i in range(autoCorrs.nModels):
        S2_list[i] *= zeta
        consts_list[i] *= zeta
    out = dict(names=np.array([int(k) for k in autoCorrs.model.keys()]), zeta=zeta, S2=S2_raw, C=C_raw, tau=T_raw, nComps=K,
               Diso=synth.DISO, Dani=synth.DANI, fields=np.array(FIELDS))
    resIDs, vecXH, weights = quiet(ref.calcRelax.read_vector_distribution_from_file, npz_fn)   # :644
    out['binvecs'] = vecXH[0]
    out['weights'] = weights
    n = autoCorrs.nModels
    csa_alt = -170e-6 + 1e-6 * np.linspace(-10, 10, n)
    out['csa_alt'] = csa_alt
    for fi, MHz in enumerate(FIELDS):
        B0 = 2.0 * np.pi * (MHz * 1e6) / 267.513e6                                # :567
        RObj = ref.sd.relaxationModel('NH', B0)                                   # :579
        RObj.set_time_unit('ps')                                                  # :580
        out['omega_%d' % fi] = RObj.omega.copy()
        # isotropic
        RObj.set_rotdif_model('rigid_sphere_D', synth.DISO)                       # :619
        csa = np.repeat(RObj.gX.csa, n)                                           # :705
        blk = quiet(ref.calcRelax._obtain_R1R2NOErho, RObj, n, S2_list, consts_list, taus_list, None,
                    weights=None, CSAvaluesArray=csa)                             # :763
        out['iso_f32_%d' % fi] = blk
        Jiso = np.array([ref.sd.J_combine_isotropic_exp_decayN(RObj.omega, 1.0 / (6.0 * RObj.rotdifModel.D),
                                                              S2_list[i], consts_list[i], taus_list[i]) for i in range(n)])
        out['iso_J_%d' % fi] = Jiso
        iso64 = np.zeros((4, n))
        for i in range(n):
            R1, R2, NOE = RObj.get_relax_from_J(Jiso[i], CSAvalue=csa[i])
            iso64[:, i] = [R1, R2, NOE, RObj.get_rho_from_J(Jiso[i])]
        out['iso_f64_%d' % fi] = iso64
        blkJ = quiet(ref.calcRelax._obtain_Jomega, RObj, n, S2_list, consts_list, taus_list, None)
        out['iso_Jblock_f32_%d' % fi] = blkJ
        # axisymmetric with the histogram distribution
        Dperp = 3. * synth.DISO / (2 + synth.DANI)                                # :621
        Dpar = synth.DANI * Dperp
        RObj.set_rotdif_model('rigid_symmtop_D', Dpar, Dperp)                     # :625
        for nm, cs in (('sym', csa), ('symcsa', csa_alt)):
            blk = quiet(ref.calcRelax._obtain_R1R2NOErho, RObj, n, S2_list, consts_list, taus_list, vecXH,
                        weights=weights, CSAvaluesArray=cs)
            out['%s_f32_%d' % (nm, fi)] = blk
            b64 = np.zeros((4, n, 2))
            for i in range(n):
                Jm = ref.sd.J_combine_symmtop_exp_decayN(RObj.omega, vecXH[i], Dpar, Dperp, S2_list[i], consts_list[i], taus_list[i])
                r1, r2, noe = RObj.get_relax_from_J_simd(Jm, CSAvalue=cs[i])
                rho = RObj.get_rho_from_J_simd(Jm)
                for k, a in enumerate((r1, r2, noe, rho)):
                    b64[k, i] = ref.gm.weighted_average_stdev(a, weights[i])
                if i == 0 and nm == 'sym':
                    out['sym_J_res0_%d' % fi] = Jm
            out['%s_f64_%d' % (nm, fi)] = b64
        # single (average) vector per site branch (:177-187): use the first bin-centre vectors
        one = vecXH[0][: n]
        blk1 = quiet(ref.calcRelax._obtain_R1R2NOErho, RObj, n, S2_list, consts_list, taus_list, one,
                     weights=None, CSAvaluesArray=csa)
        out['sym1_f32_%d' % fi] = blk1
        out['sym1_vecs'] = one
        if fi == 1:
            # byte-exact output files of the reference's writers (:1028-1036).  numpy>=2 raises in
            # general_scripts.py:235 (`dy==[]` on an ndarray), so dy is handed over as a list of the
            # same float32 scalars -- the printed text is identical.
            hdr = ref.calcRelax.print_fitting_params_headers(
                names=("Diso", "zeta", "CSA", "chi"),
                values=np.multiply((1.0, zeta, 1.0e6, 1.0), (synth.DISO, 1.0, RObj.gX.csa, 0.0)),
                units=('ps^-1', 'a.u.', 'ppm', 'a.u.'), bFit=(False, False, False, False))
            sim_resid = [int(k) for k in autoCorrs.model.keys()]
            b = out['sym_f32_%d' % fi]
            ref.gs.print_xydy(os.path.join(GOLD, '%s_sym_R1.dat' % tag), sim_resid, b[0, :, 0], list(b[0, :, 1]), header=hdr)
            ref.gs.print_xydy(os.path.join(GOLD, '%s_sym_rho.dat' % tag), sim_resid, b[3, :, 0], list(b[3, :, 1]))
            b = out['iso_f32_%d' % fi]
            ref.gs.print_xy(os.path.join(GOLD, '%s_iso_NOE.dat' % tag), sim_resid, b[2, :], header=hdr)
    # ---- new class API (spectral_densities.py:463-603, 820-907) ----
    localCt = ref.fitCt.read_fittedCt_parameters(fit_fn)
    grd = ref.sd.globalRotationalDiffusion_Axisymmetric(D=[synth.DISO, synth.DANI])
    quiet(grd.import_frame_vectors_npz, npz_fn)
    localCt.set_zeta(zeta)
    for fi, MHz in enumerate(FIELDS):
        w = ref.sd.angularFrequencies(nucleiA='15N', nucleiB='1H', fieldStrength=MHz, fieldUnit='MHz')
        out['new_omega_%d' % fi] = w.omega.copy()
        out['new_fDD'] = w.get_factor_DD()
        for kind, cls in (('R1', ref.sd.spinRelaxationR1), ('R2', ref.sd.spinRelaxationR2), ('NOE', ref.sd.spinRelaxationNOE)):
            sp = cls(kind, angFreq=w, globalRotDif=grd, localCtModels=localCt)
            sp.eval()
            out['new_%s_val_%d' % (kind, fi)] = np.array(sp.values)
            out['new_%s_err_%d' % (kind, fi)] = np.array(sp.errors)
    # isotropic new API
    gri = ref.sd.globalRotationalDiffusion_Isotropic(D=synth.DISO)
    w = ref.sd.angularFrequencies(nucleiA='15N', nucleiB='1H', fieldStrength=FIELDS[1], fieldUnit='MHz')
    for kind, cls in (('R1', ref.sd.spinRelaxationR1), ('R2', ref.sd.spinRelaxationR2), ('NOE', ref.sd.spinRelaxationNOE)):
        sp = cls(kind, angFreq=w, globalRotDif=gri, localCtModels=localCt)
        sp.eval()
        out['newiso_%s_val' % kind] = np.array(sp.values)
    save('%s_relax.npz' % tag, **out)
    return npz_fn, out


# ------------------------------------------------------------------------------------------
# (v) multi-field residue-specific CSA fit (spectral_densities.py:909-1447)
# ------------------------------------------------------------------------------------------
def gen_rscsa(tag, fit_fn, npz_fn, relax):
    zeta = synth.ZETA
    n = len(relax['names'])
    planted = -170e-6 + 1e-6 * 10.0 * (2.0 * ((np.arange(n) * 0.6180339887498949) % 1.0) - 1.0)
    localCt = ref.fitCt.read_fittedCt_parameters(fit_fn)
    grd = ref.sd.globalRotationalDiffusion_Axisymmetric(D=[synth.DISO, synth.DANI])
    quiet(grd.import_frame_vectors_npz, npz_fn)
    # synthetic "experiments": model values at the planted CSA, 2 % errors
    gen = ref.sd.spinRelaxationExperiments(grd, localCt)
    exp_files = []
    tmpdir = os.path.join(GOLD, '_tmp_expt')
    os.makedirs(tmpdir, exist_ok=True)
    expt_vals = []
    expt_errs = []
    expt_meta = []
    localCt.set_zeta(zeta)
    for MHz in FIELDS:
        w = ref.sd.angularFrequencies(nucleiA='15N', nucleiB='1H', fieldStrength=MHz, fieldUnit='MHz')
        w.initialise_CSA_array(n, planted)
        for kind, cls in (('R1', ref.sd.spinRelaxationR1), ('R2', ref.sd.spinRelaxationR2), ('NOE', ref.sd.spinRelaxationNOE)):
            sp = cls(kind, angFreq=w, globalRotDif=grd, localCtModels=localCt)
            sp.eval()
            vals = np.array(sp.values)
            errs = 0.02 * np.abs(vals)
            fn = os.path.join(tmpdir, 'expt_%s_%d.dat' % (kind, round(MHz)))
            with open(fn, 'w') as fp:
                print('# Type %s' % kind, file=fp)
                print('# NucleiA 15N', file=fp)
                print('# NucleiB 1H', file=fp)
                print('# Frequency %.3f' % MHz, file=fp)
                for nm, v, e in zip(localCt.get_names(), vals, errs):
                    print('%s %.12g %.12g' % (nm, v, e), file=fp)
            exp_files.append(fn)
            back = np.loadtxt(fn, comments='#')
            expt_vals.append(back[:, 1])
            expt_errs.append(back[:, 2])
            expt_meta.append((kind, MHz))
    # fresh objects for the optimisation, as calculate-relaxations-multi-field.py:109-215 builds them
    localCt2 = ref.fitCt.read_fittedCt_parameters(fit_fn)
    grd2 = ref.sd.globalRotationalDiffusion_Axisymmetric(D=[synth.DISO, synth.DANI])
    quiet(grd2.import_frame_vectors_npz, npz_fn)
    objExpts = ref.sd.spinRelaxationExperiments(grd2, localCt2)
    for f in exp_files:
        objExpts.add_experiment(f)
    objExpts.set_global_zeta(zeta)
    quiet(objExpts.map_experiment_peaknames_to_models)
    quiet(objExpts.parse_optimisation_params, ['rsCSA'])
    chisq = quiet(objExpts.perform_optimisation, maxCycles=10, tol=1e-6)
    fitted = np.array(objExpts.get_first_csa())
    vals_after = [np.array(sp.values) for sp in objExpts.spinrelax]
    errs_after = [np.array(sp.errors) for sp in objExpts.spinrelax]
    # one exported xvg as a format fixture (spectral_densities.py:1178-1194)
    quiet(objExpts.export_xvg, os.path.join(tmpdir, 'out'), bIncludeExpt=True)
    xvg = sorted(f for f in os.listdir(tmpdir) if f.endswith('.xvg'))
    os.replace(os.path.join(tmpdir, xvg[0]), os.path.join(GOLD, '%s_%s' % (tag, xvg[0])))
    manifest['%s_xvg_name' % tag] = xvg[0]
    # the objective at a grid of CSA values for residue 0, to pin the objective itself
    grid = -170e-6 + 1e-6 * np.linspace(-15, 15, 7)
    obj0 = np.array([ref.sd.optimisation_loop_rsCSA_inner_function([g], objExpts, 0, objExpts.mapExptCoverage[0]) for g in grid])
    objExpts.set_all_csa(fitted[0], ind=0)
    save('%s_rscsa.npz' % tag, planted=planted, fitted=fitted, chisq=chisq,
         expt_kind=np.array([m[0] for m in expt_meta]), expt_MHz=np.array([m[1] for m in expt_meta]),
         expt_vals=np.array(expt_vals), expt_errs=np.array(expt_errs),
         vals_after=np.array(vals_after), errs_after=np.array(errs_after), obj_grid=grid, obj_res0=obj0)
    for f in os.listdir(tmpdir):
        os.remove(os.path.join(tmpdir, f))
    os.rmdir(tmpdir)


# ------------------------------------------------------------------------------------------
# (vii) known answers
# ------------------------------------------------------------------------------------------
def gen_known():
    w = ref.sd.angularFrequencies()
    B0 = 2.0 * np.pi * 600.133e6 / 267.513e6
    RObj = ref.sd.relaxationModel('NH', B0)
    RObj.set_time_unit('ps')
    RObj.set_rotdif_model('rigid_sphere_D', 3.7383e-5)
    blk = quiet(ref.calcRelax._obtain_R1R2NOErho, RObj, 1, [0.890023], [[0.]], [[99999.]], [])
    J = ref.sd.J_combine_isotropic_exp_decayN(RObj.omega, 1.0 / (6.0 * 3.7383e-5), 0.890023, [0.], [99999.])
    R1, R2, NOE = RObj.get_relax_from_J(J)
    x = np.array([1e-5, 2.5e-4, 0.3, 7.0])
    y = np.array([0.0, 3.8e-4, 4.1e-3])
    known = dict(f_DD=float(w.get_factor_DD()), omega_600_133_ps=[float(v) for v in RObj.omega],
                 rigid_sphere=dict(Diso=3.7383e-5, S2=0.890023, MHz=600.133, R1=float(R1), R2=float(R2), NOE=float(NOE),
                                   theoretical_f32=[float(v) for v in blk[:, 0]]),
                 Jomega_outer=dict(x=x.tolist(), y=y.tolist(), out=ref.npufunc.Jomega.outer(x, y).tolist()),
                 Jomega_types=list(ref.npufunc.Jomega.types))
    with open(os.path.join(GOLD, 'known_answers.json'), 'w') as fp:
        json.dump(known, fp, indent=1)
    print('wrote known_answers.json')


def main():
    import scipy
    manifest['versions'] = dict(numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0])
    gen_known()
    # cfg1: the full chain on 32 residues
    s1, vecs1, v41, t1, (C32, dC32), (C64, dC64) = gen_ct('cfg1', 1)
    names1 = list(range(2, 2 + s1['V']))
    rot, avg, hist, edges, S2 = gen_vec('cfg1', s1, v41, synth.Q_EXT)
    ref.gs.print_xylist(os.path.join(GOLD, 'cfg1_avgvec.dat'), names1, np.array(avg).T, True)        # :582
    ref.gs.print_xylist(os.path.join(GOLD, 'cfg1_S2.dat'), names1, (S2.T) * (1.02 / 1.04) ** 6, True)  # :646
    # native-f32 text file exactly as the reference pipeline writes it
    ac, res, fit_fn = gen_fit('cfg1', names1, t1, C32, dC32, keep_text=True)
    npz_fn, relax = gen_relax('cfg1', fit_fn, hist, edges, names1)
    gen_rscsa('cfg1', fit_fn, npz_fn, relax)
    # float64 C(t) written through the same writer (the GPU path's native precision)
    fn64 = os.path.join(GOLD, 'cfg1_Ctint_f64.dat')
    ref.gs.print_sxylist(fn64, names1, t1, np.stack((C64.T, dC64.T), axis=-1))
    # cfg2: C(t), histogram with the README quaternion, fits for 16 residues at L=512
    s2, vecs2, v42, t2, (C32, dC32), (C64, dC64) = gen_ct('cfg2', 2)
    gen_vec('cfg2', s2, v42, synth.Q_EXT)
    gen_fit('cfg2', list(range(2, 18)), t2, C64[:, :16], dC64[:, :16])
    # cfg3 slice: 8 vectors, R=24, F=4096, L=2048
    s3, vecs3, v43, t3, (C32, dC32), (C64, dC64) = gen_ct('cfg3s', 3, nvec=8)
    gen_fit('cfg3s', list(range(2, 10)), t3, C64, dC64)
    with open(os.path.join(GOLD, 'MANIFEST.json'), 'w') as fp:
        json.dump(manifest, fp, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
