"""
Legacy single-field optimisation modes of calculate-relaxations-from-Ct.py:
`--opt Diso | DisoS2 | DisoCSA | DisoS2CSA | new` (calculate-relaxations-from-Ct.py:193-316, 775-1004).

scipy's Powell search runs on the host exactly as in the reference for the few-variable global searches; every objective
evaluation is ONE launch of the relaxation kernel over all residues and histogram bins
(`spectral_densities._obtain_R1R2NOErho`).  The per-residue CSA refinement of mode `new` (one fmin_powell per residue,
:935-1000) is ONE launch of the device-side search (`sr_legacy_csa_search_f64`: scipy's one-variable Powell restated, a
workgroup per residue, the objective over the residue's histogram bins with the relaxation kernel's own arithmetic) when the
model is the axisymmetric one with a vector histogram; `SR_LEGACY_HOST_SEARCH=1` (and every other case) keeps the host loop,
which the tests use as the cross-check.  Function names, argument tuples, printed lines and the float32 datablocks follow the
reference so that the optimiser sees the same objective.
"""
import os
import sys

import numpy as np
from scipy.optimize import fmin_powell

from . import spectral_densities as sd

_ctx = sd._ctx


def optfunc_R1R2NOE_inner(datablock, expblock):
    """:193-208"""
    if len(datablock.shape) == 3 and len(expblock.shape) == 3:
        chisq = np.square(datablock[..., 0] - expblock[..., 0])
        sigsq = np.square(datablock[..., 1]) + np.square(expblock[..., 1])
        return np.mean(chisq / sigsq)
    elif len(datablock.shape) == 3:
        chisq = np.square(datablock[..., 0] - expblock)
        sigsq = np.square(datablock[..., 1])
        return np.mean(chisq / sigsq)
    elif len(expblock.shape) == 3:
        chisq = np.square(datablock - expblock[..., 0])
        sigsq = np.square(expblock[..., 1])
        return np.mean(chisq / sigsq)
    return np.mean(np.square(datablock - expblock))


def optfunc_R1R2NOE_new(params, *args):
    """:210-258 -- single-residue CSA fitting; args = (RObj, S2, consts, taus, vecXH, w, expblock)."""
    RObj = args[0]
    S2, consts, taus = args[1], args[2], args[3]
    vecXH, w = args[4], args[5]
    expblock = args[-1]
    if len(expblock.shape) > 1:
        exp = expblock[:, 0]
        sigsq = np.square(expblock[:, 1])
    else:
        exp = expblock
        sigsq = np.zeros(3, dtype=np.float32)
    if RObj.rotdifModel.name != 'rigid_symmtop':
        # the reference indexes rotdifModel.D[0], D[1] here: only the symmetric top works
        raise IndexError('optfunc_R1R2NOE_new needs an axisymmetric diffusion model')
    csa = float(np.ravel(params)[0])
    fcsa = RObj.get_f_CSA(np.array([csa]))
    S2a, C, T, K = sd._pack([S2], [consts], [taus])
    D = [RObj.rotdifModel.D[0], RObj.rotdifModel.D[1]]
    vecXH = np.asarray(vecXH, dtype=float)
    common = (2, D, RObj.omega, RObj.get_f_DD(), fcsa[None, :], RObj.time_fact, RObj.gH.gamma / RObj.gX.gamma, S2a, C, T, K)
    if len(vecXH.shape) > 1:
        wts = None if w is None else np.asarray(w, dtype=float)[None, :]
        out, _ = _ctx(None).relax(*common, noe_mode=0, binvecs=vecXH, weights=wts)
        datablock = np.zeros((3, 2), dtype=np.float32)
        datablock[:] = out[0, 0, :3, :]
        chisq = np.square(datablock[:, 0] - exp)
        sigsq = sigsq + np.square(datablock[:, 1])
        return np.mean(chisq / sigsq)
    out, _ = _ctx(None).relax(*common, noe_mode=0, resvecs=vecXH[None, :])
    datablock = np.zeros(3, dtype=np.float32)
    datablock[:] = out[0, 0, :3, 0]
    chisq = np.square(out[0, 0, :3, 0] - exp)
    if sigsq[0] != 0.0:
        return np.mean(chisq / sigsq)
    return np.mean(chisq)


def optfunc_R1R2NOE_DisoS2CSA(params, *args):
    """:260-277 -- Diso, global S2 scaling and global CSA."""
    Diso, S2s, csa = params[0], params[1], params[2]
    RObj = args[0]
    nVecs, listS2, consts, taus = args[1], args[2], args[3], args[4]
    vecXH, w = args[5], args[6]
    expblock = args[-1]
    RObj.rotdifModel.change_Diso(Diso)
    S2loc = [S2s * k for k in listS2]
    consts_loc = [[S2s * k for k in j] for j in consts]
    RObj.gX.csa = csa
    datablock = sd._obtain_R1R2NOErho(RObj, nVecs, S2loc, consts_loc, taus, vecXH, weights=w)
    chisq = optfunc_R1R2NOE_inner(datablock[0:3, ...], expblock)
    print("= = optimisations params( %s ) returns chi^2 %g" % (params, chisq))
    return chisq


def optfunc_R1R2NOE_DisoCSA(params, *args):
    """:279-291"""
    Diso, csa = params[0], params[1]
    RObj = args[0]
    nVecs, S2, consts, taus = args[1], args[2], args[3], args[4]
    vecXH, w = args[5], args[6]
    expblock = args[-1]
    RObj.rotdifModel.change_Diso(Diso)
    RObj.gX.csa = csa
    datablock = sd._obtain_R1R2NOErho(RObj, nVecs, S2, consts, taus, vecXH, weights=w)
    chisq = optfunc_R1R2NOE_inner(datablock[0:3, ...], expblock)
    print("= = optimisations params( %s ) returns chi^2 %g" % (params, chisq))
    return chisq


def optfunc_R1R2NOE_DisoS2(params, *args):
    """:293-307"""
    Diso, S2s = params[0], params[1]
    RObj = args[0]
    nVecs, S2, consts, taus = args[1], args[2], args[3], args[4]
    vecXH, w, CSAarray = args[5], args[6], args[7]
    expblock = args[-1]
    RObj.rotdifModel.change_Diso(Diso)
    S2loc = [S2s * k for k in S2]
    consts_loc = [[S2s * k for k in j] for j in consts]
    datablock = sd._obtain_R1R2NOErho(RObj, nVecs, S2loc, consts_loc, taus, vecXH, weights=w, CSAvaluesArray=CSAarray)
    chisq = optfunc_R1R2NOE_inner(datablock[0:3, ...], expblock)
    print("= = optimisations params( %s ) returns chi^2 %g" % (params, chisq))
    return chisq


def optfunc_R1R2NOE_Diso(params, *args):
    """:309-320"""
    Diso = params[0]
    RObj = args[0]
    nVecs, S2, consts, taus = args[1], args[2], args[3], args[4]
    vecXH, w, CSAarray = args[5], args[6], args[7]
    expblock = args[-1]
    RObj.rotdifModel.change_Diso(Diso)
    datablock = sd._obtain_R1R2NOErho(RObj, nVecs, S2, consts, taus, vecXH, weights=w, CSAvaluesArray=CSAarray)
    chisq = optfunc_R1R2NOE_inner(datablock[0:3, ...], expblock)
    print("= = Optimisations params( %s ) returns Chi^2 %g" % (params, chisq))
    return chisq


def read_experiment(expt_data_file, rotdif_name, load_xys):
    """:776-803 -- 4- or 7-column experiment file -> (exp_resid, expblock (3, nres) or (3, nres, 2))."""
    exp_resid, expblock = load_xys(expt_data_file)
    nres = len(exp_resid)
    ny = expblock.shape[1]
    if ny == 6:
        expblock = expblock.reshape((nres, 3, 2))
        if np.any(expblock[..., 1] == 0):
            print("= = = WARNING: Experimental data %s contains entries with 0.00 uncertainty!" % expt_data_file, file=sys.stderr)
            if rotdif_name == 'rigid_sphere':
                print("= = = ERROR: Experimental data with partial zero uncertainties will break isotropic rotational diffusion optimisations.\n"
                      "      Please clean up your data with an appropriate uncertainty estimator.")
                sys.exit(1)
    elif ny != 3:
        print("= = = ERROR: The column format of the experimental relaxation file is not recognised!", file=sys.stderr)
        sys.exit(1)
    if ny == 3:
        expblock = expblock.T
    else:
        expblock = np.swapaxes(expblock, 0, 1)
    return exp_resid, expblock


def match_residues(sim_resid, exp_resid, expblock, S2_list, consts_list, taus_list, vecXH, vecXHweights, CSAvaluesArray):
    """:805-851 -- restrict simulation and experiment to the residues both have."""
    sim_resid = list(sim_resid)
    same = len(sim_resid) == len(exp_resid) and all(int(a) == int(b) for a, b in zip(sim_resid, exp_resid))
    if same:
        return None, sim_resid, S2_list, consts_list, taus_list, vecXH, vecXHweights, CSAvaluesArray, expblock
    print("= = WARNING: The resids between the simulation and experiment are not the same!", file=sys.stderr)
    print("...removing elements from the vector files that do not match.", file=sys.stderr)
    print("Debug (before):", len(S2_list), None if vecXH is None else vecXH.shape, expblock.shape)
    print("(resid - sim)", sim_resid)
    print("(resid - exp)", exp_resid)
    shared = np.sort(list(set(int(x) for x in sim_resid) & set(int(x) for x in exp_resid)))
    print("(resid - shared)", shared)
    if len(shared) == 0:
        print("= = ERROR: there is no overlap between experimental and simulation residue indices!", file=sys.stderr)
        sys.exit(1)
    sim_arr = np.array([int(x) for x in sim_resid])
    exp_arr = np.array([int(x) for x in exp_resid])
    sim_ind = np.array([np.where(sim_arr == x)[0][0] for x in shared])
    exp_ind = np.array([np.where(exp_arr == x)[0][0] for x in shared])
    fvec = None if vecXH is None else vecXH.take(sim_ind, axis=0)
    fw = None if vecXHweights is None else vecXHweights.take(sim_ind, axis=0)
    expblock = np.take(expblock, exp_ind, axis=1)
    print("Debug (after):", len(sim_ind), None if fvec is None else fvec.shape, expblock.shape)
    return (sim_ind, [sim_resid[x] for x in sim_ind], [S2_list[x] for x in sim_ind], [consts_list[x] for x in sim_ind],
            [taus_list[x] for x in sim_ind], fvec, fw, CSAvaluesArray.take(sim_ind), expblock)


def _device_csa_search_applies(relax_obj, fvecXH, fw, expblock):
    """the one-launch search covers the case run-all.bash produces: axisymmetric diffusion, one histogram of bin-centre vectors
    shared by every residue with per-residue weights, measured values with uncertainties"""
    if os.environ.get('SR_LEGACY_HOST_SEARCH'):
        return False
    if relax_obj.rotdifModel.name != 'rigid_symmtop' or fw is None or np.ndim(expblock) != 3:
        return False
    v = np.asarray(fvecXH, dtype=float)
    return v.ndim == 3 and v.shape[2] == 3 and bool(np.all(v == v[0]))


def csa_search_device(relax_obj, fS2, fconsts, ftaus, fvecXH, fw, expblock, csa0, ctx=None):
    """fmin_powell(optfunc_R1R2NOE_new, x0=csa0[i], ...) for every residue i (:210-258, :983-989) as one launch.
    Returns (csa (n), chi^2 (n), objective calls (n))."""
    S2a, C, T, K = sd._pack(fS2, fconsts, ftaus)
    gb = (relax_obj.gX.gamma * relax_obj.B_0) ** 2                       # get_f_CSA: 2.0 / 15.0 * csa ** 2.0 * (gamma B0) ** 2
    return _ctx(ctx).legacy_csa_search([relax_obj.rotdifModel.D[0], relax_obj.rotdifModel.D[1]], relax_obj.omega, relax_obj.get_f_DD(), gb,
                                       relax_obj.time_fact, relax_obj.gH.gamma / relax_obj.gX.gamma, S2a, C, T, K,
                                       np.asarray(fvecXH, dtype=float)[0], np.asarray(fw, dtype=float),
                                       np.ascontiguousarray(np.swapaxes(np.asarray(expblock, dtype=float), 0, 1)), np.asarray(csa0, dtype=float))


def run(optMode, relax_obj, Diso, matched, expblock_unused, nRefinementCycles, refinementTolerance, out_pref, header_fn,
        param_names, param_scaling, param_units, print_xy, sim_resid_all, CSAvaluesArray, S2_list):
    """The mode switch of :853-1002.  Returns (optHeader, CSAvaluesArray, S2_list)."""
    sim_ind, fsim_resid, fS2, fconsts, ftaus, fvecXH, fw, fCSAs, expblock = matched
    fnum = len(fS2)
    Diso_init = Diso
    if optMode == 'new':
        print("= = Conducting global-Diso + local-CSA refinement... this may take a while.")
        DisoOpt = Diso_init
        fCSAsOpt = np.copy(fCSAs)
        fCSAsChiSq = np.zeros(fnum, dtype=np.float32)
        DisoPrev = fCSAsPrev = None
        bFirst = True
        ChiSqDiso = np.nan
        r = 0
        for r in range(nRefinementCycles):
            out = fmin_powell(optfunc_R1R2NOE_Diso, x0=DisoOpt, direc=[0.1 * DisoOpt],
                              args=(relax_obj, fnum, fS2, fconsts, ftaus, fvecXH, fw, fCSAsOpt, expblock), full_output=True)
            DisoOpt, ChiSqDiso = out[0], out[1]
            if (not bFirst) and np.allclose(DisoOpt, DisoPrev, rtol=refinementTolerance):
                print("= = = BREAK at Diso test.")
                break
            DisoPrev = DisoOpt
            if _device_csa_search_applies(relax_obj, fvecXH, fw, expblock):
                csa_dev, chi_dev, _ = csa_search_device(relax_obj, fS2, fconsts, ftaus, fvecXH, fw, expblock, fCSAsOpt)
                fCSAsOpt[:] = csa_dev
                fCSAsChiSq[:] = chi_dev
            else:
                for i in range(fnum):
                    out = fmin_powell(optfunc_R1R2NOE_new, x0=fCSAsOpt[i],
                                      args=(relax_obj, fS2[i], fconsts[i], ftaus[i], fvecXH[i], None if fw is None else fw[i],
                                            expblock[:, i, :] if expblock.ndim == 3 else expblock[:, i]), full_output=True, disp=False)
                    fCSAsOpt[i], fCSAsChiSq[i] = np.ravel(out[0])[0], out[1]
            if (not bFirst) and np.allclose(fCSAsOpt, fCSAsPrev, rtol=refinementTolerance):
                print("= = = BREAK at CSA test")
                break
            fCSAsPrev = fCSAsOpt                     # (sic) the reference keeps a reference, not a copy
            if bFirst:
                bFirst = False
            print("    ...round %i complete." % r)
        print("    ....optimisation complete at round %i." % r)
        optHeader = header_fn(names=param_names,
                              values=np.multiply(param_scaling, (float(np.ravel(DisoOpt)[0]), 1.0, np.nan, np.sqrt(ChiSqDiso))),
                              units=param_units, bFit=(True, False, False, True))
        optHeader = optHeader + "\n# See %s_CSA_values.dat for individual CSA optimisations." % out_pref
        print(optHeader)
        if sim_ind is not None:
            for i, j in enumerate(sim_ind):
                CSAvaluesArray[j] = fCSAsOpt[i]
        else:
            CSAvaluesArray = fCSAsOpt
        print_xy(out_pref + '_CSA_values.dat', sim_resid_all, CSAvaluesArray)
    elif optMode == 'DisoS2CSA':
        print("= = Fitting both Diso, S2, as well as average CSA..")
        p_init = (Diso_init, 1.0, relax_obj.gX.csa)
        dmat = np.array([[np.sqrt(1.0 / 3.0), np.sqrt(1.0 / 3.0), np.sqrt(1.0 / 3.0)],
                         [-np.sqrt(2.0 / 3.0), np.sqrt(1.0 / 6.0), np.sqrt(1.0 / 6.0)],
                         [0, np.sqrt(1.0 / 2.0), -np.sqrt(1.0 / 2.0)]])
        d_init = np.multiply(0.1 * dmat, p_init)
        fminOut = fmin_powell(optfunc_R1R2NOE_DisoS2CSA, x0=p_init, direc=d_init,
                              args=(relax_obj, fnum, fS2, fconsts, ftaus, fvecXH, fw, expblock), full_output=True)
        print(fminOut)
        Diso_opt, S2s_opt, csa_opt, chisq = fminOut[0][0], fminOut[0][1], fminOut[0][2], fminOut[1]
        optHeader = header_fn(names=param_names, values=np.multiply(param_scaling, (Diso_opt, S2s_opt, csa_opt, np.sqrt(chisq))),
                              units=param_units, bFit=(True, True, True, True))
        print(optHeader)
    elif optMode == 'DisoCSA':
        print("= = Fitting both Diso and the average CSA..")
        p_init = (Diso_init, relax_obj.gX.csa)
        d_init = ((0.1 * p_init[0], 0.1 * p_init[1]), (0.1 * p_init[0], -0.1 * p_init[1]))
        fminOut = fmin_powell(optfunc_R1R2NOE_DisoCSA, x0=p_init, direc=d_init,
                              args=(relax_obj, fnum, fS2, fconsts, ftaus, fvecXH, fw, expblock), full_output=True)
        print(fminOut)
        Diso_opt, csa_opt, chisq = fminOut[0][0], fminOut[0][1], fminOut[1]
        optHeader = header_fn(names=param_names, values=np.multiply(param_scaling, (Diso_opt, 1.0, csa_opt, np.sqrt(chisq))),
                              units=param_units, bFit=(True, False, True, True))
        print(optHeader)
    elif optMode == 'DisoS2':
        print("= = Fitting both D_iso and overall S2 scaling..")
        p_init = (Diso_init, 1.0)
        d_init = ((0.1 * p_init[0], 0.1 * p_init[1]), (0.1 * p_init[0], -0.1 * p_init[1]))
        fminOut = fmin_powell(optfunc_R1R2NOE_DisoS2, x0=p_init, direc=d_init,
                              args=(relax_obj, fnum, fS2, fconsts, ftaus, fvecXH, fw, CSAvaluesArray, expblock), full_output=True)
        print(fminOut)
        Diso_opt, S2s_opt, chisq = fminOut[0][0], fminOut[0][1], fminOut[1]
        optHeader = header_fn(names=param_names, values=np.multiply(param_scaling, (Diso_opt, S2s_opt, relax_obj.gX.csa, np.sqrt(chisq))),
                              units=param_units, bFit=(True, True, False, True))
        print(optHeader)
        S2_list = [S2s_opt * k for k in S2_list]
    elif optMode == 'Diso':
        print("= = Fitting D_iso..")
        fminOut = fmin_powell(optfunc_R1R2NOE_Diso, x0=Diso_init, direc=[0.1 * Diso_init],
                              args=(relax_obj, fnum, fS2, fconsts, ftaus, fvecXH, fw, CSAvaluesArray, expblock), full_output=True)
        print(fminOut)
        Diso_opt, chisq = fminOut[0], fminOut[1]
        optHeader = header_fn(names=param_names,
                              values=np.multiply(param_scaling, (float(np.ravel(Diso_opt)[0]), 1.0, relax_obj.gX.csa, np.sqrt(chisq))),
                              units=param_units, bFit=(True, False, False, True))
        print(optHeader)
    else:
        print("= = Invalid optimisation mode!", file=sys.stderr)
        sys.exit(1)
    return optHeader, CSAvaluesArray, S2_list
