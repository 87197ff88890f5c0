#!/usr/bin/env python3
"""
Drop-in for the reference's calculate-relaxations-from-Ct.py (run-all.bash:510-528): reads the fitted C(t)
parameters (`_fittedCt.dat`), the vector distribution (`_vecHistogram.npz`), the global diffusion tensor and
the field, and writes `<o>_R1.dat`, `<o>_R2.dat`, `<o>_NOE.dat`, `<o>_rho.dat` (or `<o>_Jw.dat` with
--Jomega) in the reference's formats.  J(omega) and the relaxation rates of all residues, histogram bins
and frequencies are evaluated in one batched GPU launch (sr_jomega_relax_f64).

The legacy single-field optimisation modes (--opt Diso|DisoS2|DisoCSA|DisoS2CSA|new; marked deprecated in
the reference's README.md:106) run scipy's Powell search on the host with every
objective evaluation being one launch of the same kernel (spinrelax_amd/legacy_opt.py).
"""
import argparse
import os
import sys
import time
from re import split as regexp_split

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spinrelax_amd import general_scripts as gs             # noqa: E402
from spinrelax_amd import fitting_Ct_functions as fitCt     # noqa: E402
from spinrelax_amd import spectral_densities as sd          # noqa: E402
from spinrelax_amd import dist as srdist                    # noqa: E402


def sanity_check_two_list(listA, listB, string):
    if len(listA) != len(listB) or any(a != b for a, b in zip(listA, listB)):
        print("= = ERROR: Sanity checked failed for %s!" % string)
        print("    ...first residues:", listA[0], listB[0])
        print("    ...set intersection (unordered):", set(listA).intersection(set(listB)))
        sys.exit(1)


def print_fitting_params_headers(names, values, units, bFit):
    """calculate-relaxations-from-Ct.py:331-340."""
    out = ""
    for i in range(len(names)):
        out += "# %s %s: %g %s\n" % ("Optimised" if bFit[i] else "Fixed", names[i], values[i], units[i])
    return out


def build_parser():
    p = argparse.ArgumentParser(description='Read fitted-Ct values and calculate relaxation parameters assuming '
                                            'Ct = C_internal(t) * C_external(t); global tumbling must be given.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-f', '--infn', type=str, dest='in_Ct_fn', help='Fitted C_internal(t) parameter file.')
    p.add_argument('-o', '--outpref', type=str, dest='out_pref', default='out', help='Output file prefix.')
    p.add_argument('-v', '--vecfn', type=str, dest='vecfn', default=None, help='Average vector orientations (resid x y z).')
    p.add_argument('--distfn', type=str, dest='distfn', default=None, help='Vector orientation distribution (.npz histogram).')
    p.add_argument('--shiftres', type=int, default=0, help='Shift the MD residue indices.')
    p.add_argument('-e', '--expfn', type=str, dest='expfn', default=None, help='Experimental R1/R2/NOE file (needed by --opt).')
    p.add_argument('--ref', type=str, dest='reffn', default=None, help='Reference PDB (not implemented in the reference either).')
    p.add_argument('--refHsel', type=str, default='name H')
    p.add_argument('--refXsel', type=str, default='name N and not resname PRO')
    p.add_argument('--traj', type=str, dest='trjfn', default=None)
    p.add_argument('-q', '--q_rot', type=str, dest='qrot_str', default='', help='Rotation quaternion "w x y z" into the PAF.')
    p.add_argument('-n', '--nuclei', type=str, dest='nuclei', default='NH', help='Nuclei pair: NH or CH.')
    p.add_argument('-B', '--B0', type=float, dest='B0', default=None, help='Magnetic field in T.')
    p.add_argument('-F', '--freq', type=float, dest='Hz', default=None, help='Proton frequency in Hz (overrides B0).')
    p.add_argument('--Jomega', action='store_true', help='Output J(omega) instead of R1, R2, NOE and rho.')
    p.add_argument('--tu', '--time_units', type=str, dest='time_unit', default='ps', help='Time units of the autocorrelation file.')
    p.add_argument('--tau', type=float, dest='tau', default=None, help='Isotropic tumbling time.')
    p.add_argument('--aniso', type=float, dest='aniso', default=1.0, help='Diffusion anisotropy.')
    p.add_argument('-D', '--DTensor', type=str, dest='D', default=None, help='Diffusion tensor "Diso [Daniso [Drhomb]]".')
    p.add_argument('--rXH', type=float, default=np.nan)
    p.add_argument('--zeta', type=float, default=0.890023, help='Zero-point vibration scaling of all C(t) amplitudes.')
    p.add_argument('--csa', type=str, default=None, help='CSA value, or a file of per-residue CSA values.')
    p.add_argument('--opt', '--fit', type=str, default=None,
                   help='Legacy single-field optimisation against --expfn: Diso, DisoS2, DisoCSA, DisoS2CSA or new (global Diso + per-residue CSA).')
    p.add_argument('--cycles', type=int, default=100, help='Maximum refinement cycles of --opt new.')
    p.add_argument('--tol', type=float, default=1e-6, help='Relative tolerance that ends the refinement cycles of --opt new.')
    p.add_argument('--theoretical', dest='bTheoretical', action='store_true',
                   help='Print the rigid-body relaxation (S2 = zeta, no internal motion) and exit.')
    return p


def main():
    time_start = time.time()
    args = build_parser().parse_args()
    srdist.start()          # under torchrun: residues are split over the ranks inside every batched evaluation, rank 0 writes
    out_pref = srdist.output_prefix(args.out_pref)
    if args.opt is not None:
        if args.expfn is None:
            print("= = = ERROR: Cannot conduct optimisation without a target experimental scattering file! (Missing --expfn )", file=sys.stderr)
            sys.exit(1)
        if args.opt not in ('new', 'Diso', 'DisoS2', 'DisoCSA', 'DisoS2CSA'):
            print("= = Invalid optimisation mode!", file=sys.stderr)
            sys.exit(1)
    zeta = args.zeta
    if zeta != 1.0:
        print(" = = Applying scaling of all C(t) magnitudes to account for zero-point QM vibrations (zeta) of %g" % zeta)
    if args.Hz is not None:
        B0 = 2.0 * np.pi * args.Hz / 267.513e6
    elif args.B0 is not None:
        B0 = args.B0
    else:
        print("= = = ERROR: Must give either the background magnetic field or the frequency! E.g., --B0 14.0956", file=sys.stderr)
        sys.exit(1)

    relax_obj = sd.relaxationModel(args.nuclei, B0)
    relax_obj.set_time_unit(args.time_unit)
    print("= = = Setting up magnetic field:", B0, "T")
    print("= = = Angular frequencies in ps^-1 based on given parameters:")
    relax_obj.print_frequencies()
    print("= = = Gamma values: (X) %g , (H) %g rad s^-1 T^-1" % (relax_obj.gX.gamma, relax_obj.gH.gamma))

    if args.D is None:
        if args.tau is None:
            diff_type, Diso = 'direct', 0.0
        else:
            Diso = 1.0 / (6 * args.tau)
            aniso = args.aniso
            diff_type = 'symmtop' if args.aniso != 1.0 else 'spherical'
    else:
        tmp = [float(x) for x in regexp_split('[, ]', args.D) if len(x) > 0]
        Diso = tmp[0]
        if len(tmp) == 1:
            diff_type = 'spherical'
        elif len(tmp) == 2:
            aniso = tmp[1]
            diff_type = 'symmtop'
        else:
            print("= = = ERROR: fully anisotropic diffusion is not implemented (neither in the reference).", file=sys.stderr)
            sys.exit(1)

    vecXH, vecXHweights, resNH = None, None, None
    bHaveDy = False
    if diff_type == 'direct':
        print("= = = No global rotational diffusion selected. Calculating the direct transform.")
        relax_obj.set_rotdif_model('direct_transform')
    elif diff_type == 'spherical':
        print("= = = Using a spherical rotational diffusion model.")
        relax_obj.set_rotdif_model('rigid_sphere_D', Diso)
    else:
        Dperp = 3. * Diso / (2 + aniso)
        Dpar = aniso * Dperp
        print("= = = Calculated anisotropy to be: ", aniso)
        print("= = = With Dpar, Dperp: %g, %g %s^-1" % (Dpar, Dperp, args.time_unit))
        relax_obj.set_rotdif_model('rigid_symmtop_D', Dpar, Dperp)
        if args.vecfn is not None:
            print("= = = Using average vectors. Reading X-H vectors from %s ..." % args.vecfn)
            resNH, vecXH = gs.load_xys(args.vecfn)
            resNH = [int(x) + args.shiftres for x in resNH]
        elif args.distfn is not None:
            print("= = = Using vector distribution in spherical coordinates. Reading X-H vector distribution from %s ..." % args.distfn)
            resNH, vecXH, vecXHweights = sd.read_vector_distribution_from_file(args.distfn)
            resNH = [int(x) + args.shiftres for x in resNH]
            bHaveDy = True
        elif not args.bTheoretical:
            print("= = = ERROR: non-spherical diffusion models require a vector source! "
                  "Please supply the average vectors or a trajectory and reference!", file=sys.stderr)
            sys.exit(1)
        if vecXH is not None:
            print("= = = Note: the shape of the X-H vector distribution is:", vecXH.shape)
            if args.qrot_str != "":
                q_rot = np.array([float(v) for v in args.qrot_str.split()])
                print("    ....rotating input vectors into PAF frame using q_rot.")
                from spinrelax_amd import ct as hostct
                sh = vecXH.shape
                vecXH = hostct.rotate_vector_simd(np.ascontiguousarray(vecXH, dtype=np.float32).reshape(-1, 1, 3), q_rot).reshape(sh)

    if args.bTheoretical:
        if diff_type == 'direct':
            print("= = = ERROR: Rigid-sphere argument cannot be applied without an input for the global rotational diffusion!", file=sys.stderr)
            sys.exit(1)
        if diff_type == 'spherical':
            num_vecs, vecs = 1, None
        else:
            num_vecs, vecs = 3, np.identity(3)
        datablock = sd._obtain_R1R2NOErho(relax_obj, num_vecs, [zeta] * num_vecs, [[0.]] * num_vecs, [[99999.]] * num_vecs, vecs)
        print("...Isotropic baseline values:" if diff_type == 'spherical' else "...Anistropic axial baseline values (x/y/z):")
        print("R1:", str(datablock[0]).strip('[]'))
        print("R2:", str(datablock[1]).strip('[]'))
        print("NOE:", str(datablock[2]).strip('[]'))
        sys.exit()

    autoCorrs = fitCt.read_fittedCt_parameters(args.in_Ct_fn)
    if autoCorrs.nModels == 0:
        print("= = = ERROR: The fitted-Ct file %s was read, but did not yield any usable parameters!" % args.in_Ct_fn)
        sys.exit(1)
    num_vecs = autoCorrs.nModels
    sim_resid = [int(k) for k in autoCorrs.model.keys()]
    if diff_type == 'symmtop':
        sanity_check_two_list(sim_resid, resNH, "resid from fitted_Ct -versus- vectors as defined in anisotropy")

    # CSA input (calculate-relaxations-from-Ct.py:702-743): a number, or a file with one value per residue
    if args.csa is None:
        print("= = = Using default CSA value: %g" % relax_obj.gX.csa)
        CSAvaluesArray = np.repeat(relax_obj.gX.csa, num_vecs)
    elif os.path.isfile(args.csa):
        residCSA, CSAvaluesArray = gs.load_xy(args.csa)
        relax_obj.gX.csa = np.nan
        print("= = = Using input CSA values from file %s - please ensure that the resid definitions are identical to the other files." % args.csa)
        sanity_check_two_list(sim_resid, [int(x) for x in residCSA], "resid from fitted_Ct -versus- as defined in CSA file ")
        if np.fabs(CSAvaluesArray[0]) > 1.0:
            print("= = = NOTE: the first value is > 1.0, so assume a necessary conversion to ppm.")
            CSAvaluesArray *= 1e-6
    else:
        try:
            tmp = float(args.csa)
        except ValueError:
            print("= = = ERROR at parsing the --csa argument!", file=sys.stderr)
            sys.exit(1)
        print("= = = Using user-input CSA value: %g" % tmp)
        relax_obj.gX.csa = tmp
        if np.fabs(tmp) > 1.0:
            print("= = = NOTE: this value is > 1.0, so assume a necessary conversion to ppm.")
            relax_obj.gX.csa *= 1e-6
        CSAvaluesArray = np.repeat(relax_obj.gX.csa, num_vecs)

    S2_list, consts_list, taus_list, _ = autoCorrs.get_params_as_list()
    for i in range(num_vecs):                                 # :747-750
        S2_list[i] *= zeta
        consts_list[i] *= zeta

    param_names = ("Diso", "zeta", "CSA", "chi")
    param_scaling = (1.0, zeta, 1.0e6, 1.0)
    param_units = (relax_obj.timeUnit + "^-1", "a.u.", "ppm", "a.u.")
    if args.opt is None:
        if args.Jomega:
            datablock = sd._obtain_Jomega(relax_obj, num_vecs, S2_list, consts_list, taus_list, vecXH, weights=vecXHweights)
        else:
            datablock = sd._obtain_R1R2NOErho(relax_obj, num_vecs, S2_list, consts_list, taus_list, vecXH, weights=vecXHweights,
                                              CSAvaluesArray=CSAvaluesArray)
        optHeader = print_fitting_params_headers(names=param_names,
                                                 values=np.multiply(param_scaling, (Diso, 1.0, relax_obj.gX.csa, 0.0)),
                                                 units=param_units, bFit=(False, False, False, False))
    else:
        # Section 2 of the reference (:775-1004): minimisation against experimental data, legacy single-field modes
        from spinrelax_amd import legacy_opt
        print("= = = Reading Experimental relaxation parameter files")
        exp_resid, expblock = legacy_opt.read_experiment(args.expfn, relax_obj.rotdifModel.name, gs.load_xys)
        matched = legacy_opt.match_residues(sim_resid, [int(x) for x in exp_resid], expblock, S2_list, consts_list, taus_list,
                                            vecXH, vecXHweights, CSAvaluesArray)
        optHeader, CSAvaluesArray, S2_list = legacy_opt.run(
            args.opt, relax_obj, Diso, matched, None, args.cycles, args.tol, out_pref, print_fitting_params_headers,
            param_names, param_scaling, param_units, gs.print_xy, sim_resid, CSAvaluesArray, S2_list)
        datablock = sd._obtain_R1R2NOErho(relax_obj, num_vecs, S2_list, consts_list, taus_list, vecXH, weights=vecXHweights,
                                          CSAvaluesArray=CSAvaluesArray)
    print(" = = Completed Relaxation calculations.")

    if args.Jomega:
        with open(out_pref + '_Jw.dat', 'w') as fp:
            if optHeader != '':
                print('%s' % optHeader, file=fp)
            if bHaveDy:
                print('@type xydy', file=fp)
            xdat = np.fabs(relax_obj.omega)
            for i in range(num_vecs):
                print('@s%d legend "Resid: %d"' % (i, sim_resid[i]), file=fp)
                for j in np.argsort(xdat):
                    if bHaveDy:
                        print('%g %g %g' % (xdat[j], datablock[j, i, 0], datablock[j, i, 1]), file=fp)
                    else:
                        print('%g %g' % (xdat[j], datablock[j, i]), file=fp)
                print('&', file=fp)
    elif not bHaveDy:
        gs.print_xy(out_pref + '_R1.dat', sim_resid, datablock[0, :], header=optHeader)
        gs.print_xy(out_pref + '_R2.dat', sim_resid, datablock[1, :], header=optHeader)
        gs.print_xy(out_pref + '_NOE.dat', sim_resid, datablock[2, :], header=optHeader)
        gs.print_xy(out_pref + '_rho.dat', sim_resid, datablock[3, :])
    else:
        gs.print_xydy(out_pref + '_R1.dat', sim_resid, datablock[0, :, 0], datablock[0, :, 1], header=optHeader)
        gs.print_xydy(out_pref + '_R2.dat', sim_resid, datablock[1, :, 0], datablock[1, :, 1], header=optHeader)
        gs.print_xydy(out_pref + '_NOE.dat', sim_resid, datablock[2, :, 0], datablock[2, :, 1], header=optHeader)
        gs.print_xydy(out_pref + '_rho.dat', sim_resid, datablock[3, :, 0], datablock[3, :, 1])
    print("= = Finished. Total seconds elapsed: %g" % (time.time() - time_start))
    srdist.finish()


if __name__ == '__main__':
    main()
