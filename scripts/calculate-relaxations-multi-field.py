#!/usr/bin/env python3
"""
Drop-in for the reference's calculate-relaxations-multi-field.py (run-all.bash:537-543): predicts every given
experiment (R1 / R2 / NOE at arbitrary fields) from the fitted C(t) parameters, the vector distribution and the
global diffusion tensor, optionally optimising Diso, Daniso, zeta, CSA (global) and/or rsCSA (per residue), and
writes `<o>_<15N1H>_<MHz>MHz_<Type>.xvg` per experiment and `<o>_CSA_opt.dat`.  All experiments, residues and
histogram bins are evaluated in one batched GPU launch per objective call; the rsCSA step uses the closed-form
CSA dependence (spinrelax_amd/spin_relaxation.py).
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


def parse_rotdif_params(D=None, tau=None, aniso=None):
    """calculate-relaxations-multi-field.py:13-37."""
    if D is None:
        if tau is None:
            print("= = ERROR: No global tumbling parameters given!", file=sys.stderr)
            sys.exit(1)
        Diso = 1.0 / (6 * tau)
        if aniso is None or aniso == 1.0:
            return sd.globalRotationalDiffusion_Isotropic(D=Diso)
        return sd.globalRotationalDiffusion_Axisymmetric(D=[Diso, aniso])
    tmp = [float(x) for x in regexp_split('[, ]', D) if len(x) > 0]
    if len(tmp) == 1:
        if aniso is None:
            return sd.globalRotationalDiffusion_Isotropic(D=tmp[0])
        return sd.globalRotationalDiffusion_Axisymmetric(D=[tmp[0], aniso])
    if len(tmp) == 2:
        return sd.globalRotationalDiffusion_Axisymmetric(D=tmp, bConvert=True)
    print("WARNING: fully anisotropic global rotdif not implemented.", file=sys.stderr)
    sys.exit(1)


def main():
    p = argparse.ArgumentParser(description='Prediction / optimisation of spin relaxation against several experiments at once.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('expFiles', type=str, nargs='+', help='Experiment files (# Type, # NucleiA, # NucleiB, # Frequency header).')
    p.add_argument('-o', '--outpref', type=str, dest='out_pref', default='out', help='Output file prefix.')
    p.add_argument('-f', '--infn', type=str, dest='in_Ct_fn', required=True, help='Fitted C_internal(t) parameter file.')
    p.add_argument('--refpdb', type=str, dest='refPDBFile', default=None, help='(MDTraj only; not supported by the GPU build)')
    p.add_argument('--distfn', type=str, dest='distfn', default=None, help='Vector orientation distribution (.npz histogram).')
    p.add_argument('--tau', type=float, dest='tau', default=None, help='Isotropic relaxation time constant.')
    p.add_argument('--aniso', type=float, dest='aniso', default=None, help='Diffusion anisotropy.')
    p.add_argument('-D', '--DTensor', type=str, dest='D', default=None, help='Diso, or "Dpar Dperp".')
    p.add_argument('--zeta', type=float, default=0.890023, help='Zero-point vibration scaling.')
    p.add_argument('--csa', type=str, default=None, help='CSA value or per-residue CSA file.')
    p.add_argument('--opt', '--fit', type=str, dest='listOptParams', default=None,
                   help='Comma-separated parameters to optimise: %s' % sd.spinRelaxationExperiments.listAllowedOptimisationVariables)
    p.add_argument('--cycles', type=int, default=10, help='Maximum global/local cycles.')
    p.add_argument('--tol', type=float, default=1e-6, help='Fractional-change tolerance of the global/local cycles.')
    args = p.parse_args()
    time_start = time.time()
    srdist.start()          # under torchrun: every batched evaluation splits the residues over the ranks; rank 0 writes
    args.out_pref = srdist.output_prefix(args.out_pref)

    localCtModel = fitCt.read_fittedCt_parameters(args.in_Ct_fn)
    if localCtModel.nModels == 0:
        print("= = = ERROR: The fitted-Ct file %s was read, but did not yield any usable parameters!" % args.in_Ct_fn)
        sys.exit(1)
    globalRotDif = parse_rotdif_params(args.D, args.tau, args.aniso)
    if args.distfn is not None:
        globalRotDif.import_frame_vectors(args.distfn)
    elif args.refPDBFile is not None:
        print("= = = ERROR: --refpdb needs MDTraj; give --distfn instead.", file=sys.stderr)
        sys.exit(1)

    objExpts = sd.spinRelaxationExperiments(globalRotDif, localCtModel)
    for f in args.expFiles:
        objExpts.add_experiment(f)
    if args.zeta != 1.0:
        print(" = = Applying scaling of all C(t) magnitudes to account for zero-point QM vibrations (zeta) of %g" % args.zeta)
        objExpts.set_global_zeta(args.zeta)
    objExpts.map_experiment_peaknames_to_models()
    objExpts.report_maps()

    if args.csa is None:
        print("= = = Using default CSA value respective to each experiment.")
    elif os.path.isfile(args.csa):
        residCSA, CSAvaluesArray = gs.load_xy(args.csa)
        residCSA = [str(int(x)) for x in residCSA]
        print("= = = Using input CSA values from file %s - please ensure that the names match those found in C(t) models." % args.csa)
        if np.fabs(CSAvaluesArray[0]) > 1.0:
            print("= = = NOTE: the first value is > 1.0, so assume a necessary conversion to ppm.")
            CSAvaluesArray *= 1e-6
        objExpts.initialise_CSA_array(residCSA, CSAvaluesArray)
    else:
        try:
            tmp = float(args.csa)
        except ValueError:
            print("= = = ERROR at parsing the --csa argument!", file=sys.stderr)
            sys.exit(1)
        print("= = = Using user-input CSA value: %g" % tmp)
        if np.fabs(tmp) > 1.0:
            print("= = = NOTE: this value is > 1.0, so assume a necessary conversion to ppm.")
            tmp *= 1e-6
        objExpts.initialise_CSA_array(objExpts.localCtModels.get_names(), np.repeat(tmp, objExpts.localCtModels.nModels))

    if args.listOptParams is None:
        objExpts.eval_all(bVerbose=True)
        objExpts.export_xvg(args.out_pref, bIncludeExpt=False)
        print("= = Finished. Total seconds elapsed: %g" % (time.time() - time_start))
        srdist.finish()
        sys.exit()

    objExpts.parse_optimisation_params(args.listOptParams.split(','))
    print("= = = Parsed optimiser input %s." % args.listOptParams)
    print("    ... conducting global optimisations over parameters %s ..." % (objExpts.listUpdateVariables))
    if objExpts.bDoLocalOpt:
        print("    ... conducting local optimisations over residue-specific CSA...")
    chisq = objExpts.perform_optimisation(maxCycles=args.cycles, tol=args.tol)
    print("= = = Optimisation complete. Final chi-value: %g" % np.sqrt(chisq))
    objExpts.export_xvg(args.out_pref, bIncludeExpt=True)
    if objExpts.bDoLocalOpt and objExpts.bOptCompleted:
        with open(args.out_pref + '_CSA_opt.dat', 'w') as fp:
            for x, y in zip(objExpts.localCtModels.get_names(), objExpts.get_first_csa()):
                print("%s %g" % (x, y), file=fp)
    print("= = Finished. Total seconds elapsed: %g (%d objective evaluations)" % (time.time() - time_start, objExpts.nObjectiveCalls))
    srdist.finish()


if __name__ == '__main__':
    main()
