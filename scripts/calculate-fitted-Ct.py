#!/usr/bin/env python3
"""
Drop-in for the reference's calculate-fitted-Ct.py (run-all.bash:488-491): reads `_Ctint.dat`, fits every
residue's C(t) with 1..4 exponentials (+ optional free S2) and writes `<o>_fittedCt.dat` in the same
format.  All residues are fitted together: one batched GPU trust-region solve per model order
(spinrelax_amd.fitting_Ct_functions.autoCorrelations.fit_all), with the reference's accept/reject rules.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spinrelax_amd import general_scripts as gs             # noqa: E402
from spinrelax_amd import fitting_Ct_functions as fitCt     # noqa: E402
from spinrelax_amd import dist as srdist                    # noqa: E402


def main():
    p = argparse.ArgumentParser(description='Fit the raw autocorrelation functions C(t) with a small set of exponential '
                                            'decays; the number of components is chosen by chi-square improvement.',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-f', '--infn', type=str, dest='in_Ct_fn', nargs='+', help='One or more C(t) files (xmgrace sets with legends).')
    p.add_argument('-o', '--outpref', type=str, dest='out_pref', default='out', help='Output file prefix.')
    p.add_argument('--nc', type=int, default=-1, help='Number of transient components; -1 searches for the best number.')
    p.add_argument('--nofast', dest='bNoFast', action='store_true', default=False, help='Forbid the S_fast component (C(0) must be one).')
    args = p.parse_args()
    time_start = time.time()
    srdist.start()          # under torchrun: residues are split over the ranks inside fit_all, rank 0 writes the file

    files = args.in_Ct_fn
    print("= = = Found %d input C(t) files." % len(files))
    if len(files) == 1:
        legs, dt, Ct, Cterr = gs.load_sxydylist(files[0], 'legend')
        legs = [int(x) for x in legs]
        if len(Cterr) == 0:
            Cterr = None
    else:
        # calculate-fitted-Ct.py:111-147: average the curves of several files (equal weights)
        print("    ...will perform averaging to obtain averaged C(t).")
        all_C, all_E = [], []
        for fn in files:
            legs, dt, C1, E1 = gs.load_sxydylist(fn, 'legend')
            legs = [int(x) for x in legs]
            all_C.append(C1)
            all_E.append(E1)
        all_C = np.array(all_C, dtype=float)
        Ct = np.mean(all_C, axis=0)
        if len(all_E[0]) == 0:
            Cterr = np.std(all_C, axis=0)
        else:
            all_E = np.array(all_E, dtype=float)
            gm_ = np.mean(all_C, axis=0)
            # general_maths.py:89-98 simple_total_mean_square: (between-copy + within-copy sum of squares) / copies
            Cterr = (np.sum((all_C - gm_) ** 2.0, axis=0) + np.sum(all_E ** 2.0, axis=0)) / all_C.shape[0]

    autoCorrs = fitCt.autoCorrelations()
    autoCorrs.import_target_array(keys=legs, DeltaT=dt, Decay=Ct, dDecay=Cterr)
    bUseSFast = not args.bNoFast
    listDoGs = [2, 3, 5, 7, 9] if bUseSFast else [2, 4, 6, 8]
    print("...Running C(t)-fits for %d residues on the GPU (orders %s)." % (len(legs), str(listDoGs if args.nc == -1 else args.nc)))
    autoCorrs.fit_all(listDoG=listDoGs, chiSqThreshold=0.5, nc=args.nc, bUseSFast=bUseSFast)
    out_fn = srdist.output_prefix(args.out_pref) + '_fittedCt.dat'
    autoCorrs.export(fileName=out_fn, style='xmgrace')
    print(" = = Completed C(t)-fits.")
    print("= = Finished. Total seconds elapsed: %g" % (time.time() - time_start))
    srdist.finish()


if __name__ == '__main__':
    main()
