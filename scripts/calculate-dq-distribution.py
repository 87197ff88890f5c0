#!/usr/bin/env python3
"""
Drop-in for the reference's calculate-dq-distribution.py (run-all.bash:383-387: `--iso --aniso -f colvar-qorient -o <pref>
--mindt t --skip t --maxdt tau --num_chunk n`): global rotational diffusion from the orientation trajectory.  Reads the
PLUMED quaternion file (or a `gmx rotmat` .xvg), computes the difference-quaternion statistics for every lag on the GPU
(one launch, spinrelax_amd/csrc/sr_dq.hip), fits the decays and writes `<pref>-iso.dat`, `<pref>-aniso2.dat`,
`<pref>-aniso_q.dat`, `<pref>-moi.xyz` (and `<pref>-tensor.dat` with --fulltensor) in the reference's formats.
The per-lag 3-D histograms of --hist are not part of the hot path and are not produced.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spinrelax_amd import dq_distribution as dq            # noqa: E402
from spinrelax_amd import general_scripts as gs            # noqa: E402
from spinrelax_amd import plumedcolvario as pl             # noqa: E402


def main():
    p = argparse.ArgumentParser(description='Calculates the difference quaternions from PLUMED output: a time-series of '
                                            'quaternion representation of orientations, then manipulate them in various ways',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-f', '--infn', type=str, dest='infn', default='colvar-q',
                   help='Input file in PLUMED quaternion form or GROMACS rotational-matrix .xvg file. Assumes that dt is identical between every frame!')
    p.add_argument('-o', '--outpref', type=str, dest='out_pref', default='out', help='Output file prefix for the quaternion-decay curves.')
    p.add_argument('--hist', dest='bDoHist', action='store_true', default=False, help='(not supported here) 3D-histogram of dq at each delay time.')
    p.add_argument('-o2', '--outtype', type=str, dest='out_suff', default='dat', help='File format of the histograms: dx, dat or none.')
    p.add_argument('--iso', dest='bDoIso', action='store_true', default=False, help='Record the isotropic decay of dq.')
    p.add_argument('--aniso', dest='bDoAniso', action='store_true', default=False,
                   help='Record an estimate of the anisotropic decay of dq, using the coordinate axes as a guide.')
    p.add_argument('--fulltensor', dest='bDoFullTensor', action='store_true', default=False,
                   help='Record all nine components of the tensor <q_i q_j> in the PAF frame.')
    p.add_argument('-n', '--num_bins', type=int, dest='num_bins', default=101, help='Number of histogram bins spanning [-1,1].')
    p.add_argument('--mindt', '--min_dt', type=float, dest='min_dt', default=0.0, help='Minimum interval delta_t in picoseconds [ps].')
    p.add_argument('--num_chunk', '--num_chunks', type=int, dest='num_chunk', default=0,
                   help='Uncertainty estimation from sub-chunks of the trajectory: reports their standard deviation and their plots.')
    p.add_argument('--maxdt', '--max_dt', type=float, dest='max_dt', default=1000.0, help='Maximum interval delta_t in picoseconds [ps].')
    p.add_argument('--skip', '--skip_dt', type=float, dest='skip_dt', default=0.0, help='The interval of time in which the calculation should be carried out.')
    args = p.parse_args()
    time_start = time.time()
    if args.out_suff not in ('dx', 'dat', 'none'):
        print("= = ERROR in input: histogram output type must be either dx, or dat, or none.")
        sys.exit()
    if args.bDoHist:
        print("= = ERROR: --hist (3D histograms of dq per lag) is not implemented in the MI355X build.", file=sys.stderr)
        sys.exit(1)

    if args.infn.endswith('.xvg'):
        t, m = gs.load_xys(args.infn)
        data = dq.rotmatrix_to_quaternion(t, m, bInvert=True)
    else:
        res = pl.read_from_plumedprint(args.infn)
        if res == -1:
            sys.exit(1)
        fields, data = res
        nfield, ndat = data.shape
        print("= = Input data found to be %i fields and %i entries. = =" % (nfield, ndat))
    print(data[1:5, 0])
    qprev = data[1:5, 0]
    print("= = Initial quaternion read: (%f %f %f %f) = =" % (qprev[0], qprev[1], qprev[2], qprev[3]))

    out = dq.analyse(data, min_dt=args.min_dt, max_dt=args.max_dt, skip_dt=args.skip_dt, num_chunk=args.num_chunk,
                     bDoIso=args.bDoIso, bDoAniso=args.bDoAniso, bDoFullTensor=args.bDoFullTensor)
    time_chk1 = time.time()
    dq.fit_and_write(out, args.out_pref, num_chunk=args.num_chunk, bDoIso=args.bDoIso, bDoAniso=args.bDoAniso,
                     bDoFullTensor=args.bDoFullTensor)
    time_stop = time.time()
    print("= = Total seconds elapsed: %g" % (time_stop - time_start))
    print("= = Time of Read and fit halves: %g , %g" % (time_chk1 - time_start, time_stop - time_chk1))


if __name__ == '__main__':
    main()
