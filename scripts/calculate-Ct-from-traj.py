#!/usr/bin/env python3
"""
Drop-in for the reference's calculate-Ct-from-traj.py (run-all.bash:476-481): same flags, same output
files (<o>_Ctext.dat, <o>_Ctint.dat, <o>_vecHistogram.npz | _vecPhiTheta.npz|.dat, <o>_avgvec.dat,
<o>_S2.dat).  C(t), the rotation into the PAF, the spherical histogram, the mean vector and S2 are
computed on the MI355X (libspinrelax_hip.so); this script only parses arguments and moves files.

Several GPUs: run under torchrun (`torchrun --nproc-per-node N scripts/calculate-Ct-from-traj.py ...`): rank r computes the
contiguous vector range spinrelax_amd.dist.shard_range gives it, the results are all-gathered (RCCL) and rank 0 writes the
same files a single process writes.

Trajectory input:
  * with MDTraj installed: -s <pdb> -f <xtc ...> exactly like the reference; MDTraj only reads the files and
    resolves the atom selections, the bond vectors, the centring and the per-frame superposition (reference lines
    64-86, 462-470) run on the GPU (csrc/sr_traj.hip);
  * without MDTraj (or for synthetic data): -f <file.npz|.npy>.  An .npz holds either raw coordinates --
    `xyz` (frames, atoms, 3) float32 like traj.xyz, `indexX`, `indexH` and, for the superposition, `ref_xyz` (atoms, 3)
    and `fit_indices` -- which go through the same GPU front end, or precomputed unit vectors `vecs` (frames, bonds, 3;
    body frame) with optional `vecs_lab` (lab frame for _Ctext.dat; defaults to `vecs`); plus `names` (resSeq per bond;
    default 2..V+1) and `dt` (ps; default --dt).  A .npy holds unit vectors.  -s is still parsed and ignored then.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spinrelax_amd import ct as hostct                      # noqa: E402
from spinrelax_amd import dist as srdist                    # noqa: E402
from spinrelax_amd import general_scripts as gs             # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description='Obtain the unit X-H vectors from one of more trajectories, and compute '
                                            'S^2, C(t) and the vector distribution on the GPU (Palmer memory-time chunks).',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('-s', type=str, dest='topfn', required=True, nargs='+', help='Topology PDB (reference frame for fits).')
    p.add_argument('-f', '--infn', type=str, dest='infn', required=True, nargs='+',
                   help='One or more trajectories, or .npy/.npz files of precomputed unit vectors.')
    p.add_argument('-o', '--outpref', type=str, dest='out_pref', default='out', help='Output file prefix.')
    p.add_argument('--split', type=int, dest='nSplitFrames', default=-1, help='Read trajectories in chunks of N frames (MDTraj input).')
    p.add_argument('-t', '--tau', type=float, dest='tau', default=None, help='Memory time (global tumbling estimate), same units as the trajectory.')
    p.add_argument('--prefact', type=float, dest='zeta', default=(1.02 / 1.04) ** 6, help='Zero-point vibration prefactor applied to S2 output.')
    p.add_argument('--S2', dest='bDoS2', action='store_true', default=False, help='Calculate order parameters S2.')
    p.add_argument('--Ct', dest='bDoCt', action='store_true', default=False, help='Calculate autocorrelation C(t).')
    p.add_argument('--vecDist', dest='bDoVecDistrib', action='store_true', default=False, help='Print the vector distribution in spherical coordinates.')
    p.add_argument('--binary', action='store_true', default=False, help='Numpy binary storage for the vector distribution.')
    p.add_argument('--vecHist', dest='bDoVecHist', action='store_true', default=False, help='Print the 2D histogram instead of all vectors.')
    p.add_argument('--histBin', type=int, default=72, help='Number of bins along phi; theta uses half this number.')
    p.add_argument('--vecAvg', dest='bDoVecAverage', action='store_true', default=False, help='Print the average unit XH-vector.')
    p.add_argument('--vecRot', dest='vecRotQ', type=str, default='', help='Rotation quaternion "w x y z" into the PAF frame.')
    p.add_argument('--Hsel', '--selection', type=str, dest='Hseltxt', default='name H', help='MDTraj selection of the H atoms.')
    p.add_argument('--Xsel', type=str, dest='Xseltxt', default='name N and not resname PRO', help='MDTraj selection of the X atoms.')
    p.add_argument('--fitsel', type=str, dest='fittxt', default='custom occupancy', help='MDTraj selection used for the superposition.')
    p.add_argument('--help_sel', action='store_true', help='Display help for selection texts and exit.')
    p.add_argument('--dt', type=float, default=10.0, help='[extension] frame spacing in ps for .npy vector input.')
    p.add_argument('--qfile', type=str, nargs='+', default=None,
                   help='[extension, SURVEY 8(f)-1] PLUMED colvar-qorient file(s), one per trajectory file (fields time q.w q.x q.y q.z): '
                        'the fitted (body-frame) vectors are the lab-frame vectors rotated on the GPU by the inverse orientation '
                        'quaternion of every frame, instead of coming from a superposition.')
    p.add_argument('--exact', action='store_true', help='[extension] float64 validation mode of the C(t) kernel.')
    return p


def load_vector_files(files, default_dt):
    """.npy / .npz vector input -> (resXH, [vecXH per file], [vecXHfit per file], deltaT)."""
    resXH, lab, body, dt = None, [], [], None
    for fn in files:
        if fn.endswith('.npy'):
            v = np.load(fn)
            names, vl, d = None, v, default_dt
        else:
            z = np.load(fn, allow_pickle=True)
            if 'xyz' in z:
                # raw coordinates (traj.xyz) + the index lists MDTraj's selections would give: the front end
                # (obtain_XHvecs, centre + superpose, calculate-Ct-from-traj.py:64-86, 462-470) runs on the GPU
                print("= = = Reading coordinate file %s ..." % fn)
                if 'ref_xyz' in z:
                    vl, v = hostct.superpose_XHvecs(z['xyz'], z['ref_xyz'], z['fit_indices'], z['indexX'], z['indexH'])
                    print("= = = Molecule centered and fitted.")
                else:
                    v = vl = hostct.obtain_XHvecs(z['xyz'], z['indexX'], z['indexH'])
            else:
                v = z['vecs']
                vl = z['vecs_lab'] if 'vecs_lab' in z else v
            names = list(z['names']) if 'names' in z else None
            d = np.float32(z['dt']) if 'dt' in z else np.float32(default_dt)      # float32 like MDTraj's frame times (see load_mdtraj)
        if v.ndim != 3 or v.shape[2] != 3:
            print("= = = ERROR: vector file %s does not hold a (frames, bonds, 3) array!" % fn, file=sys.stderr)
            sys.exit(1)
        names = list(range(2, v.shape[1] + 2)) if names is None else [int(x) for x in names]
        if resXH is None:
            resXH, dt = names, d
        elif dt != d or resXH != names:
            print("= = = ERROR: Differences in trajectories have been detected! Aborting.", file=sys.stderr)
            sys.exit(1)
        body.append(np.ascontiguousarray(v, dtype=np.float32))
        lab.append(np.ascontiguousarray(vl, dtype=np.float32))
    return resXH, lab, body, dt


def load_mdtraj(args, frames_per_chunk_of):
    """The reference's loader (calculate-Ct-from-traj.py:396-498) on top of MDTraj, STREAMING: MDTraj reads the file (in
    chunks of --split frames, `md.iterload`, like the reference's :426-453, or 1000 frames otherwise) and resolves the
    selections; every chunk's coordinates go to the GPU, where the bond vectors, the centring and the per-frame superposition
    are computed and APPENDED to this rank's resident vectors (spinrelax_amd.hip.ResidentVectors) -- the host never holds
    more than one chunk of coordinates and no vectors at all.  A file's tail that does not fill a block of memory time is
    cut when the file ends (reformat_vecs_by_tau, :259-272).  frames_per_chunk_of(dt) -> frames per block or None.
    Returns (resXH, lab vectors, fitted vectors, deltaT, total vectors V, first vector i0, frames kept)."""
    try:
        import mdtraj as md
    except ImportError:
        print("= = = ERROR: MDTraj is not installed; give precomputed vectors as .npy/.npz to -f instead.", file=sys.stderr)
        sys.exit(1)
    from spinrelax_amd import hip

    def select(traj):
        iX = traj.topology.select(args.Xseltxt)
        iH = traj.topology.select(args.Hseltxt)
        if len(iX) == 0 or len(iH) == 0 or len(iX) != len(iH):
            print("= = = ERROR: selection text failed to find matching atoms! N(%s) = %i , N(%s) = %i"
                  % (args.Xseltxt, len(iX), args.Hseltxt, len(iH)), file=sys.stderr)
            sys.exit(1)
        return iX, iH

    def fit_indices(ref, fn):
        if args.fittxt == 'custom occupancy':
            pdb = md.formats.pdb.pdbstructure.PdbStructure(open(fn))
            mask = [a.get_occupancy() for a in pdb.iter_atoms()]
            inds = ref.topology.select('all')
            return [inds[i] for i in range(len(mask)) if mask[i] > 0.0]
        return ref.topology.select(args.fittxt)

    ctx = hip.default_context()
    resXH, dt, lab, fit, V, i0, nloc = None, None, None, None, None, 0, 0
    for i, fn in enumerate(args.infn):
        top = args.topfn[i] if len(args.topfn) > 1 else args.topfn[0]
        ref = md.load(top)
        fi = fit_indices(ref, top)
        nchunk = args.nSplitFrames if args.nSplitFrames > 0 else 1000
        file_start = 0 if lab is None else lab.frames
        nfile = 0
        iX = iH = None
        for trj in md.iterload(fn, chunk=nchunk, top=top):
            if iX is None:
                # names, selections and the time step come from a file's FIRST chunk only, as in the reference (:436-438):
                # a later chunk may hold a single frame (MDTraj's .timestep raises for it) and float32 frame times of
                # later chunks give a time step that differs from the first one in its last bits
                names = [trj.topology.atom(k).residue.resSeq for k in trj.topology.select(args.Hseltxt)]
                d = trj.timestep          # MDTraj's np.float32, KEPT float32: the reference's int(tau/dt) and int(0.5*tau/dt)
                                          # (:241, :255) are float32 divisions under NumPy >= 2 -- tau = 10, dt = 0.1 gives 100, not 99
                iX, iH = select(trj)
                if resXH is None:
                    resXH, dt, V = names, d, len(iX)
                    i0, nloc = srdist.my_range(V) if srdist.world() > 1 else (0, V)
                    if nloc > 0:
                        lab, fit = ctx.vectors(nloc), ctx.vectors(nloc)
                elif dt != d or resXH != names:
                    print("= = = ERROR: Differences in trajectories have been detected! Aborting.", file=sys.stderr)
                    print("      ...delta-t: %g vs.%g " % (dt, d), file=sys.stderr)
                    print("      ...n-bonds: %g vs.%g " % (V, len(iX)), file=sys.stderr)
                    sys.exit(1)
            if nloc > 0:
                hip.append_xyz(ctx, lab, fit, trj.xyz, iX[i0:i0 + nloc], iH[i0:i0 + nloc], fi, ref.xyz[0])
            nfile += trj.xyz.shape[0]
        if iX is None:
            print("= = = ERROR: trajectory file %s holds no frames!" % fn, file=sys.stderr)
            sys.exit(1)
        F = frames_per_chunk_of(dt)
        keep = nfile if F is None else (nfile // F) * F
        if nloc > 0 and keep != nfile:
            lab.truncate(file_start + keep)
            fit.truncate(file_start + keep)
        print("= = = Molecule centered and fitted: %s, %i frames read, %i kept." % (fn, nfile, keep))
    frames = 0 if lab is None else lab.frames
    return resXH, lab, fit, dt, V, i0, frames


def main():
    args = build_parser().parse_args()
    time_start = time.time()
    rank, world = srdist.start()
    if args.help_sel:
        print("Notes: This program uses MDTraj selection syntax, e.g. 'chain A and resname GLY and name HA1 HA2'.")
        sys.exit(0)
    tau_memory = args.tau
    if args.bDoCt and tau_memory is None:
        print("= = = Refusing to do C(t)-analysis without using a block averaging over memory_time tau!", file=sys.stderr)
        sys.exit(1)
    bDoVecDistrib = args.bDoVecDistrib or args.bDoVecHist
    q_rot = None
    if args.vecRotQ != '':
        q_rot = np.array([float(v) for v in args.vecRotQ.split()])
        if len(q_rot) != 4 or not np.allclose(np.dot(q_rot, q_rot), 1):
            print("= = = ERROR: input rotation quaternion is malformed!", q_rot)
            sys.exit(23)
    if len(args.topfn) > 1 and len(args.topfn) != len(args.infn):
        print("= = ERROR: When giving multiple reference files, you must have one for each trajecfile file given!", file=sys.stderr)
        sys.exit(1)

    def frames_per_chunk_of(dt):
        if tau_memory is None:
            return None
        if dt > 0.5 * tau_memory:
            print("= = = ERROR: delta-t form the trajectory is too small relative to tau! %g vs. %g" % (dt, tau_memory), file=sys.stderr)
            sys.exit(1)
        return int(tau_memory / dt)

    host_fit = host_lab = None   # host copies of the vectors (vector-file input only: the --vecDist listing and the chunk-sharded C(t) need them)
    if all(f.endswith('.npy') or f.endswith('.npz') for f in args.infn):
        resXH, vecXH, vecXHfit, deltaT = load_vector_files(args.infn, args.dt)
        if args.qfile is not None:
            from spinrelax_amd import plumedcolvario
            if len(args.qfile) != len(vecXH):
                print("= = = ERROR: --qfile needs one orientation file per trajectory file!", file=sys.stderr)
                sys.exit(1)
            print("= = = De-tumbling the lab-frame vectors with the per-frame orientation quaternions.")
            vecXHfit = []
            for fn, lab in zip(args.qfile, vecXH):
                _, q = plumedcolvario.read_qorient(fn)
                if q.shape[0] != lab.shape[0]:
                    print("= = = ERROR: %s holds %i frames, the trajectory %i!" % (fn, q.shape[0], lab.shape[0]), file=sys.stderr)
                    sys.exit(1)
                vecXHfit.append(hostct.detumble_vectors(lab, q))
        F = frames_per_chunk_of(deltaT)
        # ONE upload per vector set, of this rank's columns only; when the file holds no separate lab-frame vectors the two
        # sets are the same arrays and share the resident copy
        same = all(a is b for a, b in zip(vecXH, vecXHfit)) and len(vecXH) == len(vecXHfit)
        # fewer vectors than ranks: C(t) is sharded over the replicate chunks instead; the vector
        # distribution keeps the vector sharding (the surplus ranks idle there)
        repl = args.bDoCt and srdist.replicate_sharding(vecXHfit[0].shape[1])
        rv_fit, V, i0, N = hostct.upload_shard(vecXHfit, F)
        rv_lab = rv_fit if same or not args.bDoCt or repl else hostct.upload_shard(vecXH, F)[0]
        host_fit, host_lab = vecXHfit, vecXH
        del vecXH
    else:
        if args.qfile is not None:
            print("= = = ERROR: --qfile works on vector-file input (.npy/.npz), not on MDTraj input.", file=sys.stderr)
            sys.exit(1)
        resXH, rv_lab, rv_fit, deltaT, V, i0, N = load_mdtraj(args, frames_per_chunk_of)
        repl = False
        if srdist.replicate_sharding(V):
            print("= = = ERROR: %i ranks for %i vectors: trajectory input is sharded by vector; use at most %i ranks (vector-file "
                  "input can shard the replicate chunks instead)." % (srdist.world(), V, V), file=sys.stderr)
            sys.exit(1)
        F = frames_per_chunk_of(deltaT)
        if N < 1:
            print("= = = ERROR: no trajectory holds a full block of memory time tau!", file=sys.stderr)
            sys.exit(1)
    print("= = Loading finished.")
    out_pref = srdist.output_prefix(args.out_pref)

    if tau_memory is not None:
        print("= = Reformatting all vecXH information into chunks of tau ( %g ) " % tau_memory)
        print("    ...debug: Using %i frames per chunk based on tau/dt (%g/%g)." % (F, tau_memory, deltaT))
        R = N // F

    if args.bDoCt:
        mode = 1 if args.exact else 0
        dt = hostct.calculate_dt(deltaT, tau_memory)
        print("= = = Conducting Ct_external using Palmer's approach.")
        print("= = = timestep: ", deltaT, "ps")
        print("= = = tau_memory: ", tau_memory, "ps")
        if repl:
            Ct, dCt = hostct.calculate_Ct_chunk_sharded(host_lab, F, mode=mode)
        else:
            Ct, dCt = hostct.calculate_Ct_resident(rv_lab, V, R, F, mode=mode)
        gs.print_sxylist(out_pref + '_Ctext.dat', resXH, dt, np.stack((Ct.T, dCt.T), axis=-1))
        print("= = = Conducting Ct_internal using Palmer's approach.")
        if repl:
            if not same:
                Ct, dCt = hostct.calculate_Ct_chunk_sharded(host_fit, F, mode=mode)
        elif rv_fit is not rv_lab:
            Ct, dCt = hostct.calculate_Ct_resident(rv_fit, V, R, F, mode=mode)
        gs.print_sxylist(out_pref + '_Ctint.dat', resXH, dt, np.stack((Ct.T, dCt.T), axis=-1))
    if rv_lab is not None and rv_lab is not rv_fit:
        rv_lab.close()

    need_dist = args.bDoVecAverage or args.bDoS2 or (bDoVecDistrib and args.bDoVecHist)
    if need_dist:
        if q_rot is not None:
            print("= = = Rotating all fitted vectors by the input quaternion into PAF.")
        dist = hostct.vector_distribution_resident(rv_fit, V, N, q_rot, histBinX=args.histBin,
                                                   delta_t=deltaT if tau_memory is not None else -1,
                                                   tau_memory=tau_memory if tau_memory is not None else -1)
    if bDoVecDistrib and not args.bDoVecHist:
        # the full (phi, theta) listing needs every rotated vector on the host
        if host_fit is not None:
            used = [v[: (v.shape[0] // F) * F] if tau_memory is not None else v for v in host_fit]
            vec3d = used[0] if len(used) == 1 else np.concatenate(used, axis=0)
        else:
            part = rv_fit.download() if rv_fit is not None else np.empty((N, 0, 3), dtype=np.float32)
            vec3d = srdist.gather_rows(part, V, axis=1) if srdist.world() > 1 else part
    if rv_fit is not None:
        rv_fit.close()
    if args.bDoVecAverage:
        gs.print_xylist(out_pref + '_avgvec.dat', resXH, np.array(dist['avgvec']).T, True)
    if bDoVecDistrib:
        print("= = = Converting vectors into spherical coordinates.")
        if not args.bDoVecHist:
            # full (phi, theta) listing: rotation on the GPU, the two angle maps are plain numpy ufuncs
            rot = hostct.rotate_vector_simd(vec3d, q_rot) if q_rot is not None else vec3d
            r = np.linalg.norm(rot, axis=-1)
            pt = np.stack((np.arctan2(rot[..., 1], rot[..., 0]), np.arccos(rot[..., 2] / r)), axis=-1)
            pt = np.transpose(pt, axes=(1, 0, 2))
            if args.binary:
                gs.save_vecPhiTheta_npz(out_pref + '_vecPhiTheta.npz', resXH, pt)
            else:
                rtp = np.concatenate((np.transpose(r)[..., None], pt), axis=-1)
                gs.print_s3d(out_pref + '_vecPhiTheta.dat', resXH, rtp, (1, 2))
        else:
            print("= = = Histgrams will use Lambert Cylindrical projection by converting Theta spanning (0,pi) to cos(Theta) spanning (-1,1)")
            if args.binary:
                gs.save_vecHistogram_npz(out_pref + '_vecHistogram.npz', resXH, dist['hist'], dist['edges'])
            else:
                for i in range(len(resXH)):
                    ofile = out_pref + '_vecXH_' + str(resXH[i]) + '.hist'
                    gs.print_gplot_hist(ofile, dist['hist'][i], dist['edges'],
                                        header='# Lamber Cylindrical Histogram over phi,cos(theta).', bSphere=True)
                    print("= = = Written to output: ", ofile)
    if args.bDoS2:
        if tau_memory is not None:
            print("= = = Conducting S2 analysis using memory time to chop input-trajectories", tau_memory, "ps")
        else:
            print("= = = Conducting S2 analysis directly from trajectories.")
        S2 = dist['S2']
        gs.print_xylist(out_pref + '_S2.dat', resXH, (S2.T) * args.zeta, True)
        print("      ...complete.")
    print("= = Finished. Total seconds elapsed: %g" % (time.time() - time_start))
    srdist.finish()


if __name__ == '__main__':
    main()
