"""
Host-side mirror of the C(t) / vector-distribution functions of the reference's
calculate-Ct-from-traj.py, with the same names, argument meaning and error behaviour, computing on the
MI355X through libspinrelax_hip.so.  There is no CPU compute path here: index arithmetic and
formatting only; the arithmetic of the hot loops runs in the HIP kernels.
"""
import sys

import numpy as np

from . import hip
from . import dist as srdist


def _ctx(ctx):
    return ctx if ctx is not None else hip.default_context()


def calculate_dt(dt, tau):
    """calculate-Ct-from-traj.py:240-243: the lag axis (arange(int(0.5*tau/dt)) + 1) * dt."""
    nPts = int(0.5 * tau / dt)
    return (np.arange(nPts) + 1.0) * dt


def concat_with_chunk_starts(vec_list, F):
    """Index form of reformat_vecs_by_tau (calculate-Ct-from-traj.py:245-275): instead of copying the
    used frames into a (R, F, V, 3) array, return the concatenation of the files, the start frame of
    each of the R chunks and R.  The tail of every file that does not fill a chunk is skipped."""
    starts = []
    off = 0
    for v in vec_list:
        n = v.shape[0]
        for c in range(n // F):
            starts.append(off + c * F)
        off += n
    cat = vec_list[0] if len(vec_list) == 1 else np.concatenate(vec_list, axis=0)
    return np.ascontiguousarray(cat, dtype=np.float32), np.array(starts, dtype=np.int64), len(starts)


def reformat_vecs_by_tau(vecs, dt, tau):
    """calculate-Ct-from-traj.py:245-275, same signature and prints: list of (frames, bonds, 3) arrays ->
    (nchunk, frames_per_chunk, bonds, 3).  Kept for callers that want the 4-D array; the GPU path itself
    uses concat_with_chunk_starts and never materialises it."""
    nFiles = len(vecs)
    nFramesPerChunk = int(tau / dt)
    print("    ...debug: Using %i frames per chunk based on tau/dt (%g/%g)." % (nFramesPerChunk, tau, dt))
    kept = []
    for i in range(nFiles):
        nFrames = vecs[i].shape[0]
        used = int(nFrames / nFramesPerChunk) * nFramesPerChunk
        print("    ...Source %i divided into %i chunks. Usage rate: %g %%" % (i, used / nFramesPerChunk, 100.0 * used / nFrames))
        kept.append(vecs[i][0:used, ...])
    out = np.concatenate(kept, axis=0)
    nTot = out.shape[0]
    print("    ...Done. vecs reformatted into %i chunks." % (nTot / nFramesPerChunk))
    return out.reshape((int(nTot / nFramesPerChunk), nFramesPerChunk, out.shape[-2], out.shape[-1]))


def calculate_Ct_Palmer(vecs, ctx=None, mode=0, v0=0, nV=None):
    """calculate-Ct-from-traj.py:200-238.  vecs (nReplicates, nFrames, nResidues, 3) -> Ct, dCt of shape
    (nDeltas, nResidues), float64 (the reference returns the dtype of vecs; the values here are the
    float64 evaluation, see DESIGN.md "precision")."""
    sh = vecs.shape
    print("= = = Debug of calculate_Ct_Palmer confirming the dimensions of vecs:", sh)
    if len(sh) != 4:
        print("= = = ERROR: The input vectors to calculate_Ct_Palmer is not of the expected 4-dimensional form! ", sh,
              file=sys.stderr)
        sys.exit(1)
    if sh[1] < 50:
        print("= = = WARNING: there are less than 50 frames per block of memory-time!", file=sys.stderr)
    R, F, V = sh[0], sh[1], sh[2]
    flat = np.ascontiguousarray(vecs, dtype=np.float32).reshape(R * F, V, 3)
    return _ctx(ctx).ct_palmer(flat, R, F, v0=v0, nV=nV, mode=mode)


def calculate_Ct_from_files(vec_list, dt, tau, ctx=None, mode=0, v0=0, nV=None):
    """reformat_vecs_by_tau + calculate_Ct_Palmer without the intermediate copy."""
    F = int(tau / dt)
    cat, starts, R = concat_with_chunk_starts(vec_list, F)
    if R < 1:
        print("= = = ERROR: no trajectory holds a full block of memory time tau!", file=sys.stderr)
        sys.exit(1)
    if srdist.world() > 1 and nV is None and v0 == 0:
        # several ranks (torchrun): vectors are independent -- every rank computes its contiguous range of columns, all
        # ranks receive the whole (lags, vectors) arrays
        V = cat.shape[1]
        i0, nloc = srdist.my_range(V)
        L = F // 2
        if nloc > 0:
            Ct, dCt = _ctx(ctx).ct_palmer(cat, R, F, v0=i0, nV=nloc, chunk_start=starts, mode=mode)
        else:
            Ct, dCt = np.empty((L, 0)), np.empty((L, 0))
        return srdist.gather_rows(Ct, V, axis=1), srdist.gather_rows(dCt, V, axis=1)
    return _ctx(ctx).ct_palmer(cat, R, F, v0=v0, nV=nV, chunk_start=starts, mode=mode)


def upload_shard(vec_list, frames_per_chunk=None, ctx=None):
    """The product path's single upload: this rank's vector range (srdist.my_range; everything in a single process) of the
    files' vectors, each file cut to whole chunks of frames_per_chunk frames when given (reformat_vecs_by_tau,
    calculate-Ct-from-traj.py:245-275), appended one after the other to a hip.ResidentVectors object.  Only the rank's
    columns cross PCIe, once; C(t) and the vector distribution then read the same planes.
    Returns (resident vectors, total vectors V, first vector i0, frames held)."""
    c = _ctx(ctx)
    V = vec_list[0].shape[1]
    i0, nloc = srdist.my_range(V) if srdist.world() > 1 else (0, V)
    used = []
    for v in vec_list:
        n = v.shape[0] if frames_per_chunk is None else (v.shape[0] // frames_per_chunk) * frames_per_chunk
        if n > 0:
            used.append((v, n))
    if not used:
        print("= = = ERROR: no trajectory holds a full block of memory time tau!", file=sys.stderr)
        sys.exit(1)
    N = sum(n for _, n in used)
    if nloc == 0:
        return None, V, i0, N
    rv = c.vectors(nloc, N)
    for v, n in used:
        rv.append(v[:n], v0=i0)
    return rv, V, i0, N


def chunk_table(vec_list, F):
    """(file index, first frame) of every whole chunk of F frames, in the order reformat_vecs_by_tau concatenates them."""
    return [(i, c * F) for i, v in enumerate(vec_list) for c in range(v.shape[0] // F)]


def calculate_Ct_chunk_sharded(vec_list, F, mode=0, ctx=None, sums_fn=None, finalize_fn=None):
    """C(t) with the replicate CHUNKS sharded over the ranks (fewer vectors than GPUs): this rank uploads the frames of its
    chunk range -- all vectors --, computes their raw sums, the (V, R, L) sums of all ranks are gathered and every rank
    runs the single-process mean / std kernel on them: the same bits as one process.
    sums_fn(list of (F, V, 3) chunks) -> (V, n, L) and finalize_fn((V, R, L)) -> (Ct, dCt) replace the GPU calls in the
    CPU test of the sharding logic (tests/test_cabi_and_dist.py)."""
    table = chunk_table(vec_list, F)
    R = len(table)
    if R < 1:
        print("= = = ERROR: no trajectory holds a full block of memory time tau!", file=sys.stderr)
        sys.exit(1)
    V = vec_list[0].shape[1]
    L = F // 2
    r0, nR = srdist.my_chunk_range(R)
    mine = [vec_list[i][f0:f0 + F] for i, f0 in table[r0:r0 + nR]]
    if nR == 0:
        sums = np.empty((V, 0, L))
    elif sums_fn is not None:
        sums = sums_fn(mine)
    else:
        with _ctx(ctx).vectors(V, nR * F) as rv:
            for c in mine:
                rv.append(c)
            sums = rv.ct_sums(nR, F, mode=mode)
    sums = srdist.gather_chunk_axis(sums, R)
    if finalize_fn is not None:
        return finalize_fn(sums)
    return _ctx(ctx).ct_finalize_sums(sums, F)


def calculate_Ct_resident(rv, V, R, F, mode=0):
    """calculate_Ct_Palmer of resident vectors (regular chunks r*F: the unused tails were dropped at upload); all ranks
    receive the whole (lags, V) arrays."""
    L = F // 2
    if rv is not None:
        Ct, dCt = rv.ct(R, F, mode=mode)
    else:
        Ct, dCt = np.empty((L, 0)), np.empty((L, 0))
    if srdist.world() > 1:
        return srdist.gather_rows(Ct, V, axis=1), srdist.gather_rows(dCt, V, axis=1)
    return Ct, dCt


def vector_distribution_resident(rv, V, N, q_rot=None, histBinX=72, delta_t=-1, tau_memory=-1):
    """vector_distribution() on resident vectors: same result dictionary."""
    edges = lambert_edges(histBinX)
    Fb = 0 if (delta_t < 0 or tau_memory < 0) else int(tau_memory / delta_t)
    if rv is not None:
        hist, vecsum, outer = rv.hist(q_rot, edges[0], edges[1], block_len=Fb, N_hist=N)
    else:
        nB = N // Fb if Fb else 1
        hist, vecsum, outer = np.empty((0, histBinX, int(histBinX / 2))), np.empty((0, 3)), np.empty((nB, 0, 6))
    if srdist.world() > 1:
        hist, vecsum = srdist.gather_rows(hist, V, axis=0), srdist.gather_rows(vecsum, V, axis=0)
        outer = srdist.gather_rows(outer, V, axis=1)
    mean = vecsum / N
    avgvec = mean / np.sqrt((mean ** 2).sum(-1))[..., np.newaxis]
    return dict(hist=hist, edges=edges, avgvec=avgvec, S2=S2_from_outer_sums(outer, Fb if Fb else N, blocked=bool(Fb)))


def lambert_edges(histBinX=72):
    """The bin edges numpy.histogramdd builds at calculate-Ct-from-traj.py:618 for
    bins=(histBinX, int(histBinX/2)), range=((-pi,pi),(-1,1))."""
    histBinY = int(histBinX / 2)
    return [np.linspace(-np.pi, np.pi, histBinX + 1), np.linspace(-1.0, 1.0, histBinY + 1)]


def vector_distribution(vecs, q_rot=None, histBinX=72, delta_t=-1, tau_memory=-1, ctx=None, v0=0, nV=None):
    """One pass replacing calculate-Ct-from-traj.py:541-646 for the Histogram storage mode:
    rotation into the PAF (rotate_vector_simd), mean vector (:579-583), Lambert-cylindrical histogram
    (:585-626) and calculate_S2_by_outerProduct (:96-145).

    vecs (N, V, 3) float32 (the 3-D array after the 4-D -> 3-D reshape at :535-536).
    Returns dict(hist (V,X,Y) float64, edges [X+1, Y+1], avgvec (V,3), S2 (V,2) or (V,))."""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    N = vecs.shape[0]
    edges = lambert_edges(histBinX)
    if delta_t < 0 or tau_memory < 0:
        Fb = 0
    else:
        Fb = int(tau_memory / delta_t)
    if srdist.world() > 1 and nV is None and v0 == 0:
        V = vecs.shape[1]
        i0, nloc = srdist.my_range(V)
        if nloc > 0:
            hist, vecsum, outer = _ctx(ctx).rotate_hist(vecs, q_rot, edges[0], edges[1], v0=i0, nV=nloc, block_len=Fb)
        else:
            nB = N // Fb if Fb else 1
            hist, vecsum, outer = np.empty((0, histBinX, int(histBinX / 2))), np.empty((0, 3)), np.empty((nB, 0, 6))
        hist, vecsum = srdist.gather_rows(hist, V, axis=0), srdist.gather_rows(vecsum, V, axis=0)
        outer = srdist.gather_rows(outer, V, axis=1)
    else:
        hist, vecsum, outer = _ctx(ctx).rotate_hist(vecs, q_rot, edges[0], edges[1], v0=v0, nV=nV, block_len=Fb)
    mean = vecsum / N
    avgvec = mean / np.sqrt((mean ** 2).sum(-1))[..., np.newaxis]
    return dict(hist=hist, edges=edges, avgvec=avgvec, S2=S2_from_outer_sums(outer, Fb if Fb else N, blocked=bool(Fb)))


def S2_from_outer_sums(outer, frames_per_block, blocked=True):
    """calculate_S2_by_outerProduct (calculate-Ct-from-traj.py:125-142) from per-block sums of
    xx,yy,zz,xy,xz,yz: S2 = 1.5*sum_ij <u_i u_j>^2 - 0.5; with blocks: mean and std/(sqrt(nBlocks)-1)."""
    m = outer / frames_per_block
    s = 1.5 * (m[..., 0] ** 2 + m[..., 1] ** 2 + m[..., 2] ** 2
               + 2.0 * (m[..., 3] ** 2 + m[..., 4] ** 2 + m[..., 5] ** 2)) - 0.5
    if not blocked:
        return s[0]
    nBlocks = s.shape[0]
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.stack((np.mean(s, axis=0), np.std(s, axis=0) / (np.sqrt(nBlocks) - 1.0)), axis=-1)


def vecnorm_NDarray(v, axis=-1):
    """transforms3d_supplement.py:40-52: normalise along an axis, 0/0 -> 0."""
    v = np.asarray(v)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.nan_to_num(v / np.linalg.norm(v, axis=axis, keepdims=True))


def rotate_vector_simd(v, q, ctx=None):
    """transforms3d_supplement.py:270-296 for v (N, V, 3) float32 and either one quaternion q (4,) or one per
    frame, q (N, 4) / (N, 1, 4) (numpy broadcasting over the vector axis): float64 result, rotation on the GPU."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    shp = v.shape
    q = np.asarray(q, dtype=np.float64)
    if q.ndim == 3 and q.shape[1] == 1:
        q = q[:, 0, :]
    if q.ndim == 2:
        q = vecnorm_NDarray(q)
    out = _ctx(ctx).rotate_vectors(v.reshape(-1, 1, 3) if v.ndim == 2 else v, q)
    return out.reshape(shp)


def detumble_vectors(vecs_lab, q_orient, ctx=None):
    """Body-frame vectors from lab-frame vectors and the orientation trajectory q_orient(t) (PLUMED colvar-qorient,
    float32 fields, plumedcolvario.py:24-81): v_body(t) = R(q(t))^-1 v_lab(t), i.e. rotate_vector_simd with the
    conjugate quaternion per frame.  Returns float32 like the vectors the superposition route produces
    (calculate-Ct-from-traj.py:64-86, 466-467)."""
    q = np.array(q_orient, dtype=np.float64)
    if q.ndim != 2 or q.shape[1] != 4 or q.shape[0] != np.shape(vecs_lab)[0]:
        raise ValueError('detumble_vectors: need one quaternion (w x y z) per frame, got %s for %d frames'
                         % (q.shape, np.shape(vecs_lab)[0]))
    q[:, 1:] *= -1.0
    return rotate_vector_simd(vecs_lab, q, ctx=ctx).astype(np.float32)


def obtain_XHvecs(xyz, indexX, indexH, ctx=None, bSuppressPrint=False):
    """obtain_XHvecs (calculate-Ct-from-traj.py:64-86) on a coordinate array instead of an MDTraj trajectory: xyz is
    traj.xyz (frames, atoms, 3) float32, indexX / indexH what topology.select() returned for the X and H selections.
    Unit vectors (x_H - x_X)/|x_H - x_X|, float32, 0/0 -> 0, computed on the GPU (bit-identical to the numpy expression)."""
    if not bSuppressPrint:
        print("= = = Obtaining XH-vectors from trajectory...")
    numX, numH = len(indexX), len(indexH)
    if numX == 0 or numH == 0:
        print("= = = ERROR: selection text failed to find atoms!")
        print("     ....debug: N(X) = %i , N(H) = %i" % (numX, numH))
        sys.exit(1)
    if numX != numH:
        print("= = = ERROR: selection text found different number of atoms!")
        print("     ....debug: N(X) = %i , N(H) = %i" % (numX, numH))
        sys.exit(1)
    lab, _, _ = _ctx(ctx).xh_vectors(xyz, indexX, indexH)
    return lab


def superpose_XHvecs(xyz, ref_xyz, fit_indices, indexX, indexH, ctx=None, want_quat=False):
    """The reference's two obtain_XHvecs calls around trj.center_coordinates(); trj.superpose(ref, frame=0,
    atom_indices=fit_indices) (calculate-Ct-from-traj.py:462-470) in one pass over the coordinates on the GPU:
    returns (vecXH lab frame, vecXHfit after the per-frame least-squares superposition [, rotation quaternions])."""
    lab, fitv, quat = _ctx(ctx).xh_vectors(xyz, indexX, indexH, fit_indices=fit_indices, ref_xyz=ref_xyz, want_quat=want_quat)
    return (lab, fitv, quat) if want_quat else (lab, fitv)
