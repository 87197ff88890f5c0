"""
Small quaternion helpers the global-rotational-diffusion analysis needs on the host.

The reference delegates these to the third-party package `transforms3d` (requirements.txt:5, unpinned; NOT installed in
this image, so the reference's own quat_frame_transform_min cannot be executed here and its parity is pinned by
properties only -- tests/test_formats_and_hostlogic.py::test_quat_frame_transform_*).  What follows restates the
published definitions of transforms3d.quaternions (qmult, qconjugate, rotate_vector, axangle2quat, nearly_equivalent,
mat2quat -- Bar-Itzhack's eigenvector method) and the reference's own compositions of them
(transforms3d_supplement.py:71-83 quat_v1v2, :137-149 quat_frame_transform_min).  Quaternions are (w, x, y, z).
"""
import math

import numpy as np


def qeye():
    return np.array([1.0, 0.0, 0.0, 0.0])


def qmult(q1, q2):
    w1, x1, y1, z1 = q1
    w2, x2, y2, z2 = q2
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                     w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2])


def qconjugate(q):
    q = np.asarray(q, dtype=float)
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qinverse(q):
    q = np.asarray(q, dtype=float)
    return qconjugate(q) / np.dot(q, q)


def rotate_vector(v, q):
    """v rotated by q: vector part of q (0, v) q*."""
    varr = np.zeros(4)
    varr[1:] = v
    return qmult(q, qmult(varr, qconjugate(q)))[1:]


def axangle2quat(vector, theta, is_normalized=False):
    vector = np.asarray(vector, dtype=float)
    if not is_normalized:
        vector = vector / math.sqrt(np.dot(vector, vector))
    t2 = theta / 2.0
    return np.concatenate(([math.cos(t2)], vector * math.sin(t2)))


def nearly_equivalent(q1, q2, rtol=1e-5, atol=1e-8):
    q1 = np.asarray(q1, dtype=float)
    q2 = np.asarray(q2, dtype=float)
    return bool(np.allclose(q1, q2, rtol, atol) or np.allclose(q1 * -1, q2, rtol, atol))


def mat2quat(M):
    """Rotation matrix -> quaternion by the largest eigenvector of Bar-Itzhack's K matrix (robust to slightly
    non-orthogonal input, as gmx rotmat output is)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, dtype=float).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = q * -1
    return q


def quat_v1v2(v1, v2):
    """Minimum-angle rotation taking v1 onto v2 (transforms3d_supplement.py:71-83; identical vectors give a NaN axis
    after normalisation there and the identity here and there)."""
    th = math.acos(np.dot(v1, v2))
    ax = np.cross(v1, v2)
    if all(np.isnan(ax)):
        return qeye()
    return axangle2quat(ax, th)


def quat_frame_transform_min(axes):
    """Rotation that brings the frame `axes` (rows = x, y, z axes) onto the coordinate axes, choosing for z and then x
    the nearer of the two senses (transforms3d_supplement.py:137-149)."""
    q1a = quat_v1v2(axes[2], (0, 0, 1))
    q1b = quat_v1v2(axes[2], (0, 0, -1))
    q1 = q1a if q1a[0] > q1b[0] else q1b
    arot = [rotate_vector(axes[i], q1) for i in range(3)]
    q2a = quat_v1v2(arot[0], (1, 0, 0))
    q2b = quat_v1v2(arot[0], (-1, 0, 0))
    q2 = q2a if q2a[0] > q2b[0] else q2b
    return qmult(q2, q1)


def rotation_matrix(q):
    """3 x 3 matrix R with R v = rotate_vector(v, q) for a unit quaternion."""
    w, x, y, z = np.asarray(q, dtype=float) / math.sqrt(np.dot(q, q))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
