"""
Small host-side (numpy) set-up maths: quantities computed ONCE per run from a handful of numbers
(diffusion-tensor coefficients, the 2592 bin-centre vectors of the Lambert histogram).  Nothing here is
on the hot path; the per-frame / per-residue / per-bin loops are all in the HIP kernels.
"""
import numpy as np


def D_coefficients_symmtop(D):
    """spectral_densities.py:1874-1884: D = (Dpar, Dperp) -> [5Dperp+Dpar, 2Dperp+4Dpar, 6Dperp]."""
    Dpar, Dperp = D[0], D[1]
    return np.array([5 * Dperp + Dpar, 2 * Dperp + 4 * Dpar, 6 * Dperp])


def A_coefficients_symmtop(v, bProlate=True):
    """spectral_densities.py:1886-1906."""
    v = np.asarray(v)
    z2 = np.square(v[..., 2] if bProlate else v[..., 0])
    w = 1 - z2
    return np.stack((3.0 * (z2 * w), 0.75 * np.square(w), 0.25 * np.square(3 * z2 - 1)), axis=-1)


def symmtop_from_iso(Diso, aniso):
    """calculate-relaxations-from-Ct.py:621-622 -> (Dpar, Dperp)."""
    Dperp = 3. * Diso / (2 + aniso)
    return aniso * Dperp, Dperp


def rtp_to_xyz_unit(pt):
    """general_maths.py:176-180 (bUnit=True, vaxis=-1): (..., [phi, theta]) -> unit vectors."""
    pt = np.asarray(pt)
    uv = np.zeros(pt.shape[:-1] + (3,), dtype=pt.dtype)
    uv[..., 0] = np.cos(pt[..., 0]) * np.sin(pt[..., 1])
    uv[..., 1] = np.sin(pt[..., 0]) * np.sin(pt[..., 1])
    uv[..., 2] = np.cos(pt[..., 1])
    return uv


def lambert_bin_vectors(edges):
    """spectral_densities.py:2338-2341: bin-centre directions of the (phi, cos theta) histogram,
    flattened phi-major -> (nphi*ncos, 3)."""
    phis = 0.5 * (edges[0][:-1] + edges[0][1:])
    thetas = np.arccos(0.5 * (edges[1][:-1] + edges[1][1:]))
    pt = np.moveaxis(np.array(np.meshgrid(phis, thetas, indexing='ij')), 0, -1)
    bv = rtp_to_xyz_unit(pt)
    return bv.reshape(bv.shape[0] * bv.shape[1], 3)
