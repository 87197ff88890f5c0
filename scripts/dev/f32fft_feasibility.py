#!/usr/bin/env python3
"""
Development study (build container only; needs /root/reference + oracle/_ref): does a FLOAT32 Wiener-Khinchin C(t) --
signals mean-removed per (chunk, signal), transformed in float32, mean terms restored in float64 -- keep the reference
CHAIN (fit order selection, R1/R2/NOE) within 1e-6 of the float64 chain?  Emulated with scipy's float32 transforms.

    python scripts/dev/f32fft_feasibility.py cfg2full|cfg3s|cfg4s [traceless]
"""
import os
import sys

import numpy as np
import scipy.fft as sfft

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

from spinrelax_amd import synth   # noqa: E402


def ct_f32fft(vecs, R, F, L, traceless=True, noise=0.0):
    """vecs (frames, V, 3) float32 -> raw sums S (V, R, L+1) float64 by the float32 transform."""
    V = vecs.shape[1]
    M = 6144 if F + L > 4096 else (4096 if F + L > 2048 else 2048)
    S = np.zeros((V, R, L + 1))
    lag = np.arange(L + 1)
    for v in range(V):
        u = vecs[:R * F, v, :].reshape(R, F, 3)
        x, y, z = (u[..., k] for k in range(3))
        x64, y64, z64 = (a.astype(np.float64) for a in (x, y, z))
        if traceless:
            sigs = [(2 * z64 * z64 - x64 * x64 - y64 * y64, 1.0 / 6.0), (x64 * x64 - y64 * y64, 0.5), (x64 * y64, 2.0),
                    (x64 * z64, 2.0), (y64 * z64, 2.0)]
            eps = (x64 * x64 + y64 * y64 + z64 * z64) - 1.0
            e = eps / 3.0
            const = (F - lag)[None, :] / 3.0 * np.ones((R, 1))
        else:
            sigs = [(x64 * x64, 1.0), (y64 * y64, 1.0), (z64 * z64, 1.0), (x64 * y64, 2.0), (x64 * z64, 2.0), (y64 * z64, 2.0)]
            e = np.zeros((R, F))
            const = np.zeros((R, L + 1))
        P = np.zeros((R, M // 2 + 1), dtype=np.float32)
        for a, w in sigs:
            m = a.mean(axis=1, keepdims=True).astype(np.float32)           # any constant works: float32 mean
            d = (a - m.astype(np.float64)).astype(np.float32)               # rounded once (the kernel: fma chain)
            D = sfft.rfft(d, n=M, axis=1)
            assert D.dtype == np.complex64
            P += np.float32(w) * (D.real * D.real + D.imag * D.imag)
            m64 = m.astype(np.float64)
            e = e + w * m64 * d.astype(np.float64)
            const = const + w * (m64 * m64) * (F - lag)[None, :]
        s = sfft.irfft(P, n=M, axis=1)
        assert s.dtype == np.float32
        if noise:
            s = s * (1 + noise * np.random.default_rng(v).standard_normal(s.shape)).astype(np.float32)
        # mean terms: PE[F - d] + (PE[F] - PE[d]),  PE = prefix sums of e
        PE = np.concatenate([np.zeros((R, 1)), np.cumsum(e, axis=1)], axis=1)
        corr = PE[:, F - lag] + (PE[:, F:F + 1] - PE[:, lag])
        S[v] = s[:, :L + 1].astype(np.float64) + corr + const
    return S


def finalize(S, F, L):
    """calculate-Ct-from-traj.py:226-228"""
    lag = np.arange(1, L + 1)
    p = 1.5 * (S[:, :, 1:] / (F - lag)[None, None, :]) - 0.5          # (V, R, L)
    Ct = p.mean(axis=1).T
    R = S.shape[1]
    dCt = (p.std(axis=1) / (np.sqrt(R) - 1.0)).T
    return Ct, dCt


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'cfg3s'
    tr = len(sys.argv) > 2 and sys.argv[2] == 'traceless'
    noise = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    import gen_golden_chain as gc
    gg = gc.gg
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', '%s_chain.npz' % tag), allow_pickle=True))
    cfg = {'cfg2': 2, 'cfg2full': 2, 'cfg3s': 3}[tag]
    s = synth.config_shapes(cfg)
    nvec = {'cfg2': 16, 'cfg2full': None, 'cfg3s': 8}[tag]
    vecs = synth.synth_config(cfg, nvec=nvec)
    V = vecs.shape[1]
    S = ct_f32fft(vecs, s['R'], s['F'], s['L'], traceless=tr, noise=noise)
    Ct, dCt = finalize(S, s['F'], s['L'])
    eC = np.max(np.abs(Ct / g['Ct64'] - 1.0))
    eD = np.max(np.abs(dCt - g['dCt64']))
    eDr = np.max(np.abs(dCt - g['dCt64']) / np.abs(g['dCt64']))
    print('[%s tr=%d] C(t) rel %.2e   dC(t) abs %.2e rel %.2e  (min dCt %.2e)' % (tag, tr, eC, eD, eDr, g['dCt64'].min()))
    names = [int(x) for x in g['names']]
    t = g['t']
    ac, res = gc.fit_in_memory(names, t, Ct, dCt)
    S2, C, tau = gc.params_of(ac)
    hist = g['hist']
    v4 = gg.quiet(gc.ref.calcCt.reformat_vecs_by_tau, [vecs], s['dt'], s['tau_memory'])
    _, _, _, hist2, edges, _, _ = gg.ref_vec_stage(s, v4, synth.Q_EXT)
    assert np.array_equal(hist2, hist)
    tab = gc.ref_table_f64(S2, C, tau, hist, edges, names, synth.FIELD_MHZ, synth.ZETA, synth.DISO, synth.DANI)
    same = res['sel_nParams'] == g['mem_sel_nParams']
    rel_sym = np.abs(tab['sym64'][:3, :, 0] / g['mem_sym64'][:3, :, 0] - 1.0).max(axis=0)
    rel_iso = np.abs(tab['iso64'][:3] / g['mem_iso64'][:3] - 1.0).max(axis=0)
    chi = np.abs(res['sel_chi'] / g['mem_sel_chi'] - 1.0)
    print('[%s] same order %d/%d   R sym: max %.2e beyond1e-6 %d   iso: max %.2e beyond %d   chi rel max %.2e' % (
        tag, same.sum(), V, rel_sym.max(), (rel_sym > 1e-6).sum(), rel_iso.max(), (rel_iso > 1e-6).sum(), chi[same].max()))
    bad = np.where((rel_sym > 1e-6) | ~same)[0]
    for i in bad:
        print('   residue %d: nP %d vs %d  rel %.2e chi %.6g vs %.6g trial chi %s | %s' % (
            i, res['sel_nParams'][i], g['mem_sel_nParams'][i], rel_sym[i], res['sel_chi'][i], g['mem_sel_chi'][i],
            res['trial_chi'][i], g['mem_trial_chi'][i]))


if __name__ == '__main__':
    main()
