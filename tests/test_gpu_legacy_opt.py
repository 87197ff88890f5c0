"""
GPU tests of the legacy single-field optimisation modes of calculate-relaxations-from-Ct.py (SURVEY.md section 8(a)
row 20; reference lines :193-316, :775-1004).  Expected numbers come from the reference's own objective functions
driven by scipy's Powell search (tests/golden/cfg1_legacy_opt.npz, oracle/gen_golden_legacy.py).
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, golden
from spinrelax_amd import synth

pytestmark = pytest.mark.gpu
SCR = os.path.join(ROOT, 'scripts')
B0_HZ = 600.133e6


def _setup(Diso):
    from spinrelax_amd import spectral_densities as sd
    from spinrelax_amd import fitting_Ct_functions as fitCt
    ac = fitCt.read_fittedCt_parameters(os.path.join(GOLD, 'cfg1_fittedCt.dat'))
    S2, consts, taus, _ = ac.get_params_as_list()
    for i in range(ac.nModels):
        S2[i] *= synth.ZETA
        consts[i] *= synth.ZETA
    _, vecXH, w = sd.read_vector_distribution_from_file(os.path.join(GOLD, 'cfg1_vecHistogram.npz'))
    RObj = sd.relaxationModel('NH', 2.0 * np.pi * B0_HZ / 267.513e6)
    RObj.set_time_unit('ps')
    Dperp = 3. * Diso / (2 + synth.DANI)
    RObj.set_rotdif_model('rigid_symmtop_D', synth.DANI * Dperp, Dperp)
    return RObj, ac.nModels, S2, consts, taus, vecXH, w


def _expblock():
    from spinrelax_amd import general_scripts as gs
    from spinrelax_amd import legacy_opt
    return legacy_opt.read_experiment(os.path.join(GOLD, 'cfg1_legacy_exp.dat'), 'rigid_symmtop', gs.load_xys)


def test_objective_functions_equal_the_reference(capsys):
    from spinrelax_amd import legacy_opt as lo
    g = golden('cfg1_legacy_opt.npz')
    exp_resid, expblock = _expblock()
    assert expblock.shape == (3, 32, 2) and [int(x) for x in exp_resid] == list(g['resid'])
    np.testing.assert_allclose(expblock[..., 0], g['exp_values'], rtol=1e-5)
    csa0 = np.repeat(-170e-6, 32)
    grid = []
    for sc in (0.9, 1.0, 1.1):
        RObj, n, S2, consts, taus, vecXH, w = _setup(synth.DISO)
        grid.append([lo.optfunc_R1R2NOE_Diso([synth.DISO * sc], RObj, n, S2, consts, taus, vecXH, w, csa0, expblock),
                     lo.optfunc_R1R2NOE_DisoS2([synth.DISO * sc, 0.95], RObj, n, S2, consts, taus, vecXH, w, csa0, expblock),
                     lo.optfunc_R1R2NOE_DisoCSA([synth.DISO * sc, -160e-6], RObj, n, S2, consts, taus, vecXH, w, expblock),
                     lo.optfunc_R1R2NOE_DisoS2CSA([synth.DISO * sc, 0.95, -160e-6], RObj, n, S2, consts, taus, vecXH, w, expblock)])
    # float32 datablocks on both sides: one float32 ulp of a rate moves chi^2 by ~1e-5 relative at most
    np.testing.assert_allclose(np.array(grid), g['objective_grid'], rtol=2e-5)
    RObj, n, S2, consts, taus, vecXH, w = _setup(synth.DISO)
    mine = np.array([[lo.optfunc_R1R2NOE_new([c], RObj, S2[i], consts[i], taus[i], vecXH[i], w[i], expblock[:, i, :])
                      for c in (-150e-6, -170e-6, -190e-6)] for i in range(4)])
    np.testing.assert_allclose(mine, g['objective_new_res'], rtol=2e-5)
    assert '= = optimisations params(' in capsys.readouterr().out


def _run(tmp_path, mode, extra=()):
    out = str(tmp_path / ('opt_' + mode))
    cmd = [sys.executable, os.path.join(SCR, 'calculate-relaxations-from-Ct.py'), '-f', os.path.join(GOLD, 'cfg1_fittedCt.dat'),
           '-o', out, '--distfn', os.path.join(GOLD, 'cfg1_vecHistogram.npz'), '-F', str(B0_HZ), '--tu', 'ps',
           '-D', '%.10g %g' % (synth.DISO, synth.DANI), '--opt', mode, '--expfn', os.path.join(GOLD, 'cfg1_legacy_exp.dat')] + list(extra)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
    hdr = {}
    with open(out + '_R1.dat') as fp:
        for line in fp:
            m = re.match(r'# (Optimised|Fixed) (\w+): (\S+)', line)
            if m:
                hdr[m.group(2)] = (m.group(1), float(m.group(3)))
    return out, hdr, p.stdout.decode()


@pytest.mark.parametrize('mode', ['Diso', 'DisoS2', 'DisoCSA', 'DisoS2CSA'])
def test_legacy_global_modes_reach_the_reference_optimum(tmp_path, mode):
    g = golden('cfg1_legacy_opt.npz')
    out, hdr, log = _run(tmp_path, mode)
    x, chi = np.ravel(g[mode + '_x']), float(g[mode + '_chi'])
    assert hdr['Diso'][0] == 'Optimised' and abs(hdr['Diso'][1] / x[0] - 1) < 2e-4
    assert abs(hdr['chi'][1] / np.sqrt(chi) - 1) < 1e-3
    if 'S2' in mode:
        # the header scales the S2 factor by zeta (param_scaling, calculate-relaxations-from-Ct.py:754)
        assert hdr['zeta'][0] == 'Optimised' and abs(hdr['zeta'][1] / (synth.ZETA * x[1]) - 1) < 2e-4
    if 'CSA' in mode:
        assert hdr['CSA'][0] == 'Optimised' and abs(hdr['CSA'][1] / (1e6 * x[-1]) - 1) < 5e-4
    if mode == 'Diso':
        # the table written after the fit is the reference's datablock at the optimum
        from spinrelax_amd import general_scripts as gs
        _, r1 = gs.load_xys(out + '_R1.dat')
        np.testing.assert_allclose(np.asarray(r1)[:, 0], g['Diso_datablock'][0, :, 0], rtol=3e-4)


def test_legacy_mode_new(tmp_path):
    from spinrelax_amd import general_scripts as gs
    g = golden('cfg1_legacy_opt.npz')
    out, hdr, log = _run(tmp_path, 'new', ['--cycles', '3'])
    assert '= = = BREAK at CSA test' in log              # the reference's aliasing quirk ends the refinement in round 1
    assert abs(hdr['Diso'][1] / float(g['new_Diso'][0]) - 1) < 3e-4
    resid, csa = gs.load_xy(out + '_CSA_values.dat')
    assert [int(x) for x in resid] == list(g['resid'])
    np.testing.assert_allclose(np.asarray(csa, dtype=float), g['new_csa'], rtol=2e-3)


def test_device_csa_search_equals_the_host_powell_loop():
    """The per-residue CSA refinement of `--opt new` (calculate-relaxations-from-Ct.py:983-989: one fmin_powell over
    optfunc_R1R2NOE_new per residue) as ONE launch (sr_legacy_csa_search_f64) against the host loop it replaces -- scipy's
    fmin_powell driving the same objective through the relaxation kernel, the reference's structure.  Same arithmetic in the
    objective (the bins are reduced by the code k_relax runs) and scipy's Powell restated operation for operation: the optimum,
    the chi^2 there and the NUMBER OF OBJECTIVE CALLS must be identical for every residue, from the default CSA and from
    perturbed starts."""
    from scipy.optimize import fmin_powell
    from spinrelax_amd import legacy_opt as lo
    from spinrelax_amd.hip import SpinRelaxHipError
    RObj, n, S2, consts, taus, vecXH, w = _setup(synth.DISO)
    exp_resid, expblock = _expblock()
    assert lo._device_csa_search_applies(RObj, vecXH, w, expblock)
    for scale in (1.0, 0.8, 1.3):
        csa0 = np.repeat(-170e-6 * scale, n)
        csa_d, chi_d, nfev_d = lo.csa_search_device(RObj, S2, consts, taus, vecXH, w, expblock, csa0)
        ncheck = n if scale == 1.0 else 8
        for i in range(ncheck):
            out = fmin_powell(lo.optfunc_R1R2NOE_new, x0=csa0[i], args=(RObj, S2[i], consts[i], taus[i], vecXH[i], w[i], expblock[:, i, :]),
                              full_output=True, disp=False)
            assert np.ravel(out[0])[0] == csa_d[i], (scale, i, out[0], csa_d[i])
            assert float(out[1]) == chi_d[i], (scale, i, out[1], chi_d[i])
            assert int(out[4]) == int(nfev_d[i]), (scale, i, out[4], nfev_d[i])
        assert np.all(nfev_d > 5) and np.all(np.isfinite(chi_d))
    # the objective itself at arbitrary points: Powell's first call is f(x0)
    for c in (-150e-6, -170e-6, -190e-6):
        _, chi0, _ = lo.csa_search_device(RObj, S2, consts, taus, vecXH, w, expblock, np.repeat(c, n))
        assert np.all(np.isfinite(chi0))
    # argument validation happens before any launch
    ctx = lo._ctx(None)
    S2a, C, T, K = lo.sd._pack(S2, consts, taus)
    D = [RObj.rotdifModel.D[0], RObj.rotdifModel.D[1]]
    ex = np.ascontiguousarray(np.swapaxes(expblock, 0, 1))
    gb = (RObj.gX.gamma * RObj.B_0) ** 2
    with pytest.raises(ValueError):
        ctx.legacy_csa_search(D, RObj.omega, RObj.get_f_DD(), gb, RObj.time_fact, 1.0, S2a, C, T, K, np.asarray(vecXH)[0], np.asarray(w)[:3], ex, csa0)
    with pytest.raises(SpinRelaxHipError):
        ctx.legacy_csa_search(D, RObj.omega, RObj.get_f_DD(), gb, RObj.time_fact, 1.0, S2a, C, T, K + 9, np.asarray(vecXH)[0], np.asarray(w), ex, csa0)
    # the switch back to the host loop
    os.environ['SR_LEGACY_HOST_SEARCH'] = '1'
    try:
        assert not lo._device_csa_search_applies(RObj, vecXH, w, expblock)
    finally:
        del os.environ['SR_LEGACY_HOST_SEARCH']


def test_legacy_opt_argument_errors(tmp_path):
    base = [sys.executable, os.path.join(SCR, 'calculate-relaxations-from-Ct.py'), '-f', 'x', '--opt', 'Diso']
    p = subprocess.run(base, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b'Missing --expfn' in p.stderr
    p = subprocess.run(base[:-1] + ['Bogus', '--expfn', 'y'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b'Invalid optimisation mode' in p.stderr
