"""
CPU tests of the drop-in boundary and the host logic: the writers must reproduce, byte for byte, files the
REFERENCE's own writers produced (tests/golden/*.dat, made by oracle/gen_golden.py); the readers must
recover the reference's numbers; the vectorised model-order search must reproduce the reference's
selection when fed the reference's own per-order fit results.  No GPU, no oracle compute.
"""
import filecmp
import os

import numpy as np
import pytest

from conftest import GOLD, golden
from spinrelax_amd import general_scripts as gs
from spinrelax_amd import fitting_Ct_functions as fitCt
from spinrelax_amd import _hostmath as hm
from spinrelax_amd import ct as hostct
from spinrelax_amd import dist as srdist


def same(a, b):
    assert filecmp.cmp(a, b, shallow=False), 'files differ: %s %s' % (a, b)


def test_Ctint_writer_byte_exact(tmp_path):
    g = golden('cfg1_ct.npz')
    names = list(range(2, 34))
    fn = str(tmp_path / 'f64.dat')
    gs.print_sxylist(fn, names, g['t'], np.stack((g['Ct64'].T, g['dCt64'].T), axis=-1))
    same(fn, os.path.join(GOLD, 'cfg1_Ctint_f64.dat'))
    fn = str(tmp_path / 'f32.dat')
    gs.print_sxylist(fn, names, g['t'], np.stack((g['Ct32'].T, g['dCt32'].T), axis=-1))
    same(fn, os.path.join(GOLD, 'cfg1_Ctint.dat'))


def test_Ctint_reader():
    f = golden('cfg1_fit.npz')
    legs, t, y, dy = gs.load_sxydylist(os.path.join(GOLD, 'cfg1_Ctint.dat'), 'legend')
    assert [int(x) for x in legs] == list(f['names'])
    np.testing.assert_array_equal(t, f['t'])
    np.testing.assert_array_equal(y, f['y'])
    np.testing.assert_array_equal(dy, f['dy'])


def test_avgvec_S2_writers_byte_exact(tmp_path):
    g = golden('cfg1_vec.npz')
    names = list(range(2, 34))
    fn = str(tmp_path / 'avg.dat')
    gs.print_xylist(fn, names, np.array(g['avgvec']).T, True)
    same(fn, os.path.join(GOLD, 'cfg1_avgvec.dat'))
    fn = str(tmp_path / 's2.dat')
    gs.print_xylist(fn, names, (g['S2_tau'].T) * (1.02 / 1.04) ** 6, True)
    same(fn, os.path.join(GOLD, 'cfg1_S2.dat'))
    # S2 from the per-block outer-product sums (what kernel 2 returns) equals the reference formula
    dt = hostct.calculate_dt(10.0, 1000.0)
    assert dt.shape == (50,) and dt[0] == 10.0 and dt[-1] == 500.0


def _models_from_golden(f):
    ac = fitCt.autoCorrelations()
    ac.import_target_array(keys=list(f['names']), DeltaT=f['t'], Decay=f['y'], dDecay=f['dy'])
    for i, k in enumerate(f['names']):
        K, free = fitCt.split_nparams(int(f['sel_nParams'][i]))
        m = ac.add_model(k)
        m._load_fit(dict(nParams=int(f['sel_nParams'][i]), C=f['sel_C'][i, :K], tau=f['sel_tau'][i, :K], S2=f['sel_S2'][i],
                         dC=f['sel_dC'][i, :K], dtau=f['sel_dtau'][i, :K], dS2=f['sel_dS2'][i], chiSq=f['sel_chi'][i]))
    return ac


def test_fittedCt_writer_byte_exact_and_reader(tmp_path):
    f = golden('cfg1_fit.npz')
    ac = _models_from_golden(f)
    fn = str(tmp_path / 'fitted.dat')
    ac.export(fn)
    same(fn, os.path.join(GOLD, 'cfg1_fittedCt.dat'))
    back = fitCt.read_fittedCt_parameters(os.path.join(GOLD, 'cfg1_fittedCt.dat'))
    r = golden('cfg1_relax.npz')
    S2, C, tau, K = back.get_params_as_arrays(Kmax=4)
    np.testing.assert_array_equal(S2, r['S2'])
    np.testing.assert_array_equal(C, r['C'])
    np.testing.assert_array_equal(tau, r['tau'])
    np.testing.assert_array_equal(K, r['nComps'])
    assert [int(k) for k in back.model.keys()] == list(r['names'])
    # %g keeps 6 significant digits: the file is a lossy view of the fit
    assert np.max(np.abs(S2 / f['sel_S2'] - 1)) < 1e-5


def test_relaxation_writers_byte_exact(tmp_path):
    r = golden('cfg1_relax.npz')
    names = [int(x) for x in r['names']]
    hdr = "# Fixed Diso: %g ps^-1\n# Fixed zeta: %g a.u.\n# Fixed CSA: %g ppm\n# Fixed chi: %g a.u.\n" % (
        3.7383e-5, 0.890023, -170e-6 * 1e6, 0.0)
    b = r['sym_f32_1']
    fn = str(tmp_path / 'R1.dat')
    gs.print_xydy(fn, names, b[0, :, 0], b[0, :, 1], header=hdr)
    same(fn, os.path.join(GOLD, 'cfg1_sym_R1.dat'))
    fn = str(tmp_path / 'rho.dat')
    gs.print_xydy(fn, names, b[3, :, 0], b[3, :, 1])
    same(fn, os.path.join(GOLD, 'cfg1_sym_rho.dat'))
    b = r['iso_f32_1']
    fn = str(tmp_path / 'NOE.dat')
    gs.print_xy(fn, names, b[2, :], header=hdr)
    same(fn, os.path.join(GOLD, 'cfg1_iso_NOE.dat'))


def test_vecHistogram_npz_roundtrip(tmp_path):
    v = golden('cfg1_vec.npz')
    r = golden('cfg1_relax.npz')
    ref = np.load(os.path.join(GOLD, 'cfg1_vecHistogram.npz'), allow_pickle=True)
    fn = str(tmp_path / 'h.npz')
    edges = hostct.lambert_edges(72)
    np.testing.assert_array_equal(edges[0], v['edges_phi'])
    np.testing.assert_array_equal(edges[1], v['edges_cos'])
    gs.save_vecHistogram_npz(fn, list(range(2, 34)), v['hist'].astype(np.float64), edges)
    mine = np.load(fn, allow_pickle=True)
    for k in ('names', 'dataType', 'bHistogram', 'axisLabels', 'data'):
        np.testing.assert_array_equal(mine[k], ref[k])
    np.testing.assert_array_equal(mine['edges'][0], ref['edges'][0])
    np.testing.assert_array_equal(mine['edges'][1], ref['edges'][1])
    np.testing.assert_array_equal(hm.lambert_bin_vectors(mine['edges']), r['binvecs'])
    from spinrelax_amd import spectral_densities as sd
    ids, vecs, w = sd.read_vector_distribution_from_file(fn)
    np.testing.assert_array_equal(w, r['weights'])
    np.testing.assert_array_equal(vecs[3], r['binvecs'])


@pytest.mark.parametrize('tag', ['cfg1', 'cfg2', 'cfg3s'])
def test_order_search_reproduces_reference_selection(tag):
    """Feed the vectorised model-order search with the reference's own per-order results: the initial guesses
    must be bit-identical to the reference's and the accept/reject sequence must pick the same model."""
    f = golden('%s_fit.npz' % tag)
    t, y = f['t'], f['y']
    orders = [int(x) for x in f['listDoG']]
    seen = []

    def runner(nP, p0, idx):
        j = orders.index(nP)
        np.testing.assert_array_equal(p0, f['trial_p0'][idx, j, :nP])        # same initial guess, bitwise
        seen.append((nP, idx.copy()))
        ok = f['trial_quality'][idx, j, 0]
        K = nP // 2
        popt = f['trial_popt'][idx, j, :nP].copy()
        dP = f['trial_dP'][idx, j, :nP].copy()
        # hand back values whose "dP > popt" flag equals the reference's flag 1 (the golden popt are sorted by tau,
        # the comparison is permutation invariant within C / tau because errors are permuted alike)
        status = np.where(ok, 2, 0).astype(np.int32)
        popt[~ok] = 0.5
        dP[~ok] = 0.0
        return popt, dP, f['trial_chi'][idx, j], status

    best, per_order = fitCt.order_search_batch(t, y, runner, orders, 0.5)
    for i in range(y.shape[0]):
        assert best[i] >= 0
        assert orders[best[i]] == f['sel_nParams'][i]
        assert per_order[best[i]]['chiSq'][i] == f['sel_chi'][i]
        for j, res in enumerate(per_order):
            if np.isfinite(res['p0'][i, 0]):
                assert list(res['quality'][i]) == list(f['trial_quality'][i, j])
    # residues stop individually: later orders are only solved for those still searching
    assert all(len(ix) <= y.shape[0] for _, ix in seen)
    assert len(seen[-1][1]) <= len(seen[0][1])


def test_initial_guess_batch_equals_scalar():
    f = golden('cfg2_fit.npz')
    for nP in (2, 3, 4, 5, 7, 9):
        p0b, C0b, S2b = fitCt.initial_guess_batch(f['t'], f['y'], nP)
        for i in range(f['y'].shape[0]):
            p0, C0, S2 = fitCt.initial_guess(f['t'][i], f['y'][i], nP)
            np.testing.assert_array_equal(p0b[i], p0)
            assert S2b[i] == S2


def test_chunk_starts_match_reformat():
    a = np.arange(7 * 2 * 3, dtype=np.float32).reshape(7, 2, 3)
    b = 100 + np.arange(5 * 2 * 3, dtype=np.float32).reshape(5, 2, 3)
    cat, starts, R = hostct.concat_with_chunk_starts([a, b], 3)
    assert R == 3 and list(starts) == [0, 3, 7]
    v4 = np.stack([cat[s:s + 3] for s in starts])
    ref = np.concatenate([a[:6], b[:3]]).reshape(3, 3, 2, 3)
    np.testing.assert_array_equal(v4, ref)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        np.testing.assert_array_equal(hostct.reformat_vecs_by_tau([a, b], 1.0, 3.0), ref)


def test_shard_ranges_cover_everything():
    for V in (1, 7, 512, 2048, 4096, 4099):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                v0, nV = srdist.shard_range(V, r, world)
                cover += list(range(v0, v0 + nV))
            assert cover == list(range(V))
            sizes = srdist.shard_sizes(V, world)
            assert max(sizes) - min(sizes) <= 1


def test_plumed_colvar_reader_equals_reference_reader(capsys):
    """spinrelax_amd.plumedcolvario against what the reference's plumedcolvario.read_from_plumedprint returned for the
    same file (tests/golden/cfg1_colvar-qorient, oracle/gen_golden_detumble.py)."""
    from spinrelax_amd import plumedcolvario
    g = golden('cfg1_detumble.npz')
    names, data = plumedcolvario.read_from_plumedprint(os.path.join(GOLD, 'cfg1_colvar-qorient'))
    assert names == list(g['field_names'])
    assert data.dtype == np.float32 and data.flags['F_CONTIGUOUS']
    np.testing.assert_array_equal(data, g['colvar'])
    t, q = plumedcolvario.read_qorient(os.path.join(GOLD, 'cfg1_colvar-qorient'))
    np.testing.assert_array_equal(q, g['q32'])
    np.testing.assert_array_equal(t, g['colvar'][0])
    out = capsys.readouterr().out
    assert 'Found 1000 data-like lines' in out and "'time', 'q.w', 'q.x', 'q.y', 'q.z', 'rest0.bias'" in out


def test_plumed_colvar_reader_rejects_malformed(tmp_path):
    from spinrelax_amd import plumedcolvario
    bad = tmp_path / 'bad'
    bad.write_text(' 0.0 1.0 0.0 0.0 0.0\n')
    assert plumedcolvario.read_from_plumedprint(str(bad)) == -1
    bad.write_text('#! FIELDS time q.w q.x\n 0.0 1.0\n')
    assert plumedcolvario.read_from_plumedprint(str(bad)) == -1
    bad.write_text('#! FIELDS time a b\n 0.0 1.0 2.0\n')
    with pytest.raises(ValueError):
        plumedcolvario.read_qorient(str(bad))


def test_fast_pair_formatter_equals_numpy_str():
    """general_scripts._numpy_str_pairs formats the "<Ct> <dCt>" part of every _Ctint.dat line for a whole residue at once;
    the reference writes str(np.array([Ct, dCt])).strip('[]') per line (general_scripts.py:283-289).  Same text, always:
    positional and scientific rows, padding on both sides of the point, signs, zeros, exact decimals, the 1e-4 / 1e8 / ratio
    1000 switch points, and the rows it hands back to numpy (non-finite values, three-digit exponents)."""
    from spinrelax_amd import general_scripts as gs
    rng = np.random.RandomState(1)
    n = 20000
    sets = [np.stack((rng.rand(n), rng.rand(n) * 1e-2), -1), np.stack((rng.rand(n), rng.rand(n) * 1e-4), -1),
            np.stack((rng.rand(n) * 2 - 1, 10.0 ** rng.uniform(-12, 9, n) * rng.choice([-1, 1], n)), -1)]
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 0.25, 1e-4, 9.9999e-5, 1e-5, 1e8, 99999999.0, 123456789.0, 1e3, 999.9, 1000.1, 0.1,
                   0.123456785, 0.123456775, 1.23449999999e-3, 2.5, 1 / 3., 2 / 3., 1e-100, 1e100, 1e-99, np.nan, np.inf, -np.inf,
                   7.0, 1e22, 5e-324, 123.456])
    sets.append(np.array([(a, b) for a in sp for b in sp]))
    dec = rng.randint(0, 9, n)
    sets.append(np.stack([np.round(rng.rand(n), d) for d in range(9)], 0)[dec, np.arange(n)][:, None] * 10.0 ** rng.randint(-6, 6, (n, 2)))
    for y in sets:
        fast = gs._numpy_str_pairs(y)
        for j in range(len(y)):
            assert fast[j] == str(y[j]).strip('[]'), (y[j], fast[j])


def _pair_sets(n=20000):
    rng = np.random.RandomState(5)
    sets = [np.stack((rng.rand(n), rng.rand(n) * 1e-2), -1), np.stack((rng.rand(n), rng.rand(n) * 1e-4), -1),
            np.stack((rng.rand(n) * 2 - 1, 10.0 ** rng.uniform(-12, 9, n) * rng.choice([-1, 1], n)), -1)]
    sp = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 0.25, 1e-4, 9.9999e-5, 1e-5, 1e8, 99999999.0, 123456789.0, 1e3, 999.9, 1000.1, 0.1,
                   0.123456785, 0.123456775, 1.23449999999e-3, 2.5, 1 / 3., 2 / 3., 1e-98, 1e99, 7.0, 1e22, 123.456, 9.5e-5, 1e-7])
    sets.append(np.array([(a, b) for a in sp for b in sp]))
    dec = rng.randint(0, 9, n)
    sets.append(np.stack([np.round(rng.rand(n), d) for d in range(9)], 0)[dec, np.arange(n)][:, None] * 10.0 ** rng.randint(-6, 6, (n, 2)))
    return sets


def test_native_Ctint_writer_and_reader_equal_the_python_ones(tmp_path, monkeypatch):
    """csrc/sr_textio.hip (host code of the shared library): sr_text_write_sxydy_f64 writes the bytes print_sxylist writes, and
    sr_text_open_sxydy reads the doubles and legends load_sxydylist reads -- over positional and scientific rows, the switch
    points of numpy's printer, headers, quoted legends; rows it does not format (non-finite, three-digit exponents) and files
    that are not regular (ragged sets, no closing '&', junk) make it decline, and the Python path takes over."""
    from spinrelax_amd import general_scripts as gs
    assert gs._native_lib() is not None, 'libspinrelax_hip.so must load (host entry points need no GPU)'
    sets = _pair_sets()
    m = min(len(y) for y in sets)
    ylist = np.stack([y[:m] for y in sets], 0)                       # (5, m, 2)
    x = np.arange(1, m + 1) * 0.001
    legend = ['%d' % (i + 1) for i in range(len(ylist))]
    header = ['# first line', '@ title "x"']
    fa, fb = str(tmp_path / 'native.dat'), str(tmp_path / 'python.dat')
    gs.print_sxylist(fa, legend, x, ylist, header)
    with monkeypatch.context() as mp:
        mp.setattr(gs, '_native_lib', lambda: None)
        gs.print_sxylist(fb, legend, x, ylist, header)
        want = gs.load_sxydylist(fb)
    assert open(fa, 'rb').read() == open(fb, 'rb').read()
    got = gs.load_sxydylist(fa)
    assert gs._load_sxydylist_native(fa, 'legend') is not None
    assert got[0] == want[0] == legend
    for a, b in zip(got[1:], want[1:]):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    # rows the native formatter declines: the file is still written, by the Python path, byte for byte what numpy prints
    odd = ylist.copy()
    odd[2, 7] = (np.nan, 1.0)
    odd[3, 9] = (1e-120, 1.0)
    gs.print_sxylist(fa, legend, x, odd)
    with monkeypatch.context() as mp:
        mp.setattr(gs, '_native_lib', lambda: None)
        gs.print_sxylist(fb, legend, x, odd)
    assert open(fa, 'rb').read() == open(fb, 'rb').read()
    # two-column sets (no dy), and irregular files
    two = str(tmp_path / 'two.dat')
    gs.print_xylist(two, x[:50], ylist[:3, :50, 0])
    a = gs._load_sxydylist_native(two, 'legend')
    with monkeypatch.context() as mp:
        mp.setattr(gs, '_native_lib', lambda: None)
        b = gs.load_sxydylist(two)
    assert a is not None and a[0] == b[0] == [] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == [] and b[3] == []
    for text in ('@s0 legend "a"\n1 2 3\n2 3 4\n&\n@s1 legend "b"\n1 2 3\n&\n',          # ragged
                 '@s0 legend "a"\n1 2 3\n2 3 4\n',                                      # no closing '&'
                 '@s0 legend "a"\n1 2 3\n2 x 4\n&\n',                                   # junk
                 '@s0 legend "a"\n1 2 3 4\n&\n',                                        # more columns
                 '@s0 legend "a"\n&\n'):                                                # empty set
        f = str(tmp_path / 'irr.dat')
        open(f, 'w').write(text)
        assert gs._load_sxydylist_native(f, 'legend') is None


def test_native_g8_rows_equal_python_formatting(tmp_path):
    """the "%8g %8g" rows of _fittedCt.dat (fitting_Ct_functions.py:107-126) from sr_text_format_g8_pairs"""
    import ctypes
    from spinrelax_amd import general_scripts as gs
    lib = gs._native_lib()
    rng = np.random.RandomState(2)
    a = np.concatenate([10.0 ** rng.uniform(-12, 12, 5000) * rng.choice([-1, 1], 5000), [0.0, -0.0, 1.0, 100000.0, 999999.5, 1e6, 1e-5, 0.0001, 123456.7]])
    b = np.concatenate([rng.rand(5000), [1.0, 0.5, 1e-300, 1e300, 12345678.0, 0.1, 1 / 3., 2.5e-7, 5e-324]])
    buf = ctypes.create_string_buffer(32 * a.size)
    bounds = np.array([0, 17, 17, 4000, a.size], dtype=np.int64)
    offs = np.empty(bounds.size, dtype=np.int64)
    n = lib.sr_text_format_g8_pairs(a.ctypes.data, b.ctypes.data, a.size, buf, 32 * a.size, 4, bounds.ctypes.data, bounds.size, offs.ctypes.data)
    assert n > 0
    rows = ["%8g %8g\n" % (a[j], b[j]) for j in range(a.size)]
    assert buf.raw[:n].decode('ascii') == ''.join(rows)
    assert offs.tolist() == [sum(len(r) for r in rows[:k]) for k in bounds]
