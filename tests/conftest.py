import os
import sys

import numpy as np
import pytest

# before the HIP runtime starts: one hardware queue per stream of a pipeline (bench.py does the same; spinrelax_amd/pipeline.py)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=True)


@pytest.fixture(scope='session')
def gold():
    return golden


@pytest.fixture(scope='session')
def synth_cache():
    """Session cache of synthetic trajectories keyed by (cfg, nvec)."""
    from spinrelax_amd import synth
    cache = {}

    def get(cfg, nvec=None):
        key = (cfg, nvec)
        if key not in cache:
            cache[key] = synth.synth_config(cfg, nvec=nvec)
        return cache[key]
    return get


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


def files_equal_numeric(fa, fb, rtol=1e-6):
    """Same lines and same tokens; tokens that parse as numbers may differ by rtol (a printed last digit that sat on a
    rounding boundary).  Returns (ok, first difference or None)."""
    with open(fa) as a, open(fb) as b:
        la, lb = a.read().splitlines(), b.read().splitlines()
    if len(la) != len(lb):
        return False, 'line count %d vs %d' % (len(la), len(lb))
    for i, (x, y) in enumerate(zip(la, lb)):
        if x == y:
            continue
        tx, ty = x.split(), y.split()
        if len(tx) != len(ty):
            return False, 'line %d: %r vs %r' % (i + 1, x, y)
        for u, v in zip(tx, ty):
            if u == v:
                continue
            try:
                fu, fv = float(u), float(v)
            except ValueError:
                return False, 'line %d: %r vs %r' % (i + 1, x, y)
            if abs(fu - fv) > rtol * max(abs(fu), abs(fv)):
                return False, 'line %d: %r vs %r' % (i + 1, x, y)
    return True, None


def committed_tally(section, key, measured):
    """Counts that are part of the parity contract (tests/golden/fit_trial_tallies.json, measured on MI355X and committed).
    Returns the committed entry for (section, key); with SR_DUMP_TALLIES=<dir> the measured one is also written there
    (how the committed file is refreshed after a deliberate change: scripts/collect_tallies.py merges the dumps)."""
    import json
    dump = os.environ.get('SR_DUMP_TALLIES')
    if dump:
        os.makedirs(dump, exist_ok=True)
        with open(os.path.join(dump, 'tally__%s__%s.json' % (section, key)), 'w') as fp:
            json.dump(measured, fp, indent=1)
    try:
        with open(os.path.join(GOLD, 'fit_trial_tallies.json')) as fp:
            return json.load(fp)[section][key]
    except (OSError, KeyError):
        if dump:                      # first collection of a new entry: nothing committed to hold it to yet
            return measured
        raise


def ct_f32_transform(F, ct_fft=3):
    """True when kernel 1 runs FLOAT32 transforms for this chunk length (k_ct_rfft32, sr_ct32.hip): the production dispatch
    (ct_fft = 3) for 4096 < F + L <= 8192, ct_fft = 4 for every 1024 < F + L <= 8192.  Its bars: C(t) 1e-7 relative; the replicate means p_r it feeds into dC(t)
    5e-8 absolute (measured 3e-8: the rounding of a float32 transform does not average down with the chunk length the way the
    direct kernel's per-product rounding does), i.e. |d dC(t)| <= 5e-8 / (sqrt(R) - 1)."""
    return (4096 if ct_fft == 3 else 1024) < F + F // 2 <= 8192 and ct_fft in (3, 4)


def dct_close_f32_transform(dCt, ref, R, Ct=None):
    """Ct: the C(t) values, for series that are not unit vectors (C(t) then is not bounded by 1 and the absolute bar scales
    with it, per vector)"""
    import numpy as np
    scale = 1.0 if Ct is None else np.maximum(1.0, np.max(np.abs(Ct), axis=0))[None, :]
    return bool(np.all(np.abs(dCt - ref) <= np.maximum(1e-6 * np.abs(ref), scale * 5e-8 / (np.sqrt(R) - 1.0))))
