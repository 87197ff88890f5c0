import os
import sys

import numpy as np
import pytest

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
