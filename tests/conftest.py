import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def make_table(n_rows, F, seed, scale):
    """Same deterministic table as tests/golden/make_golden.py:make_table (numpy Generator is platform-stable)."""
    return (np.random.default_rng(seed).random((n_rows, F), dtype=np.float32) * 2.0 - 1.0).astype(np.float32) * np.float32(scale)


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
