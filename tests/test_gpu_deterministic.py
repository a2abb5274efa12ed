"""ARCN_DETERMINISTIC=1: two runs of the same NGP training give bit-identical parameters (the reference's CPU path is deterministic;
round 2 reported +-1.3 dB between identical runs at 10k iterations, from the order of float additions in the hash-grid scatter)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _train(env_extra, steps=200):
    env = {k: v for k, v in os.environ.items() if not k.startswith('ARCN_')}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'det_train.py'), str(steps)], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])


def test_two_deterministic_trainings_are_bit_identical():
    """200 steps of the full NGP configuration (L16 T2^19, 128^3 occupancy refresh applied on its own stream, prefetch two batches ahead,
    density noise): same sha256 of the 12.2 M parameters in two processes; no scatter bin overflowed (that path is order-dependent);
    the run trains as well as the default mode."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    a = _train({'ARCN_DETERMINISTIC': '1'})
    b = _train({'ARCN_DETERMINISTIC': '1'})
    assert a['deterministic'] and b['deterministic'] and not a['scatter_overflowed'] and not b['scatter_overflowed']
    assert a['sha256'] == b['sha256'] and a['loss'] == b['loss']
    ref = _train({})
    assert not ref['deterministic']
    assert abs(a['loss'] - ref['loss']) <= 0.25 * ref['loss'] + 1e-3, (a['loss'], ref['loss'])
    print('deterministic %.3f ms/step, default %.3f ms/step' % (a['ms_per_step'], ref['ms_per_step']))
