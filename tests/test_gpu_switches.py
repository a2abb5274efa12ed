"""Every ARCN_* environment switch (README.md) at its NON-default value: the code behind it must still give the default path's results.
The switches are read once per process, so each group runs tests/switch_smoke.py in a subprocess; groups combine switches that act on
different kernels.  (ARCN_GRAD_SEGMENTS / ARCN_GRAD_LEVEL_CUTS / ARCN_DIST_BACKEND are exercised by tests/test_gpu_distributed.py;
ARCN_GATHER_ONE_XCD / ARCN_GATHER_ONLY_XCD restrict the hash gather to a part of the chip for the counter passes of tools/pmc_gather.sh
- they change the result by design and are only run, not compared.  Round 4 removed the switches whose non-default value was a slower
variant with no other role: the round-1 gather, the non-temporal gather variants, the bit-lock consumer for two-feature tables, the
deferred dW reductions, the single-launch marcher behind an environment variable, the generic MLP kernels forced onto the NGP shapes.)"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = {
    'ngp': [
        {'ARCN_OCC_ASYNC': '0', 'ARCN_FUSED_COMPOSITE': '0', 'ARCN_EMA_ALIAS': '0'},
        {'ARCN_PREFETCH_AT': '1'},
        {'ARCN_PREFETCH_AT': '2', 'ARCN_MAIN_PRIORITY': '0'},
        # one batch in flight instead of two: the batches meet the sampler's pcg32 launches in a different order (jitter of the ray
        # starts), so the trajectory is another draw of the same training - compared through the loss it reaches, not bit by bit
        {'ARCN_PREFETCH_DEPTH': '1'},
        {'ARCN_DETERMINISTIC': '1'},     # the order-independent fixed-point scatter: same gradients as the float one
        {'ARCN_FUSE_ADAM': '0'},         # scatter and optimiser as two passes (what several ranks run) instead of the fused consumer
        {'ARCN_STEP_TAIL': '0', 'ARCN_MARCH_CULL': '0'},   # dW reductions, rest of the optimiser and the counter fill as four launches instead of one; no ray culling
        {'ARCN_MARCH_WAVES': '256'},     # the marching of the batches in flight as 256 persistent wavefronts (4 rays each here): the same samples
        {'ARCN_PREFETCH_AT': '5', 'ARCN_AUX_PRIORITY': '-1'},   # the marching chain issued behind the NEXT step's gather; the sampling stream at the high priority
    ],
    'nets': [
        {'ARCN_GEMM_SPLIT': '0', 'ARCN_LINEAR_FUSED_RELU': '0', 'ARCN_LINEAR_SOFTPLUS': '0', 'ARCN_TONEMAP_FUSED': '0', 'ARCN_NEUS_UPSAMPLE_GRAPH': '1'},
        {'ARCN_RELU_BITS': '0', 'ARCN_SOFTPLUS_FUSED': '0'},
        {'ARCN_LINEAR_GEMM': '0'},
        {'ARCN_FIELD_CHAIN': '0', 'ARCN_SDF_CHAIN': '0', 'ARCN_RADIANCE_CHAIN': '0'},        # the layer-by-layer modules instead of the one-node fields (several chunks)
        {'ARCN_FIELD_CHAIN': '0', 'ARCN_SDF_CHAIN': '0', 'ARCN_RADIANCE_CHAIN': '0', 'ARCN_SKIP_CAT_FUSED': '0', 'ARCN_SPLIT_SCOPE': '0'},   # ... and without their share of it
    ],
    'neusngp': [
        {'ARCN_SDF_JACOBIAN': '0', 'ARCN_LINEAR_FUSED': '0', 'ARCN_PACKED_OVERFLOW_CHECK': '0'},
        {'ARCN_DETERMINISTIC': '1'},     # ... including the second-order table scatter
        {'ARCN_BKG_PRESAMPLE': '0'},     # the background's sampler queued after the foreground instead of before it
    ],
}
_default = {}


def _run(which, env_extra, tmp):
    env = {k: v for k, v in os.environ.items() if not k.startswith('ARCN_')}
    env.update(env_extra)
    path = os.path.join(tmp, which + '_' + '_'.join(sorted(env_extra)) + '.npz')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'switch_smoke.py'), which, path], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (env_extra, r.stderr[-3000:])
    return dict(np.load(path))


@pytest.mark.parametrize('which,idx', [(w, i) for w, gs in GROUPS.items() for i in range(len(gs))])
def test_switches_at_non_default_values_give_the_default_results(which, idx):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    with tempfile.TemporaryDirectory() as tmp:
        if which not in _default:
            _default[which] = _run(which, {}, tmp)
        ref, got = _default[which], _run(which, GROUPS[which][idx], tmp)
    assert set(ref) == set(got)
    if 'ARCN_PREFETCH_DEPTH' in GROUPS[which][idx] or GROUPS[which][idx].get('ARCN_PREFETCH_AT') == '5':
        # (the batches of the lead-in meet the sampler's launches in another order: another draw of the same training)
        assert abs(float(got['ngp_loss']) - float(ref['ngp_loss'])) <= 0.05 * float(ref['ngp_loss']) and float(got['ngp_moved']) > 1e-3
        return
    for k in ref:
        a, b = got[k], ref[k]
        if k == 'ngp_params':
            # Adam (eps 1e-15) turns the summation-order noise of a near-zero gradient entry into a full-size step of either sign: all but a
            # handful of the 4.5e5 parameters within 2 % of the distance travelled, none further than two steps; the renders below are exact
            far = np.abs(a - b) > 2e-2 * float(ref['ngp_moved'])
            assert far.mean() < 1e-3 and np.abs(a - b).max() <= 2.0 * float(ref['ngp_moved']), (k, far.mean(), np.abs(a - b).max())
        elif k == 'ngp_moved':
            assert float(a) > 1e-3
        elif k == 'ngp_loss':
            assert abs(float(a) - float(b)) <= 1e-3 * float(b)
        elif k.endswith('_grad'):
            assert np.abs(a - b).max() <= 2e-3 * np.abs(b).max() + 1e-9, (k, np.abs(a - b).max() / np.abs(b).max())
        else:
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3, err_msg=k)


def test_profiling_aids_run():
    """the level / XCD restrictions used by the counter passes: the kernels run (results are partial by design)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    with tempfile.TemporaryDirectory() as tmp:
        for env in ({'ARCN_GATHER_ONE_XCD': '1'}, {'ARCN_GATHER_ONLY_XCD': '3'}):
            out = _run('ngp', env, tmp)
            assert np.isfinite(out['ngp_params']).all()
