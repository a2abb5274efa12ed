"""The measured alternatives of the step's schedule that are still selectable - NgpPipeline's keyword switches (no environment variables:
round 6 removed the ARCN_* switchboard of the package; what remains in the environment is the library's ARCN_DETERMINISTIC, the profiling
aids ARCN_GATHER_ONE_XCD / ARCN_GATHER_ONLY_XCD, ARCN_POISON_OUTPUTS of the test suite and the launcher's ARCN_DIST_* variables) - and the
library's deterministic mode, each at its NON-default value: the code behind it must still give the default path's results.  The library
reads its variables once per process, so every case runs tests/switch_smoke.py in a subprocess (keyword switches as a JSON argument).
(ARCN_GRAD_SYNC / ARCN_GRAD_LEVEL_CUTS / ARCN_DIST_BACKEND are bench.py's and are exercised by tests/test_gpu_distributed.py;
ARCN_GATHER_ONE_XCD / ARCN_GATHER_ONLY_XCD restrict the hash gather to a part of the chip for the counter passes of tools/pmc_gather.sh
- they change the result by design and are only run, not compared.)"""
import json
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
        {'kw': {'occ_async': False, 'fused_composite': False}},
        {'kw': {'prefetch_at': 1}},
        {'kw': {'prefetch_at': 2}},
        # one batch in flight instead of two: the batches meet the sampler's pcg32 launches in a different order (jitter of the ray
        # starts), so the trajectory is another draw of the same training - compared through the loss it reaches, not bit by bit
        {'kw': {'prefetch_depth': 1}},
        {'env': {'ARCN_DETERMINISTIC': '1'}},     # the order-independent fixed-point scatter: same gradients as the float one
        {'kw': {'fuse_adam': False}},             # scatter and optimiser as two passes (what several ranks run) instead of the fused consumer
        {'kw': {'step_tail': False, 'march_cull': False}},   # dW reductions, rest of the optimiser and the counter fill as four launches instead of one; no ray culling
        {'kw': {'march_waves': 256}},             # the marching of the batches in flight as 256 persistent wavefronts (4 rays each here): the same samples
        {'kw': {'planned_scatter': True}},        # the scatter's position-only half with the batches marched ahead (arcn_hashgrid_bwd_plan), fill pass + owners in the step
        {'kw': {'fused_nets': False}},            # the two nets' forward as two launches (arcn_mlp_fwd_lm + arcn_mlp_fwd_cat) instead of arcn_ngp_nets_fwd
        {'kw': {'prefetch_at': 5, 'aux_priority': -1}},   # the marching chain issued behind the NEXT step's gather; the sampling stream at the high priority
    ],
    'neusngp': [
        {'env': {'ARCN_DETERMINISTIC': '1'}},     # ... including the second-order table scatter
    ],
}
_default = {}


def _run(which, case, tmp):
    env = {k: v for k, v in os.environ.items() if not k.startswith('ARCN_')}
    env.update(case.get('env', {}))
    kw = case.get('kw', {})
    path = os.path.join(tmp, which + '_' + '_'.join(sorted(list(case.get('env', {})) + list(kw))) + '.npz')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'switch_smoke.py'), which, path, json.dumps(kw)], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (case, r.stderr[-3000:])
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
    kw = GROUPS[which][idx].get('kw', {})
    if 'prefetch_depth' in kw or kw.get('prefetch_at') == 5:
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
            out = _run('ngp', {'env': env}, tmp)
            assert np.isfinite(out['ngp_params']).all()
