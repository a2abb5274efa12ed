"""Feeds the mirror's sampling helpers (arcnerf_amd/render/ray_helper.uniform) with the uniforms a reference run drew from torch.rand,
recorded call by call by the fixture generators (tests/golden/tie_probe.py:RandTape): a perturb=True training pass of the reference can
then be reproduced sample for sample."""
import torch


class RandFeed:
    def __init__(self, draws, device):
        self.draws = [torch.as_tensor(d).to(device) for d in draws]
        self.used = 0

    def __enter__(self):
        from arcnerf_amd.render import ray_helper as RH
        self._rh, self._orig = RH, RH.uniform

        def fed(shape, dtype, device):
            assert self.used < len(self.draws), 'the mirror draws more uniforms than the reference run did'
            t = self.draws[self.used]
            assert tuple(t.shape) == tuple(shape), ('draw', self.used, tuple(t.shape), tuple(shape))
            self.used += 1
            return t.to(dtype=dtype, device=device).clone()
        RH.uniform = fed
        return self

    def __exit__(self, exc_type, *a):
        self._rh.uniform = self._orig
        if exc_type is None:
            assert self.used == len(self.draws), 'the mirror drew {} of the reference run\'s {} uniform tensors'.format(self.used, len(self.draws))
