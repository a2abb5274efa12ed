"""TruncExp activation module (arcnerf/ops/trunc_exp.py:40-60) on the HIP elementwise kernel."""
import torch.nn as nn

from .autograd import TruncExpFn


class TruncExp(nn.Module):
    def __init__(self, clip=15.0):
        super().__init__()
        assert float(clip) == 15.0, 'the kernel implements the reference default clip of 15'
        self.clip = clip

    def forward(self, x):
        return TruncExpFn.apply(x)
