"""FreqEmbedder (arcnerf/models/base_modules/encoding/freq_encoder.py:10-88): [x, sin(2^k x), cos(2^k x)]_k, one kernel.

When the INPUT requires grad (BaseGeoNet.forward_with_grad: normals = d sdf / d x with create_graph=True, needed again by the
Eikonal term and by everything downstream of the normals) the encoding is built from torch sin / cos instead, so that autograd can
differentiate the input gradient a second time; the kernel's hand-written backward is first-order only."""
import torch
import torch.nn as nn

from ....ops.autograd import FreqFn
from ....utils.registry import ENCODER_REGISTRY


@ENCODER_REGISTRY.register()
class FreqEmbedder(nn.Module):
    def __init__(self, input_dim, n_freqs, log_sampling=True, include_input=True, periodic_fns=(torch.sin, torch.cos),
                 *args, **kwargs):
        super().__init__()
        assert log_sampling and tuple(periodic_fns) == (torch.sin, torch.cos), 'the kernel implements log-sampled sin/cos'
        assert n_freqs > 0 or include_input
        self.input_dim, self.n_freqs, self.include_input = input_dim, int(n_freqs), include_input
        self.out_dim = input_dim * (1 if include_input else 0) + input_dim * 2 * self.n_freqs

    def get_output_dim(self):
        return self.out_dim

    def forward(self, x):
        assert x.shape[-1] == self.input_dim, 'Input shape should be (B, {})'.format(self.input_dim)
        if self.n_freqs == 0:
            return x
        if x.requires_grad and torch.is_grad_enabled():
            out = [x] if self.include_input else []
            for k in range(self.n_freqs):   # same order as the kernel / the reference: per frequency (sin, cos) over all dims
                out += [torch.sin(x * (2.0 ** k)), torch.cos(x * (2.0 ** k))]
            return torch.cat(out, dim=-1)
        return FreqFn.apply(x, self.n_freqs, self.include_input)
