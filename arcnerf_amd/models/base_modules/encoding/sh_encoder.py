"""SHEmbedder (arcnerf/models/base_modules/encoding/sh_encoder.py:19-185): real SH basis up to degree 5 evaluated, like
the reference's torch branch, on the (d+1)/2-mapped direction.  backend 'tcnn' and 'torch' both run the HIP kernel."""
import torch
import torch.nn as nn

from ....ops import functional as F
from ....utils.registry import ENCODER_REGISTRY


@ENCODER_REGISTRY.register()
class SHEmbedder(nn.Module):
    def __init__(self, input_dim=3, n_freqs=4, include_input=True, backend=None, dtype='torch.float16', *args, **kwargs):
        super().__init__()
        assert input_dim == 3, 'SHEmbedder should has input_dim==3...'
        assert 1 <= n_freqs <= 5, 'Should have degree 1~5 for encoding...'
        backend = backend or 'torch'
        assert backend in ('torch', 'tcnn'), 'Invalid backend used, only torch/tcnn allowed'
        self.input_dim, self.n_freqs, self.include_input, self.backend = input_dim, n_freqs, include_input, backend
        self.out_dim = n_freqs ** 2 + include_input * input_dim

    def get_output_dim(self):
        return self.out_dim

    def forward(self, xyz):
        assert xyz.dim() == 2 and xyz.shape[-1] == 3, 'Must be (B, 3) direction'
        return F.sh_fwd(xyz.detach(), self.n_freqs, self.include_input)
