"""`tinycudann`-shaped shim: the two entry points ArcNerf uses backed by the HIP kernels.

    tcnn.Encoding(n_input_dims, encoding_config={'otype': 'HashGrid' | 'SphericalHarmonics', ...})
    tcnn.Network(n_input_dims, n_output_dims, network_config={'otype': 'FullyFusedMLP', ...})

Both are nn.Modules whose parameters appear in model.named_parameters() (`params`, like tcnn).  Numerics follow the
reference's TORCH back-ends in fp32 (the parity contract of this port), not tcnn's fp16 — outputs are fp32 tensors, the
callers' `.type(dtype)` casts still work.  HashGrid expects inputs normalised to [0,1]^3 like tcnn.
"""
import math

import torch
import torch.nn as nn

from arcnerf_amd import _native as N
from arcnerf_amd.ops.autograd import FusedMlpFn, hashgrid_encode
from arcnerf_amd.ops import functional as F
from arcnerf_amd.pipeline import hashgrid_level_table

_ACT = {'ReLU': 'relu', 'None': None, 'Sigmoid': 'sigmoid', 'Exponential': 'truncexp', 'Softplus': 'softplus'}


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=None, seed=1337):
        super().__init__()
        assert n_input_dims == 3
        self.otype = encoding_config['otype']
        if self.otype == 'HashGrid':
            L = int(encoding_config.get('n_levels', 16))
            nf = int(encoding_config.get('n_features_per_level', 2))
            T = int(encoding_config.get('log2_hashmap_size', 19))
            base = int(encoding_config.get('base_resolution', 16))
            pls = float(encoding_config.get('per_level_scale', 2.0))
            max_res = base * pls ** (L - 1)
            res, offs = hashgrid_level_table(L, T, base, int(round(max_res)))
            self.desc = N.make_hashgrid_desc(res, offs, nf, [0.0] * 3, [1.0] * 3)
            g = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter((torch.rand(offs[-1] * nf, generator=g) * 2e-4 - 1e-4))
            self.n_output_dims = L * nf
            self._ws = None
        elif self.otype == 'SphericalHarmonics':
            self.degree = int(encoding_config['degree'])
            self.n_output_dims = self.degree ** 2
            self.params = nn.Parameter(torch.zeros(0))
        else:
            raise NotImplementedError('encoding {} is not provided by the HIP shim'.format(self.otype))

    def forward(self, x):
        if self.otype == 'HashGrid':
            return hashgrid_encode(x, self.params.view(-1, self.desc.n_feat), self.desc, True)
        # tcnn maps [0,1] -> [-1,1] internally; the reference's torch branch evaluates the polynomials on the [0,1] value
        # (sh_encoder.py:116,140-185).  The parity target is the torch branch: undo the caller's (d+1)/2 and re-apply it
        # inside the kernel, i.e. evaluate on exactly the value the torch branch sees.
        return F.sh_fwd((x * 2.0 - 1.0).contiguous(), self.degree, False)


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        assert network_config.get('otype', 'FullyFusedMLP') in ('FullyFusedMLP', 'CutlassMLP')
        W = int(network_config['n_neurons'])
        D = int(network_config['n_hidden_layers'])
        self.dims = [int(n_input_dims)] + [W] * D + [int(n_output_dims)]
        self.desc = N.make_mlp_desc(self.dims, _ACT[network_config.get('activation', 'ReLU')],
                                    _ACT[network_config.get('output_activation', 'None')], has_bias=False)
        g = torch.Generator().manual_seed(seed)
        ws = []
        for i in range(len(self.dims) - 1):  # torch.nn.Linear default init (the reference torch path)
            bound = 1.0 / math.sqrt(self.dims[i])
            ws.append((torch.rand(self.dims[i + 1] * self.dims[i], generator=g) * 2 - 1) * bound)
        self.params = nn.Parameter(torch.cat(ws))
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)

    def forward(self, x):
        return FusedMlpFn.apply(x, self.params, None, self.desc)
