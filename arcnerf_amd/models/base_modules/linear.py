"""DenseLayer / SirenLayer (arcnerf/models/base_modules/linear.py:11-71): nn.Linear + activation.  Plain library GEMMs
(rocBLAS/hipBLASLt through torch) — used by the wide vanilla-NeRF stacks; the small NGP nets run on the fused kernel."""
import math

import torch
import torch.nn as nn

from .activation import Sine


class DenseLayer(nn.Linear):
    def __init__(self, input_dim, out_dim, activation=None, bias=True):
        super().__init__(input_dim, out_dim, bias=bias)
        self.activation = activation if activation is not None else nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activation(super().forward(x))


class SirenLayer(nn.Linear):
    def __init__(self, input_dim, out_dim, is_first=False, bias=True):
        self.is_first, self.input_dim, self.w0, self.c = is_first, input_dim, 30, 6
        super().__init__(input_dim, out_dim, bias=bias)
        self.activation = Sine(self.w0)

    def reset_parameters(self):
        super().reset_parameters()
        with torch.no_grad():
            std = (1 / self.input_dim) if self.is_first else (math.sqrt(self.c / self.input_dim) / self.w0)
            self.weight.uniform_(-std, std)

    def forward(self, x):
        return self.activation(super().forward(x))
